"""The transformer oracle (oracle/transformer_oracle.py) against the committed fixtures, which hold the
REFERENCE's results (cpc/transformers.py imported by oracle/make_golden_transformer.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import transformer_oracle as T
from oracle.make_golden import checksums


@pytest.mark.parametrize("case", ["transformer_ar_b2", "transformer_pred_b2", "transformer_abspos_b1"])
def test_transformer_oracle_matches_reference_fixture(case, golden_dir):
    with open(os.path.join(golden_dir, "transformer_meta.json")) as f:
        m = json.load(f)["cases"][case]
    fx = np.load(os.path.join(golden_dir, case + ".npz"))
    B, S, abspos = m["batch"], m["size_seq"], m["abspos"]
    first = 1 if abspos else 0
    p = T.make_layer_params(m["param_seed"], 256, S, abspos, prefix=f"{first}.")
    g = torch.Generator().manual_seed(m["input_seed"])
    x = torch.randn(B, S, 256, generator=g)
    dy = torch.randn(B, S, 256, generator=g)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    y = T.ar_forward(leaves, xr, 1, abspos)
    (y * dy).sum().backward()
    assert np.abs(y.detach()[:, ::8, :].numpy() - fx["y_slice"]).max() < 2e-6
    assert np.abs(xr.grad[:, ::8, :].numpy() - fx["dx_slice"]).max() < 2e-5
    # checksums = (sum, norm, cosine probe); sum and probe cancel heavily, so their tolerance scales with the norm
    def close(got, want, what):
        got, want = np.array(got), np.array(want)
        assert abs(got[1] - want[1]) <= 2e-5 * want[1], what
        assert np.abs(got[[0, 2]] - want[[0, 2]]).max() <= 2e-5 * want[1] + 1e-6, what

    close(checksums(y), fx["y_sums"], "y")
    for k, v in leaves.items():
        close(checksums(v.grad), fx["g:" + k], k)
