"""Pin oracle/transformer_oracle.py against the imported reference (cpc/transformers.py) and write the
fixtures tests/golden/transformer_*.npz.  TEST INFRASTRUCTURE ONLY; run in the build container:

    python oracle/make_golden_transformer.py

For each case the reference ``buildTransformerAR(d_model, 1, size_seq, abspos)`` is built, put in eval mode
(its dropout of 0.1 is hard-coded, transformers.py:93), loaded with the deterministic parameters of
transformer_oracle.make_layer_params and run forward + backward on a seeded input with a seeded output
gradient.  The oracle must agree (outputs <= 2e-6 abs, input/parameter gradients <= 1e-5 relative); the
fixture stores slices and checksums of the REFERENCE's results (data only).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_import                    # noqa: E402
from oracle import transformer_oracle as T       # noqa: E402
from oracle.make_golden import checksums         # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")

# name, batch, size_seq, abspos, param seed, input seed
CASES = [
    ("transformer_ar_b2", 2, 128, False, 5, 11),        # --arMode transformer: sequence 128
    ("transformer_pred_b2", 2, 116, False, 6, 12),      # --rnnMode transformer predictor: sequence 128 - 12
    ("transformer_abspos_b1", 1, 128, True, 7, 13),
]


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    ref_import.import_reference()
    import cpc.transformers as RT
    meta = {"torch": torch.__version__, "cases": {}}
    for name, B, S, abspos, pseed, xseed in CASES:
        first = 1 if abspos else 0
        p = T.make_layer_params(pseed, 256, S, abspos, prefix=f"{first}.")
        net = RT.buildTransformerAR(256, 1, S, abspos)
        missing = net.load_state_dict(p, strict=False)
        assert not missing.unexpected_keys, missing
        assert all(k.endswith(("Att.z", "Att.mask", ".pe")) for k in missing.missing_keys), missing   # buffers only
        net.eval()
        g = torch.Generator().manual_seed(xseed)
        x = torch.randn(B, S, 256, generator=g)
        dy = torch.randn(B, S, 256, generator=g)
        xr = x.clone().requires_grad_(True)
        yr = net(xr)
        (yr * dy).sum().backward()
        ref_grads = {k: v.grad.clone() for k, v in net.named_parameters()}

        leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        xo = x.clone().requires_grad_(True)
        yo = T.ar_forward(leaves, xo, 1, abspos)
        (yo * dy).sum().backward()
        err = (yo - yr).abs().max().item()
        assert err <= 2e-6, (name, err)
        assert rel(xo.grad, xr.grad) <= 1e-5, (name, rel(xo.grad, xr.grad))
        for k, gr in ref_grads.items():
            assert rel(leaves[k].grad, gr) <= 1e-5, (name, k, rel(leaves[k].grad, gr))
        print(f"{name}: oracle == reference (max|dy| {err:.2e}, dx rel {rel(xo.grad, xr.grad):.2e})")

        arrays = {
            "y_slice": yr.detach()[:, ::8, :].numpy().astype(np.float32),
            "dx_slice": xr.grad[:, ::8, :].numpy().astype(np.float32),
            "y_sums": np.array(checksums(yr), dtype=np.float64),
            "dx_sums": np.array(checksums(xr.grad), dtype=np.float64),
        }
        for k, gr in ref_grads.items():
            arrays["g:" + k] = np.array(checksums(gr), dtype=np.float64)
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, **arrays)
        meta["cases"][name] = {"batch": B, "size_seq": S, "abspos": abspos, "param_seed": pseed, "input_seed": xseed,
                               "bytes": os.path.getsize(path)}
    with open(os.path.join(GOLDEN_DIR, "transformer_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
