"""Fixture for the prediction networks of the reference's other ``--rnnMode`` choices (cpc/criterion/criterion.py:63-81:
RNN, LSTM, ffd, conv4, conv8, conv12) -- tests/golden/predictors.npz + predictors_meta.json.  TEST INFRASTRUCTURE ONLY; run in
the build container:

    python oracle/make_golden_predictors.py

For each mode the reference ``PredictionNetwork(3, 256, 256, rnnMode)`` is built, its parameters are overwritten by seeded
values (named by the reference's own state-dict keys, which the fixture records with their shapes), and its forward
(criterion.py:97-118: per head mean_d(prediction * candidates)) is run on a seeded context and seeded candidates together with
the gradient of the sum of its outputs w.r.t. the context.  The fixture stores the REFERENCE's outputs (data only);
tests/test_abi_symbols.py loads the same parameters into cpc_audio_amd.criterion.PredictionNetwork by those keys and must
reproduce them.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_import                    # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")
MODES = ["RNN", "LSTM", "ffd", "conv4", "conv8", "conv12"]
K, B, W, NEG, C = 3, 2, 6, 4, 256


def seeded_state(shapes, seed):
    """Deterministic parameters by key order: N(0, 1) scaled to 0.05 (biases included: a zero bias would hide a missing one)."""
    g = torch.Generator().manual_seed(seed)
    return {k: 0.05 * torch.randn(*s, generator=g) for k, s in shapes.items()}


def inputs(seed):
    g = torch.Generator().manual_seed(seed)
    c = torch.randn(B, W, C, generator=g)
    cand = [torch.randn(B, NEG, W, C, generator=g) for _ in range(K)]
    return c, cand


def main():
    _, ref_criterion = ref_import.import_reference()
    import cpc.criterion.criterion as RC
    arrays, meta = {}, {"torch": torch.__version__, "heads": K, "batch": B, "window": W, "candidates": NEG, "modes": {}}
    for i, mode in enumerate(MODES):
        net = RC.PredictionNetwork(K, C, C, rnnMode=mode, dropout=False, sizeInputSeq=W)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict(seeded_state(shapes, 100 + i), strict=True)
        c, cand = inputs(200 + i)
        cr = c.clone().requires_grad_(True)
        out = net(cr, cand)
        sum(o.sum() for o in out).backward()
        arrays[f"{mode}:out"] = torch.stack(out).detach().numpy().astype(np.float32)        # (K, B, NEG, W)
        arrays[f"{mode}:dc"] = cr.grad.numpy().astype(np.float32)
        meta["modes"][mode] = {"param_seed": 100 + i, "input_seed": 200 + i, "keys": {k: list(s) for k, s in shapes.items()}}
        print(mode, len(shapes), "parameters,", "out", tuple(arrays[f"{mode}:out"].shape))
    np.savez_compressed(os.path.join(GOLDEN_DIR, "predictors.npz"), **arrays)
    with open(os.path.join(GOLDEN_DIR, "predictors_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
