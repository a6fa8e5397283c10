"""A/B timing of the train step inside ONE process (box-to-box spread is ~1 %, more than most single changes):
alternates two settings of a module-level switch over several rounds and prints ms/step for each.

    python tools/ab_step.py "ops.PIN_NEGATIVES" False True [--rounds 6 --steps 20]

The first argument is an attribute path inside cpc_audio_amd (ops.X, criterion.X ...) or ENV:NAME for an
environment variable read at call time, or CALL:cpc_set_xxx for a setter of the C ABI (an int, or a tuple of ints for several arguments); the two values are Python literals."""
import ast
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    target, va, vb = sys.argv[1], ast.literal_eval(sys.argv[2]), ast.literal_eval(sys.argv[3])
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 6
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 20
    from cpc_audio_amd.train import Trainer, build_criterion, build_model

    def setv(v):
        if target.startswith("ENV:"):
            os.environ[target[4:]] = str(v)
        elif target.startswith("CALL:"):                  # a setter of the C ABI, e.g. CALL:cpc_set_conv_tile
            from cpc_audio_amd import _lib
            lib = _lib.get()
            lib.check(getattr(lib, target[5:])(*(v if isinstance(v, tuple) else (v,))), target)
        else:
            mod, attr = target.rsplit(".", 1)
            setattr(importlib.import_module("cpc_audio_amd." + mod), attr, v)

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model, crit = build_model().to(dev), build_criterion().to(dev)
    tr = Trainer(model, crit)
    wave = (0.1 * torch.randn(64, 1, 20480)).clamp_(-1, 1).to(dev)
    label = torch.zeros(64, dtype=torch.long, device=dev)
    res = {0: [], 1: []}
    for r in range(rounds + 1):
        for k, v in enumerate((va, vb)):
            setv(v)
            for _ in range(5):
                tr.step(wave, label)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                tr.step(wave, label)
            torch.cuda.synchronize()
            if r:                                   # round 0 warms up
                res[k].append(1e3 * (time.perf_counter() - t0) / steps)
    for k, v in enumerate((va, vb)):
        xs = sorted(res[k])
        print(f"{target} = {v!r}: median {xs[len(xs) // 2]:.3f} ms/step  min {xs[0]:.3f}  max {xs[-1]:.3f}", flush=True)


if __name__ == "__main__":
    main()
