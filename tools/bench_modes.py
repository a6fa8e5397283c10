"""Full train-step time under each GEMM arithmetic mode (cpc_set_mfma_mode): 0 exact-f32 MFMA, 1 three bf16 pieces,
2 two fp16 pieces (conv layers).  usage: python tools/bench_modes.py [B]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib                                          # noqa: E402
from cpc_audio_amd.train import Trainer, build_criterion, build_model   # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda:0")
    lib = _lib.get()
    out = {"B": B}
    modes = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 1, 2, 0)
    for mode in modes:
        lib.check(lib.cpc_set_mfma_mode(mode))
        torch.manual_seed(0)
        model, crit = build_model().to(dev), build_criterion().to(dev)
        tr = Trainer(model, crit)
        wave = (0.1 * torch.randn(B, 1, 20480)).clamp_(-1, 1).to(dev)
        label = torch.zeros(B, dtype=torch.long, device=dev)
        torch.manual_seed(5)
        for _ in range(4):
            losses, _ = tr.step(wave, label)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            losses, _ = tr.step(wave, label)
        torch.cuda.synchronize()
        ms = 1000 * (time.perf_counter() - t0) / n
        out.setdefault(f"mode{mode}_ms", []).append(round(ms, 3))
        out[f"mode{mode}_loss"] = round(float(losses.mean()), 6)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
