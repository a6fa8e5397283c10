"""A few train steps of BASELINE config 4 (transformer AR + 12 transformer predictors, B = 64) for profiling:
    rocprofv3 --kernel-trace --stats -d /tmp/p4 -o res -- python tools/run_config4.py [steps] [B]
prints ms/step (wall clock, after warm-up)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd.train import Trainer, build_criterion, build_model      # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    dev = torch.device("cuda:0")
    if os.environ.get("CPC_GEMM_DMA"):            # A/B: cpc_set_gemm_dma
        from cpc_audio_amd import _lib
        _lib.get().check(_lib.get().cpc_set_gemm_dma(int(os.environ["CPC_GEMM_DMA"])), "set_gemm_dma")
    for spec in filter(None, os.environ.get("CPC_CALLS", "").split(";")):      # A/B: "cpc_set_nce_fused=1;cpc_set_gemm_dma=0"
        from cpc_audio_amd import _lib
        name, _, vals = spec.partition("=")
        _lib.get().check(getattr(_lib.get(), name)(*[int(v) for v in vals.split(",")]), name)
    torch.manual_seed(0)
    model = build_model(arMode="transformer").to(dev)
    crit = build_criterion(rnnMode="transformer").to(dev)
    tr = Trainer(model, crit)
    wave = (0.1 * torch.randn(B, 1, 20480)).clamp_(-1, 1).to(dev)
    label = torch.zeros(B, dtype=torch.long, device=dev)
    for _ in range(3):
        tr.step(wave, label)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, _ = tr.step(wave, label)
    torch.cuda.synchronize()
    print(f"config 4, B = {B}: {1e3 * (time.perf_counter() - t0) / steps:.3f} ms/step, loss {float(loss.mean()):.4f}")


if __name__ == "__main__":
    main()
