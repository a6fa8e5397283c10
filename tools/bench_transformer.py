"""Time the HIP transformer layer (forward, backward) and a whole BASELINE config-4 train step on the GPU.
usage: python tools/bench_transformer.py [B]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd.train import Trainer, build_criterion, build_model      # noqa: E402
from cpc_audio_amd.transformers import buildTransformerAR                  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    out = {"B": B}
    for S in (128, 116):
        net = buildTransformerAR(256, 1, S, False, dropout=0.0).to(dev)
        x = torch.randn(B, S, 256, device=dev, requires_grad=True)
        dy = torch.randn(B, S, 256, device=dev)
        with torch.no_grad():
            out[f"layer_fwd_ms_S{S}"] = round(timeit(lambda: net(x)), 4)

        def fb():
            net.zero_grad(set_to_none=True)
            x.grad = None
            (net(x) * dy).sum().backward()
        out[f"layer_fwd_bwd_ms_S{S}"] = round(timeit(fb), 4)
    for name, ar, pred in (("gru_linear", "GRU", "linear"), ("transformer_ar_linear", "transformer", "linear"),
                           ("transformer_ar_transformer_pred", "transformer", "transformer")):
        model = build_model(arMode=ar, transformerDropout=0.0).to(dev)
        crit = build_criterion(rnnMode=pred, transformerDropout=0.0).to(dev)
        tr = Trainer(model, crit)
        wave = (0.1 * torch.randn(B, 1, 20480)).clamp_(-1, 1).to(dev)
        label = torch.zeros(B, dtype=torch.long, device=dev)
        for _ in range(3):
            tr.step(wave, label)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            tr.step(wave, label)
        torch.cuda.synchronize()
        ms = 1000 * (time.perf_counter() - t0) / n
        out[f"step_ms_{name}"] = round(ms, 3)
        out[f"audio_s_per_s_{name}"] = round(B * 1.28 / (ms * 1e-3), 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
