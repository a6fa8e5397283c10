"""Where layer 1's forward DMA kernel spends its time: the same 65536 x 256 output tile grid (256 workgroups, same epilogue)
with contraction lengths K = 512, 1024, 2048 (k = 2, 4, 8 taps) separates the per-stage cost of the main loop from the fixed
part (prologue + ChannelNorm epilogue + stores).  Random H2 input; hip events on torch's current stream.
usage (GPU): python tools/bench_conv_k.py [pipe ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402


def timeit(fn, iters=30, warm=5):
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
    pipes = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
    dev = torch.device("cuda:0")
    lib = _lib.get()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(0)
    B, Lout = 64, 1024
    bias = torch.randn(256, device=dev) * 0.1
    nw, nb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    bound = torch.tensor([4.0], device=dev)
    zeros = torch.zeros(32, device=dev)
    out = {}
    lib.cpc_set_mfma_mode(3)
    for k, s, p in ((8, 4, 2), (4, 2, 1), (2, 1, 0)):
        Lin = (Lout - 1) * s + k - 2 * p
        x = torch.randn(B, Lin, 256, device=dev).clamp_(-4, 4)
        if os.environ.get("CPC_DATA") == "zero":       # DVFS check: no toggling in the operands
            x.zero_()
        elif os.environ.get("CPC_DATA") == "relu":     # what the layer sees in the step: half the elements exactly zero
            x.relu_()
        xh2 = torch.empty(B, Lin, 256, device=dev)
        lib.check(lib.cpc_h2_encode(P(x), P(xh2), B * Lin, P(bound), st))
        w = torch.randn(256, 256, k, device=dev) / (16.0 * k ** 0.5)
        if os.environ.get("CPC_DATA") == "zero":
            w.zero_()
        wq = torch.empty(256 * k * 256 + 64, device=dev)
        lib.check(lib.cpc_conv_weight_relayout_h2(P(w), P(wq), k, st))
        yh = torch.empty(B, Lout, 256, device=dev)
        xh = torch.empty(B, Lout, 256, device=dev)
        rs = torch.empty(B * Lout, device=dev)
        for pipe in pipes:
            lib.cpc_set_dma_pipeline(pipe)

            def run():
                lib.check(lib.cpc_conv_gemm_forward_h2(P(xh2), P(wq), P(bias), P(nw), P(nb), P(yh), P(xh), P(rs), P(bound),
                                                       P(bound), P(zeros), B, Lin, k, s, p, 256, st))
            out[f"k{k}_pipe{pipe}_us"] = round(1e3 * timeit(run), 1)
    lib.cpc_set_dma_pipeline(_lib.DEFAULT_DMA_PIPELINE)
    for pipe in pipes:
        t8, t2 = out[f"k8_pipe{pipe}_us"], out[f"k2_pipe{pipe}_us"]
        per32 = (t8 - t2) / 48.0               # 64 vs 16 stages of 32 k
        out[f"pipe{pipe}_us_per_32k"] = round(per32, 3)
        out[f"pipe{pipe}_fixed_us"] = round(t8 - 64 * per32, 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
