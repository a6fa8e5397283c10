"""Forward conv kernels of layers 1 and 2 in isolation, register-staged (mode 2, conv_fwd_kernel) vs DMA (mode 3,
conv_fwd_dma_kernel) on the activations conv0 produces for white noise; hip events on torch's current stream.
usage (GPU): python tools/bench_conv_fwd.py [B=64]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402


def timeit(fn, iters=20, warm=3):
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
    lib = _lib.get()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(0)
    L = 20480
    wave = (0.1 * torch.randn(B, L, device=dev)).clamp_(-1, 1)
    w0 = torch.randn(256, 10, device=dev) * 0.3
    bias = torch.randn(256, device=dev) * 0.1
    nw, nb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    bound = (15.968719 * nw.abs().max() + nb.abs().max()).view(1).clone()
    zeros = torch.zeros(32, device=dev)
    out = {"B": B}
    L0 = 4096
    y0 = torch.empty(B, L0, 256, device=dev)
    y0h = torch.empty(B, L0, 256, device=dev)
    m0, r0 = torch.empty(B * L0, device=dev), torch.empty(B * L0, device=dev)

    def c0(h2):
        lib.check(lib.cpc_conv0_forward_h2(P(wave), P(w0), P(bias), P(nw), P(nb), P(y0h if h2 else y0), P(m0), P(r0),
                                           P(bound) if h2 else None, B, L, st))
    out["conv0_fp32_ms"] = round(timeit(lambda: c0(False)), 4)
    out["conv0_h2_ms"] = round(timeit(lambda: c0(True)), 4)
    geom = {1: (8, 4, 2, 4096, 536870912), 2: (4, 2, 1, 1024, 134217728)}
    xin, xin_h = y0, y0h
    for layer in (1, 2):
        k, s, p, Lin, macs = geom[layer]
        Lout = Lin // s
        w = torch.randn(256, 256, k, device=dev) / (16.0 * k ** 0.5)
        wp = torch.empty(256 * k * 256 * 3 // 2, device=dev)
        wq = torch.empty(256 * k * 256 + 64, device=dev)
        y = torch.empty(B, Lout, 256, device=dev)
        yh = torch.empty(B, Lout, 256, device=dev)
        xh = torch.empty(B, Lout, 256, device=dev)
        rs = torch.empty(B * Lout, device=dev)
        lib.cpc_set_mfma_mode(2)
        lib.check(lib.cpc_conv_weight_relayout(P(w), P(wp), k, st))
        lib.check(lib.cpc_conv_weight_relayout_h2(P(w), P(wq), k, st))
        flop = 2.0 * macs * B

        def old():
            lib.check(lib.cpc_conv_gemm_forward(P(xin), P(wp), P(bias), P(nw), P(nb), P(y), P(xh), P(rs), P(bound), B, Lin, k, s, p, st))
        t = timeit(old)
        out[f"conv{layer}_regstaged_ms"] = round(t, 4)
        out[f"conv{layer}_regstaged_TF"] = round(flop / t / 1e9, 1)
        for pipe in (0, 1):
            lib.cpc_set_dma_pipeline(pipe)
            for bm in (256, 128):
                for rot in (0, 3, 5, 7, 13):
                    lib.cpc_set_dma_rotation(rot)

                    def new():
                        lib.check(lib.cpc_conv_gemm_forward_h2(P(xin_h), P(wq), P(bias), P(nw), P(nb), P(yh), P(xh), P(rs), P(bound),
                                                               P(bound), P(zeros), B, Lin, k, s, p, bm, st))
                    t = timeit(new)
                    out[f"conv{layer}_dma_pipe{pipe}_bm{bm}_rot{rot}"] = [round(t, 4), round(flop / t / 1e9, 1)]
        lib.cpc_set_dma_rotation(5)
        lib.cpc_set_dma_pipeline(_lib.DEFAULT_DMA_PIPELINE)
        # agreement of the two kernels on the same data
        old()
        xh_old = xh.clone()
        lib.check(lib.cpc_conv_gemm_forward_h2(P(xin_h), P(wq), P(bias), P(nw), P(nb), P(yh), P(xh), P(rs), P(bound),
                                               P(bound), P(zeros), B, Lin, k, s, p, 0, st))
        torch.cuda.synchronize()
        out[f"conv{layer}_max_abs_xhat_diff"] = float((xh - xh_old).abs().max())
        xin, xin_h = y, yh
    lib.cpc_set_mfma_mode(_lib.DEFAULT_MFMA_MODE)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
