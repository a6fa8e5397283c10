"""Time individual C-ABI entry points on the GPU (hip events on torch's current stream).
usage: python tools/bench_kernels.py [B]"""
import ctypes
import json
import os
import sys

import torch


def _lib_default_mode():
    from cpc_audio_amd._lib import DEFAULT_MFMA_MODE
    return DEFAULT_MFMA_MODE

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402
from oracle import cpc_oracle as O       # noqa: E402


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
    L = 20480
    dev = torch.device("cuda:0")
    lib = _lib.get()
    p = O.make_params(seed=0)
    names = [f"gEncoder.{n}{i}.{w}" for i in range(5)
             for n, w in (("conv", "weight"), ("conv", "bias"), ("batchNorm", "weight"), ("batchNorm", "bias"))]
    plist = [p[n].contiguous().to(dev) for n in names]
    wave = O.make_waveform(B, L, seed=5).to(dev)
    sizes = (ctypes.c_long * 22)()
    lib.check(lib.cpc_encoder_layout(B, L, sizes))
    Ls = [sizes[3 + i] for i in range(5)]
    saved = torch.empty(sizes[0], device=dev)
    fscr = torch.empty(max(1, sizes[1]), device=dev)
    bscr = torch.empty(sizes[2], device=dev)
    z = torch.empty(B, Ls[4], 256, device=dev)
    dz = torch.randn(B, Ls[4], 256, device=dev)
    grads = [torch.empty_like(t) for t in plist]
    parr = (ctypes.c_void_p * 20)(*[P(t) for t in plist])
    garr = (ctypes.c_void_p * 20)(*[P(t) for t in grads])
    st = torch.cuda.current_stream().cuda_stream
    out = {"B": B}

    def fwd():
        lib.check(lib.cpc_encoder_forward(P(wave), parr, P(saved), P(fscr), P(z), B, L, st))

    def bwd():
        lib.check(lib.cpc_encoder_backward(P(wave), parr, P(saved), P(z), P(dz), P(bscr), garr, B, L, st))

    out["encoder_fwd_ms"] = timeit(fwd)
    out["encoder_bwd_ms"] = timeit(bwd)
    lib.cpc_set_mfma_mode(2)          # the per-layer calls below read the saved activations as fp32 (mode 3 keeps y0 as pieces)
    fwd()
    # conv0 alone (the HBM-bound layer)
    y0 = saved[sizes[8]: sizes[8] + B * Ls[0] * 256]
    mean0 = saved[sizes[21]: sizes[21] + B * Ls[0]]
    rstd0 = saved[sizes[16]: sizes[16] + B * Ls[0]]

    def c0():
        lib.check(lib.cpc_conv0_forward(P(wave), P(plist[0]), P(plist[1]), P(plist[2]), P(plist[3]), P(y0),
                                        P(mean0), P(rstd0), B, L, st))
    t = timeit(c0, iters=20)
    out["conv0_fwd_ms"] = t
    out["conv0_fwd_GBps"] = (B * Ls[0] * 256 * 4 + B * L * 4) / t / 1e6
    macs = {1: 536870912, 2: 134217728, 3: 67108864, 4: 33554432}
    geom = {1: (8, 4, 2), 2: (4, 2, 1), 3: (4, 2, 1), 4: (4, 2, 1)}
    for bm, mode in ((0, 1), (0, 0), (32, 1), (128, 1), (64, 1)):
        lib.cpc_set_conv_tile(bm)
        lib.cpc_set_mfma_mode(mode)
        for i in (1, 2, 3, 4):
            k, s, pd = geom[i]
            xin = saved[sizes[8 + i - 1]: sizes[8 + i - 1] + B * Ls[i - 1] * 256]
            yo = z if i == 4 else saved[sizes[8 + i]: sizes[8 + i] + B * Ls[i] * 256]
            xh = saved[sizes[11 + i]: sizes[11 + i] + B * Ls[i] * 256]
            rs = saved[sizes[16 + i]: sizes[16 + i] + B * Ls[i]]
            wp = torch.empty(256 * k * 256 * 3 // 2, device=dev)

            def cf():
                lib.check(lib.cpc_conv_layer_forward(P(xin), P(plist[4 * i]), P(plist[4 * i + 1]), P(plist[4 * i + 2]),
                                                     P(plist[4 * i + 3]), P(wp), P(yo), P(xh), P(rs), B, Ls[i - 1], k, s, pd, st))
            t = timeit(cf)
            out[f"conv{i}_fwd_bm{bm}_mode{mode}_ms"] = round(t, 4)
            out[f"conv{i}_fwd_bm{bm}_mode{mode}_TFLOPs"] = round(2 * macs[i] * B / t / 1e9, 1)
    lib.cpc_set_conv_tile(0)
    lib.cpc_set_mfma_mode(_lib_default_mode())
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
