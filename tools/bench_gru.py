"""Time the two-layer GRU entry points (persistent single launch vs one launch per step) on the GPU.
usage: python tools/bench_gru.py [B] [S]"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402


def _lib_default_gru_mode():
    from cpc_audio_amd._lib import DEFAULT_GRU_MODE
    return DEFAULT_GRU_MODE


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
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    dev = torch.device("cuda:0")
    lib = _lib.get()
    torch.manual_seed(0)
    shapes = [(768, 256), (768, 256), (768,), (768,)] * 2
    plist = [(torch.randn(s, device=dev) / 16.0) for s in shapes]
    x = torch.randn(B, S, 256, device=dev)
    dy = torch.randn(B, S, 256, device=dev)
    sizes = (ctypes.c_long * 3)()
    lib.check(lib.cpc_gru_layout(B, S, 2, sizes))
    saved = torch.empty(sizes[0], device=dev)
    fscr = torch.empty(sizes[1], device=dev)
    bscr = torch.empty(sizes[2], device=dev)
    y = torch.empty(B, S, 256, device=dev)
    hN = torch.empty(2, B, 256, device=dev)
    dx = torch.empty(B, S, 256, device=dev)
    grads = [torch.empty_like(t) for t in plist]
    parr = (ctypes.c_void_p * 8)(*[P(t) for t in plist])
    garr = (ctypes.c_void_p * 8)(*[P(t) for t in grads])
    st = torch.cuda.current_stream().cuda_stream

    def fwd():
        lib.check(lib.cpc_gru_forward(P(x), None, parr, P(saved), P(fscr), P(y), P(hN), B, S, 2, st))

    def bwd():
        lib.check(lib.cpc_gru_backward(P(x), None, parr, P(saved), P(y), P(dy), P(bscr), P(dx), garr, B, S, 2, st))

    out = {"B": B, "S": S}
    ref = None
    for mode, pack in ((0, 1), (1, 1), (2, 1), (2, 0), (1, 0)):
        lib.check(lib.cpc_set_gru_mode(mode))
        lib.check(lib.cpc_set_gru_xcd_pack(pack))
        mode = f"{mode}_pack{pack}"
        out[f"fwd_ms_mode{mode}"] = round(timeit(fwd), 4)
        out[f"bwd_ms_mode{mode}"] = round(timeit(bwd), 4)
        cur = [y.clone(), dx.clone()] + [g.clone() for g in grads]
        if ref is None:
            ref = cur
        else:
            out["bit_identical"] = all(torch.equal(a, b) for a, b in zip(ref, cur))
            out["finite"] = all(bool(torch.isfinite(a).all()) for a in cur)
    lib.cpc_set_gru_mode(_lib_default_gru_mode())
    lib.cpc_set_gru_xcd_pack(0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
