"""A/B of cpc_set_gru_poll_plain masks on the stand-alone recurrence (persistent kernels only, hip events), alternated.
usage: python tools/ab_gru_plain.py [B]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402
from tools.bench_gru import timeit        # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = 128
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

    ref = None
    for rnd in range(3):
        for mask, loc in ((0, 0), (15, 0), (15, 1), (15, 5), (15, 15), (5, 5)):
            lib.check(lib.cpc_set_gru_poll_plain(mask))
            lib.check(lib.cpc_set_gru_xcd_local(loc))
            f, b = timeit(fwd, 30), timeit(bwd, 30)
            cur = [y.clone(), dx.clone()] + [g.clone() for g in grads]
            same = True if ref is None else all(torch.equal(a, c) for a, c in zip(ref, cur))
            ref = ref or cur
            flags = lib.cpc_device_error_flags(1)
            print(f"round {rnd} plain {mask:2d} local {loc}: forward call {f * 1e3:7.1f} us  backward call {b * 1e3:7.1f} us  same bits {same}  flags {flags}", flush=True)
    lib.cpc_set_gru_poll_plain(_lib.DEFAULT_GRU_POLL_PLAIN)
    lib.cpc_set_gru_xcd_local(_lib.DEFAULT_GRU_XCD_LOCAL)


if __name__ == "__main__":
    main()
