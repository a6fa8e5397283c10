"""Stand-alone timings of the criterion's forward / backward entry points on the GPU (B = 64 x 128 frames, K = 12, N = 128), for every
cpc_set_nce_fused mode, and -- mode 2 -- with parts of the scoring kernel left out (cpc_set_nce_debug): what each part costs.
usage: python tools/time_nce.py [B]        (run under rocprofv3 --kernel-trace --stats for per-kernel durations)"""
import ctypes
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
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S, K, N = 128, 12, 128
    W = S - K
    dev = torch.device("cuda:0")
    lib = _lib.get()
    torch.manual_seed(0)
    c = torch.tanh(torch.randn(B, S, 256, device=dev))
    z = torch.relu(torch.randn(B, S, 256, device=dev))
    wall = torch.randn(K * 256, 256, device=dev) / 16
    bi = torch.randint(0, B, (B * N * W,), device=dev)
    si = torch.randint(1, S, (B * N * W,), device=dev)
    ext = torch.zeros(B * W * N, dtype=torch.int32, device=dev)
    perm = torch.zeros(B * W * (N + K), dtype=torch.int32, device=dev)
    row_ptr = torch.zeros(B * S + 1, dtype=torch.int32, device=dev)
    work = torch.zeros(B * W * (N + K) + 2 * B * S + 2, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lib.check(lib.cpc_nce_prepare(P(bi), P(si), P(ext), P(perm), P(row_ptr), P(work), B, S, K, N, st), "prepare")
    sizes = (ctypes.c_long * 6)()
    lib.check(lib.cpc_nce_layout(B, S, K, N, sizes), "layout")
    saved = torch.zeros(sizes[0], device=dev)
    fscr, bscr = torch.zeros(sizes[1], device=dev), torch.zeros(sizes[2], device=dev)
    losses, acc = torch.zeros(K, device=dev), torch.zeros(K, device=dev)
    gl = torch.ones(K, device=dev)
    dc, dz, dwall = torch.zeros_like(c), torch.zeros_like(z), torch.zeros_like(wall)

    def fwd():
        lib.check(lib.cpc_nce_forward(P(c), P(z), P(wall), P(ext), P(saved), P(fscr), P(losses), P(acc), B, S, K, N, st), "fwd")

    def bwd():
        lib.check(lib.cpc_nce_backward(P(c), P(z), P(wall), P(ext), P(perm), P(row_ptr), P(saved), P(gl), P(bscr), P(dc), P(dz),
                                       P(dwall), B, S, K, N, st), "bwd")

    ref = None
    for mode, grid, dbg in ((1, 0, 0), (2, 0, 0), (3, 0, 0), (2, -1, 0), (2, 0, 1), (2, 0, 2), (2, 0, 4), (2, 0, 8), (2, 0, 15), (0, 0, 0)):
        lib.check(lib.cpc_set_nce_fused(mode), "mode")
        lib.check(lib.cpc_set_nce_grid(grid), "grid")
        lib.check(lib.cpc_set_nce_debug(dbg), "dbg")
        tf = timeit(fwd)
        tb = timeit(bwd) if dbg == 0 else float("nan")
        torch.cuda.synchronize()
        note = ""
        if dbg == 0:
            if ref is None:
                ref = (losses.clone(), dz.clone(), dc.clone())
            else:
                note = (f" dloss {float((losses - ref[0]).abs().max()):.2e} dz rel {float((dz - ref[1]).norm() / ref[1].norm()):.2e}"
                        f" dc rel {float((dc - ref[2]).norm() / ref[2].norm()):.2e}")
        print(f"mode {mode} grid {grid:2d} dbg {dbg:2d}: forward call {tf:7.1f} us   backward call {tb:7.1f} us{note}")
    lib.cpc_set_nce_fused(_lib.DEFAULT_NCE_FUSED); lib.cpc_set_nce_grid(0); lib.cpc_set_nce_debug(0)


if __name__ == "__main__":
    main()
