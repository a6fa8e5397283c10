"""Does the ORDER in which a window gathers its negatives matter?  cpc_nce_scores_forward (nce_fwd_kernel) and the full
criterion backward at B = 64 with the reference's random negative rows as drawn, and with each window's 128 rows sorted
ascending (the loss is invariant under a permutation of a window's negatives): sorted lists make all resident waves walk z
in step, so the rows they gather at any moment lie in a narrow band that fits the 4 MB L2 of an XCD instead of coming from
Infinity Cache.  usage: python tools/probe_gather_order.py [B]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402
from cpc_audio_amd.ops import candidate_destinations  # noqa: E402

lib = _lib.get()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S, K, N = 128, 12, 128
W = S - K
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
sizes = (ctypes.c_long * 6)()
lib.check(lib.cpc_nce_layout(B, S, K, N, sizes))
c = torch.tanh(torch.randn(B, S, 256, device=dev))
z = torch.relu(torch.randn(B, S, 256, device=dev))
wall = torch.randn(K * 256, 256, device=dev) / 16
saved = torch.empty(sizes[0], device=dev)
fscr = torch.empty(sizes[1], device=dev)
bscr = torch.empty(sizes[2], device=dev)
losses, acc = torch.empty(K, device=dev), torch.empty(K, device=dev)
gl = torch.ones(K, device=dev)
dc, dz, dwall = torch.empty_like(c), torch.empty_like(z), torch.empty_like(wall)
ext_rand = torch.randint(0, B * S, (B, W, N), device=dev, dtype=torch.int32)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


for name, ext in (("as drawn", ext_rand), ("sorted per window", torch.sort(ext_rand, dim=2).values.contiguous())):
    perm, row_ptr = candidate_destinations(ext, B, S, K)

    def fwd():
        lib.check(lib.cpc_nce_forward(P(c), P(z), P(wall), P(ext), P(saved), P(fscr), P(losses), P(acc), B, S, K, N, st))

    def bwd():
        lib.check(lib.cpc_nce_backward(P(c), P(z), P(wall), P(ext), P(perm), P(row_ptr), P(saved), P(gl), P(bscr), P(dc), P(dz),
                                       P(dwall), B, S, K, N, st))

    tf = timeit(fwd)
    tb = timeit(bwd)
    print(f"{name:20s}: criterion forward {tf:7.1f} us   backward {tb:7.1f} us   loss0 {losses[0].item():.5f}", flush=True)
