"""In-process A/B of the persistent recurrence under a C-ABI setter (default: cpc_set_gru_xcd_pack 0 / 1), forward and
backward timed separately with hip events, settings alternated round by round.
usage: python tools/ab_gru.py [B] [setter] [a] [b]"""
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
setter = sys.argv[2] if len(sys.argv) > 2 else "cpc_set_gru_xcd_pack"
va, vb = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 1)
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
saved, fscr, bscr = (torch.empty(sizes[i], device=dev) for i in range(3))
y, hN, dx = torch.empty(B, S, 256, device=dev), torch.empty(2, B, 256, device=dev), torch.empty(B, S, 256, device=dev)
grads = [torch.empty_like(t) for t in plist]
parr = (ctypes.c_void_p * 8)(*[P(t) for t in plist])
garr = (ctypes.c_void_p * 8)(*[P(t) for t in grads])
st = torch.cuda.current_stream().cuda_stream


def fwd():
    lib.check(lib.cpc_gru_forward(P(x), None, parr, P(saved), P(fscr), P(y), P(hN), B, S, 2, st))


def bwd():
    lib.check(lib.cpc_gru_backward(P(x), None, parr, P(saved), P(y), P(dy), P(bscr), P(dx), garr, B, S, 2, st))


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if setter == "pacing":          # sweep (first_fwd, first_bwd); -1 = self-steering
    import itertools
    lib.check(lib.cpc_set_gru_xcd_pack(0))
    cfgs = [(0, 0), (-1, -1), (20, 20), (24, 24), (28, 28), (-1, -1), (32, 32)]
    out = {c: ([], []) for c in cfgs}
    for r in range(6):
        for c in (cfgs if r % 2 == 0 else cfgs[::-1]):
            lib.check(lib.cpc_set_gru_poll_pacing(*c))
            out[c][0].append(timeit(fwd))
            out[c][1].append(timeit(bwd))
    for c in cfgs:
        print(f"pacing {c}: fwd {statistics.median(out[c][0]):.1f} us  bwd {statistics.median(out[c][1]):.1f} us")
    sys.exit(0)
res = {(v, k): [] for v in (va, vb) for k in "fb"}
for r in range(12):
    for v in ((va, vb) if r % 2 == 0 else (vb, va)):
        lib.check(getattr(lib, setter)(v))
        res[(v, "f")].append(timeit(fwd))
        res[(v, "b")].append(timeit(bwd))
for k in "fb":
    a, b = res[(va, k)], res[(vb, k)]
    print(f"{setter} {'fwd' if k == 'f' else 'bwd'}: {va}: {statistics.median(a):.1f} us (min {min(a):.1f})   "
          f"{vb}: {statistics.median(b):.1f} us (min {min(b):.1f})")
