"""What slows the persistent recurrence down when other kernels share the chip?  Times cpc_gru_forward / cpc_gru_backward
(B = 64, S = 128) alone and beside a side stream that keeps one kind of load running: an HBM copy, an L2-resident copy, an
fp16 matmul (matrix pipes + power), a random 1 KB-row gather (the criterion's access pattern), a fill.
usage (GPU): python tools/probe_gru_beside.py [B]"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402


def main():
    B, S = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 128
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
    y = torch.empty(B, S, 256, device=dev)
    hN = torch.empty(2, B, 256, device=dev)
    dx = torch.empty(B, S, 256, device=dev)
    grads = [torch.empty_like(t) for t in plist]
    parr = (ctypes.c_void_p * 8)(*[P(t) for t in plist])
    garr = (ctypes.c_void_p * 8)(*[P(t) for t in grads])
    main_s, side_s = torch.cuda.Stream(), torch.cuda.Stream()

    def fwd():
        lib.check(lib.cpc_gru_forward(P(x), None, parr, P(saved), P(fscr), P(y), P(hN), B, S, 2, main_s.cuda_stream))

    def bwd():
        lib.check(lib.cpc_gru_backward(P(x), None, parr, P(saved), P(y), P(dy), P(bscr), P(dx), garr, B, S, 2, main_s.cuda_stream))

    big_a, big_b = torch.empty(128 << 20, device=dev), torch.empty(128 << 20, device=dev)       # 512 MB each
    small_a, small_b = torch.empty(256 << 10, device=dev), torch.empty(256 << 10, device=dev)   # 1 MB each
    ma, mb = torch.randn(8192, 8192, device=dev, dtype=torch.float16), torch.randn(8192, 8192, device=dev, dtype=torch.float16)
    table = torch.randn(8192, 256, device=dev)
    idx = torch.randint(0, 8192, (1 << 20,), device=dev)
    loads = {
        "alone": (None, 0),
        "hbm_copy_512MB": (lambda: big_b.copy_(big_a), 12),
        "l2_copy_1MB": (lambda: small_b.copy_(small_a), 600),
        "fp16_matmul_8192": (lambda: torch.matmul(ma, mb), 8),
        "gather_1KB_rows": (lambda: torch.index_select(table, 0, idx), 10),
        "fill_512MB": (lambda: big_b.fill_(1.0), 20),
    }
    out = {"B": B}
    for name, (load, reps) in loads.items():
        for what, fn in (("fwd", fwd), ("bwd", bwd)):
            with torch.cuda.stream(main_s):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if load is not None:
                with torch.cuda.stream(side_s):
                    load()                                    # already running when the recurrence starts
                    s0.record()
                    for _ in range(reps):
                        load()
                    s1.record()
            with torch.cuda.stream(main_s):
                e0.record()
                for _ in range(3):
                    fn()
                e1.record()
            torch.cuda.synchronize()
            out[f"{what}_ms_{name}"] = round(e0.elapsed_time(e1) / 3, 4)
            if load is not None:
                out[f"{what}_side_ms_{name}"] = round(s0.elapsed_time(s1), 3)     # must exceed 3 x the call for full cover
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
