"""Encoder forward + backward alone on ONE stream (every kernel runs by itself, back to back), B = 64 by default: the
per-kernel durations without co-runners, for rocprofv3 --kernel-trace --stats.
usage (GPU): rocprofv3 --kernel-trace --stats -d out -- python tools/trace_encoder.py [B] [reps]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402
from cpc_audio_amd.model import CPCEncoder  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    L = 20480
    dev = torch.device("cuda:0")
    lib = _lib.get()
    torch.manual_seed(0)
    enc = CPCEncoder(256, "layerNorm").to(dev)
    names = [f"{n}{i}.{w}" for i in range(5) for n, w in (("conv", "weight"), ("conv", "bias"), ("batchNorm", "weight"), ("batchNorm", "bias"))]
    sd = dict(enc.named_parameters())
    plist = [sd[n].detach().contiguous() for n in names]
    wave = (0.1 * torch.randn(B, L, device=dev)).clamp_(-1, 1)
    sizes = (ctypes.c_long * 22)()
    assert lib.cpc_encoder_layout(B, L, sizes) == 0
    Ls = [sizes[3 + i] for i in range(5)]
    saved = torch.empty(sizes[0], device=dev)
    fscr = torch.empty(max(1, sizes[1]), device=dev)
    bscr = torch.empty(sizes[2], device=dev)
    z = torch.empty(B, Ls[4], 256, device=dev)
    dz = torch.randn(B, Ls[4], 256, device=dev) * 1e-3
    grads = [torch.empty_like(t) for t in plist]
    parr = (ctypes.c_void_p * 20)(*[P(t) for t in plist])
    garr = (ctypes.c_void_p * 20)(*[P(t) for t in grads])
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(reps):
        lib.check(lib.cpc_encoder_forward(P(wave), parr, P(saved), P(fscr), P(z), B, L, st), "encoder_forward")
        lib.check(lib.cpc_encoder_backward(P(wave), parr, P(saved), P(z), P(dz), P(bscr), garr, B, L, st), "encoder_backward")
    torch.cuda.synchronize()
    print("ok", float(z.abs().mean()), float(grads[4].abs().mean()))


if __name__ == "__main__":
    main()
