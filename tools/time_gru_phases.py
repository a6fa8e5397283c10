"""Phase trace of the persistent two-layer recurrence (gru.hip, PhaseClock): shader clocks per time step, averaged over the
steps 16 .. S - 16, per layer and wave kind.  Needs the instrumented library:
    bash tools/build_timing_lib.sh && CPC_HIP_LIB=tools/_bin/libcpc_gru_timing.so python tools/time_gru_phases.py [B] [S]
MFMA waves: poll | operand math + MFMAs (until the partials are in LDS-store flight) | LDS stores + barrier;
gate waves: barrier wait | partial sums + gate math until the coherent store is issued | the other stores + next prefetch."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    dev = torch.device("cuda:0")
    lib = _lib.get()
    raw = ctypes.CDLL(os.environ["CPC_HIP_LIB"])
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
    for _ in range(4):
        lib.check(lib.cpc_gru_forward(P(x), None, parr, P(saved), P(fscr), P(y), P(hN), B, S, 2, st))
        lib.check(lib.cpc_gru_backward(P(x), None, parr, P(saved), P(y), P(dy), P(bscr), P(dx), garr, B, S, 2, st))
    torch.cuda.synchronize()
    host = (ctypes.c_ulonglong * (2 * 512 * 12 * 8))()
    assert raw.cpc_debug_gru_phases(host) == 0
    a = np.ctypeslib.as_array(host).reshape(2, 512, 12, 8).astype(np.float64)
    G = (B + 15) // 16
    nsteps = S - 32
    nwg = 32 * G
    names = ["poll", "math+MFMA", "LDS+barrier", "barrier wait", "sums+gate math -> coherent store", "stores+prefetch"]
    for d, dname in enumerate(("forward", "backward")):
        for slot in (0, 1):                       # layer field of PersistIds (backward: slot 0 = the top layer, which leads)
            layer = slot if d == 0 else 1 - slot
            wgs = [b for b in range(nwg) if (b // G) >> 4 == slot]
            sub = a[d][wgs] / nsteps              # (wg, wave, phase)
            print(f"== {dname} layer {layer}: {len(wgs)} workgroups, clocks per step")
            kinds = [("MFMA waves 0-3", range(0, 4), (0, 1, 2)), ("MFMA waves 4-7", range(4, 8), (0, 1, 2)),
                     ("gate waves", range(8, 12), (3, 4, 5))]
            for kname, waves, phases in kinds:
                v = sub[:, list(waves), :][:, :, list(phases)]
                tot = v.sum(-1)
                txt = "  ".join(f"{names[p]} {v[:, :, i].mean():7.0f} (max {v[:, :, i].max():6.0f})" for i, p in enumerate(phases))
                print(f"   {kname:15s} step {tot.mean():7.0f}  |  {txt}")


if __name__ == "__main__":
    main()
