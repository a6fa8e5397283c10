"""conv0 forward (H2 output) alone at B = 64 for several groups-per-wave settings (cpc_set_conv0_groups), and the dominant
pair conv0 -> conv1 as the step runs it.  usage: python tools/sweep_conv0.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402

lib = _lib.get()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L, L0 = 20480, 4096
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
wave = (0.1 * torch.randn(B, L, device=dev)).clamp_(-1, 1)
w0 = torch.randn(256, 10, device=dev) * 0.3
bias = torch.randn(256, device=dev) * 0.1
nw, nb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
bound = (15.968719 * nw.abs().max() + nb.abs().max()).view(1).clone()
y0 = torch.empty(B, L0, 256, device=dev)
m0, r0 = torch.empty(B * L0, device=dev), torch.empty(B * L0, device=dev)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)          # pushes y0 out of the 256 MB Infinity Cache
byts = B * (L * 4 + L0 * 256 * 4 + 2 * L0 * 4)
configs = [(4, 1)] if len(sys.argv) > 2 else [(g, n) for n in (1, 0) for g in (2, 4, 6, 8)]
for groups, nt in configs:
    lib.check(lib.cpc_set_conv0_tuning(groups, nt))
    ts = []
    for it in range(12):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.check(lib.cpc_conv0_forward_h2(P(wave), P(w0), P(bias), P(nw), P(nb), P(y0), P(m0), P(r0), P(bound), B, L, st))
        e1.record()
        torch.cuda.synchronize()
        if it >= 2:
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"groups {groups:2d} nt {nt}: median {1e3 * med:6.1f} us  min {1e3 * ts[0]:6.1f}  = {byts / med / 1e6:6.0f} GB/s "
          f"({byts / med / 1e6 / 8000:.3f} of 8 TB/s)", flush=True)
