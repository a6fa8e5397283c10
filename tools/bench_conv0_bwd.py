"""Time cpc_conv0_backward (conv0 + ChannelNorm backward from dy0; hip events) alone.  usage: python tools/bench_conv0_bwd.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = 20480
dev = torch.device("cuda:0")
lib = _lib.get()
g = torch.Generator(device="cpu").manual_seed(1)
wave = (0.1 * torch.randn(B, L, generator=g)).clamp_(-1, 1).to(dev)
w0 = (torch.randn(256, 10, generator=g) * 0.3).to(dev)
b0 = (torch.randn(256, generator=g) * 0.1).to(dev)
nw, nb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
y0 = torch.empty(B, 4096, 256, device=dev)
m0, r0 = torch.empty(B * 4096, device=dev), torch.empty(B * 4096, device=dev)
lib.check(lib.cpc_conv0_forward(P(wave), P(w0), P(b0), P(nw), P(nb), P(y0), P(m0), P(r0), B, L, None))
dy0 = (torch.randn(B, 4096, 256, generator=g) * 1e-3).to(dev)
scr = torch.empty(lib.cpc_conv0_backward_scratch_floats(B, L), device=dev)
grads = [torch.empty(256, 10, device=dev)] + [torch.empty(256, device=dev) for _ in range(3)]
st = torch.cuda.current_stream().cuda_stream


def f():
    lib.check(lib.cpc_conv0_backward(P(wave), P(w0), P(b0), P(nw), P(nb), P(m0), P(r0), P(dy0), P(scr), P(grads[0]),
                                     P(grads[1]), P(grads[2]), P(grads[3]), B, L, st))


for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f()
e1.record()
torch.cuda.synchronize()
print(f"conv0_backward B={B}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  checksum {float(grads[0].double().abs().sum()):.6e}")
