"""Launch the two roofline kernels a few times (for rocprofv3 --pmc runs):
conv_fwd_kernel<128,X3> on encoder layer 1 and conv0_fwd_kernel, B = 64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib
from cpc_audio_amd._lib import ptr as P
lib = _lib.get(); dev = torch.device("cuda:0")
B, Lin, k, s, p = 64, 4096, 8, 4, 2
x = torch.randn(B, Lin, 256, device=dev).relu_(); w = torch.randn(256, 256, k, device=dev) / 45; wp = torch.empty(256 * k * 256 * 3 // 2, device=dev)
bias = torch.randn(256, device=dev) * 0.1; nw = torch.ones(256, device=dev); nb = torch.zeros(256, device=dev)
y = torch.empty(B, 1024, 256, device=dev); xh = torch.empty_like(y); rs = torch.empty(B * 1024, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.check(lib.cpc_conv_weight_relayout(P(w), P(wp), k, st))
L = 20480
wave = torch.randn(B, L, device=dev) * 0.1; w0 = torch.randn(256, 10, device=dev) * 0.3
y0 = torch.empty(B, 4096, 256, device=dev); m0 = torch.empty(B * 4096, device=dev); r0 = torch.empty(B * 4096, device=dev)
xamax = torch.zeros(1, device=dev)
lib.check(lib.cpc_conv0_forward(P(wave), P(w0), P(bias), P(nw), P(nb), P(y0), P(m0), P(r0), B, L, st))
lib.check(lib.cpc_absmax(P(y0), y0.numel(), P(xamax), st))
for _ in range(5):      # layer 1 runs on the activations conv0 produces, as inside the train step (and as bench.py's probe)
    lib.check(lib.cpc_conv0_forward(P(wave), P(w0), P(bias), P(nw), P(nb), P(y0), P(m0), P(r0), B, L, st))
    lib.check(lib.cpc_conv_gemm_forward(P(y0), P(wp), P(bias), P(nw), P(nb), P(y), P(xh), P(rs), P(xamax), B, Lin, k, s, p, st))
torch.cuda.synchronize()
print("done")
