"""Launch the roofline kernels a few times each, for rocprofv3 --pmc runs (tools/pmc_collect.sh), B = 64:
conv0_fwd_kernel<true>, conv_fwd_dma_kernel<256,32,2> on encoder layer 1 (its input produced by conv0, as in the step), and
nce_fwd_kernel (the contrastive score matrix)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib           # noqa: E402
from cpc_audio_amd._lib import ptr as P  # noqa: E402

lib = _lib.get()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream
L, Lin, k, s, p = 20480, 4096, 8, 4, 2
wave = (0.1 * torch.randn(B, L, device=dev)).clamp_(-1, 1)
w0 = torch.randn(256, 10, device=dev) * 0.3
bias = torch.randn(256, device=dev) * 0.1
nw, nb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
bound = (15.968719 * nw.abs().max() + nb.abs().max()).view(1).clone()
zeros = torch.zeros(32, device=dev)
y0 = torch.empty(B, Lin, 256, device=dev)
m0, r0 = torch.empty(B * Lin, device=dev), torch.empty(B * Lin, device=dev)
w = torch.randn(256, 256, k, device=dev) / 45
wq = torch.empty(256 * k * 256 + 64, device=dev)
lib.check(lib.cpc_conv_weight_relayout_h2(P(w), P(wq), k, st))
y = torch.empty(B, 1024, 256, device=dev)
xh = torch.empty_like(y)
rs = torch.empty(B * 1024, device=dev)
# the score matrix
S, K, N = 128, 12, 128
W = S - K
sizes = (ctypes.c_long * 6)()
lib.check(lib.cpc_nce_layout(B, S, K, N, sizes))
pred = torch.randn(B, W, K * 256, device=dev)
z = torch.randn(B, S, 256, device=dev)
ext = torch.randint(0, B * S, (B, W, N), device=dev, dtype=torch.int32)
saved = torch.empty(sizes[0], device=dev)
scratch = torch.empty(sizes[1], device=dev)
losses, acc = torch.empty(K, device=dev), torch.empty(K, device=dev)
for _ in range(5):
    lib.check(lib.cpc_conv0_forward_h2(P(wave), P(w0), P(bias), P(nw), P(nb), P(y0), P(m0), P(r0), P(bound), B, L, st))
    lib.check(lib.cpc_conv_gemm_forward_h2(P(y0), P(wq), P(bias), P(nw), P(nb), P(y), P(xh), P(rs), P(bound), P(bound), P(zeros),
                                           B, Lin, k, s, p, 0, st))
    lib.check(lib.cpc_nce_scores_forward(P(pred), P(z), P(ext), P(saved), P(scratch), P(losses), P(acc), B, S, K, N, st))
torch.cuda.synchronize()
print("done")
