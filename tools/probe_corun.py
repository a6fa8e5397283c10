"""Co-residency probe: does a kernel give bit-identical results when another kernel of this library runs beside it on a
second stream?  (It must: the kernels share no memory.)  Written while moving the weight-gradient GEMMs onto their own
stream.  Findings on MI355X, ROCm 7.2, run after run:
  * conv_dgrad_kernel<128,false,2> beside conv_wgrad_kernel<2>: both bit-exact
  * conv0_bwd_kernel built with the default flags (SLP vectoriser -> v_pk_fma_f32 with op_sel on ds_read2_b32 pairs)
    beside conv_wgrad_kernel<1|2> or conv_dgrad_kernel<128,false,2> (16-bit MFMA kernels): ~20 % of its workgroups
    returned partial sums off by ~1e-3 relative; beside the exact-f32 wgrad, a rocBLAS GEMM or device copies: exact
  * the same kernel built with -fno-slp-vectorize (no packed fp32; what build.py does now): exact beside all of them
Usage (GPU): python tools/probe_corun.py"""
import sys

import torch

sys.path.insert(0, '.')
from cpc_audio_amd import _lib
from cpc_audio_amd._lib import ptr as P
lib = _lib.get(); dev = torch.device('cuda:0')
torch.manual_seed(0)
B, Lin, k, s, p = 16, 4096, 8, 4, 2
Lout = 1024
dx = torch.randn(B, Lout, 256, device=dev) * 1e-3
x = torch.relu(torch.randn(B, Lin, 256, device=dev))
w = torch.randn(256, 256, k, device=dev) / 45
wd = torch.empty(256 * k * 256 * 3 // 2 + 64, device=dev)
dprev = torch.empty(B, Lin, 256, device=dev)
amax = torch.zeros(4, device=dev)
lib.check(lib.cpc_absmax(P(dx), dx.numel(), P(amax), None))
xam = torch.zeros(1, device=dev); lib.check(lib.cpc_absmax(P(x), x.numel(), P(xam), None))
M = B * Lout; K = k * 256; tiles = 2 * (K // 128); S = -(-768 // tiles); rows = -(-(-(-M // S)) // 32) * 32
rows = max(rows, 256); S = -(-M // rows)
part = torch.empty(S * 256 * K, device=dev); dW = torch.empty_like(w)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def dgrad(st):
    lib.check(lib.cpc_conv_layer_dgrad(P(dx), P(w), P(wd), 0, None, None, None, None, P(dprev), None, None, None, P(amax), None,
                                       B, Lin, k, s, p, st.cuda_stream))
def wgrad(st):
    lib.check(lib.cpc_conv_layer_wgrad(P(dx), P(x), P(part), P(dW), P(amax), P(xam), B, Lin, k, s, p, S, rows, st.cuda_stream))
torch.cuda.synchronize()
dgrad(s1); torch.cuda.synchronize(); ref = dprev.clone()
wgrad(s2); torch.cuda.synchronize(); refw = dW.clone()
for it in range(4):
    dprev.fill_(float('nan')); dW.fill_(float('nan')); torch.cuda.synchronize()
    wgrad(s2); dgrad(s1); torch.cuda.synchronize()
    print('concurrent', it, 'dgrad equal', torch.equal(ref, dprev), ((ref - dprev).norm() / ref.norm()).item(),
          'wgrad equal', torch.equal(refw, dW), flush=True)
# conv0 backward beside wgrad
L = 20480
wave = (0.1 * torch.randn(B, L, device=dev)).clamp_(-1, 1)
w0 = torch.randn(256, 10, device=dev) * 0.3; b0 = torch.randn(256, device=dev) * 0.1
nw = torch.ones(256, device=dev); nb = torch.zeros(256, device=dev)
y0 = torch.empty(B, 4096, 256, device=dev); m0 = torch.empty(B * 4096, device=dev); r0 = torch.empty(B * 4096, device=dev)
lib.check(lib.cpc_conv0_forward(P(wave), P(w0), P(b0), P(nw), P(nb), P(y0), P(m0), P(r0), B, L, None))
dy0 = torch.randn(B, 4096, 256, device=dev) * 1e-3
scr = torch.empty(lib.cpc_conv0_backward_scratch_floats(B, L), device=dev)
g = [torch.empty(256, 10, device=dev), torch.empty(256, device=dev), torch.empty(256, device=dev), torch.empty(256, device=dev)]
def c0b(st):
    lib.check(lib.cpc_conv0_backward(P(wave), P(w0), P(b0), P(nw), P(nb), P(m0), P(r0), P(dy0), P(scr), P(g[0]), P(g[1]), P(g[2]), P(g[3]), B, L, st.cuda_stream))
torch.cuda.synchronize(); c0b(s1); torch.cuda.synchronize(); refs = [t.clone() for t in g]
for it in range(4):
    for t in g: t.fill_(float('nan'))
    torch.cuda.synchronize(); wgrad(s2); c0b(s1); torch.cuda.synchronize()
    print('concurrent conv0_bwd', it, [torch.equal(a, b) for a, b in zip(refs, g)], ((refs[0] - g[0]).norm() / refs[0].norm()).item(), flush=True)
# --- finer: partial rows of conv0_bwd under concurrency, and other co-runners
nblk = (4096 // 64) * B
n = 13 * 256
torch.cuda.synchronize(); scr.zero_(); c0b(s1); torch.cuda.synchronize()
ref_part = scr[:nblk * n].clone(); ref_tmp = scr[nblk * n:(nblk + 128) * n].clone()
A = torch.randn(4096, 4096, device=dev); Bm = torch.randn(4096, 4096, device=dev)
def corun(kind):
    if kind == 'wgrad': wgrad(s2)
    elif kind == 'matmul':
        with torch.cuda.stream(s2): torch.matmul(A, Bm)
    elif kind == 'copy':
        with torch.cuda.stream(s2): Bm.copy_(A); A.copy_(Bm); Bm.copy_(A)
for kind in ('wgrad', 'matmul', 'copy', 'none'):
    for it in range(2):
        scr.zero_(); torch.cuda.synchronize(); corun(kind); c0b(s1); torch.cuda.synchronize()
        pd = (scr[:nblk * n] != ref_part).view(nblk, n).any(1)
        td = (scr[nblk * n:(nblk + 128) * n] != ref_tmp).view(128, n).any(1)
        print(kind, it, 'bad partial rows', int(pd.sum()), 'of', nblk, 'first', pd.nonzero()[:5].flatten().tolist(),
              '| bad tmp rows', int(td.sum()), '| final equal', [torch.equal(a, b) for a, b in zip(refs, g)], flush=True)
scr.zero_(); torch.cuda.synchronize(); wgrad(s2); c0b(s1); torch.cuda.synchronize()
cur = scr[:nblk * n].view(nblk, 13, 256); ref3 = ref_part.view(nblk, 13, 256)
bad = (cur != ref3).view(nblk, -1).any(1).nonzero().flatten().tolist()
for r in bad[:3]:
    d = (cur[r] - ref3[r]).abs(); 
    print('row', r, 'per-acc max abs diff', [f'{v:.2e}' for v in d.max(1).values.tolist()], 'ref scale', f'{ref3[r].abs().max().item():.2e}',
          'n diff elems', int((cur[r] != ref3[r]).sum()), 'channels differing', (cur[r] != ref3[r]).any(0).nonzero().flatten()[:12].tolist(), flush=True)
print('--- which co-runner breaks conv0_bwd', flush=True)
def corun2(kind):
    if kind == 'dgrad': dgrad(s2)
    elif kind.startswith('wgrad_mode'):
        lib.cpc_set_mfma_mode(int(kind[-1])); wgrad(s2)
for kind in ('dgrad', 'wgrad_mode1', 'wgrad_mode0', 'wgrad_mode2'):
    scr.zero_(); torch.cuda.synchronize(); corun2(kind); lib.cpc_set_mfma_mode(2); c0b(s1); torch.cuda.synchronize()
    pd = (scr[:nblk * n] != ref_part).view(nblk, n).any(1)
    print(kind, 'bad partial rows', int(pd.sum()), flush=True)
lib.cpc_set_mfma_mode(2)
