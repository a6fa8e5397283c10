"""GPU debug: encoder backward under different tile sizes, run twice each; compare everything."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib
from cpc_audio_amd._lib import ptr as P
from oracle import cpc_oracle as O

B, L = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
lib = _lib.get()
names = [f"gEncoder.{n}{i}.{w}" for i in range(5)
         for n, w in (("conv", "weight"), ("conv", "bias"), ("batchNorm", "weight"), ("batchNorm", "bias"))]
p = O.make_params(seed=0)
plist = [p[n].contiguous().to(dev) for n in names]
wave = O.make_waveform(B, L, seed=5)
wd = wave.to(dev)
g = torch.Generator().manual_seed(11)
st = torch.cuda.current_stream().cuda_stream
results = {}
dz = None
for bm in (32, 128, 128, 64, 32):
    lib.cpc_set_conv_tile(bm)
    sizes = (ctypes.c_long * 22)()
    lib.cpc_encoder_layout(B, L, sizes)
    Ls = [sizes[3 + i] for i in range(5)]
    if dz is None:
        dz = torch.randn(B, Ls[4], 256, generator=g)
        dzd = dz.to(dev)
    saved = torch.full((sizes[0],), float("nan"), device=dev)
    fscr = torch.full((max(1, sizes[1]),), float("nan"), device=dev)
    bscr = torch.full((sizes[2],), float("nan"), device=dev)
    z = torch.full((B, Ls[4], 256), float("nan"), device=dev)
    parr = (ctypes.c_void_p * 20)(*[P(t) for t in plist])
    grads = [torch.full_like(t, float("nan")) for t in plist]
    garr = (ctypes.c_void_p * 20)(*[P(t) for t in grads])
    lib.check(lib.cpc_encoder_forward(P(wd), parr, P(saved), P(fscr), P(z), B, L, st))
    lib.check(lib.cpc_encoder_backward(P(wd), parr, P(saved), P(z), P(dzd), P(bscr), garr, B, L, st))
    torch.cuda.synchronize()
    results.setdefault(bm, []).append([x.cpu() for x in grads])
lib.cpc_set_conv_tile(0)
leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith("gEncoder")}
zr = O.encoder_forward(leaves, wave).permute(0, 2, 1)
(zr * dz).sum().backward()
ref = [leaves[n].grad for n in names]
def rel(a, b):
    return ((a.reshape(-1) - b.reshape(-1)).norm() / (b.norm() + 1e-30)).item()
for bm, runs in results.items():
    for ri, gr in enumerate(runs):
        errs = {n.replace("gEncoder.", ""): f"{rel(a, b):.1e}" for n, a, b in zip(names, gr, ref)}
        print(f"bm={bm} run{ri}: ", errs)
    if len(runs) > 1:
        print(f"bm={bm} run0 vs run1 max rel:", max(rel(a, b) for a, b in zip(runs[0], runs[1])))
