"""Per-parameter relative difference of config 4's gradients between the DMA-fed and the generic feed-forward GEMMs (B = 64)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib
from cpc_audio_amd.train import build_criterion, build_model
from oracle import cpc_oracle as O

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(11)
model = build_model(arMode="transformer", transformerDropout=0.0).to(dev)
crit = build_criterion(rnnMode="transformer", transformerDropout=0.0).to(dev)
model.train(); crit.train()
wave = O.make_waveform(B, 20480, seed=71).to(dev)
g = torch.Generator().manual_seed(19)
bi, si = O.draw_negative_indices(B, 128, 116, 128, generator=g)
neg = (bi.to(dev), si.to(dev))
names = [n for n, _ in model.named_parameters()] + [n for n, _ in crit.named_parameters()]
params = list(model.parameters()) + list(crit.parameters())

def run(mode):
    _lib.get().check(_lib.get().cpc_set_gemm_dma(mode), "dma")
    for q in params:
        q.grad = None
    c, z, _ = model(wave, None)
    losses, acc = crit(c, z, None, negatives=neg)
    losses.sum().backward()
    torch.cuda.synchronize()
    return losses.detach().clone(), [q.grad.clone() for q in params]

l0, g0 = run(0)
l1, g1 = run(1 if len(sys.argv) < 3 else int(sys.argv[2]))
print("loss diff", (l0 - l1).abs().max().item())
rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-30)).item()
rows = sorted(((rel(a, b), n, tuple(a.shape)) for n, a, b in zip(names, g1, g0)), reverse=True)
for r in rows[:14]:
    print(f"{r[0]:.3e}  {r[1]}  {r[2]}")
