import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpc_oracle as O
from cpc_audio_amd import _lib
from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
dev = torch.device("cuda:0")
for split in (1, 0):
    _lib.get().cpc_set_gemm_split(split)
    B = 4; S, K, N = 128, 12, 128; W = S - K
    p = O.make_params(seed=31, head_scale=64.0)
    model, crit = build_model(keepHidden=True).to(dev), build_criterion().to(dev)
    load_flat_params(model, crit, p)
    model.train(); crit.train()
    tr = Trainer(model, crit)
    cpu = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.Adam(list(cpu.values()), lr=2e-4, betas=(0.9, 0.999), eps=1e-8)
    g = torch.Generator().manual_seed(9)
    h = None
    for i in range(3):
        wave = O.make_waveform(B, 20480, seed=100 + i)
        bi, si = O.draw_negative_indices(B, S, W, N, generator=g)
        losses, _ = tr.step(wave.to(dev), None, negatives=(bi.to(dev), si.to(dev)))
        ora = O.train_step({k: v.detach() for k, v in cpu.items()}, wave, bi, si, h0=h)
        h = ora["hN"]
        print(split, i, "loss diff", (losses.cpu() - ora["losses"]).abs().max().item())
        if i == 0:
            grads = {k: q.grad for k, q in list(model.named_parameters()) + list(crit.named_parameters())}
        for k, v in cpu.items():
            v.grad = ora["grads"][k]
        opt.step()
    new = dict(model.state_dict()); new.update(crit.state_dict())
    rows = []
    for k in cpu:
        d = (new[k].cpu() - cpu[k].detach()).abs()
        rows.append((float((d > 2e-6).float().mean()), float(d.max()), k, d.numel()))
    rows.sort(reverse=True)
    for r in rows[:6]:
        print(split, "frac_far %.5f worst %.2e %s n=%d" % r)
