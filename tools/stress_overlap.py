"""Repeats the train step with all stream overlaps on and compares every gradient, bit for bit, with the single-stream
run on the same inputs (tests/test_gpu_train_step.py does this three times; a hazard between co-resident kernels can
be rarer than that).  Usage (GPU): python tools/stress_overlap.py [n_steps=60] [batch=64]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    from cpc_audio_amd import ops
    from cpc_audio_amd.train import build_criterion, build_model
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model, crit = build_model().to(dev), build_criterion().to(dev)
    params = list(model.parameters()) + list(crit.parameters())
    wave = (0.1 * torch.randn(B, 1, 20480)).clamp_(-1, 1).to(dev)
    label = torch.zeros(B, dtype=torch.long, device=dev)
    g = torch.Generator(device="cpu").manual_seed(1)
    negs = (torch.randint(0, B, (B * 128 * 116,), generator=g).to(dev),
            torch.randint(1, 128, (B * 128 * 116,), generator=g).to(dev))

    def step(overlap):
        for p in params:
            p.grad = None
        with ops.StepContext(overlap=overlap) as sc:
            c, z, _ = model(wave, label)
            losses, _ = crit(c, z, None, negatives=negs)
            torch.autograd.backward([losses], [torch.ones_like(losses)])
            sc.wait()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in params], losses.detach().clone()

    ref, lref = step(False)
    ref2, _ = step(False)
    assert all(torch.equal(a, b) for a, b in zip(ref, ref2)), "single-stream run is not reproducible"
    bad = 0
    for i in range(n):
        cur, l = step(True)
        diff = [k for k, (a, b) in enumerate(zip(ref, cur)) if not torch.equal(a, b)]
        if diff or not torch.equal(l, lref):
            bad += 1
            print(f"step {i}: {len(diff)} gradient tensors differ (first index {diff[:4]}), loss equal {torch.equal(l, lref)}",
                  flush=True)
    print(f"{n} overlapped steps at B = {B}: {bad} with any difference", flush=True)


if __name__ == "__main__":
    main()
