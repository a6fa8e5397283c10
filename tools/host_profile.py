"""Where the host time of one eager train step goes (cProfile over a few steps; GPU box).
usage: python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd.train import Trainer, build_criterion, build_model      # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model, crit = build_model().to(dev), build_criterion().to(dev)
    tr = Trainer(model, crit, graph=False)
    wave = (0.1 * torch.randn(64, 1, 20480)).clamp_(-1, 1).to(dev)
    label = torch.zeros(64, dtype=torch.long, device=dev)
    for _ in range(10):
        tr.step(wave, label)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        tr.step(wave, label)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
