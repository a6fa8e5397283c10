"""cpc_audio_amd/harness.py against what the REFERENCE's own code produced (tests/golden/harness.json, written by
oracle/make_golden_harness.py): the learning rate of every epoch under cpc/train.py:351-370's scheduler (StepLR, the ramp of
cpc/utils/misc.py:77-81, SchedulerCombiner :84-121, incl. a resumed run's fast-forward), and what cpc/feature_loader.py:228-269
(buildFeature) feeds its feature maker and returns for a table of file lengths x strict x seqNorm."""
import json
import os

import pytest
import torch

from cpc_audio_amd import harness as H

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "harness.json")))


class Recorder(torch.nn.Module):
    """The recording feature maker of oracle/make_golden_harness.py: downsampling 160; frame t = (mean of its 160 samples, first
    sample of the chunk); notes (first sample, length) of every chunk row it is given (the waveform is i * 2**-20 at sample i)."""

    def __init__(self):
        super().__init__()
        self.seen = []

    def getDownsamplingFactor(self):
        return 160

    def forward(self, data):
        x, _ = data
        for row in x[:, 0]:
            self.seen.append([int(round(float(row[0]) * 2 ** 20)), int(row.numel())])
        k, n = x.shape[0], x.shape[2]
        t = n // 160
        frames = x[:, 0, :t * 160].reshape(k, t, 160).double().mean(dim=2).float()
        first = x[:, 0, :1].expand(k, t)
        return torch.stack([frames, first], dim=2)


def waveform(n):
    return (torch.arange(n, dtype=torch.float32) * 2.0 ** -20).view(1, n)


@pytest.mark.parametrize("case", GOLD["lr"], ids=lambda c: f"step{c['schedulerStep']}-ramp{c['schedulerRamp']}-resume{c['logged']}")
def test_learning_rate_schedule_against_the_reference(case):
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=2e-4)
    sched = H.build_scheduler(opt, scheduler_step=case["schedulerStep"], scheduler_ramp=case["schedulerRamp"])
    if sched is not None:
        for _ in range(case["logged"]):                 # a resumed run (harness.run does the same: cpc/train.py:368-370)
            sched.step()
    lrs = []
    for _ in case["lrs"]:
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        if sched is not None:
            sched.step()
    assert lrs == case["lrs"]                           # the same torch schedulers driven the same way: equal to the last bit


def check_chunks(case, device=None):
    rec = Recorder()
    seq = waveform(case["n"])
    if device is not None:
        seq = seq.to(device)
    out = H.build_feature(rec, seq, strict=case["strict"], max_size_seq=64000, seq_norm=case["seqNorm"])
    # the same chunks in the same order (this package batches equally long consecutive chunks into one call: rows of it)
    assert rec.seen == case["seen"]
    assert list(out.shape) == case["shape"]
    fin = torch.isfinite(out)
    assert int((~fin).sum()) == case["n_nonfinite"]
    o = torch.where(fin, out, torch.zeros_like(out)).double()
    tol = 1e-9 if not case["seqNorm"] else 1e-4 * max(1.0, case["abs_sum"])      # (normalised: a mean / variance per chunk in fp32)
    assert abs(float(o.sum()) - case["sum"]) <= tol and abs(float(o.abs().sum()) - case["abs_sum"]) <= tol
    if out.shape[1]:
        eps = 0.0 if not case["seqNorm"] else 1e-4
        assert all(abs(float(a) - b) <= eps for a, b in zip(o[0, 0], case["first"]))
        assert all(abs(float(a) - b) <= eps for a, b in zip(o[0, -1], case["last"]))


@pytest.mark.parametrize("case", GOLD["chunks"], ids=lambda c: f"n{c['n']}-strict{int(c['strict'])}-norm{int(c['seqNorm'])}")
def test_chunked_feature_extraction_against_the_reference(case):
    check_chunks(case)
    # ... and chunk_plan says the same cut
    plan = H.chunk_plan(case["n"], 64000, case["strict"], 160)
    assert [[a, b - a] for a, b, _ in plan] == case["seen"]
