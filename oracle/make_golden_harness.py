"""Fixtures for the train harness and chunked feature extraction from the REFERENCE's own
code -- tests/golden/harness.json.  TEST INFRASTRUCTURE ONLY; run in the build container:

    python oracle/make_golden_harness.py

* ``lr``: the learning rate every epoch trains with, for a table of (--schedulerStep, --schedulerRamp, epochs already logged),
  produced by the scheduler the reference builds in cpc/train.py:351-370 -- ``StepLR(gamma=0.5)``, the ``LambdaLR`` ramp of
  ``cpc/utils/misc.py:77-81`` and their ``SchedulerCombiner`` (:84-121) -- stepped once per epoch as ``trainStep`` does
  (cpc/train.py:113-114), including the fast-forward of a resumed run (:368-370).
* ``chunks``: what ``cpc/feature_loader.py:228-269`` (``buildFeature``) feeds its feature maker and returns, for a table of
  (file length, strict, seqNorm): a recording feature maker (downsampling factor 160; output frame t, channel 0 = mean of its 160
  samples, channel 1 = first sample of the chunk) notes every chunk's first sample and length -- the waveform is ``i * 2**-20`` at
  sample i, so a chunk identifies itself --; ``torchaudio.load`` is replaced by the in-memory waveform and ``Tensor.cuda`` by the
  identity (no GPU in this container).

tests/test_harness_golden.py holds cpc_audio_amd.harness (build_scheduler / SchedulerCombiner, chunk_plan / build_feature /
seq_normalization) to them on the CPU; tests/test_gpu_harness.py runs build_feature with the same recording module on device tensors.
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ref_import                    # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")

LR_CASES = [  # (schedulerStep, schedulerRamp, epochs already in the logs, epochs to run)
    (-1, None, 0, 6), (3, None, 0, 10), (-1, 4, 0, 8), (2, 3, 0, 12), (5, 10, 0, 24), (2, 3, 5, 8), (1, 1, 0, 5), (4, 2, 3, 9)]
CHUNK_CASES = [64000 * 2 + 12345, 64000, 63999, 64001, 1000, 64000 * 3, 128000 + 159, 128000 + 160, 64000 + 12354, 159]


class Recorder(torch.nn.Module):
    """The feature maker the chunk fixtures were recorded with; tests build the same one."""

    def __init__(self):
        super().__init__()
        self.seen = []

    def getDownsamplingFactor(self):
        return 160

    def forward(self, data):
        x, _ = data                                   # (k, 1, n)
        for row in x[:, 0]:
            self.seen.append((int(round(float(row[0]) * 2 ** 20)), int(row.numel())))
        k, n = x.shape[0], x.shape[2]
        t = n // 160
        frames = x[:, 0, :t * 160].reshape(k, t, 160).double().mean(dim=2).float()
        first = x[:, 0, :1].expand(k, t)
        return torch.stack([frames, first], dim=2)


def waveform(n):
    return (torch.arange(n, dtype=torch.float32) * 2.0 ** -20).view(1, n)


def main():
    ref_import.import_reference()
    import cpc.feature_loader as F
    import cpc.utils.misc as M
    res = {"torch": torch.__version__, "lr": [], "chunks": []}

    for step, ramp, logged, epochs in LR_CASES:
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=2e-4)
        sched = None                                   # cpc/train.py:351-367, with the reference's own classes
        if step > 0:
            sched = torch.optim.lr_scheduler.StepLR(opt, step, gamma=0.5)
        if ramp is not None:
            n_epoch = ramp
            sr = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda e: M.ramp_scheduling_function(n_epoch, e), last_epoch=-1)
            sched = sr if sched is None else M.SchedulerCombiner([sr, sched], [0, ramp])
        if sched is not None:
            for _ in range(logged):                    # :368-370
                sched.step()
        lrs = []
        for _ in range(epochs):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            if sched is not None:
                sched.step()                           # cpc/train.py:113-114
        res["lr"].append({"schedulerStep": step, "schedulerRamp": ramp, "logged": logged, "lrs": lrs})

    F.torchaudio.load = lambda path: (waveform(int(path)), 16000)
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for n in CHUNK_CASES:
            for strict in (False, True):
                for norm in (False, True):
                    rec = Recorder()
                    out = F.buildFeature(rec, str(n), strict=strict, maxSizeSeq=64000, seqNorm=norm)
                    fin = torch.isfinite(out)               # (seqNorm of a one-frame chunk: var() of one element is NaN)
                    o = torch.where(fin, out, torch.zeros_like(out)).double()
                    T = out.shape[1]
                    res["chunks"].append({"n": n, "strict": strict, "seqNorm": norm, "seen": rec.seen, "shape": list(out.shape),
                                          "n_nonfinite": int((~fin).sum()), "sum": float(o.sum()), "abs_sum": float(o.abs().sum()),
                                          "first": [float(v) for v in o[0, 0]] if T else [],
                                          "last": [float(v) for v in o[0, -1]] if T else []})
    finally:
        torch.Tensor.cuda = cuda

    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, "harness.json")
    with open(path, "w") as f:
        json.dump(res, f, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
