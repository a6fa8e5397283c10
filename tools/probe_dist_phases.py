"""What the data-parallel form of the step costs on ONE rank (RCCL group of one): the composite in one call, in two phases without
any collective, with only the early / only the late bucket, and in full.  usage (GPU): python tools/probe_dist_phases.py [steps=60]"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd.train import Trainer, build_criterion, build_model   # noqa: E402


def run(tr, wave, label, steps):
    for _ in range(8):
        tr.step(wave, label)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(wave, label)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device("cuda:0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    torch.manual_seed(0)
    model, crit = build_model().to(dev), build_criterion().to(dev)
    tr = Trainer(model, crit)
    wave = (0.1 * torch.randn(64, 1, 20480)).clamp_(-1, 1).to(dev)
    label = torch.zeros(64, dtype=torch.long, device=dev)
    ar = tr.allreduce
    real_begin, real_call, real_reduce = ar.begin, type(ar).__call__, ar._reduce
    print(f"one call                       {run(tr, wave, label, steps):.3f} ms/step")
    ar.single_rank_too = True
    ar.begin = lambda step=None: None
    type(ar).__call__ = lambda self: None
    print(f"two phases, no collective      {run(tr, wave, label, steps):.3f} ms/step")
    ar.begin = real_begin
    type(ar).__call__ = lambda self: (self._pending.wait() if self._pending is not None else None,
                                      torch.cuda.current_stream().wait_event(self._pending_event) if self._pending_event is not None else None,
                                      setattr(self, "_pending", None), setattr(self, "_pending_event", None)) and None
    print(f"two phases + early bucket      {run(tr, wave, label, steps):.3f} ms/step")
    ar.begin = lambda step=None: None
    type(ar).__call__ = real_call
    print(f"two phases + one late bucket   {run(tr, wave, label, steps):.3f} ms/step")
    ar.begin = real_begin
    print(f"full (early + late)            {run(tr, wave, label, steps):.3f} ms/step")
    # the collectives alone, back to back on the main stream
    buf = ar.buf
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        dist.all_reduce(buf[:ar.n_early])
        dist.all_reduce(buf[ar.n_early:])
    e1.record()
    torch.cuda.synchronize()
    print(f"the two all-reduces alone      {e0.elapsed_time(e1) / 50:.3f} ms per pair ({buf.numel() * 4 / 1e6:.1f} MB)")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
