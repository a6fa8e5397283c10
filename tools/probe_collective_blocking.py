"""Does a collective that is PENDING on the side stream (queued behind a long kernel there) hold back kernels of the other streams of the
step?  RCCL orders the launches of one communicator through an internal stream of its own; that stream shares a hardware queue with
whichever streams the runtime mapped there, and a wait queued on it blocks everything behind it in that queue.
usage (GPU, one rank): python tools/probe_collective_blocking.py        prints, per stream, whether its kernel finished while the
collective was still pending."""
import ctypes
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import _lib, ops   # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29523")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    ctx = ops.StepContext(overlap=True)
    main_s = torch.cuda.current_stream(dev)
    s0, s1, s2 = ctx.reserve(dev)
    buf = torch.ones(1 << 20, device=dev)
    small = [torch.zeros(64, device=dev) for _ in range(3)]
    dist.all_reduce(buf)                      # communicator up
    torch.cuda.synchronize()
    for trial in range(3):
        spin_done, coll_done = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(s0):
            torch.cuda._sleep(int(2.4e9 * 0.004))            # ~4 ms of one wavefront
            spin_done.record()
            dist.all_reduce(buf)                              # synchronous op: launched on s0 itself, pending behind the spin
            coll_done.record()
        res = {}
        for name, st, t in (("main", main_s, small[0]), ("prep", s1, small[1]), ("wgrad", s2, small[2])):
            e = torch.cuda.Event()
            with torch.cuda.stream(st):
                t.add_(1.0)
                e.record()
            e.synchronize()
            res[name] = "free" if not spin_done.query() else "finished only after the spin (blocked, or the spin was too short)"
        torch.cuda.synchronize()
        print(f"trial {trial}: " + ", ".join(f"{k}: {v}" for k, v in res.items()), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
