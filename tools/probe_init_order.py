"""ms per train step by start-up order: init_first = GPU touched, then init_process_group("nccl"), then the
Trainer; steps_first = Trainer and steps first, the process group after.  usage (GPU): python tools/probe_init_order.py init_first|steps_first [steps]
(AMD_LOG_LEVEL=3 on stderr shows the hardware queues being created.)"""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd.train import Trainer, build_criterion, build_model
from tools.probe_dist_phases import run
dev = torch.device("cuda:0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
order = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
torch.manual_seed(0)
wave = (0.1 * torch.randn(64, 1, 20480)).clamp_(-1, 1).to(dev)
label = torch.zeros(64, dtype=torch.long, device=dev)
print("MARK before init", file=sys.stderr, flush=True)
if order == "init_first":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
print("MARK after init", file=sys.stderr, flush=True)
model, crit = build_model().to(dev), build_criterion().to(dev)
tr = Trainer(model, crit)
print("MARK trainer built", file=sys.stderr, flush=True)
print(order, f"{run(tr, wave, label, steps):.3f}")
if order == "steps_first":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    print(order, "after init", f"{run(tr, wave, label, steps):.3f}")
