"""Which streams share a hardware queue: the overlap matrix of the default stream, six normal- and five high-priority pool streams
(cpc_streams_overlap; 0 = the pair serialises).  usage (GPU): python tools/probe_stream_queues.py [init]   (init: after init_process_group("nccl"))"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpc_audio_amd import ops
dev = torch.device("cuda:0")
torch.zeros(4, device=dev)
if len(sys.argv) > 1 and sys.argv[1] == "init":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    t = torch.zeros(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
ss = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(6)] + [torch.cuda.Stream(dev, priority=-1) for _ in range(5)]
print("queues", os.environ.get("GPU_MAX_HW_QUEUES"), sys.argv[1:])
for i, a in enumerate(ss):
    print(i, "".join("-" if i == j else ("1" if ops.streams_overlap(a, b) else "0") for j, b in enumerate(ss)))
