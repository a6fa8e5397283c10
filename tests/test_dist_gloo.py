"""N > 1 path on CPU: world_size-2 gloo runs of the flat-bucket gradient SUM all-reduce and of the
per-rank sharding of the synthetic batch (one process per device, no data-path collective)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cpc_audio_amd.dist import FlatGradAllReduce
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7)),
              torch.nn.Parameter(torch.zeros(1, 4, 1))]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    ar = FlatGradAllReduce(params)
    ar()
    ok = all(torch.equal(p.grad, torch.full_like(p, float(3 * (i + 1)))) for i, p in enumerate(params))
    # second call reuses the persistent bucket
    for p in params:
        p.grad.fill_(float(rank))
    ar()
    ok = ok and all(torch.equal(p.grad, torch.full_like(p, 1.0)) for p in params)
    # two buckets: the early one is sent off by begin(), the rest by the call; a parameter without gradient counts 0
    early = [params[2], params[0]]
    ar2 = FlatGradAllReduce(params, early=early)
    for rep in range(2):
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, float((rank + 1) * (i + 1) + rep))
        if rank == 1:
            params[0].grad = None
        ar2.begin()
        ar2()
        want = [1.0 + rep, 6.0 + 2 * rep, 9.0 + 2 * rep]
        ok = ok and all(torch.equal(p.grad, torch.full_like(p, w)) for p, w in zip(params, want))
    # ... and without begin() the same object falls back to one collective
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + i))
    ar2()
    ok = ok and all(torch.equal(p.grad, torch.full_like(p, float(1 + 2 * i))) for i, p in enumerate(params))
    # a step that raised after begin(): abort() lets the early bucket's collective finish on every rank and forgets it; the next
    # step's two-bucket exchange starts clean
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, 100.0)
    ar2.begin()
    ar2.abort()
    ok = ok and ar2._pending is None and ar2._pending_event is None
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    ar2.begin()
    ar2()
    ok = ok and all(torch.equal(p.grad, torch.full_like(p, float(3 * (i + 1)))) for i, p in enumerate(params))
    # three buckets (early | mid | late): on CPU tensors the mid bucket has no stream of its own and travels with the late one;
    # buffer order and sums as with two
    ar3 = FlatGradAllReduce(params, early=[params[1]], mid=[params[2]])
    ok = ok and [id(p) for p in ar3.params] == [id(params[1]), id(params[2]), id(params[0])]
    ok = ok and (ar3.n_early, ar3.n_mid, ar3.numel) == (7, 4, 26)
    for rep in range(2):
        for i, p in enumerate(params):
            p.grad = torch.full_like(p, float((rank + 1) * (i + 1) + rep))
        ar3.begin()
        ar3(mid_wait=lambda stream: None)
        ok = ok and all(torch.equal(p.grad, torch.full_like(p, float(3 * (i + 1) + 2 * rep))) for i, p in enumerate(params))
    # sharding: each rank draws its own sequences; no overlap, identical parameters
    g = torch.Generator().manual_seed(1234 + rank)
    wave = (0.1 * torch.randn(2, 1, 64, generator=g)).clamp_(-1, 1)
    gathered = [torch.zeros_like(wave) for _ in range(world)]
    dist.all_gather(gathered, wave)
    ok = ok and not torch.equal(gathered[0], gathered[1])
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_grad_allreduce_sum_world2():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_allreduce_is_noop_without_process_group():
    from cpc_audio_amd.dist import FlatGradAllReduce
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    FlatGradAllReduce([p])()
    assert torch.equal(p.grad, torch.ones(4))
