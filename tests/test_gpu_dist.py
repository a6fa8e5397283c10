"""The overlapped two-bucket gradient all-reduce on the GPU's streams, in a one-rank RCCL group (the box has one GPU;
the world_size-2 logic is covered on CPU by tests/test_dist_gloo.py): a SUM over one rank is the identity, so two
optimiser steps with the collectives switched on must leave bit-identical parameters."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from oracle import cpc_oracle as O

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_bucket_allreduce_on_side_stream_single_rank():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    dev = torch.device("cuda:0")
    B = 4
    p = O.make_params(seed=14, head_scale=64.0)
    wave = O.make_waveform(B, 20480, seed=24).to(dev)
    label = torch.zeros(B, dtype=torch.long, device=dev)
    g = torch.Generator().manual_seed(8)
    draws = [O.draw_negative_indices(B, 128, 116, 128, generator=g) for _ in range(3)]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        finals = []
        for on in (False, True):
            model, crit = build_model().to(dev), build_criterion().to(dev)
            load_flat_params(model, crit, p)
            model.train(); crit.train()
            tr = Trainer(model, crit)
            tr.allreduce.single_rank_too = on
            assert tr.allreduce.early and tr.allreduce.late
            assert {id(q) for q in tr.allreduce.late} == {id(q) for q in model.gEncoder.parameters()}
            for bidx, sidx in draws:
                tr.step(wave, label, negatives=(bidx.to(dev), sidx.to(dev)))
            torch.cuda.synchronize()
            assert tr.allreduce._pending is None
            assert (tr.allreduce.buf is not None) == on
            finals.append({k: v.detach().cpu() for k, v in list(model.state_dict().items())
                           + list(crit.state_dict().items())})
        for k in finals[0]:
            assert torch.equal(finals[0][k], finals[1][k]), k
    finally:
        dist.destroy_process_group()
