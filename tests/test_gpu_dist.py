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
            assert tr.allreduce.early and tr.allreduce.mid and tr.allreduce.late
            # three buckets: heads + recurrence | conv2..4 weights | the rest of the encoder (2.1 MB: the exposed collective)
            assert [id(q) for q in tr.allreduce.mid] == [id(getattr(model.gEncoder, f"conv{i}").weight) for i in (2, 3, 4)]
            assert ({id(q) for q in tr.allreduce.late} | {id(q) for q in tr.allreduce.mid}
                    == {id(q) for q in model.gEncoder.parameters()})
            assert 4 * sum(q.numel() for q in tr.allreduce.late) <= 2.2e6
            if on:      # a step that raised after its early bucket went out: abort() drains it, the next step starts clean
                tr.allreduce.begin(tr.ctx)
                assert tr.allreduce._pending is not None and tr.allreduce._pending_event is not None
                tr.allreduce.abort()
                assert tr.allreduce._pending is None and tr.allreduce._pending_event is None
            for bidx, sidx in draws:
                tr.step(wave, label, negatives=(bidx.to(dev), sidx.to(dev)))
            torch.cuda.synchronize()
            assert tr.allreduce._pending is None
            assert tr.allreduce.buf is not None and tr._fused is not None     # the composite step keeps its gradients in the flat buffer
            finals.append({k: v.detach().cpu() for k, v in list(model.state_dict().items())
                           + list(crit.state_dict().items())})
        for k in finals[0]:
            assert torch.equal(finals[0][k], finals[1][k]), k
    finally:
        dist.destroy_process_group()


# ---- world_size 2: the REAL train step on two ranks ------------------------------------------------------------------
_WS2_B, _WS2_STEPS = 4, 2


def _ws2_inputs(rank):
    """Rank ``rank``'s half of the global batch: its own sequences and its own negatives, drawn inside the half
    (criterion.py:176-184 runs per replica: batchIdx in [0, B_local))."""
    waves = [O.make_waveform(_WS2_B, 20480, seed=300 + 10 * i + rank) for i in range(_WS2_STEPS)]
    g = torch.Generator().manual_seed(500 + rank)
    draws = [O.draw_negative_indices(_WS2_B, 128, 116, 128, generator=g) for _ in range(_WS2_STEPS)]
    return waves, draws


def _ws2_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # both ranks on the box's one GPU: RCCL refuses two ranks on one device, so the group is gloo and FlatGradAllReduce stages
    # its two buckets through the host -- same buckets, same order, same stream dependencies as over RCCL
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
        dev = torch.device("cuda:0")
        model, crit = build_model().to(dev), build_criterion().to(dev)
        load_flat_params(model, crit, O.make_params(seed=21, head_scale=64.0))
        model.train(); crit.train()
        tr = Trainer(model, crit)
        assert tr.allreduce._active() and tr.allreduce.early and tr.allreduce.late
        waves, draws = _ws2_inputs(rank)
        label = torch.zeros(_WS2_B, dtype=torch.long, device=dev)
        losses = []
        for wave, (bi, si) in zip(waves, draws):
            l, _ = tr.step(wave.to(dev), label, negatives=(bi.to(dev), si.to(dev)))
            losses.append(l.cpu())
        torch.cuda.synchronize()
        state = {k: v.detach().cpu() for k, v in list(model.state_dict().items()) + list(crit.state_dict().items())}
        torch.save({"state": state, "losses": torch.stack(losses)}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_world_size_2_real_step_equals_one_rank_on_the_summed_gradients(tmp_path):
    """cpc/train.py:83-87,372-375: every replica runs the step on its own sub-batch with negatives from that sub-batch, the
    per-replica losses are SUMMED and one optimiser step follows.  Two processes (one per rank) run Trainer.step on their
    halves with the two-bucket all-reduce live -- early bucket packed on the side stream from the encoder-backward hook,
    behind the head gradient and the recurrence's weight-gradient stream; late bucket after backward -- for two steps.
    Required: (a) both ranks end with bit-identical parameters; (b) they equal, bit for bit, ONE rank that forms g(half 0)
    and g(half 1) in two backward passes, adds them and takes the same Adam steps (a SUM over two ranks is one fp32
    addition per value); (c) the summed gradient of step 0 and each half's losses match the CPU oracle."""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    import torch.multiprocessing as mp
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    world = 2
    mp.spawn(_ws2_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(world)]
    for k in got[0]["state"]:                                                              # (a)
        assert torch.equal(got[0]["state"][k], got[1]["state"][k]), k
    # (b) one rank, two backward passes per step
    dev = torch.device("cuda:0")
    p = O.make_params(seed=21, head_scale=64.0)
    model, crit = build_model().to(dev), build_criterion().to(dev)
    load_flat_params(model, crit, p)
    model.train(); crit.train()
    tr = Trainer(model, crit)
    params = [q for grp in tr.optimizer.param_groups for q in grp["params"]]
    real_step, real_zero = tr.optimizer.step, tr.optimizer.zero_grad
    inputs = [_ws2_inputs(r) for r in range(world)]
    label = torch.zeros(_WS2_B, dtype=torch.long, device=dev)
    named = dict(model.state_dict(keep_vars=True))
    named.update(crit.state_dict(keep_vars=True))
    for i in range(_WS2_STEPS):
        halves = []
        for r in range(world):
            tr.optimizer.step, tr.optimizer.zero_grad = (lambda *a, **k: None), (lambda *a, **k: None)
            for q in params:
                q.grad = None
            bi, si = inputs[r][1][i]
            l, _ = tr.step(inputs[r][0][i].to(dev), label, negatives=(bi.to(dev), si.to(dev)))
            torch.cuda.synchronize()
            assert torch.equal(l.cpu(), got[r]["losses"][i]), (i, r)       # same parameters, same half: same losses, bit for bit
            halves.append([q.grad.clone() for q in params])
        if i == 0:                                                                         # (c)
            for r in range(world):
                ora = O.train_step(p, inputs[r][0][0], *inputs[r][1][0])
                assert (got[r]["losses"][0] - ora["losses"]).abs().max().item() < 1e-4
                halves_ora = ora["grads"] if r == 0 else {k: halves_ora[k] + ora["grads"][k] for k in halves_ora}
            summed = {id(q): a + b for q, a, b in zip(params, *halves)}
            for k in ("gAR.baseNet.weight_hh_l0", "gAR.baseNet.bias_ih_l1", "wPrediction.predictors.3.weight",
                      "gEncoder.conv4.weight", "gEncoder.conv2.weight"):
                mine, ref = summed[id(named[k])].cpu(), halves_ora[k]
                rel = ((mine - ref).norm() / ref.norm()).item()
                assert rel < (5e-3 if "conv2" in k else 2e-4), (k, rel)     # conv2: a ReLU tie may flip a row
        for q, a, b in zip(params, *halves):
            q.grad = a + b
        tr.optimizer.step, tr.optimizer.zero_grad = real_step, real_zero
        tr.optimizer.step()
        tr.optimizer.zero_grad()
    torch.cuda.synchronize()
    mine = {k: v.detach().cpu() for k, v in list(model.state_dict().items()) + list(crit.state_dict().items())}
    moved = 0
    for k in mine:
        assert torch.equal(mine[k], got[0]["state"][k]), k
        moved += int(not torch.equal(mine[k], p[k]))
    assert moved == len(mine)                                 # every tensor took the optimiser steps


# ---- the launcher path the driver's scaling run uses --------------------------------------------------------------------
def _run_bench(args, env_extra, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable] + args, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_through_the_launcher_walks_the_multi_rank_code_paths_on_one_gpu():
    """``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N
    ...`` is how the driver measures N = 2, 4, 8.  A 1-GPU box can host one rank, so the same command runs with N = 1 and
    CPC_BENCH_FORCE_DIST=1, which switches on everything N > 1 adds: ``init_process_group("nccl", device_id=...)``, the barriers
    around the timed region, the three-bucket RCCL all-reduce (early and mid buckets as synchronous collectives of the side stream, behind
    the heads' / the recurrence's / the short conv layers' gradient streams), the MAX over ranks of the elapsed time, graph replay off, the sustained
    run.  The line must be well-formed, its loss must equal the plain single-process run's (a SUM over one rank is the
    identity), and the data-parallel form of the step must not cost more than 12 % over the plain one on the same box (it cost
    16 % while the early bucket went through torch.distributed's internal stream: DESIGN.md section 5a; 1-2 % now)."""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    common = ["bench.py", "--gpus", "1", "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-probes",
              "--sustained-seconds", "1.0", "--launch", "eager"]
    forced = _run_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                         "--master-port", str(_free_port())] + common, {"CPC_BENCH_FORCE_DIST": "1"})
    plain = _run_bench(common + ["--no-pipeline-tail"], {})      # (the closed-tail step: what the data-parallel form is built on)
    for line in (forced, plain):
        assert line["n_gpus"] == 1 and line["steps"] == 6 and line["scaling"] == "weak" and line["higher_is_better"] is True
        assert line["config"]["parallelism"] == "dp1" and line["config"]["global_batch"] == 64
        assert line["value"] > 0 and line["value"] == line["value"] and line["ms_per_step"] > 0
        assert abs(line["value"] - 64 * 1.28 / (line["ms_per_step"] * 1e-3)) < 0.01 * line["value"]
        assert line["sustained"]["steps"] >= 6 and line["sustained"]["ms_per_step"] > 0
        assert line["config"]["launch"].startswith("eager")
    assert "forced_dist" in forced["config"] and "forced_dist" not in plain["config"]
    # the N > 1 line carries what its two gradient all-reduces cost on their own (11.6 MB in two buckets)
    assert "dist" not in plain and forced["dist"]["backend"] == "nccl"
    assert sum(forced["dist"]["bytes"].values()) == 4 * 2893056
    assert set(forced["dist"]["bytes"]) == {"early_bucket", "mid_bucket", "late_bucket_exposed"}
    assert forced["dist"]["bytes"]["mid_bucket"] == 4 * 3 * 256 * 256 * 4               # conv2..4 weights
    assert forced["dist"]["bytes"]["late_bucket_exposed"] <= 2.2e6                       # conv0, conv1, biases / norms
    assert all(v > 0 for v in forced["dist"]["allreduce_ms"].values()), forced["dist"]
    # same seeds, same steps; the all-reduce of one rank adds nothing
    assert forced["config"]["loss_mean_over_heads"] == plain["config"]["loss_mean_over_heads"]
    ratio = forced["sustained"]["ms_per_step"] / plain["sustained"]["ms_per_step"]
    print(f"data-parallel form on one rank: {forced['sustained']['ms_per_step']:.3f} ms/step against {plain['sustained']['ms_per_step']:.3f} "
          f"plain ({ratio:.3f}x)")
    # a wall-clock ratio of two separate processes on a shared box: reported, and only a gross regression fails the suite
    if ratio >= 1.12:
        import warnings
        warnings.warn(f"data-parallel form of the step {ratio:.3f}x the plain one on this box (1.01-1.02x expected)")
    assert ratio < 1.5, (forced["sustained"], plain["sustained"])


def test_bench_self_spawn_refuses_more_ranks_than_gpus():
    """``python bench.py --gpus 2`` on a 1-GPU box: the self-spawning launcher must fail loudly, not print a 1-GPU line."""
    import subprocess
    import sys
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one visible GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and '"metric"' not in r.stdout
    assert "only 1 GPU" in (r.stdout + r.stderr)
