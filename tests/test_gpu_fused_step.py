"""train.Trainer's composite step (cpc_train_step, csrc/train_step.hip: forward + backward of the north-star configuration issued
by ONE C call on the trainer's four streams, gradients written into the flat all-reduce buffer) on a real MI355X: bit-identical
to the autograd-driven step it replaces -- same kernels in the same order -- over several optimiser steps, for drawn and for
caller-supplied negatives, with the recurrent state carried (BASELINE config 5), and in the two-call form a data-parallel rank
uses (reference semantics: cpc/train.py:78-99)."""
import os
import socket
import time

import pytest
import torch

from oracle import cpc_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    return torch.device("cuda:0")


def _trainer(p, dev, fused, keepHidden=False):
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    model, crit = build_model(keepHidden=keepHidden).to(dev), build_criterion().to(dev)
    load_flat_params(model, crit, p)
    model.train(); crit.train()
    return Trainer(model, crit, fused=fused), model, crit


def _state(model, crit):
    return {k: v.detach().cpu().clone() for k, v in list(model.state_dict().items()) + list(crit.state_dict().items())}


@pytest.mark.parametrize("given,keepHidden", [(True, False), (False, False), (True, True)])
def test_composite_step_equals_the_autograd_step_bit_for_bit(given, keepHidden):
    dev = _dev()
    B, steps = 4, 3
    p = O.make_params(seed=31, head_scale=64.0)
    waves = [O.make_waveform(B, 20480, seed=40 + i).to(dev) for i in range(steps)]
    label = torch.zeros(B, dtype=torch.long, device=dev)
    g = torch.Generator().manual_seed(17)
    draws = [O.draw_negative_indices(B, 128, 116, 128, generator=g) for _ in range(steps)]
    res = []
    for fused in (False, True):
        tr, model, crit = _trainer(p, dev, fused, keepHidden)
        torch.manual_seed(123)                       # drawn negatives: torch's generator, consumed in the same order by both
        losses = []
        for i in range(steps):
            neg = (draws[i][0].to(dev), draws[i][1].to(dev)) if given else None
            l, a = tr.step(waves[i], label, negatives=neg)
            losses.append(torch.cat([l, a]).cpu())
        torch.cuda.synchronize()
        assert (tr._fused is not None) == fused
        from cpc_audio_amd import ops
        ops.check_device_errors()
        hid = None if not keepHidden else model.gAR.hidden.cpu()
        res.append((torch.stack(losses), _state(model, crit), hid, tr.optimizer.state_dict()))
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
        assert not torch.equal(res[1][1][k], p[k]), k                  # every tensor took the optimiser steps
    if keepHidden:
        assert torch.equal(res[0][2], res[1][2])
    s0, s1 = res[0][3]["state"], res[1][3]["state"]
    for i in s0:                                                       # optimiser state incl. the step counts (fast path of optim.Adam)
        assert float(s0[i]["step"]) == float(s1[i]["step"]) == steps
        assert torch.equal(s0[i]["exp_avg"], s1[i]["exp_avg"]) and torch.equal(s0[i]["exp_avg_sq"], s1[i]["exp_avg_sq"])


def test_composite_step_first_step_matches_the_oracle():
    dev = _dev()
    B = 3
    p = O.make_params(seed=32, head_scale=64.0)
    wave = O.make_waveform(B, 20480, seed=50)
    g = torch.Generator().manual_seed(18)
    bidx, sidx = O.draw_negative_indices(B, 128, 116, 128, generator=g)
    tr, model, crit = _trainer(p, dev, True)
    tr.optimizer.step = lambda *a, **k: None              # keep the gradients and the parameters of step 0
    tr.optimizer.zero_grad = lambda *a, **k: None
    l, a = tr.step(wave.to(dev), torch.zeros(B, dtype=torch.long, device=dev), negatives=(bidx.to(dev), sidx.to(dev)))
    torch.cuda.synchronize()
    ora = O.train_step(p, wave, bidx, sidx)
    assert (l.cpu() - ora["losses"]).abs().max().item() < 1e-4
    assert (a.cpu() - ora["acc"]).abs().max().item() < 1.5 / (B * 116)
    named = dict(model.state_dict(keep_vars=True))
    named.update(crit.state_dict(keep_vars=True))
    bad = {}
    for k, ref in ora["grads"].items():
        rel = ((named[k].grad.cpu() - ref).norm() / (ref.norm() + 1e-30)).item()
        if not rel < (5e-3 if k.startswith("gEncoder") else 2e-4):     # encoder: a ReLU tie may flip a row
            bad[k] = rel
    assert not bad, bad


def test_composite_step_in_two_phases_around_the_early_gradient_bucket():
    """world_size > 1 issues the step as phase 1, early bucket, phase 2, late bucket.  In a one-rank RCCL group (a SUM over one
    rank is the identity) the trajectory must equal the single-call step's, bit for bit."""
    dev = _dev()
    import torch.distributed as dist
    B, steps = 4, 3
    p = O.make_params(seed=33, head_scale=64.0)
    wave = O.make_waveform(B, 20480, seed=60).to(dev)
    label = torch.zeros(B, dtype=torch.long, device=dev)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        finals = []
        for on in (False, True):
            tr, model, crit = _trainer(p, dev, True)
            tr.allreduce.single_rank_too = on
            torch.manual_seed(7)
            for _ in range(steps):
                tr.step(wave, label)
            torch.cuda.synchronize()
            assert tr._fused is not None and tr.allreduce._pending is None
            finals.append(_state(model, crit))
        for k in finals[0]:
            assert torch.equal(finals[0][k], finals[1][k]), k
    finally:
        dist.destroy_process_group()


def test_composite_step_costs_the_host_a_fraction_of_the_autograd_step():
    """What the composite is for: host time to ENQUEUE a step (B = 64, the benchmark size).  The bound is loose (a busy box), the
    ratio is the claim: the autograd-driven step needs 1.4-1.8 ms from Python on a quiet box."""
    dev = _dev()
    B = 64
    p = O.make_params(seed=34)
    wave = O.make_waveform(B, 20480, seed=70).to(dev)
    label = torch.zeros(B, dtype=torch.long, device=dev)
    host = {}
    for fused in (False, True):
        tr, _, _ = _trainer(p, dev, fused)
        for _ in range(6):
            tr.step(wave, label)
        torch.cuda.synchronize()
        tr.wait_seconds = 0.0
        t0 = time.perf_counter()
        for _ in range(20):
            tr.step(wave, label)
        host[fused] = (time.perf_counter() - t0 - tr.wait_seconds) / 20
        torch.cuda.synchronize()
    print(f"host enqueue per step: autograd {1e3 * host[False]:.3f} ms, composite {1e3 * host[True]:.3f} ms")
    assert host[True] < 0.6 * host[False]
    assert host[True] < 1.0e-3


def test_a_hidden_state_assigned_from_outside_drops_the_a_priori_bound():
    """ADVICE (round 3): |c| <= 1 is taken a priori only for a recurrence started from zero or from its OWN final state.  A state
    assigned from outside with |h| >> 1 makes |c| large; both step paths must then reduce max|c| in line and still match the oracle
    (with the a-priori scale 2^13 such a c would overflow the fp16 pieces)."""
    dev = _dev()
    B = 2
    p = O.make_params(seed=35, head_scale=64.0)
    wave = O.make_waveform(B, 20480, seed=80)
    g = torch.Generator().manual_seed(19)
    bidx, sidx = O.draw_negative_indices(B, 128, 116, 128, generator=g)
    h0 = 40.0 * torch.randn(2, B, 256, generator=g)
    ora = O.train_step(p, wave, bidx, sidx, h0=h0)
    assert ora["c"].abs().max().item() > 8.0
    for fused in (False, True):
        tr, model, crit = _trainer(p, dev, fused, keepHidden=True)
        model.gAR.hidden = h0.to(dev)
        l, _ = tr.step(wave.to(dev), torch.zeros(B, dtype=torch.long, device=dev), negatives=(bidx.to(dev), sidx.to(dev)))
        torch.cuda.synchronize()
        assert torch.isfinite(l).all()
        assert (l.cpu() - ora["losses"]).abs().max().item() < 2e-4 * max(1.0, ora["losses"].abs().max().item()), fused


@pytest.mark.parametrize("B", [128, 256])
def test_composite_step_equals_the_autograd_step_at_the_large_per_gpu_batches(B):
    """BASELINE configs[2] is B = 256 per GPU: the recurrence runs as two persistent launches over 8 batch tiles each, layer 2 moves
    to the DMA-fed kernels (from B ~ 100 on), the weight-gradient plans change.  Two optimiser steps, composite against autograd:
    bit for bit."""
    dev = _dev()
    p = O.make_params(seed=36, head_scale=64.0)
    wave = O.make_waveform(B, 20480, seed=90).to(dev)
    label = torch.zeros(B, dtype=torch.long, device=dev)
    res = []
    for fused in (False, True):
        tr, model, crit = _trainer(p, dev, fused)
        torch.manual_seed(321)
        losses = [torch.cat(tr.step(wave, label)).cpu() for _ in range(2)]
        torch.cuda.synchronize()
        from cpc_audio_amd import ops
        ops.check_device_errors()
        res.append((torch.stack(losses), _state(model, crit)))
        del tr, model, crit
        torch.cuda.empty_cache()
    assert torch.isfinite(res[0][0]).all() and torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


@pytest.mark.parametrize("B,steps", [(4, 6), (64, 12)])
def test_open_tailed_steps_equal_the_closed_steps_bit_for_bit(B, steps):
    """Trainer(pipeline_tail=True): the next step's conv0 starts under this step's last kernel (layer 1's weight gradient), the
    optimiser's update is split over two streams, the weight layouts are prepared at the tail for the other parity of the
    workspace (train.CompositeStep.finish, cpc_train_step_tail).  The same kernels on the same values: every loss of the
    trajectory, every parameter and the optimiser state after it equal the closed-tail run's bit for bit -- at B = 64 with the
    tail genuinely overlapping the next step (a missing cross-stream dependency would show as a differing bit)."""
    dev = _dev()
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    p = O.make_params(seed=37, head_scale=64.0)
    waves = [O.make_waveform(B, 20480, seed=95 + (i % 3)).to(dev) for i in range(steps)]
    label = torch.zeros(B, dtype=torch.long, device=dev)
    res = []
    for pipe in (False, True):
        model, crit = build_model().to(dev), build_criterion().to(dev)
        load_flat_params(model, crit, p)
        tr = Trainer(model, crit, pipeline_tail=pipe)
        torch.manual_seed(5)
        losses = []
        for i in range(steps):
            l, a = tr.step(waves[i], label)
            losses.append(torch.cat([l, a]))
            if i == steps // 2:
                # a parameter changed in place by torch between two steps (a scheduler, a load_state_dict): the prepared layouts
                # are stale -- detected through the version counter; the step joins and prepares at its head
                tr.join()
                with torch.no_grad():
                    model.gEncoder.conv1.weight.mul_(1.0009765625)
                    model.gEncoder.batchNorm0.weight.mul_(1.03125)
        f = tr._fused
        assert f is not None and (f["ready"] is not None) == pipe          # (the tail really was open: layouts prepared for the next step)
        tr.join()
        assert f["ready"] is None        # whoever joined may change parameters through raw pointers: the layouts are not trusted afterwards
        torch.cuda.synchronize()
        from cpc_audio_amd import ops
        ops.check_device_errors()
        res.append((torch.stack(losses).cpu(), _state(model, crit), tr.optimizer.state_dict()))
        del tr, model, crit
    assert torch.isfinite(res[0][0]).all() and torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
    s0, s1 = res[0][2]["state"], res[1][2]["state"]
    for i in s0:
        assert float(s0[i]["step"]) == float(s1[i]["step"]) == steps
        assert torch.equal(s0[i]["exp_avg"], s1[i]["exp_avg"]) and torch.equal(s0[i]["exp_avg_sq"], s1[i]["exp_avg_sq"])


def test_an_autograd_step_between_open_tailed_composite_steps_does_not_leave_stale_weight_layouts():
    """Round-5 advice: an open-tailed composite step prepares the NEXT step's weight layouts at its tail and marks them ready by
    torch's version counters -- which this package's Adam (raw pointers) never bumps.  If the next step takes the autograd path
    (here: ops.KEEP_DEBUG for one step) and updates every weight, a following composite step must not run on the layouts prepared
    two updates ago.  join() now drops them; the trajectory equals the closed-tail Trainer's bit for bit."""
    dev = _dev()
    from cpc_audio_amd import ops
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    B, steps = 4, 6
    p = O.make_params(seed=39, head_scale=64.0)
    waves = [O.make_waveform(B, 20480, seed=80 + i).to(dev) for i in range(steps)]
    label = torch.zeros(B, dtype=torch.long, device=dev)
    res = []
    for pipe in (False, True):
        model, crit = build_model().to(dev), build_criterion().to(dev)
        load_flat_params(model, crit, p)
        tr = Trainer(model, crit, lr=2e-3, pipeline_tail=pipe)               # (a large step: stale layouts move the losses visibly)
        torch.manual_seed(6)
        losses = []
        for i in range(steps):
            ops.KEEP_DEBUG = i == 2                                         # step 2 on the autograd path (CompositeStep.ok() refuses)
            try:
                l, a = tr.step(waves[i], label)
            finally:
                ops.KEEP_DEBUG = False
            losses.append(torch.cat([l, a]))
        tr.join()
        torch.cuda.synchronize()
        ops.check_device_errors()
        res.append((torch.stack(losses).cpu(), _state(model, crit)))
        del tr, model, crit
    assert torch.isfinite(res[0][0]).all() and torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


def test_open_tailed_steps_in_the_bf16_storage_variant_and_across_a_mode_switch():
    """The same in mode 4 (other kernels read y0 / the bounds), then back to fp32 in the same Trainer: the switch rebuilds the
    workspace, which must first join the tail still running in the old one."""
    dev = _dev()
    import cpc_audio_amd
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    B, steps = 16, 4
    p = O.make_params(seed=38, head_scale=64.0)
    wave = O.make_waveform(B, 20480, seed=99).to(dev)
    label = torch.zeros(B, dtype=torch.long, device=dev)
    res = []
    for pipe in (False, True):
        model, crit = build_model().to(dev), build_criterion().to(dev)
        load_flat_params(model, crit, p)
        tr = Trainer(model, crit, pipeline_tail=pipe)
        torch.manual_seed(6)
        losses = []
        try:
            for mode in ("bf16", "fp32"):
                cpc_audio_amd.set_activation_storage(mode)
                for _ in range(steps):
                    losses.append(torch.cat(tr.step(wave, label)))
        finally:
            cpc_audio_amd.set_activation_storage("fp32")
        tr.join()
        torch.cuda.synchronize()
        res.append((torch.stack(losses).cpu(), _state(model, crit)))
    assert torch.isfinite(res[0][0]).all() and torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


def test_whole_step_with_100_negatives_matches_the_oracle_on_both_step_paths():
    """negativeSamplingExt = 100 (criterion.py:176-189 draws any number; not a multiple of the kernels' 16-wide candidate tile):
    the lists are padded to 112 and the padding masked by the scoring kernels.  First step against the oracle run with exactly
    100 negatives (losses, accuracies, every gradient), composite and autograd paths bit-identical over three steps."""
    dev = _dev()
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    B, N, steps = 3, 100, 3
    p = O.make_params(seed=39, head_scale=64.0)
    wave = O.make_waveform(B, 20480, seed=111)
    label = torch.zeros(B, dtype=torch.long, device=dev)
    g = torch.Generator().manual_seed(23)
    draws = [O.draw_negative_indices(B, 128, 116, N, generator=g) for _ in range(steps)]
    ora = O.train_step(p, wave, draws[0][0], draws[0][1], n_neg=N)
    res = []
    for fused in (False, True):
        model, crit = build_model().to(dev), build_criterion(negativeSamplingExt=N).to(dev)
        load_flat_params(model, crit, p)
        tr = Trainer(model, crit, fused=fused)
        opt_step, tr.optimizer.step = tr.optimizer.step, (lambda *a, **k: None)       # keep the gradients of step 0
        zero, tr.optimizer.zero_grad = tr.optimizer.zero_grad, (lambda *a, **k: None)
        l, a = tr.step(wave.to(dev), label, negatives=(draws[0][0].to(dev), draws[0][1].to(dev)))
        torch.cuda.synchronize()
        assert (tr._fused is not None) == fused
        assert (l.cpu() - ora["losses"]).abs().max().item() < 1e-4
        assert (a.cpu() - ora["acc"]).abs().max().item() < 1.5 / (B * 116)
        named = dict(model.state_dict(keep_vars=True))
        named.update(crit.state_dict(keep_vars=True))
        bad = {}
        for k, ref in ora["grads"].items():
            rel = ((named[k].grad.cpu() - ref).norm() / (ref.norm() + 1e-30)).item()
            if not rel < (5e-3 if k.startswith("gEncoder") else 2e-4):
                bad[k] = rel
        assert not bad, (fused, bad)
        tr.optimizer.step, tr.optimizer.zero_grad = opt_step, zero
        tr.optimizer.zero_grad()
        losses = []
        for i in range(steps):
            l, a = tr.step(wave.to(dev), label, negatives=(draws[i][0].to(dev), draws[i][1].to(dev)))
            losses.append(torch.cat([l, a]).cpu())
        torch.cuda.synchronize()
        res.append((torch.stack(losses), _state(model, crit)))
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


@pytest.mark.parametrize("rnnMode", ["linear", "transformer"])
def test_whole_step_with_20_prediction_steps_matches_the_oracle(rnnMode):
    """nPredicts = 20 (criterion.py:225-257 takes any; the score tiles hold 16 heads): the criterion walks the heads in groups
    (cpc_nce_head_group) -- with 100 negatives on top, so both paddings are in play.  One step through the Trainer (which takes
    the autograd path: the composite step covers K <= 16) against the oracle's 20-head step: losses, accuracies, every gradient;
    with transformer predictors (their predictions handed over as a tensor) losses and the gradients in front of them."""
    dev = _dev()
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    B, N, K = 2, 100, 20
    W = 128 - K
    p = O.make_params(seed=41, n_predicts=K, head_scale=64.0)
    wave = O.make_waveform(B, 20480, seed=113)
    label = torch.zeros(B, dtype=torch.long, device=dev)
    bi, si = O.draw_negative_indices(B, 128, W, N, generator=torch.Generator().manual_seed(29))
    if rnnMode == "linear":
        ora = O.train_step(p, wave, bi, si, n_predicts=K, n_neg=N)
        model, crit = build_model().to(dev), build_criterion(nPredicts=K, negativeSamplingExt=N).to(dev)
        load_flat_params(model, crit, p)
        tr = Trainer(model, crit)
        tr.optimizer.step = tr.optimizer.zero_grad = lambda *a, **k: None           # keep the gradients of the step
        l, a = tr.step(wave.to(dev), label, negatives=(bi.to(dev), si.to(dev)))
        torch.cuda.synchronize()
        assert tr._fused is None
        assert l.shape == (1, K) and (l.cpu() - ora["losses"]).abs().max().item() < 1e-4
        assert (a.cpu() - ora["acc"]).abs().max().item() < 1.5 / (B * W)
        named = dict(model.state_dict(keep_vars=True))
        named.update(crit.state_dict(keep_vars=True))
        bad = {}
        for k, ref in ora["grads"].items():
            rel = ((named[k].grad.cpu() - ref).norm() / (ref.norm() + 1e-30)).item()
            if not rel < (5e-3 if k.startswith("gEncoder") else 2e-4):
                bad[k] = rel
        assert not bad, bad
        return
    from cpc_audio_amd import ops
    from oracle import transformer_oracle as T
    for k in range(K):
        p.pop(f"wPrediction.predictors.{k}.weight")
        p.update(T.make_layer_params(60 + k, 256, W, False, prefix=f"wPrediction.predictors.{k}.0."))
    model = build_model().to(dev)
    crit = build_criterion(nPredicts=K, negativeSamplingExt=N, rnnMode="transformer", transformerDropout=0.0).to(dev)
    model.load_state_dict({k: v for k, v in p.items() if not k.startswith("wPrediction")}, strict=True)
    cm = crit.load_state_dict({k: v for k, v in p.items() if k.startswith("wPrediction")}, strict=False)
    assert not cm.unexpected_keys and all(k.endswith(("Att.z", "Att.mask")) for k in cm.missing_keys), cm
    ops.debug_last.pop("transformer", None)
    ops.KEEP_DEBUG = True
    try:
        c, z, _ = model(wave.to(dev), None)
        losses, acc = crit(c, z, None, negatives=(bi.to(dev), si.to(dev)))
    finally:
        ops.KEEP_DEBUG = False
    # the device's ReLU decisions in the K feed-forward blocks, for the oracle (an element within rounding of zero may fall
    # either way; tests/test_gpu_transformer.py does the same)
    tmasks = [(saved[sizes[7]: sizes[7] + B * W * 2048].view(B, W, 2048) > 0).cpu() for saved, sizes in ops.debug_last["transformer"]]
    assert len(tmasks) == K
    losses.sum().backward()
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    co, zo, _ = O.model_forward(leaves, wave)
    ext = O.negative_rows(bi, si, B, 128, W, N)
    lo, ao = O.criterion_forward(leaves, co, zo, ext, K,
                                 predict=lambda k, cw: T.layer_forward(leaves, cw, prefix=f"wPrediction.predictors.{k}.0.",
                                                                       relu_override=tmasks[k]))
    lo.sum().backward()
    assert (losses.detach().cpu() - lo.detach()).abs().max().item() < 1e-4
    assert (acc.detach().cpu() - ao).abs().max().item() < 1.5 / (B * W)
    rel = lambda a_, b_: ((a_ - b_).norm() / (b_.norm() + 1e-30)).item()
    bad = {}
    for name, v in crit.named_parameters():
        r = rel(v.grad.cpu(), leaves[name].grad)
        if not r < 2e-4:
            bad[name] = r
    assert not bad, bad
    assert rel(model.gAR.baseNet.weight_hh_l1.grad.cpu(), leaves["gAR.baseNet.weight_hh_l1"].grad) < 2e-4
    assert rel(model.gEncoder.conv4.weight.grad.cpu(), leaves["gEncoder.conv4.weight"].grad) < 5e-3
