"""The BASELINE.json configurations at their real sizes on a MI355X:

* configs[1]  B = 64 x 20480: the whole train step (forward, ``sum().backward()``) against the CPU oracle, single-stream
  and with every stream overlap on (bit-identical to each other);
* configs[2]  B = 256 per GPU: a different code path for the recurrence (more workgroups than can be co-resident), 32,768
  destination rows in the index preparation, 4.3 GB of candidate rows.  The oracle cannot run B = 256 in reasonable
  time, so the batch is built from four self-contained 64-sequence slices (every sequence draws its negatives inside its
  own slice): the B = 256 loss must be the mean of the four B = 64 losses and every gradient the mean of the four B = 64
  gradients (the criterion averages over B*W rows) -- and slice 0 IS the B = 64 case checked against the oracle above;
* configs[4]  ``--samplingType sequential`` (keepHidden): three consecutive optimiser steps carrying the GRU state,
  against the oracle carrying it the same way (cpc/model.py:193-198, cpc/feature_loader.py:149);
* criterion mode 'reverse' through the overlapped Trainer (a torch.flip sits between the criterion and the encoder).
"""
import os

import pytest
import torch

from oracle import cpc_oracle as O

pytestmark = pytest.mark.gpu

S, K, N, W = 128, 12, 128, 116


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    return torch.device("cuda:0")


def _rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _sliced_draws(B, slice_size, seed):
    """The two draws of sampleClean (flat (b,n,t) order) with batchIdx confined to the drawing sequence's own slice."""
    g = torch.Generator().manual_seed(seed)
    local = torch.randint(0, slice_size, (B, N, W), generator=g)
    base = (torch.arange(B) // slice_size * slice_size).view(B, 1, 1)
    sidx = torch.randint(1, S, (B * N * W,), generator=g)
    return (local + base).reshape(-1), sidx, local


def _step(model, crit, wave, bidx, sidx, dev, overlap, keep_masks=False):
    from cpc_audio_amd import ops
    for q in list(model.parameters()) + list(crit.parameters()):
        q.grad = None
    B = wave.shape[0]
    ops.KEEP_DEBUG = keep_masks
    try:
        with ops.StepContext(overlap=overlap) as sc:
            c, z, _ = model(wave, torch.zeros(B, dtype=torch.long, device=dev))
            masks = None
            if keep_masks:
                saved, sizes, zz = ops.debug_last["encoder"]
                masks = [(yi > 0).permute(0, 2, 1).cpu() for yi in ops.saved_encoder_activations(saved, B, wave.shape[2])] \
                    + [(zz > 0).permute(0, 2, 1).cpu()]
            losses, acc = crit(c, z, None, negatives=(bidx, sidx))
            torch.autograd.backward([losses], [torch.ones_like(losses)])
            sc.wait()
    finally:
        ops.KEEP_DEBUG = False
    torch.cuda.synchronize()
    grads = {k: v.grad.clone() for k, v in list(model.state_dict(keep_vars=True).items())
             + list(crit.state_dict(keep_vars=True).items())}
    return dict(c=c.detach(), z=z.detach(), losses=losses.detach(), acc=acc.detach(), grads=grads, masks=masks)


@pytest.fixture(scope="module")
def world():
    dev = _dev()
    from cpc_audio_amd.train import build_criterion, build_model, load_flat_params
    p = O.make_params(seed=21, head_scale=64.0)
    model, crit = build_model().to(dev), build_criterion().to(dev)
    load_flat_params(model, crit, p)
    model.train(); crit.train()
    wave = O.make_waveform(256, 20480, seed=77)
    bidx, sidx, local = _sliced_draws(256, 64, seed=5)
    return dict(dev=dev, p=p, model=model, crit=crit, wave=wave, bidx=bidx, sidx=sidx, local=local)


def _slice(world, j):
    n = 64 * N * W
    return (world["wave"][64 * j:64 * (j + 1)].to(world["dev"]),
            world["local"][64 * j:64 * (j + 1)].reshape(-1).to(world["dev"]),
            world["sidx"][n * j:n * (j + 1)].to(world["dev"]))


def test_config2_full_step_b64_matches_oracle(world):
    """B = 64 x 20480 (the benchmark workload): z, c, losses within the north-star 1e-4 of the CPU oracle, every
    parameter gradient within 2e-4 relative; the overlapped step is bit-identical to the single-stream one."""
    dev, model, crit = world["dev"], world["model"], world["crit"]
    wave, bidx, sidx = _slice(world, 0)
    single = _step(model, crit, wave, bidx, sidx, dev, overlap=False, keep_masks=True)
    over = _step(model, crit, wave, bidx, sidx, dev, overlap=True)
    assert torch.equal(single["z"], over["z"]) and torch.equal(single["c"], over["c"])
    assert torch.equal(single["losses"], over["losses"])
    for k in single["grads"]:
        assert torch.equal(single["grads"][k], over["grads"][k]), k
    O.tie_report()
    ora = O.train_step(world["p"], wave.cpu(), bidx.cpu(), sidx.cpu(), relu_override=single["masks"])
    ties = O.tie_report()                    # 64 x (4096 + 1024 + 512 + 256 + 128) x 256 = 98.6 M activations
    print(f"relu ties [B=64]: {ties}")
    assert O.tie_ok(ties) and ties["disagree_outside"] == 0, ties
    assert (single["z"].cpu() - ora["z"]).abs().max().item() < 1e-4
    assert (single["c"].cpu() - ora["c"]).abs().max().item() < 1e-4
    assert (single["losses"].cpu() - ora["losses"]).abs().max().item() < 1e-4
    assert (single["acc"].cpu() - ora["acc"]).abs().max().item() <= 2.0 / (W * 64) + 1e-7
    bad = {k: _rel(single["grads"][k].cpu(), g) for k, g in ora["grads"].items()}
    bad = {k: v for k, v in bad.items() if not v < 2e-4}
    assert not bad, bad


def test_config3_b256_is_the_mean_of_its_self_contained_slices(world):
    """B = 256 per GPU (BASELINE configs[2]) through the overlapped path the train loops use."""
    dev, model, crit = world["dev"], world["model"], world["crit"]
    full = _step(model, crit, world["wave"].to(dev), world["bidx"].to(dev), world["sidx"].to(dev), dev, overlap=True)
    assert all(torch.isfinite(v).all() for v in full["grads"].values())
    again = _step(model, crit, world["wave"].to(dev), world["bidx"].to(dev), world["sidx"].to(dev), dev, overlap=False)
    for k in full["grads"]:                                   # single-stream == overlapped, bit for bit, at this size too
        assert torch.equal(full["grads"][k], again["grads"][k]), k
    loss_sum, grad_sum = 0.0, None
    for j in range(4):
        wave, bidx, sidx = _slice(world, j)
        part = _step(model, crit, wave, bidx, sidx, dev, overlap=True)
        # per-sequence outputs do not depend on which batch the sequence sits in
        assert (full["z"][64 * j:64 * (j + 1)] - part["z"]).abs().max().item() < 1e-5
        assert (full["c"][64 * j:64 * (j + 1)] - part["c"]).abs().max().item() < 1e-5
        loss_sum = loss_sum + part["losses"].double()
        grad_sum = {k: v.double() for k, v in part["grads"].items()} if grad_sum is None else \
            {k: grad_sum[k] + v.double() for k, v in part["grads"].items()}
    assert (full["losses"].double() - loss_sum / 4).abs().max().item() < 2e-5
    bad = {k: _rel(full["grads"][k].double(), grad_sum[k] / 4) for k in grad_sum}
    # every gradient but one agrees to < 8e-5.  conv0.weight -- 2560 values, each a sum with heavy cancellation over the 16.8
    # million rows of layer 0 whose blocks group differently at the two sizes, behind ReLU masks that flip wherever the two runs
    # round a near-zero pre-activation differently -- was measured at 1.9e-4 / 2.6e-4 / 2.3e-4 / 2.3e-4 for the two-stage walk, the
    # tap-pair walk, and either of them with the K-walk rotation off (which makes layer 1's summation order independent of where a
    # sequence sits in the batch): the deviation is a property of the quantity, not of a kernel, so it gets its own bound.
    bad = {k: v for k, v in bad.items() if not v < (6e-4 if k.endswith("conv0.weight") else 2e-4)}
    assert not bad, bad


def test_config5_sequential_sampling_three_steps_carry_hidden_state():
    """keepHidden=True (what --samplingType sequential switches on, feature_loader.py:149): the final GRU state of
    step i, detached, is the initial state of step i+1; parameters move by Adam in between."""
    dev = _dev()
    from cpc_audio_amd import ops
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    B = 4
    p = O.make_params(seed=31, head_scale=64.0)
    model, crit = build_model(keepHidden=True).to(dev), build_criterion().to(dev)
    load_flat_params(model, crit, p)
    model.train(); crit.train()
    tr = Trainer(model, crit)
    cpu = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.Adam(list(cpu.values()), lr=2e-4, betas=(0.9, 0.999), eps=1e-8)
    g = torch.Generator().manual_seed(9)
    h = None
    for i in range(3):
        wave = O.make_waveform(B, 20480, seed=100 + i)          # consecutive windows of the same B streams
        bi, si = O.draw_negative_indices(B, S, W, N, generator=g)
        ops.KEEP_DEBUG = True                                    # keeps the forward's saved activations: the ReLU masks
        try:
            losses, _ = tr.step(wave.to(dev), None, negatives=(bi.to(dev), si.to(dev)))
        finally:
            ops.KEEP_DEBUG = False
        # ReLU derivative of numerically tied pre-activations follows the device: a single tie of the first
        # layers moves conv0's gradients by ~1e-3, which Adam's sign-like first steps turn into whole-tensor differences --
        # whether the CPU's rounding produces a tie where the GPU's does not depends on the host (thread count, blocking)
        saved, sizes, zz = ops.debug_last["encoder"]
        masks = [(y.cpu() > 0).permute(0, 2, 1) for y in ops.saved_encoder_activations(saved, B, 20480)] + \
                [(zz.cpu() > 0).permute(0, 2, 1)]
        O.tie_report()
        ora = O.train_step({k: v.detach() for k, v in cpu.items()}, wave, bi, si, h0=h, relu_override=masks)
        ties = O.tie_report()
        # (from step 1 on the two parameter trajectories differ by Adam's sign-like steps, so device and oracle masks may also
        # disagree OUTSIDE the window; on identical parameters -- step 0 -- they may not)
        print(f"relu ties [keepHidden step {i}]: {ties}")
        assert O.tie_ok(ties) and (i > 0 or ties["disagree_outside"] == 0), (i, ties)
        h = ora["hN"]
        assert (losses.cpu() - ora["losses"]).abs().max().item() < 1e-4, i
        assert model.gAR.hidden is not None and not model.gAR.hidden.requires_grad
        # step 0 runs on identical parameters; afterwards the two Adam trajectories differ by up to ~lr per step in a few
        # weights (sign flips of ~0 gradients), which the carried state sees through 128 recurrent steps each (measured 4e-4)
        assert (model.gAR.hidden.cpu() - h).abs().max().item() < (1e-4 if i == 0 else 3e-3), i
        for k, v in cpu.items():
            v.grad = ora["grads"][k]
        opt.step()
    new = dict(model.state_dict())
    new.update(crit.state_dict())
    # three Adam steps of lr 2e-4: every weight within 3 * lr of the CPU trajectory (+ sign flips of ~0 gradients)
    worst = max((new[k].cpu() - cpu[k].detach()).abs().max().item() for k in cpu)
    assert worst <= 3 * 4.1e-4, worst
    frac_far = max(((new[k].cpu() - cpu[k].detach()).abs() > 2e-6).float().mean().item() for k in cpu)
    assert frac_far < 2e-3, frac_far


def test_reverse_mode_train_step_through_the_overlapped_trainer_matches_oracle():
    """cpc_mode 'reverse' (criterion.py:227-229; CPCAR reverse=True): torch.flip sits between the criterion and the
    encoder, so dz must exist when the criterion's backward returns -- the overlapped Trainer must not defer it."""
    dev = _dev()
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    B = 3
    p = O.make_params(seed=41, head_scale=64.0)
    model, crit = build_model(reverse=True).to(dev), build_criterion(mode="reverse").to(dev)
    load_flat_params(model, crit, p)
    model.train(); crit.train()
    tr = Trainer(model, crit, lr=0.0)                          # lr 0: parameters stay, gradients are what is checked
    tr.optimizer.zero_grad = lambda *a, **k: None              # keep .grad for inspection
    wave = O.make_waveform(B, 20480, seed=55)
    g = torch.Generator().manual_seed(10)
    bi, si = O.draw_negative_indices(B, S, W, N, generator=g)
    for q in list(model.parameters()) + list(crit.parameters()):
        q.grad = None
    losses, _ = tr.step(wave.to(dev), None, negatives=(bi.to(dev), si.to(dev)))
    torch.cuda.synchronize()
    # the oracle of the same computation: encoder, GRU over flipped time, criterion on flipped c and z
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    z = O.encoder_forward(leaves, wave).permute(0, 2, 1)
    c, _ = O.gru_forward(leaves, torch.flip(z, [1]))
    c = torch.flip(c, [1])
    ext = O.negative_rows(bi, si, B, S, W, N)
    lr_, _ = O.criterion_forward(leaves, torch.flip(c, [1]), torch.flip(z, [1]), ext)
    lr_.sum().backward()
    assert (losses.cpu() - lr_.detach()).abs().max().item() < 1e-4
    got = dict(model.state_dict(keep_vars=True))
    got.update(crit.state_dict(keep_vars=True))
    bad = {k: _rel(got[k].grad.cpu(), v.grad) for k, v in leaves.items()}
    bad = {k: v for k, v in bad.items() if not v < 5e-3}       # loose on the encoder: ReLU ties
    assert not bad, bad
    for k in ("gAR.baseNet.weight_hh_l0", "wPrediction.predictors.3.weight", "gEncoder.conv4.weight"):
        assert _rel(got[k].grad.cpu(), leaves[k].grad) < 2e-4, k


def test_a_systematically_misrounding_epilogue_is_caught_by_the_tie_accounting():
    """The gradient-parity tests hand the device's ReLU mask to the oracle for pre-activations within 1e-5 of zero and assert how
    OFTEN that happened (oracle.tie_report: two correct fp32 paths disagree about one element per million).  A kernel that
    rounds every such pre-activation to one side stays inside every output tolerance (it moves values by < 1e-5) -- this test
    runs exactly that kernel, the test-only build lib/libcpc_hip_misround.so (-DCPC_TEST_MISROUND, cpc_common.h), on the B = 64
    encoder forward and requires the accounting to reject it, layer by layer where the layer is large enough to tell."""
    dev = _dev()
    import ctypes
    from cpc_audio_amd import _lib
    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libcpc_hip_misround.so")
    if not os.path.exists(path):
        pytest.fail("lib/libcpc_hip_misround.so missing: __graft_entry__.build() builds it")
    B, L = 64, 20480
    p = O.make_params(seed=41)
    wave = O.make_waveform(B, L, seed=141)
    names = [f"gEncoder.{n}{i}.{w}" for i in range(5)
             for n, w in (("conv", "weight"), ("conv", "bias"), ("batchNorm", "weight"), ("batchNorm", "bias"))]
    plist = [p[n].to(dev).contiguous() for n in names]
    wd = wave.view(B, L).to(dev).contiguous()
    reports = {}
    for tag, lib in (("product", _lib.get()), ("misround", _lib.bind(path))):
        sizes = (ctypes.c_long * 22)()
        lib.check(lib.cpc_encoder_layout(B, L, sizes), "encoder_layout")
        saved, fscr = torch.empty(sizes[0], device=dev), torch.empty(max(1, sizes[1]), device=dev)
        Ls = [sizes[3 + i] for i in range(5)]
        z = torch.empty(B, Ls[4], 256, device=dev)
        parr = (ctypes.c_void_p * 20)(*[t.data_ptr() for t in plist])
        st = torch.cuda.current_stream().cuda_stream
        lib.check(lib.cpc_encoder_forward(wd.data_ptr(), parr, saved.data_ptr(), fscr.data_ptr(), z.data_ptr(), B, L, st), "encoder_forward")
        ys = []
        for i in range(4):
            y = torch.empty(B, Ls[i], 256, device=dev)
            lib.check(lib.cpc_encoder_saved_activation(saved.data_ptr(), i, y.data_ptr(), B, L, st), "saved_activation")
            ys.append(y.cpu())
        ys.append(z.cpu())
        torch.cuda.synchronize()
        O.tie_report()
        with torch.no_grad():
            zr = O.encoder_forward({k: v for k, v in p.items() if k.startswith("gEncoder")}, wave,
                                   relu_override=[(y > 0).permute(0, 2, 1) for y in ys]).permute(0, 2, 1)
        per_layer = [dict(e) for e in O.TIE_LOG]
        total = O.tie_report()
        assert (ys[4] - zr).abs().max().item() < 1e-4, tag              # both builds are inside the output tolerance
        reports[tag] = (total, per_layer)
        print(f"{tag}: total {total}; per layer overridden {[e['overridden'] for e in per_layer]} of eligible "
              f"{[e['eligible'] for e in per_layer]}")
    total, per_layer = reports["product"]
    assert O.tie_ok(total) and total["disagree_outside"] == 0 and all(O.tie_ok(e) for e in per_layer), reports["product"]
    total, per_layer = reports["misround"]
    # every eligible element with a non-positive oracle value now takes the override: about half the window, ~4 per million
    assert not O.tie_ok(total), total
    assert not O.tie_ok(per_layer[0]) and not O.tie_ok(per_layer[1]), per_layer
