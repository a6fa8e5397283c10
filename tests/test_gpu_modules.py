"""Module-level behaviour on a real MI355X vs the CPU oracle: hidden-state carry (keepHidden /
samplingType=sequential, BASELINE config 5), reverse mode, other K / N / window lengths, and one
complete optimiser step."""
import pytest
import torch

from oracle import cpc_oracle as O

pytestmark = pytest.mark.gpu


def _lib_default_gru_mode():
    from cpc_audio_amd._lib import DEFAULT_GRU_MODE
    return DEFAULT_GRU_MODE


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_gru_hidden_carry_matches_oracle():
    """CPCAR(keepHidden=True): the final state of call 1 seeds call 2 (cpc/model.py:193-198)."""
    dev = _dev()
    from cpc_audio_amd.model import CPCAR
    p = O.make_params(seed=2)
    ar = CPCAR(256, 256, True, 2, mode="GRU").to(dev)
    ar.load_state_dict({k[len("gAR."):]: v for k, v in p.items() if k.startswith("gAR.")})
    g = torch.Generator().manual_seed(0)
    x1, x2 = torch.randn(5, 40, 256, generator=g), torch.randn(5, 40, 256, generator=g)
    y1 = ar(x1.to(dev))
    assert ar.hidden is not None and not ar.hidden.requires_grad
    x2d = x2.to(dev).requires_grad_(True)
    y2 = ar(x2d)
    dy = torch.randn(5, 40, 256, generator=g)
    (y2 * dy.to(dev)).sum().backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith("gAR.")}
    r1, h1 = O.gru_forward(leaves, x1)
    x2r = x2.clone().requires_grad_(True)
    r2, _ = O.gru_forward(leaves, x2r, h0=h1.detach())
    (r2 * dy).sum().backward()
    assert (y1.detach().cpu() - r1).abs().max().item() < 1e-4
    assert (y2.detach().cpu() - r2).abs().max().item() < 1e-4
    assert _rel(x2d.grad.cpu(), x2r.grad) < 1e-4
    for n, prm in ar.baseNet.named_parameters():
        assert _rel(prm.grad.cpu(), leaves["gAR.baseNet." + n].grad) < 1e-4, n


def test_reverse_mode_matches_flipped_oracle():
    """cpc_mode='reverse': CPCAR(reverse=True) and criterion mode='reverse' flip time
    (cpc/model.py:187-188,202-203; criterion.py:227-229)."""
    dev = _dev()
    from cpc_audio_amd.criterion import CPCUnsupersivedCriterion
    from cpc_audio_amd.model import CPCAR
    p = O.make_params(seed=3, head_scale=64.0)
    ar = CPCAR(256, 256, False, 2, mode="GRU", reverse=True).to(dev)
    ar.load_state_dict({k[len("gAR."):]: v for k, v in p.items() if k.startswith("gAR.")})
    crit = CPCUnsupersivedCriterion(12, 256, 256, 128, mode="reverse", rnnMode="linear", sizeInputSeq=128).to(dev)
    crit.load_state_dict({k: v for k, v in p.items() if k.startswith("wPrediction")})
    g = torch.Generator().manual_seed(1)
    B, S = 3, 128
    z = torch.relu(torch.randn(B, S, 256, generator=g))
    c = ar(z.to(dev))
    cr, _ = O.gru_forward(p, torch.flip(z, [1]))
    cr = torch.flip(cr, [1])
    assert (c.cpu() - cr).abs().max().item() < 1e-4
    bi, si = O.draw_negative_indices(B, S, S - 12, 128, generator=g)
    losses, acc = crit(c, z.to(dev), None, negatives=(bi.to(dev), si.to(dev)))
    ext = O.negative_rows(bi, si, B, S, S - 12, 128)
    lr, ar_ = O.criterion_forward(p, torch.flip(cr, [1]), torch.flip(z, [1]), ext)
    assert (losses.cpu() - lr).abs().max().item() < 1e-4
    assert (acc.cpu() - ar_).abs().max().item() <= 2.0 / (116 * B) + 1e-7


@pytest.mark.parametrize("K,N,L", [(5, 256, 10240), (16, 64, 20480), (12, 512, 20480)])
def test_other_heads_negatives_and_window(K, N, L):
    """nPredicts != 12, large-negative stress (BASELINE config 5 sweeps N in {128,256,512}: ~476 candidate slots per
    destination row of the re-associated dz path at N = 512) and a shorter window (S = L/160)."""
    _heads_negatives_window(_dev(), K, N, L)


def _heads_negatives_window(dev, K, N, L):
    from cpc_audio_amd.train import build_criterion, build_model, load_flat_params
    B = 2
    S = L // 160
    p = O.make_params(seed=5, head_scale=128.0, n_predicts=K)
    model = build_model().to(dev)
    crit = build_criterion(nPredicts=K, negativeSamplingExt=N, sizeWindow=L).to(dev)
    load_flat_params(model, crit, p)
    wave = O.make_waveform(B, L, seed=9)
    g = torch.Generator().manual_seed(4)
    bi, si = O.draw_negative_indices(B, S, S - K, N, generator=g)
    c, z, _ = model(wave.to(dev), None)
    losses, acc = crit(c, z, None, negatives=(bi.to(dev), si.to(dev)))
    losses.sum().backward()
    ora = O.train_step(p, wave, bi, si, n_predicts=K, n_neg=N)
    assert (z.detach().cpu() - ora["z"]).abs().max().item() < 1e-4
    assert (c.detach().cpu() - ora["c"]).abs().max().item() < 1e-4
    assert (losses.detach().cpu() - ora["losses"]).abs().max().item() < 1e-4
    for k_ in range(K):
        name = f"wPrediction.predictors.{k_}.weight"
        assert _rel(crit.wPrediction.predictors[k_].weight.grad.cpu(), ora["grads"][name]) < 2e-4, name
    assert _rel(model.gAR.baseNet.weight_hh_l1.grad.cpu(), ora["grads"]["gAR.baseNet.weight_hh_l1"]) < 2e-4
    # dz (criterion -> encoder) reaches the last conv layer's weight gradient directly
    assert _rel(model.gEncoder.conv4.weight.grad.cpu(), ora["grads"]["gEncoder.conv4.weight"]) < 5e-3


def test_trainer_step_updates_parameters_like_cpu_adam():
    """forward + sum().backward() + Adam (cpc/train.py:83-91) vs the same step on the CPU oracle."""
    dev = _dev()
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    B = 2
    p = O.make_params(seed=6, head_scale=64.0)
    model, crit = build_model().to(dev), build_criterion().to(dev)
    load_flat_params(model, crit, p)
    tr = Trainer(model, crit)
    wave = O.make_waveform(B, 20480, seed=3)
    g = torch.Generator().manual_seed(8)
    bi, si = O.draw_negative_indices(B, 128, 116, 128, generator=g)
    losses, _ = tr.step(wave.to(dev), None, negatives=(bi.to(dev), si.to(dev)))
    ora = O.train_step(p, wave, bi, si)
    cpu = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    opt = torch.optim.Adam(list(cpu.values()), lr=2e-4, betas=(0.9, 0.999), eps=1e-8)
    for k, v in cpu.items():
        v.grad = ora["grads"][k]
    opt.step()
    assert (losses.cpu() - ora["losses"]).abs().max().item() < 1e-4
    new = dict(model.state_dict())
    new.update(crit.state_dict())
    worst = max((new[k].cpu() - cpu[k].detach()).abs().max().item() for k in cpu)
    # Adam's first step moves every weight by ~lr; allow a few sign flips of near-zero gradients
    assert worst <= 4.1e-4, worst
    frac_far = max(((new[k].cpu() - cpu[k].detach()).abs() > 1e-6).float().mean().item() for k in cpu)
    assert frac_far < 1e-3, frac_far
    assert all(prm.grad is None or float(prm.grad.abs().sum()) == 0.0 for prm in model.parameters())


@pytest.mark.parametrize("B,S", [(64, 128), (7, 33), (128, 16)])
def test_gru_persistent_launch_equals_stepwise(B, S):
    """The single-launch two-layer recurrence (workgroups hand h_t / gate gradients over through the output
    arrays) and the launch-per-step path share arithmetic and summation order: bit-identical results."""
    dev = _dev()
    from cpc_audio_amd import _lib
    from cpc_audio_amd.model import CPCAR
    lib = _lib.get()
    p = O.make_params(seed=4)
    ar = CPCAR(256, 256, False, 2, mode="GRU").to(dev)
    ar.load_state_dict({k[len("gAR."):]: v for k, v in p.items() if k.startswith("gAR.")})
    g = torch.Generator().manual_seed(B * 1000 + S)
    x = torch.randn(B, S, 256, generator=g).to(dev)
    dy = torch.randn(B, S, 256, generator=g).to(dev)
    outs = []
    for mode in (0, 1, 1):
        assert lib.cpc_set_gru_mode(mode) == 0
        try:
            ar.zero_grad(set_to_none=True)
            xd = x.clone().requires_grad_(True)
            y = ar(xd)
            (y * dy).sum().backward()
            torch.cuda.synchronize()
            outs.append([y.detach().clone(), xd.grad.clone()] + [q.grad.clone() for q in ar.parameters()])
        finally:
            lib.cpc_set_gru_mode(_lib_default_gru_mode())
    assert all(torch.isfinite(t).all() for t in outs[1])
    for a, b, c in zip(*outs):
        assert torch.equal(a, b) and torch.equal(b, c)


@pytest.mark.parametrize("B,S", [(64, 128), (20, 33), (128, 16), (256, 24)])
def test_gru_handover_modes_change_no_bit(B, S):
    """Round 6: how the persistent recurrence hands h / dh over is a matter of memory scope and placement, never of arithmetic --
    plain (L2) first looks (cpc_set_gru_poll_plain), one batch tile per XCD with the placement check and plain stores
    (cpc_set_gru_xcd_local, for either direction and forced for any launch plan: bits 2 / 3), the steering constants of the poll
    pacing: every combination leaves output, input gradient and parameter gradients bit-identical to the device-scope hand-over of
    rounds 2-5, with no polling time-out.  B = 20: a ragged second tile; B = 256: two batch tiles per workgroup."""
    dev = _dev()
    from cpc_audio_amd import _lib, ops
    from cpc_audio_amd.model import CPCAR
    lib = _lib.get()
    p = O.make_params(seed=4)
    ar = CPCAR(256, 256, False, 2, mode="GRU").to(dev)
    ar.load_state_dict({k[len("gAR."):]: v for k, v in p.items() if k.startswith("gAR.")})
    g = torch.Generator().manual_seed(B * 1000 + S)
    x = torch.randn(B, S, 256, generator=g).to(dev)
    dy = torch.randn(B, S, 256, generator=g).to(dev)
    outs = []
    try:
        for plain, local, pace in ((0, 0, -1), (_lib.DEFAULT_GRU_POLL_PLAIN, _lib.DEFAULT_GRU_XCD_LOCAL, -1), (15, 15, -1), (31, 3, -(16 * 4 + 4)),
                                   (5, 12, -(16 * 1 + 1))):
            assert lib.cpc_set_gru_poll_plain(plain) == 0 and lib.cpc_set_gru_xcd_local(local) == 0
            assert lib.cpc_set_gru_poll_pacing(pace, pace) == 0
            ar.zero_grad(set_to_none=True)
            xd = x.clone().requires_grad_(True)
            y = ar(xd)
            (y * dy).sum().backward()
            torch.cuda.synchronize()
            ops.check_device_errors()
            outs.append([y.detach().clone(), xd.grad.clone()] + [q.grad.clone() for q in ar.parameters()])
    finally:
        lib.cpc_set_gru_poll_plain(_lib.DEFAULT_GRU_POLL_PLAIN)
        lib.cpc_set_gru_xcd_local(_lib.DEFAULT_GRU_XCD_LOCAL)
        lib.cpc_set_gru_poll_pacing(-1, -1)
    assert all(torch.isfinite(t).all() for t in outs[0])
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


def test_gru_fp16_split_forward_is_within_fp32_rounding_of_exact_products():
    """cpc_set_gru_mode(2) (default) against mode 1 (exact-f32 MFMAs) over the full 128-step recurrence at B = 64:
    the two-piece fp16 split keeps 22 mantissa bits per operand; tolerance 2e-6 absolute on |y| < 1."""
    dev = _dev()
    from cpc_audio_amd import _lib
    from cpc_audio_amd.model import CPCAR
    lib = _lib.get()
    p = O.make_params(seed=4)
    ar = CPCAR(256, 256, False, 2, mode="GRU").to(dev)
    ar.load_state_dict({k[len("gAR."):]: v for k, v in p.items() if k.startswith("gAR.")})
    x = torch.randn(64, 128, 256, generator=torch.Generator().manual_seed(3)).to(dev)
    ys = []
    for mode in (1, 2):
        assert lib.cpc_set_gru_mode(mode) == 0
        try:
            with torch.no_grad():
                ys.append(ar(x).clone())
            torch.cuda.synchronize()
        finally:
            lib.cpc_set_gru_mode(_lib_default_gru_mode())
    d = (ys[0] - ys[1]).abs().max().item()
    assert 0.0 < d < 2e-6, d


def test_device_error_flags_negative_index_and_recurrence_timeout():
    """The two things a kernel can notice but not raise (include/cpc_hip.h, cpc_device_error_flags):
    * cpc_nce_prepare handed an index outside the batch -> clamped, flagged, ops.check_device_errors raises;
    * a wave of the persistent recurrence that runs out of its polling budget -> flagged (and NaN in the outputs, which
      is what reaches the loss).  Driven here by a budget of zero re-reads (cpc_set_gru_spin_limit(0)): every hand-over
      that is not complete at the first look counts as a timeout."""
    dev = _dev()
    from cpc_audio_amd import _lib, ops
    from cpc_audio_amd._lib import CpcHipError
    from cpc_audio_amd.model import CPCAR
    lib = _lib.get()
    lib.cpc_device_error_flags(1)
    B, S, K, N = 4, 128, 12, 128
    g = torch.Generator().manual_seed(0)
    bi, si = O.draw_negative_indices(B, S, S - K, N, generator=g)
    ops.prepare_negatives(bi.to(dev), si.to(dev), B, S, K, N)
    ops.check_device_errors()                               # clean draws: nothing flagged
    bi[17] = B
    ext, perm, row_ptr = ops.prepare_negatives(bi.to(dev), si.to(dev), B, S, K, N)
    with pytest.raises(CpcHipError, match="negative-sample indices"):
        ops.check_device_errors()
    ops.check_device_errors()                               # cleared by the raising call
    assert int(ext.max()) < B * S and int(row_ptr[-1]) == B * (S - K) * (N + K)

    p = O.make_params(seed=4)
    ar = CPCAR(256, 256, False, 2, mode="GRU").to(dev)
    ar.load_state_dict({k[len("gAR."):]: v for k, v in p.items() if k.startswith("gAR.")})
    x = torch.randn(64, 128, 256, generator=g).to(dev)
    assert lib.cpc_set_gru_spin_limit(0) == 0
    try:
        with torch.no_grad():
            y = ar(x)
        torch.cuda.synchronize()
    finally:
        lib.cpc_set_gru_spin_limit(-1)
    assert lib.cpc_device_error_flags(0) & 1
    assert not torch.isfinite(y).all()                      # the fill pattern (a NaN) went through, the launch ended
    with pytest.raises(CpcHipError, match="recurrence timed out"):
        ops.check_device_errors()
    with torch.no_grad():
        y = ar(x)                                           # default budget: fine again
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    ops.check_device_errors()


def test_two_trainers_on_two_threads_of_one_device_do_not_disturb_each_other():
    """The overlap state (side streams, events, launches held back) belongs to a Trainer's StepContext, the event sets of the C
    side are per stream and created under a lock: two train loops driven from two Python threads on ONE device -- each on its
    own stream, what a threaded data-parallel wrapper does -- must each produce, bit for bit, what they produce alone."""
    import threading
    dev = _dev()
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params

    def make(seed):
        p = O.make_params(seed=seed, head_scale=64.0)
        model, crit = build_model().to(dev), build_criterion().to(dev)
        load_flat_params(model, crit, p)
        wave = O.make_waveform(4, 20480, seed=seed + 1).to(dev)
        g = torch.Generator().manual_seed(seed + 2)
        bi, si = O.draw_negative_indices(4, 128, 116, 128, generator=g)
        return Trainer(model, crit, graph=False), wave, (bi.to(dev), si.to(dev))

    def run(job, out, key):
        try:
            tr, wave, neg = job
            with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                losses = [tr.step(wave, None, negatives=neg)[0].clone() for _ in range(3)]
                torch.cuda.current_stream().synchronize()
            state = dict(tr.model.state_dict())
            state.update(tr.criterion.state_dict())
            out[key] = (torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in state.items()})
        except BaseException as e:                # surfaced by the assert below
            out[key] = e

    alone = {}
    for key, seed in (("a", 31), ("b", 57)):
        run(make(seed), alone, key)
    both = {}
    threads = [threading.Thread(target=run, args=(make(seed), both, key)) for key, seed in (("a", 31), ("b", 57))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    for key in ("a", "b"):
        assert not isinstance(alone[key], BaseException), alone[key]
        assert not isinstance(both[key], BaseException), both[key]
        assert torch.equal(alone[key][0], both[key][0]), key
        for k, v in alone[key][1].items():
            assert torch.equal(v, both[key][1][k]), (key, k)


class _FixedMask(torch.nn.Module):
    """Stands in for nn.Dropout(p=0.5) with a mask the oracle can use too: keep -> x / (1 - p)."""

    def __init__(self, keep):
        super().__init__()
        self.keep = keep

    def forward(self, x):
        return x * self.keep * 2.0 if self.training else x


def test_prediction_dropout_option_matches_the_oracle_with_the_same_masks():
    """criterion.py:59,113-114 (``--dropout``): nn.Dropout(p = 0.5) on every head's prediction before the scores.  With the
    module's masks fixed, losses, accuracies and the gradients of everything in front of the criterion must equal the oracle's
    (its ``predict`` hook: linear head, then the same mask); in eval mode the option is the identity."""
    dev = _dev()
    from cpc_audio_amd.train import build_criterion, build_model, load_flat_params
    B, K, N, L = 2, 12, 128, 20480
    S, C = L // 160, 256
    W = S - K
    p = O.make_params(seed=7, head_scale=128.0)
    model = build_model().to(dev)
    crit = build_criterion(dropout=True).to(dev)
    assert isinstance(crit.wPrediction.dropout, torch.nn.Dropout) and crit.wPrediction.dropout.p == 0.5
    load_flat_params(model, crit, p)
    keep = (torch.rand(B, W, K * C, generator=torch.Generator().manual_seed(21)) < 0.5).float()
    crit.wPrediction.dropout = _FixedMask(keep.to(dev))
    wave = O.make_waveform(B, L, seed=10)
    bi, si = O.draw_negative_indices(B, S, W, N, generator=torch.Generator().manual_seed(5))
    c, z, _ = model(wave.to(dev), None)
    losses, acc = crit(c, z, None, negatives=(bi.to(dev), si.to(dev)))
    losses.sum().backward()
    # oracle: the same step with predict(k, cw) = dropout_k(W_k cw)
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    co, zo, _ = O.model_forward(leaves, wave)
    ext = O.negative_rows(bi, si, B, S, W, N)
    predict = lambda k, cw: (cw @ leaves[f"wPrediction.predictors.{k}.weight"].t()) * keep[:, :, k * C:(k + 1) * C] * 2.0
    lo, ao = O.criterion_forward(leaves, co, zo, ext, K, predict=predict)
    lo.sum().backward()
    assert (losses.detach().cpu() - lo.detach()).abs().max().item() < 1e-4
    assert (acc.detach().cpu() - ao).abs().max().item() < 2e-3
    for k_ in range(K):
        name = f"wPrediction.predictors.{k_}.weight"
        assert _rel(crit.wPrediction.predictors[k_].weight.grad.cpu(), leaves[name].grad) < 2e-4, name
    assert _rel(model.gAR.baseNet.weight_hh_l1.grad.cpu(), leaves["gAR.baseNet.weight_hh_l1"].grad) < 2e-4
    assert _rel(model.gEncoder.conv4.weight.grad.cpu(), leaves["gEncoder.conv4.weight"].grad) < 5e-3
    # the masked step differs from the plain one (the option does something), and eval mode is the plain criterion
    plain = O.criterion_forward(leaves, co, zo, ext, K)[0].detach()
    assert (plain - lo.detach()).abs().max().item() > 1e-3
    crit.eval()
    with torch.no_grad():
        le, _ = crit(c.detach(), z.detach(), None, negatives=(bi.to(dev), si.to(dev)))
    assert (le.cpu() - plain).abs().max().item() < 1e-4


@pytest.mark.parametrize("mode", ["LSTM", "RNN", "ffd", "conv8"])
def test_other_prediction_networks_score_through_the_hip_kernels(mode):
    """criterion.py:63-81 (``--rnnMode LSTM / RNN / ffd / conv4-12``): torch modules with the reference's parameter names
    (tests/test_abi_symbols.py pins them against the reference's outputs) whose predictions the HIP score kernels take as a
    tensor.  Losses, accuracies and the gradients of the predictors and of everything in front of the criterion must equal the
    oracle's with ``predict(k, cw)`` = a CPU copy of head k."""
    import copy
    dev = _dev()
    from cpc_audio_amd.train import build_criterion, build_model, load_flat_params
    B, K, N, L = 2, 12, 128, 20480
    S = L // 160
    W = S - K
    p = O.make_params(seed=7, head_scale=128.0)
    model = build_model()
    torch.manual_seed(3)
    crit = build_criterion(rnnMode=mode)
    assert crit.wPrediction.scores_apart
    lin = build_criterion()                                   # the linear heads only carry the flat parameters' names here
    load_flat_params(model, lin, p)
    heads = copy.deepcopy(crit.wPrediction.predictors)        # CPU copies for the oracle
    model, crit = model.to(dev), crit.to(dev)
    wave = O.make_waveform(B, L, seed=10)
    bi, si = O.draw_negative_indices(B, S, W, N, generator=torch.Generator().manual_seed(5))
    c, z, _ = model(wave.to(dev), None)
    losses, acc = crit(c, z, None, negatives=(bi.to(dev), si.to(dev)))
    losses.sum().backward()
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    co, zo, _ = O.model_forward(leaves, wave)
    ext = O.negative_rows(bi, si, B, S, W, N)

    def predict(k, cw):
        y = heads[k](cw)
        return y[0] if isinstance(y, tuple) else y
    lo, ao = O.criterion_forward(leaves, co, zo, ext, K, predict=predict)
    lo.sum().backward()
    assert (losses.detach().cpu() - lo.detach()).abs().max().item() < 1e-4
    assert (acc.detach().cpu() - ao).abs().max().item() < 2e-3
    for (name, got), want in zip(crit.wPrediction.predictors.named_parameters(), heads.parameters()):
        assert _rel(got.grad.cpu(), want.grad) < 2e-4, name
    assert _rel(model.gAR.baseNet.weight_hh_l1.grad.cpu(), leaves["gAR.baseNet.weight_hh_l1"].grad) < 2e-4
    assert _rel(model.gEncoder.conv4.weight.grad.cpu(), leaves["gEncoder.conv4.weight"].grad) < 5e-3


def test_prediction_dropout_with_torchs_own_masks_trains_through_the_trainer():
    """The real nn.Dropout: the Trainer takes the autograd path (the composite step does not cover it), losses are finite and
    above the no-dropout loss of the same parameters on average."""
    dev = _dev()
    from cpc_audio_amd.train import Trainer, build_criterion, build_model
    torch.manual_seed(3)
    model, crit = build_model().to(dev), build_criterion(dropout=True).to(dev)
    tr = Trainer(model, crit)
    wave = O.make_waveform(4, 20480, seed=2).to(dev)
    first = None
    for _ in range(3):
        losses, acc = tr.step(wave, None)
        assert torch.isfinite(losses).all()
        first = losses if first is None else first
    assert tr._fused is None                    # never went through cpc_train_step
    assert float(losses.mean()) < float(first.mean())


def test_a_second_thread_issuing_on_the_same_stream_is_refused():
    """ops.issuing_step: the C side's pooled events are keyed by the main stream, so two threads issuing steps on ONE stream of
    a device at the same time have no defined order (csrc/capi.hip).  The second one gets a RuntimeError that says what to do;
    the same thread may nest, another stream is fine, and the claim is gone when the step has been issued."""
    import threading
    dev = _dev()
    from cpc_audio_amd import ops
    seen = {}

    def other(use_own_stream):
        try:
            with torch.cuda.device(dev):
                if use_own_stream:
                    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                        with ops.issuing_step(dev):
                            seen[use_own_stream] = "ok"
                else:
                    with ops.issuing_step(dev):
                        seen[use_own_stream] = "ok"
        except RuntimeError as e:
            seen[use_own_stream] = str(e)

    with torch.cuda.device(dev), ops.StepContext():
        with ops.issuing_step(dev):                           # the same thread nests
            for own in (False, True):
                t = threading.Thread(target=other, args=(own,))
                t.start()
                t.join()
    assert "its own stream" in seen[False] and seen[True] == "ok", seen
    t = threading.Thread(target=other, args=(False,))         # ... and afterwards the stream is free again
    t.start()
    t.join()
    assert seen[False] == "ok" and not ops._issuing
