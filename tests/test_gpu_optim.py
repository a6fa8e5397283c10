"""cpc_adam_step behind cpc_audio_amd.optim.Adam against torch.optim.Adam (the reference's optimiser,
cpc/train.py:335-337) on the same gradients: same update rule, fp32; tolerance 2e-6 relative to lr-sized steps."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    return torch.device("cuda:0")


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(256, 256, 8), (256,), (3, 5, 7), (1,), (768, 256), (4099,), (256, 1, 10)]
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]


def test_adam_matches_torch_adam_over_steps():
    dev = _dev()
    from cpc_audio_amd.optim import Adam
    a, b = _params(dev, 3), _params(dev, 3)
    oa = Adam(a, lr=2e-4, betas=(0.9, 0.999), eps=1e-8)
    ob = torch.optim.Adam(b, lr=2e-4, betas=(0.9, 0.999), eps=1e-8)
    g = torch.Generator().manual_seed(9)
    for step in range(12):
        for pa, pb in zip(a, b):
            # gradients of very different scale, some exactly zero
            gr = torch.randn(pa.shape, generator=g) * (10.0 ** ((step % 5) - 3))
            gr[gr.abs() < 1e-4 * gr.abs().max()] = 0.0
            pa.grad = gr.to(dev)
            pb.grad = gr.to(dev).clone()
        oa.step()
        ob.step()
    for pa, pb in zip(a, b):
        # roundings of p itself (ulp of |p|) dominate: the update is lr-sized
        assert (pa - pb).abs().max().item() <= 2.5e-7 * max(1.0, pb.abs().max().item())
        sa, sb = oa.state[pa], ob.state[pb]
        assert float(sa["step"]) == float(sb["step"]) == 12
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-6 * sb["exp_avg"].abs().max().item())
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-6 * sb["exp_avg_sq"].abs().max().item())


def test_adam_state_dict_round_trips_with_torch_adam():
    dev = _dev()
    from cpc_audio_amd.optim import Adam
    a, b = _params(dev, 4), _params(dev, 4)
    oa = Adam(a, lr=1e-3)
    for p in a:
        p.grad = torch.ones_like(p)
    oa.step()
    ob = torch.optim.Adam(b, lr=1e-3)
    ob.load_state_dict(copy.deepcopy(oa.state_dict()))     # ours -> torch (load_state_dict keeps the tensors it is given)
    oc = Adam(_params(dev, 4), lr=1e-3)
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))     # torch -> ours
    for pa, pc in zip(a, oc.param_groups[0]["params"]):
        pc.data.copy_(pa.data)
        pa.grad = torch.full_like(pa, 0.5)
        pc.grad = torch.full_like(pc, 0.5)
    oa.step()
    oc.step()
    for pa, pc in zip(a, oc.param_groups[0]["params"]):
        assert torch.equal(pa, pc)
        assert float(oc.state[pc]["step"]) == 2


def test_adam_skips_parameters_without_gradient_and_other_groups_use_torch():
    dev = _dev()
    from cpc_audio_amd.optim import Adam
    p1, p2, p3 = (torch.nn.Parameter(torch.ones(10, device=dev)) for _ in range(3))
    opt = Adam([{"params": [p1, p2]}, {"params": [p3], "weight_decay": 0.1}], lr=1e-2)
    p1.grad = torch.ones_like(p1)
    p3.grad = torch.ones_like(p3)
    opt.step()
    assert torch.equal(p2, torch.ones_like(p2)) and len(opt.state[p2]) == 0
    assert (p1 - 0.99).abs().max().item() < 1e-6
    q = torch.nn.Parameter(torch.ones(10, device=dev))
    ref = torch.optim.Adam([q], lr=1e-2, weight_decay=0.1)
    q.grad = torch.ones_like(q)
    ref.step()
    assert torch.allclose(p3, q)


def test_graph_replayed_step_equals_the_eager_step():
    """Trainer(graph=True): the whole step (forward, backward on all streams, Adam with a device-side step counter,
    zero_grad) captured once as a HIP graph and replayed.  Given the same parameters and the same generator state the
    replayed step must leave the parameters the eager step leaves (same kernels, same order per stream), over several steps
    -- including Adam's bias correction, which a frozen host scalar would get wrong from the second replay on -- and the
    optimiser's state dict must report the true step count."""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    from oracle import cpc_oracle as O
    dev = torch.device("cuda:0")
    B, n_steps = 4, 4
    p = O.make_params(seed=15, head_scale=64.0)
    waves = [O.make_waveform(B, 20480, seed=60 + i).to(dev) for i in range(n_steps)]
    label = torch.zeros(B, dtype=torch.long, device=dev)
    finals, losses = [], []
    for graph in (False, True):
        model, crit = build_model().to(dev), build_criterion().to(dev)
        load_flat_params(model, crit, p)
        model.train(); crit.train()
        tr = Trainer(model, crit, graph=graph)
        if graph:
            tr.capture(waves[0], label)                    # warm-up + capture first, so that the generator can be aligned
            assert tr._captured is not None
        torch.manual_seed(4242)
        ls = []
        for i in range(n_steps):
            l, _ = tr.step(waves[i], label)
            ls.append(l.clone())
        torch.cuda.synchronize()
        assert (tr._captured is not None) == graph
        finals.append({k: v.detach().cpu() for k, v in list(model.state_dict().items()) + list(crit.state_dict().items())})
        losses.append(torch.stack(ls).cpu())
        sd = tr.optimizer.state_dict()
        assert {int(s["step"]) for s in sd["state"].values()} == {n_steps}
    assert torch.isfinite(losses[1]).all()
    assert torch.equal(losses[0], losses[1]), (losses[0] - losses[1]).abs().max()
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k


@pytest.mark.gpu
def test_graph_recapture_after_lr_change_and_eager_interlude():
    """The paths around a captured step that one capture does not exercise: a learning-rate change re-captures (the lr
    schedule does that every epoch), a step with caller-supplied negatives runs eagerly in between (leaving capturable mode,
    device step count -> host state) and the next plain step captures again.  The trajectory must equal the all-eager one
    bit for bit, each capture's warm-up must leave no trace, and the host-side throttle must not have kept an event that
    was recorded into a capture."""
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    from oracle import cpc_oracle as O
    dev = torch.device("cuda:0")
    B = 4
    p = O.make_params(seed=16, head_scale=64.0)
    plan = [(2e-4, False), (2e-4, False), (1e-4, False), (1e-4, True), (1e-4, False), (5e-5, False)]   # (lr, supplied negatives)
    waves = [O.make_waveform(B, 20480, seed=80 + i).to(dev) for i in range(len(plan))]
    label = torch.zeros(B, dtype=torch.long, device=dev)
    g = torch.Generator().manual_seed(77)
    bi, si = O.draw_negative_indices(B, 128, 116, 128, generator=g)
    neg = (bi.to(dev), si.to(dev))
    finals, losses = [], []
    for graph in (False, True):
        model, crit = build_model().to(dev), build_criterion().to(dev)
        load_flat_params(model, crit, p)
        model.train(); crit.train()
        tr = Trainer(model, crit, graph=graph)
        ls, captures = [], 0
        for i, (lr, supplied) in enumerate(plan):
            for grp in tr.optimizer.param_groups:
                grp["lr"] = lr
            if graph and not supplied and (tr._captured is None or tr._captured[0] != tr._graph_key(waves[i])):
                tr.capture(waves[i], label)                # explicit, so that the generator can be aligned behind the warm-up
                captures += 1
                assert not tr._done_events                 # nothing recorded during the capture is kept for the host to wait on
            torch.manual_seed(9000 + i)
            l, _ = tr.step(waves[i], label, negatives=neg if supplied else None)
            ls.append(l.clone())
            assert (tr._captured is not None) == (graph and not supplied)
        torch.cuda.synchronize()
        if graph:
            assert captures == 4                           # first use, lr 2e-4 -> 1e-4, after the eager interlude, lr -> 5e-5
        finals.append({k: v.detach().cpu() for k, v in list(model.state_dict().items()) + list(crit.state_dict().items())})
        losses.append(torch.stack(ls).cpu())
        sd = tr.optimizer.state_dict()
        assert {int(s["step"]) for s in sd["state"].values()} == {len(plan)}
    assert torch.isfinite(losses[1]).all()
    assert torch.equal(losses[0], losses[1]), (losses[0] - losses[1]).abs().max()
    for k in finals[0]:
        assert torch.equal(finals[0][k], finals[1][k]), k
