"""cpc_audio_amd.optim.Adam on CPU parameters is torch.optim.Adam (the HIP launch is for GPU tensors only), with the
same state layout -- what the reference's checkpoints store under "optimizer" (cpc/train.py:139)."""
import torch


def test_cpu_parameters_take_torch_adam_path():
    from cpc_audio_amd.optim import Adam
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    oa, ob = Adam(a, lr=2e-4), torch.optim.Adam(b, lr=2e-4)
    for _ in range(3):
        for pa, pb in zip(a, b):
            g = torch.randn_like(pa)
            pa.grad, pb.grad = g, g.clone()
        oa.step()
        ob.step()
    for pa, pb in zip(a, b):
        assert torch.equal(pa, pb)
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"][0].keys() == sb["param_groups"][0].keys()
    assert set(sa["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}


def test_capturable_mode_keeps_its_moment_tensors_when_an_empty_or_partial_state_is_loaded():
    """A step-0 checkpoint (empty state) or a partial one loaded in graph-capturable mode: the moment tensors a captured step
    points at must stay in self.state (zeroed in place) and the device count must follow the loaded one."""
    from cpc_audio_amd.optim import Adam
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
    opt = Adam(a, lr=2e-4).device_step_counter(True)
    moments = [(opt.state[p]["exp_avg"], opt.state[p]["exp_avg_sq"]) for p in a]
    for m, v in moments:
        m.fill_(3.0)
        v.fill_(2.0)
    opt._device_step.fill_(7.0)
    fresh = Adam([torch.nn.Parameter(p.detach().clone()) for p in a], lr=2e-4).state_dict()     # no state at all
    assert fresh["state"] == {}
    opt.load_state_dict(fresh)
    for p, (m, v) in zip(a, moments):
        st = opt.state[p]
        assert st["exp_avg"] is m and st["exp_avg_sq"] is v and float(st["step"]) == 0.0
        assert not m.any() and not v.any()
    assert float(opt._device_step) == 0.0
    # a partial state: only parameter 0 present, at step 5
    part = {"state": {0: {"step": torch.tensor(5.0), "exp_avg": torch.full((4, 3), 0.5), "exp_avg_sq": torch.full((4, 3), 0.25)}},
            "param_groups": fresh["param_groups"]}
    moments[1][0].fill_(9.0)
    opt.load_state_dict(part)
    assert opt.state[a[0]]["exp_avg"] is moments[0][0] and torch.all(moments[0][0] == 0.5)
    assert opt.state[a[1]]["exp_avg"] is moments[1][0] and not moments[1][0].any()
    assert float(opt.state[a[1]]["step"]) == 5.0 and float(opt._device_step) == 5.0
