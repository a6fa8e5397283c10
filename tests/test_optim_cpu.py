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
