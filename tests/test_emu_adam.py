"""adam.hip on the host SIMT emulator against torch.optim.Adam (CPU): same update, fp32 (tolerance: a few ulp of the
lr-sized step).  Covers several tensors per launch, tails that are not a multiple of 4, an unaligned tensor, and more
tensors than one launch takes."""
import ctypes
import math

import torch

from emu_util import emu


def _run(lib, ps, gs, ms, vs, lr, b1, b2, eps, step):
    n = len(ps)
    arr = ctypes.c_void_p * n
    rc = lib.cpc_adam_step(arr(*[t.data_ptr() for t in ps]), arr(*[t.data_ptr() for t in gs]),
                           arr(*[t.data_ptr() for t in ms]), arr(*[t.data_ptr() for t in vs]),
                           (ctypes.c_long * n)(*[t.numel() for t in ps]), n, lr, b1, b2, eps, 1.0 - b1 ** step,
                           math.sqrt(1.0 - b2 ** step), None)
    assert rc == 0, rc


def test_adam_kernel_emulated():
    lib = emu()
    g = torch.Generator().manual_seed(2)
    big = torch.randn(9001, generator=g)
    shapes = [(4096,), (4100,), (3,), (257, 5)] + [(7,)] * 50          # 54 tensors: two launches
    ps = [torch.randn(s, generator=g) for s in shapes] + [big[1:]]      # the last one is 4-byte aligned only
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = torch.optim.Adam(ref, lr=2e-4, betas=(0.9, 0.999), eps=1e-8)
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    for step in range(1, 4):
        gs = [torch.randn(p.shape, generator=g) * 10.0 ** (step - 2) for p in ps]
        for r, gr in zip(ref, gs):
            r.grad = gr.clone()
        opt.step()
        _run(lib, ps, gs, ms, vs, 2e-4, 0.9, 0.999, 1e-8, step)
    for p, r, m, v in zip(ps, ref, ms, vs):
        # one rounding of p per step (ulp of |p| <= 4 is 4.8e-7) + the rounding inside the lr-sized update
        assert (p - r.detach()).abs().max().item() <= 2.5e-7 * max(1.0, r.abs().max().item())
        assert torch.allclose(m, opt.state[r]["exp_avg"], rtol=1e-5, atol=1e-6 * m.abs().max().item())
        assert torch.allclose(v, opt.state[r]["exp_avg_sq"], rtol=1e-5, atol=1e-6 * v.abs().max().item())
    assert big[0].item() == big[0].item()                               # untouched neighbour element stays finite


def test_adam_rejects_bad_arguments_emulated():
    lib = emu()
    p = torch.zeros(4)
    arr = (ctypes.c_void_p * 1)(p.data_ptr())
    n = (ctypes.c_long * 1)(4)
    assert lib.cpc_adam_step(arr, arr, arr, arr, n, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0.1, None) != 0
    assert lib.cpc_adam_step(None, arr, arr, arr, n, 1, 1e-3, 0.9, 0.999, 1e-8, 0.1, 0.1, None) != 0
    assert lib.cpc_adam_step(None, None, None, None, None, 0, 1e-3, 0.9, 0.999, 1e-8, 0.1, 0.1, None) == 0
