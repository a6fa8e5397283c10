"""Transformer layer kernels (csrc/transformer.hip) on the host SIMT emulator vs the oracle
(oracle/transformer_oracle.py, pinned against cpc/transformers.py by oracle/make_golden_transformer.py)."""
import ctypes

import numpy as np
import pytest
import torch

from emu_util import P, emu, rel_err
from oracle import transformer_oracle as T

ORDER = ["multihead.Wo.weight", "multihead.Wk.weight", "multihead.Wq.weight", "multihead.Wv.weight",
         "multihead.Att.Krelpos", "ln_multihead.weight", "ln_multihead.bias", "ffnetwork.lin1.weight",
         "ffnetwork.lin1.bias", "ffnetwork.lin2.weight", "ffnetwork.lin2.bias", "ln_ffnetwork.weight",
         "ln_ffnetwork.bias"]


def run_layer(lib, p, x, dy, S):
    B = x.size(0)
    plist = [p[k].contiguous() if k in p else None for k in ORDER]
    sizes = (ctypes.c_long * 8)()
    assert lib.cpc_transformer_layout(B, S, sizes) == 0
    saved = torch.full((sizes[0],), float("nan"))
    fscr = torch.full((sizes[1],), float("nan"))
    bscr = torch.full((sizes[2],), float("nan"))
    out = torch.full((B, S, 256), float("nan"))
    parr = (ctypes.c_void_p * 13)(*[P(t) for t in plist])
    assert lib.cpc_transformer_layer_forward(P(x), parr, P(saved), P(fscr), P(out), B, S, None) == 0
    dx = torch.full((B, S, 256), float("nan"))
    grads = [torch.full_like(t, float("nan")) if t is not None else None for t in plist]
    garr = (ctypes.c_void_p * 13)(*[P(t) for t in grads])
    assert lib.cpc_transformer_layer_backward(P(x), parr, P(saved), P(dy), P(bscr), P(dx), garr, B, S, None) == 0
    return out, dx, {k: g for k, g in zip(ORDER, grads) if g is not None}


@pytest.mark.parametrize("B,S,abspos", [(2, 128, False), (1, 116, False), (1, 40, False), (1, 128, True)])
def test_transformer_layer_forward_backward_emulated(B, S, abspos):
    lib = emu()
    p = T.make_layer_params(seed=3 + S, size_seq=S, abspos=abspos)
    g = torch.Generator().manual_seed(S)
    x = torch.randn(B, S, 256, generator=g)
    dy = torch.randn(B, S, 256, generator=g)
    out, dx, grads = run_layer(lib, p, x, dy, S)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    yr = T.layer_forward(leaves, xr)
    (yr * dy).sum().backward()
    assert (out - yr).abs().max().item() < 1e-5         # tolerance: fp32 parity, see DESIGN.md section 2
    assert rel_err(dx, xr.grad) < 1e-5
    bad = {k: rel_err(g, leaves[k].grad) for k, g in grads.items() if not rel_err(g, leaves[k].grad) < 1e-5}
    assert not bad, bad
