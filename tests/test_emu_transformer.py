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


@pytest.mark.parametrize("B,S,abspos", [(2, 128, False), (1, 116, False), (1, 40, False), (1, 128, True), (1, 37, False), (1, 1, False), (2, 33, True), (3, 65, False)])
def test_transformer_layer_forward_backward_emulated(B, S, abspos):
    # (S = 37: not a multiple of four -- the element-wise staging of Krelpos and a ragged last Philox row block)
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


@pytest.mark.parametrize("B,S,abspos", [(1, 160, False), (2, 129, False), (1, 400, False), (1, 257, True)])
def test_transformer_layer_forward_beyond_128_steps_emulated(B, S, abspos):
    """128 < S <= 512: forward only (attn_fwd_long_kernel: 128-row query blocks, key blocks up to the diagonal, running softmax,
    the relative-position term from the 63 distance columns a 32 x 32 tile sees) against the oracle; the backward entry point
    refuses, the layout keeps no attention probabilities.  S = 400: a 64000-sample feature-extraction chunk."""
    lib = emu()
    p = T.make_layer_params(seed=3 + S, size_seq=S, abspos=abspos)
    g = torch.Generator().manual_seed(S)
    x = torch.randn(B, S, 256, generator=g)
    plist = [p[k].contiguous() if k in p else None for k in ORDER]
    sizes = (ctypes.c_long * 8)()
    assert lib.cpc_transformer_layout(B, S, sizes) == 0
    assert sizes[5] == sizes[4]                                     # A (B*8,S,S) takes no room
    saved = torch.full((sizes[0],), float("nan"))
    fscr = torch.full((sizes[1],), float("nan"))
    out = torch.full((B, S, 256), float("nan"))
    parr = (ctypes.c_void_p * 13)(*[P(t) for t in plist])
    assert lib.cpc_transformer_layer_forward(P(x), parr, P(saved), P(fscr), P(out), B, S, None) == 0
    yr = T.layer_forward(p, x)
    assert (out - yr).abs().max().item() < 1e-5
    assert lib.cpc_transformer_layer_forward_dropout(P(x), parr, P(saved), P(fscr), P(out), B, S, 0.1, 5, None) != 0
    bscr = torch.zeros(max(1, sizes[2]))
    grads = [torch.zeros_like(t) if t is not None else None for t in plist]
    garr = (ctypes.c_void_p * 13)(*[P(t) for t in grads])
    dx = torch.zeros(B, S, 256)
    assert lib.cpc_transformer_layer_backward(P(x), parr, P(saved), P(out), P(bscr), P(dx), garr, B, S, None) != 0
    assert lib.cpc_transformer_layout(1, 513, sizes) != 0


@pytest.fixture(params=[1, 3, "dma", "dma-pass"], ids=["elementwise-relu", "relu-in-gemm-epilogue", "dma-fed-ffn", "dma-fed-ffn-relu-pass"])
def gemm_split(request):
    """cpc_set_gemm_split(3) puts every product on the wide fp16-piece tile however small the grid, which is the tile whose
    epilogue carries the feed-forward ReLU + dropout (forward) and the ReLU derivative (backward) -- at the sizes of these tests
    the default takes the small tiles with the elementwise kernels behind them.  "dma": cpc_set_gemm_dma(2), the feed-forward
    network on the DMA-fed tiles of gemm_dma.hip (hidden layer, y and the gradients kept as two fp16 pieces per element; what the
    K predictors run as a group on MI355X).  All must produce what the oracle produces with the masks cpc_dropout_keep_mask
    reports.  (Returned value: 1 / 3 = the gemm_split setting; the DMA variant reports 1 -- small generic tiles around it.)"""
    lib = emu()
    if request.param in ("dma", "dma-pass"):
        # ("dma": ReLU + dropout + the fp16 pieces + the mask bits in lin1's epilogue, the default; "dma-pass": + 8 = as a pass behind it)
        assert lib.cpc_set_gemm_dma(2 if request.param == "dma" else 10) == 0
        yield 1
        lib.cpc_set_gemm_dma(1)
        return
    assert lib.cpc_set_gemm_split(request.param) == 0
    yield request.param
    lib.cpc_set_gemm_split(1)


def test_dma_fed_ffn_with_the_tail_split_and_several_row_tiles_emulated():
    """The DMA-fed feed-forward path at M = 640 rows (three 256-row tiles, the last one ragged; two row splits of the TN kernels)
    and with the one-tile-wide NT products cut into a 256-row and a 128-row launch (cpc_set_gemm_tail_cus(2): what 348 tiles on 256
    CUs do on MI355X): forward, input gradient and every parameter gradient against the oracle.  (No ReLU-tie override at this size:
    the bar is 1e-5 only where no hidden unit sits within rounding of zero -- seeded so.)"""
    lib = emu()
    B, S = 5, 128
    p = T.make_layer_params(seed=3 + S, size_seq=S, abspos=False)
    g = torch.Generator().manual_seed(S)
    x = torch.randn(B, S, 256, generator=g)
    dy = torch.randn(B, S, 256, generator=g)
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    yr = T.layer_forward(leaves, xr)
    (yr * dy).sum().backward()
    assert lib.cpc_set_gemm_dma(2) == 0
    try:
        ref = {}
        for cus, rows in ((0, 64), (2, 64), (0, 128), (2, 128)):
            assert lib.cpc_set_gemm_tail_cus(cus) == 0
            assert lib.cpc_set_dma_wave_rows(rows) == 0
            out, dx, grads = run_layer(lib, p, x, dy, S)
            assert (out - yr).abs().max().item() < 1e-5, (cus, rows)
            assert rel_err(dx, xr.grad) < 1e-5, (cus, rows)
            bad = {k: rel_err(gr, leaves[k].grad) for k, gr in grads.items() if not rel_err(gr, leaves[k].grad) < 1e-5}
            assert not bad, (cus, rows, bad)
            # four 128 x 128 waves (one per SIMD, the four-stage 16-k loop) accumulate the same products in the same order as the
            # eight 64 x 128 waves: the same bits
            if rows == 64:
                ref[cus] = (out, dx, grads)
            else:
                assert torch.equal(out, ref[cus][0]) and torch.equal(dx, ref[cus][1]), (cus, rows)
                assert all(torch.equal(grads[k], ref[cus][2][k]) for k in grads), (cus, rows)
    finally:
        lib.cpc_set_dma_wave_rows(64)
        lib.cpc_set_gemm_tail_cus(0)
        lib.cpc_set_gemm_dma(1)


def test_hidden_layer_readback_on_either_storage_emulated():
    """cpc_transformer_hidden: the saved hidden layer as fp32 whether the forward kept it as fp32 (generic tiles) or as two fp16
    pieces per element (DMA-fed feed-forward GEMMs) -- equal to 2^-21 of its bound -- and both equal to the oracle's."""
    lib = emu()
    B, S = 1, 40
    prm = T.make_layer_params(seed=5, size_seq=S, abspos=False)
    x = torch.randn(B, S, 256, generator=torch.Generator().manual_seed(2))
    plist = [prm[k].contiguous() if k in prm else None for k in ORDER]
    sizes = (ctypes.c_long * 8)()
    assert lib.cpc_transformer_layout(B, S, sizes) == 0
    parr = (ctypes.c_void_p * 13)(*[P(t) for t in plist])
    hids = []
    for mode in (0, 2):
        assert lib.cpc_set_gemm_dma(mode) == 0
        try:
            saved = torch.full((sizes[0],), float("nan")); fscr = torch.full((sizes[1],), float("nan"))
            out = torch.full((B, S, 256), float("nan"))
            assert lib.cpc_transformer_layer_forward(P(x), parr, P(saved), P(fscr), P(out), B, S, None) == 0
            hid = torch.full((B * S, 2048), float("nan"))
            assert lib.cpc_transformer_hidden(P(saved), P(hid), B, S, None) == 0
            hids.append(hid)
        finally:
            lib.cpc_set_gemm_dma(1)
    assert torch.isfinite(hids[0]).all() and torch.isfinite(hids[1]).all()
    assert (hids[0] - hids[1]).abs().max().item() <= 2e-6 * hids[0].abs().max().item()
    assert ((hids[0] > 0) == (hids[1] > 0)).all()


@pytest.mark.parametrize("B,S,abspos,p_drop", [(1, 40, False, 0.1), (1, 48, True, 0.3), (1, 37, False, 0.2)])
def test_transformer_layer_training_dropout_emulated(B, S, abspos, p_drop, gemm_split):
    """Training-mode dropout (cpc/transformers.py:18,50,93,100): the layer run with dropout probability p and a seed must
    equal the oracle run with the masks that seed generates (cpc_dropout_keep_mask: Philox4x32-10 over the element index),
    forward and every gradient; the masks keep ~(1 - p) of the elements and the same seed reproduces the call."""
    if gemm_split == 3 and S != 40:
        pytest.skip("the wide tile is slow on the emulator: one shape is enough for the epilogue")
    lib = emu()
    prm = T.make_layer_params(seed=9 + S, size_seq=S, abspos=abspos)
    g = torch.Generator().manual_seed(S + 1)
    x = torch.randn(B, S, 256, generator=g)
    dy = torch.randn(B, S, 256, generator=g)
    seed = 0x1234ABCD5678EF01
    plist = [prm[k].contiguous() if k in prm else None for k in ORDER]
    sizes = (ctypes.c_long * 8)()
    assert lib.cpc_transformer_layout(B, S, sizes) == 0
    parr = (ctypes.c_void_p * 13)(*[P(t) for t in plist])

    def run(sd, forward_only=False):
        saved = torch.full((sizes[0],), float("nan")); fscr = torch.full((sizes[1],), float("nan"))
        bscr = torch.full((sizes[2],), float("nan")); out = torch.full((B, S, 256), float("nan"))
        assert lib.cpc_transformer_layer_forward_dropout(P(x), parr, P(saved), P(fscr), P(out), B, S, p_drop, sd, None) == 0
        if forward_only:
            return out, None, None
        dx = torch.full((B, S, 256), float("nan"))
        grads = [torch.full_like(t, float("nan")) if t is not None else None for t in plist]
        garr = (ctypes.c_void_p * 13)(*[P(t) for t in grads])
        assert lib.cpc_transformer_layer_backward_dropout(P(x), parr, P(saved), P(dy), P(bscr), P(dx), garr, B, S, p_drop, sd,
                                                          None) == 0
        return out, dx, {k: gr for k, gr in zip(ORDER, grads) if gr is not None}

    out, dx, grads = run(seed)
    attn_keep = torch.full((B * 8, S, S), float("nan"))
    ffn_keep = torch.full((B, S, 2048), float("nan"))
    assert lib.cpc_dropout_keep_mask(P(attn_keep), attn_keep.numel(), 0, S, p_drop, seed, None) == 0
    assert lib.cpc_dropout_keep_mask(P(ffn_keep), ffn_keep.numel(), 1, S, p_drop, seed, None) == 0
    for m in (attn_keep, ffn_keep):
        assert set(m.unique().tolist()) <= {0.0, float(np.float32(1.0) / (np.float32(1.0) - np.float32(p_drop)))}
        assert abs((m > 0).float().mean().item() - (1 - p_drop)) < 0.01
    leaves = {k: v.clone().requires_grad_(True) for k, v in prm.items()}
    xr = x.clone().requires_grad_(True)
    yr = T.layer_forward(leaves, xr, attn_keep=attn_keep, ffn_keep=ffn_keep)
    (yr * dy).sum().backward()
    assert (out - yr).abs().max().item() < 1e-5
    assert rel_err(dx, xr.grad) < 1e-5
    bad = {k: rel_err(gr, leaves[k].grad) for k, gr in grads.items() if not rel_err(gr, leaves[k].grad) < 1e-5}
    assert not bad, bad
    out2, _, _ = run(seed, forward_only=True)
    assert torch.equal(out, out2)                                   # the same seed reproduces the call ...
    other = torch.full_like(ffn_keep, float("nan"))
    assert lib.cpc_dropout_keep_mask(P(other), other.numel(), 1, S, p_drop, seed + 1, None) == 0
    assert not torch.equal(other, ffn_keep)                         # ... another seed draws other masks


@pytest.mark.parametrize("abspos,p_drop", [(False, 0.0), (False, 0.2), (True, 0.0)])
def test_group_of_layers_equals_single_layer_calls_emulated(abspos, p_drop, gemm_split):
    """cpc_transformer_group_forward / _backward (the K transformer predictors of the criterion run in lock-step, one launch
    per kernel): layer g of the group == a single-layer call on the same input with layer g's parameters (and seed + g),
    bit for bit -- outputs interleaved as (B*S, G*256), dx = the sum of the layers' input gradients."""
    if gemm_split == 3 and abspos:
        pytest.skip("the wide tile is slow on the emulator: p = 0 (ReLU epilogue) and p > 0 (ReLU-derivative epilogue) suffice")
    lib = emu()
    B, S, G = 1, 40, (2 if gemm_split == 3 else 3)              # (the wide tile is slow on the emulator: two layers there)
    prms = [T.make_layer_params(seed=20 + q, size_seq=S, abspos=abspos) for q in range(G)]
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, S, 256, generator=g)
    dy = torch.randn(B * S, G * 256, generator=g)
    seed = 0x0123456789ABCDE
    sizes = (ctypes.c_long * 8)()
    assert lib.cpc_transformer_layout(B, S, sizes) == 0
    kinds = [k for k in ORDER if k in prms[0]]
    stacked = {k: torch.stack([p[k] for p in prms]).contiguous() for k in kinds}
    parr = (ctypes.c_void_p * 13)(*[P(stacked[k]) if k in stacked else None for k in ORDER])
    saved = torch.full((G * sizes[0],), float("nan")); fscr = torch.full((G * sizes[1],), float("nan"))
    bscr = torch.full((G * sizes[2],), float("nan")); out = torch.full((B * S, G * 256), float("nan"))
    assert lib.cpc_transformer_group_forward(P(x), parr, P(saved), P(fscr), P(out), B, S, G, p_drop, seed, None) == 0
    dx = torch.full((B, S, 256), float("nan"))
    sgrads = {k: torch.full_like(v, float("nan")) for k, v in stacked.items()}
    garr = (ctypes.c_void_p * 13)(*[P(sgrads[k]) if k in sgrads else None for k in ORDER])
    assert lib.cpc_transformer_group_backward(P(x), parr, P(saved), P(dy), P(bscr), P(dx), garr, B, S, G, p_drop, seed, None) == 0
    dx_sum = torch.zeros(B, S, 256)
    for q in range(G):
        plist = [prms[q][k].contiguous() if k in prms[q] else None for k in ORDER]
        pq = (ctypes.c_void_p * 13)(*[P(t) for t in plist])
        sv = torch.full((sizes[0],), float("nan")); fs = torch.full((sizes[1],), float("nan"))
        bs = torch.full((sizes[2],), float("nan")); o1 = torch.full((B, S, 256), float("nan"))
        assert lib.cpc_transformer_layer_forward_dropout(P(x), pq, P(sv), P(fs), P(o1), B, S, p_drop, seed + q, None) == 0
        assert torch.equal(out[:, q * 256:(q + 1) * 256], o1.view(B * S, 256)), q
        d1 = torch.full((B, S, 256), float("nan"))
        g1 = [torch.full_like(t, float("nan")) if t is not None else None for t in plist]
        gq = (ctypes.c_void_p * 13)(*[P(t) for t in g1])
        dyq = dy[:, q * 256:(q + 1) * 256].contiguous()
        assert lib.cpc_transformer_layer_backward_dropout(P(x), pq, P(sv), P(dyq), P(bs), P(d1), gq, B, S, p_drop, seed + q,
                                                          None) == 0
        for k, t in zip(ORDER, g1):
            if t is not None:
                assert torch.equal(sgrads[k][q], t), (q, k)
        dx_sum += d1
    assert (dx - dx_sum).abs().max().item() <= 1e-6 * dx_sum.abs().max().item()
