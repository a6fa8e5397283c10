"""BASELINE.json config 4 on a real MI355X: the transformer layer as auto-regressive network (--arMode
transformer) and as the K prediction networks (--rnnMode transformer), vs the CPU oracle
(oracle/transformer_oracle.py) and the committed reference fixtures.  Dropout 0 (the only setting with a defined
parity)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cpc_oracle as O
from oracle import transformer_oracle as T

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _ffn_masks(ops, B_S):
    """[hid_device > 0] of every transformer layer call recorded under ops.KEEP_DEBUG, in call order."""
    from cpc_audio_amd import _lib
    from cpc_audio_amd._lib import ptr as P
    lib = _lib.get()
    out = []
    for (saved, sizes), (B, S) in zip(ops.debug_last["transformer"], B_S):
        # (cpc_transformer_hidden: fp32 whatever the storage -- the predictors' group keeps it as two fp16 pieces per element)
        hid = torch.empty(B * S, 2048, device=saved.device)
        lib.check(lib.cpc_transformer_hidden(P(saved), P(hid), B, S, torch.cuda.current_stream().cuda_stream), "hidden")
        out.append((hid.view(B, S, 2048) > 0).cpu())
    return out


@pytest.mark.parametrize("case", ["transformer_ar_b2", "transformer_pred_b2", "transformer_abspos_b1"])
def test_transformer_layer_matches_oracle_and_reference_fixture(case, golden_dir):
    dev = _dev()
    from cpc_audio_amd.transformers import buildTransformerAR
    with open(os.path.join(golden_dir, "transformer_meta.json")) as f:
        m = json.load(f)["cases"][case]
    fx = np.load(os.path.join(golden_dir, case + ".npz"))
    B, S, abspos = m["batch"], m["size_seq"], m["abspos"]
    first = 1 if abspos else 0
    p = T.make_layer_params(m["param_seed"], 256, S, abspos, prefix=f"{first}.")
    net = buildTransformerAR(256, 1, S, abspos, dropout=0.0).to(dev)
    missing = net.load_state_dict(p, strict=False)
    assert not missing.unexpected_keys and all(k.endswith(("Att.z", "Att.mask", ".pe")) for k in missing.missing_keys)
    net.train()
    from cpc_audio_amd import ops
    g = torch.Generator().manual_seed(m["input_seed"])
    x = torch.randn(B, S, 256, generator=g)
    dy = torch.randn(B, S, 256, generator=g)
    xd = x.to(dev).requires_grad_(True)
    ops.debug_last.pop("transformer", None)
    ops.KEEP_DEBUG = True
    y = net(xd)
    ops.KEEP_DEBUG = False
    (y * dy.to(dev)).sum().backward()
    torch.cuda.synchronize()
    masks = _ffn_masks(ops, [(B, S)])
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x.clone().requires_grad_(True)
    yr = T.ar_forward(leaves, xr, 1, abspos, relu_override=masks)
    (yr * dy).sum().backward()
    # tolerance: 1e-4 on outputs is the north-star bar; expect ~1e-6
    assert (y.detach().cpu() - yr).abs().max().item() < 1e-5
    assert np.abs(y.detach().cpu()[:, ::8, :].numpy() - fx["y_slice"]).max() < 1e-5      # the REFERENCE's values
    assert _rel(xd.grad.cpu(), xr.grad) < 1e-5
    bad = {}
    for k, v in net.named_parameters():
        r = _rel(v.grad.cpu(), leaves[k].grad)
        if not r < 1e-5:
            bad[k] = r
    assert not bad, bad


@pytest.mark.parametrize("B", [3, 16])
def test_config4_transformer_ar_and_predictors_train_step(B):
    """Whole config-4 step: conv encoder -> transformer AR (S=128) -> 12 transformer predictors (S=116) -> InfoNCE, against the
    oracle; B = 16 puts 2048 / 1856 rows through the projection and feed-forward GEMMs (other tile choices than B = 3's 384 / 348)."""
    dev = _dev()
    from cpc_audio_amd import ops
    from cpc_audio_amd.train import build_criterion, build_model
    K, N = 12, 128
    base = O.make_params(seed=9, head_scale=1.0)
    p = {k: v for k, v in base.items() if k.startswith("gEncoder.")}
    p.update(T.make_layer_params(31, 256, 128, False, prefix="gAR.0."))
    for k in range(K):
        p.update(T.make_layer_params(40 + k, 256, 116, False, prefix=f"wPrediction.predictors.{k}.0."))
    model = build_model(arMode="transformer", transformerDropout=0.0).to(dev)
    crit = build_criterion(rnnMode="transformer", transformerDropout=0.0).to(dev)
    mm = model.load_state_dict({k: v for k, v in p.items() if not k.startswith("wPrediction")}, strict=False)
    cm = crit.load_state_dict({k: v for k, v in p.items() if k.startswith("wPrediction")}, strict=False)
    for miss in (mm, cm):
        assert not miss.unexpected_keys and all(k.endswith(("Att.z", "Att.mask")) for k in miss.missing_keys), miss
    model.train(); crit.train()
    wave = O.make_waveform(B, 20480, seed=77)
    gen = torch.Generator().manual_seed(3)
    bidx, sidx = O.draw_negative_indices(B, 128, 116, N, generator=gen)
    ops.debug_last.pop("transformer", None)
    ops.KEEP_DEBUG = True
    c, z, _ = model(wave.to(dev), torch.zeros(B, dtype=torch.long, device=dev))
    saved, sizes, zz = ops.debug_last["encoder"]
    acts_dev = ops.saved_encoder_activations(saved, B, 20480)      # fp32 copies whatever the storage (y0 is kept as fp16 pieces)
    losses, acc = crit(c, z, None, negatives=(bidx.to(dev), sidx.to(dev)))
    ops.KEEP_DEBUG = False
    tmasks = _ffn_masks(ops, [(B, 128)] + [(B, 116)] * K)       # call order: the AR layer, then predictors 0..K-1
    losses.sum().backward()
    torch.cuda.synchronize()
    ys = [t.cpu() for t in acts_dev] + [zz.cpu()]
    masks = [(y > 0).permute(0, 2, 1) for y in ys]

    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    zr = O.encoder_forward(leaves, wave, relu_override=masks).permute(0, 2, 1)
    cr = T.ar_forward(leaves, zr, 1, False, prefix="gAR.", relu_override=tmasks[:1])
    ext = O.negative_rows(bidx, sidx, B, 128, 116, N)
    lr, ar = O.criterion_forward(leaves, cr, zr, ext, K,
                                 predict=lambda k, cw: T.layer_forward(leaves, cw, prefix=f"wPrediction.predictors.{k}.0.",
                                                                       relu_override=tmasks[1 + k]))
    lr.sum().backward()
    assert (z.detach().cpu() - zr.detach()).abs().max().item() < 1e-4
    assert (c.detach().cpu() - cr.detach()).abs().max().item() < 1e-4
    assert (losses.detach().cpu() - lr.detach()).abs().max().item() < 1e-4
    assert (acc.cpu() - ar).abs().max().item() <= 2.0 / (116 * B) + 1e-7
    bad = {}
    grads = dict(model.named_parameters()); grads.update(dict(crit.named_parameters()))
    for k, v in grads.items():
        r = _rel(v.grad.cpu(), leaves[k].grad)
        if not r < 2e-4:
            bad[k] = r
    assert not bad, bad


def test_transformer_layer_trains_with_the_references_dropout():
    """The reference's layers always carry nn.Dropout(0.1) (cpc/transformers.py:18,93,100).  In training mode the HIP layer
    applies it with in-kernel Philox masks: checked against the oracle run with the masks of the call's seed (forward and
    gradients at fp32 parity), eval mode stays deterministic, and the default builders (dropout 0.1) train."""
    dev = _dev()
    import ctypes
    from cpc_audio_amd import _lib
    from cpc_audio_amd._lib import ptr as P
    from cpc_audio_amd.transformers import buildTransformerAR
    lib = _lib.get()
    B, S, p_drop = 3, 128, 0.1
    prm = T.make_layer_params(77, 256, S, False, prefix="0.")
    net = buildTransformerAR(256, 1, S, False).to(dev)              # default dropout: 0.1, as the reference
    net.load_state_dict(prm, strict=False)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, S, 256, generator=g)
    dy = torch.randn(B, S, 256, generator=g)
    net.train()
    torch.manual_seed(123)
    seed = int(torch.empty((), dtype=torch.int64).random_().item())  # what the layer will draw next under this manual_seed
    torch.manual_seed(123)
    xd = x.to(dev).requires_grad_(True)
    y = net(xd)
    (y * dy.to(dev)).sum().backward()
    torch.cuda.synchronize()
    attn_keep = torch.empty(B * 8, S, S, device=dev)
    ffn_keep = torch.empty(B, S, 2048, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lib.check(lib.cpc_dropout_keep_mask(P(attn_keep), attn_keep.numel(), 0, S, p_drop, seed, st), "keep_mask")
    lib.check(lib.cpc_dropout_keep_mask(P(ffn_keep), ffn_keep.numel(), 1, S, p_drop, seed, st), "keep_mask")
    torch.cuda.synchronize()
    assert abs((ffn_keep > 0).float().mean().item() - 0.9) < 2e-3 and abs((attn_keep > 0).float().mean().item() - 0.9) < 5e-3
    leaves = {k: v.clone().requires_grad_(True) for k, v in prm.items()}
    xr = x.clone().requires_grad_(True)
    yr = T.layer_forward(leaves, xr, prefix="0.", attn_keep=attn_keep.cpu(), ffn_keep=ffn_keep.cpu())
    (yr * dy).sum().backward()
    assert (y.detach().cpu() - yr).abs().max().item() < 1e-4
    assert _rel(xd.grad.cpu(), xr.grad) < 1e-4
    bad = {k: _rel(v.grad.cpu(), leaves[k].grad) for k, v in net.named_parameters()}
    bad = {k: v for k, v in bad.items() if not v < 1e-4}
    assert not bad, bad
    # eval mode: no dropout, deterministic, equal to the p = 0 oracle
    net.eval()
    with torch.no_grad():
        e1, e2 = net(x.to(dev)), net(x.to(dev))
    assert torch.equal(e1, e2)
    assert (e1.cpu() - T.layer_forward(prm, x, prefix="0.")).abs().max().item() < 1e-4


def test_config4_at_its_quoted_batch_is_finite_reproducible_and_grouped_equals_per_head():
    """BASELINE.json configs[3] at the size it is quoted on (B = 64 x 20480, transformer AR + 12 transformer predictors,
    dropout 0): the grouped-predictor path (one launch per kernel for the 12 layers, 58 x 12 GEMM tiles, the score kernels on
    foreign predictions with their 1 GB candidate-row buffer) is not what the B = 3 oracle test exercises.  Size-independent
    properties: finite losses near ln 129 at random init, two identical steps bit-identical, and the grouped path equal to the
    head-by-head path (same layers, same inputs; the GEMM tiling differs with the group size, so to fp32 rounding order)."""
    dev = _dev()
    from cpc_audio_amd.train import build_criterion, build_model
    B = 64
    torch.manual_seed(11)
    model = build_model(arMode="transformer", transformerDropout=0.0).to(dev)
    crit = build_criterion(rnnMode="transformer", transformerDropout=0.0).to(dev)
    model.train(); crit.train()
    wave = O.make_waveform(B, 20480, seed=71).to(dev)
    g = torch.Generator().manual_seed(19)
    bi, si = O.draw_negative_indices(B, 128, 116, 128, generator=g)
    neg = (bi.to(dev), si.to(dev))
    params = list(model.parameters()) + list(crit.parameters())

    def run(grouped):
        crit.wPrediction.group_predictors = grouped
        for q in params:
            q.grad = None
        c, z, _ = model(wave, None)
        losses, acc = crit(c, z, None, negatives=neg)
        losses.sum().backward()
        torch.cuda.synchronize()
        return losses.detach().clone(), [q.grad.clone() for q in params]

    # The feed-forward GEMMs of the GROUP run on the DMA-fed tiles by default and those of a single layer on the generic ones
    # (cpc_set_gemm_dma(1): where the launches fill the chip) -- two arithmetic paths with a ReLU between them, and a hidden unit
    # within rounding of zero may then fall on the other side (a handful of 182 M: ~1e-4 on a gradient's norm).  The property
    # here is group == per head, so both runs take the SAME path: once the generic tiles (0), once the DMA-fed ones (2).
    from cpc_audio_amd import _lib
    lib = _lib.get()
    try:
        for mode in (2, 0):
            lib.check(lib.cpc_set_gemm_dma(mode), "set_gemm_dma")
            l1, g1 = run(True)
            l2, g2 = run(True)
            l3, g3 = run(False)
            crit.wPrediction.group_predictors = True
            assert torch.isfinite(l1).all() and all(torch.isfinite(t).all() for t in g1)
            assert (l1 - 4.8598).abs().max().item() < 0.5, l1                       # ln(129) at random init (SURVEY.md trap T9)
            assert torch.equal(l1, l2) and all(torch.equal(a, b) for a, b in zip(g1, g2)), "two identical steps differ"
            assert (l1 - l3).abs().max().item() < 1e-5
            worst = max(_rel(a, b) for a, b in zip(g1, g3))
            assert worst < 2e-5, (mode, worst)
            if mode == 2:
                ldma, gdma = l1, g1
                # the plain NT products' 256-row tiles as four 128 x 128 waves, one per SIMD (cpc_set_dma_wave_rows(128): the
                # four-stage 16-k loop of dma_tile.h): the same products in the same order -- the same bits
                lib.check(lib.cpc_set_dma_wave_rows(128), "set_dma_wave_rows")
                try:
                    l4, g4 = run(True)
                finally:
                    lib.cpc_set_dma_wave_rows(64)
                assert torch.equal(l4, l1) and all(torch.equal(a, b) for a, b in zip(g4, g1)), "wave tiles of 128 rows differ"
        # ... and the two paths against each other: losses to rounding, gradients to the few ReLU ties
        assert (ldma - l1).abs().max().item() < 1e-5
        assert max(_rel(a, b) for a, b in zip(gdma, g1)) < 2e-3
    finally:
        lib.cpc_set_gemm_dma(1)


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_feed_forward_epilogues_equal_the_elementwise_kernels_bit_for_bit(p_drop):
    """At B = 64 the feed-forward GEMMs run on the wide fp16-piece tile, whose epilogue carries the ReLU derivative (backward, any
    dropout) and the ReLU (forward, dropout 0) -- cpc_set_gemm_fuse(0) puts the elementwise kernels back behind the same GEMMs.
    Same products, same order, same masks: every output and gradient of a layer call must agree bit for bit (but lin1's bias
    gradient, whose column sums the fused backward takes per tile inside the epilogue: rounding order only)."""
    dev = _dev()
    import ctypes
    from cpc_audio_amd import _lib
    from cpc_audio_amd._lib import ptr as P
    lib = _lib.get()
    B, S = 64, 128
    prm = T.make_layer_params(31, 256, S, False)
    order = ["multihead.Wo.weight", "multihead.Wk.weight", "multihead.Wq.weight", "multihead.Wv.weight", "multihead.Att.Krelpos",
             "ln_multihead.weight", "ln_multihead.bias", "ffnetwork.lin1.weight", "ffnetwork.lin1.bias", "ffnetwork.lin2.weight",
             "ffnetwork.lin2.bias", "ln_ffnetwork.weight", "ln_ffnetwork.bias"]
    plist = [prm[k].contiguous().to(dev) for k in order]
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, S, 256, generator=g).to(dev)
    dy = torch.randn(B, S, 256, generator=g).to(dev)
    sizes = (ctypes.c_long * 8)()
    lib.check(lib.cpc_transformer_layout(B, S, sizes), "layout")
    parr = (ctypes.c_void_p * 13)(*[P(t) for t in plist])
    st = torch.cuda.current_stream().cuda_stream
    runs = []
    try:
        for fuse in (1, 0):
            lib.check(lib.cpc_set_gemm_fuse(fuse), "fuse")
            saved = torch.empty(sizes[0], device=dev); fscr = torch.empty(sizes[1], device=dev); bscr = torch.empty(sizes[2], device=dev)
            out = torch.empty(B, S, 256, device=dev); dx = torch.empty(B, S, 256, device=dev)
            grads = [torch.empty_like(t) for t in plist]
            garr = (ctypes.c_void_p * 13)(*[P(t) for t in grads])
            lib.check(lib.cpc_transformer_layer_forward_dropout(P(x), parr, P(saved), P(fscr), P(out), B, S, p_drop, 99, st), "fwd")
            lib.check(lib.cpc_transformer_layer_backward_dropout(P(x), parr, P(saved), P(dy), P(bscr), P(dx), garr, B, S, p_drop, 99,
                                                                 st), "bwd")
            torch.cuda.synchronize()
            hid = saved[sizes[7]:sizes[7] + B * S * 2048].clone()
            runs.append([out, dx, hid] + grads)
    finally:
        lib.cpc_set_gemm_fuse(1)
    assert torch.isfinite(runs[0][0]).all() and (runs[0][2] == 0).float().mean().item() > 0.3     # the ReLU (and dropout) cut
    for i, (a, b) in enumerate(zip(*runs)):
        if i == 3 + order.index("ffnetwork.lin1.bias"):      # summed per 128-row tile in the epilogue, then over the tiles:
            assert (a - b).abs().max().item() <= 2e-6 * b.abs().max().item()       # another order of the same 8192 terms
        else:
            assert torch.equal(a, b), i


@pytest.mark.parametrize("B,S,abspos,p_drop", [(64, 128, False, 0.1), (5, 116, False, 0.0), (3, 37, True, 0.2)])
def test_attention_forward_kernels_agree_bit_for_bit(B, S, abspos, p_drop):
    """cpc_set_attn_fwd: the two-workgroups-per-CU forward attention kernel (Q in registers, Krelpos from L2, E per score tile; the
    default since round 6) feeds the same operands into the same MFMA chains as the one-tile-in-LDS kernel of rounds 1-5: layer
    output, attention output and the saved probabilities are bit-identical, with and without dropout / relative positions."""
    dev = _dev()
    from cpc_audio_amd import _lib
    from cpc_audio_amd._lib import ptr as P
    import ctypes
    lib = _lib.get()
    prm = T.make_layer_params(41 + S, 256, S, abspos)
    order = ["multihead.Wo.weight", "multihead.Wk.weight", "multihead.Wq.weight", "multihead.Wv.weight", "multihead.Att.Krelpos",
             "ln_multihead.weight", "ln_multihead.bias", "ffnetwork.lin1.weight", "ffnetwork.lin1.bias", "ffnetwork.lin2.weight",
             "ffnetwork.lin2.bias", "ln_ffnetwork.weight", "ln_ffnetwork.bias"]
    plist = [prm[k].contiguous().to(dev) if k in prm else None for k in order]
    x = torch.randn(B, S, 256, generator=torch.Generator().manual_seed(S)).to(dev)
    sizes = (ctypes.c_long * 8)()
    lib.check(lib.cpc_transformer_layout(B, S, sizes), "layout")
    parr = (ctypes.c_void_p * 13)(*[P(t) for t in plist])
    st = torch.cuda.current_stream().cuda_stream
    res = []
    try:
        for variant in (0, 1):
            lib.check(lib.cpc_set_attn_fwd(variant), "attn_fwd")
            saved = torch.full((sizes[0],), float("nan"), device=dev); fscr = torch.empty(sizes[1], device=dev)
            out = torch.empty(B, S, 256, device=dev)
            lib.check(lib.cpc_transformer_layer_forward_dropout(P(x), parr, P(saved), P(fscr), P(out), B, S, p_drop, 4242, st), "fwd")
            torch.cuda.synchronize()
            res.append((out.clone(), saved[sizes[4]:sizes[4] + B * 8 * S * S].clone(), saved[sizes[5]:sizes[5] + B * S * 256].clone()))
    finally:
        lib.cpc_set_attn_fwd(1)
    assert torch.isfinite(res[1][0]).all()
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,S,abspos", [(2, 400, False), (3, 129, False), (1, 512, False), (2, 257, True)])
def test_transformer_layer_beyond_128_steps_runs_on_the_kernels_in_inference(B, S, abspos):
    """A layer built for more than 128 steps (sizeSeq = 400: a 64000-sample feature-extraction window, cpc/feature_loader.py
    :247-266 with cpc/transformers.py:22-49 at that length) runs on the HIP kernels when no gradient is asked for -- attention as
    a running-softmax walk over key blocks (attn_fwd_long_kernel) -- and matches the oracle; the same module asked for a
    gradient, or in training mode with dropout, composes torch ops (differentiable) and matches as well."""
    dev = _dev()
    from cpc_audio_amd import _lib
    from cpc_audio_amd.transformers import TransformerLayer, buildTransformerAR
    first = 1 if abspos else 0
    p = T.make_layer_params(11 + S, 256, S, abspos, prefix=f"{first}.")
    net = buildTransformerAR(256, 1, S, abspos, dropout=0.1).to(dev)
    net.load_state_dict(p, strict=False)
    layer = net[-1]
    assert isinstance(layer, TransformerLayer) and layer.fused and not layer.fused_train
    x = torch.randn(B, S, 256, generator=torch.Generator().manual_seed(S))
    yr = T.ar_forward(p, x, 1, abspos)
    net.eval()
    calls = []
    lib = _lib.get()
    orig = lib.cpc_transformer_layer_forward_dropout
    lib.cpc_transformer_layer_forward_dropout = lambda *a: (calls.append(a[5:7]), orig(*a))[1]     # (B, S) of every kernel call
    try:
        with torch.no_grad():
            y = net(x.to(dev))
        torch.cuda.synchronize()
        assert calls == [(B, S)]                                  # the HIP layer ran, once
        assert (y.cpu() - yr).abs().max().item() < 1e-5
        # with a gradient: torch ops (no kernel call), same values, gradients against the oracle
        xd = x.to(dev).requires_grad_(True)
        y2 = net(xd)
        assert len(calls) == 1 and (y2.detach().cpu() - yr).abs().max().item() < 1e-4
        g = torch.randn(B, S, 256, generator=torch.Generator().manual_seed(1))
        (y2 * g.to(dev)).sum().backward()
        xr = x.clone().requires_grad_(True)
        (T.ar_forward(p, xr, 1, abspos) * g).sum().backward()
        assert _rel(xd.grad.cpu(), xr.grad) < 1e-4
    finally:
        lib.cpc_transformer_layer_forward_dropout = orig


def test_feature_extraction_with_a_transformer_context_network_built_for_400_frames():
    """build_feature (cpc/feature_loader.py:228-269) on a model whose --arMode transformer network was built for the 64000-sample
    window (sizeSeq = 400): encoder and the transformer layer on the HIP kernels, chunks batched, against the oracle."""
    dev = _dev()
    from cpc_audio_amd import harness as H
    from cpc_audio_amd.train import build_model
    model = build_model(arMode="transformer", sizeWindow=64000, transformerDropout=0.1).to(dev)
    p = O.make_params(seed=5)
    ep = {k: v for k, v in p.items() if k.startswith("gEncoder")}
    tp = T.make_layer_params(21, 256, 400, False, prefix="0.")
    missing = model.load_state_dict({**ep, **{"gAR." + k: v for k, v in tp.items()}}, strict=False)
    assert not missing.unexpected_keys and all(k.endswith(("Att.z", "Att.mask")) for k in missing.missing_keys), missing
    n = 64000 * 3
    seq = (0.1 * torch.randn(1, n, generator=torch.Generator().manual_seed(2))).clamp_(-1, 1)
    fm = H.FeatureModule(model, get_encoded=False).eval()
    feats = H.build_feature(fm, seq, strict=True, max_size_seq=64000)
    ref = []
    for start in range(0, n, 64000):
        z = O.encoder_forward(p, seq[:, start:start + 64000].reshape(1, 1, -1)).permute(0, 2, 1)
        ref.append(T.ar_forward(tp, z, 1, False))
    ref = torch.cat(ref, dim=1)
    assert feats.shape == ref.shape == (1, 1200, 256)
    assert (feats - ref).abs().max().item() < 1e-4
