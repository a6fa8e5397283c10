"""InfoNCE criterion kernels (csrc/nce.hip) on the host SIMT emulator vs the oracle."""
import ctypes

import pytest
import torch

from cpc_audio_amd import _lib as _L

from emu_util import P, emu, rel_err
from oracle import cpc_oracle as O


@pytest.mark.parametrize("B,S,K,N,scale,wide,fused", [(2, 20, 12, 16, 1.0, 0, 1), (3, 19, 5, 32, 40.0, 0, 1), (2, 20, 12, 16, 1.0, 1, 1),
                                                          (2, 21, 7, 32, 3000.0, 0, 1), (1, 20, 16, 16, 1.0, 0, 1),
                                                          (3, 19, 5, 32, 40.0, 0, 0), (2, 21, 7, 32, 3000.0, 0, 0),
                                                          (2, 20, 12, 16, 1.0, 0, 3), (3, 19, 5, 32, 40.0, 0, 3), (2, 21, 7, 32, 3000.0, 0, 3),
                                                          (1, 20, 16, 16, 1.0, 0, 3), (3, 19, 5, 32, 40.0, 0, 2)])
def test_nce_forward_backward_emulated(B, S, K, N, scale, wide, fused):
    """wide: the prediction GEMM on the 128 x 256 pipelined tile (cpc_set_gemm_split(3) forces it at test sizes).
    scale 3000: logits hundreds apart, the softmax is saturated (most score gradients are exactly 0) and the running reference of
    the online softmax moves (scale 40 as well).  fused: the one-pass criterion (default: scores and the unit-gradient dPred from
    one gather pass, cpc_set_nce_fused) or the two-pass kernels; 2 / 3: the one-pass criterion on fp16 pieces (H2 gather sources, DMA'd
    tiles, transposing LDS reads: nce_fwd_h2_kernel; 3: + nce_bwd_g_h2_kernel)."""
    lib = emu()
    assert lib.cpc_set_gemm_split(3 if wide else 1) == 0 and lib.cpc_set_nce_fused(fused) == 0
    # (the prediction product runs on the DMA-fed tile by default -- cpc_set_nce_heads_dma, round 6 --; the `wide` case and the
    # two-pass criterion keep the generic tiles covered)
    assert lib.cpc_set_nce_heads_dma(0 if (wide or fused == 0) else 1) == 0
    try:
        _nce_forward_backward(lib, B, S, K, N, scale)
    finally:
        lib.cpc_set_gemm_split(1)
        lib.cpc_set_nce_fused(_L.DEFAULT_NCE_FUSED)
        lib.cpc_set_nce_heads_dma(1)


@pytest.mark.parametrize("grid", [0, 1, 3])
def test_scoring_kernel_with_a_capped_grid_emulated(grid):
    """cpc_set_nce_grid: the fp16-piece scoring kernel as at most `grid` workgroups that walk their windows with the grid's stride
    (the default, -1, is two per CU: 512 workgroups for the 1856 of B = 64 on MI355X; 0 = one per four windows) -- every window is
    still scored by one wave on its own, so losses, logits and every gradient are those of the oracle whatever the cap."""
    lib = emu()
    assert lib.cpc_set_nce_grid(grid) == 0
    try:
        _nce_forward_backward(lib, 3, 19, 5, 32, 40.0)
    finally:
        lib.cpc_set_nce_grid(-1)


def _nce_forward_backward(lib, B, S, K, N, scale):
    torch.manual_seed(2)
    W = S - K
    p = O.make_params(seed=4, n_predicts=K, head_scale=scale)
    heads = O.head_weights(p, K)
    wall = torch.cat(heads, dim=0).contiguous()
    c = torch.tanh(torch.randn(B, S, 256))
    z = torch.relu(torch.randn(B, S, 256))
    g = torch.Generator().manual_seed(9)
    bi, si = O.draw_negative_indices(B, S, W, N, generator=g)
    ext = O.negative_rows(bi, si, B, S, W, N)                       # (B,N,W)
    ext_t = ext.permute(0, 2, 1).contiguous().to(torch.int32)       # (B,W,N)
    sizes = (ctypes.c_long * 6)()
    assert lib.cpc_nce_layout(B, S, K, N, sizes) == 0
    saved = torch.full((sizes[0],), float("nan"))
    fscr = torch.full((sizes[1],), float("nan"))
    losses = torch.full((K,), float("nan")); acc = torch.full((K,), float("nan"))
    assert lib.cpc_nce_forward(P(c), P(z), P(wall), P(ext_t), P(saved), P(fscr), P(losses), P(acc), B, S, K, N, None) == 0
    # the same forward on a workspace whose GEMM operand bounds were written ahead of time (cpc_nce_bounds: max|wall| reduced,
    # |c| <= 1 a priori -- c is a tanh here): what the train loops queue beside the encoder (criterion.prepare_step)
    saved2 = torch.full((sizes[0],), float("nan"))
    l2 = torch.full((K,), float("nan")); a2 = torch.full((K,), float("nan"))
    assert lib.cpc_nce_bounds(None, 1.0, P(wall), P(saved2), B, S, K, N, None) == 0
    assert lib.cpc_nce_forward_prepared(P(c), P(z), P(wall), P(ext_t), P(saved2), P(fscr), P(l2), P(a2), B, S, K, N, None) == 0
    assert (l2 - losses).abs().max().item() < 1e-5 * max(1.0, losses.abs().max().item()) and torch.equal(a2, acc)
    assert lib.cpc_nce_bounds(None, 0.0, P(wall), P(saved2), B, S, K, N, None) != 0     # neither a bound nor the tensor
    leaves = {f"wPrediction.predictors.{k}.weight": heads[k].clone().requires_grad_(True) for k in range(K)}
    cr = c.clone().requires_grad_(True); zr = z.clone().requires_grad_(True)
    lr, ar = O.criterion_forward(leaves, cr, zr, ext, K)
    tol = 1e-5 * max(1.0, lr[0].abs().max().item())                # fp32: relative once the values are large (scale 3000)
    assert (losses - lr[0]).abs().max().item() < tol, (losses, lr)
    assert (acc - ar[0]).abs().max().item() < 1e-6
    lg = O.criterion_logits(leaves, cr, zr, ext, K)
    mine = saved[sizes[4]: sizes[4] + B * W * K * (N + 1)].view(B, W, K, N + 1)
    for k in range(K):
        assert (mine[:, :, k, :].permute(0, 2, 1) - lg[k]).abs().max().item() < 1e-5 * max(1.0, lg[k].abs().max().item())
    gl = torch.randn(K)
    (lr[0] * gl).sum().backward()
    bscr = torch.full((sizes[2],), float("nan"))
    dc = torch.full((B, S, 256), float("nan")); dz = torch.full((B, S, 256), float("nan"))
    dwall = torch.full((K * 256, 256), float("nan"))
    from cpc_audio_amd.ops import candidate_destinations
    perm_ref, row_ptr_ref = candidate_destinations(ext_t, B, S, K)
    # device-side index preparation must reproduce ext and the destination-sorted slot lists
    ext_k = torch.full((B, W, N), -1, dtype=torch.int32)
    perm = torch.full((B * W * (N + K),), -1, dtype=torch.int32)
    row_ptr = torch.full((B * S + 1,), -1, dtype=torch.int32)
    work = torch.zeros(B * W * (N + K) + 2 * B * S + 2, dtype=torch.int32)
    assert lib.cpc_nce_prepare(P(bi), P(si), P(ext_k), P(perm), P(row_ptr), P(work), B, S, K, N, None) == 0
    # (a window's negatives come back in ascending order -- the criterion is invariant under their permutation)
    assert torch.equal(ext_k, torch.sort(ext_t, dim=2).values) and torch.equal(row_ptr, row_ptr_ref)
    perm_sorted_ref, _ = candidate_destinations(ext_k, B, S, K)
    for r in range(B * S):
        lo, hi = int(row_ptr[r]), int(row_ptr[r + 1])
        assert sorted(perm[lo:hi].tolist()) == perm_sorted_ref[lo:hi].tolist()
    # (the forward above ran on the draw-order rows ext_t: the slot lists that go with THEM)
    assert lib.cpc_nce_backward(P(c), P(z), P(wall), P(ext_t), P(perm_ref), P(row_ptr_ref), P(saved), P(gl), P(bscr), P(dc),
                                P(dz), P(dwall), B, S, K, N, None) == 0
    assert rel_err(dc, cr.grad) < 1e-5
    assert rel_err(dz, zr.grad) < 1e-5
    ref_dw = torch.cat([leaves[f"wPrediction.predictors.{k}.weight"].grad for k in range(K)], dim=0)
    assert rel_err(dwall, ref_dw) < 1e-5
    # ... and the whole criterion on what cpc_nce_prepare made of the same draws (rows ascending per window, slot lists placed by
    # atomic cursors): losses and every gradient are those of the draw-order computation
    for t_ in (saved, losses, acc, dc, dz, dwall, bscr):
        t_.fill_(float("nan"))
    assert lib.cpc_nce_forward(P(c), P(z), P(wall), P(ext_k), P(saved), P(fscr), P(losses), P(acc), B, S, K, N, None) == 0
    assert (losses - lr[0].detach()).abs().max().item() < tol and (acc - ar[0]).abs().max().item() < 1e-6
    assert lib.cpc_nce_backward(P(c), P(z), P(wall), P(ext_k), P(perm), P(row_ptr), P(saved), P(gl), P(bscr), P(dc),
                                P(dz), P(dwall), B, S, K, N, None) == 0
    assert rel_err(dc, cr.grad) < 1e-5 and rel_err(dz, zr.grad) < 1e-5 and rel_err(dwall, ref_dw) < 1e-5


@pytest.mark.parametrize("B,S,K,N,fused", [(1, 13, 12, 1, 1), (1, 13, 12, 1, 0), (3, 5, 1, 3, 1), (1, 6, 2, 1040, 1), (1, 6, 2, 1040, 0),
                                            (2, 9, 16, 5, 1), (1, 13, 12, 1, 3), (3, 5, 1, 3, 3), (1, 6, 2, 1040, 3), (2, 9, 16, 5, 3)])
def test_criterion_at_the_edges_of_its_shapes_emulated(B, S, K, N, fused):
    """One window per sequence (S = K + 1), a single negative, a single head, more negatives than the per-window sort holds
    (kSortMax = 1024: such lists stay in draw order), S <= K (no window: refused)."""
    lib = emu()
    sizes = (ctypes.c_long * 6)()
    if S <= K:
        assert lib.cpc_nce_layout(B, S, K, N, sizes) != 0
        return
    assert lib.cpc_set_nce_fused(fused) == 0
    try:
        torch.manual_seed(6)
        W = S - K
        p = O.make_params(seed=5, n_predicts=K, head_scale=8.0)
        heads = O.head_weights(p, K)
        wall = torch.cat(heads, dim=0).contiguous()
        c = torch.tanh(torch.randn(B, S, 256))
        z = torch.relu(torch.randn(B, S, 256))
        bi, si = O.draw_negative_indices(B, S, W, N, generator=torch.Generator().manual_seed(3))
        ext = O.negative_rows(bi, si, B, S, W, N)
        Np = lib.cpc_nce_padded_negatives(N)
        assert lib.cpc_nce_layout(B, S, K, N, sizes) == 0
        ext_k = torch.full((B, W, Np), -1, dtype=torch.int32)
        perm = torch.full((B * W * (Np + K),), -1, dtype=torch.int32)
        row_ptr = torch.full((B * S + 1,), -1, dtype=torch.int32)
        work = torch.zeros(B * W * (Np + K) + 2 * B * S + 2, dtype=torch.int32)
        assert lib.cpc_nce_prepare(P(bi), P(si), P(ext_k), P(perm), P(row_ptr), P(work), B, S, K, N, None) == 0
        got_rows = ext_k[:, :, :N]
        want_rows = ext.permute(0, 2, 1).to(torch.int32)
        if N <= 1024:
            assert torch.equal(got_rows, torch.sort(want_rows, dim=2).values)
        else:
            assert torch.equal(got_rows, want_rows)                      # beyond the sort tile: draw order
        saved = torch.full((sizes[0],), float("nan")); fscr = torch.full((sizes[1],), float("nan"))
        bscr = torch.full((sizes[2],), float("nan"))
        losses = torch.full((K,), float("nan")); acc = torch.full((K,), float("nan"))
        assert lib.cpc_nce_forward(P(c), P(z), P(wall), P(ext_k), P(saved), P(fscr), P(losses), P(acc), B, S, K, N, None) == 0
        leaves = {f"wPrediction.predictors.{k}.weight": heads[k].clone().requires_grad_(True) for k in range(K)}
        cr = c.clone().requires_grad_(True); zr = z.clone().requires_grad_(True)
        lr, ar = O.criterion_forward(leaves, cr, zr, ext, K)
        assert (losses - lr[0].detach()).abs().max().item() < 1e-5 * max(1.0, lr[0].abs().max().item())
        assert (acc - ar[0]).abs().max().item() < 1e-6
        gl = torch.randn(K)
        (lr[0] * gl).sum().backward()
        dc = torch.full((B, S, 256), float("nan")); dz = torch.full((B, S, 256), float("nan"))
        dwall = torch.full((K * 256, 256), float("nan"))
        assert lib.cpc_nce_backward(P(c), P(z), P(wall), P(ext_k), P(perm), P(row_ptr), P(saved), P(gl), P(bscr), P(dc), P(dz),
                                    P(dwall), B, S, K, N, None) == 0
        ref_dw = torch.cat([leaves[f"wPrediction.predictors.{k}.weight"].grad for k in range(K)], dim=0)
        # (the fp16-piece kernels at N = 1040, 700 unsorted slots per destination row: 1.0e-5 on dz against an fp64 oracle where the
        # exact-f32 kernels measure 5e-6 -- operands rounded to 2^-22 on top of the same fp32 accumulation; the bar of the path is 2e-4)
        tol = 2e-5 if fused == 3 and N > 512 else 1e-5
        assert rel_err(dc, cr.grad) < tol and rel_err(dz, zr.grad) < tol and rel_err(dwall, ref_dw) < tol
    finally:
        lib.cpc_set_nce_fused(_L.DEFAULT_NCE_FUSED)


@pytest.mark.parametrize("B,S,K,N,fused", [(2, 28, 20, 16, 1), (1, 41, 35, 24, 1), (2, 28, 20, 16, 0), (1, 19, 17, 5, 1), (1, 34, 32, 16, 1),
                                            (2, 28, 20, 16, 3), (1, 41, 35, 24, 3)])
def test_more_than_sixteen_heads_walked_in_groups_emulated(B, S, K, N, fused):
    """criterion.py:225-257 takes any nPredicts; the score tiles hold 16 heads.  cpc_nce_head_group(k0, K) makes the following
    calls work on heads k0 .. of a K-step criterion (W = S - K windows, positives z[t + k0 + k + 1]): the groups' losses /
    accuracies side by side and their dc / dz summed must be the oracle's K-head criterion; the linear heads' path and the
    given-predictions path (cpc_nce_scores_*), with the lists cpc_nce_prepare makes under the same setting."""
    lib = emu()
    assert lib.cpc_set_nce_fused(fused) == 0
    try:
        torch.manual_seed(3)
        W = S - K
        p = O.make_params(seed=4, n_predicts=K, head_scale=20.0)
        heads = O.head_weights(p, K)
        c = torch.tanh(torch.randn(B, S, 256))
        z = torch.relu(torch.randn(B, S, 256))
        bi, si = O.draw_negative_indices(B, S, W, N, generator=torch.Generator().manual_seed(9))
        ext = O.negative_rows(bi, si, B, S, W, N)
        leaves = {f"wPrediction.predictors.{k}.weight": heads[k].clone().requires_grad_(True) for k in range(K)}
        cr = c.clone().requires_grad_(True); zr = z.clone().requires_grad_(True)
        lr, ar = O.criterion_forward(leaves, cr, zr, ext, K)
        gl = torch.randn(K)
        (lr[0] * gl).sum().backward()
        Np = lib.cpc_nce_padded_negatives(N)
        assert lib.cpc_nce_layout(B, S, K, N, (ctypes.c_long * 6)()) != 0          # no group set: K > 16 is not one call
        for apart in (False, True):
            dc_sum, dz_sum = torch.zeros(B, S, 256), torch.zeros(B, S, 256)
            for k0 in range(0, K, 16):
                kg = min(16, K - k0)
                wall = torch.cat(heads[k0:k0 + kg], dim=0).contiguous()
                assert lib.cpc_nce_head_group(k0, K) == 0
                try:
                    sizes = (ctypes.c_long * 6)()
                    assert lib.cpc_nce_layout(B, S, kg, N, sizes) == 0
                    ext_k = torch.full((B, W, Np), -1, dtype=torch.int32)
                    perm = torch.full((B * W * (Np + kg),), -1, dtype=torch.int32)
                    row_ptr = torch.full((B * S + 1,), -1, dtype=torch.int32)
                    work = torch.zeros(B * W * (Np + kg) + 2 * B * S + 2, dtype=torch.int32)
                    assert lib.cpc_nce_prepare(P(bi), P(si), P(ext_k), P(perm), P(row_ptr), P(work), B, S, kg, N, None) == 0
                    saved = torch.full((sizes[0],), float("nan")); fscr = torch.full((sizes[1],), float("nan"))
                    bscr = torch.full((sizes[2],), float("nan"))
                    losses = torch.full((kg,), float("nan")); acc = torch.full((kg,), float("nan"))
                    dc = torch.full((B, S, 256), float("nan")); dz = torch.full((B, S, 256), float("nan"))
                    g_ = gl[k0:k0 + kg].contiguous()
                    if not apart:
                        dwall = torch.full((kg * 256, 256), float("nan"))
                        assert lib.cpc_nce_forward(P(c), P(z), P(wall), P(ext_k), P(saved), P(fscr), P(losses), P(acc), B, S, kg, N,
                                                   None) == 0
                        assert lib.cpc_nce_backward(P(c), P(z), P(wall), P(ext_k), P(perm), P(row_ptr), P(saved), P(g_), P(bscr),
                                                    P(dc), P(dz), P(dwall), B, S, kg, N, None) == 0
                        ref_dw = torch.cat([leaves[f"wPrediction.predictors.{k}.weight"].grad for k in range(k0, k0 + kg)], dim=0)
                        assert rel_err(dwall, ref_dw) < 1e-5
                    else:
                        pred = (c[:, :W] @ wall.t()).contiguous()                       # (B, W, kg*256)
                        dpred = torch.full_like(pred, float("nan"))
                        assert lib.cpc_nce_scores_forward(P(pred), P(z), P(ext_k), P(saved), P(fscr), P(losses), P(acc), B, S, kg, N,
                                                          None) == 0
                        assert lib.cpc_nce_scores_backward(P(pred), P(z), P(ext_k), P(perm), P(row_ptr), P(saved), P(g_), P(bscr),
                                                           P(dpred), P(dz), B, S, kg, N, None) == 0
                        dc = torch.zeros(B, S, 256)
                        dc[:, :W] = dpred @ wall
                finally:
                    assert lib.cpc_nce_head_group(0, 0) == 0
                tol = 1e-5 * max(1.0, lr[0].abs().max().item())
                assert (losses - lr[0, k0:k0 + kg].detach()).abs().max().item() < tol, (k0, losses, lr)
                assert (acc - ar[0, k0:k0 + kg]).abs().max().item() < 1e-6
                dc_sum += dc
                dz_sum += dz
            assert rel_err(dc_sum, cr.grad) < 1e-5, apart
            assert rel_err(dz_sum, zr.grad) < 1e-5, apart
        assert lib.cpc_nce_head_group(3, 2) != 0 and lib.cpc_nce_head_group(-1, 4) != 0 and lib.cpc_nce_head_group(1, 0) != 0
        assert lib.cpc_nce_head_group(16, 20) == 0
        assert lib.cpc_nce_layout(B, S, 5, N, (ctypes.c_long * 6)()) != 0          # heads 16..20 of 20: at most 4
        assert lib.cpc_nce_head_group(0, 0) == 0
    finally:
        lib.cpc_nce_head_group(0, 0)
        lib.cpc_set_nce_fused(_L.DEFAULT_NCE_FUSED)


def test_index_preparation_with_a_capped_grid_gives_the_same_lists():
    """cpc_set_index_prep_groups(n): at most n workgroups per launch of cpc_nce_prepare's kernels, each walking several windows /
    slots.  ext and row_ptr must be identical, every destination row's slot set too (the order inside a row is set by atomics)."""
    lib = emu()
    B, S, K, N = 3, 25, 5, 24
    W = S - K
    Np = lib.cpc_nce_padded_negatives(N)
    bi, si = O.draw_negative_indices(B, S, W, N, generator=torch.Generator().manual_seed(17))

    def prepare(cap):
        assert lib.cpc_set_index_prep_groups(cap) == 0
        try:
            ext = torch.full((B, W, Np), -1, dtype=torch.int32)
            perm = torch.full((B * W * (Np + K),), -1, dtype=torch.int32)
            row_ptr = torch.full((B * S + 1,), -1, dtype=torch.int32)
            work = torch.zeros(B * W * (Np + K) + 2 * B * S + 2, dtype=torch.int32)
            assert lib.cpc_nce_prepare(P(bi), P(si), P(ext), P(perm), P(row_ptr), P(work), B, S, K, N, None) == 0
        finally:
            assert lib.cpc_set_index_prep_groups(-1) == 0
        return ext, perm, row_ptr

    ref = prepare(0)
    for cap in (1, 3):
        got = prepare(cap)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2]), cap
        for r in range(B * S):
            lo, hi = int(ref[2][r]), int(ref[2][r + 1])
            assert sorted(got[1][lo:hi].tolist()) == sorted(ref[1][lo:hi].tolist()), (cap, r)
    assert lib.cpc_set_index_prep_groups(-2) != 0


def test_out_of_range_negative_indices_are_clamped_and_flagged():
    """cpc_nce_prepare validates caller-supplied draws (criterion.py:181-189 draws batchIdx in [0,B), seqIdx in [1,S)):
    an index outside the batch would otherwise count and gather out of bounds.  The device flag is read through
    cpc_device_error_flags (bit CPC_DEVERR_NEGATIVE_INDEX = 2)."""
    lib = emu()
    B, S, K, N = 2, 20, 12, 16
    W = S - K
    g = torch.Generator().manual_seed(1)
    bi, si = O.draw_negative_indices(B, S, W, N, generator=g)

    def prepare(bi, si):
        ext = torch.full((B, W, N), -1, dtype=torch.int32)
        perm = torch.full((B * W * (N + K),), -1, dtype=torch.int32)
        row_ptr = torch.full((B * S + 1,), -1, dtype=torch.int32)
        work = torch.zeros(B * W * (N + K) + 2 * B * S + 2, dtype=torch.int32)
        assert lib.cpc_nce_prepare(P(bi), P(si), P(ext), P(perm), P(row_ptr), P(work), B, S, K, N, None) == 0
        return ext, row_ptr

    lib.cpc_device_error_flags(1)
    ext, row_ptr = prepare(bi, si)
    assert lib.cpc_device_error_flags(0) == 0
    bad_b, bad_s = bi.clone(), si.clone()
    bad_b[5] = B + 3
    bad_s[11] = -7
    ext, row_ptr = prepare(bad_b, bad_s)
    assert lib.cpc_device_error_flags(0) & 2
    assert lib.cpc_device_error_flags(1) & 2            # read + clear
    assert lib.cpc_device_error_flags(0) == 0
    assert int(ext.min()) >= 0 and int(ext.max()) < B * S
    assert int(row_ptr[-1]) == B * W * (N + K)


@pytest.mark.parametrize("B,S,K,N", [(2, 20, 12, 16), (2, 21, 7, 32)])
def test_nce_scores_of_foreign_predictions_emulated(B, S, K, N):
    """cpc_nce_scores_{forward,backward}: the criterion on predictions some other network made (criterion.py:82-88,
    --rnnMode transformer).  Such predictions are not linear in c, so dz goes through the per-candidate gradient rows and the
    destination-sorted gather rather than the re-associated path of the linear heads."""
    lib = emu()
    torch.manual_seed(3)
    W = S - K
    z = torch.relu(torch.randn(B, S, 256))
    pred = (2.0 * torch.randn(B, W, K * 256)).requires_grad_(True)
    zr = z.clone().requires_grad_(True)
    g = torch.Generator().manual_seed(11)
    bi, si = O.draw_negative_indices(B, S, W, N, generator=g)
    ext = O.negative_rows(bi, si, B, S, W, N)                       # (B,N,W)
    ext_t = ext.permute(0, 2, 1).contiguous().to(torch.int32)
    # reference: criterion.py:115-116, 245-257 on the given predictions
    neg = zr.reshape(B * S, 256)[ext.reshape(-1)].view(B, N, W, 256)
    ref_losses = []
    for k in range(K):
        cand = torch.cat([zr[:, k + 1:k + 1 + W].unsqueeze(1), neg], dim=1)            # (B,1+N,W,256)
        sc = (pred[:, :, k * 256:(k + 1) * 256].unsqueeze(1) * cand).mean(dim=3)      # (B,1+N,W)
        sc = sc.permute(0, 2, 1).reshape(B * W, 1 + N)
        ref_losses.append(torch.nn.functional.cross_entropy(sc, torch.zeros(B * W, dtype=torch.long)))
    ref_losses = torch.stack(ref_losses)
    gl = torch.randn(K)
    (ref_losses * gl).sum().backward()
    sizes = (ctypes.c_long * 6)()
    assert lib.cpc_nce_layout(B, S, K, N, sizes) == 0
    saved = torch.full((sizes[0],), float("nan"))
    fscr = torch.full((sizes[1],), float("nan"))
    losses = torch.full((K,), float("nan")); acc = torch.full((K,), float("nan"))
    pd = pred.detach().contiguous()
    assert lib.cpc_nce_scores_forward(P(pd), P(z), P(ext_t), P(saved), P(fscr), P(losses), P(acc), B, S, K, N, None) == 0
    assert (losses - ref_losses.detach()).abs().max().item() < 1e-5
    from cpc_audio_amd.ops import candidate_destinations
    perm, row_ptr = candidate_destinations(ext_t, B, S, K)
    bscr = torch.full((sizes[2],), float("nan"))
    dpred = torch.full((B, W, K * 256), float("nan")); dz = torch.full((B, S, 256), float("nan"))
    assert lib.cpc_nce_scores_backward(P(pd), P(z), P(ext_t), P(perm), P(row_ptr), P(saved), P(gl), P(bscr), P(dpred),
                                       P(dz), B, S, K, N, None) == 0
    assert rel_err(dpred, pred.grad) < 1e-5
    assert rel_err(dz, zr.grad) < 1e-5


@pytest.mark.parametrize("B,S,K,N,fused", [(2, 20, 12, 24, 1), (2, 19, 5, 40, 1), (2, 20, 12, 24, 0), (1, 18, 3, 7, 1),
                                            (2, 20, 12, 24, 3), (1, 18, 3, 7, 3)])
def test_negatives_not_a_multiple_of_16_emulated(B, S, K, N, fused):
    """criterion.py:176-189 draws any number of negatives; the kernels walk candidates in 16-wide MFMA tiles.  The lists are padded
    to the tile (cpc_nce_padded_negatives; padding = a valid row of z) and the scoring kernels force the padding's logits to -3e38:
    weight 0 in the softmax, the arg-max and every gradient.  Whole criterion against the oracle with exactly N negatives."""
    lib = emu()
    assert lib.cpc_set_nce_fused(fused) == 0
    try:
        torch.manual_seed(3)
        W = S - K
        Np = lib.cpc_nce_padded_negatives(N)
        assert Np % 16 == 0 and 0 <= Np - N < 16
        p = O.make_params(seed=5, n_predicts=K, head_scale=20.0)
        heads = O.head_weights(p, K)
        wall = torch.cat(heads, dim=0).contiguous()
        c = torch.tanh(torch.randn(B, S, 256))
        z = torch.relu(torch.randn(B, S, 256))
        g = torch.Generator().manual_seed(11)
        bi, si = O.draw_negative_indices(B, S, W, N, generator=g)
        ext_ref = O.negative_rows(bi, si, B, S, W, N)                   # (B, N, W)
        ext = torch.full((B, W, Np), -1, dtype=torch.int32)
        perm = torch.full((B * W * (Np + K),), -1, dtype=torch.int32)
        row_ptr = torch.full((B * S + 1,), -1, dtype=torch.int32)
        work = torch.zeros(B * W * (Np + K) + 2 * B * S + 2, dtype=torch.int32)
        assert lib.cpc_nce_prepare(P(bi), P(si), P(ext), P(perm), P(row_ptr), P(work), B, S, K, N, None) == 0
        assert torch.equal(ext[:, :, :N], torch.sort(ext_ref.permute(0, 2, 1).to(torch.int32), dim=2).values)
        assert (ext[:, :, N:] >= 0).all() and (ext[:, :, N:] < B * S).all()
        sizes = (ctypes.c_long * 6)()
        assert lib.cpc_nce_layout(B, S, K, N, sizes) == 0
        saved = torch.full((sizes[0],), float("nan"))
        fscr = torch.full((sizes[1],), float("nan"))
        losses = torch.full((K,), float("nan")); acc = torch.full((K,), float("nan"))
        assert lib.cpc_nce_forward(P(c), P(z), P(wall), P(ext), P(saved), P(fscr), P(losses), P(acc), B, S, K, N, None) == 0
        leaves = {f"wPrediction.predictors.{k}.weight": heads[k].clone().requires_grad_(True) for k in range(K)}
        cr = c.clone().requires_grad_(True); zr = z.clone().requires_grad_(True)
        lr, ar = O.criterion_forward(leaves, cr, zr, ext_ref, K)
        assert (losses - lr[0]).abs().max().item() < 1e-5 * max(1.0, lr[0].abs().max().item())
        assert (acc - ar[0]).abs().max().item() < 1e-6
        gl = torch.randn(K)
        (lr[0] * gl).sum().backward()
        bscr = torch.full((sizes[2],), float("nan"))
        dc = torch.full((B, S, 256), float("nan")); dz = torch.full((B, S, 256), float("nan"))
        dwall = torch.full((K * 256, 256), float("nan"))
        assert lib.cpc_nce_backward(P(c), P(z), P(wall), P(ext), P(perm), P(row_ptr), P(saved), P(gl), P(bscr), P(dc), P(dz),
                                    P(dwall), B, S, K, N, None) == 0
        assert rel_err(dc, cr.grad) < 1e-5 and rel_err(dz, zr.grad) < 1e-5
        ref_dw = torch.cat([leaves[f"wPrediction.predictors.{k}.weight"].grad for k in range(K)], dim=0)
        assert rel_err(dwall, ref_dw) < 1e-5
    finally:
        lib.cpc_set_nce_fused(_L.DEFAULT_NCE_FUSED)
