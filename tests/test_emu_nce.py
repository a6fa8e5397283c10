"""InfoNCE criterion kernels (csrc/nce.hip) on the host SIMT emulator vs the oracle."""
import ctypes

import pytest
import torch

from emu_util import P, emu, rel_err
from oracle import cpc_oracle as O


@pytest.mark.parametrize("B,S,K,N,scale,wide,fuse", [(2, 20, 12, 16, 1.0, 0, 1), (3, 19, 5, 32, 40.0, 0, 1), (2, 20, 12, 16, 1.0, 1, 1),
                                                         (2, 20, 12, 16, 1.0, 0, 0), (2, 21, 7, 32, 3000.0, 0, 1), (2, 21, 7, 32, 3000.0, 0, 0)])
def test_nce_forward_backward_emulated(B, S, K, N, scale, wide, fuse):
    """wide: the prediction GEMM on the 128 x 256 pipelined tile (cpc_set_gemm_split(3) forces it at test sizes).
    fuse: 1 = the forward's scoring kernel also forms the softmax-weighted row sums the backward needs, 0 (default) = the
    backward gathers the rows again.  scale 3000: logits hundreds apart, so the running reference of the fused kernel's softmax
    weights has to move (its rescaling path) and the softmax is saturated."""
    lib = emu()
    assert lib.cpc_set_gemm_split(3 if wide else 1) == 0 and lib.cpc_set_nce_fuse(fuse) == 0
    try:
        _nce_forward_backward(lib, B, S, K, N, scale)
    finally:
        lib.cpc_set_gemm_split(1)
        lib.cpc_set_nce_fuse(0)


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
    assert torch.equal(ext_k, ext_t) and torch.equal(row_ptr, row_ptr_ref)
    for r in range(B * S):
        lo, hi = int(row_ptr[r]), int(row_ptr[r + 1])
        assert sorted(perm[lo:hi].tolist()) == perm_ref[lo:hi].tolist()
    assert lib.cpc_nce_backward(P(c), P(z), P(wall), P(ext_t), P(perm), P(row_ptr), P(saved), P(gl), P(bscr), P(dc),
                                P(dz), P(dwall), B, S, K, N, None) == 0
    assert rel_err(dc, cr.grad) < 1e-5
    assert rel_err(dz, zr.grad) < 1e-5
    ref_dw = torch.cat([leaves[f"wPrediction.predictors.{k}.weight"].grad for k in range(K)], dim=0)
    assert rel_err(dwall, ref_dw) < 1e-5


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
