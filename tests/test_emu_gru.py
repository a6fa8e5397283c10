"""GEMM + GRU kernels (csrc/gemm.hip, gru.hip) on the host SIMT emulator vs the oracle."""
import ctypes

import pytest
import torch


def _lib_default_gru_mode():
    from cpc_audio_amd._lib import DEFAULT_GRU_MODE
    return DEFAULT_GRU_MODE


def _lib_default_mode():
    from cpc_audio_amd._lib import DEFAULT_MFMA_MODE
    return DEFAULT_MFMA_MODE

from cpc_audio_amd import _lib as _L
from emu_util import P, emu, rel_err
from oracle import cpc_oracle as O


@pytest.mark.parametrize("mode", [1, 0])
def test_gemm_nt_tn_emulated(mode):
    lib = emu()
    assert lib.cpc_set_mfma_mode(mode) == 0
    try:
        _gemm_checks(lib)
    finally:
        lib.cpc_set_mfma_mode(_lib_default_mode())


def _gemm_checks(lib):
    torch.manual_seed(0)
    M, N, K = 200, 256, 96
    A = torch.randn(M, K); Bm = torch.randn(N, K); bias = torch.randn(N)
    C = torch.full((M, N), float("nan"))
    assert lib.cpc_gemm_nt(P(A), K, P(Bm), K, P(bias), P(C), N, M, N, K, None) == 0
    assert rel_err(C, A @ Bm.t() + bias) < 1e-6
    M, N1, N2 = 300, 128, 256
    A = torch.randn(M, N1); Bm = torch.randn(M, N2)
    part = torch.full((lib.cpc_gemm_tn_scratch_floats(M, N1, N2),), float("nan"))
    C = torch.full((N1, N2), float("nan"))
    assert lib.cpc_gemm_tn(P(A), N1, P(Bm), N2, P(part), P(C), M, N1, N2, 0, None) == 0
    assert rel_err(C, A.t() @ Bm) < 1e-6
    C0 = C.clone()
    assert lib.cpc_gemm_tn(P(A), N1, P(Bm), N2, P(part), P(C), M, N1, N2, 1, None) == 0
    assert rel_err(C, 2 * C0) < 1e-6


def test_narrow_products_with_split_k_emulated():
    """N = 256 products (the GRU's dx = dGi . W_ih, the criterion's dc) on the wide tile with the K walk split over several
    workgroups + one reduction launch (gemm.hip, SplitK): cpc_set_gemm_split(3) forces it at test sizes; same results as
    the plain tiles to fp32 rounding (the oracle comparison inside _run_gru: 1e-5)."""
    lib = emu()
    assert lib.cpc_set_gemm_split(3) == 0
    try:
        a = _run_gru(lib, 3, 6, 2, False)
    finally:
        lib.cpc_set_gemm_split(1)
    b = _run_gru(lib, 3, 6, 2, False)
    assert (a[2] - b[2]).abs().max().item() <= 1e-6 * b[2].abs().max().item()      # dx
    assert not torch.equal(a[2], b[2])                                              # the path did change


def test_gemm_nt_wide_tile_emulated():
    """The 128 x 256 pipelined tile of the plain NT GEMM on three bf16 pieces (taken for N % 256 == 0, K % 64 == 0 once the
    grid fills the chip; cpc_set_gemm_split(3) forces it at test sizes): ragged M, two column tiles, bias."""
    lib = emu()
    assert lib.cpc_set_mfma_mode(1) == 0 and lib.cpc_set_gemm_split(3) == 0
    try:
        torch.manual_seed(1)
        M, N, K = 200, 512, 128
        A = torch.randn(M, K); Bm = torch.randn(N, K); bias = torch.randn(N)
        C = torch.full((M, N), float("nan"))
        assert lib.cpc_gemm_nt(P(A), K, P(Bm), K, P(bias), P(C), N, M, N, K, None) == 0
        assert rel_err(C, A @ Bm.t() + bias) < 1e-6
    finally:
        lib.cpc_set_mfma_mode(_lib_default_mode())
        lib.cpc_set_gemm_split(1)


@pytest.mark.parametrize("B,S,use_h0", [(3, 6, False), (20, 5, True)])
def test_gru_persistent_equals_stepwise_emulated(B, S, use_h0):
    """The single-launch recurrence (polling hand-over between co-resident workgroups) and the
    launch-per-step path run the same arithmetic in the same order: outputs must be bit-identical."""
    lib = emu()
    outs = []
    for mode in (0, 1):
        assert lib.cpc_set_gru_mode(mode) == 0
        try:
            outs.append(_run_gru(lib, B, S, 2, use_h0, check=(mode == 1)))
        finally:
            lib.cpc_set_gru_mode(_lib_default_gru_mode())
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,S,nl,use_h0", [(3, 6, 2, False), (17, 4, 1, True), (2, 5, 2, True), (2, 3, 3, True), (20, 1, 2, False)])
def test_gru_forward_backward_emulated(B, S, nl, use_h0):
    _run_gru(emu(), B, S, nl, use_h0)


def _run_gru(lib, B, S, nl, use_h0, check=True):
    torch.manual_seed(1)
    p = O.make_params(seed=3, n_levels_gru=nl)
    names = [f"gAR.baseNet.{w}_l{l}" for l in range(nl) for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    plist = [p[n].contiguous() for n in names]
    x = torch.randn(B, S, 256)
    h0 = 0.5 * torch.randn(nl, B, 256) if use_h0 else None
    sizes = (ctypes.c_long * 3)()
    assert lib.cpc_gru_layout(B, S, nl, sizes) == 0
    saved = torch.full((sizes[0],), float("nan"))
    fscr = torch.full((sizes[1],), float("nan"))
    y = torch.full((B, S, 256), float("nan"))
    hN = torch.full((nl, B, 256), float("nan"))
    parr = (ctypes.c_void_p * (4 * nl))(*[P(t) for t in plist])
    assert lib.cpc_gru_forward(P(x), P(h0), parr, P(saved), P(fscr), P(y), P(hN), B, S, nl, None) == 0
    leaves = {n: p[n].clone().requires_grad_(True) for n in names}
    xr = x.clone().requires_grad_(True)
    yr, hr = O.gru_forward(leaves, xr, n_levels=nl, h0=h0)
    assert (y - yr).abs().max().item() < 1e-5
    assert (hN - hr).abs().max().item() < 1e-5
    dy = torch.randn(B, S, 256)
    (yr * dy).sum().backward()
    bscr = torch.full((sizes[2],), float("nan"))
    dx = torch.full((B, S, 256), float("nan"))
    grads = [torch.full_like(t, float("nan")) for t in plist]
    garr = (ctypes.c_void_p * (4 * nl))(*[P(t) for t in grads])
    assert lib.cpc_gru_backward(P(x), P(h0), parr, P(saved), P(y), P(dy), P(bscr), P(dx), garr, B, S, nl, None) == 0
    if check:
        assert rel_err(dx, xr.grad) < 1e-5
        bad = {n: rel_err(g, leaves[n].grad) for n, g in zip(names, grads) if not rel_err(g, leaves[n].grad) < 1e-5}
        assert not bad, bad
    return [y, hN, dx] + grads


@pytest.mark.parametrize("B,S", [(3, 6), (20, 5)])
def test_gru_fp16_split_forward_emulated(B, S):
    """cpc_set_gru_mode(2): the recurrent products of the persistent forward on the fp16 pipe (two-piece split
    operands, three MFMAs per product).  Same tolerance against the oracle as the exact-f32 path (_run_gru: 1e-5 on y
    and hN, 1e-5 relative on every gradient), and within 2e-6 of the f32 path itself."""
    lib = emu()
    outs = []
    for mode in (1, 2):
        assert lib.cpc_set_gru_mode(mode) == 0
        try:
            outs.append(_run_gru(lib, B, S, 2, False))
        finally:
            lib.cpc_set_gru_mode(_lib_default_gru_mode())
    assert (outs[0][0] - outs[1][0]).abs().max().item() < 2e-6
    assert not torch.equal(outs[0][0], outs[1][0])          # the mode did switch the arithmetic
    # with a caller-supplied h0 (|h| not bounded by 1) mode 2 keeps the exact-f32 products
    assert lib.cpc_set_gru_mode(2) == 0
    try:
        a = _run_gru(lib, B, S, 2, True)
    finally:
        lib.cpc_set_gru_mode(_lib_default_gru_mode())
    b = _run_gru(lib, B, S, 2, True)
    assert torch.equal(a[0], b[0])


def test_persistent_recurrence_with_one_batch_tile_per_xcd_emulated():
    """cpc_set_gru_xcd_pack: the packed workgroup numbering (grid of 256 * ceil(tiles / 8), tile (slot / 32) * 8 + id % 8,
    surplus workgroups exit) computes exactly what the interleaved numbering computes; 2 forces it on the emulator, which
    has no XCDs to report.  B = 20: two batch tiles, the second one ragged."""
    lib = emu()
    outs = []
    for pack in (0, 2):
        assert lib.cpc_set_gru_xcd_pack(pack) == 0
        try:
            outs.append(_run_gru(lib, 20, 5, 2, False))
        finally:
            lib.cpc_set_gru_xcd_pack(0)
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    assert lib.cpc_set_gru_xcd_pack(3) != 0


def test_persistent_recurrence_with_plain_first_looks_emulated():
    """cpc_set_gru_poll_plain: which waves take their first look at a hand-over fragment through their XCD's L2 -- a matter of
    memory scope only, the same bits for every mask (on the emulator both loads are host atomics: this pins the plumbing)."""
    lib = emu()
    outs = []
    for mask in (0, 15, 3):
        assert lib.cpc_set_gru_poll_plain(mask) == 0
        try:
            outs.append(_run_gru(lib, 20, 5, 2, False))
        finally:
            lib.cpc_set_gru_poll_plain(_L.DEFAULT_GRU_POLL_PLAIN)
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])) and all(torch.equal(a, b) for a, b in zip(outs[0], outs[2]))
    assert lib.cpc_set_gru_poll_plain(32) != 0


def test_persistent_recurrence_with_xcd_local_handover_emulated():
    """cpc_set_gru_xcd_local: packed numbering + the placement check at the start of the launch (every workgroup of a tile clears
    the bit of its XCD and counts itself in) + plain stores where a tile sits on one XCD -- the emulator reports XCD 0 for every
    workgroup, so with the packed numbering forced (2) the tiles take the local path: same bits; forward and backward bits apart."""
    lib = emu()
    outs = []
    for mask, pack in ((0, 0), (15, 2), (5, 2), (10, 2), (3, 0), (3, 2)):
        assert lib.cpc_set_gru_xcd_local(mask) == 0 and lib.cpc_set_gru_xcd_pack(pack) == 0
        try:
            outs.append(_run_gru(lib, 20, 5, 2, False))
        finally:
            lib.cpc_set_gru_xcd_local(_L.DEFAULT_GRU_XCD_LOCAL)
            lib.cpc_set_gru_xcd_pack(0)
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o))
    assert lib.cpc_set_gru_xcd_local(16) != 0


def test_persistent_recurrence_in_chunks_of_batch_tiles_emulated():
    """A batch whose workgroups cannot all be resident runs as several persistent launches over chunks of 16-sequence tiles
    (B = 256 on MI355X: two launches of 8 tiles).  cpc_set_gru_chunk_tiles(1) forces one tile per launch: same bits."""
    lib = emu()
    outs = []
    # (cap, tiles per workgroup): everything resident; one tile per launch, serial chunks (rounds 2-4); TWO tiles per workgroup --
    # launches of 1 slot x 2 tiles: (0, 1), then (2, none) -- and 2 slots x 2 tiles in ONE launch: (0, 2), (1, none).  With two tiles
    # a workgroup alternates between its tiles inside every time step (what B = 256 runs as on MI355X); same bits, forward and backward.
    for cap, tpw in ((0, 2), (1, 1), (1, 2), (2, 2)):
        assert lib.cpc_set_gru_chunk_tiles(cap) == 0 and lib.cpc_set_gru_tiles_per_wg(tpw) == 0
        try:
            outs.append(_run_gru(lib, 36, 5, 2, False))          # three tiles, the last one ragged
        finally:
            lib.cpc_set_gru_chunk_tiles(0)
            lib.cpc_set_gru_tiles_per_wg(2)
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o))
    # ... and with an initial state (the exact-f32 forward kernel; h0 rows of the second tile)
    outs = []
    for cap, tpw in ((0, 2), (1, 2)):
        assert lib.cpc_set_gru_chunk_tiles(cap) == 0 and lib.cpc_set_gru_tiles_per_wg(tpw) == 0
        try:
            outs.append(_run_gru(lib, 20, 4, 2, True))
        finally:
            lib.cpc_set_gru_chunk_tiles(0)
            lib.cpc_set_gru_tiles_per_wg(2)
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    assert lib.cpc_set_gru_tiles_per_wg(3) != 0


def test_gru_backward_with_early_coefficients_emulated():
    """cpc_gru_backward_coef (forward-only part + pre-filled hand-over buffers, run ahead of time by the overlapped train
    loops) + cpc_gru_backward_with_coef / _streams give bit-identical results to the one-call backward."""
    lib = emu()
    B, S, nl = 5, 6, 2
    torch.manual_seed(2)
    p = O.make_params(seed=3, n_levels_gru=nl)
    names = [f"gAR.baseNet.{w}_l{l}" for l in range(nl) for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    plist = [p[n].contiguous() for n in names]
    x = torch.randn(B, S, 256)
    sizes = (ctypes.c_long * 3)()
    assert lib.cpc_gru_layout(B, S, nl, sizes) == 0
    saved = torch.full((sizes[0],), float("nan"))
    fscr = torch.full((sizes[1],), float("nan"))
    y = torch.full((B, S, 256), float("nan"))
    hN = torch.full((nl, B, 256), float("nan"))
    parr = (ctypes.c_void_p * (4 * nl))(*[P(t) for t in plist])
    assert lib.cpc_gru_forward(P(x), None, parr, P(saved), P(fscr), P(y), P(hN), B, S, nl, None) == 0
    dy = torch.randn(B, S, 256)

    def backward(kind):
        bscr = torch.full((sizes[2],), float("nan"))
        dx = torch.full((B, S, 256), float("nan"))
        grads = [torch.full_like(t, float("nan")) for t in plist]
        garr = (ctypes.c_void_p * (4 * nl))(*[P(t) for t in grads])
        if kind == "plain":
            rc = lib.cpc_gru_backward(P(x), None, parr, P(saved), P(y), P(dy), P(bscr), P(dx), garr, B, S, nl, None)
        else:
            n = lib.cpc_gru_coef_floats(B, S, nl)
            assert n > 0
            coef = torch.full((n,), float("nan"))
            assert lib.cpc_gru_backward_coef(None, parr, P(saved), P(y), P(coef), 0, B, S, nl, None) == 0
            if kind == "coef":
                rc = lib.cpc_gru_backward_with_coef(P(x), None, parr, P(saved), P(y), P(dy), P(coef), P(bscr), P(dx), garr,
                                                    B, S, nl, None)
            else:
                rc = lib.cpc_gru_backward_streams(P(x), None, parr, P(saved), P(y), P(dy), P(coef), P(bscr), P(dx), garr,
                                                  B, S, nl, None, ctypes.c_void_p(0x10))
        assert rc == 0
        return [dx] + grads

    ref = backward("plain")
    for kind in ("coef", "streams"):
        for a, b in zip(ref, backward(kind)):
            assert torch.equal(a, b), kind
    # the coefficients written by the forward recurrence itself (cpc_gru_forward_coef) instead of read back by
    # gru_bwd_coef_kernel: the same values, the same backward
    n = lib.cpc_gru_coef_floats(B, S, nl)
    coef_k = torch.full((n,), float("nan"))
    assert lib.cpc_gru_backward_coef(None, parr, P(saved), P(y), P(coef_k), 0, B, S, nl, None) == 0
    coef_f = torch.full((n,), float("nan"))
    y2 = torch.full((B, S, 256), float("nan")); hN2 = torch.full((nl, B, 256), float("nan"))
    saved2 = torch.full_like(saved, float("nan"))
    assert lib.cpc_gru_forward_coef(P(x), None, parr, P(saved2), P(fscr), P(y2), P(hN2), P(coef_f), B, S, nl, None) == 0
    assert torch.equal(y2, y) and torch.equal(hN2, hN)
    assert lib.cpc_gru_backward_coef(None, parr, P(saved2), P(y2), P(coef_f), 1, B, S, nl, None) == 0
    assert torch.equal(coef_f.view(torch.int32), coef_k.view(torch.int32))      # (bitwise: the hand-over buffers hold the 0xFFFFFFFF fill)
    bscr = torch.full((sizes[2],), float("nan"))
    dx = torch.full((B, S, 256), float("nan"))
    grads = [torch.full_like(t, float("nan")) for t in plist]
    garr = (ctypes.c_void_p * (4 * nl))(*[P(t) for t in grads])
    assert lib.cpc_gru_backward_with_coef(P(x), None, parr, P(saved2), P(y2), P(dy), P(coef_f), P(bscr), P(dx), garr,
                                          B, S, nl, None) == 0
    for a, b in zip(ref, [dx] + grads):
        assert torch.equal(a, b)
