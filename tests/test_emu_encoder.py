"""Encoder kernels (cpc_audio_amd/csrc/enc_conv0.hip, enc_conv.hip) executed on the host
SIMT emulator and compared with the oracle: validates tile indexing, MFMA fragment maps,
padding/ragged edges, the fused ChannelNorm epilogues and every gradient -- on CPU."""
import ctypes

import pytest
import torch


def _lib_default_mode():
    from cpc_audio_amd._lib import DEFAULT_MFMA_MODE
    return DEFAULT_MFMA_MODE

from emu_util import P, emu, rel_err
from oracle import cpc_oracle as O


def _params(seed=0):
    p = O.make_params(seed=seed)
    return p, [p[f"gEncoder.{n}{i}.{w}"].contiguous() for i in range(5)
               for n, w in (("conv", "weight"), ("conv", "bias"), ("batchNorm", "weight"), ("batchNorm", "bias"))]


def _oracle_encoder(p, wave, dz, relu_override=None):
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith("gEncoder")}
    acts = []
    z = O.encoder_forward(leaves, wave, collect=acts, relu_override=relu_override).permute(0, 2, 1)
    (z * dz).sum().backward()
    return z.detach(), [a.detach().permute(0, 2, 1).contiguous() for a in acts], leaves


@pytest.mark.parametrize("B,L,bm,mode", [(2, 1280, 0, 1), (1, 1370, 64, 1), (1, 1600, 128, 1), (3, 1290, 128, 0),
                                          (2, 1280, 0, 0), (1, 1600, 128, 2), (2, 1280, 0, 2), (1, 1370, 64, 2)])
def test_encoder_forward_backward_emulated(B, L, bm, mode):
    """mode 1: NT GEMMs on the bf16 pipe with 3-piece split operands; mode 0: exact-f32 MFMA; mode 2: fp16 pipe with
    scaled 2-piece split operands."""
    lib = emu()
    assert lib.cpc_set_conv_tile(bm) == 0
    assert lib.cpc_set_mfma_mode(mode) == 0
    try:
        torch.manual_seed(0)
        p, plist = _params()
        wave = O.make_waveform(B, L, seed=5)
        sizes = (ctypes.c_long * 22)()
        assert lib.cpc_encoder_layout(B, L, sizes) == 0
        saved = torch.full((sizes[0],), float("nan"))
        fscr = torch.full((max(1, sizes[1]),), float("nan"))
        Ls = [sizes[3 + i] for i in range(5)]
        z = torch.full((B, Ls[4], 256), float("nan"))
        parr = (ctypes.c_void_p * 20)(*[P(t) for t in plist])
        rc = lib.cpc_encoder_forward(P(wave), parr, P(saved), P(fscr), P(z), B, L, None)
        assert rc == 0
        dz = torch.randn(B, Ls[4], 256)
        # ReLU derivative of numerically tied pre-activations follows the device path (see oracle)
        ys = [saved[sizes[8 + i]: sizes[8 + i] + B * Ls[i] * 256].view(B, Ls[i], 256) for i in range(4)] + [z]
        z_ref, acts, leaves = _oracle_encoder(p, wave, dz, [(y > 0).permute(0, 2, 1) for y in ys])
        assert z_ref.shape == z.shape
        # intermediate activations y0..y3 live in the saved workspace
        for i in range(4):
            yi = saved[sizes[8 + i]: sizes[8 + i] + B * Ls[i] * 256].view(B, Ls[i], 256)
            assert (yi - acts[i]).abs().max().item() < 2e-5, f"layer {i}"
        assert (z - z_ref).abs().max().item() < 2e-5

        bscr = torch.full((sizes[2],), float("nan"))
        grads = [torch.full_like(t, float("nan")) for t in plist]
        garr = (ctypes.c_void_p * 20)(*[P(t) for t in grads])
        rc = lib.cpc_encoder_backward(P(wave), parr, P(saved), P(z), P(dz.contiguous()), P(bscr), garr,
                                      B, L, None)
        assert rc == 0
        names = [f"gEncoder.{n}{i}.{w}" for i in range(5)
                 for n, w in (("conv", "weight"), ("conv", "bias"), ("batchNorm", "weight"), ("batchNorm", "bias"))]
        bad = {}
        for n, g in zip(names, grads):
            ref = leaves[n].grad
            e = rel_err(g.view_as(ref), ref) if torch.isfinite(g).all() else float("inf")
            if not e < 2e-5:
                bad[n] = e
        assert not bad, bad
        # the two-stream entry point (weight gradients on their own stream) runs the same kernels: identical results
        bscr2 = torch.full((sizes[2],), float("nan"))
        grads2 = [torch.full_like(t, float("nan")) for t in plist]
        garr2 = (ctypes.c_void_p * 20)(*[P(t) for t in grads2])
        other = ctypes.c_void_p(0x10)                      # any handle != stream: the emulator ignores streams
        rc = lib.cpc_encoder_backward_streams(P(wave), parr, P(saved), P(z), P(dz.contiguous()), P(bscr2), garr2,
                                              B, L, None, other)
        assert rc == 0
        for a, b in zip(grads, grads2):
            assert torch.equal(a, b)
    finally:
        lib.cpc_set_conv_tile(0)
        lib.cpc_set_mfma_mode(_lib_default_mode())


@pytest.mark.parametrize("mode", [1, 2])
def test_fused_and_unfused_dgrad_agree_emulated(mode):
    """cpc_conv_layer_dgrad(fuse=1) (ReLU'/ChannelNorm backward in the GEMM epilogue) against the path the encoder
    uses (plain dgrad + cpc_norm_backward): same dprev and the same three small gradients."""
    lib = emu()
    assert lib.cpc_set_mfma_mode(mode) == 0
    try:
        torch.manual_seed(3)
        B, Lin, k, s, p = 2, 40, 4, 2, 1
        Lout = (Lin + 2 * p - k) // s + 1
        dx = torch.randn(B, Lout, 256) * 0.01
        w = torch.randn(256, 256, k) / 32.0
        xhat = torch.randn(B, Lin, 256)
        nw = 1.0 + 0.1 * torch.randn(256)
        y = (xhat * nw + 0.1 * torch.randn(256)).relu()
        rstd = torch.rand(B * Lin) + 0.5
        wd = torch.zeros(256 * k * 256 * 3 // 2)
        nblk = (B * (Lout + 1) + 31) // 32 * s + (B * Lin + 31) // 32 + 8
        colpart = torch.zeros(nblk * 768); tmp = torch.zeros(128 * 768)
        outs = []
        for fuse in (1, 0):
            dprev = torch.full((B, Lin, 256), float("nan")); small = torch.full((768,), float("nan"))
            amax = torch.zeros(2)
            if fuse:
                assert lib.cpc_conv_layer_dgrad(P(dx), P(w), P(wd), 1, P(xhat), P(y), P(rstd), P(nw), P(dprev), P(colpart),
                                                P(tmp), P(small), None, amax.data_ptr(), B, Lin, k, s, p, None) == 0
            else:
                raw = torch.full((B, Lin, 256), float("nan"))
                assert lib.cpc_conv_layer_dgrad(P(dx), P(w), P(wd), 0, None, None, None, None, P(raw), None, None, None,
                                                None, None, B, Lin, k, s, p, None) == 0
                assert lib.cpc_norm_backward(P(raw), P(xhat), P(y), P(rstd), P(nw), P(dprev), P(colpart), P(tmp), P(small),
                                             amax.data_ptr(), B * Lin, None) == 0
            outs.append((dprev, small, amax[0].item()))
        assert rel_err(outs[0][0], outs[1][0]) < 1e-6 and rel_err(outs[0][1], outs[1][1]) < 1e-5
        assert abs(outs[0][2] - outs[1][2]) <= 1e-6 * outs[1][2] and outs[1][2] == outs[1][0].abs().max().item()
    finally:
        lib.cpc_set_mfma_mode(_lib_default_mode())


@pytest.mark.parametrize("xs,ws", [(1e-12, 1e-9), (3e7, 2e4), (1.0, 1e-30)])
def test_fp16_split_scaling_extremes_emulated(xs, ws):
    """Mode 2 maps every operand into fp16's range with a power-of-two scale taken from its exact max|.|: operands of
    any magnitude must give the same normalised output as the exact-f32 MFMA path."""
    lib = emu()
    torch.manual_seed(5)
    B, Lin, k, s, p = 1, 64, 4, 2, 1
    Lout = (Lin + 2 * p - k) // s + 1
    x = (torch.randn(B, Lin, 256).relu() * xs).contiguous()
    w = (torch.randn(256, 256, k) * ws).contiguous()
    bias = torch.zeros(256); nw = torch.ones(256); nb = torch.zeros(256)
    outs = []
    for mode in (0, 2):
        assert lib.cpc_set_mfma_mode(mode) == 0
        try:
            wp = torch.zeros(256 * k * 256 * 3 // 2)
            y = torch.full((B, Lout, 256), float("nan")); xh = torch.full_like(y, float("nan")); rs = torch.zeros(B * Lout)
            assert lib.cpc_conv_layer_forward(P(x), P(w), P(bias), P(nw), P(nb), P(wp), P(y), P(xh), P(rs), B, Lin, k, s, p,
                                              None) == 0
            outs.append(xh)
        finally:
            lib.cpc_set_mfma_mode(_lib_default_mode())
    assert torch.isfinite(outs[1]).all()
    if xs * ws > 1e-25:        # below that the conv output itself underflows against the norm's epsilon in BOTH modes
        assert (outs[0] - outs[1]).abs().max().item() < 2e-5
