"""Encoder kernels (cpc_audio_amd/csrc/enc_conv0.hip, enc_conv.hip) executed on the host
SIMT emulator and compared with the oracle: validates tile indexing, MFMA fragment maps,
padding/ragged edges, the fused ChannelNorm epilogues and every gradient -- on CPU."""
import ctypes

from cpc_audio_amd import _lib as _L

import pytest
import torch


def _lib_default_pipeline():
    from cpc_audio_amd._lib import DEFAULT_DMA_PIPELINE
    return DEFAULT_DMA_PIPELINE


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
    # (oneDNN off: torch 2.10's oneDNN conv1d backward returns wrong input gradients for the first six steps of every sequence at
    # some shapes -- e.g. B = 2, C = 256, k 8 / s 4 / p 2, 58 or 121 input steps: 1.6e-2 off its own float64 and native-fp32
    # results -- which the odd window lengths of this file hit; the reference's shapes and the fixtures' are not among them)
    with torch.backends.mkldnn.flags(enabled=False):
        z = O.encoder_forward(leaves, wave, collect=acts, relu_override=relu_override).permute(0, 2, 1)
        (z * dz).sum().backward()
    return z.detach(), [a.detach().permute(0, 2, 1).contiguous() for a in acts], leaves


def _saved_acts(lib, saved, B, L, Ls):
    """fp32 copies of y0..y3 whatever their storage (mode 3 keeps y0, y1 as fp16 piece pairs)"""
    out = []
    for i in range(4):
        y = torch.full((B, Ls[i], 256), float("nan"))
        assert lib.cpc_encoder_saved_activation(P(saved), i, P(y), B, L, None) == 0
        out.append(y)
    return out


@pytest.mark.parametrize("B,L,bm,mode", [(2, 1280, 0, 1), (1, 1370, 64, 1), (1, 1600, 128, 1), (3, 1290, 128, 0),
                                          (2, 1280, 0, 0), (1, 1600, 128, 2), (2, 1280, 0, 2), (1, 1370, 64, 2),
                                          (2, 1280, 0, 3), (1, 1370, 64, 3), (3, 1290, 128, 3), (2, 1280, 0, 32),
                                          (1, 1370, 0, 132), (2, 1280, 0, 30), (2, 1280, 0, 34), (1, 1370, 64, 34),
                                          (3, 1290, 128, 34), (4, 2560, 0, 234), (2, 2560, 32, 334), (1, 1370, 64, 334), (4, 2560, 0, 434), (2, 1280, 0, 534), (2, 2560, 0, 634),
                                          (2, 2560, 0, 734), (3, 1290, 0, 734), (1, 170, 0, 734), (2, 485, 0, 734), (1, 170, 0, 0),
                                          # 978 / 2594 samples: layer 0's output is 4 L1 + 3 steps long -- its last step feeds no window of
                                          # layer 1 and must get a ZERO gradient (found by a shape sweep in round 5: it got none);
                                          # 290 / 2319: shapes at which torch's oneDNN conv backward is wrong (_oracle_encoder)
                                          (2, 978, 0, 734), (2, 978, 0, 0), (1, 2594, 0, 34), (2, 290, 0, 734), (2, 2319, 0, 3)])
def test_encoder_forward_backward_emulated(B, L, bm, mode):
    """mode 1: NT GEMMs on the bf16 pipe with 3-piece split operands; mode 0: exact-f32 MFMA; mode 2: fp16 pipe with
    scaled 2-piece split operands; mode 3 (default): mode 2 + layers 1, 2 on the DMA kernel reading H2 activations
    (ragged lengths: partial 128-row tiles, padding rows from the zero buffer)."""
    lib = emu()
    # mode 32: mode 3 with conv2 on the DMA kernel as well (what B >= ~100 selects); 132: that with two 32-k LDS stages;
    # mode 30: mode 3 with layer 1's gradient kept fp32 (cpc_set_h2_dx(0): register-staged data gradient) instead of H2 storage;
    # mode 34: every activation y0..y3 and every gradient dx1..dx4 in H2 storage (cpc_set_h2_layers(4)): the short layers on the
    # register-staged tiles fed H2 rows as they lie (32 / 64 / 128-row tiles), all weight gradients on the DMA kernel, one reduction
    # (234: that with the weight-gradient splits down to 128 rows, so that the short layers run several splits + the batched reduction)
    # (334: mode 34 with the short tiles on the software-pipelined 16-k schedule, cpc_set_conv_small_pipe(1))
    # (434: mode 234 with the weight-gradient kernel on four 16-row LDS stages, cpc_set_wgrad_dma_stages(4))
    min_rows = 128 if mode in (234, 434) else 512
    assert lib.cpc_set_conv_small_pipe(1 if mode == 334 else 0) == 0
    assert lib.cpc_set_wgrad_dma_stages(4 if mode == 434 else 2) == 0
    # (534: mode 34 with layer 2 on the DMA-fed kernels as well -- what B >= ~100 selects: conv2 writes y2 in H2 from the DMA
    # kernel's epilogue, its data gradient runs on the DMA kernel and leaves max|dy| for layer 1's H2 gradient)
    assert lib.cpc_set_dma_layer2(1 if mode == 534 else 0) == 0
    # (634: mode 34 with the short layers' data gradients on 128 x 128 tiles, two workgroups per row tile: cpc_set_dgrad_nsplit)
    assert lib.cpc_set_dgrad_nsplit(1 if mode == 634 else 0) == 0
    # (734: mode 34 with the short layers' FORWARD on 128 x 128 tiles, two workgroups per row tile exchanging their ChannelNorm
    # statistics through global memory: cpc_set_fwd_nsplit; the ragged case ends in a partial row tile)
    assert lib.cpc_set_fwd_nsplit(1 if mode == 734 else 0, -1) == 0
    mode = 34 if mode in (234, 334, 434, 534, 634, 734) else mode
    assert lib.cpc_set_wgrad_dma_min_rows(min_rows) == 0
    h2_layers, pipe = (4, 0) if mode == 34 else ((2, mode // 100) if mode >= 32 else (0, 0))
    h2_dx = 0 if mode == 30 else 1
    mode = 3 if mode >= 30 else mode
    assert lib.cpc_set_h2_dx(h2_dx) == 0
    assert lib.cpc_set_conv_tile(bm) == 0
    assert lib.cpc_set_mfma_mode(mode) == 0
    assert lib.cpc_set_h2_layers(h2_layers) == 0 and lib.cpc_set_dma_pipeline(pipe) == 0
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
        ys = _saved_acts(lib, saved, B, L, Ls) + [z]
        O.tie_report()
        z_ref, acts, leaves = _oracle_encoder(p, wave, dz, [(y > 0).permute(0, 2, 1) for y in ys])
        # how many elements actually took the device's ReLU derivative: the rounding-tie rate of two correct fp32 paths (about one
        # per million), never a systematic share of the window; none may disagree outside the window
        ties = O.tie_report()
        print(f"relu ties: {ties}")
        assert O.tie_ok(ties) and ties["disagree_outside"] == 0, ties
        assert z_ref.shape == z.shape
        # intermediate activations y0..y3 live in the saved workspace
        for i in range(4):
            assert (ys[i] - acts[i]).abs().max().item() < 2e-5, f"layer {i}"
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
        lib.cpc_set_h2_layers(0)
        lib.cpc_set_dma_pipeline(_lib_default_pipeline())
        lib.cpc_set_h2_dx(1)
        lib.cpc_set_wgrad_dma_min_rows(512)
        lib.cpc_set_conv_small_pipe(_L.DEFAULT_CONV_SMALL_PIPE)
        lib.cpc_set_wgrad_dma_stages(_L.DEFAULT_WGRAD_DMA_STAGES)
        lib.cpc_set_dma_layer2(0)
        lib.cpc_set_dgrad_nsplit(_L.DEFAULT_DGRAD_NSPLIT)
        lib.cpc_set_fwd_nsplit(_L.DEFAULT_FWD_NSPLIT, -1)


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


@pytest.mark.parametrize("B,Lin,k,s,p,bm,y_h2,pipe", [(1, 300, 8, 4, 2, 128, True, 0), (2, 131, 4, 2, 1, 256, False, 0),
                                                    (1, 70, 8, 4, 2, 256, True, 1), (1, 300, 4, 2, 1, 128, False, 1),
                                                    (2, 1024, 8, 4, 2, 256, True, 2), (1, 512, 4, 2, 1, 256, False, 2),
                                                    (1, 300, 8, 4, 2, 256, True, 3), (2, 131, 4, 2, 1, 256, False, 3),
                                                    (1, 300, 8, 4, 2, 256, True, 4), (2, 131, 4, 2, 1, 256, False, 5),
                                                    (1, 300, 8, 4, 2, 256, True, 6), (2, 131, 4, 2, 1, 256, False, 6),
                                                    (1, 300, 8, 4, 2, 256, True, 2),        # Lout % 256 != 0: the pair walk falls back
                                                    (1, 300, 8, 4, 2, 256, True, 7), (2, 131, 4, 2, 1, 256, False, 7)])   # 7: four 128 x 128 waves
def test_dma_conv_kernel_matches_the_register_staged_kernel_emulated(B, Lin, k, s, p, bm, y_h2, pipe):
    """cpc_conv_gemm_forward_h2 (both operands DMA'd into XOR-swizzled LDS rows, H2 storage) against
    cpc_conv_layer_forward in mode 2 on the same fp32 data: same pieces, same products, same ChannelNorm -- results agree
    to fp32 summation-order noise.  Covers padding rows (zero buffer), ragged last tiles, both tile heights, both output
    storages, and the H2 encode / decode round trip."""
    lib = emu()
    torch.manual_seed(7)
    Lout = (Lin + 2 * p - k) // s + 1
    x = torch.randn(B, Lin, 256).relu().contiguous()
    w = (torch.randn(256, 256, k) / (16.0 * k ** 0.5)).contiguous()
    bias = 0.1 * torch.randn(256); nw = 1 + 0.1 * torch.randn(256); nb = 0.1 * torch.randn(256)
    assert lib.cpc_set_mfma_mode(2) == 0
    try:
        wp = torch.zeros(256 * k * 256 * 3 // 2)
        y_ref = torch.full((B, Lout, 256), float("nan")); xh_ref = torch.full_like(y_ref, float("nan")); rs_ref = torch.zeros(B * Lout)
        assert lib.cpc_conv_layer_forward(P(x), P(w), P(bias), P(nw), P(nb), P(wp), P(y_ref), P(xh_ref), P(rs_ref), B, Lin, k,
                                          s, p, None) == 0
    finally:
        lib.cpc_set_mfma_mode(_lib_default_mode())
    xamax = x.abs().max().view(1).clone() * 1.7                 # any bound works
    x_h2 = torch.zeros(B, Lin, 256)
    assert lib.cpc_h2_encode(P(x), P(x_h2), B * Lin, P(xamax), None) == 0
    back = torch.full_like(x, float("nan"))
    assert lib.cpc_h2_decode(P(x_h2), P(back), B * Lin, P(xamax), None) == 0
    assert (back - x).abs().max().item() <= 2.0 ** -21 * x.abs().max().item()
    wq = torch.zeros(256 * k * 256 + 64)
    assert lib.cpc_conv_weight_relayout_h2(P(w), P(wq), k, None) == 0
    zeros = torch.zeros(32)
    yamax = (15.968719 * nw.abs().max() + nb.abs().max()).view(1).clone()
    y = torch.full((B, Lout, 256), float("nan")); xh = torch.full_like(y, float("nan")); rs = torch.zeros(B * Lout)
    # 0: four 16-k LDS stages, 1: two 32-k stages, 2: the pair walk (every input row DMA'd once; 256-row tiles inside one sequence)
    assert lib.cpc_set_dma_pipeline(pipe) == 0
    try:
        assert lib.cpc_conv_gemm_forward_h2(P(x_h2), P(wq), P(bias), P(nw), P(nb), P(y), P(xh), P(rs), P(xamax),
                                            P(yamax) if y_h2 else None, P(zeros), B, Lin, k, s, p, bm, None) == 0
    finally:
        lib.cpc_set_dma_pipeline(_lib_default_pipeline())
    if y_h2:
        dec = torch.full_like(y, float("nan"))
        assert lib.cpc_h2_decode(P(y), P(dec), B * Lout, P(yamax), None) == 0
        y = dec
    assert (xh - xh_ref).abs().max().item() < 2e-5
    assert (rs - rs_ref).abs().max().item() < 2e-5 * rs_ref.abs().max().item()
    assert (y - y_ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,L", [(2, 1280), (1, 1370), (2, 978)])          # (978: an input step of layer 1 that no window touches)
def test_bf16_storage_encoder_emulated(B, L):
    """cpc_set_mfma_mode(4), the bf16-storage variant of BASELINE configs[1]: activations y0..y3, the saved xhat1..4 and every
    gradient tensor of the encoder are bf16 (half the bytes), weights are rounded to bf16 by the re-layout, every product
    is ONE bf16 MFMA with fp32 accumulation, ChannelNorm statistics stay fp32, z is written in fp32.
    The parity bar is NOT the fp32 path's 1e-4: a bf16 value carries 8 significant bits (relative rounding 2^-9 = 2e-3) and
    each of the five layers rounds its input once.  Stated bound, checked here and on the GPU: max|z - z_ref| < 6e-2 on
    z = O(1) and < 2e-2 relative in the Frobenius norm; every parameter gradient within 6e-2 relative (ReLU masks follow the
    device inside the band the bf16 rounding makes ambiguous)."""
    lib = emu()
    assert lib.cpc_set_mfma_mode(4) == 0
    try:
        p, plist = _params()
        wave = O.make_waveform(B, L, seed=5)
        sizes = (ctypes.c_long * 22)()
        assert lib.cpc_encoder_layout(B, L, sizes) == 0
        saved = torch.full((sizes[0],), float("nan"))
        fscr = torch.full((max(1, sizes[1]),), float("nan"))
        Ls = [sizes[3 + i] for i in range(5)]
        z = torch.full((B, Ls[4], 256), float("nan"))
        parr = (ctypes.c_void_p * 20)(*[P(t) for t in plist])
        assert lib.cpc_encoder_forward(P(wave), parr, P(saved), P(fscr), P(z), B, L, None) == 0
        torch.manual_seed(1)
        dz = torch.randn(B, Ls[4], 256)
        ys = _saved_acts(lib, saved, B, L, Ls) + [z]
        leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith("gEncoder")}
        acts = []
        z_ref = O.encoder_forward(leaves, wave, collect=acts, relu_override=[(y > 0).permute(0, 2, 1) for y in ys],
                                  tie_eps=0.08).permute(0, 2, 1)
        (z_ref * dz).sum().backward()
        err = (z - z_ref.detach()).abs().max().item()
        rel = rel_err(z, z_ref.detach())
        assert torch.isfinite(z).all() and err < 6e-2 and rel < 2e-2, (err, rel)
        for i in range(4):
            assert rel_err(ys[i], acts[i].detach().permute(0, 2, 1)) < 2e-2, i
        bscr = torch.full((sizes[2],), float("nan"))
        grads = [torch.full_like(t, float("nan")) for t in plist]
        garr = (ctypes.c_void_p * 20)(*[P(t) for t in grads])
        assert lib.cpc_encoder_backward(P(wave), parr, P(saved), P(z), P(dz.contiguous()), P(bscr), garr, B, L, None) == 0
        names = [f"gEncoder.{n}{i}.{w}" for i in range(5)
                 for n, w in (("conv", "weight"), ("conv", "bias"), ("batchNorm", "weight"), ("batchNorm", "bias"))]
        worst = {}
        for n, g in zip(names, grads):
            ref = leaves[n].grad
            worst[n] = rel_err(g.view_as(ref), ref) if torch.isfinite(g).all() else float("inf")
        bad = {k: v for k, v in worst.items() if not v < 6e-2}
        assert not bad, bad
    finally:
        lib.cpc_set_mfma_mode(_lib_default_mode())
