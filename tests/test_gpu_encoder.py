"""Encoder HIP kernels on a real MI355X vs the CPU oracle (through the C ABI)."""
import ctypes

from cpc_audio_amd import _lib as _L

import pytest
import torch


def _lib_default_mode():
    from cpc_audio_amd._lib import DEFAULT_MFMA_MODE
    return DEFAULT_MFMA_MODE

from oracle import cpc_oracle as O

pytestmark = pytest.mark.gpu
from cpc_audio_amd._lib import DEFAULT_DMA_PIPELINE as DMA_PIPELINE_DEFAULT  # noqa: E402


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _names():
    return [f"gEncoder.{n}{i}.{w}" for i in range(5)
            for n, w in (("conv", "weight"), ("conv", "bias"), ("batchNorm", "weight"), ("batchNorm", "bias"))]


def _run(lib, B, L, dev, pseed=0, bm=0, mode=1, dma=None, onednn=True):
    from cpc_audio_amd._lib import ptr as P
    h2_layers, stages, small_pipe = 0, 2, 0
    nsplit = 1 if mode == 634 else 0             # 634: + the short layers' data gradients on 128 x 128 tiles (cpc_set_dgrad_nsplit)
    assert lib.cpc_set_dgrad_nsplit(nsplit) == 0
    # 734: + the short layers' FORWARD on 128 x 128 tiles with the ChannelNorm statistics exchanged between the two workgroups of a
    # row tile (cpc_set_fwd_nsplit; 1 = wherever the layer reads H2 input, i.e. conv2..4 whatever their size)
    assert lib.cpc_set_fwd_nsplit(1 if mode == 734 else 0, -1) == 0
    if mode == 734:
        h2_layers, mode = 4, 3
    dma2 = 1 if mode == 534 else 0               # 534: + layer 2 on the DMA-fed kernels (what B >= ~100 selects by itself: the B = 128 case)
    assert lib.cpc_set_dma_layer2(dma2) == 0
    if mode in (34, 334, 434, 534, 634):  # mode 3 with every activation and every gradient of layers 1..4 in H2 storage (cpc_set_h2_layers(4));
        stages = 4 if mode == 434 else 2         # 434: + the weight-gradient kernel on four 16-row LDS stages
        small_pipe = 1 if mode == 334 else 0     # 334: + the short tiles on the software-pipelined 16-k schedule
        mode, h2_layers = 3, 4
    assert lib.cpc_set_h2_layers(h2_layers) == 0 and lib.cpc_set_wgrad_dma_stages(stages) == 0
    assert lib.cpc_set_conv_small_pipe(small_pipe) == 0
    p = O.make_params(seed=pseed)
    plist = [p[n].contiguous().to(dev) for n in _names()]
    wave = O.make_waveform(B, L, seed=5)
    sizes = (ctypes.c_long * 22)()
    assert lib.cpc_set_conv_tile(bm) == 0
    assert lib.cpc_set_mfma_mode(mode) == 0
    if dma is not None:                      # (cpc_set_dma_tile, cpc_set_dma_pipeline)
        assert lib.cpc_set_dma_tile(dma[0]) == 0 and lib.cpc_set_dma_pipeline(dma[1]) == 0
    assert lib.cpc_encoder_layout(B, L, sizes) == 0
    Ls = [sizes[3 + i] for i in range(5)]
    saved = torch.full((sizes[0],), float("nan"), device=dev)
    fscr = torch.empty(max(1, sizes[1]), device=dev)
    z = torch.full((B, Ls[4], 256), float("nan"), device=dev)
    wd = wave.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    parr = (ctypes.c_void_p * 20)(*[P(t) for t in plist])
    lib.check(lib.cpc_encoder_forward(P(wd), parr, P(saved), P(fscr), P(z), B, L, st), "encoder_forward")
    g = torch.Generator().manual_seed(11)
    dz = torch.randn(B, Ls[4], 256, generator=g)
    bscr = torch.empty(sizes[2], device=dev)
    grads = [torch.full_like(t, float("nan")) for t in plist]
    garr = (ctypes.c_void_p * 20)(*[P(t) for t in grads])
    dzd = dz.to(dev)
    lib.check(lib.cpc_encoder_backward(P(wd), parr, P(saved), P(z), P(dzd), P(bscr), garr, B, L, st),
              "encoder_backward")
    ys = []
    for i in range(4):                       # fp32 copies of y0..y3 whatever their storage (mode 3: y0, y1 as fp16 piece pairs)
        yi = torch.full((B, Ls[i], 256), float("nan"), device=dev)
        lib.check(lib.cpc_encoder_saved_activation(P(saved), i, P(yi), B, L, st), "encoder_saved_activation")
        ys.append(yi)
    torch.cuda.synchronize()
    lib.cpc_set_conv_tile(0)
    lib.cpc_set_h2_layers(0)
    lib.cpc_set_wgrad_dma_stages(_L.DEFAULT_WGRAD_DMA_STAGES)
    lib.cpc_set_conv_small_pipe(_L.DEFAULT_CONV_SMALL_PIPE)
    lib.cpc_set_dma_layer2(0)
    lib.cpc_set_dgrad_nsplit(_L.DEFAULT_DGRAD_NSPLIT)
    lib.cpc_set_fwd_nsplit(_L.DEFAULT_FWD_NSPLIT, -1)
    lib.cpc_set_mfma_mode(_lib_default_mode())
    if dma is not None:
        lib.cpc_set_dma_tile(0)
        lib.cpc_set_dma_pipeline(DMA_PIPELINE_DEFAULT)
    # oracle
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith("gEncoder")}
    acts = []
    # ReLU derivative of numerically tied pre-activations follows the device path (see oracle)
    ys = [t.cpu() for t in ys] + [z.cpu()]
    O.tie_report()
    # onednn=False (tests/test_gpu_shapes.py): torch 2.10's oneDNN conv1d backward returns wrong input gradients for the first
    # six steps of every sequence at some odd lengths (tests/test_emu_encoder._oracle_encoder); the fixtures' shapes are unaffected
    with torch.backends.mkldnn.flags(enabled=onednn):
        zr = O.encoder_forward(leaves, wave, collect=acts,
                               relu_override=[(y > 0).permute(0, 2, 1) for y in ys]).permute(0, 2, 1)
        ties = O.tie_report()                    # (how many elements took the device's ReLU derivative: oracle.tie_report)
        print(f"relu ties [B={B} L={L} mode={mode}]: {ties}")
        assert O.tie_ok(ties) and ties["disagree_outside"] == 0, ties
        (zr * dz).sum().backward()
    return dict(z=z.cpu(), z_ref=zr.detach(), grads=[g_.cpu() for g_ in grads],
                ref_grads=[leaves[n].grad for n in _names()], saved=saved, sizes=sizes, Ls=Ls, acts=acts, ys=ys)


@pytest.mark.parametrize("pipe,B", [(1, 8), (2, 8), (3, 8), (5, 8), (6, 8), (7, 8), (15, 64)])
def test_encoder_dma_pipelines_match_oracle(pipe, B):
    """Layer 1 on 256-row tiles of the DMA kernel at a size the oracle finishes quickly: the two-stage walk (1), the ping-pong slots (3) and the
    tap-pair walk (2: every input row brought to LDS once, used by both taps that read it); 7: four 128 x 128 waves, one per SIMD
    (dma_tile.h, the W128 loop); 15 = 7 + 8: layer 1's data gradient on that tile as well (256-row tiles: the benchmark's batch)."""
    dev = _dev()
    from cpc_audio_amd import _lib
    r = _run(_lib.get(), B, 20480, dev, mode=3, dma=(256, pipe))
    assert (r["z"] - r["z_ref"]).abs().max().item() < 1e-4
    for i in range(4):
        assert (r["ys"][i] - r["acts"][i].permute(0, 2, 1)).abs().max().item() < 1e-4, i
    for n, g, ref in zip(_names(), r["grads"], r["ref_grads"]):
        assert ((g.view_as(ref) - ref).norm() / (ref.norm() + 1e-30)).item() < 1e-4, n


@pytest.mark.parametrize("B,L,bm,mode", [(2, 20480, 0, 1), (3, 20480, 128, 1), (1, 4330, 64, 1), (8, 20480, 0, 1),
                                          (3, 20480, 0, 0), (2, 10240, 32, 0), (8, 20480, 0, 2), (3, 20480, 128, 2),
                                          (1, 4330, 64, 2), (2, 10240, 32, 2), (8, 20480, 0, 3), (3, 20480, 128, 3),
                                          (1, 4330, 64, 3), (2, 10240, 32, 3), (64, 20480, 0, 3), (8, 20480, 0, 34),
                                          (3, 20480, 128, 34), (1, 4330, 64, 34), (2, 10240, 32, 34), (64, 20480, 0, 34),
                                          (8, 20480, 0, 434), (1, 4330, 64, 434), (64, 20480, 0, 434), (8, 20480, 0, 334),
                                          (2, 10240, 64, 334), (64, 20480, 0, 334), (8, 20480, 0, 3), (8, 20480, 0, 534), (128, 20480, 0, 34), (8, 20480, 0, 634), (64, 20480, 0, 634),
                                          (8, 20480, 0, 734), (3, 4330, 0, 734), (64, 20480, 0, 734)])
def test_encoder_matches_oracle(B, L, bm, mode):
    """mode 1 = bf16 pipe with 3-piece split operands, mode 0 = exact-f32 MFMA, mode 2 = fp16 pipe with scaled
    2-piece split operands, mode 3 (default) = mode 2 + layers 1, 2 on the DMA kernel reading H2 activations (B = 64:
    256-row tiles on layer 1, 128-row tiles on layer 2 -- the benchmark's shapes); 34 = mode 3 with y0..y3 and dx1..dx4 all in H2
    storage: the short layers on the register-staged tiles fed H2 rows as they lie, every weight gradient on the DMA kernel."""
    dev = _dev()
    from cpc_audio_amd import _lib
    lib = _lib.get()
    r = _run(lib, B, L, dev, bm=bm, mode=mode)
    # encoder output within 1e-4 of the CPU reference path (north-star tolerance); expect ~1e-5
    err = (r["z"] - r["z_ref"]).abs().max().item()
    assert err < 1e-4, err
    for i in range(4):
        e = (r["ys"][i] - r["acts"][i].permute(0, 2, 1)).abs().max().item()
        assert e < 1e-4, (i, e)
    bad = {}
    for n, g, ref in zip(_names(), r["grads"], r["ref_grads"]):
        rel = ((g.view_as(ref) - ref).norm() / (ref.norm() + 1e-30)).item() if torch.isfinite(g).all() else float("inf")
        if not rel < 1e-4:
            bad[n] = rel
    assert not bad, bad
