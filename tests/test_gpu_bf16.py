"""The bf16-storage variant of BASELINE.json configs[1] (cpc_audio_amd.set_activation_storage("bf16"), cpc_set_mfma_mode(4)) on
a real MI355X: encoder activations y0..y3, the saved xhat1..4 and every encoder gradient tensor as bf16, conv weights rounded to
bf16 per step, one bf16 MFMA per product with fp32 accumulation, fp32 ChannelNorm statistics, fp32 encoder output.

This variant does NOT meet the fp32 path's 1e-4 bar and does not claim to: a bf16 value has 8 significant bits (relative
rounding error 2^-9 = 2e-3) and each of the five layers rounds its input once.  The bounds it is held to -- per layer, at the
benchmark's size B = 64 as well as at B <= 8, each about twice what is measured: the activations y0..y3 and z within
4e-3 / 6e-3 / 8e-3 / 9e-3 / 1e-2 relative (Frobenius) of the fp32 oracle's, max|z - z_ref| < 6e-2 on z = O(1); every encoder
parameter gradient within 4e-2 relative; the InfoNCE losses of a whole B = 64 train step within 2e-3 of the fp32 oracle, its
recurrence / head gradients within 2e-2 relative."""
import ctypes

import pytest
import torch

from oracle import cpc_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    return torch.device("cuda:0")


def _rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.fixture
def bf16_mode():
    import cpc_audio_amd
    cpc_audio_amd.set_activation_storage("bf16")
    yield
    cpc_audio_amd.set_activation_storage("fp32")


# (3, 978) / (2, 20494): window lengths whose layer-0 output is 4 L1 + 3 steps long -- the last step feeds no window of layer 1 and its
# gradient row is zero-filled (enc_conv.hip: zero_uncovered_rows, the bf16 branch); oracle with oneDNN off there (torch's oneDNN conv
# backward is wrong at some odd lengths, tests/test_gpu_shapes.py)
@pytest.mark.parametrize("B,L", [(8, 20480), (1, 4330), (64, 20480), (3, 978), (2, 20494)])
def test_bf16_storage_encoder_against_the_fp32_oracle(bf16_mode, B, L):
    dev = _dev()
    from cpc_audio_amd import _lib
    from cpc_audio_amd._lib import ptr as P
    lib = _lib.get()
    p = O.make_params(seed=0)
    names = [f"gEncoder.{n}{i}.{w}" for i in range(5)
             for n, w in (("conv", "weight"), ("conv", "bias"), ("batchNorm", "weight"), ("batchNorm", "bias"))]
    plist = [p[n].contiguous().to(dev) for n in names]
    wave = O.make_waveform(B, L, seed=5)
    sizes = (ctypes.c_long * 22)()
    assert lib.cpc_encoder_layout(B, L, sizes) == 0
    Ls = [sizes[3 + i] for i in range(5)]
    saved = torch.full((sizes[0],), float("nan"), device=dev)
    fscr = torch.empty(max(1, sizes[1]), device=dev)
    z = torch.full((B, Ls[4], 256), float("nan"), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    parr = (ctypes.c_void_p * 20)(*[P(t) for t in plist])
    wd = wave.to(dev)
    lib.check(lib.cpc_encoder_forward(P(wd), parr, P(saved), P(fscr), P(z), B, L, st), "encoder_forward")
    ys = []
    for i in range(4):
        yi = torch.full((B, Ls[i], 256), float("nan"), device=dev)
        lib.check(lib.cpc_encoder_saved_activation(P(saved), i, P(yi), B, L, st), "saved_activation")
        ys.append(yi.cpu())
    dz = torch.randn(B, Ls[4], 256, generator=torch.Generator().manual_seed(11))
    bscr = torch.empty(sizes[2], device=dev)
    grads = [torch.full_like(t, float("nan")) for t in plist]
    garr = (ctypes.c_void_p * 20)(*[P(t) for t in grads])
    dzd = dz.to(dev)
    lib.check(lib.cpc_encoder_backward(P(wd), parr, P(saved), P(z), P(dzd), P(bscr), garr, B, L, st), "encoder_backward")
    torch.cuda.synchronize()
    assert torch.isfinite(z).all() and all(torch.isfinite(g).all() for g in grads)
    # the fp32 oracle at the same size (B = 64 included: the variant's quoted configuration), layer by layer
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k.startswith("gEncoder")}
    acts = []
    with torch.backends.mkldnn.flags(enabled=(L % 160 == 0 or L == 4330)):
        zr = O.encoder_forward(leaves, wave, collect=acts, relu_override=[(y > 0).permute(0, 2, 1) for y in ys + [z.cpu()]],
                               tie_eps=0.08).permute(0, 2, 1)
        (zr * dz).sum().backward()
    layer_rel = [_rel(ys[i], acts[i].detach().permute(0, 2, 1)) for i in range(4)] + [_rel(z.cpu(), zr.detach())]
    err = (z.cpu() - zr.detach()).abs().max().item()
    grad_rel = {n: _rel(g.cpu().view_as(leaves[n].grad), leaves[n].grad) for n, g in zip(names, grads)}
    print(f"bf16 storage B={B} L={L}: y0..y3, z rel {[round(v, 5) for v in layer_rel]}  max|dz| {err:.4f}  "
          f"worst gradient {max(grad_rel.values()):.4f} ({max(grad_rel, key=grad_rel.get)})")
    assert err < 6e-2, err
    # measured 1.7e-3 / 2.8e-3 / 3.6e-3 / 4.4e-3 / 4.7e-3 at every size: one bf16 rounding (2^-9) more per layer; the bounds are ~2x
    # that, per layer -- a mis-scaled layer shows at ITS index
    for i, (v, bound) in enumerate(zip(layer_rel, (4e-3, 6e-3, 8e-3, 9e-3, 1e-2))):
        assert v < bound, (i, layer_rel)
    bad = {k: v for k, v in grad_rel.items() if not v < 4e-2}
    assert not bad, bad


def test_bf16_storage_train_step_at_the_quoted_size_close_to_the_fp32_oracle(bf16_mode):
    """One full bf16-storage train step at B = 64 x 20480 (BASELINE configs[1] as written) against the fp32 oracle: losses,
    accuracies and the gradients of everything behind the encoder (they see its bf16 error only through z)."""
    dev = _dev()
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    B = 64
    p = O.make_params(seed=16, head_scale=64.0)
    model, crit = build_model().to(dev), build_criterion().to(dev)
    load_flat_params(model, crit, p)
    tr = Trainer(model, crit)
    tr.optimizer.step = lambda *a, **k: None          # keep step 0's gradients and parameters
    tr.optimizer.zero_grad = lambda *a, **k: None
    wave = O.make_waveform(B, 20480, seed=13)
    g = torch.Generator().manual_seed(18)
    bi, si = O.draw_negative_indices(B, 128, 116, 128, generator=g)
    losses, acc = tr.step(wave.to(dev), None, negatives=(bi.to(dev), si.to(dev)))
    torch.cuda.synchronize()
    ora = O.train_step(p, wave, bi, si)
    assert torch.isfinite(losses).all()
    dl = (losses.cpu() - ora["losses"]).abs().max().item()
    da = (acc.cpu() - ora["acc"]).abs().max().item()
    named = dict(model.state_dict(keep_vars=True))
    named.update(crit.state_dict(keep_vars=True))
    rel = {k: _rel(named[k].grad.cpu(), ref) for k, ref in ora["grads"].items()}
    tail = {k: v for k, v in rel.items() if not k.startswith("gEncoder")}
    print(f"bf16 storage step B=64: max|dloss| {dl:.4f}  max|dacc| {da:.4f}  worst AR/head gradient {max(tail.values()):.4f}  "
          f"worst encoder gradient {max(v for k, v in rel.items() if k.startswith('gEncoder')):.4f}")
    assert dl < 2e-3 and da < 5e-3, (dl, da)                      # measured 1e-4 / 3e-4
    bad = {k: v for k, v in tail.items() if not v < 2e-2}         # measured 7e-3
    assert not bad, bad
    bad = {k: v for k, v in rel.items() if k.startswith("gEncoder") and not v < 8e-2}      # (oracle's own ReLU masks here: ties flip rows)
    assert not bad, bad


def test_bf16_storage_train_step_close_to_the_fp32_oracle(bf16_mode):
    dev = _dev()
    from cpc_audio_amd.train import Trainer, build_criterion, build_model, load_flat_params
    B = 4
    p = O.make_params(seed=6, head_scale=64.0)
    model, crit = build_model().to(dev), build_criterion().to(dev)
    load_flat_params(model, crit, p)
    tr = Trainer(model, crit)
    wave = O.make_waveform(B, 20480, seed=3)
    g = torch.Generator().manual_seed(8)
    bi, si = O.draw_negative_indices(B, 128, 116, 128, generator=g)
    losses, acc = tr.step(wave.to(dev), None, negatives=(bi.to(dev), si.to(dev)))
    ora = O.train_step(p, wave, bi, si)
    assert torch.isfinite(losses).all()
    assert (losses.cpu() - ora["losses"]).abs().max().item() < 5e-2
    # and the fp32 path on the same inputs still is what it was (the switch is process-wide: set back by the fixture)
    for _ in range(3):
        losses2, _ = tr.step(wave.to(dev), None, negatives=(bi.to(dev), si.to(dev)))
    assert torch.isfinite(losses2).all() and float(losses2.mean()) < float(losses.mean())      # it trains
