"""Shape sweep on a real MI355X (round-5 review, "hardware shape sweep"): the encoder and the whole composite step at window
lengths the 20480-sample suite never reaches -- layer-0 outputs of 4 L1 + 3 steps (the last step feeds no window of layer 1: the
case commit 329b5f2 fixed with emulator evidence only), odd lengths, fewer than three frames' worth of samples, batch sizes off
every tile (1, 3, 5) -- and chunked feature extraction on a file whose ragged tail has that property.  Everything against the CPU
oracle; its conv backward runs with oneDNN OFF here (torch 2.10's oneDNN conv1d backward is wrong for the first six input steps
of every sequence at some of these lengths -- 1.6e-2 off torch's own float64 and native-fp32 results, tests/test_emu_encoder).
The reference accepts any length (cpc/model.py:99-105; cpc/feature_loader.py:253-254 feeds ragged tails)."""
import random

import pytest
import torch

from oracle import cpc_oracle as O
from test_emu_train_step import check_composite_step, frames
from test_gpu_encoder import _names, _run

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _l0(L):
    return (L + 6 - 10) // 5 + 1


def test_the_lengths_below_have_the_uncovered_step():
    """(documentation of the parameter lists: L0 = 4 L1 + 3 <=> L0 % 4 == 3)"""
    for L in (978, 20494, 1398, 12354):
        assert _l0(L) % 4 == 3, L


# (B, L, mode): 978 -> 195 -> 48 (the fixed defect), 1398 -> 279 -> 69, 20494 -> 4099 -> 1024 (benchmark scale); odd lengths;
# 405 samples = 2 frames; modes: 3 = default (H2 storage), 34 = every layer in H2 storage, 2 = register-staged fp16 pieces
@pytest.mark.parametrize("B,L,mode", [(1, 978, 3), (3, 978, 3), (5, 978, 34), (3, 1398, 3), (2, 20494, 3), (1, 20494, 34),
                                       (3, 2319, 3), (5, 1397, 3), (1, 405, 3), (3, 477, 34), (5, 4331, 2), (1, 290, 3)])
def test_encoder_at_ragged_window_lengths_matches_oracle(B, L, mode):
    dev = _dev()
    from cpc_audio_amd import _lib
    r = _run(_lib.get(), B, L, dev, mode=mode, onednn=False)
    assert torch.isfinite(r["z"]).all()
    assert (r["z"] - r["z_ref"]).abs().max().item() < 1e-4
    for i in range(4):
        assert (r["ys"][i] - r["acts"][i].permute(0, 2, 1)).abs().max().item() < 1e-4, i
    bad = {}
    for n, g, ref in zip(_names(), r["grads"], r["ref_grads"]):
        rel = ((g.view_as(ref) - ref).norm() / (ref.norm() + 1e-30)).item() if torch.isfinite(g).all() else float("inf")
        if not rel < 1e-4:
            bad[n] = rel
    assert not bad, bad            # (before 329b5f2: NaN in conv0's gradients at L0 = 4 L1 + 3)


# the composite step (cpc_train_step) -- encoder, recurrence, criterion, every gradient -- and, bit for bit, the stage-wise entry
# points: (B, L, K, N, carried state)
_FIXED = [(1, 978, 2, 7, False), (3, 978, 3, 16, True), (5, 978, 5, 33, False), (3, 1398, 4, 17, True), (2, 20494, 12, 128, False),
          (1, 405, 1, 1, False), (5, 1397, 4, 40, True), (3, 2319, 6, 128, False)]


def _random_shapes(n, seed):
    rng = random.Random(seed)
    out = []
    while len(out) < n:
        B = rng.choice([1, 2, 3, 5, 7])
        L = rng.randint(480, 6000)
        S = frames(L)
        K = rng.randint(1, min(S - 1, 12))
        N = rng.choice([1, 3, 16, 17, 40, 128])
        out.append((B, L, K, N, rng.random() < 0.5))
    return out


@pytest.mark.parametrize("B,L,K,N,use_h0", _FIXED + _random_shapes(12, seed=20261001))
def test_composite_step_at_ragged_shapes_matches_oracle_and_the_stagewise_step(B, L, K, N, use_h0):
    dev = _dev()
    from cpc_audio_amd import _lib
    with torch.cuda.device(dev):
        check_composite_step(_lib.get(), B, L, K, N, use_h0, seed=11, device=dev)


def test_chunked_feature_extraction_with_an_uncovered_tail_matches_oracle():
    """build_feature (cpc/feature_loader.py:228-269) on a file whose last chunk is 12354 samples long: 2471 layer-0 steps
    = 4 * 617 + 3, 77 frames -- and with B = 1 rows everywhere."""
    dev = _dev()
    from cpc_audio_amd import harness as H
    from cpc_audio_amd.criterion import CPCUnsupersivedCriterion
    from cpc_audio_amd.train import build_model, load_flat_params
    p = O.make_params(seed=7)
    model = build_model(keepHidden=True).to(dev)
    load_flat_params(model, CPCUnsupersivedCriterion(12, 256, 256, 128), p)
    n = 64000 + 12354
    assert _l0(12354) % 4 == 3
    seq = (0.1 * torch.randn(1, n, generator=torch.Generator().manual_seed(4))).clamp_(-1, 1)
    fm = H.FeatureModule(model, get_encoded=False).eval()
    feats = H.build_feature(fm, seq, strict=False, max_size_seq=64000)
    outs, h = [], None
    for start in range(0, n, 64000):
        sub = seq[:, start:start + 64000].reshape(1, 1, -1)
        z = O.encoder_forward(p, sub).permute(0, 2, 1)
        c, h = O.gru_forward(p, z, h0=h)
        outs.append(c)
    ref = torch.cat(outs, dim=1)
    assert feats.shape == ref.shape == (1, 400 + frames(12354), 256)
    assert torch.isfinite(feats).all() and (feats - ref).abs().max().item() < 1e-4
