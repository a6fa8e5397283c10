"""The C-ABI library builds for gfx950, loads, and exports every symbol include/cpc_hip.h declares
(no kernel is launched: this runs without a GPU)."""
import os
import re
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "cpc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cpc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_signature_table_agree():
    from cpc_audio_amd import _lib
    assert _declared() == sorted(_lib.SIGNATURES.keys())


def test_library_builds_loads_and_exports_every_declared_symbol():
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    from cpc_audio_amd import _lib, build
    path = build.build()
    bound = _lib.bind(path)          # raises if any declared symbol is missing
    assert bound.cpc_abi_version() == _lib.EXPECTED_ABI
    # argument validation happens before any launch, so it is testable without a GPU
    import ctypes
    sizes = (ctypes.c_long * 22)()
    assert bound.cpc_encoder_layout(0, 20480, sizes) == 1          # CPC_ERR_SHAPE
    assert bound.cpc_encoder_layout(2, 20480, sizes) == 0 and sizes[7] == 128
    assert bound.cpc_nce_layout(2, 128, 17, 128, sizes) == 1        # K > 16
    assert bound.cpc_nce_layout(2, 128, 12, 100, sizes) == 0        # (N % 16 != 0: candidate tiles padded and masked)
    assert bound.cpc_nce_padded_negatives(100) == 112 and bound.cpc_nce_padded_negatives(128) == 128
    assert bound.cpc_nce_layout(2, 128, 12, 0, sizes) == 1          # no negatives
    assert bound.cpc_set_conv_tile(48) == 2                         # CPC_ERR_ARG


def test_product_has_no_cpu_path():
    import torch
    from cpc_audio_amd.train import build_model
    model = build_model()
    with pytest.raises(RuntimeError, match="no CPU path"):
        model(torch.zeros(1, 1, 20480), torch.zeros(1, dtype=torch.long))


def test_module_api_matches_reference_surface():
    """Names, constructor signatures, attributes and state-dict keys of the drop-in modules
    (SURVEY.md section 8b)."""
    from cpc_audio_amd.criterion import CPCUnsupersivedCriterion
    from cpc_audio_amd.model import CPCAR, CPCEncoder, CPCModel
    from oracle import cpc_oracle as O
    enc = CPCEncoder(256, "layerNorm")
    ar = CPCAR(256, 256, False, 2, mode="GRU", reverse=False)
    m = CPCModel(enc, ar)
    crit = CPCUnsupersivedCriterion(12, 256, 256, 128, mode=None, rnnMode="linear", dropout=False,
                                    nSpeakers=0, speakerEmbedding=0, sizeInputSeq=128)
    keys = list(m.state_dict().keys()) + list(crit.state_dict().keys())
    shapes = O.param_shapes()
    assert keys == list(shapes.keys())
    for k, v in list(m.state_dict().items()) + list(crit.state_dict().items()):
        assert tuple(v.shape) == shapes[k], k
    assert enc.DOWNSAMPLING == 160 and enc.dimEncoded == 256 and enc.getDimOutput() == 256
    assert ar.getDimOutput() == 256 and ar.hidden is None and ar.keepHidden is False
    assert crit.warmUp() is False and crit.update() is None
    with pytest.raises(ValueError):
        CPCEncoder(256, "nope")
    with pytest.raises(ValueError):
        CPCUnsupersivedCriterion(12, 256, 256, 128, mode="bogus")


def test_prediction_dropout_option_has_the_reference_surface():
    """cpc/criterion/criterion.py:59,113-114: ``dropout=True`` holds nn.Dropout(p=0.5) (no parameters: the state dict is unchanged),
    applies it to each head's prediction in the reference-API forward while training, and is the identity in eval mode."""
    import torch
    from cpc_audio_amd.criterion import CPCUnsupersivedCriterion, PredictionNetwork
    plain = PredictionNetwork(12, 256, 256, rnnMode="linear", dropout=False)
    pn = PredictionNetwork(12, 256, 256, rnnMode="linear", dropout=True)
    assert plain.dropout is None and not plain.scores_apart
    assert isinstance(pn.dropout, torch.nn.Dropout) and pn.dropout.p == 0.5 and pn.scores_apart
    assert list(pn.state_dict().keys()) == list(plain.state_dict().keys())
    pn.load_state_dict(plain.state_dict())
    c = torch.randn(2, 5, 256)
    cand = [torch.randn(2, 3, 5, 256) for _ in range(12)]
    torch.manual_seed(0)
    dropped = pn(c, cand)
    pn.eval()
    assert not pn.scores_apart
    for a, b, d in zip(pn(c, cand), plain(c, cand), dropped):
        assert torch.equal(a, b) and a.shape == (2, 3, 5) and not torch.allclose(a, d)
    crit = CPCUnsupersivedCriterion(12, 256, 256, 128, dropout=True)
    assert isinstance(crit.wPrediction.dropout, torch.nn.Dropout)


@pytest.mark.parametrize("mode", ["RNN", "LSTM", "ffd", "conv4", "conv8", "conv12"])
def test_other_prediction_networks_reproduce_the_reference(mode):
    """cpc/criterion/criterion.py:63-81: the reference's other ``--rnnMode`` choices.  tests/golden/predictors.npz holds what the
    REFERENCE's PredictionNetwork returned (oracle/make_golden_predictors.py: seeded parameters under the reference's own
    state-dict keys, seeded context and candidates); this package's class must take the same state dict (strict) and give the
    same per-head scores and the same gradient w.r.t. the context -- including nn.RNN walking the batch axis (criterion.py:64-65)."""
    import json
    import os
    import numpy as np
    import torch
    from cpc_audio_amd.criterion import PredictionNetwork
    from oracle.make_golden_predictors import inputs, seeded_state
    gold = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(gold, "predictors_meta.json")))
    data = np.load(os.path.join(gold, "predictors.npz"))
    m = meta["modes"][mode]
    net = PredictionNetwork(meta["heads"], 256, 256, rnnMode=mode, dropout=False, sizeInputSeq=meta["window"])
    shapes = {k: tuple(v) for k, v in m["keys"].items()}
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == shapes          # names, order and shapes
    assert net.scores_apart
    net.load_state_dict(seeded_state(shapes, m["param_seed"]), strict=True)
    c, cand = inputs(m["input_seed"])
    cr = c.clone().requires_grad_(True)
    out = net(cr, cand)
    sum(o.sum() for o in out).backward()
    ref_out, ref_dc = torch.from_numpy(data[f"{mode}:out"]), torch.from_numpy(data[f"{mode}:dc"])
    assert (torch.stack(out).detach() - ref_out).abs().max().item() <= 2e-6 * max(1.0, ref_out.abs().max().item())
    assert ((cr.grad - ref_dc).norm() / ref_dc.norm()).item() <= 1e-5
    # the (B, W, K*256) tensor the HIP score kernels read is the same predictions, head k at columns k*256..
    pred = net.predictions(c)
    assert pred.shape == (c.shape[0], c.shape[1], meta["heads"] * 256)
    for k in range(meta["heads"]):
        s_k = (pred[:, :, k * 256:(k + 1) * 256].unsqueeze(1) * cand[k]).mean(dim=3)
        assert (s_k.detach() - ref_out[k]).abs().max().item() <= 2e-6 * max(1.0, ref_out.abs().max().item())


def test_library_contains_no_packed_fp32_arithmetic():
    """build.py's gate with an EMPTY allow-list (round 4): no code object of the library contains v_pk_{fma,mul,add}_f32 -- the
    instruction class behind the co-residency corruption of rounds 1-2."""
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    import glob
    from cpc_audio_amd import build
    build.build()
    assert build.PACKED_FP32_ALLOWED == ()
    objs = sorted(glob.glob(os.path.join(build.LIBDIR, "obj", "*.o")))
    assert len(objs) >= len(build.sources())
    found = {}
    for o in objs:
        found.update(build.packed_fp32_kernels(o))
    assert found == {}, found
