"""Import the reference's hot-path modules from /root/reference in THIS container.

TEST INFRASTRUCTURE ONLY (see oracle/cpc_oracle.py).  The reference never travels
to the GPU box; this helper is used by oracle/make_golden.py to pin the oracle
and by tests that are skipped when /root/reference is absent.

The reference imports three packages that are not installed here and that the
hot path never touches (torchaudio: cpc/model.py:7, soundfile: cpc/dataset.py:10,
progressbar: cpc/criterion/seq_alignment.py:5); they are replaced by empty stub
modules.
"""
import importlib.machinery
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("CPC_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "cpc", "model.py"))


def import_reference():
    """-> (cpc.model, cpc.criterion) modules of the reference."""
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    for name in ("torchaudio", "torchaudio.transforms", "soundfile", "progressbar"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            sys.modules[name] = m
    sys.modules["torchaudio"].transforms = sys.modules["torchaudio.transforms"]
    sys.modules["torchaudio.transforms"].MFCC = None
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import cpc.model as ref_model            # noqa: E402
    import cpc.criterion as ref_criterion    # noqa: E402
    return ref_model, ref_criterion


def build_reference(params, n_levels=2, n_predicts=12, n_neg=128, size_input_seq=128,
                    hidden=256):
    """North-star architecture (--arMode GRU --nLevelsGRU 2 --rnnMode linear,
    SURVEY.md T1) built exactly as cpc/train.py:307-321 does, loaded with ``params``
    (reference state-dict keys)."""
    ref_model, ref_criterion = import_reference()
    enc = ref_model.CPCEncoder(hidden, "layerNorm")
    ar = ref_model.CPCAR(hidden, hidden, False, n_levels, mode="GRU", reverse=False)
    model = ref_model.CPCModel(enc, ar)
    crit = ref_criterion.CPCUnsupersivedCriterion(
        n_predicts, hidden, hidden, n_neg, mode=None, rnnMode="linear", dropout=False,
        nSpeakers=0, speakerEmbedding=0, sizeInputSeq=size_input_seq)
    msd = {k: v for k, v in params.items() if not k.startswith("wPrediction")}
    csd = {k: v for k, v in params.items() if k.startswith("wPrediction")}
    model.load_state_dict(msd, strict=True)
    crit.load_state_dict(csd, strict=True)
    return model, crit
