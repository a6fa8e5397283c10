"""Harness end to end on a real MI355X: epochs, logs, checkpoint + resume, validation, chunked inference."""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import cpc_oracle as O

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def test_run_epochs_checkpoint_and_resume(tmp_path):
    dev = _dev()
    from cpc_audio_amd import harness as H
    from cpc_audio_amd.train import build_criterion, build_model
    torch.manual_seed(0)
    model, crit = build_model().to(dev), build_criterion().to(dev)
    params = list(crit.parameters()) + list(model.parameters())
    opt = torch.optim.Adam(params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8)
    sched = H.build_scheduler(opt, scheduler_step=2, scheduler_ramp=2)
    prefix = os.path.join(str(tmp_path), "checkpoint")
    args = argparse.Namespace(hiddenEncoder=256, hiddenGar=256, arMode="GRU", nLevelsGRU=2, rnnMode="linear")
    tr = lambda: H.SyntheticLoader(3, 4, 20480, seed=1, device=dev)      # noqa: E731
    va = lambda: H.SyntheticLoader(2, 4, 20480, seed=2, device=dev)      # noqa: E731
    logs = H.run(tr, va, model, crit, 2, prefix, opt, sched, args=args, save_step=1, verbose=False)
    assert logs["epoch"] == [0, 1]
    for key in ("locLoss_train", "locAcc_train", "locLoss_val", "locAcc_val"):
        assert len(logs[key]) == 2 and len(logs[key][0]) == 12
        assert np.isfinite(np.array(logs[key])).all()
    assert 3.5 < np.mean(logs["locLoss_train"][0]) < 6.0            # ~ln(129) at init
    assert np.mean(logs["locLoss_train"][1]) < np.mean(logs["locLoss_train"][0])   # it learns the fixed batches
    for f in ("checkpoint_0.pt", "checkpoint_1.pt", "checkpoint_logs.json", "checkpoint_args.json"):
        assert os.path.exists(os.path.join(str(tmp_path), f)), f
    # resume: newest checkpoint + logs + args are found and training continues from epoch 2
    path, saved_logs, saved_args = H.get_checkpoint_data(str(tmp_path))
    assert path.endswith("checkpoint_1.pt") and saved_args.arMode == "GRU"
    model2, crit2 = build_model().to(dev), build_criterion().to(dev)
    opt2 = torch.optim.Adam(list(crit2.parameters()) + list(model2.parameters()), lr=2e-4)
    H.load_checkpoint(path, model2, crit2, opt2)
    for (k, v), (_, v2) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(v.cpu(), v2.cpu()), k
    logs2 = H.run(tr, va, model2, crit2, 3, prefix, opt2, None, logs=saved_logs, args=args, save_step=1, verbose=False)
    assert logs2["epoch"] == [0, 1, 2] and os.path.exists(os.path.join(str(tmp_path), "checkpoint_2.pt"))


def test_chunked_feature_extraction_matches_oracle():
    """build_feature (cpc/feature_loader.py:228-269): 64000-sample chunks, GRU state carried across chunks."""
    dev = _dev()
    from cpc_audio_amd import harness as H
    from cpc_audio_amd.train import build_model, load_flat_params
    from cpc_audio_amd.criterion import CPCUnsupersivedCriterion
    p = O.make_params(seed=7)
    model = build_model(keepHidden=True).to(dev)
    load_flat_params(model, CPCUnsupersivedCriterion(12, 256, 256, 128), p)
    n = 64000 * 2 + 12345
    g = torch.Generator().manual_seed(3)
    seq = (0.1 * torch.randn(1, n, generator=g)).clamp_(-1, 1)
    fm = H.FeatureModule(model, get_encoded=False).eval()
    feats = H.build_feature(fm, seq, strict=False, max_size_seq=64000)
    # oracle: same chunking with carried hidden state
    outs, h = [], None
    for start in range(0, n, 64000):
        sub = seq[:, start:start + 64000].reshape(1, 1, -1)
        z = O.encoder_forward(p, sub).permute(0, 2, 1)
        c, h = O.gru_forward(p, z, h0=h)
        outs.append(c)
    ref = torch.cat(outs, dim=1)
    assert feats.shape == ref.shape == (1, 400 + 400 + 77, 256)
    assert (feats - ref).abs().max().item() < 1e-4
    # encoder output path + strict tail handling
    model.gAR.hidden = None
    fz = H.FeatureModule(model, get_encoded=True).eval()
    zf = H.build_feature(fz, seq, strict=True, max_size_seq=64000)
    assert zf.shape == (1, 400 + 400 + 77, 256)


def test_device_resident_dataset_feeds_the_train_loop(tmp_path):
    """AudioBatchData.to(cuda): the packed waveform lives in HBM and batches are gathered there (dataset.py);
    two epochs over a small synthetic corpus through the harness's train loop."""
    dev = _dev()
    import wave
    from cpc_audio_amd import harness as H
    from cpc_audio_amd.dataset import AudioBatchData, findAllSeqs
    from cpc_audio_amd.train import build_criterion, build_model
    rng = np.random.default_rng(0)
    for spk in range(3):
        for utt in range(2):
            d = tmp_path / "db" / f"s{spk}" / "c0"
            d.mkdir(parents=True, exist_ok=True)
            pcm = (rng.standard_normal(20480 * 3 + 100 * utt) * 3000).astype("<i2")
            with wave.open(str(d / f"s{spk}-c0-{utt}.wav"), "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(pcm.tobytes())
    seqs, speakers = findAllSeqs(str(tmp_path / "db"), extension=".wav")
    data = AudioBatchData(tmp_path / "db", 20480, seqs, None, len(speakers)).to(dev)
    assert data.data.is_cuda
    loader = data.getDataLoader(4, "uniform", True)
    batch, labels = next(iter(loader))
    assert batch.is_cuda and batch.shape == (4, 1, 20480) and labels.is_cuda
    torch.manual_seed(0)
    model, crit = build_model().to(dev), build_criterion().to(dev)
    opt = torch.optim.Adam(list(crit.parameters()) + list(model.parameters()), lr=2e-4)
    logs = [H.train_epoch(data.getDataLoader(4, "samespeaker", True), model, crit, opt) for _ in range(2)]
    for lg in logs:
        assert np.isfinite(np.array(lg["locLoss_train"])).all() and len(lg["locLoss_train"]) == 12


@pytest.mark.parametrize("kind", ["uniform", "sequential", "samespeaker", "samesequence"])
def test_device_side_window_plans_against_the_per_item_path(tmp_path, kind):
    """The four sampling types of cpc/dataset.py:318-408 with the pack resident in HBM: the plan (all window starts of the
    pass) is built on the device, every batch is one device-side gather -- and equals, item by item, what the reference's
    per-item path (__getitem__ on a CPU copy of the same pack) returns for the same indices.  'sequential' additionally
    feeds the hidden-state carry it exists for (keepHidden): item b of consecutive batches is contiguous audio."""
    dev = _dev()
    import wave
    from cpc_audio_amd.dataset import AudioBatchData, findAllSeqs
    from cpc_audio_amd.train import Trainer, build_criterion, build_model
    W = 20480
    rng = np.random.default_rng(1)
    for spk in range(3):
        for utt in range(3):
            d = tmp_path / "db" / f"s{spk}" / "c0"
            d.mkdir(parents=True, exist_ok=True)
            pcm = (rng.standard_normal(W * (3 + utt) + 321 * spk) * 3000).astype("<i2")
            with wave.open(str(d / f"s{spk}-c0-{utt}.wav"), "wb") as f:
                f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(pcm.tobytes())
    seqs, speakers = findAllSeqs(str(tmp_path / "db"), extension=".wav")
    random_state = __import__("random").getstate()
    __import__("random").seed(4)
    data = AudioBatchData(tmp_path / "db", W, seqs, None, len(speakers)).to(dev)
    __import__("random").setstate(random_state)
    host = {"data": data.data.cpu(), "bounds": list(data.speakerLabel), "seq": list(data.seqLabel)}
    B = 4
    plan = data.window_plan(kind, B, 777)
    assert plan.starts.is_cuda and len(plan) > 0
    served = []
    for index in plan:
        assert index.is_cuda
        batch, labels = data.get_batch(index)
        assert batch.is_cuda and batch.shape[1:] == (1, W)
        for row, sidx in enumerate(index.tolist()):
            assert torch.equal(batch[row, 0].cpu(), host["data"][sidx:sidx + W])
            spk = next(i for i, b in enumerate(host["bounds"]) if b > sidx) - 1          # cpc/dataset.py:181-183
            assert int(labels[row]) == spk
        served.append(index.tolist())
    flat = [x for b in served for x in b]
    assert len(flat) == len(set(flat))                                                   # no window twice in a pass
    if kind == "uniform":
        assert all(len(b) == B for b in served)
    if kind in ("samespeaker", "samesequence"):
        bounds = host["bounds"] if kind == "samespeaker" else host["seq"]
        for b in served:
            assert len({next(i for i, e in enumerate(bounds) if e > x) for x in b}) == 1
    if kind == "sequential":
        for prev, cur in zip(served, served[1:]):
            assert [c - p for p, c in zip(prev, cur)] == [W] * B
        # ... which is what lets the GRU state of batch i seed batch i+1 (feature_loader.py:149, model.py:193-198)
        torch.manual_seed(0)
        model, crit = build_model(keepHidden=True).to(dev), build_criterion().to(dev)
        tr = Trainer(model, crit)
        n = 0
        for batch, labels in data.getDataLoader(B, "sequential", False):
            losses, _ = tr.step(batch, labels)
            assert torch.isfinite(losses).all() and model.gAR.hidden is not None
            n += 1
        assert n == len(data.window_plan("sequential", B, 0)) >= 2


def test_train_epoch_through_the_composite_step_equals_the_autograd_loop():
    """harness.train_epoch issues forward + backward through cpc_train_step where the configuration allows (train.CompositeStep);
    the logged losses / accuracies and the parameters after an epoch must equal the autograd-driven loop's bit for bit, with the
    package's Adam and with torch's own (the gradients are views of a flat buffer either optimiser reads)."""
    dev = _dev()
    from cpc_audio_amd import harness as H
    from cpc_audio_amd.optim import Adam
    from cpc_audio_amd.train import build_criterion, build_model, load_flat_params
    p = O.make_params(seed=41, head_scale=64.0)
    for make_opt in (lambda ps: Adam(ps, lr=2e-4), lambda ps: torch.optim.Adam(ps, lr=2e-4)):
        res = []
        for composite in (False, True):
            model, crit = build_model().to(dev), build_criterion().to(dev)
            load_flat_params(model, crit, p)
            opt = make_opt(list(crit.parameters()) + list(model.parameters()))
            H.COMPOSITE_STEP = composite
            try:
                torch.manual_seed(77)
                logs = H.train_epoch(H.SyntheticLoader(3, 4, 20480, seed=5, device=dev), model, crit, opt)
            finally:
                H.COMPOSITE_STEP = True
            torch.cuda.synchronize()
            state = {k: v.detach().cpu().clone() for k, v in list(model.state_dict().items()) + list(crit.state_dict().items())}
            res.append((logs, state))
        assert res[0][0]["iter"] == res[1][0]["iter"] == 3
        for k in ("locLoss_train", "locAcc_train"):
            assert (res[0][0][k] == res[1][0][k]).all(), k
        for k in res[0][1]:
            assert torch.equal(res[0][1][k], res[1][1][k]), k


def test_graft_entry_smoke_runs():
    """__graft_entry__.smoke(): the driver's one-call check (a small train step on cuda:0 against the oracle, tie accounting
    included) -- run here too, so that an interface change that breaks it fails the suite and not the round-end run."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import importlib
    entry = importlib.import_module("__graft_entry__")
    entry.smoke()


@pytest.mark.parametrize("kind", ["uniform", "sequential", "samespeaker", "samesequence"])
@pytest.mark.parametrize("offset_on", [False, True])
def test_device_window_plans_against_the_reference_fixture(tmp_path, kind, offset_on):
    """The device-resident data path against what the REFERENCE's AudioBatchData / samplers / loaders produced on the same
    corpus (tests/golden/dataset.json, oracle/make_golden_dataset.py; cpc/dataset.py:20-258,318-408): boundaries, the plan of a
    pass at the reference's own random offset -- bit-equal for 'sequential', the same window grid / whole batches for 'uniform'
    (another generator), per interval the same windows and batch sizes for the grouped types (order random in both) -- and
    every served batch's samples and speaker labels."""
    dev = _dev()
    from test_dataset_golden import GOLD, W, _load, check_grouped, interval_of, write_corpus
    write_corpus(tmp_path)
    g = GOLD["one_pack"]
    ds = _load(tmp_path).to(dev)
    assert ds.data.is_cuda and ds.data.numel() == g["n_samples"]
    assert [int(x) for x in ds.speakerLabel] == g["speakerLabel"] and [int(x) for x in ds.seqLabel] == g["seqLabel"]
    assert float(ds.data.double().sum()) == g["sum"]
    ref = [b["starts"] for b in g["loaders"][f"{kind}/{int(offset_on)}"]["batches"]]
    bounds = g["speakerLabel"] if kind != "samesequence" else g["seqLabel"]
    if kind in ("uniform", "sequential"):
        offset = min(x for b in ref for x in b)                                        # the first window of the pack starts at it
    else:
        offset = min(x - bounds[interval_of(bounds, x)] for b in ref for x in b)
    assert (offset > 0) == offset_on
    plan = ds.window_plan(kind, 3, offset)
    assert plan.starts.is_cuda
    served = plan.batches()
    if kind == "sequential":
        assert served == ref
    elif kind == "uniform":
        flat = [x for b in served for x in b]
        n_win = g["n_samples"] // W - (1 if offset > 0 else 0)
        assert len(served) == len(ref) and all(len(b) == 3 for b in served) and len(set(flat)) == len(flat)
        assert set(flat) <= {offset + W * i for i in range(n_win)} and {x for b in ref for x in b} <= {offset + W * i for i in range(n_win)}
    else:
        check_grouped(served, ref, bounds, 3)
    host = ds.data.cpu()
    for index in plan:
        batch, labels = ds.get_batch(index)
        assert batch.is_cuda and labels.is_cuda
        for row, s in enumerate(index.tolist()):
            assert torch.equal(batch[row, 0].cpu(), host[s:s + W])
            assert int(labels[row]) == interval_of(g["speakerLabel"], s)               # cpc/dataset.py:181-183 on the reference's bounds


def test_chunked_feature_extraction_on_the_device_against_the_reference_fixture():
    """build_feature with the waveform and the recording feature maker on the GPU: the chunks cpc/feature_loader.py:228-269 cut
    and the features it returned (tests/golden/harness.json), incl. strict tails that keep 0 frames' worth and files shorter
    than a chunk."""
    dev = _dev()
    from test_harness_golden import GOLD as HG, check_chunks
    for case in HG["chunks"]:
        check_chunks(case, device=dev)
