"""Data path (cpc_audio_amd/dataset.py) against the behaviour the reference pins in cpc/unit_tests.py:15-200
(TestDataLoader, TestPhonemParser), on a synthetic tree of .wav files with the same speaker / chapter / file layout
(the reference's fixtures are .flac, which this image cannot decode)."""
import os
import random
import wave

import numpy as np
import pytest
import torch

from cpc_audio_amd.dataset import (AudioBatchData, SameSpeakerSampler, SequentialSampler, UniformAudioSampler, WindowPlan,
                                   filterSeqs, findAllSeqs, parseSeqLabels)

ALL = ["2911/12359/2911-12359-0007", "4051/11218/4051-11218-0044", "4397/15668/4397-15668-0003",
       "4397/15668/4397-15668-0007", "5393/19218/5393-19218-0024", "5678/43301/5678-43301-0021",
       "5678/43303/5678-43303-0024", "5678/43303/5678-43303-0032", "6476/57446/6476-57446-0019"]
LISTED = ALL[2:]                      # the 7 sequences of the reference's seq_list.txt
W = 20480


@pytest.fixture(scope="module")
def db(tmp_path_factory):
    root = tmp_path_factory.mktemp("db")
    rng = np.random.default_rng(0)
    lengths = {}
    for i, rel in enumerate(ALL):
        p = root / "test_db" / (rel + ".wav")
        p.parent.mkdir(parents=True, exist_ok=True)
        n = W * (3 + (i % 4)) + 777 * i                   # 3..6 windows each, ragged
        pcm = (rng.standard_normal(n) * 3000).astype("<i2")
        with wave.open(str(p), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(pcm.tobytes())
        lengths[rel] = n
    with open(root / "seq_list.txt", "w") as f:
        f.write("\n".join(os.path.basename(s) for s in reversed(LISTED)) + "\n")
    with open(root / "phone_labels.txt", "w") as f:
        for rel in ALL[:2]:
            n = lengths[rel] // 160
            f.write(os.path.basename(rel) + " " + " ".join(str((j // 7) % 41) for j in range(n)) + "\n")
    return root, lengths


def test_find_all_seqs(db):
    root, _ = db
    seq_names, speakers = findAllSeqs(str(root / "test_db"), extension=".wav")
    assert len(speakers) == 6 and set(speakers) == {"2911", "4051", "4397", "5393", "5678", "6476"}
    assert {x[1] for x in seq_names} == {s + ".wav" for s in ALL} and len(seq_names) == 9
    for spk, name in seq_names:
        assert speakers[spk] == os.path.basename(name).split("-")[0]
    # cache round trip
    again, spk2 = findAllSeqs(str(root / "test_db"), extension=".wav", loadCache=True)
    assert again == seq_names and spk2 == speakers
    # speaker_level = 2 / 0 (unit_tests.py:70-105)
    seq2, speakers2 = findAllSeqs(str(root / "test_db"), extension=".wav", speaker_level=2)
    assert set(speakers2) == {"2911/12359", "4051/11218", "4397/15668", "5393/19218", "5678/43301", "5678/43303",
                              "6476/57446"}
    assert findAllSeqs(str(root / "test_db" / "2911" / "12359"), extension=".wav")[1] == [""]
    assert findAllSeqs(str(root / "test_db"), extension=".wav", speaker_level=0)[1] == [""]


def test_load_data_counts(db):
    root, _ = db
    seq_names, speakers = findAllSeqs(str(root / "test_db"), extension=".wav")
    seq_names = filterSeqs(root / "seq_list.txt", seq_names)
    assert {x[1] for x in seq_names} == {s + ".wav" for s in LISTED} and len(seq_names) == 7
    data = AudioBatchData(root / "test_db", W, seq_names, None, 9)
    assert data.getNSpeakers() == 9 and data.getNSeqs() == 7 and data.getNPacks() == 1
    assert data.speakerLabel[0] == 0 and data.speakerLabel[-1] == len(data.data) == data.seqLabel[-1]


@pytest.mark.parametrize("max_size,packs", [(4000000000, 1), (300000, 2)])
def test_samespeaker_loader(db, max_size, packs):
    """unit_tests.py:130-170: every batch of the 'samespeaker' loader carries ONE label; 4 speakers are visited."""
    root, _ = db
    random.seed(1)
    seq_names, speakers = findAllSeqs(str(root / "test_db"), extension=".wav")
    seq_names = filterSeqs(root / "seq_list.txt", seq_names)
    data = AudioBatchData(root / "test_db", W, seq_names, None, len(speakers), MAX_SIZE_LOADED=max_size)
    assert data.getNPacks() >= packs
    loader = data.getDataLoader(2, "samespeaker", True, numWorkers=2)
    visited = set()
    n = 0
    for batch, labels in loader:
        assert batch.shape[1:] == (1, W) and batch.dtype == torch.float32
        assert int((labels == labels[0]).sum()) == labels.numel()
        visited.add(int(labels[0]))
        n += 1
    assert len(visited) == 4 and n > 0


def test_get_batch_equals_getitem(db):
    root, _ = db
    seq_names, speakers = findAllSeqs(str(root / "test_db"), extension=".wav")
    data = AudioBatchData(root / "test_db", W, seq_names, None, len(speakers))
    starts = [0, 5000, len(data.data) - W - 2, data.speakerLabel[2], data.speakerLabel[3] - 1]
    batch, labels = data.get_batch(starts)
    for row, s in enumerate(starts):
        x, lab = data[s]
        assert torch.equal(batch[row], x) and int(labels[row]) == int(lab)


def test_samplers():
    torch.manual_seed(0)
    u = UniformAudioSampler(10 * W + 5, W, 100)
    idx = list(u)
    assert len(idx) == len(u) == 9 and sorted(idx) == [100 + W * i for i in range(9)]
    s = SequentialSampler(40 * W, W, 0, 4)
    batches = list(s)
    assert len(batches) == 10 and batches[0] == [0, 10 * W, 20 * W, 30 * W] and batches[3][1] == 10 * W + 3 * W
    ss = SameSpeakerSampler(3, [0, 5 * W, 5 * W, 12 * W + 7], W, 0)        # an empty interval in the middle
    assert len(ss) == 2 + 3
    for b in ss:
        assert all(x < 5 * W for x in b) or all(x >= 5 * W for x in b)
    with pytest.raises(AttributeError):
        SameSpeakerSampler(3, [W, 2 * W], W, 0)


def test_phone_labels(db):
    """unit_tests.py:173-200: '<seq> labels...' parsing and per-window phone labels (sizeWindow 640 = 4 labels)."""
    root, lengths = db
    phones, n_phones = parseSeqLabels(root / "phone_labels.txt")
    assert phones["step"] == 160 and len(phones) == 3 and n_phones == 41
    seq_names = [(0, ALL[0] + ".wav"), (1, ALL[1] + ".wav")]
    data = AudioBatchData(root / "test_db", 640, seq_names, phones, 2)
    first = os.path.basename(ALL[0])
    assert data.getPhonem(0) == phones[first][:4]
    assert data.getPhonem(160 * 9 + 3) == phones[first][9:13]
    batch, labels = data.get_batch([0, 160 * 9 + 3])
    assert labels.tolist() == [phones[first][:4], phones[first][9:13]] and batch.shape == (2, 1, 640)
    data.doubleLabels = True
    _, spk, ph = data.get_batch([0, len(data.data) - 700])
    assert spk.tolist() == [0, 1] and ph.shape == (2, 4)


@pytest.mark.parametrize("offset", [0, 1234])
def test_window_plans_serve_every_window_once_and_respect_their_grouping(offset):
    """The four sampling types of cpc/dataset.py:318-408 as tensor-built plans (WindowPlan): what each type promises."""
    g = torch.Generator().manual_seed(3)
    dev = torch.device("cpu")
    n, B = 37 * W + 99, 4
    cut = 1 if offset else 0
    # uniform: every window of the pack once, whole batches only
    u = WindowPlan.uniform(n, W, B, offset, dev, g)
    flat = [x for b in u.batches() for x in b]
    assert len(u) == (37 - cut) // B and all(len(b) == B for b in u.batches())
    assert len(set(flat)) == len(flat) and set(flat) <= {offset + W * i for i in range(37 - cut)}
    # sequential: item b of consecutive batches walks contiguous audio inside its own 1/B of the pack
    q = WindowPlan.sequential(n, W, B, offset, dev).batches()
    assert len(q) == 37 // B - cut
    for i, b in enumerate(q):
        assert b == [offset + W * i + lane * (n // B) for lane in range(B)]
    assert all(x + W <= n for x in q[-1])
    # grouped: one interval per batch, every window of every interval once, short last batches kept
    bounds = [0, 5 * W + 17, 5 * W + 17, 6 * W, 19 * W + 3, n]   # an empty interval and one shorter than a window (with offset)
    p = WindowPlan.grouped(torch.tensor(bounds), W, B, offset, dev, g)
    seen = []
    for b in p.batches():
        which = {max(i for i in range(len(bounds) - 1) if bounds[i] <= x) for x in b}
        assert len(which) == 1 and 1 <= len(b) <= B
        i = which.pop()
        assert all((x - offset - bounds[i]) % W == 0 and x + W <= bounds[i + 1] + (W if offset else 0) for x in b)
        seen += b
    expect = sum(max(0, (bounds[i + 1] - bounds[i]) // W - cut) for i in range(len(bounds) - 1))
    assert len(seen) == len(set(seen)) == expect
    assert len(p) == sum(-(-max(0, (bounds[i + 1] - bounds[i]) // W - cut) // B) for i in range(len(bounds) - 1))
    # two plans from one generator state differ (the order is random), the same seed reproduces
    a1 = WindowPlan.grouped(torch.tensor(bounds), W, B, offset, dev, torch.Generator().manual_seed(5)).batches()
    a2 = WindowPlan.grouped(torch.tensor(bounds), W, B, offset, dev, torch.Generator().manual_seed(5)).batches()
    assert a1 == a2 and a1 != p.batches()


@pytest.mark.parametrize("kind", ["uniform", "sequential", "samespeaker", "samesequence"])
def test_loader_batches_equal_per_item_getitem(db, kind):
    """Every batch the loader serves is what __getitem__ (the reference's per-item path, cpc/dataset.py:185-202) returns
    for the plan's indices, stacked."""
    root, _ = db
    random.seed(2)
    torch.manual_seed(2)
    seq_names, speakers = findAllSeqs(str(root / "test_db"), extension=".wav")
    data = AudioBatchData(root / "test_db", W, seq_names, None, len(speakers))
    plan = data.getBaseSampler(kind, 3, 500)
    assert len(plan) > 0
    for index in plan:
        batch, labels = data.get_batch(index)
        for row, sidx in enumerate(index.tolist()):
            x, lab = data[sidx]
            assert torch.equal(batch[row], x) and int(labels[row]) == int(lab)
        if kind == "samespeaker":
            assert len(set(labels.tolist())) == 1
    n = sum(1 for _ in data.getDataLoader(3, kind, True))
    assert n > 0


def test_pack_boundaries_and_pack_cutting_follow_the_reference_accounting():
    """The two helpers behind AudioBatchData: speaker boundaries exist for every speaker index up to the last one present
    (absent speakers are empty intervals), an out-of-range speaker raises, labelled sequences are cut where their labels
    end; the pack cutter opens a new pack with the sequence that overflowed and counts it with the pack it overflowed
    (cpc/dataset.py:104-118, 142-170)."""
    from cpc_audio_amd.dataset import _Pack, _cut_into_packs
    loaded = [(3, "b", torch.arange(5.)), (0, "z", torch.arange(4.)), (3, "a", torch.arange(7.))]
    pk = _Pack(loaded, 5, None, 0)
    assert pk.seq_bounds.tolist() == [0, 4, 11, 16]                     # (0,z), (3,a), (3,b)
    assert pk.speaker_bounds.tolist() == [0, 4, 4, 4, 16]               # speakers 1, 2 are empty, speaker 4 is not listed
    assert torch.equal(pk.wave, torch.cat([torch.arange(4.), torch.arange(7.), torch.arange(5.)]))
    with pytest.raises(ValueError):
        _Pack(loaded, 3, None, 0)
    labelled = _Pack([(0, "u", torch.arange(10.))], 1, {"step": 4, "u": [7, 8]}, 4)
    assert labelled.wave.numel() == 8 and labelled.phones == [7, 8]
    # [5, 5 | 5, 5 |]: the second and fourth sequences overflow; each opens the next pack with a running size of zero, so a
    # pack that would consist of the overflowing LAST sequence alone is never closed (the reference drops it the same way)
    packs, counted = _cut_into_packs([5, 5, 5, 5], 9)
    assert packs == [(0, 1), (1, 3)] and counted == 10 + 10
    packs, counted = _cut_into_packs([5, 5, 5, 5, 2], 9)
    assert packs == [(0, 1), (1, 3), (3, 5)] and counted == 10 + 10 + 2
    assert _cut_into_packs([3, 3], 100) == ([(0, 2)], 6)
