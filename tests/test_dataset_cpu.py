"""Data path (cpc_audio_amd/dataset.py) against the behaviour the reference pins in cpc/unit_tests.py:15-200
(TestDataLoader, TestPhonemParser), on a synthetic tree of .wav files with the same speaker / chapter / file layout
(the reference's fixtures are .flac, which this image cannot decode)."""
import os
import random
import wave

import numpy as np
import pytest
import torch

from cpc_audio_amd.dataset import (AudioBatchData, SameSpeakerSampler, SequentialSampler, UniformAudioSampler,
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
