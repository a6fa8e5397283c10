"""Data path with the semantics of the reference's cpc/dataset.py.

Same public names and behaviour -- ``findAllSeqs``, ``filterSeqs``, ``parseSeqLabels``, ``AudioBatchData`` (packed
in-RAM waveform, speaker / sequence boundaries, pack-by-pack loading with the next pack prefetched in the
background), the window samplers (``uniform`` / ``samespeaker`` / ``samesequence`` / ``sequential``, random offset of
up to sizeWindow/2) and ``AudioLoader`` -- so cpc/train.py's ``loadArgs`` / ``getDataLoader`` calls work unchanged.

What is different, on purpose (MI355X first):
  * a pack (default 4e9 samples = 16 GB) fits the GPU's 288 GB of HBM many times over, so ``AudioBatchData.to(device)``
    keeps the packed waveform resident on the GPU and ``AudioLoader`` assembles every batch there with ONE gather
    (window indices + arange) and a ``bucketize`` for the labels: no per-item ``__getitem__``, no worker processes,
    no H2D copy per step.  The CPU path (``numWorkers``-free, vectorised gather into pinned memory) is kept for hosts
    without the memory budget.
  * files are decoded by a pluggable reader (``register_reader``): 16/32-bit PCM and float ``.wav`` through the
    standard library, ``.npy`` / ``.pt`` tensors, and anything ``soundfile`` reads (e.g. LibriSpeech ``.flac``) when
    that package is installed -- it is not in this image, which is also why the reference's own flac fixtures
    cannot be decoded here (tests build an equivalent tree of .wav files).
  * the background loader is a thread pool (decoding releases the GIL in numpy / soundfile), not a process pool.

Reference: cpc/dataset.py:20-258 (AudioBatchData), :261-316 (AudioLoader), :318-408 (samplers), :411-520 (helpers).
"""
import os
import random
import wave
from concurrent.futures import ThreadPoolExecutor
from copy import deepcopy
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

# ----------------------------------------------------------------------------------------------- file readers
_READERS = {}


def register_reader(extension, read_fn, length_fn=None):
    """read_fn(path) -> 1-D float32 numpy array or tensor (mono); length_fn(path) -> number of samples."""
    _READERS[extension.lower()] = (read_fn, length_fn)


def _wav_read(path):
    with wave.open(str(path), "rb") as f:
        nch, width, n = f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported PCM sample width {width}")
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)
    return x


def _wav_len(path):
    with wave.open(str(path), "rb") as f:
        return f.getnframes()


def _npy_read(path):
    x = np.load(str(path)).astype(np.float32)
    return x.mean(axis=1) if x.ndim == 2 else x


def _npy_len(path):
    return int(np.load(str(path), mmap_mode="r").shape[0])


def _pt_read(path):
    x = torch.load(str(path)).float()
    return x.mean(dim=1) if x.dim() == 2 else x


register_reader(".wav", _wav_read, _wav_len)
register_reader(".npy", _npy_read, _npy_len)
register_reader(".pt", _pt_read, lambda p: int(_pt_read(p).shape[0]))


def _reader_for(path):
    ext = Path(path).suffix.lower()
    if ext in _READERS:
        return _READERS[ext]
    try:                                              # cpc/dataset.py:249-258 reads everything with soundfile
        import soundfile as sf
    except ImportError as e:
        raise RuntimeError(f"no reader for '{ext}' files: install soundfile or call "
                           f"cpc_audio_amd.dataset.register_reader('{ext}', fn)") from e

    def read(p):
        x = sf.read(str(p), dtype="float32")[0]
        return x.mean(axis=1) if x.ndim == 2 else x
    return read, lambda p: int(sf.info(str(p)).frames)


def loadFile(data):
    """(speaker, path) -> (speaker, sequence name, mono float waveform); what the reference's loader of the same name
    returns (cpc/dataset.py:249-258), read through the pluggable readers."""
    spk, where = data
    wav = _reader_for(where)[0](where)
    if not torch.is_tensor(wav):
        wav = torch.from_numpy(np.ascontiguousarray(wav))
    return spk, Path(where).stem, wav.float()


def extractLength(couple):
    """Number of samples of a (speaker, path) entry without decoding it where the reader can tell (cpc/dataset.py:411-414)."""
    where = couple[1]
    read, length = _reader_for(where)
    return int(read(where).shape[0]) if length is None else length(where)


# ----------------------------------------------------------------------------------------------- the dataset
def _cut_into_packs(lengths, cap):
    """Greedy cut of a sequence list into packs of at most ~cap samples -> ([(first, last)], samples counted).  The sequence
    that overflows a pack opens the next one, and its samples are counted with the pack it overflowed -- the reference's
    accounting (cpc/dataset.py:104-118), which len(dataset) and the loaders' progress estimates are built on."""
    packs, counted, first, running = [], 0, 0, 0
    for i, n in enumerate(lengths):
        running += n
        if running > cap:
            packs.append((first, i))
            counted += running
            first, running = i, 0
    if running:
        packs.append((first, len(lengths)))
        counted += running
    return packs, counted


class _Pack:
    """One pack resident in memory: the waveforms of its sequences back to back, ordered by (speaker, name), with the
    boundaries the samplers and label look-ups need.  Boundaries are computed with tensor ops from the per-sequence sizes:
      seq_bounds      (n_seq + 1,)   sample offset of every sequence
      speaker_bounds  (last speaker + 2,)   sample offset of every speaker index up to the last one present
    (cpc/dataset.py:142-170 builds the same two lists in a Python loop)."""

    def __init__(self, loaded, n_speakers, phone_labels, phone_size):
        loaded = sorted(loaded, key=lambda item: (item[0], item[1]))
        spk = torch.tensor([item[0] for item in loaded], dtype=torch.long)
        if len(loaded) and (int(spk.min()) < 0 or int(spk.max()) >= n_speakers):
            bad = int(spk[(spk < 0) | (spk >= n_speakers)][0])
            raise ValueError(f"{bad} invalid speaker")
        waves, phones = [], []
        for _, name, wav in loaded:
            if phone_labels is not None:                 # a labelled sequence ends where its labels end
                lab = phone_labels[name]
                phones.extend(lab)
                wav = wav[:len(lab) * phone_size]
            waves.append(wav)
        sizes = torch.tensor([w.size(0) for w in waves], dtype=torch.long)
        zero = sizes.new_zeros(1)
        self.seq_bounds = torch.cat([zero, torch.cumsum(sizes, 0)])
        per_speaker = torch.zeros(int(spk.max()) + 1 if len(loaded) else 0, dtype=torch.long).index_add_(0, spk, sizes)
        self.speaker_bounds = torch.cat([zero, torch.cumsum(per_speaker, 0)])
        self.phones = phones
        self.wave = torch.cat(waves, dim=0) if waves else torch.zeros(0)


class AudioBatchData(Dataset):
    """The reference's in-memory audio data set (cpc/dataset.py:20-246) by name, constructor and public attributes:
    sequences are shuffled and cut into packs of at most MAX_SIZE_LOADED samples; one pack is resident (``data``, with
    ``speakerLabel`` / ``seqLabel`` sample boundaries and ``phoneLabels``) while the next one is decoded in the background."""

    def __init__(self, path, sizeWindow, seqNames, phoneLabelsDict, nSpeakers, nProcessLoader=50,
                 MAX_SIZE_LOADED=4000000000):
        self.dbPath, self.sizeWindow = Path(path), sizeWindow
        self.MAX_SIZE_LOADED, self.nProcessLoader = MAX_SIZE_LOADED, nProcessLoader
        self.seqNames = [(spk, self.dbPath / rel) for spk, rel in seqNames]
        self.speakers = list(range(nSpeakers))
        self.doubleLabels = False
        self.device = torch.device("cpu")
        self.reload_pool = ThreadPoolExecutor(max_workers=max(1, min(nProcessLoader, os.cpu_count() or 1)))
        self._set_phone_labels(phoneLabelsDict, None if phoneLabelsDict is None else phoneLabelsDict["step"])
        self.prepare()
        self.data = []
        self.loadNextPack(first=True)                    # starts decoding pack 0 ...
        self.loadNextPack()                              # ... installs it and starts on pack 1

    def _set_phone_labels(self, labels, step):
        self.phoneLabelsDict = deepcopy(labels)
        self.phoneSize = 0 if labels is None else step
        self.phoneStep = 0 if labels is None else self.sizeWindow // step

    # -- device residency (addition): keep the packed waveform on the GPU
    def to(self, device):
        self.device = torch.device(device)
        self._place()
        return self

    def _place(self):
        if torch.is_tensor(self.data):
            self.data = self.data.to(self.device, non_blocking=True)
        self._speakerBounds = self._bounds_tensor(self.speakerLabel)
        self._phones = self._bounds_tensor(self.phoneLabels) if self.phoneSize > 0 else None

    def resetPhoneLabels(self, newPhoneLabels, step):
        self._set_phone_labels(newPhoneLabels, step)
        self.loadNextPack()

    @staticmethod
    def splitSeqTags(seqName):
        return os.path.normpath(seqName).split(os.sep)

    def getSeqNames(self):
        return [str(where) for _, where in self.seqNames]

    def clear(self):
        for field in ("data", "speakerLabel", "seqLabel", "phoneLabels"):
            self.__dict__.pop(field, None)

    def prepare(self):
        """A new random order of the sequences and its packs (cpc/dataset.py:92-121)."""
        random.shuffle(self.seqNames)
        lengths = list(self.reload_pool.map(extractLength, self.seqNames))
        packs, self.totSize = _cut_into_packs(lengths, self.MAX_SIZE_LOADED)
        self.packageIndex = [list(p) for p in packs]
        self.currentPack, self.nextPack = -1, 0

    def getNPacks(self):
        return len(self.packageIndex)

    def loadNextPack(self, first=False):
        """Install the pack that was being decoded and start decoding the one after it; a full turn over the packs
        reshuffles (cpc/dataset.py:126-140)."""
        self.clear()
        if not first:
            self.currentPack = self.nextPack
            self.nextData = [job.result() for job in self._pending]
            self.parseNextDataBlock()
            del self.nextData
        self.nextPack = (self.currentPack + 1) % self.getNPacks()
        if self.nextPack == 0 and self.getNPacks() > 1:
            self.prepare()
        lo, hi = self.packageIndex[self.nextPack]
        self._pending = [self.reload_pool.submit(loadFile, entry) for entry in self.seqNames[lo:hi]]

    def parseNextDataBlock(self):
        """self.nextData (decoded sequences) -> the resident pack and its boundary lists (cpc/dataset.py:142-170)."""
        pack = _Pack(self.nextData, len(self.speakers), self.phoneLabelsDict, self.phoneSize)
        self.data = pack.wave
        self.seqLabel = pack.seq_bounds.tolist()
        self.speakerLabel = pack.speaker_bounds.tolist()
        self.phoneLabels = pack.phones
        self._place()

    def getPhonem(self, idx):
        first = idx // self.phoneSize
        return self.phoneLabels[first:first + self.phoneStep]

    def getSpeakerLabel(self, idx):
        """Index of the speaker whose samples contain position idx (empty speakers share a boundary and are skipped)."""
        return int(torch.bucketize(torch.tensor(idx), torch.as_tensor(self.speakerLabel), right=True)) - 1

    def __len__(self):
        return self.totSize // self.sizeWindow

    def __getitem__(self, idx):
        """One window starting at sample idx, (1, sizeWindow), with its speaker label -- or its phone labels when the data
        set has them -- or both when ``doubleLabels`` is set (cpc/dataset.py:179-201)."""
        window = self.data[idx:idx + self.sizeWindow].view(1, -1)
        speaker = torch.tensor(self.getSpeakerLabel(idx), dtype=torch.long)
        phones = torch.tensor(self.getPhonem(idx), dtype=torch.long) if self.phoneSize > 0 else torch.zeros(1)
        if self.doubleLabels:
            return window, speaker, phones
        return window, (phones if self.phoneSize > 0 else speaker)

    def get_batch(self, starts):
        """Vectorised counterpart of __getitem__ + default collate for a list of window starts: one gather on the
        device the pack lives on.  -> (B,1,sizeWindow) float, labels as __getitem__ would give them, stacked."""
        idx = torch.as_tensor(starts, dtype=torch.long, device=self.device)
        win = idx[:, None] + torch.arange(self.sizeWindow, device=self.device)[None, :]
        batch = self.data[win].unsqueeze(1)
        speaker = torch.bucketize(idx, self._speakerBounds, right=True) - 1
        if self.phoneSize > 0:
            pidx = (idx // self.phoneSize)[:, None] + torch.arange(self.phoneStep, device=self.device)[None, :]
            phones = self._phones[pidx]
        else:
            phones = torch.zeros(len(idx), 1, device=self.device)
        if self.doubleLabels:
            return batch, speaker, phones
        return batch, (phones if self.phoneSize > 0 else speaker)

    # -- counts (names as in cpc/dataset.py:203-213)
    def getNSpeakers(self):
        return len(self.speakers)

    def getNSeqs(self):
        return len(self.seqLabel) - 1

    def getNLoadsPerEpoch(self):
        return self.getNPacks()

    # -- window plans
    def window_plan(self, type, batchSize, offset, generator=None):
        """All window starts of the pack that is loaded now, for one pass, as ONE tensor on the pack's device (WindowPlan)."""
        common = (self.sizeWindow, batchSize, offset, self.device)
        if type in ("samespeaker", "samesequence"):
            bounds = self.speakerLabel if type == "samespeaker" else self.seqLabel
            return WindowPlan.grouped(self._bounds_tensor(bounds), *common, generator)
        if type == "sequential":
            return WindowPlan.sequential(len(self.data), *common)
        return WindowPlan.uniform(len(self.data), *common, generator)

    def _bounds_tensor(self, bounds):
        return torch.as_tensor(bounds, dtype=torch.long, device=self.device)

    def getBaseSampler(self, type, batchSize, offset):
        """cpc/dataset.py:215-229's name for window_plan: an iterable of index batches with a length."""
        return self.window_plan(type, batchSize, offset)

    def getDataLoader(self, batchSize, type, randomOffset, numWorkers=0, onLoop=-1):
        """cpc/dataset.py:231-258: one pass over every pack (or over pack ``onLoop`` only), a fresh plan -- and a fresh
        random offset in [0, sizeWindow/2] -- per pack.  ``numWorkers`` is accepted and ignored: a batch is one gather."""
        if onLoop < 0:
            return AudioLoader(self, type, batchSize, randomOffset, self.getNPacks())
        self.currentPack = onLoop - 1                    # make pack `onLoop` the resident one
        self.loadNextPack()
        return AudioLoader(self, type, batchSize, randomOffset, 1)


class WindowPlan:
    """The window starts of one pass over a pack, for every sampling type of cpc/dataset.py:318-408, built with tensor
    ops on the device the pack lives on (no Python loop over windows, no host round trip per batch):

      ``starts``  (n_windows,) int64   the windows in the order they are served
      ``ptr``     (n_batches + 1,)     batch j is starts[ptr[j] : ptr[j + 1]]  (the grouped types have short last batches)
      ``order``   (n_batches,)         the order in which the batches are served

    Iterating yields the index tensor of each batch (AudioBatchData.get_batch gathers it in one shot)."""

    def __init__(self, starts, ptr, order):
        self.starts, self.ptr, self.order = starts, ptr, order
        self._ptr_host = ptr.tolist()                    # one transfer per pack: slicing needs host integers
        self._order_host = order.tolist()

    def __len__(self):
        return len(self._order_host)

    def __iter__(self):
        for j in self._order_host:
            yield self.starts[self._ptr_host[j]:self._ptr_host[j + 1]]

    def batches(self):
        """The plan as Python lists (tests, debugging)."""
        return [b.tolist() for b in self]

    @staticmethod
    def _n_windows(span, sizeWindow, offset):
        """Windows that fit in ``span`` samples (an int, or a tensor of interval lengths) once the first ``offset`` are
        skipped; with an offset the reference gives up one window instead of computing (span - offset) // sizeWindow
        (cpc/dataset.py:324-325,347-348,382-383)."""
        n = torch.as_tensor(span) // sizeWindow
        n = (n - (1 if offset > 0 else 0)).clamp(min=0)
        return n if torch.is_tensor(span) else int(n)

    @classmethod
    def uniform(cls, dataSize, sizeWindow, batchSize, offset, device, generator=None):
        """Every window once, in random order, whole batches only (BatchSampler(..., drop_last=True), dataset.py:229)."""
        n = cls._n_windows(dataSize, sizeWindow, offset)
        nb = n // batchSize
        perm = torch.randperm(n, device=device, generator=generator)[:nb * batchSize]
        return cls(offset + sizeWindow * perm, torch.arange(nb + 1, device=device) * batchSize,
                   torch.arange(nb, device=device))

    @classmethod
    def sequential(cls, dataSize, sizeWindow, batchSize, offset, device):
        """Batch item b walks its own contiguous dataSize // batchSize samples, window after window, so that a recurrent
        state carried from batch to batch (CPCAR.keepHidden) always continues the audio it has seen (dataset.py:339-358)."""
        n = max(0, (dataSize // sizeWindow) // batchSize - (1 if offset > 0 else 0))    # one whole step less with an offset
        step = torch.arange(n, device=device).view(n, 1) * sizeWindow
        lane = torch.arange(batchSize, device=device).view(1, batchSize) * (dataSize // batchSize)
        return cls((offset + step + lane).reshape(-1), torch.arange(n + 1, device=device) * batchSize,
                   torch.arange(n, device=device))

    @classmethod
    def grouped(cls, bounds, sizeWindow, batchSize, offset, device, generator=None):
        """'samespeaker' / 'samesequence': every batch comes from ONE interval [bounds[i], bounds[i+1]) (a speaker, a
        sequence); inside an interval every window once in random order, cut into batches of at most batchSize; the
        batches of all intervals are served in random order (dataset.py:361-408)."""
        if int(bounds[0]) != 0:
            raise AttributeError("Sampling intervals should start at zero")
        bounds = bounds.to(device)
        count = cls._n_windows(bounds[1:] - bounds[:-1], sizeWindow, offset)           # windows per interval
        first = torch.cumsum(count, 0) - count                                         # first slot of each interval
        total = int(count.sum())
        interval = torch.repeat_interleave(torch.arange(count.numel(), device=device), count)
        slot = torch.arange(total, device=device) - first[interval]                    # position inside its interval
        # a random order inside each interval: sort by (interval, uniform key) -- the integer part keeps intervals apart
        key = interval.double() + torch.rand(total, device=device, dtype=torch.float64, generator=generator)
        window = slot[torch.argsort(key)]                                              # which window sits at each slot
        starts = offset + window * sizeWindow + bounds[interval]
        nb = (count + batchSize - 1) // batchSize                                      # batches per interval
        batch_of = (torch.cumsum(nb, 0) - nb)[interval] + slot // batchSize            # batch each slot belongs to
        n_batches = int(nb.sum())
        sizes = torch.bincount(batch_of, minlength=n_batches)
        ptr = torch.cat([sizes.new_zeros(1), torch.cumsum(sizes, 0)])
        return cls(starts, ptr, torch.randperm(n_batches, device=device, generator=generator))


# The reference's sampler classes (cpc/dataset.py:318-408), kept by name and constructor for callers that build them
# directly; each is a WindowPlan of the matching kind on the CPU.  The first two iterate single indices / index lists as
# the reference's do (UniformAudioSampler is wrapped in a BatchSampler there).
class UniformAudioSampler(WindowPlan):
    def __init__(self, dataSize, sizeWindow, offset):
        p = WindowPlan.uniform(dataSize, sizeWindow, 1, offset, torch.device("cpu"))
        super().__init__(p.starts, p.ptr, p.order)

    def __iter__(self):
        return iter(self.starts.tolist())


class SequentialSampler(WindowPlan):
    def __init__(self, dataSize, sizeWindow, offset, batchSize):
        p = WindowPlan.sequential(dataSize, sizeWindow, batchSize, offset, torch.device("cpu"))
        super().__init__(p.starts, p.ptr, p.order)

    def __iter__(self):
        return iter(self.batches_as_lists())

    def batches_as_lists(self):
        return [b.tolist() for b in WindowPlan.__iter__(self)]


class SameSpeakerSampler(WindowPlan):
    def __init__(self, batchSize, samplingIntervals, sizeWindow, offset):
        p = WindowPlan.grouped(torch.as_tensor(samplingIntervals, dtype=torch.long), sizeWindow, batchSize, offset,
                               torch.device("cpu"))
        super().__init__(p.starts, p.ptr, p.order)

    def __iter__(self):
        return iter([b.tolist() for b in WindowPlan.__iter__(self)])


class AudioLoader:
    """One pass over the data set: for every pack a fresh WindowPlan, every batch one device-side gather
    (AudioBatchData.get_batch); the next pack -- prefetched in the background while this one was served -- is swapped in
    between.  len() is the reference's estimate, total windows // batchSize (cpc/dataset.py:233,298-299)."""

    def __init__(self, dataset, samplingType, batchSize, randomOffset, nPacks):
        self.dataset, self.samplingType, self.batchSize = dataset, samplingType, batchSize
        self.randomOffset, self.nPacks = randomOffset, nPacks
        self.size = dataset.totSize // (dataset.sizeWindow * batchSize)

    def __len__(self):
        return self.size

    def plan(self):
        offset = random.randint(0, self.dataset.sizeWindow // 2) if self.randomOffset else 0
        return self.dataset.window_plan(self.samplingType, self.batchSize, offset)

    def __iter__(self):
        for pack in range(self.nPacks):
            if pack:
                self.dataset.loadNextPack()
            for index in self.plan():
                yield self.dataset.get_batch(index)


# ----------------------------------------------------------------------------------------------- helpers
def findAllSeqs(dirName, extension=".flac", loadCache=False, speaker_level=1):
    """cpc/dataset.py:417-488: -> ([(speaker index, path relative to dirName)], [speaker names]); the speaker of a file is
    the first ``speaker_level`` directory levels below dirName ('' when 0 or when the files sit in dirName itself), and
    speakers are numbered in the order the walk meets them.  The listing is cached in dirName/_seqs_cache.txt."""
    cache = os.path.join(dirName, "_seqs_cache.txt")
    if loadCache and os.path.isfile(cache):
        try:
            return tuple(torch.load(cache))
        except OSError:
            pass
    base = dirName if dirName.endswith(os.sep) else dirName + os.sep
    index_of, found = {}, []
    for here, _, names in os.walk(base):
        rel = here[len(base):]
        hits = [n for n in names if n.endswith(extension)]
        if hits:
            tag = os.sep.join(rel.split(os.sep)[:speaker_level])
            spk = index_of.setdefault(tag, len(index_of))
            found.extend((spk, os.path.join(rel, n)) for n in hits)
    names_by_index = sorted(index_of, key=index_of.get)
    try:
        torch.save((found, names_by_index), cache)
    except OSError:
        pass
    return found, names_by_index


def parseSeqLabels(pathLabels):
    """cpc/dataset.py:491-501: '<seq> l0 l1 ...' lines -> ({'step': 160, seq: [labels]}, number of classes)."""
    table, top = {"step": 160}, -1
    with open(pathLabels, "r") as f:
        for fields in map(str.split, f):
            if fields:
                labels = list(map(int, fields[1:]))
                table[fields[0]] = labels
                top = max([top] + labels)
    return table, top + 1


def filterSeqs(pathTxt, seqCouples):
    """cpc/dataset.py:504-520: the sequences whose file stem is listed in pathTxt, ordered by stem (the caller's list is
    sorted in place, as the reference does)."""
    def stem(couple):
        return os.path.splitext(os.path.basename(couple[1]))[0]
    with open(pathTxt, "r") as f:
        wanted = set(filter(None, map(str.strip, f)))
    seqCouples.sort(key=stem)
    return [c for c in seqCouples if stem(c) in wanted]
