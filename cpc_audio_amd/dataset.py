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
    """(speaker, path) -> (speaker, sequence name, mono float waveform), cpc/dataset.py:249-258."""
    speaker, fullPath = data
    seq = _reader_for(fullPath)[0](fullPath)
    seq = seq if torch.is_tensor(seq) else torch.from_numpy(np.ascontiguousarray(seq))
    return speaker, Path(fullPath).stem, seq.float()


def extractLength(couple):
    """cpc/dataset.py:411-414."""
    _, locPath = couple
    read, length = _reader_for(locPath)
    return length(locPath) if length is not None else int(read(locPath).shape[0])


# ----------------------------------------------------------------------------------------------- the dataset
class AudioBatchData(Dataset):
    """cpc/dataset.py:20-246."""

    def __init__(self, path, sizeWindow, seqNames, phoneLabelsDict, nSpeakers, nProcessLoader=50,
                 MAX_SIZE_LOADED=4000000000):
        self.MAX_SIZE_LOADED = MAX_SIZE_LOADED
        self.nProcessLoader = nProcessLoader
        self.dbPath = Path(path)
        self.sizeWindow = sizeWindow
        self.seqNames = [(s, self.dbPath / x) for s, x in seqNames]
        self.reload_pool = ThreadPoolExecutor(max_workers=max(1, min(nProcessLoader, os.cpu_count() or 1)))
        self.device = torch.device("cpu")

        self.prepare()
        self.speakers = list(range(nSpeakers))
        self.data = []

        self.phoneSize = 0 if phoneLabelsDict is None else phoneLabelsDict["step"]
        self.phoneStep = 0 if phoneLabelsDict is None else self.sizeWindow // self.phoneSize
        self.phoneLabelsDict = deepcopy(phoneLabelsDict)
        self.loadNextPack(first=True)
        self.loadNextPack()
        self.doubleLabels = False

    # -- device residency (addition): keep the packed waveform on the GPU
    def to(self, device):
        self.device = torch.device(device)
        self._place()
        return self

    def _place(self):
        if torch.is_tensor(self.data):
            self.data = self.data.to(self.device, non_blocking=True)
        self._speakerBounds = torch.tensor(self.speakerLabel, dtype=torch.long, device=self.device)
        self._phones = (torch.tensor(self.phoneLabels, dtype=torch.long, device=self.device)
                        if self.phoneSize > 0 else None)

    def resetPhoneLabels(self, newPhoneLabels, step):
        self.phoneSize = step
        self.phoneStep = self.sizeWindow // self.phoneSize
        self.phoneLabelsDict = deepcopy(newPhoneLabels)
        self.loadNextPack()

    @staticmethod
    def splitSeqTags(seqName):
        return os.path.normpath(seqName).split(os.sep)

    def getSeqNames(self):
        return [str(x[1]) for x in self.seqNames]

    def clear(self):
        for name in ("data", "speakerLabel", "phoneLabels", "seqLabel"):
            if name in self.__dict__:
                delattr(self, name)

    def prepare(self):
        """Shuffle the sequences and cut them into packs of at most MAX_SIZE_LOADED samples (cpc/dataset.py:92-121)."""
        random.shuffle(self.seqNames)
        allLength = list(self.reload_pool.map(extractLength, self.seqNames))
        self.packageIndex, self.totSize = [], 0
        start, packageSize = 0, 0
        for index, length in enumerate(allLength):
            packageSize += length
            if packageSize > self.MAX_SIZE_LOADED:
                self.packageIndex.append([start, index])
                self.totSize += packageSize
                start, packageSize = index, 0
        if packageSize > 0:
            self.packageIndex.append([start, len(self.seqNames)])
            self.totSize += packageSize
        self.currentPack = -1
        self.nextPack = 0

    def getNPacks(self):
        return len(self.packageIndex)

    def loadNextPack(self, first=False):
        """Install the pack that was being prefetched and start prefetching the following one (cpc/dataset.py:126-140)."""
        self.clear()
        if not first:
            self.currentPack = self.nextPack
            self.nextData = [f.result() for f in self._pending]
            self.parseNextDataBlock()
            del self.nextData
        self.nextPack = (self.currentPack + 1) % len(self.packageIndex)
        seqStart, seqEnd = self.packageIndex[self.nextPack]
        if self.nextPack == 0 and len(self.packageIndex) > 1:
            self.prepare()
        self._pending = [self.reload_pool.submit(loadFile, s) for s in self.seqNames[seqStart:seqEnd]]

    def parseNextDataBlock(self):
        """Concatenate the pack, speaker-major, and record speaker / sequence boundaries (cpc/dataset.py:142-170)."""
        self.speakerLabel = [0]
        self.seqLabel = [0]
        self.phoneLabels = []
        speakerSize = 0
        indexSpeaker = 0
        self.nextData.sort(key=lambda x: (x[0], x[1]))
        chunks = []
        for speaker, seqName, seq in self.nextData:
            while self.speakers[indexSpeaker] < speaker:
                indexSpeaker += 1
                self.speakerLabel.append(speakerSize)
            if self.speakers[indexSpeaker] != speaker:
                raise ValueError(f"{speaker} invalid speaker")
            if self.phoneLabelsDict is not None:
                self.phoneLabels += self.phoneLabelsDict[seqName]
                seq = seq[:len(self.phoneLabelsDict[seqName]) * self.phoneSize]
            sizeSeq = seq.size(0)
            chunks.append(seq)
            self.seqLabel.append(self.seqLabel[-1] + sizeSeq)
            speakerSize += sizeSeq
        self.speakerLabel.append(speakerSize)
        self.data = torch.cat(chunks, dim=0)
        self._place()

    def getPhonem(self, idx):
        idPhone = idx // self.phoneSize
        return self.phoneLabels[idPhone:(idPhone + self.phoneStep)]

    def getSpeakerLabel(self, idx):
        return next(i for i, bound in enumerate(self.speakerLabel) if bound > idx) - 1

    def __len__(self):
        return self.totSize // self.sizeWindow

    def __getitem__(self, idx):
        outData = self.data[idx:(self.sizeWindow + idx)].view(1, -1)
        label = torch.tensor(self.getSpeakerLabel(idx), dtype=torch.long)
        if self.phoneSize > 0:
            label_phone = torch.tensor(self.getPhonem(idx), dtype=torch.long)
            if not self.doubleLabels:
                label = label_phone
        else:
            label_phone = torch.zeros(1)
        if self.doubleLabels:
            return outData, label, label_phone
        return outData, label

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
            if self.doubleLabels:
                return batch, speaker, phones
            return batch, phones
        if self.doubleLabels:
            return batch, speaker, torch.zeros(len(idx), 1, device=self.device)
        return batch, speaker

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
        n = len(self.data)
        if type == "samespeaker":
            return WindowPlan.grouped(self._bounds_tensor(self.speakerLabel), self.sizeWindow, batchSize, offset,
                                      self.device, generator)
        if type == "samesequence":
            return WindowPlan.grouped(self._bounds_tensor(self.seqLabel), self.sizeWindow, batchSize, offset, self.device,
                                      generator)
        if type == "sequential":
            return WindowPlan.sequential(n, self.sizeWindow, batchSize, offset, self.device)
        return WindowPlan.uniform(n, self.sizeWindow, batchSize, offset, self.device, generator)

    def _bounds_tensor(self, bounds):
        return torch.as_tensor(bounds, dtype=torch.long, device=self.device)

    def getBaseSampler(self, type, batchSize, offset):
        """cpc/dataset.py:215-229's name for window_plan: an iterable of index batches with a length."""
        return self.window_plan(type, batchSize, offset)

    def getDataLoader(self, batchSize, type, randomOffset, numWorkers=0, onLoop=-1):
        """cpc/dataset.py:231-258: one pass over every pack (or over pack ``onLoop`` only), a fresh plan -- and a fresh
        random offset in [0, sizeWindow/2] -- per pack.  ``numWorkers`` is accepted and ignored: a batch is one gather."""
        packs = self.getNPacks()
        if onLoop >= 0:
            self.currentPack = onLoop - 1
            self.loadNextPack()
            packs = 1
        return AudioLoader(self, type, batchSize, randomOffset, packs)


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
    """cpc/dataset.py:417-488: -> ([(speaker index, path relative to dirName)], [speaker names]); the speaker is the
    first ``speaker_level`` directory levels below dirName ('' when 0 or when files sit in dirName itself)."""
    cache_path = os.path.join(dirName, "_seqs_cache.txt")
    if loadCache:
        try:
            outSequences, speakers = torch.load(cache_path)
            return outSequences, speakers
        except OSError:
            pass
    if dirName[-1] != os.sep:
        dirName += os.sep
    prefixSize = len(dirName)
    speakersTarget = {}
    outSequences = []
    for root, _, filenames in os.walk(dirName):
        files = [f for f in filenames if f.endswith(extension)]
        if not files:
            continue
        speakerStr = os.sep.join(root[prefixSize:].split(os.sep)[:speaker_level])
        speaker = speakersTarget.setdefault(speakerStr, len(speakersTarget))
        for filename in files:
            outSequences.append((speaker, os.path.join(root[prefixSize:], filename)))
    outSpeakers = [None] * len(speakersTarget)
    for key, index in speakersTarget.items():
        outSpeakers[index] = key
    try:
        torch.save((outSequences, outSpeakers), cache_path)
    except OSError:
        pass
    return outSequences, outSpeakers


def parseSeqLabels(pathLabels):
    """cpc/dataset.py:491-501: '<seq> l0 l1 ...' lines -> ({'step': 160, seq: [labels]}, number of classes)."""
    output = {"step": 160}
    maxPhone = 0
    with open(pathLabels, "r") as f:
        for line in f:
            data = line.split()
            if not data:
                continue
            output[data[0]] = [int(x) for x in data[1:]]
            maxPhone = max(maxPhone, max(output[data[0]]))
    return output, maxPhone + 1


def filterSeqs(pathTxt, seqCouples):
    """cpc/dataset.py:504-520: keep the sequences whose file stem is listed in pathTxt."""
    with open(pathTxt, "r") as f:
        wanted = {line.strip() for line in f if line.strip()}
    seqCouples.sort(key=lambda x: os.path.basename(os.path.splitext(x[1])[0]))
    return [x for x in seqCouples if os.path.basename(os.path.splitext(x[1])[0]) in wanted]
