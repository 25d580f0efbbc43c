"""Window feeder: the step in front of the hot path.

Same names and behaviour as the reference's cpc/dataset.py for the parts the training loop uses --
findAllSeqs (:771-948, format=None branch), filterSeqs (:963-978), AudioBatchData (:23-408: sequences sorted by
(speaker, name) and concatenated into ONE flat audio vector, speaker / sequence interval tables, chunked
"packs" of at most MAX_SIZE_LOADED samples, getDataLoader with the uniform / sequential / samespeaker /
samesequence / temporalsamespeaker samplers :603-757 and the random window offset :395-403) -- but MI355X-first:
the flat vector lives in HBM (288 GB: the reference's 4e9-sample pack is 16 GB) and a batch is cut out of it
by one gather kernel (cpc_window_gather); the host only produces b int64 offsets per step.  Files are decoded
by the library's own FLAC / WAV readers (cpc2_amd/audio.py).

Signal-quality side files (per-file snr / c50 estimates, dataset.py:69-77,106-120,257-281) are read too and come out as
the third element of a batch.  Not on this path: data augmentation (sox / WavAugment), phone labels.
Without augmentation the reference yields past == future (dataset.py:308-321): batches are returned as an
expanded [b, 2, 1, W] view whose two halves alias, which cpcStep(dedup=True) can exploit.
"""
import os
import random
from pathlib import Path

import torch

from . import _lib, audio
from ._lib import check, ptr, stream_ptr


def findAllSeqs(dirName, extension='.flac', speaker_level=1, **unused):
    """(outSequences [(speaker_index, relative_path)], outSpeakers) -- dataset.py:771-948 with format=None."""
    dirName = str(dirName)
    if dirName[-1] != os.sep:
        dirName += os.sep
    prefixSize = len(dirName)
    speakersTarget, outSequences = {}, []
    for root, _dirs, filenames in os.walk(dirName, followlinks=True):
        filtered_files = [f for f in filenames if f.endswith(extension)]
        if filtered_files:
            speakerStr = os.sep.join(root[prefixSize:].split(os.sep)[:speaker_level])
            speaker = speakersTarget.setdefault(speakerStr, len(speakersTarget))
            for filename in filtered_files:
                outSequences.append((speaker, os.path.join(root[prefixSize:], filename)))
    outSpeakers = [None] * len(speakersTarget)
    for key, index in speakersTarget.items():
        outSpeakers[index] = key
    return outSequences, outSpeakers


def filterSeqs(pathTxt, seqCouples):
    """keep the sequences whose base name is listed in pathTxt -- dataset.py:963-978."""
    with open(pathTxt, 'r') as f:
        wanted = {p.strip() for p in f.readlines() if p.strip()}
    seqCouples = sorted(seqCouples, key=lambda x: os.path.basename(os.path.splitext(x[1])[0]))
    return [x for x in seqCouples if os.path.basename(os.path.splitext(x[1])[0]) in wanted]


def shard_for_rank(files, rank, world_size):
    """The contiguous slice of a sequence list one data-parallel rank loads -- cpc/train.py:389-393 (`filter_distributed`):
    files[len * rank // world : len * (rank + 1) // world].  The slices of the world_size ranks partition the list in order,
    sizes differ by at most one, and world_size == 1 returns the list itself.  Applied by the caller to the train, validation
    (and noise) lists alike, after the train / validation split, as the reference does (:396-399)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"shard_for_rank: rank {rank} outside a world of {world_size}")
    start = len(files) * rank // world_size
    end = len(files) * (rank + 1) // world_size
    return files[start:end]


# --------------------------------------------------------------------------- samplers (dataset.py:603-757)
def _uniform_batches(dataSize, sizeWindow, offset, batchSize):
    n = dataSize // sizeWindow - (1 if offset > 0 else 0)
    idx = (offset + sizeWindow * torch.randperm(max(n, 0))).tolist()
    return [idx[i:i + batchSize] for i in range(0, len(idx) - batchSize + 1, batchSize)]      # drop_last=True


def _sequential_batches(dataSize, sizeWindow, offset, batchSize):
    n = (dataSize // sizeWindow) // batchSize - (1 if offset > 0 else 0)
    starts = [x * (dataSize // batchSize) for x in range(batchSize)]
    return [[offset + sizeWindow * i + s for s in starts] for i in range(max(n, 0))]


def _same_interval_batches(intervals, sizeWindow, offset, batchSize):
    """SameSpeakerSampler: every batch comes from one interval (speaker or sequence)."""
    if intervals[0] != 0:
        raise AttributeError("Sampling intervals should start at zero")
    sizes = [(intervals[i + 1] - intervals[i]) // sizeWindow for i in range(len(intervals) - 1)]
    if offset > 0:
        sizes = [max(0, x - 1) for x in sizes]
    batches = []
    for i, val in enumerate(sizes):
        if val <= 0:
            continue
        perm = torch.randperm(val).tolist()
        for s in range(0, val, batchSize):
            batches.append([offset + x * sizeWindow + intervals[i] for x in perm[s:s + batchSize]])
    random.shuffle(batches)
    return batches


def _temporal_same_interval_batches(intervals, sizeWindow, offset, batchSize):
    """TemporalSameSpeakerSampler: a batch is batchSize CONSECUTIVE windows of one interval."""
    if intervals[0] != 0:
        raise AttributeError("Sampling intervals should start at zero")
    sizes = [(intervals[i + 1] - intervals[i]) // (sizeWindow * batchSize) for i in range(len(intervals) - 1)]
    if offset > 0:
        sizes = [max(0, x - 1) for x in sizes]
    if sum(sizes) == 0:
        raise ValueError("No sampling intervals can be found. Try to increase --max_size_loaded or to reduce the batch size.")
    batches = []
    for i, val in enumerate(sizes):
        for x in torch.randperm(val).tolist():
            beg = offset + x * sizeWindow * batchSize + intervals[i]
            batches.append(list(range(beg, beg + sizeWindow * batchSize, sizeWindow)))
    random.shuffle(batches)
    return batches


class AudioBatchData:
    """dataset.py:23-408 (see the module docstring for what is and is not carried over)."""

    def __init__(self, path, sizeWindow, seqNames, phoneLabelsDict, nSpeakers, nProcessLoader=10,
                 MAX_SIZE_LOADED=4000000000, transform=None, augment_past=False, augment_future=False,
                 augmentation=None, keep_temporality=True, past_equal_future=False, signal_quality_path=None,
                 signal_quality_step=1600, signal_quality_mode=None, device=None):
        if phoneLabelsDict is not None:
            raise NotImplementedError("phone labels are not on the MI355X feeder path")
        if transform is not None or augment_past or augment_future or augmentation is not None:
            raise NotImplementedError("audio augmentation is not on the MI355X feeder path")
        self.MAX_SIZE_LOADED = MAX_SIZE_LOADED
        self.dbPath = Path(path)
        self.sizeWindow = sizeWindow
        self.seqNames = [(s, self.dbPath / x) for s, x in seqNames]
        self.keep_temporality = keep_temporality
        self.speakers = list(range(nSpeakers))
        self.device = torch.device(device) if device is not None else \
            torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.doubleLabels = False
        self.phoneSize = 0
        # signal-quality estimates (dataset.py:69-77,106-120): one .pt per audio file under signal_quality_path (a list of
        # tensors that concatenate along dim 1 to [frames, 2] = (snr, c50), one frame per signal_quality_step samples) and
        # min_max.csv with the normalisation bounds
        self.signal_quality_path = Path(signal_quality_path) if signal_quality_path is not None else None
        self.signal_quality_step = signal_quality_step
        self.signal_quality_size = sizeWindow // signal_quality_step
        self.signal_quality_mode = signal_quality_mode
        self.data_quality = None
        if self.signal_quality_path is not None:
            self._init_min_max_signal_quality()
        self.prepare()
        self.loadNextPack(first=True)
        self.loadNextPack()

    def _init_min_max_signal_quality(self):
        import csv
        file_path = self.signal_quality_path / "min_max.csv"
        if not file_path.is_file():
            raise FileNotFoundError("Can not find file containing min/max values of snr and c50 under: %s" % file_path)
        with open(file_path, "r") as fin:
            reader = csv.reader(fin)
            bounds = dict(zip(next(reader), next(reader)))
        try:
            self.min_snr, self.max_snr = float(bounds["min_snr"]), float(bounds["max_snr"])
            self.min_c50, self.max_c50 = float(bounds["min_c50"]), float(bounds["max_c50"])
        except (KeyError, ValueError):
            raise ValueError("min_max.csv should contain the following keys: min_snr, max_snr, min_c50, max_c50.")

    def _quality_file(self, audio_path):
        rel = os.path.relpath(str(audio_path), str(self.dbPath))
        return self.signal_quality_path / (os.path.splitext(rel)[0] + ".pt")       # dataset.py:166-168

    # ---- pack bookkeeping (what dataset.py:147-190 computes, restated on cumulative sizes)
    def prepare(self):
        """New random order of the files (sessions -- runs of equal speaker index -- stay together when keep_temporality) and the
        packs [first file, one past the last) of at most MAX_SIZE_LOADED samples.  The reference's rule, kept because its unit
        tests pin it: a pack is closed by the first file that takes the running size beyond the limit; that file OPENS the next
        pack but is not counted in the next pack's running size, while totSize counts it with the pack it closed."""
        import itertools
        import numpy as np
        if self.keep_temporality:
            sessions = [list(run) for _spk, run in itertools.groupby(self.seqNames, key=lambda item: item[0])]
            random.shuffle(sessions)
            self.seqNames = list(itertools.chain.from_iterable(sessions))
        else:
            random.shuffle(self.seqNames)
        ends = np.cumsum([audio.info(p)[2] for _, p in self.seqNames], dtype=np.int64)      # samples up to and including file i
        n_files = len(ends)
        self.packageIndex, self.totSize = [], 0
        first, counted_from = 0, 0          # the open pack starts at file `first`; its running size counts files >= counted_from
        while True:
            before = int(ends[counted_from - 1]) if counted_from > 0 else 0
            closing = int(np.searchsorted(ends, before + self.MAX_SIZE_LOADED, side="right"))   # first file that exceeds the limit
            if closing >= n_files:
                tail = int(ends[-1]) - before if counted_from < n_files else 0
                if tail > 0:
                    self.packageIndex.append([first, n_files])
                    self.totSize += tail
                break
            self.packageIndex.append([first, closing])
            self.totSize += int(ends[closing]) - before
            first, counted_from = closing, closing + 1
        self.currentPack = -1
        self.nextPack = 0

    def getNPacks(self):
        return len(self.packageIndex)

    def _load_pack(self, pack):
        start, end = self.packageIndex[pack]
        items = []
        for speaker, p in self.seqNames[start:end]:
            wav = audio.load(p)[0].mean(dim=0)              # dataset.py:425: mono mix
            item = (speaker, os.path.splitext(os.path.basename(str(p)))[0], wav)
            if self.signal_quality_path is not None:        # dataset.py:427-430: the audio is cut to whole quality frames
                quality = torch.cat(torch.load(self._quality_file(p)), dim=1).float()
                item = item[:2] + (wav[:quality.shape[0] * self.signal_quality_step], quality)
            items.append(item)
        return items

    def loadNextPack(self, first=False):
        """dataset.py:192-223: make the pack that was read ahead the current one (unless this is the very first call), then read
        the following pack ahead -- cyclically; when the cycle restarts and there is more than one pack, the files are re-shuffled
        and re-packed first."""
        if not first:
            self.currentPack = self.nextPack
            ahead, self.nextData = self.nextData, None
            self.parseNextDataBlock(ahead)
        n_packs = len(self.packageIndex)
        self.nextPack = (self.currentPack + 1) % n_packs
        if n_packs > 1 and self.nextPack == 0:
            self.prepare()
        self.nextData = self._load_pack(self.nextPack)

    def parseNextDataBlock(self, nextData):
        """dataset.py:225-268: the pack's files in (speaker, name) order as ONE flat vector on the device, with the tables of
        where every sequence (seqLabel) and every speaker index up to the last one present (speakerLabel) begins."""
        import numpy as np
        ordered = sorted(nextData, key=lambda item: (item[0], item[1]))
        sizes = np.array([item[2].size(0) for item in ordered], dtype=np.int64)
        who = np.array([item[0] for item in ordered], dtype=np.int64)
        valid = set(self.speakers)
        for speaker in who.tolist():
            if speaker not in valid:
                raise ValueError(f'{speaker} invalid speaker')
        self.seqLabel = [0] + np.cumsum(sizes).tolist()
        per_speaker = np.bincount(who, weights=sizes, minlength=int(who.max()) + 1 if len(who) else 1).astype(np.int64)
        self.speakerLabel = [0] + np.cumsum(per_speaker).tolist()
        self.data = torch.cat([item[2] for item in ordered], dim=0).to(self.device)     # resident on the device
        quality = [item[3] for item in ordered if len(item) > 3]
        if quality:                                              # dataset.py:257-265: min-max normalised, third column = mean
            q = torch.cat(quality, dim=0)
            q[:, 0] = (q[:, 0] - self.min_snr) / (self.max_snr - self.min_snr)
            q[:, 1] = (q[:, 1] - self.min_c50) / (self.max_c50 - self.min_c50)
            self.data_quality = torch.cat((q, torch.mean(q, dim=1).view(-1, 1)), dim=1).to(self.device)

    # ---- accessors
    def getSpeakerLabel(self, idx):
        import bisect
        return bisect.bisect_right(self.speakerLabel, idx) - 1

    def getSignalQuality(self, idx):
        """dataset.py:271-281: the window's signal_quality_size estimates of the selected kind."""
        column = {"snr": 0, "c50": 1, "snr_c50": 2}.get(self.signal_quality_mode)
        if column is None:
            raise ValueError("--signal_quality_mode should be in ['snr', 'c50', 'snr_c50'].")
        first = idx // self.signal_quality_step
        return self.data_quality[first:first + self.signal_quality_size, column]

    def __len__(self):
        return self.totSize // self.sizeWindow

    def getNSpeakers(self):
        return len(self.speakers)

    def getNSeqs(self):
        return len(self.seqLabel) - 1

    def getNLoadsPerEpoch(self):
        return len(self.packageIndex)

    def windows(self, offsets):
        """[b, 2, 1, sizeWindow] batch (past and future alias: no augmentation) for int window offsets."""
        b = len(offsets)
        off = torch.tensor(offsets, dtype=torch.int64)
        if self.data.is_cuda:
            off = off.to(self.device)
            out = torch.empty(b, 1, self.sizeWindow, dtype=torch.float32, device=self.device)
            check(_lib.load().cpc_window_gather(ptr(self.data), self.data.numel(), ptr(off), ptr(out), b, self.sizeWindow,
                                                stream_ptr(self.device)), "window_gather")
        else:
            out = torch.stack([self.data[o:o + self.sizeWindow] for o in offsets]).view(b, 1, self.sizeWindow)
        return out.unsqueeze(1).expand(b, 2, 1, self.sizeWindow)

    def windows_from(self, off_dev):
        """The same batch from offsets that already are a device int64 tensor (the loader uploads a whole pack's offsets at once)."""
        b = off_dev.numel()
        out = torch.empty(b, 1, self.sizeWindow, dtype=torch.float32, device=self.device)
        check(_lib.load().cpc_window_gather(ptr(self.data), self.data.numel(), ptr(off_dev), ptr(out), b, self.sizeWindow,
                                            stream_ptr(self.device)), "window_gather")
        return out.unsqueeze(1).expand(b, 2, 1, self.sizeWindow)

    def getBaseSampler(self, type, batchSize, offset, batchSizePerGPU=None):
        n = self.data.numel()
        if type == "samespeaker":
            return _same_interval_batches(self.speakerLabel, self.sizeWindow, offset, batchSize)
        if type == "samesequence":
            return _same_interval_batches(self.seqLabel, self.sizeWindow, offset, batchSize)
        if type == "temporalsamespeaker":
            return _temporal_same_interval_batches(self.speakerLabel, self.sizeWindow, offset, batchSize)
        if type == "sequential":
            return _sequential_batches(n, self.sizeWindow, offset, batchSize)
        if type == "uniform":
            return _uniform_batches(n, self.sizeWindow, offset, batchSize)
        raise ValueError("--samplingType should belong to %s" % ["samespeaker", "samesequence", "temporalsamespeaker",
                                                                "sequential", "uniform"])

    def getDataLoader(self, batchSize, type, randomOffset, numWorkers=0, onLoop=-1, nLoops=-1, **unused):
        """Iterable of (sequence [b,2,1,W] on the device, speaker label [b]) over nLoops packs -- dataset.py:366-408."""
        if onLoop >= 0:
            self.currentPack = onLoop - 1
            self.loadNextPack()
            nLoops = 1 if nLoops <= 0 else nLoops
        elif nLoops <= 0:
            nLoops = len(self.packageIndex)
        return _AudioLoader(self, batchSize, type, randomOffset, nLoops)


class _AudioLoader:
    def __init__(self, dataset, batchSize, type, randomOffset, nLoops):
        self.dataset, self.batchSize, self.type, self.randomOffset, self.nLoops = dataset, batchSize, type, randomOffset, nLoops

    def _sampler(self):
        d = self.dataset
        if self.randomOffset:                                # dataset.py:395-403
            offset = random.randint(0, d.sizeWindow * self.batchSize) if self.type == "temporalsamespeaker" \
                else random.randint(0, d.sizeWindow // 2)
        else:
            offset = 0
        return d.getBaseSampler(self.type, self.batchSize, offset)

    def __len__(self):
        return self.dataset.totSize // (self.dataset.sizeWindow * self.batchSize)

    def __iter__(self):
        d = self.dataset
        for loop in range(self.nLoops):
            limit = d.data.numel() - d.sizeWindow
            batches = [[o for o in batch if 0 <= o <= limit] for batch in self._sampler()]
            batches = [batch for batch in batches if batch]
            if d.data.is_cuda and batches:
                # The whole pack's window offsets go to the device in ONE pinned, asynchronous copy and the speaker labels are looked
                # up there (bucketize over the speaker table == dataset.py:254's bisect): a step of the loop then uploads nothing --
                # a pageable `torch.tensor(...).to(device)` per step held the host until the device had caught up, i.e. the loop
                # ran in lock step with the GPU (round 6: host 0.15 ms ahead of the device instead of two steps)
                width = max(len(batch) for batch in batches)
                host = torch.zeros(len(batches), width, dtype=torch.int64).pin_memory()
                for i, batch in enumerate(batches):
                    host[i, :len(batch)] = torch.tensor(batch, dtype=torch.int64)
                offs = host.to(d.device, non_blocking=True)
                table = torch.tensor(d.speakerLabel, dtype=torch.int64).pin_memory().to(d.device, non_blocking=True)
                labels = torch.bucketize(offs, table, right=True) - 1
                for i, batch in enumerate(batches):
                    off_dev, label = offs[i, :len(batch)], labels[i, :len(batch)]
                    if d.signal_quality_path is not None:    # dataset.py:327-330: a third element per sample
                        yield d.windows_from(off_dev), label, torch.stack([d.getSignalQuality(o) for o in batch])
                    else:
                        yield d.windows_from(off_dev), label
            else:
                for batch in batches:
                    label = torch.tensor([d.getSpeakerLabel(o) for o in batch], dtype=torch.long, device=d.device)
                    if d.signal_quality_path is not None:
                        yield d.windows(batch), label, torch.stack([d.getSignalQuality(o) for o in batch])
                    else:
                        yield d.windows(batch), label
            if loop + 1 < self.nLoops or len(d.packageIndex) > 1:
                d.loadNextPack()
