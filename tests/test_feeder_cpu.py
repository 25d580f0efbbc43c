"""Window feeder on CPU: the reference's own data-loader expectations (cpc/unit_tests.py:15-170, TestDataLoader)
on the reference's own fixture (cpc/test_data: 9 LibriSpeech FLACs, copied under tests/golden/test_db), plus the
native FLAC decoder (every file is verified against the MD5 stored in its STREAMINFO)."""
import os
from pathlib import Path

import torch

from cpc2_amd import audio
from cpc2_amd.dataset import AudioBatchData, filterSeqs, findAllSeqs, shard_for_rank

GOLD = Path(__file__).parent / "golden"
DB = GOLD / "test_db"
SEQ_LIST = GOLD / "seq_list.txt"
EXPECTED = [(0, '2911/12359/2911-12359-0007.flac'), (1, '4051/11218/4051-11218-0044.flac'),
            (2, '4397/15668/4397-15668-0003.flac'), (2, '4397/15668/4397-15668-0007.flac'),
            (3, '5393/19218/5393-19218-0024.flac'), (4, '5678/43301/5678-43301-0021.flac'),
            (4, '5678/43303/5678-43303-0024.flac'), (4, '5678/43303/5678-43303-0032.flac'),
            (5, '6476/57446/6476-57446-0019.flac')]


def test_flac_decoder_md5_and_lengths():
    total = 0
    for _spk, rel in EXPECTED:
        wav, sr = audio.load(DB / rel)                 # raises if the MD5 of the decoded PCM differs
        assert sr == 16000 and wav.shape[0] == 1 and wav.dtype == torch.float32
        assert audio.info(DB / rel) == (16000, 1, wav.shape[1])
        assert float(wav.abs().max()) < 1.0 and float(wav.std()) > 0.01
        total += wav.shape[1]
    assert total == 1864080                             # 116.505 s of 16 kHz audio = 91 windows of 20480


def test_find_all_seqs():                               # unit_tests.py:32-66
    seq_names, speakers = findAllSeqs(str(DB), extension=".flac")
    assert len(speakers) == 6 and set(speakers) == {'2911', '4051', '4397', '5393', '5678', '6476'}
    assert {x[0] for x in seq_names} == {x[0] for x in EXPECTED}
    assert {x[1] for x in seq_names} == {x[1] for x in EXPECTED}
    for index_speaker, seq_name in seq_names:
        assert speakers[index_speaker] == str(Path(seq_name).stem).split('-')[0]


def test_find_all_seqs_speaker_levels():                # unit_tests.py:68-105
    seq_names, speakers = findAllSeqs(str(DB), extension=".flac", speaker_level=2)
    assert set(speakers) == {'2911/12359', '4051/11218', '4397/15668', '5393/19218', '5678/43301', '5678/43303', '6476/57446'}
    for index_speaker, seq_name in seq_names:
        assert speakers[index_speaker] == '/'.join(str(Path(seq_name).stem).split('-')[:2])
    assert findAllSeqs(str(DB / '2911/12359/'), extension=".flac")[1] == ['']
    assert findAllSeqs(str(DB), extension=".flac", speaker_level=0)[1] == ['']


def _filtered():
    seq_names, speakers = findAllSeqs(str(DB), extension=".flac")
    return filterSeqs(SEQ_LIST, seq_names), speakers


def test_load_data():                                   # unit_tests.py:107-129
    seq_names, _ = _filtered()
    assert {x[1] for x in seq_names} == {x[1] for x in EXPECTED[2:]} and len(seq_names) == 7
    data = AudioBatchData(DB, 20480, seq_names, None, 9, device="cpu")
    assert data.getNSpeakers() == 9 and data.getNSeqs() == 7
    assert data.data.numel() == sum(audio.info(DB / x[1])[2] for x in seq_names)


def test_data_loader_samespeaker():                     # unit_tests.py:131-149
    seq_names, speakers = _filtered()
    data = AudioBatchData(DB, 20480, seq_names, None, len(speakers), device="cpu")
    visited = set()
    for seq, labels in data.getDataLoader(2, "samespeaker", True, numWorkers=2):
        assert seq.shape[1:] == (2, 1, 20480) and seq.shape[0] == labels.shape[0]
        assert torch.equal(seq[:, 0], seq[:, 1])        # no augmentation: past == future
        p = labels[0].item()
        visited.add(p)
        assert int(torch.sum(labels == p)) == labels.size(0)
    assert len(visited) == 4


def test_partial_loader_two_packs():                    # unit_tests.py:151-170
    seq_names, speakers = _filtered()
    data = AudioBatchData(DB, 20480, seq_names, None, len(speakers), MAX_SIZE_LOADED=1000000, device="cpu")
    assert data.getNPacks() == 2
    visited = set()
    for _seq, labels in data.getDataLoader(16, "samespeaker", True, numWorkers=2):
        p = labels[0].item()
        assert int(torch.sum(labels == p)) == labels.size(0)
        visited.add(p)
    assert len(visited) == 4


def test_other_samplers_yield_valid_windows():
    seq_names, speakers = _filtered()
    data = AudioBatchData(DB, 20480, seq_names, None, len(speakers), device="cpu")
    flat = data.data
    for kind in ("uniform", "sequential", "samesequence", "temporalsamespeaker"):
        n = 0
        for seq, labels in data.getDataLoader(4, kind, False):
            assert seq.shape[1:] == (2, 1, 20480) and labels.shape[0] == seq.shape[0] <= 4
            assert kind == "samesequence" or seq.shape[0] == 4      # only interval samplers emit short last batches
            n += 1
        assert n >= 1, kind
    # a window is a verbatim slice of the flat vector
    w = data.windows([12345])
    assert torch.equal(w[0, 0, 0], flat[12345:12345 + 20480])


def test_signal_quality_side_files(tmp_path):
    """dataset.py:69-77,106-120,166-168,257-281,327-330: per-file estimates (snr, c50) every signal_quality_step samples,
    min-max normalised with min_max.csv, a third column with their mean, the audio cut to whole estimate frames, and one
    [signal_quality_size] slice per window as a third element of every batch."""
    seq_names, _ = _filtered()
    step, window = 1600, 20480
    truth = {}
    for i, (_spk, rel) in enumerate(seq_names):
        frames = audio.info(DB / rel)[2] // step - (i % 2)             # some files carry fewer estimate frames than audio
        snr = torch.linspace(-5.0, 30.0, frames) + i
        c50 = torch.linspace(60.0, 0.0, frames) - i
        out = tmp_path / (os.path.splitext(rel)[0] + ".pt")
        out.parent.mkdir(parents=True, exist_ok=True)
        torch.save([snr.view(-1, 1), c50.view(-1, 1)], out)          # a list that concatenates along dim 1
        truth[os.path.splitext(os.path.basename(rel))[0]] = (snr, c50)
    with open(tmp_path / "min_max.csv", "w") as fh:
        fh.write("min_snr,max_snr,min_c50,max_c50\n-10,40,-10,70\n")
    for mode, column in (("snr", 0), ("c50", 1), ("snr_c50", 2)):
        data = AudioBatchData(DB, window, seq_names, None, 9, device="cpu", signal_quality_path=tmp_path,
                              signal_quality_step=step, signal_quality_mode=mode)
        assert data.signal_quality_size == 12
        # the flat audio holds whole estimate frames only, and as many frames as the quality table
        assert data.data.numel() == data.data_quality.shape[0] * step and data.data_quality.shape[1] == 3
        assert torch.allclose(data.data_quality[:, 2], data.data_quality[:, :2].mean(dim=1))
        batches = list(data.getDataLoader(4, "sequential", False))
        assert batches and all(len(bt) == 3 for bt in batches)
        seq, label, quality = batches[0]
        assert seq.shape == (4, 2, 1, window) and quality.shape == (4, 12) and quality.dtype == torch.float32
        # first window of the flat vector = start of the first sequence in (speaker, name) order
        first = sorted(seq_names, key=lambda x: (x[0], os.path.basename(x[1])))[0][1]
        snr, c50 = truth[os.path.splitext(os.path.basename(first))[0]]
        n_snr, n_c50 = (snr[:12] + 10) / 50, (c50[:12] + 10) / 80
        want = (n_snr, n_c50, (n_snr + n_c50) / 2)[column]
        assert torch.allclose(quality[0], want, atol=1e-6)
    # errors of the reference
    import pytest
    with pytest.raises(ValueError):
        AudioBatchData(DB, window, seq_names, None, 9, device="cpu", signal_quality_path=tmp_path,
                       signal_quality_mode="loudness").getSignalQuality(0)
    with pytest.raises(FileNotFoundError):
        AudioBatchData(DB, window, seq_names, None, 9, device="cpu", signal_quality_path=tmp_path / "nowhere")


def test_pack_boundaries_follow_the_reference_rule(monkeypatch):
    """dataset.py:172-186's rule stated as the loop it is (a pack is closed by the first file that takes the running size beyond
    MAX_SIZE_LOADED; that file opens the next pack uncounted, totSize counts it with the pack it closed) against
    AudioBatchData.prepare, which computes the same boundaries from cumulative sizes, on random length lists."""
    import random as rnd
    from cpc2_amd import audio, dataset

    def rule(lengths, limit):
        packs, total, start, size = [], 0, 0, 0
        for index, length in enumerate(lengths):
            size += length
            if size > limit:
                packs.append([start, index])
                total += size
                start, size = index, 0
        if size > 0:
            packs.append([start, len(lengths)])
            total += size
        return packs, total

    gen = rnd.Random(7)
    for case in range(200):
        n = gen.randint(1, 40)
        lengths = [gen.randint(1, 50) for _ in range(n)]
        limit = gen.choice([1, 5, 30, 60, 100, 10 ** 6]) if case % 7 else max(lengths) - 1
        monkeypatch.setattr(audio, "info", lambda p, lengths=lengths: (16000, 1, lengths[int(str(p))]))
        obj = dataset.AudioBatchData.__new__(dataset.AudioBatchData)
        obj.keep_temporality, obj.MAX_SIZE_LOADED = False, limit
        obj.seqNames = [(0, str(i)) for i in range(n)]
        monkeypatch.setattr(dataset.random, "shuffle", lambda x: None)           # the order under test is the given one
        obj.prepare()
        assert (obj.packageIndex, obj.totSize) == rule(lengths, limit), (lengths, limit)


def test_shard_for_rank_partitions_like_the_reference():
    """cpc/train.py:389-393: rank r of ws loads files[len * r // ws : len * (r + 1) // ws] -- a partition in order, sizes within
    one of each other; on the reference's own fixture (9 files) with 2, 4 and 8 ranks, and each shard feeds an AudioBatchData."""
    import pytest
    seq_names, speakers = findAllSeqs(str(DB), extension=".flac")
    seq_names = sorted(seq_names)
    for ws in (1, 2, 4, 8, 9, 11):
        shards = [shard_for_rank(seq_names, r, ws) for r in range(ws)]
        assert [x for sh in shards for x in sh] == seq_names                       # a partition, in order
        sizes = [len(sh) for sh in shards]
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == len(seq_names)
        for r, sh in enumerate(shards):                                              # the reference's arithmetic, literally
            assert sh == seq_names[len(seq_names) * r // ws:len(seq_names) * (r + 1) // ws]
    assert shard_for_rank(seq_names, 0, 1) is not None and len(shard_for_rank(seq_names, 0, 1)) == 9
    with pytest.raises(ValueError):
        shard_for_rank(seq_names, 2, 2)
    # two ranks load disjoint audio whose windows add up to the whole fixture's
    total = 0
    for r in range(2):
        data = AudioBatchData(str(DB), 20480, shard_for_rank(seq_names, r, 2), None, len(speakers), nProcessLoader=1)
        total += len(data)
    whole = AudioBatchData(str(DB), 20480, seq_names, None, len(speakers), nProcessLoader=1)
    assert abs(total - len(whole)) <= 1                # (each shard drops its own ragged tail window)
