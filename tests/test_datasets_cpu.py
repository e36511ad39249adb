"""On-disk dataset readers against the contract spelled out by the reference's datasets
(reference datasets/syn.py:46-74,88-121,124-170; datasets/asap.py:296-328), on small corpora written to a tmp folder."""
import numpy as np
import pytest
import torch

from piano_a2s_amd import spec as a2s_spec
from piano_a2s_amd.spec import EOS, PAD
from tests import disk_corpus

CFG = a2s_spec.default_cfg(freq_bins=24, conv_feature_size=32, hidden_size=32, max_length=(12, 8))
HP = {"max_frame_num": 41, "max_length": (12, 8)}


def _expect_rows(tokens, maxlen):
    """reference pad_single_measure (syn.py:67-74): truncate, <pad>-fill, <eos> right after the tokens when there is room."""
    row = [PAD] * maxlen
    tokens = list(tokens)[:maxlen]
    row[:len(tokens)] = tokens
    if len(tokens) < maxlen:
        row[len(tokens)] = EOS
    return row


def _check_item(item, spec_np, score, name, version):
    spectrogram, ts, key, up, up_len, lo, lo_len, got_name, got_version = item
    assert spectrogram.shape == (1, 41, 24) and spectrogram.dtype == torch.float32
    n = spec_np.shape[0]
    assert torch.equal(spectrogram[0, :n], torch.from_numpy(spec_np).float()) and float(spectrogram[0, n:].abs().sum()) == 0.0
    assert key.tolist() == [int(bar[0]) + 6 for bar in score] and key.dtype == torch.int64                      # sharps/flats + 6
    assert ts.tolist() == [disk_corpus.TIME_SIGS.index(bar[1]) for bar in score] and ts.dtype == torch.int64
    assert up.tolist() == [_expect_rows(bar[3], 12) for bar in score]                                             # index 3 = UPPER staff
    assert lo.tolist() == [_expect_rows(bar[2], 8) for bar in score]                                              # index 2 = LOWER staff
    assert up_len.tolist() == [min(len(bar[3]), 12) for bar in score] and lo_len.tolist() == [min(len(bar[2]), 8) for bar in score]
    assert got_name == name and got_version == version


def test_rendered_corpus_readers(tmp_path):
    from datasets.syn import TestDataset, TrainDataset
    hp = dict(HP, feature_folder=str(tmp_path))
    # a 10 % tail of full-length rows: truncation to max_length and the "no <eos> when the row is full" case are exercised
    written = disk_corpus.write_rendered_corpus(str(tmp_path), CFG, "train", range(3), 4, 41, seed=5, upper_range=(3, 14), lower_range=(2, 10), full_tail=0.1)
    train = TrainDataset(hp, "train", "cpu", range(3))
    assert len(train) == 8                                             # 4 chunks x 2 soundfonts per version (max over versions, syn.py:86)
    # one random rendering version per access, drawn with np.random.randint(len(version)) exactly as the reference does (syn.py:90):
    # seeding numpy reproduces the reference's version sequence
    np.random.seed(77)
    got = [train[i] for i in range(11)]
    np.random.seed(77)
    for i, item in enumerate(got):
        v = list(range(3))[np.random.randint(3)]
        names = sorted(written[v])
        name = names[i % len(names)]                                  # idx % length (syn.py:93)
        _check_item(item, *written[v][name], name, v)                 # target looked up by name.split('~')[0] (syn.py:96)
    assert any("~pianoB" in item[7] for item in got) and len({item[8] for item in got}) > 1
    test = TestDataset(hp, "train", "cpu", [0, 2])
    assert len(test) == 16                                            # every (version, rendering) once (syn.py:133-137)
    seen = set()
    for i in range(len(test)):
        item = test[i]
        _check_item(item, *written[item[8]][item[7]], item[7], item[8])
        seen.add((item[7], item[8]))
    assert seen == {(n, v) for v in (0, 2) for n in written[v]}


def test_asap_reader_and_collation(tmp_path):
    from datasets.asap import ASAPDataset
    hp = dict(HP, feature_folder=str(tmp_path))
    written = disk_corpus.write_asap_corpus(str(tmp_path), CFG, "test", 5, 41, seed=9, upper_range=(3, 14), lower_range=(2, 10), full_tail=0.1)
    ds = ASAPDataset(hp, "test", "cpu")
    assert len(ds) == 5
    for i in range(5):
        item = ds[i]
        _check_item(item, *written[item[7]], item[7], "asap")           # flat layout, same name for both files, version slot = 'asap' (asap.py:314-323)
    batch = next(iter(torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False)))
    assert batch[0].shape == (3, 1, 41, 24) and batch[3].shape == (3, 5, 12) and batch[5].shape == (3, 5, 8) and list(batch[8]) == ["asap"] * 3
