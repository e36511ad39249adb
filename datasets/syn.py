"""Clip datasets in the reference's tensor contract (reference datasets/syn.py:10-170).

``TrainDataset`` / ``TestDataset`` read the rendered-corpus layout
``<feature_folder>/<split>/<version>/spectrogram/<chunk>~<soundfont>.npy`` + ``.../target/<chunk>.pkl`` (a list of
``[key:int, time_sig:str, lower_ids, upper_ids]`` per bar; index 2 = lower, 3 = upper) and return the 9-tuple
(spectrogram (1,T,F) f32, time_sig (bars,), key (bars,), upper (bars,U), upper_len (bars,), lower (bars,L), lower_len (bars,),
name, version).  Tensors are built on the host and moved by the trainer one batch at a time (the reference moves every item).
``SyntheticClips`` yields seeded random clips of the same contract without any files (benchmarks, smoke runs).
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from piano_a2s_amd import synthetic
from piano_a2s_amd.spec import EOS, PAD
from utilities import load

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def time_signature_list():
    return load(os.path.join(_HERE, "data_processing", "metadata", "time_signature_list.json"))


def pad_measure(tokens, max_length):
    return torch.from_numpy(synthetic.pad_measure(tokens, max_length))


def read_clip(folder, spectrogram_name, target_name, hparams, ts_index):
    """One clip of the rendered-corpus / ASAP file layout -> the first seven elements of the reference's item tuple
    (reference syn.py:93-111 == asap.py:299-312): ``<folder>/spectrogram/<spectrogram_name>.npy`` (frames, bins) zero-padded /
    cut to max_frame_num -> (1, T, bins) f32; ``<folder>/target/<target_name>.pkl`` = one [key, time_sig, LOWER ids, UPPER ids]
    per bar -> key + 6, time-signature class index, rows padded with pad_measure, lengths clipped to max_length."""
    spec = torch.from_numpy(np.asarray(load(os.path.join(folder, "spectrogram", spectrogram_name + ".npy")))).float()
    T = hparams["max_frame_num"]
    padded = torch.zeros((T, spec.shape[-1]))
    n = min(spec.shape[0], T)
    padded[:n] = spec[:n]
    score = load(os.path.join(folder, "target", target_name + ".pkl"))
    U, L = hparams["max_length"]
    key = torch.tensor([int(bar[0]) for bar in score]) + 6
    ts = torch.tensor([ts_index[bar[1]] for bar in score])
    upper = torch.stack([pad_measure(bar[3], U) for bar in score])
    lower = torch.stack([pad_measure(bar[2], L) for bar in score])
    up_len = torch.tensor([min(len(bar[3]), U) for bar in score])
    lo_len = torch.tensor([min(len(bar[2]), L) for bar in score])
    return padded.unsqueeze(0), ts, key, upper, up_len, lower, lo_len


class _ClipFolder(Dataset):
    def __init__(self, hparams, split, device="cpu", version=(0,)):
        self.hp, self.split, self.device, self.version = hparams, split, device, list(version)
        self.ts_index = {s: i for i, s in enumerate(time_signature_list())}
        self.songs = {}
        for v in self.version:
            folder = os.path.join(hparams["feature_folder"], split, str(v), "spectrogram")
            self.songs[v] = sorted(f[:-4] for f in os.listdir(folder))

    def _item(self, v, name):
        """name = ``<chunk>~<soundfont>``: every soundfont rendering of a chunk shares the chunk's target (syn.py:97)."""
        base = os.path.join(self.hp["feature_folder"], self.split, str(v))
        return read_clip(base, name, name.split("~")[0], self.hp, self.ts_index) + (name, v)


class TrainDataset(_ClipFolder):
    """One random rendering version per access (reference syn.py:88-92)."""

    def __len__(self):
        return max(len(s) for s in self.songs.values())

    def __getitem__(self, idx):
        v = self.version[np.random.randint(len(self.version))]
        names = self.songs[v]
        return self._item(v, names[idx % len(names)])


class TestDataset(_ClipFolder):
    def __init__(self, hparams, split, device="cpu", version=(0,)):
        super().__init__(hparams, split, device, version)
        self.flat = [(n, v) for v in self.version for n in self.songs[v]]

    def __len__(self):
        return len(self.flat)

    def __getitem__(self, idx):
        n, v = self.flat[idx]
        return self._item(v, n)


class SyntheticClips(Dataset):
    """`n` seeded synthetic clips (piano_a2s_amd.synthetic) -- no corpus needed."""

    def __init__(self, cfg, n, seed=1234, frames=1201, **kw):
        self.cfg, self.n, self.seed, self.frames, self.kw = cfg, n, seed, frames, kw

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        b = synthetic.make_batch(1, self.cfg, self.seed + idx, frames=self.frames, **self.kw)
        return tuple(t[0] for t in b[:7]) + (b[7][0], int(b[8][0]))


class SyntheticWaveClips(SyntheticClips):
    """As SyntheticClips but the first element is a raw 16 kHz waveform (N,) -- exercises the online VQT front-end."""

    def __getitem__(self, idx):
        item = super().__getitem__(idx)
        wave = synthetic.make_waveforms(1, self.seed + idx, seconds=(self.frames - 1) / 100.0)[0]
        return (wave,) + item[1:]
