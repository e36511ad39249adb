"""ASAP fine-tuning clips in the reference's tensor contract (reference datasets/asap.py:276-366: flat layout
``<feature_folder>/<split>/{spectrogram/<name>.npy, target/<name>.pkl}``).  The corpus preprocessing (ProcessASAP: cutting real
recordings by downbeat annotations with verovio / humextra) is offline data preparation and out of scope."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from datasets.syn import pad_measure, time_signature_list
from utilities import load


class ASAPDataset(Dataset):
    def __init__(self, hparams, split, device="cpu"):
        self.hp, self.split, self.device = hparams, split, device
        self.base = os.path.join(hparams["feature_folder"], split)
        self.names = sorted(f[:-4] for f in os.listdir(os.path.join(self.base, "spectrogram")))
        self.ts_index = {s: i for i, s in enumerate(time_signature_list())}

    def __len__(self):
        return len(self.names)

    def __getitem__(self, idx):
        name = self.names[idx]
        spec = torch.from_numpy(np.asarray(load(os.path.join(self.base, "spectrogram", name + ".npy")))).float()
        T = self.hp["max_frame_num"]
        padded = torch.zeros((T, spec.shape[-1]))
        n = min(spec.shape[0], T)
        padded[:n] = spec[:n]
        score = load(os.path.join(self.base, "target", name + ".pkl"))
        U, L = self.hp["max_length"]
        key = torch.tensor([bar[0] for bar in score]) + 6
        ts = torch.tensor([self.ts_index[bar[1]] for bar in score])
        upper = torch.stack([pad_measure(bar[3], U) for bar in score])
        lower = torch.stack([pad_measure(bar[2], L) for bar in score])
        up_len = torch.tensor([min(len(bar[3]), U) for bar in score])
        lo_len = torch.tensor([min(len(bar[2]), L) for bar in score])
        return padded.unsqueeze(0), ts, key, upper, up_len, lower, lo_len, name, 0
