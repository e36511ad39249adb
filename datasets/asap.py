"""ASAP fine-tuning clips in the reference's tensor contract (reference datasets/asap.py:276-366: flat layout
``<feature_folder>/<split>/{spectrogram/<name>.npy, target/<name>.pkl}``).  The corpus preprocessing (ProcessASAP: cutting real
recordings by downbeat annotations with verovio / humextra) is offline data preparation and out of scope."""
import os

from torch.utils.data import Dataset

from datasets.syn import read_clip, time_signature_list


class ASAPDataset(Dataset):
    def __init__(self, hparams, split, device="cpu"):
        self.hp, self.split, self.device = hparams, split, device
        self.base = os.path.join(hparams["feature_folder"], split)
        self.names = sorted(f[:-4] for f in os.listdir(os.path.join(self.base, "spectrogram")))
        self.ts_index = {s: i for i, s in enumerate(time_signature_list())}

    def __len__(self):
        return len(self.names)

    def __getitem__(self, idx):
        name = self.names[idx]                       # spectrogram and target share the clip name; the version slot says 'asap' (asap.py:314-323)
        return read_clip(self.base, name, name, self.hp, self.ts_index) + (name, "asap")
