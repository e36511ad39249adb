"""TEST INFRASTRUCTURE: writes small on-disk corpora in the two file layouts the reference's datasets read
(rendered corpus: reference datasets/syn.py:28-36,88-121, written by data_processing/render.py:227; ASAP: datasets/asap.py:296-328)."""
import os
import pickle

import numpy as np

from piano_a2s_amd import synthetic
from piano_a2s_amd.spec import EOS

TIME_SIGS = ["4/4", "3/4", "2/4", "6/8", "2/2", "12/8", "3/8"]           # data_processing/metadata/time_signature_list.json


def _score(batch, b, cfg):
    """Target record of clip b: one [key (-6..7), time-signature string, LOWER ids, UPPER ids] per bar."""
    _, ts, key, up, up_len, lo, lo_len = batch[:7]
    return [[int(key[b, k]) - 6, TIME_SIGS[int(ts[b, k])], lo[b, k, :int(lo_len[b, k])].tolist(), up[b, k, :int(up_len[b, k])].tolist()]
            for k in range(cfg["max_bars"])]


def write_rendered_corpus(root, cfg, split, versions, n_chunks, frames, seed, soundfonts=("pianoA", "pianoB"), **kw):
    """<root>/<split>/<version>/{spectrogram/<chunk>~<soundfont>.npy, target/<chunk>.pkl}; returns {version: {name: (spec, score)}}."""
    out = {}
    for v in versions:
        base = os.path.join(root, split, str(v))
        os.makedirs(os.path.join(base, "spectrogram"), exist_ok=True)
        os.makedirs(os.path.join(base, "target"), exist_ok=True)
        batch = synthetic.make_batch(n_chunks, cfg, seed + 100 * v, frames=frames, **kw)
        out[v] = {}
        for b in range(n_chunks):
            chunk = f"Chunk{b:02d}"                   # upper-case initial -> style "pop" in the result record (pretrain.py:205)
            score = _score(batch, b, cfg)
            with open(os.path.join(base, "target", chunk + ".pkl"), "wb") as f:
                pickle.dump(score, f)
            for j, sf in enumerate(soundfonts):
                spec = batch[0][b, 0].numpy() * (1.0 - 0.25 * j)       # each soundfont rendering: its own spectrogram, shared target
                n = frames - 3 * ((b + j) % 2)                          # some clips shorter than max_frame_num (zero-padded by the reader)
                np.save(os.path.join(base, "spectrogram", f"{chunk}~{sf}.npy"), spec[:n])
                out[v][f"{chunk}~{sf}"] = (spec[:n], score)
    return out


def write_asap_corpus(root, cfg, split, n_clips, frames, seed, **kw):
    """<root>/<split>/{spectrogram/<name>.npy, target/<name>.pkl}; returns {name: (spec, score)}."""
    base = os.path.join(root, split)
    os.makedirs(os.path.join(base, "spectrogram"), exist_ok=True)
    os.makedirs(os.path.join(base, "target"), exist_ok=True)
    batch = synthetic.make_batch(n_clips, cfg, seed, frames=frames, **kw)
    out = {}
    for b in range(n_clips):
        name = f"Bach_Fugue_bwv_{846 + b}_perf{b}_{b * 5}"
        score = _score(batch, b, cfg)
        score = [[str(bar[0])] + bar[1:] for bar in score] if b % 2 else score       # ASAP targets may carry the key as a string (asap.py:309 int())
        spec = batch[0][b, 0].numpy()[: frames - (b % 3)]
        np.save(os.path.join(base, "spectrogram", name + ".npy"), spec)
        with open(os.path.join(base, "target", name + ".pkl"), "wb") as f:
            pickle.dump(score, f)
        out[name] = (spec, score)
    return out
