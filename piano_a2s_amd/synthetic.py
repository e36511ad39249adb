"""Seeded synthetic clips in the reference's batch contract (host side).

The reference trains on rendered corpora that are not redistributable; benchmarks and parity tests
use random "scores" of the same tensor contract instead.  One batch is the 9-tuple the
reference's datasets yield (datasets/syn.py:88-121, datasets/asap.py:296-366):

  spectrogram (B,1,T,F) f32 in [0,1] | time_sig (B,bars) i64 | key (B,bars) i64 |
  upper (B,bars,U) i64 | upper_len (B,bars) i64 | lower (B,bars,L) i64 | lower_len (B,bars) i64 |
  names [B] | versions (B,) i64

Token rows follow ``pad_single_measure`` (datasets/syn.py:67-74): ``len`` note tokens, then <eos>
if the row is not full, then <pad>.
"""
import numpy as np
import torch

from .spec import EOS, PAD, SOS, VOCAB_SIZE

_NOTE_IDS = np.array([i for i in range(VOCAB_SIZE) if i not in (SOS, EOS, PAD)], dtype=np.int64)


def pad_measure(tokens, max_length):
    """Token list -> (max_length,) row: tokens | <eos> | <pad>...  (truncates to max_length)."""
    row = np.full((max_length,), PAD, dtype=np.int64)
    tokens = np.asarray(tokens, dtype=np.int64)[:max_length]
    row[:len(tokens)] = tokens
    if len(tokens) < max_length:
        row[len(tokens)] = EOS
    return row


def _ridge_spectrogram(rng, frames, bins):
    """VQT-like picture in [0,1]: low noise floor + a few sustained partials (note events with harmonics)."""
    s = 0.15 * rng.random((frames, bins), dtype=np.float32)
    for _ in range(int(rng.integers(6, 14))):
        f0 = int(rng.integers(0, max(1, bins - 1)))
        t0 = int(rng.integers(0, frames))
        t1 = min(frames, t0 + int(rng.integers(max(2, frames // 40), max(3, frames // 3))))
        amp = 0.5 + 0.5 * float(rng.random())
        decay = np.exp(-np.arange(t1 - t0, dtype=np.float32) / max(1.0, 0.5 * (t1 - t0)))
        for h, w in ((0, 1.0), (bins // 8, 0.6), (bins // 5, 0.4)):      # fundamental + two upper partials
            f = f0 + h
            if f < bins:
                s[t0:t1, f] = np.maximum(s[t0:t1, f], amp * w * decay)
                if f + 1 < bins:
                    s[t0:t1, f + 1] = np.maximum(s[t0:t1, f + 1], 0.5 * amp * w * decay)
    return np.clip(s, 0.0, 1.0)


def make_batch(batch, cfg, seed, frames=1201, upper_range=(20, 120), lower_range=(10, 80),
               full_tail=0.01, device="cpu", spectrogram="uniform", full_rows=()):
    """Deterministic batch.  Lengths ~ U{range}; with probability ``full_tail`` per (clip, bar, staff)
    the row is full-length with no <eos> (exercises the max-steps cap).  ``spectrogram``: "uniform"
    (U[0,1) noise) or "ridges" (clip-specific sustained partials on a noise floor, so that
    different clips encode differently -- used by parity fixtures to catch batch-indexing errors).  ``full_rows``: (clip, bar, "up" | "lo")
    rows that are full-length whatever the draw says (parity fixtures place the max-steps cap where they want it)."""
    forced = {(int(b), int(k), str(s)) for b, k, s in full_rows}
    rng = np.random.default_rng(seed)
    bars = cfg["max_bars"]
    U, L = cfg["max_length"]
    if spectrogram == "ridges":
        spec = np.stack([_ridge_spectrogram(rng, frames, cfg["freq_bins"]) for _ in range(batch)])[:, None]
    else:
        spec = rng.random((batch, 1, frames, cfg["freq_bins"]), dtype=np.float32)
    ts = np.empty((batch, bars), dtype=np.int64)
    key = np.empty((batch, bars), dtype=np.int64)
    for b in range(batch):
        ts[b] = rng.integers(0, cfg["num_time_sig"])
        key[b] = rng.integers(0, cfg["num_keys"])
        for k in range(1, bars):                      # constant within the clip w.p. 0.9
            if rng.random() > 0.9:
                ts[b, k:] = rng.integers(0, cfg["num_time_sig"])
            if rng.random() > 0.9:
                key[b, k:] = rng.integers(0, cfg["num_keys"])

    def staff(maxlen, lo, hi, which):
        rows = np.empty((batch, bars, maxlen), dtype=np.int64)
        lens = np.empty((batch, bars), dtype=np.int64)
        for b in range(batch):
            for k in range(bars):
                n = maxlen if rng.random() < full_tail else int(rng.integers(min(lo, maxlen), min(hi, maxlen) + 1))
                if (b, k, which) in forced:
                    n = maxlen
                toks = _NOTE_IDS[rng.integers(0, len(_NOTE_IDS), size=n)]
                rows[b, k] = pad_measure(toks, maxlen)
                lens[b, k] = min(n, maxlen)
        return rows, lens

    up, up_len = staff(U, *upper_range, "up")
    lo, lo_len = staff(L, *lower_range, "lo")
    t = lambda a: torch.from_numpy(a).to(device)
    names = [f"syn{seed}_{b}~synthetic" for b in range(batch)]
    return (t(spec), t(ts), t(key), t(up), t(up_len), t(lo), t(lo_len), names,
            torch.zeros(batch, dtype=torch.long))


def make_waveforms(batch, seed, seconds=12.0, sr=16000, device="cpu"):
    """Synthetic 16 kHz clips for the online VQT front-end: up to 6 simultaneous decaying harmonic tones at MIDI 21-108,
    peak-normalised to 0.9.  Returns (batch, seconds*sr) float32."""
    rng = np.random.default_rng(seed)
    n = int(seconds * sr)
    t = np.arange(n, dtype=np.float64) / sr
    out = np.zeros((batch, n), dtype=np.float64)
    for b in range(batch):
        for _ in range(int(rng.integers(12, 40))):
            midi = int(rng.integers(21, 109))
            f0 = 440.0 * 2.0 ** ((midi - 69) / 12.0)
            t0 = float(rng.uniform(0, seconds * 0.95))
            dur = float(rng.uniform(0.15, 2.0))
            env = np.where(t >= t0, np.exp(-(t - t0) / (0.35 * dur)), 0.0) * (t < t0 + 2.5 * dur)
            for h, a in ((1, 1.0), (2, 0.5), (3, 0.25), (4, 0.12)):
                if f0 * h < sr / 2:
                    out[b] += a * env * np.sin(2 * np.pi * f0 * h * (t - t0))
        out[b] *= 0.9 / max(np.abs(out[b]).max(), 1e-9)
    return torch.from_numpy(out.astype(np.float32)).to(device)
