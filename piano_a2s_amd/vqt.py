"""Variable-Q transform front-end on the MI355X.

The reference computes its input features OFFLINE with ``librosa.vqt(y, sr=16000, hop_length=160, fmin=A0, n_bins=480,
bins_per_octave=60, gamma=20)`` -> ``amplitude_to_db(|.|, ref=max)/80 + 1`` (reference utilities.py:240-254; librosa 0.10.1 is a
third-party dependency that is neither under /root/reference nor installed here).  This module evaluates the same transform
DEFINITION directly -- one Hann-windowed complex exponential per bin, filter length Q*sr/(f_k + gamma/alpha), L1-normalised,
scaled by 1/sqrt(length), frames centred every `hop` samples on the zero-padded signal -- as ONE framed complex GEMM on the matrix
cores plus a log-magnitude epilogue.  librosa instead evaluates octave by octave on recursively decimated signals with a sparsified
FFT basis; the two agree up to its resampling/sparsity error, so this front-end is "librosa-0.10.1-like", NOT bit-comparable:
**parity unpinned** (no librosa, no reference .npy sample).  It is validated against its own float64 restatement
(oracle/vqt_ref.py) and analytic known answers (tests/test_gpu_vqt.py).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import hip

A0_HZ = 27.5


def filter_lengths(sr=16000, n_bins=480, bins_per_octave=60, gamma=20.0, fmin=A0_HZ):
    freqs = fmin * 2.0 ** (np.arange(n_bins) / bins_per_octave)
    r = 2.0 ** (2.0 / bins_per_octave)
    alpha = (r - 1.0) / (r + 1.0)                       # relative bandwidth of one bin
    Q = 1.0 / alpha
    lengths = Q * sr / (freqs + gamma / alpha)
    return freqs, lengths


def kernel_bank(sr=16000, n_bins=480, bins_per_octave=60, gamma=20.0, fmin=A0_HZ):
    """(taps, 2*n_bins) float32: column k = real part, column n_bins+k = MINUS imaginary part of bin k's conjugated kernel, centred."""
    freqs, lengths = filter_lengths(sr, n_bins, bins_per_octave, gamma, fmin)
    taps = int(math.ceil(lengths.max()))
    taps += (-taps) % 4                                  # multiple of 4 keeps the framed rows 16-byte aligned
    bank = np.zeros((taps, 2 * n_bins), dtype=np.float64)
    centre = taps // 2
    for k in range(n_bins):
        L = lengths[k]
        n = np.arange(-int(L // 2), int(L // 2) + 1)     # odd support centred on the frame
        n = n[np.abs(n) <= L / 2]
        win = 0.5 + 0.5 * np.cos(2.0 * np.pi * n / L)    # Hann window of (real-valued) length L
        kern = win * np.exp(2j * np.pi * freqs[k] * n / sr)
        kern = kern / np.abs(kern).sum()                 # norm=1
        kern = kern / np.sqrt(L)                         # scale=True
        idx = centre + n
        bank[idx, k] = kern.real                         # C = sum_m y[m] * conj(kern[m])
        bank[idx, n_bins + k] = -kern.imag
    return bank.astype(np.float32), taps


class VQT:
    def __init__(self, device, sr=16000, hop=160, n_bins=480, bins_per_octave=60, gamma=20.0):
        bank, self.taps = kernel_bank(sr, n_bins, bins_per_octave, gamma)
        self.bank = torch.from_numpy(bank).to(device)
        self.hop, self.n_bins, self.device = hop, n_bins, device

    def __call__(self, wave):
        """wave: (B, N) float32 on the device (16 kHz).  Returns (B, 1, 1 + N//hop, n_bins) in [0, 1] -- the model's input."""
        if not wave.is_cuda:
            raise hip.A2SError("VQT runs on the GPU only (no CPU implementation in the product)")
        B, N = wave.shape
        frames = 1 + N // self.hop
        half = self.taps // 2
        plen = (frames - 1) * self.hop + self.taps
        plen += (-plen) % 4
        padded = torch.zeros((B, plen), dtype=torch.float32, device=wave.device)
        padded[:, half:half + N] = wave                  # centre=True with zero padding
        Cc = torch.empty((B, frames, 2 * self.n_bins), dtype=torch.float32, device=wave.device)
        # framed complex GEMM: A(n, m) = padded[n*hop + m]  (row stride = hop), B = bank (taps, 2*bins)
        hip.gemm(padded, self.hop, 1, self.bank, 2 * self.n_bins, 1, Cc, 2 * self.n_bins, frames, 2 * self.n_bins, self.taps,
                 batch=B, bsA=plen, bsB=0, bsC=frames * 2 * self.n_bins)
        out = torch.empty((B, 1, frames, self.n_bins), dtype=torch.float32, device=wave.device)
        partial = torch.empty(B * 64, dtype=torch.float32, device=wave.device)
        hip.check(hip.lib().a2s_vqt_logmag(hip.stream(), hip._p(Cc), hip._p(out), hip._p(partial), B, C.c_long(frames), self.n_bins, hip.f32(80.0)),
                  "a2s_vqt_logmag")
        return out
