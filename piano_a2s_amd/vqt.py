"""Variable-Q transform front-end on the MI355X (the north-star's "batched CQT").

The reference computes its input features OFFLINE with ``librosa.vqt(y, sr=16000, hop_length=160, fmin=A0, n_bins=480,
bins_per_octave=60, gamma=20)`` -> ``amplitude_to_db(|.|, ref=max)/80 + 1`` (reference utilities.py:240-254; librosa 0.10.1 is a
third-party dependency that is neither under /root/reference nor installed here).  This module evaluates librosa's ALGORITHM for that
call on the GPU, as framed GEMMs on the matrix cores:

  * octave by octave, highest first, on a signal decimated by 2 after every octave while the hop stays even (hop 160 -> 5; the three
    lowest octaves share sr = 500 Hz): 8 framed complex GEMMs  C_o = frames(y_o) x bank_o  with frames(y)[n, m] = y[n*hop_o + m]
    (a Hankel view expressed through the GEMM's row stride -- the frames are never materialised) and 32..512 taps instead of the
    788-tap full-rate bank a direct evaluation needs (13x fewer flops);
  * bank_o is librosa's per-octave filter bank in the form the GEMM consumes: Hann-windowed complex exponentials of length
    Q sr_o / (f_k + gamma/alpha) (L1-normalised, scaled by length/n_fft), FFT'd, the non-negative half sparsified to 99 % of each
    row's magnitude, scaled by sqrt(sr/sr_o) and 1/sqrt(length at the full rate) -- and then transformed BACK to the time domain
    (response = sum_f basis[f] FFT(frame)[f] = frame . g with g[m] = sum_f basis[f] e^{-2 pi i f m / n_fft}), so that the sparsified
    basis, the dropped negative frequencies and the rectangular-window STFT of librosa are all inside one dense real x complex product;
  * the decimator is a framed GEMM as well (row stride 2, one output column): a linear-phase Kaiser-windowed sinc with soxr-HQ's
    documented band edges.  librosa decimates with libsoxr (res_type='soxr_hq'), a compiled resampler whose coefficients are not
    published: this is the one step that cannot follow librosa exactly, and why the front-end's parity stays **unpinned**;
  * log-magnitude epilogue (csrc/a2s_vqt.hip): dB relative to the clip maximum, floor 1e-5, top_db 80, /80 + 1.

oracle/vqt_ref.py restates the same algorithm on the CPU in float64 with FFTs (librosa's own formulation); tests compare the two and
report how far the round-1 direct-form definition was from it (it used the wrong channel scaling: 1/sqrt(length) without librosa's
length/n_fft factor, a 14 dB tilt across the 8 octaves).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import hip

A0_HZ = 27.5
_HANN_BW = 1.50018310546875          # equivalent noise bandwidth of the Hann window in bins (librosa.filters.window_bandwidth)


def _grid(n_bins, bins_per_octave, fmin):
    freqs = fmin * 2.0 ** (np.arange(n_bins) / bins_per_octave)
    logf = np.log2(freqs)
    bpo = np.empty_like(freqs)                                  # local bins per octave: centred differences, reflected at the ends
    bpo[0], bpo[-1] = 1.0 / (logf[1] - logf[0]), 1.0 / (logf[-1] - logf[-2])
    bpo[1:-1] = 2.0 / (logf[2:] - logf[:-2])
    r = 2.0 ** (2.0 / bpo)
    return freqs, (r - 1.0) / (r + 1.0)                        # centre frequencies, relative bandwidths alpha_k


def filter_lengths(sr=16000, n_bins=480, bins_per_octave=60, gamma=20.0, fmin=A0_HZ):
    freqs, alpha = _grid(n_bins, bins_per_octave, fmin)
    return freqs, sr / (alpha * (freqs + gamma / alpha))       # Q sr / (f + gamma/alpha), Q = 1/alpha


def decimation_filter(passband=0.913, atten_db=120.0):
    """Half-band low-pass of the decimator, scaled by sqrt(2) (librosa.resample(..., scale=True)): taps h[-half..half]."""
    width = (1.0 - passband) * 0.25
    cutoff = (1.0 + passband) * 0.125
    half = int(math.ceil((atten_db - 8.0) / (2.285 * 2.0 * math.pi * width) / 2.0))
    n = np.arange(-half, half + 1)
    h = 2.0 * cutoff * np.sinc(2.0 * cutoff * n) * np.kaiser(2 * half + 1, 0.1102 * (atten_db - 8.7))
    return h / h.sum() * math.sqrt(2.0), half


def octave_banks(sr=16000, hop=160, n_bins=480, bins_per_octave=60, gamma=20.0, fmin=A0_HZ, sparsity=0.01):
    """Per octave (highest first): dict(hop, n_fft, bins = slice into the 480 bins, bank (n_fft, 2*nb) float32 = [Re g | Im g])."""
    freqs, alpha = _grid(n_bins, bins_per_octave, fmin)
    full_len = sr / (alpha * (freqs + gamma / alpha))
    cutoff = np.max(freqs * (1.0 + 0.5 * _HANN_BW * alpha) + 0.5 * gamma)
    if cutoff > sr / 2.0:
        raise ValueError("VQT filter bank reaches beyond the Nyquist frequency")
    n_oct = int(math.ceil(n_bins / bins_per_octave))
    nf = min(bins_per_octave, n_bins)
    out = []
    my_sr, my_hop = float(sr), hop
    for o in range(n_oct):
        hi = n_bins - nf * o
        lo = max(0, hi - nf)
        f, a = freqs[lo:hi], alpha[lo:hi]
        lens = my_sr / (a * (f + gamma / a))
        n_fft = int(2 ** math.ceil(math.log2(lens.max())))
        basis = np.zeros((hi - lo, n_fft), dtype=np.complex128)
        for i, (ilen, fk) in enumerate(zip(lens, f)):
            n = np.arange(-ilen // 2, ilen // 2, dtype=float)
            m = len(n)
            sig = np.exp(2j * np.pi * fk * n / my_sr) * (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(m) / m))     # periodic Hann
            sig *= ilen / n_fft / np.abs(sig).sum()
            p = (n_fft - m) // 2
            basis[i, p:p + m] = sig
        F = np.fft.fft(basis, axis=1)[:, : n_fft // 2 + 1]
        mags = np.abs(F)                                                     # keep the entries holding 99 % of each row's magnitude
        srt = np.sort(mags, axis=1)
        cum = np.cumsum(srt / mags.sum(axis=1, keepdims=True), axis=1)
        thr = srt[np.arange(len(F)), np.argmin(cum < sparsity, axis=1)]
        F = np.where(mags >= thr[:, None], F, 0.0)
        F = F * math.sqrt(sr / my_sr) / np.sqrt(full_len[lo:hi])[:, None]
        # back to the time domain: response = sum_f F[f] * rfft(frame)[f] = frame . g,  g[m] = sum_f F[f] exp(-2 pi i f m / n_fft)
        ph = np.exp(-2j * np.pi * np.outer(np.arange(n_fft // 2 + 1), np.arange(n_fft)) / n_fft)
        g = (F @ ph).T                                                        # (n_fft, nb)
        out.append(dict(hop=my_hop, n_fft=n_fft, lo=lo, hi=hi, bank=np.ascontiguousarray(np.concatenate([g.real, g.imag], axis=1), dtype=np.float32)))       # row-major (n_fft, 2 nb)
        if my_hop % 2 == 0:
            my_hop //= 2
            my_sr /= 2.0
            out[-1]["decimate_after"] = True
    return out


class VQT:
    def __init__(self, device, sr=16000, hop=160, n_bins=480, bins_per_octave=60, gamma=20.0):
        self.octaves = octave_banks(sr, hop, n_bins, bins_per_octave, gamma)
        for o in self.octaves:
            o["bank_dev"] = torch.from_numpy(o["bank"]).to(device)
        h, self.half = decimation_filter()
        taps = 2 * self.half + 1
        self.dec_taps = taps + (-taps) % 4
        hb = np.zeros((self.dec_taps, 1), dtype=np.float32)
        hb[:taps, 0] = h[::-1]                                                # y_out[m] = sum_j ypad[2m + j] h[half - j]  (h is symmetric)
        self.dec_bank = torch.from_numpy(hb).to(device)
        self.hop, self.n_bins, self.device = hop, n_bins, device
        self.bpo = min(bins_per_octave, n_bins)
        if any(o["hi"] - o["lo"] != self.bpo for o in self.octaves):
            raise ValueError("VQT: the octave-by-octave response layout needs n_bins to be a multiple of bins_per_octave")

    def _decimate(self, y):
        B, N = y.shape
        n_out = (N + 1) // 2
        plen = 2 * n_out + self.dec_taps + 4
        plen += (-plen) % 4
        yp = torch.zeros((B, plen), dtype=torch.float32, device=y.device)
        yp[:, self.half:self.half + N] = y
        out = torch.empty((B, n_out), dtype=torch.float32, device=y.device)
        hip.check(hip.lib().a2s_vqt_decimate(hip.stream(), hip._p(yp), C.c_long(plen), hip._p(self.dec_bank), self.dec_taps, hip._p(out), C.c_long(n_out), B),
                  "a2s_vqt_decimate")
        return out

    def __call__(self, wave):
        """wave: (B, N) float32 on the device (16 kHz).  Returns (B, 1, 1 + N//hop, n_bins) in [0, 1] -- the model's input."""
        if not wave.is_cuda:
            raise hip.A2SError("VQT runs on the GPU only (no CPU implementation in the product)")
        B, N = wave.shape
        y = wave.contiguous().float()
        sigs = []
        for o in self.octaves:                                                # the decimation chain first: frame counts per octave
            sigs.append(y)
            if o.get("decimate_after") and o is not self.octaves[-1]:
                y = self._decimate(y)
        frames = min(1 + s.shape[1] // o["hop"] for s, o in zip(sigs, self.octaves))
        nb2 = 2 * self.n_bins
        Cc = torch.empty((B, frames, nb2), dtype=torch.float32, device=wave.device)
        col = 0
        for s, o in zip(sigs, self.octaves):
            n_fft, hop, nb = o["n_fft"], o["hop"], o["hi"] - o["lo"]
            plen = (frames - 1) * hop + n_fft
            plen += (-plen) % 4
            plen = max(plen, n_fft // 2 + s.shape[1])
            plen += (-plen) % 4
            padded = torch.zeros((B, plen), dtype=torch.float32, device=wave.device)
            padded[:, n_fft // 2:n_fft // 2 + s.shape[1]] = s              # centre=True with zero padding
            # framed complex GEMM: A(n, m) = padded[n*hop + m] (row stride = hop), B = [Re g | Im g] (n_fft, 2 nb): ONE product per octave writes
            # [re | im] of its bins side by side into the octave's 2 nb columns of the (frames, n_oct * 2 nb) response the epilogue reads
            hip.gemm(padded, hop, 1, o["bank_dev"], 2 * nb, 1, Cc, nb2, frames, 2 * nb, n_fft, batch=B, bsA=plen, bsB=0, bsC=frames * nb2, c_off=col)
            col += 2 * nb
        out = torch.empty((B, 1, frames, self.n_bins), dtype=torch.float32, device=wave.device)
        partial = torch.empty(B * 64, dtype=torch.float32, device=wave.device)
        hip.check(hip.lib().a2s_vqt_logmag_octaves(hip.stream(), hip._p(Cc), hip._p(out), hip._p(partial), B, C.c_long(frames), self.n_bins, self.bpo,
                                                   hip.f32(80.0)), "a2s_vqt_logmag_octaves")
        return out
