"""ORACLE -- TEST INFRASTRUCTURE ONLY.  float64 restatement of the VQT DEFINITION used by piano_a2s_amd/vqt.py.

Parity status: **UNPINNED**.  The reference's features come from librosa.vqt 0.10.1 (reference utilities.py:246-253), which is a
third-party package absent from /root/reference and from this image, and no reference spectrogram sample ships with the repository,
so there is nothing to pin this against.  The definition follows librosa's published parameterisation (frequencies
27.5*2^(k/60), alpha = (2^(2/60)-1)/(2^(2/60)+1), length = sr/(alpha*(f_k + gamma/alpha)), Hann window, norm=1, scale=True,
centred frames, amplitude_to_db(ref=max, amin=1e-5, top_db=80)/80+1); librosa's multi-rate evaluation is NOT reproduced."""
import numpy as np


def vqt_ref(y, sr=16000, hop=160, n_bins=480, bins_per_octave=60, gamma=20.0, fmin=27.5):
    y = np.asarray(y, dtype=np.float64)
    freqs = fmin * 2.0 ** (np.arange(n_bins) / bins_per_octave)
    r = 2.0 ** (2.0 / bins_per_octave)
    alpha = (r - 1.0) / (r + 1.0)
    lengths = sr / (alpha * (freqs + gamma / alpha))
    frames = 1 + len(y) // hop
    pad = int(np.ceil(lengths.max())) // 2 + 2
    yp = np.concatenate([np.zeros(pad), y, np.zeros(pad + hop)])
    Cq = np.zeros((frames, n_bins), dtype=np.complex128)
    for k in range(n_bins):
        L = lengths[k]
        n = np.arange(-int(L // 2), int(L // 2) + 1)
        n = n[np.abs(n) <= L / 2]
        kern = (0.5 + 0.5 * np.cos(2 * np.pi * n / L)) * np.exp(2j * np.pi * freqs[k] * n / sr)
        kern = kern / np.abs(kern).sum() / np.sqrt(L)
        for t in range(frames):
            c = pad + t * hop
            Cq[t, k] = np.dot(yp[c + n], np.conj(kern))
    mag = np.abs(Cq)
    db = 20 * np.log10(np.maximum(1e-5, mag)) - 20 * np.log10(max(1e-5, mag.max()))
    return np.maximum(db, -80.0) / 80.0 + 1.0
