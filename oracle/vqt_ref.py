"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatements of the VQT front-end.

The reference computes its input features with ``librosa.vqt(y, sr=16000, hop_length=160, fmin=A0, n_bins=480, bins_per_octave=60,
gamma=20)`` -> ``amplitude_to_db(|.|, ref=np.max) / 80 + 1`` (reference utilities.py:240-254).  librosa 0.10.1 (environment.yaml:52) is a
third-party package that is neither under /root/reference nor installed in this image; the repository ships no spectrogram sample and
the reference has no test for this path.

Two functions:

``vqt_librosa``  restates librosa 0.10.1's PUBLISHED ALGORITHM for exactly that call (librosa/core/constantq.py ``vqt``; helper names
    below are librosa's): per-octave evaluation, highest octave first, on a signal that is decimated by 2 after every octave while the
    hop length stays even (hop 160 -> 5: five decimations, the three lowest octaves share sr = 500 Hz); per octave a bank of
    Hann-windowed complex exponentials of length Q sr' / (f_k + gamma/alpha) (``filters.wavelet``, ``wavelet_lengths``,
    ``_relative_bandwidth``), L1-normalised, zero-padded to the next power of two, FFT'd, its non-negative half SPARSIFIED to the
    entries that hold 99 % of each row's magnitude (``util.sparsify_rows``, sparsity = 0.01), applied to a rectangular-window STFT of
    the centred, zero-padded signal (``__cqt_response``); octave stack trimmed (``__trim_stack``); ``scale=True`` division by
    sqrt(filter length at the original rate); then ``amplitude_to_db(ref=max, amin=1e-5, top_db=80) / 80 + 1``.
    ONE step cannot be restated from documentation: the decimator.  librosa resamples with ``res_type='soxr_hq'`` -- libsoxr, a
    compiled library (python-soxr 0.3.7, environment.yaml:104) whose filter coefficients are not published.  ``_decimate2`` stands in
    for it: a linear-phase Kaiser-windowed sinc with soxr HQ's documented band edges (pass band to 0.913 of the new Nyquist, >= 120 dB
    rejection), followed by librosa's ``scale=True`` factor sqrt(2).  Every filter of every octave lies inside that pass band (top bin
    of an octave: 0.87 of the Nyquist), so the result is insensitive to the stand-in's exact design -- tests/test_vqt_oracle.py measures
    that sensitivity (two different designs agree to < 2e-4 of the [0, 1] output range) instead of assuming it.

``vqt_direct``   the direct-form DEFINITION the HIP front-end evaluates (piano_a2s_amd/vqt.py): one full-rate Hann-windowed complex
    exponential per bin, no decimation, no sparsification.

Parity status: **UNPINNED** -- no golden vector from librosa itself exists in this container.  What IS checked: the HIP front-end and
``vqt_direct`` against ``vqt_librosa`` (deviation reported in dB and bounded in the tests), and analytic known answers for all three.
"""
import numpy as np

A0_HZ = 27.5
HANN_BANDWIDTH = 1.50018310546875            # librosa.filters.window_bandwidth('hann')


# ----------------------------------------------------------------------------- librosa.filters
def _relative_bandwidth(freqs):
    """librosa.filters._relative_bandwidth: alpha_k from the local bins-per-octave of the frequency grid."""
    logf = np.log2(freqs)
    bpo = np.empty_like(freqs)
    bpo[0] = 1.0 / (logf[1] - logf[0])
    bpo[-1] = 1.0 / (logf[-1] - logf[-2])
    bpo[1:-1] = 2.0 / (logf[2:] - logf[:-2])
    return (2.0 ** (2.0 / bpo) - 1.0) / (2.0 ** (2.0 / bpo) + 1.0)


def wavelet_lengths(freqs, sr, gamma, alpha, filter_scale=1.0):
    """librosa.filters.wavelet_lengths: (fractional) filter lengths and the highest frequency any filter touches."""
    Q = float(filter_scale) / alpha
    f_cutoff = np.max(freqs * (1.0 + 0.5 * HANN_BANDWIDTH / Q) + 0.5 * gamma)
    return Q * sr / (freqs + gamma / alpha), f_cutoff


def wavelet(freqs, sr, gamma, alpha):
    """librosa.filters.wavelet(norm=1, pad_fft=True, window='hann'): rows = centred, zero-padded time-domain filters."""
    lengths, _ = wavelet_lengths(freqs, sr, gamma, alpha)
    filters = []
    for ilen, freq in zip(lengths, freqs):
        n = np.arange(-ilen // 2, ilen // 2, dtype=float)                 # floor(-ilen/2) .. floor(ilen/2) - 1
        sig = np.exp(1j * n * 2.0 * np.pi * freq / sr)
        m = len(sig)
        sig = sig * (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(m) / m))   # scipy get_window('hann', m, fftbins=True)
        sig = sig / np.sum(np.abs(sig))                                    # util.normalize(norm=1)
        filters.append(sig)
    max_len = int(2.0 ** np.ceil(np.log2(max(lengths))))
    out = np.zeros((len(filters), max_len), dtype=np.complex128)
    for i, f in enumerate(filters):                                        # util.pad_center
        lpad = (max_len - len(f)) // 2
        out[i, lpad:lpad + len(f)] = f
    return out, lengths


def sparsify_rows(x, quantile=0.01):
    """librosa.util.sparsify_rows: zero the smallest entries of each row that together hold < quantile of its L1 norm."""
    mags = np.abs(x)
    norms = np.sum(mags, axis=1, keepdims=True)
    mag_sort = np.sort(mags, axis=1)
    cumulative = np.cumsum(mag_sort / norms, axis=1)
    threshold_idx = np.argmin(cumulative < quantile, axis=1)
    out = np.zeros_like(x)
    for i, j in enumerate(threshold_idx):
        keep = mags[i] >= mag_sort[i, j]
        out[i, keep] = x[i, keep]
    return out


def _vqt_filter_fft(sr, freqs, gamma, alpha, sparsity):
    """librosa.core.constantq.__vqt_filter_fft: sparsified non-negative-frequency half of the FFT'd filter bank."""
    basis, lengths = wavelet(freqs, sr, gamma, alpha)
    n_fft = basis.shape[1]
    basis = basis * (lengths[:, None] / float(n_fft))
    fft_basis = np.fft.fft(basis, n=n_fft, axis=1)[:, : n_fft // 2 + 1]
    return sparsify_rows(fft_basis, quantile=sparsity), n_fft


def _cqt_response(y, n_fft, hop, fft_basis):
    """librosa.core.constantq.__cqt_response: rectangular-window STFT (center=True, zero padding) times the basis."""
    yp = np.concatenate([np.zeros(n_fft // 2), y, np.zeros(n_fft // 2)])
    n_frames = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    D = np.fft.rfft(yp[idx], n=n_fft, axis=1)                              # (frames, n_fft/2+1)
    return fft_basis @ D.T                                                 # (bins, frames)


def _decimate2(y, passband=0.913, atten_db=120.0):
    """Stand-in for ``librosa.resample(y, orig_sr=2, target_sr=1, res_type='soxr_hq', scale=True)`` (see the module docstring):
    zero-phase Kaiser-windowed-sinc low-pass between `passband` x and 1.0 x the new Nyquist, every second sample, times sqrt(2)."""
    width = (1.0 - passband) * 0.25                                        # transition width in cycles/sample of the input rate
    cutoff = (1.0 + passband) * 0.125                                      # centre of the transition band
    beta = 0.1102 * (atten_db - 8.7)
    half = int(np.ceil((atten_db - 8.0) / (2.285 * 2.0 * np.pi * width) / 2.0))
    n = np.arange(-half, half + 1)
    h = 2.0 * cutoff * np.sinc(2.0 * cutoff * n) * np.kaiser(2 * half + 1, beta)
    h /= h.sum()
    n_out = int(np.ceil(len(y) / 2.0))                                     # librosa: ceil(n * ratio)
    full = np.convolve(y, h)                                               # full[j + half] = sum_k h[k] y[j - k]: zero-phase at offset `half`
    out = full[half: half + 2 * n_out: 2]
    if len(out) < n_out:
        out = np.concatenate([out, np.zeros(n_out - len(out))])
    return out * np.sqrt(2.0)


def vqt_librosa(y, sr=16000, hop=160, n_bins=480, bins_per_octave=60, gamma=20.0, fmin=A0_HZ, sparsity=0.01, decimator=_decimate2):
    """Complex VQT (n_bins, frames) as librosa 0.10.1's ``vqt`` computes it for the reference's call (scale=True, norm=1, hann,
    pad_mode='constant', tuning=0, filter_scale=1)."""
    y = np.asarray(y, dtype=np.float64)
    n_octaves = int(np.ceil(float(n_bins) / bins_per_octave))
    n_filters = min(bins_per_octave, n_bins)
    freqs = fmin * 2.0 ** (np.arange(n_bins) / bins_per_octave)            # interval_frequencies(intervals='equal')
    alpha = _relative_bandwidth(freqs)
    lengths, filter_cutoff = wavelet_lengths(freqs, sr, gamma, alpha)
    nyquist = sr / 2.0
    if filter_cutoff > nyquist:
        raise ValueError("Wavelet basis with max frequency would exceed the Nyquist frequency")
    # __early_downsample_count: min(max(0, ceil(log2(nyquist / cutoff)) - 2), max(0, twos(hop) - n_octaves + 1)) -- 0 for this call
    twos = 0
    while hop % (2 ** (twos + 1)) == 0:
        twos += 1
    early = min(max(0, int(np.ceil(np.log2(nyquist / filter_cutoff)) - 1) - 1), max(0, twos - n_octaves + 1))
    if early != 0:
        raise NotImplementedError("early downsampling does not occur for the reference's parameters and is not restated")
    resp = []
    my_y, my_sr, my_hop = y, float(sr), hop
    for i in range(n_octaves):
        sl = slice(-n_filters, None) if i == 0 else slice(-n_filters * (i + 1), -n_filters * i)
        fft_basis, n_fft = _vqt_filter_fft(my_sr, freqs[sl], gamma, alpha[sl], sparsity)
        fft_basis = fft_basis * np.sqrt(sr / my_sr)                        # compensate for the decimations so far
        resp.append(_cqt_response(my_y, n_fft, my_hop, fft_basis))
        if my_hop % 2 == 0:
            my_hop //= 2
            my_sr /= 2.0
            my_y = decimator(my_y)
    # __trim_stack: octaves were produced top-down
    max_col = min(c.shape[-1] for c in resp)
    V = np.empty((n_bins, max_col), dtype=np.complex128)
    end = n_bins
    for c in resp:
        n_oct = c.shape[0]
        if end < n_oct:
            V[:end] = c[-end:, :max_col]
        else:
            V[end - n_oct:end] = c[:, :max_col]
        end -= n_oct
    return V / np.sqrt(lengths)[:, None]                                   # scale=True


def amplitude_to_unit(mag):
    """reference utilities.py:253: amplitude_to_db(|V|, ref=np.max) / 80 + 1 (librosa defaults amin=1e-5, top_db=80), transposed."""
    mag = np.abs(mag)
    log_spec = 20.0 * np.log10(np.maximum(1e-5, mag)) - 20.0 * np.log10(np.maximum(1e-5, mag.max()))
    log_spec = np.maximum(log_spec, log_spec.max() - 80.0)
    return (log_spec / 80.0 + 1.0).T                                       # (frames, bins)


def vqt_features_librosa(y, **kw):
    """reference utilities.get_VQT on a waveform: (frames, 480) in [0, 1]."""
    return amplitude_to_unit(vqt_librosa(y, **kw))


# ----------------------------------------------------------------------------- the direct-form definition of the HIP front-end
def vqt_direct(y, sr=16000, hop=160, n_bins=480, bins_per_octave=60, gamma=20.0, fmin=A0_HZ):
    """(frames, n_bins) features from one full-rate kernel per bin (what piano_a2s_amd/vqt.py's framed complex GEMM evaluates)."""
    y = np.asarray(y, dtype=np.float64)
    freqs = fmin * 2.0 ** (np.arange(n_bins) / bins_per_octave)
    r = 2.0 ** (2.0 / bins_per_octave)
    alpha = (r - 1.0) / (r + 1.0)
    lengths = sr / (alpha * (freqs + gamma / alpha))
    frames = 1 + len(y) // hop
    pad = int(np.ceil(lengths.max())) // 2 + 2
    yp = np.concatenate([np.zeros(pad), y, np.zeros(pad + hop)])
    Cq = np.zeros((frames, n_bins), dtype=np.complex128)
    centres = pad + hop * np.arange(frames)
    for k in range(n_bins):
        L = lengths[k]
        n = np.arange(-int(L // 2), int(L // 2) + 1)
        n = n[np.abs(n) <= L / 2]
        kern = (0.5 + 0.5 * np.cos(2 * np.pi * n / L)) * np.exp(2j * np.pi * freqs[k] * n / sr)
        kern = kern / np.abs(kern).sum() / np.sqrt(L)
        Cq[:, k] = yp[centres[:, None] + n[None, :]] @ np.conj(kern)
    return amplitude_to_unit(Cq.T)


vqt_ref = vqt_direct            # name used by round-1 tests
