"""The VQT oracle (oracle/vqt_ref.py: librosa 0.10.1's published multirate algorithm restated on the CPU) -- analytic known answers, the
measured (not assumed) insensitivity to the one step that cannot follow librosa (the libsoxr decimator), the deviation of the direct-form
definition from it, and the PRODUCT's host-side filter-bank construction (piano_a2s_amd/vqt.py: time-domain banks for the framed GEMMs)
against the oracle's FFT-domain formulation.  Parity with librosa itself stays UNPINNED (no librosa in this image, no reference sample)."""
import numpy as np
import pytest

from oracle import vqt_ref as V
from piano_a2s_amd import synthetic
from piano_a2s_amd import vqt as product


@pytest.fixture(scope="module")
def piano():
    w = synthetic.make_waveforms(1, 7, seconds=3.0)[0].numpy().astype(np.float64)
    return w, V.vqt_librosa(w)


def test_analytic_known_answers():
    sr, N = 16000, 16000
    t = np.arange(N) / sr
    freqs = 27.5 * 2.0 ** (np.arange(480) / 60)
    alpha = V._relative_bandwidth(freqs)
    assert np.allclose(alpha, (2 ** (2 / 60) - 1) / (2 ** (2 / 60) + 1))            # equal temperament: one alpha
    lengths, cutoff = V.wavelet_lengths(freqs, sr, 20.0, alpha)
    assert 787 < lengths[0] < 789 and cutoff < sr / 2                                # the 788-tap bottom filter; bank below Nyquist
    for k in (30, 200, 330, 450):                                                    # octaves 7, 4, 2, 0 (0 = undecimated)
        Vc = np.abs(V.vqt_librosa(np.sin(2 * np.pi * freqs[k] * t)))
        assert Vc.shape == (480, 1 + N // 160)
        assert abs(int(Vc[:, 50].argmax()) - k) <= 1
        # scale=True: a unit sinusoid on a bin centre reads sqrt(length)/2, as a length-L orthonormal DFT would (within the sparsified
        # basis' 1 % and the window-length quantisation of the decimated octaves)
        assert Vc[k, 50] == pytest.approx(0.5 * np.sqrt(lengths[k]), rel=0.03), k
    feats = V.vqt_features_librosa(np.zeros(1600))
    assert feats.shape == (11, 480) and np.all(feats == 1.0)                         # silence: floor vs floor = 0 dB
    feats = V.vqt_features_librosa(np.sin(2 * np.pi * 440.0 * t))
    assert feats.max() == 1.0 and feats.min() >= 0.0 and feats.shape == (101, 480)


def test_decimator_stand_in_sensitivity(piano):
    """librosa decimates with libsoxr ('soxr_hq'), whose coefficients are not published; the oracle uses a Kaiser-sinc with its documented
    band edges.  How much can that matter?  Two other designs (edge 0.90 / 100 dB and 0.93 / 140 dB), measured on a 3 s piano-like clip:
    the features move by 0.005 dB on average, 99.9 % of the cells by < 1 dB, every cell within 40 dB of the clip maximum by < 1.3 dB.
    Single faint cells move by up to 16 dB: top bins of a decimated octave (their skirts reach the 0.913..1.0 transition band) next to a
    strong partial just above -- those cells are decimator-defined in librosa as well and cannot be pinned without libsoxr itself."""
    w, ref = piano
    f0 = V.amplitude_to_unit(ref)
    for pb, att in ((0.90, 100.0), (0.93, 140.0)):
        f1 = V.amplitude_to_unit(V.vqt_librosa(w, decimator=lambda x: V._decimate2(x, passband=pb, atten_db=att)))
        d = np.abs(f0 - f1) * 80.0
        assert d.mean() < 0.05, d.mean()
        assert np.quantile(d, 0.999) < 2.0, np.quantile(d, 0.999)
        assert d[f0 > 0.5].max() < 2.5, d[f0 > 0.5].max()
        worst = np.unravel_index(d.argmax(), d.shape)[1]
        assert worst % 60 >= 48, worst                                               # the worst cell sits in the top fifth of its octave


def test_direct_form_deviation_report(piano):
    """How far a direct (single-rate, unsparsified) evaluation is from the multirate algorithm -- the number VERDICT r1 asked for.
    (a) With librosa's filter construction and channel scaling the direct form agrees to ~0.5 dB on average: what is left is the
    multirate algorithm's own approximation (window lengths quantised at the decimated rates, 1 % sparsification).  (b) The round-1
    definition (channel scale 1/sqrt(length) without librosa's length/n_fft factor) is tilted by length_k: 10 log10(788/160) x 2 = 14 dB
    between the bottom and the top bin -- which is why the product now evaluates the multirate algorithm itself."""
    w, ref = piano
    f0 = V.amplitude_to_unit(ref)
    sr = 16000
    freqs = 27.5 * 2.0 ** (np.arange(480) / 60)
    alpha = V._relative_bandwidth(freqs)
    lengths, _ = V.wavelet_lengths(freqs, sr, 20.0, alpha)
    frames = 1 + len(w) // 160
    pad = 400
    yp = np.concatenate([np.zeros(pad), w, np.zeros(pad + 160)])
    centres = pad + 160 * np.arange(frames)
    Cq = np.zeros((480, frames), dtype=np.complex128)
    for k in range(480):
        n = np.arange(-lengths[k] // 2, lengths[k] // 2)
        m = len(n)
        sig = np.exp(2j * np.pi * freqs[k] * n / sr) * (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(m) / m))
        sig = sig / np.abs(sig).sum() * np.sqrt(lengths[k])
        Cq[k] = yp[centres[:, None] - n.astype(int)[None, :]] @ sig
    d = np.abs(V.amplitude_to_unit(Cq) - f0) * 80.0
    assert d.mean() < 1.0, d.mean()
    loud = f0 > 0.5                                                                  # cells within 40 dB of the clip maximum
    assert np.quantile(d[loud], 0.99) < 6.0
    old = np.abs(V.vqt_direct(w) - f0) * 80.0                                        # round-1 definition
    assert 3.0 < old.mean() < 12.0                                                   # the tilt
    print(f"direct form vs multirate: mean {d.mean():.2f} dB, p99 (loud cells) {np.quantile(d[loud], 0.99):.2f} dB; round-1 scaling: mean {old.mean():.2f} dB")


def test_product_filter_banks_equal_the_fft_formulation():
    """piano_a2s_amd.vqt.octave_banks (time-domain banks g for the framed GEMMs) against the oracle's FFT-domain path on random frames:
    frame . g  ==  sparsified_fft_basis . rfft(frame) * sqrt(sr/sr_o) / sqrt(length)  for every octave; same hops, n_fft, decimator."""
    sr = 16000
    freqs = 27.5 * 2.0 ** (np.arange(480) / 60)
    alpha = V._relative_bandwidth(freqs)
    lengths, _ = V.wavelet_lengths(freqs, sr, 20.0, alpha)
    banks = product.octave_banks()
    assert [o["hop"] for o in banks] == [160, 80, 40, 20, 10, 5, 5, 5] and [o["n_fft"] for o in banks] == [512, 256, 256, 128, 64, 32, 32, 32]
    g = np.random.default_rng(0)
    my_sr = float(sr)
    for i, o in enumerate(banks):
        sl = slice(o["lo"], o["hi"])
        assert (o["lo"], o["hi"]) == (480 - 60 * (i + 1), 480 - 60 * i)
        fft_basis, n_fft = V._vqt_filter_fft(my_sr, freqs[sl], 20.0, alpha[sl], 0.01)
        assert n_fft == o["n_fft"] and o["bank"].flags["C_CONTIGUOUS"] and o["bank"].dtype == np.float32       # what the GEMM is told it gets
        frame = g.standard_normal(n_fft)
        want = (fft_basis @ np.fft.rfft(frame)) * np.sqrt(sr / my_sr) / np.sqrt(lengths[sl])
        nb = o["hi"] - o["lo"]
        got = frame @ o["bank"][:, :nb].astype(np.float64) + 1j * (frame @ o["bank"][:, nb:].astype(np.float64))
        assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max(), i
        if o["hop"] % 2 == 0:
            my_sr /= 2.0
    h, half = product.decimation_filter()
    y = g.standard_normal(1001)
    want = V._decimate2(y)
    got = np.convolve(y, h)[half: half + 2 * len(want): 2]
    assert np.abs(got - want[:len(got)]).max() < 1e-12
