"""VQT front-end on the MI355X (piano_a2s_amd/vqt.py + csrc/a2s_vqt.hip): 8 framed complex GEMMs + 5 framed decimations + the
log-magnitude epilogue, against the CPU oracle that restates librosa 0.10.1's multirate algorithm with FFTs in float64
(oracle/vqt_ref.py), and analytic known answers.  Parity with librosa ITSELF stays unpinned (no librosa here; the decimator is a
stand-in for libsoxr -- tests/test_vqt_oracle.py measures what that can change)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _log(line):
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/vqt_errors.txt", "a") as f:
        f.write(line + "\n")


@pytest.mark.parametrize("kind", ["tones", "piano", "ragged"])
def test_matches_the_multirate_oracle(dev, kind):
    from oracle import vqt_ref as V
    from piano_a2s_amd import synthetic
    from piano_a2s_amd.vqt import VQT
    g = np.random.default_rng(3)
    if kind == "tones":
        N = 16000 * 2
        t = np.arange(N) / 16000.0
        y = 0.6 * np.sin(2 * np.pi * 440.0 * t) + 0.3 * np.sin(2 * np.pi * 1318.5 * t + 0.4) + 0.2 * np.sin(2 * np.pi * 55.0 * t) + 0.02 * g.standard_normal(N)
    elif kind == "piano":
        y = synthetic.make_waveforms(1, 7, seconds=3.0)[0].numpy().astype(np.float64)
    else:
        y = g.standard_normal(160 * 37 + 53) * np.linspace(0.1, 1.0, 160 * 37 + 53)       # a length that is no multiple of anything
    ref = V.vqt_features_librosa(y)
    out = VQT(dev)(torch.from_numpy(y.astype(np.float32)).to(dev).unsqueeze(0))
    torch.cuda.synchronize()
    assert out.shape == (1, 1, 1 + len(y) // 160, 480) == (1, 1) + ref.shape
    d = np.abs(out[0, 0].cpu().numpy() - ref)
    audible = ref > 0.05                                   # cells above the -76 dB region, where fp32 round-off of a 512-tap sum is not the signal
    _log(f"{kind}: max {d.max() * 80:.4f} dB, audible max {d[audible].max() * 80:.4f} dB, mean {d.mean() * 80:.5f} dB")
    assert d[audible].max() < 2.5e-3 and d.mean() < 2e-4, (d[audible].max(), d.mean())      # 0.2 dB / 0.016 dB: fp32 GEMMs vs float64 FFTs


def test_batch_and_analytic_known_answers(dev):
    from piano_a2s_amd.vqt import VQT, filter_lengths
    freqs, lengths = filter_lengths()
    vq = VQT(dev)
    N = 16000                                            # 1 s
    t = np.arange(N) / 16000.0
    ks = [30, 100, 250, 400, 450]
    waves = np.stack([np.sin(2 * np.pi * freqs[k] * t) for k in ks] + [np.zeros(N)]).astype(np.float32)
    out = vq(torch.from_numpy(waves).to(dev))[:, 0].cpu().numpy()          # (6, frames, 480): one batched call
    assert out.shape[1] == 1 + N // 160
    mid = out.shape[1] // 2
    for i, k in enumerate(ks):
        assert abs(int(out[i, mid].argmax()) - k) <= 1, (k, int(out[i, mid].argmax()))
        assert abs(out[i].max() - 1.0) < 1e-6 and out[i].min() >= 0.0
        assert out[i, mid, (k + 120) % 480] < 0.6                          # two octaves away: > 32 dB down
    # an all-zero clip: |C| = 0 everywhere -> floor vs floor = 0 dB -> exactly 1.0 (same as amplitude_to_db(ref=max) on silence)
    assert np.all(out[5] == 1.0)
    # each clip of a batch is normalised by its OWN maximum and equals the single-clip call
    one = vq(torch.from_numpy(waves[2:3]).to(dev))[0, 0].cpu().numpy()
    assert np.abs(one - out[2]).max() < 5e-4           # (the batch size changes the GEMM tiling, hence the rounding of faint cells)
