"""Weight initialisation of the drop-in module against the reference's rules: init_layer / init_bn / init_gru
(reference models.py:548-585) where the reference calls them (models.py:71-73,136-139,362-364,448-450,509-521), PyTorch defaults where
it does not (staff_emb GRU, the two MLP heads, every nn.Embedding)."""
import math

import pytest
import torch

import models


@pytest.fixture(scope="module")
def model():
    torch.manual_seed(4321)
    return models.ScoreTranscription(freq_bins=48, conv_feature_size=64, hidden_size=32, max_length=(12, 8))


def _P(model):
    return dict(model.named_parameters())


def test_conv_and_linear_layers_are_xavier_uniform_with_zero_bias(model):
    P = _P(model)
    for name in ["convstack.conv1.weight", "convstack.conv2.weight", "convstack.conv3.weight", "convstack.conv4.weight", "convstack.out.weight",
                 "encoder.fc.weight", "decoder.attn.attn.weight", "decoder.attn.v.weight", "decoder.upper_decoder.attn.attn.weight",
                 "decoder.upper_decoder.out.weight", "decoder.lower_decoder.attn.v.weight", "decoder.lower_decoder.out.weight"]:
        w = P[name]
        rf = w[0][0].numel() if w.dim() > 2 else 1
        fan_in, fan_out = w.shape[1] * rf, w.shape[0] * rf
        bound = math.sqrt(6.0 / (fan_in + fan_out))                    # nn.init.xavier_uniform_ (init_layer, models.py:548-553)
        assert float(w.abs().max()) <= bound + 1e-7, name
        if w.numel() >= 2000:                                          # uniform: std = bound / sqrt(3), and the range is actually used
            assert float(w.std()) == pytest.approx(bound / math.sqrt(3), rel=0.08), name
            assert float(w.abs().max()) > 0.9 * bound, name
    for name in ["encoder.fc.bias", "decoder.attn.attn.bias", "decoder.upper_decoder.attn.attn.bias", "decoder.upper_decoder.out.bias",
                 "decoder.lower_decoder.out.bias"]:
        assert float(P[name].abs().max()) == 0.0, name                 # init_layer fills biases with 0


def test_batchnorm_is_identity_at_init(model):
    P, Bf = _P(model), dict(model.named_buffers())
    for bn in ("convstack.bn1", "convstack.bn2", "convstack.bn3", "convstack.bn4", "convstack.out_bn"):
        assert bool((P[bn + ".weight"] == 1).all()) and bool((P[bn + ".bias"] == 0).all())                # init_bn (models.py:556-559)
        assert bool((Bf[bn + ".running_mean"] == 0).all()) and bool((Bf[bn + ".running_var"] == 1).all()) and int(Bf[bn + ".num_batches_tracked"]) == 0


def test_init_gru_rule(model):
    """init_gru (models.py:561-585): every gate block ~ U(+-sqrt(3 / fan_in)), except the n-gate block of weight_hh which is
    orthogonal; all four bias vectors zero.  It only visits weight_*_l{i}: the reverse direction of the encoder keeps torch's default."""
    P = _P(model)
    k = 1.0 / math.sqrt(32)
    for sfx in ("l0_reverse", "l1_reverse"):
        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            t = P[f"encoder.gru.{n}_{sfx}"]
            assert float(t.abs().max()) <= k + 1e-7 and float(t.abs().max()) > 0.5 * k, (n, sfx)
    for prefix, sfxs in (("encoder.gru", ("l0", "l1")), ("decoder.gru", ("l0",)),
                         ("decoder.upper_decoder.gru", ("l0",)), ("decoder.lower_decoder.gru", ("l0",))):
        for sfx in sfxs:
            for kind in ("ih", "hh"):
                w = P[f"{prefix}.weight_{kind}_{sfx}"]
                hid = w.shape[0] // 3
                for gate in range(3):
                    blk = w[gate * hid:(gate + 1) * hid]
                    if kind == "hh" and gate == 2:
                        eye = blk @ blk.t()                           # square block: orthogonal rows
                        assert torch.allclose(eye, torch.eye(hid), atol=1e-5), (prefix, sfx)
                    else:
                        bound = math.sqrt(3.0 / blk.shape[1])
                        assert float(blk.abs().max()) <= bound + 1e-7, (prefix, sfx, kind, gate)
                        assert float(blk.std()) == pytest.approx(bound / math.sqrt(3), rel=0.12), (prefix, sfx, kind, gate)
                assert float(P[f"{prefix}.bias_{kind}_{sfx}"].abs().max()) == 0.0


def test_torch_defaults_where_the_reference_does_not_initialise(model):
    P = _P(model)
    # nn.GRU default: everything U(+-1/sqrt(hidden)), biases included and therefore NOT zero (staff_emb, models.py:96-97,136-139)
    k = 1.0 / math.sqrt(32)                                            # staff_emb_size
    for sfx in ("l0", "l0_reverse"):
        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            t = P[f"decoder.staff_emb.{n}_{sfx}"]
            assert float(t.abs().max()) <= k + 1e-7 and float(t.abs().max()) > 0.5 * k
    # nn.Linear default in the heads: kaiming_uniform(a=sqrt(5)) = U(+-1/sqrt(fan_in)) for weight and bias (models.py:123-132)
    for head in ("decoder.time_sig_out", "decoder.key_out"):
        for i in (0, 2, 4):
            w, b = P[f"{head}.{i}.weight"], P[f"{head}.{i}.bias"]
            bound = 1.0 / math.sqrt(w.shape[1])
            assert float(w.abs().max()) <= bound + 1e-7 and float(b.abs().max()) <= bound + 1e-7 and float(b.abs().max()) > 0
    # nn.Embedding default: N(0, 1)
    for n in ("decoder.note_emb.weight", "decoder.upper_decoder.embedding.weight", "decoder.lower_decoder.embedding.weight"):
        t = P[n]
        assert float(t.mean()) == pytest.approx(0.0, abs=0.1) and float(t.std()) == pytest.approx(1.0, rel=0.1) and float(t.abs().max()) > 2.0


def test_parameter_order_matches_flat_layout(model):
    layout = model.flat_layout()
    assert len(layout) == 83 and [s for _, s in layout] == [tuple(p.shape) for p in model.parameters()]
    assert all(off % 4 == 0 for off, _ in layout)
