"""Shape contract of the transcription model's state (single source of truth, host side).

Names, shapes and ordering are those of the reference's ``ScoreTranscription.state_dict()``
(reference models.py:14-24,53-73,84-139,340-364,440-450,463-521; listed in SURVEY.md 8b "State"),
so a checkpoint written by the reference loads here and vice versa.  GRU tensors use PyTorch's
packing: rows [r; z; n].
"""
from collections import OrderedDict

import numpy as np
import torch

VOCAB_SIZE = 173          # LabelsMultiple(extended=True)
SOS, EOS, PAD = 145, 146, 147


def default_cfg(**kw):
    """Constructor arguments as hparams/pretrain.yaml resolves them."""
    cfg = dict(in_channels=1, freq_bins=480, conv_feature_size=256, hidden_size=256, max_bars=5,
               num_time_sig=7, num_keys=14, max_length=(398, 189), note_emb_size=16, staff_emb_size=32,
               time_sig_emb_size=5, key_emb_size=8)
    cfg.update(kw)
    cfg["max_length"] = tuple(cfg["max_length"])
    return cfg


def _gru(prefix, inp, hid, layers, bidir):
    out = []
    for l in range(layers):
        i = inp if l == 0 else hid * (2 if bidir else 1)
        for sfx in ([f"l{l}", f"l{l}_reverse"] if bidir else [f"l{l}"]):
            out += [(f"{prefix}.weight_ih_{sfx}", (3 * hid, i)), (f"{prefix}.weight_hh_{sfx}", (3 * hid, hid)),
                    (f"{prefix}.bias_ih_{sfx}", (3 * hid,)), (f"{prefix}.bias_hh_{sfx}", (3 * hid,))]
    return out


def _bn(prefix, c):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)),
            (prefix + ".running_var", (c,)), (prefix + ".num_batches_tracked", ())]


def state_spec(cfg):
    """OrderedDict name -> shape, in the reference's state_dict order (83 parameters + 15 buffers)."""
    H, C, Fb = cfg["hidden_size"], cfg["conv_feature_size"], cfg["freq_bins"]
    ne, se = cfg["note_emb_size"], cfg["staff_emb_size"]
    te, ke = cfg["time_sig_emb_size"], cfg["key_emb_size"]
    s = []
    s += [("convstack.conv1.weight", (20, cfg["in_channels"], 3, 3)), ("convstack.conv2.weight", (20, 20, 3, 3)),
          ("convstack.conv3.weight", (40, 20, 3, 3)), ("convstack.conv4.weight", (40, 40, 3, 3))]
    s += _bn("convstack.bn1", 20) + _bn("convstack.bn2", 20) + _bn("convstack.bn3", 40) + _bn("convstack.bn4", 40)
    s += [("convstack.out.weight", (C, Fb * 40))] + _bn("convstack.out_bn", C)
    s += _gru("encoder.gru", C, H, 2, True)
    s += [("encoder.fc.weight", (H, 2 * H)), ("encoder.fc.bias", (H,))]
    s += [("decoder.note_emb.weight", (VOCAB_SIZE, ne)), ("decoder.time_sig_emb.weight", (cfg["num_time_sig"] + 1, te)),
          ("decoder.key_emb.weight", (cfg["num_keys"] + 1, ke))]
    s += _gru("decoder.staff_emb", ne, se, 1, True)
    for st in ("upper_decoder", "lower_decoder"):
        p = f"decoder.{st}"
        s += [(p + ".embedding.weight", (VOCAB_SIZE, ne)), (p + ".attn.attn.weight", (H, 4 * H)),
              (p + ".attn.attn.bias", (H,)), (p + ".attn.v.weight", (1, H))]
        s += _gru(p + ".gru", ne + 2 * H, 2 * H, 1, False)
        s += [(p + ".out.weight", (VOCAB_SIZE, 4 * H)), (p + ".out.bias", (VOCAB_SIZE,))]
    s += [("decoder.attn.attn.weight", (H, 4 * H)), ("decoder.attn.attn.bias", (H,)), ("decoder.attn.v.weight", (1, H))]
    s += _gru("decoder.gru", 4 * se + te + ke + 2 * H, 2 * H, 1, False)
    for head, n in (("time_sig_out", cfg["num_time_sig"]), ("key_out", cfg["num_keys"])):
        p = f"decoder.{head}"
        s += [(p + ".0.weight", (4 * H, 4 * H)), (p + ".0.bias", (4 * H,)), (p + ".2.weight", (2 * H, 4 * H)),
              (p + ".2.bias", (2 * H,)), (p + ".4.weight", (n, 2 * H)), (p + ".4.bias", (n,))]
    return OrderedDict(s)


def is_buffer(name):
    return name.endswith(("running_mean", "running_var", "num_batches_tracked"))


# Gain presets for procedural_state(lively=...).  True / "matrix": whole-matrix gains, tuned on the reduced
# model of fixture G1.  "token": strengthens the fed-back-token path instead (note-decoder embeddings), which
# is what keeps greedy decoding of the FULL-width model from collapsing onto one symbol (fixture G2).
_LIVELY_GAINS = (("_decoder.out.weight", 6.0), ("decoder.gru.weight_ih", 3.0), ("_decoder.gru.weight_hh", 2.0),
                 ("attn.v.weight", 4.0), ("attn.attn.weight", 4.0), ("encoder.gru.weight", 2.0))
_TOKEN_GAINS = (("_decoder.out.weight", 3.0), ("_decoder.embedding.weight", 16.0),
                ("attn.v.weight", 2.0), ("attn.attn.weight", 2.0))


def procedural_state(cfg, seed, eos_bias=0.0, lively=False):
    """Deterministic, well-conditioned weights for full-size parity cases (no 65 MB checkpoint in git).

    One numpy PCG64 stream, consumed in state_spec order.  Matrices ~ U(+-sqrt(3/fan_in)), biases
    ~ U(+-0.1), embeddings ~ U(+-1), BN gamma ~ U(0.5,1.5), beta ~ U(+-0.2), running mean ~ U(+-0.1),
    running var ~ U(0.5,1.5).  ``eos_bias`` is added to the <eos> logit bias of both note decoders so
    greedy decoding terminates at data-dependent steps (exercises the early-break bookkeeping).
    ``lively`` (True/"matrix" or "token") applies one of the gain presets above so that the decoded tokens
    depend visibly on the audio and on the fed-back tokens instead of collapsing to one symbol.
    """
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, shape in state_spec(cfg).items():
        if name.endswith("num_batches_tracked"):
            out[name] = torch.tensor(3, dtype=torch.long)
            continue
        n = int(np.prod(shape)) if shape else 1
        u = rng.random(n, dtype=np.float64) * 2.0 - 1.0
        if name.endswith("running_var"):
            v = 1.0 + 0.5 * u
        elif name.endswith("running_mean"):
            v = 0.1 * u
        elif ".bn" in name or "out_bn" in name:
            v = (1.0 + 0.5 * u) if name.endswith("weight") else 0.2 * u
        elif name.endswith(("emb.weight", "embedding.weight")):
            v = u
        elif name.endswith("bias") or ".bias_" in name:
            v = 0.1 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            v = u * np.sqrt(3.0 / fan_in)
        if lively:
            for key, gain in (_TOKEN_GAINS if lively == "token" else _LIVELY_GAINS):
                if key in name:
                    v = v * gain
        t = torch.from_numpy(v.astype(np.float32)).reshape(shape)
        if eos_bias and name.endswith("_decoder.out.bias"):
            t[EOS] += eos_bias
        out[name] = t
    return out


def split_state(state):
    """-> (parameters dict, buffers dict)."""
    P = OrderedDict((k, v) for k, v in state.items() if not is_buffer(k))
    B = OrderedDict((k, v) for k, v in state.items() if is_buffer(k))
    return P, B


def flat_layout(numels, align=4):
    """Offsets (in floats) of tensors packed into one flat buffer with every tensor starting on a 16-byte boundary (the HIP kernels
    fetch weight rows with 16-byte loads); returns (offsets, total).  The padding floats stay zero: parameters, gradients and
    optimizer state all use this layout, so the fused clip + Adadelta sees zeros there and leaves them zero."""
    offs, off = [], 0
    for n in numels:
        offs.append(off)
        off += (n + align - 1) // align * align
    return offs, off
