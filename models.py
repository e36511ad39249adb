"""Drop-in for the reference's ``models.py``: ``models.ScoreTranscription`` on the MI355X HIP path.

Same import path, constructor keywords, ``forward`` signature, return values and ``state_dict`` names as
reference models.py:14-51 (so ``hparams/*.yaml`` ``!new:models.ScoreTranscription`` and the published
``model.ckpt`` work unchanged), but the module owns no torch.nn layers: parameters are plain tensors with the
reference's names and all arithmetic is done by liba2s_hip.so through piano_a2s_amd.engine (forward) and
piano_a2s_amd.engine_bwd (backward, attached to autograd by one custom Function).

There is deliberately NO CPU implementation: calling it with CPU tensors raises.  (The CPU restatement used for
parity lives under oracle/ and is test infrastructure only.)
"""
import math
import random

import torch
import torch.nn as nn

from data_processing.humdrum import LabelsMultiple
from piano_a2s_amd import engine, engine_bwd, spec

labels = LabelsMultiple(extended=True)
SOS = labels.labels_map['<sos>']
EOS = labels.labels_map['<eos>']
vocab_size = len(labels.labels_map)


class _Node(nn.Module):
    """Name-space container: gives parameters the dotted names of the reference's module tree."""


def _xavier_uniform(t):
    nn.init.xavier_uniform_(t)


def _gru_init(P, prefix, suffixes):
    """reference init_gru (models.py:561-585): per-gate U(+-sqrt(3/fan_in)); n-gate of weight_hh orthogonal; biases 0."""
    for sfx in suffixes:
        for kind in ("ih", "hh"):
            w = P[f"{prefix}.weight_{kind}_{sfx}"]
            hid = w.shape[0] // 3
            for gate in range(3):
                block = w[gate * hid:(gate + 1) * hid]
                if kind == "hh" and gate == 2:
                    nn.init.orthogonal_(block)
                else:
                    bound = math.sqrt(3.0 / block.shape[1])
                    nn.init.uniform_(block, -bound, bound)
            P[f"{prefix}.bias_{kind}_{sfx}"].zero_()


def _default_gru_init(P, prefix, suffixes):
    """torch.nn.GRU default: everything U(+-1/sqrt(hidden)) (reference leaves staff_emb at the default, models.py:136-139)."""
    for sfx in suffixes:
        hid = P[f"{prefix}.weight_hh_{sfx}"].shape[1]
        k = 1.0 / math.sqrt(hid)
        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            nn.init.uniform_(P[f"{prefix}.{n}_{sfx}"], -k, k)


def _default_linear_init(P, prefix):
    """torch.nn.Linear default (kaiming_uniform(a=sqrt(5)) weight, U(+-1/sqrt(fan_in)) bias): the heads' MLPs."""
    w = P[prefix + ".weight"]
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    bound = 1.0 / math.sqrt(w.shape[1])
    nn.init.uniform_(P[prefix + ".bias"], -bound, bound)


class _Transcribe(torch.autograd.Function):
    """One autograd node for the whole model: forward = Engine.forward, backward = engine_bwd.backward."""

    @staticmethod
    def forward(ctx, module, spectrogram, inference, ground_truth, tf_ratio, names, need_grad, *params):
        # NB grad mode is always off inside Function.forward: whether the caller wants gradients is decided by the
        # module (need_grad) before entering.
        S = dict(zip(names, params))
        S.update(module._buffer_dict())
        eng = engine.Engine(module.cfg)
        outs = eng.forward(S, spectrogram, inference=inference, ground_truth=ground_truth, teacher_forcing_ratio=tf_ratio,
                           training=module.training, rng=random, dropout=True)
        ctx.eng, ctx.S, ctx.names, ctx.can_backward = eng, S, names, module.training
        if not need_grad:
            eng.saved = None          # nothing to keep alive
        return outs

    @staticmethod
    def backward(ctx, *grad_outputs):
        if not ctx.can_backward or ctx.eng.saved is None:
            raise RuntimeError("ScoreTranscription: backward is implemented for training mode (batch-statistics BatchNorm) only")
        G = engine_bwd.backward(ctx.eng, ctx.S, [g if g is not None else torch.zeros_like(o) for g, o in zip(grad_outputs, ctx.eng.saved["outs"])])
        ctx.eng.saved = None
        return (None, None, None, None, None, None, None) + tuple(G[n] for n in ctx.names)


class ScoreTranscription(nn.Module):
    def __init__(self, in_channels=1, freq_bins=480, conv_feature_size=256,
                 hidden_size=256, max_bars=5, num_time_sig=7, num_keys=14,
                 max_length=(437, 129), note_emb_size=16, staff_emb_size=32,
                 time_sig_emb_size=5, key_emb_size=8):
        super().__init__()
        self.cfg = spec.default_cfg(in_channels=in_channels, freq_bins=freq_bins, conv_feature_size=conv_feature_size,
                                    hidden_size=hidden_size, max_bars=max_bars, num_time_sig=num_time_sig, num_keys=num_keys,
                                    max_length=tuple(max_length), note_emb_size=note_emb_size, staff_emb_size=staff_emb_size,
                                    time_sig_emb_size=time_sig_emb_size, key_emb_size=key_emb_size)
        if not (isinstance(hidden_size, int) and 1 <= hidden_size <= 512):
            # (the reference constructor takes any width; here the one-workgroup-per-clip attention kernels take any width up to 512 -- 256 = hparams/*.yaml
            # and the published checkpoints runs on the split-T / multi-row / persistent kernels, widths that are multiples of 16 on the fused step kernels)
            raise ValueError(f"hidden_size {hidden_size}: the HIP attention kernels take widths 1 .. 512")
        self._names = []
        P = {}
        for name, shape in spec.state_spec(self.cfg).items():
            *path, leaf = name.split(".")
            node = self
            for part in path:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            if spec.is_buffer(name):
                if leaf == "num_batches_tracked":
                    t = torch.tensor(0, dtype=torch.long)
                else:
                    t = torch.ones(shape) if leaf == "running_var" else torch.zeros(shape)
                node.register_buffer(leaf, t)
            else:
                p = nn.Parameter(torch.zeros(shape))
                node.register_parameter(leaf, p)
                P[name] = p
                self._names.append(name)
        self._init_weights(P)
        self._flat = None

    @torch.no_grad()
    def _init_weights(self, P):
        """reference init_weight calls (models.py:71-73,136-139,362-364,448-450,509-521) + torch defaults elsewhere."""
        for i in (1, 2, 3, 4):
            _xavier_uniform(P[f"convstack.conv{i}.weight"])
            P[f"convstack.bn{i}.weight"].fill_(1.0)
        _xavier_uniform(P["convstack.out.weight"])
        P["convstack.out_bn.weight"].fill_(1.0)
        # init_gru walks `weight_ih_l{i}` / `weight_hh_l{i}` for i < num_layers only (models.py:574-585): the REVERSE direction of the
        # bidirectional encoder GRU is never touched by it and keeps torch.nn.GRU's default U(+-1/sqrt(hidden)), biases included
        _gru_init(P, "encoder.gru", ("l0", "l1"))
        _default_gru_init(P, "encoder.gru", ("l0_reverse", "l1_reverse"))
        _xavier_uniform(P["encoder.fc.weight"])
        for n in ("decoder.note_emb.weight", "decoder.time_sig_emb.weight", "decoder.key_emb.weight",
                  "decoder.upper_decoder.embedding.weight", "decoder.lower_decoder.embedding.weight"):
            nn.init.normal_(P[n])
        _default_gru_init(P, "decoder.staff_emb", ("l0", "l0_reverse"))
        for st in ("decoder.upper_decoder", "decoder.lower_decoder"):
            _xavier_uniform(P[st + ".attn.attn.weight"])
            _xavier_uniform(P[st + ".attn.v.weight"])
            _gru_init(P, st + ".gru", ("l0",))
            _xavier_uniform(P[st + ".out.weight"])
        _xavier_uniform(P["decoder.attn.attn.weight"])
        _xavier_uniform(P["decoder.attn.v.weight"])
        _gru_init(P, "decoder.gru", ("l0",))
        for head in ("decoder.time_sig_out", "decoder.key_out"):
            for i in (0, 2, 4):
                _default_linear_init(P, f"{head}.{i}")

    # ------------------------------------------------------------------ state access
    def _param_dict(self):
        return dict(self.named_parameters())

    def _buffer_dict(self):
        return dict(self.named_buffers())

    def flatten_(self):
        """Re-home every parameter as a view of ONE flat device buffer (order = state_dict order): lets the fused
        clip+Adadelta kernel and the data-parallel all-reduce treat the model as a single tensor.  Idempotent."""
        params = self._param_dict()
        dev = next(iter(params.values())).device
        from piano_a2s_amd.spec import flat_layout
        offs, total = flat_layout([params[n].numel() for n in self._names])      # every parameter on a 16-byte boundary
        if self._flat is not None and self._flat.device == dev:
            if all(params[n].data_ptr() == self._flat.data_ptr() + 4 * off for n, off in zip(self._names, offs)):
                return self._flat
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for n, off in zip(self._names, offs):
                p = params[n]
                flat[off:off + p.numel()].copy_(p.reshape(-1))
                p.data = flat[off:off + p.numel()].view(p.shape)
        self._flat = flat
        return flat

    def flat_layout(self):
        """[(offset in floats, shape)] of every parameter inside the flat buffer, in `self.parameters()` order (the index space of a
        torch optimizer's state_dict built from `modules.parameters()`)."""
        from piano_a2s_amd.spec import flat_layout
        params = self._param_dict()
        assert list(params) == list(self._names), "parameter registration order differs from the state_dict order"
        offs, _ = flat_layout([params[n].numel() for n in self._names])
        return [(off, tuple(params[n].shape)) for n, off in zip(self._names, offs)]

    # ------------------------------------------------------------------ forward (reference models.py:26-51)
    def forward(self,
                spectrogram,
                inference=True,
                ground_truth=None,
                teacher_forcing_ratio=0.,
                device=None):
        if inference:
            assert teacher_forcing_ratio == 0
            assert ground_truth is None
        self.device = device if device is not None else spectrogram.device
        params = self._param_dict()
        names = tuple(self._names)
        need_grad = self.training and torch.is_grad_enabled() and any(p.requires_grad for p in params.values())
        outs = _Transcribe.apply(self, spectrogram, inference, ground_truth, teacher_forcing_ratio, names, need_grad, *[params[n] for n in names])
        return outs


if __name__ == "__main__":
    # the reference's own smoke block (models.py:588-602), on the GPU
    dev = "cuda"
    model = ScoreTranscription().to(dev)
    print('Number of parameters: %d' % sum(p.numel() for p in model.parameters() if p.requires_grad))
    model.eval()
    outs = model(torch.randn(1, 1, 1200, 480, device=dev), device=dev)
    for name, o in zip(("time signature", "key", "upper staff", "lower staff"), outs):
        print(f"Shape of {name} predictions: ", o.shape)
