"""The fused training step of the recipe on the HIP path (reference ASR.fit_batch, pretrain.py:121-129):

    forward  ->  4-term NLL objective (+ its gradient)  ->  backward  ->  [data-parallel all-reduce]
             ->  check_gradients (finite loss, clip_grad_norm_ 5.0)  ->  Adadelta step  ->  zero_grad

Everything between the batch arriving on the device and the updated parameters is liba2s_hip.so work on one stream;
there is no host synchronisation in the step except the one the forward needs to read the ground-truth token rows
(to plan the per-step control flow) -- the loss value is left on the device and only fetched when asked for.
Data parallelism: one process per GPU, gradients summed with ONE all-reduce over the flat gradient buffer (RCCL when the
process group's backend is nccl; gloo on CPU in tests) and divided by the world size, as DDP does.
"""
import ctypes as C
import random as _py_random

import torch
import torch.distributed as dist

from . import engine, engine_bwd, hip
from .spec import PAD, VOCAB_SIZE


import os as _os
_FORCE_COLLECTIVES = _os.environ.get("A2S_FORCE_DIST") == "1"      # debug: run the collectives even with a single rank


def average_gradients(flat_g, world):
    """Data-parallel gradient exchange: SUM all-reduce of the flat gradient buffer, then / world (DDP semantics: every rank
    contributes the gradient of ITS minibatch mean).  Backend-agnostic: RCCL (nccl) on GPUs, gloo in the CPU tests."""
    if world > 1 or (dist.is_available() and dist.is_initialized() and world == 1 and _FORCE_COLLECTIVES):
        dist.all_reduce(flat_g, op=dist.ReduceOp.SUM)
        if world > 1:
            flat_g.div_(world)
    return flat_g


class GradientExchange:
    """The same exchange, overlapped with the backward pass: engine_bwd.backward announces contiguous slices of the flat gradient
    buffer as soon as they are final (decoder + encoder parameters first -- 89 % of the bytes -- while the ConvStack backward, 40 %
    of the step, is still to run; then the ConvStack slice), each slice is all-reduced asynchronously on the collective's own stream,
    and finish() waits for all of them and divides by the world size.  One all-reduce per slice, bit-identical to average_gradients."""

    def __init__(self, world):
        self.world = world
        self.active = world > 1 or (dist.is_available() and dist.is_initialized() and world == 1 and _FORCE_COLLECTIVES)
        self.pending = []

    def slice_ready(self, flat_g, start, end):
        if self.active and end > start:
            self.pending.append(dist.all_reduce(flat_g[start:end], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self, flat_g):
        for work in self.pending:
            work.wait()
        self.pending = []
        if self.active and self.world > 1:
            flat_g.div_(self.world)
        return flat_g


def broadcast_parameters(flat_p, src=0):
    """Make every replica start from rank `src`'s parameters (what DDP's constructor does)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE_COLLECTIVES):
        dist.broadcast(flat_p, src=src)
    return flat_p


class Objective:
    """reference compute_objectives: NLLLoss on time signature and key, NLLLoss(ignore_index=<pad>) on both staves; total = sum."""

    def __init__(self, device):
        self.dev = device
        self.nblocks = 256
        self.partial = torch.empty(2 * self.nblocks, dtype=torch.float64, device=device)
        self.losses = torch.zeros((4, 2), dtype=torch.float32, device=device)     # per term: loss, 1/count

    def __call__(self, outs, targets, want_grad=True):
        """outs: 4 log-prob tensors; targets: (ts, key, upper, lower) int64.  Returns (losses (4,2) device tensor, grads or None)."""
        L = hip.lib()
        grads = [torch.zeros_like(o) for o in outs] if want_grad else [None] * 4
        for i, (o, t, ign) in enumerate(zip(outs, targets, (-1, -1, PAD, PAD))):
            V = o.shape[-1]
            rows = o.numel() // V
            t = t.contiguous()
            hip.check(L.a2s_nll_loss(hip.stream(), hip._p(o), hip._p(t), C.c_long(rows), V, C.c_longlong(ign), C.c_void_p(self.losses.data_ptr() + 8 * i),
                                     hip._p(grads[i]), hip.f32(1.0), hip._p(self.partial), self.nblocks), "a2s_nll_loss")
        return self.losses, (grads if want_grad else None)


class FusedAdadelta:
    """clip_grad_norm_(max_grad_norm) + torch.optim.Adadelta(lr, rho, eps) over the model's flat parameter buffer."""

    def __init__(self, flat_params, lr=1.0, rho=0.95, eps=1e-8, max_grad_norm=5.0):
        self.p = flat_params
        self.lr, self.rho, self.eps, self.max_grad_norm = lr, rho, eps, max_grad_norm
        dev = flat_params.device
        self.square_avg = torch.zeros_like(flat_params)
        self.acc_delta = torch.zeros_like(flat_params)
        self.ctl = torch.zeros(3, dtype=torch.float32, device=dev)       # total norm, clip coef, applied flag
        self.nblocks = 1024
        self.partial = torch.empty(self.nblocks, dtype=torch.float64, device=dev)

    def step(self, flat_grads, loss_scalar=None, zero_grad=True):
        hip.check(hip.lib().a2s_clip_adadelta(hip.stream(), hip._p(self.p), hip._p(flat_grads), hip._p(self.square_avg), hip._p(self.acc_delta),
                                              C.c_long(self.p.numel()), hip._p(loss_scalar), hip.f32(self.max_grad_norm), hip.f32(self.lr), hip.f32(self.rho),
                                              hip.f32(self.eps), hip._p(self.ctl), hip._p(self.partial), self.nblocks, 1 if zero_grad else 0), "a2s_clip_adadelta")

    def state_dict(self):
        return {"square_avg": self.square_avg, "acc_delta": self.acc_delta, "lr": self.lr, "rho": self.rho, "eps": self.eps}

    def load_state_dict(self, sd):
        self.square_avg.copy_(sd["square_avg"])
        self.acc_delta.copy_(sd["acc_delta"])
        self.lr = sd.get("lr", self.lr)


class TrainStep:
    """model: models.ScoreTranscription on a GPU.  One call = one optimizer step on one minibatch."""

    def __init__(self, model, lr=1.0, rho=0.95, eps=1e-8, max_grad_norm=5.0, dropout=True, sync_bn=None, skip_finished_rows=None,
                 fuse_bars=None):
        """skip_finished_rows (default on; A2S_SKIP_FINISHED=0 turns it off): the note decoders skip the attention of rows whose
        remaining targets are all <pad>.  fuse_bars (default on with the former; A2S_FUSE_BARS=0 turns it off): consecutive bars
        whose bar-level input is teacher-forced are decoded in one call (Engine.forward).  Loss, gradients and the update are
        unchanged (skipped rows are ignore_index positions and nothing else reads them); only `last_outputs` positions whose target
        is <pad> differ from the reference's values."""
        self.model = model
        self.sync_bn = (_os.environ.get("A2S_SYNC_BN") == "1") if sync_bn is None else bool(sync_bn)
        self.skip_finished_rows = (_os.environ.get("A2S_SKIP_FINISHED", "1") != "0") if skip_finished_rows is None else bool(skip_finished_rows)
        self.fuse_bars = (_os.environ.get("A2S_FUSE_BARS", "1") != "0") if fuse_bars is None else bool(fuse_bars)
        self.flat = model.flatten_()
        self.opt = FusedAdadelta(self.flat, lr, rho, eps, max_grad_norm)
        self.objective = Objective(self.flat.device)
        self.total = torch.zeros(1, dtype=torch.float32, device=self.flat.device)
        self.dropout = dropout
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def state(self):
        S = dict(self.model.named_parameters())
        S.update(dict(self.model.named_buffers()))
        return {k: (v.data if isinstance(v, torch.nn.Parameter) else v) for k, v in S.items()}

    def __call__(self, batch, teacher_forcing_ratio, rng=_py_random):
        """batch: the reference's 9-tuple (device tensors).  Returns the (4,2) device tensor of loss terms (col 0)."""
        spectrogram, ts_t, key_t, up_t, up_len, lo_t, lo_len = batch[:7]
        eng = engine.Engine(self.model.cfg, sync_bn=self.sync_bn)
        eng.skip_finished_rows = self.skip_finished_rows
        eng.fuse_bars = self.skip_finished_rows and self.fuse_bars
        S = self.state()
        outs = eng.forward(S, spectrogram, inference=False, ground_truth=[ts_t, key_t, up_t, up_len, lo_t, lo_len],
                           teacher_forcing_ratio=teacher_forcing_ratio, training=True, rng=rng, dropout=self.dropout)
        if eng.bar_major:            # fused bars: the staff outputs come bar-major (bars, B, len, V); the loss is a mean over rows
            losses, gouts = self.objective(outs, (ts_t, key_t, up_t.transpose(0, 1), lo_t.transpose(0, 1)))
            outs = (outs[0], outs[1], outs[2].transpose(0, 1), outs[3].transpose(0, 1))
        else:
            losses, gouts = self.objective(outs, (ts_t, key_t, up_t, lo_t))
        exchange = GradientExchange(self.world)
        G = engine_bwd.backward(eng, S, gouts, grad_ready=exchange.slice_ready)
        flat_g = exchange.finish(G[None])
        torch.sum(losses[:, 0], dim=0, keepdim=True, out=self.total)          # total loss stays on the device
        self.opt.step(flat_g, self.total, zero_grad=False)
        eng.saved = None
        self.last_outputs = outs
        return losses
