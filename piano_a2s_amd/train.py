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


def average_gradients(flat_g, world):
    """Data-parallel gradient exchange: SUM all-reduce of the flat gradient buffer, then / world (DDP semantics: every rank
    contributes the gradient of ITS minibatch mean).  Backend-agnostic: RCCL (nccl) on GPUs, gloo in the CPU tests."""
    if dist.is_available() and dist.is_initialized():
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
        # active whenever a process group exists -- also with a single rank (torchrun --nproc-per-node 1): the collectives then run
        # through RCCL exactly as with N ranks, which is what lets a 1-GPU box exercise the data-parallel path end to end
        self.active = dist.is_available() and dist.is_initialized()
        self.pending = []
        self.issued = 0

    def slice_ready(self, flat_g, start, end):
        if self.active and end > start:
            self.pending.append(dist.all_reduce(flat_g[start:end], op=dist.ReduceOp.SUM, async_op=True))
            self.issued += 1

    def finish(self, flat_g):
        for work in self.pending:
            work.wait()
        self.pending = []
        if self.active and self.world > 1:
            flat_g.div_(self.world)
        return flat_g


def broadcast_parameters(flat_p, src=0):
    """Make every replica start from rank `src`'s parameters (what DDP's constructor does)."""
    if dist.is_available() and dist.is_initialized():
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
    """clip_grad_norm_(max_grad_norm) + torch.optim.Adadelta(lr, rho, eps) over the model's flat parameter buffer.

    `layout` = [(offset, shape), ...] of the parameters inside the flat buffer, in `module.parameters()` order: with it,
    state_dict() / load_state_dict() speak torch.optim.Adadelta's own format ({"state": {i: {"step", "square_avg", "acc_delta"}},
    "param_groups": [...]}), so the `optimizer.ckpt` SpeechBrain's Brain writes for the reference recipe (init_optimizers registers
    the optimizer as a recoverable) and the one written here are interchangeable."""

    def __init__(self, flat_params, lr=1.0, rho=0.95, eps=1e-8, max_grad_norm=5.0, layout=None):
        self.p = flat_params
        self.lr, self.rho, self.eps, self.max_grad_norm = lr, rho, eps, max_grad_norm
        self.layout = layout
        self.steps = 0
        dev = flat_params.device
        self.square_avg = torch.zeros_like(flat_params)
        self.acc_delta = torch.zeros_like(flat_params)
        self.ctl = torch.zeros(3, dtype=torch.float32, device=dev)       # total norm, clip coef, applied flag
        self.nblocks = 1024
        self.partial = torch.empty(self.nblocks, dtype=torch.float64, device=dev)

    def step(self, flat_grads, loss_scalar=None, zero_grad=True):
        """The update is skipped ON THE DEVICE (ctl[2] = 0) when *loss_scalar or the gradient norm is not finite."""
        hip.check(hip.lib().a2s_clip_adadelta(hip.stream(), hip._p(self.p), hip._p(flat_grads), hip._p(self.square_avg), hip._p(self.acc_delta),
                                              C.c_long(self.p.numel()), hip._p(loss_scalar), hip.f32(self.max_grad_norm), hip.f32(self.lr), hip.f32(self.rho),
                                              hip.f32(self.eps), hip._p(self.ctl), hip._p(self.partial), self.nblocks, 1 if zero_grad else 0), "a2s_clip_adadelta")
        self.steps += 1

    @property
    def param_groups(self):          # so that sb.nnet.schedulers.update_learning_rate(optimizer, lr) works on this object too
        return [self]

    def __getitem__(self, key):      # param-group view of the hyper-parameters
        return {"lr": self.lr, "rho": self.rho, "eps": self.eps}[key]

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def state_dict(self):
        if self.layout is None:
            return {"square_avg": self.square_avg.cpu(), "acc_delta": self.acc_delta.cpu(), "lr": self.lr, "rho": self.rho, "eps": self.eps}
        state = {}
        for i, (off, shape) in enumerate(self.layout):
            n = 1
            for d in shape:
                n *= d
            state[i] = {"step": torch.tensor(float(self.steps)), "square_avg": self.square_avg[off:off + n].view(shape).clone(),
                        "acc_delta": self.acc_delta[off:off + n].view(shape).clone()}
        group = {"lr": self.lr, "rho": self.rho, "eps": self.eps, "weight_decay": 0, "foreach": None, "capturable": False, "maximize": False,
                 "differentiable": False, "params": list(range(len(self.layout)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if "state" in sd:                                   # torch.optim.Adadelta format (ours, or SpeechBrain's optimizer.ckpt)
            if sd["state"] and self.layout is None:
                raise ValueError("FusedAdadelta: a per-parameter optimizer state needs the parameter layout")
            for i, st in sd["state"].items():
                off, shape = self.layout[int(i)]
                n = st["square_avg"].numel()
                if tuple(st["square_avg"].shape) != tuple(shape):
                    raise ValueError(f"optimizer state {i}: shape {tuple(st['square_avg'].shape)} != parameter shape {tuple(shape)}")
                self.square_avg[off:off + n].copy_(st["square_avg"].reshape(-1))
                self.acc_delta[off:off + n].copy_(st["acc_delta"].reshape(-1))
                self.steps = int(float(st.get("step", self.steps)))
            g = sd["param_groups"][0]
            self.lr, self.rho, self.eps = g.get("lr", self.lr), g.get("rho", self.rho), g.get("eps", self.eps)
        else:                                               # flat format of round 1
            self.square_avg.copy_(sd["square_avg"])
            self.acc_delta.copy_(sd["acc_delta"])
            self.lr = sd.get("lr", self.lr)


def reserve_pools(device, batch, frames=1201, scale=1.0):
    """Reserve the caching allocator's pools for the working set of the fused step BEFORE the first step: one block per stream the step allocates on,
    allocated and handed straight back -- the allocator keeps the segment and carves every later request of that stream out of it (it keeps one
    pool per stream, and the step allocates on four).  Without it the pools grow by hipMalloc for as long as new (minibatch, coin) shapes keep turning
    up -- ~10 steps at 256 clips (tools/alloc_by_stream.py: 224, 28, 0, 0, 1, 9, 0, 3, 4, 0 ... segments per step), each a driver call of 50-130 ms
    under load.  Sizes: what those pools hold after 24 steps at 256 clips x 1201 frames (default 155 GiB, lower-staff stream 27, long-clip groups 6-7),
    plus 3 %, scaled by the clips x frames of the caller, in blocks of 24 GiB.  Returns the GiB reserved."""
    # (about three quarters of what the pools end up holding: the warm-up steps grow them to their final size -- reserving all of it up front left the
    # allocator at 266 of the MI355X's 268 GiB once the shapes of 20 steps had been seen)
    per = {"default": 115.0, "side0": 1.0, "side1": 22.0, "group1": 5.0, "group2": 6.0}
    f = scale * (batch / 256.0) * (frames / 1201.0)
    streams = {"default": torch.cuda.current_stream(device), "side0": engine.side_streams(device)[0], "side1": engine.side_streams(device)[1],
               "group1": engine.group_stream(device, 1), "group2": engine.group_stream(device, 2)}
    total = 0.0
    torch.cuda.synchronize(device)
    torch.cuda.empty_cache()                     # (what earlier work left cached belongs to other shapes: the reservation replaces it)
    chunk = 24.0 * max(f, 0.25)          # GiB per block: the step's largest tensors (the 40-channel activations, 22 GiB at 256 clips) fit in one; a single
    held = []                              # allocation of the whole default pool (160 GiB) is refused by the runtime
    for name, gib in sorted(per.items(), key=lambda kv: kv[1]):          # small pools first: the big one takes what is left
        want = gib * f
        with torch.cuda.stream(streams[name]):
            while want > 0.25:
                n = int(min(want, chunk) * 2 ** 30)
                free, _ = torch.cuda.mem_get_info(device)
                if n > free - (12 << 30):
                    break
                try:
                    held.append(torch.empty(n, dtype=torch.uint8, device=device))
                except RuntimeError:
                    break
                total += n / 2 ** 30
                want -= n / 2 ** 30
    del held                               # back to the allocator: the segments stay cached, one pool per stream
    torch.cuda.synchronize(device)
    return total


def plan_clip_groups(until_up, until_lo, max_tail_frac=0.5, min_gain=0.08, step_cost=150.0, jump=1.3, max_candidates=8):
    """Cut a minibatch into [ordinary clips | long clips] for Engine.forward's clip groups.

    until_up / until_lo: (B, bars) int arrays, decode steps each (clip, bar) row needs (last real target + 1).  Cost model of a group
    of clips, in units of one clip's attention pass (~0.7 us on MI355X): a decode step costs max(step_cost, rows still active) --
    bandwidth-bound while many rows are active, a latency floor (~100 us of dependent small kernels) once only a few long rows
    remain -- summed over the steps of the longer staff of every bar.  Two groups run CONCURRENTLY but share the HBM, so a cut costs
    max(cost of either group, attention work of both).  Candidate cuts: the places where the clips' longest rows, sorted, jump by
    `jump`x.  The best cut is taken if it beats the uncut minibatch by min_gain.  Returns (order, n_main): `order` = clip permutation
    (ordinary clips first, sorted by their longest row, longest first; the long clips in their original order), n_main = size of the
    first group (== B: do not split)."""
    import numpy as np
    up, lo = np.asarray(until_up, dtype=np.int64), np.asarray(until_lo, dtype=np.int64)
    B = up.shape[0]
    ident = np.arange(B)
    if B < 4:
        return ident, B

    def cost(sel):
        total, work = 0.0, 0.0
        for bar in range(up.shape[1]):
            u, l = up[sel, bar], lo[sel, bar]
            n = int(max(u.max(), l.max()))
            act = np.zeros(n, dtype=np.float64)
            for v in (u, l):
                act += len(v) - np.cumsum(np.bincount(v, minlength=n + 1))[:n]         # rows with until > t
            total += np.maximum(act, step_cost).sum()
            work += act.sum()
        return total, work

    longest = np.maximum(up.max(1), lo.max(1))
    by_len = np.argsort(-longest, kind="stable")                 # longest clips first
    srt = longest[by_len].astype(np.float64)
    kmax = max(1, int(B * max_tail_frac))
    ratio = srt[:kmax] / np.maximum(srt[1:kmax + 1], 1.0)        # jump between the k-th longest clip and the next
    cands = [int(k) + 1 for k in np.argsort(-ratio, kind="stable")[:max_candidates] if ratio[k] >= jump]
    if not cands:
        return ident, B
    whole, _ = cost(ident)
    best_k, best = None, (1.0 - min_gain) * whole
    for k in cands:
        ct, wt = cost(by_len[:k])
        cm, wm = cost(by_len[k:])
        c = max(ct, cm, wt + wm)
        if c < best:
            best_k, best = k, c
    if best_k is None:
        return ident, B
    # inside the ordinary group the clips with the longest rows come first (by_len is a stable descending sort): late in a decoder call
    # only a prefix of the clips is still running (engine.active_rows: m_active); the long clips keep their original order
    return np.concatenate([by_len[best_k:], np.sort(by_len[:best_k])]), B - best_k


def split_long_group(long_ids, until_up, until_lo, segments, min_gain=0.1):
    """Round 5.  The long-clip group's decode chain is, per bar SEGMENT of the step, as long as the longest row ANY of its clips has there -- typically a
    full-length (398-step) upper bar in every segment, although each clip holds only one such bar.  Cut it by where the clips' longest bars lie:
    sub-group A = clips whose longest upper bar is in segments 0..k, sub-group B = the rest; each then runs 398 steps only in "its" segments.  The two
    sub-groups get one stream each (engine.group_stream) and decode their staves one after the other, so a sub-group's chain is the SUM of its
    staves' steps per segment, against the MAXIMUM for the single group (staves side by side).  Returns (A, B) -- lists of clip ids -- for the
    best k when max(chain A, chain B) < (1 - min_gain) x the single group's chain, else None.  Loss, gradients and update do not depend on the
    grouping (clip groups are independent: Engine.forward)."""
    import numpy as np
    long_ids = list(long_ids)
    if len(long_ids) < 2 or len(segments) < 2:
        return None
    up, lo = np.asarray(until_up), np.asarray(until_lo)

    def chain(ids, sequential):
        total = 0
        for seg in segments:
            u, l = int(up[np.ix_(ids, seg)].max()), int(lo[np.ix_(ids, seg)].max())
            total += (u + l) if sequential else max(u, l)
        return total
    seg_of_bar = {bar: si for si, seg in enumerate(segments) for bar in seg}
    where = [seg_of_bar[int(up[c].argmax())] for c in long_ids]
    single = chain(long_ids, False)
    best = None
    for k in range(len(segments) - 1):
        a = [c for c, w in zip(long_ids, where) if w <= k]
        b = [c for c, w in zip(long_ids, where) if w > k]
        if not a or not b:
            continue
        m = max(chain(a, True), chain(b, True))
        if best is None or m < best[0]:
            best = (m, a, b)
    if best is not None and best[0] < (1.0 - min_gain) * single:
        return best[1], best[2]
    return None


def plan_step_groups(gt_host, bars, max_length, rng, teacher_forcing_ratio, group_plan=None, long_subgroups=True):
    """The host's clip-group plan of one step, from the targets alone: gt_host = (upper, lower, ...) host tensors (B, bars, len).  Returns (order,
    group_cuts, host_plan): the clip permutation, the contiguous clip ranges of the groups in that order -- [ordinary | long] or, when the long
    clips' longest bars lie in different bar segments of THIS step's coins, [ordinary | long A | long B] (split_long_group) -- and the coins if
    they had to be drawn for that (engine.draw_plan: the reference's draw order; Engine.forward then takes them as host_plan), else None.
    TrainStep._step calls it; tests/golden/make_golden.py uses it to choose fixtures that exercise the three-group path."""
    import numpy as _np
    up, lo = gt_host[0], gt_host[1]
    idx_u = torch.arange(1, up.shape[-1] + 1)
    idx_l = torch.arange(1, lo.shape[-1] + 1)
    until_u, until_l = ((up != PAD).long() * idx_u).amax(-1).numpy(), ((lo != PAD).long() * idx_l).amax(-1).numpy()
    order, n_main = plan_clip_groups(until_u, until_l, **(group_plan or {}))
    B = up.shape[0]
    group_cuts, host_plan = [(0, n_main), (n_main, B)], None
    if n_main < B and long_subgroups:
        # the coins of the step are drawn HERE, in the reference's order: the cut looks at the bar segments
        host_plan = engine.draw_plan(gt_host, bars, max_length, rng, teacher_forcing_ratio)
        sub = split_long_group(order[n_main:].tolist(), until_u, until_l, engine.plan_segments(host_plan, bars, True))
        if sub is not None:
            order = _np.concatenate([order[:n_main], _np.asarray(sub[0], dtype=order.dtype), _np.asarray(sub[1], dtype=order.dtype)])
            group_cuts = [(0, n_main), (n_main, n_main + len(sub[0])), (n_main + len(sub[0]), B)]
    return order, group_cuts, host_plan


class TrainStep:
    """model: models.ScoreTranscription on a GPU.  One call = one optimizer step on one minibatch."""

    def __init__(self, model, lr=1.0, rho=0.95, eps=1e-8, max_grad_norm=5.0, dropout=True, sync_bn=None, skip_finished_rows=None,
                 fuse_bars=None, clip_groups=None, group_plan=None):
        """skip_finished_rows (default on; A2S_SKIP_FINISHED=0 turns it off): the note decoders skip the attention of rows whose
        remaining targets are all <pad>.  fuse_bars (default on with the former; A2S_FUSE_BARS=0 turns it off): consecutive bars
        whose bar-level input is teacher-forced are decoded in one call (Engine.forward).  Loss, gradients and the update are
        unchanged (skipped rows are ignore_index positions and nothing else reads them); only `last_outputs` positions whose target
        is <pad> differ from the reference's values.  clip_groups (default on with fuse_bars; A2S_CLIP_GROUPS=0 turns it off): the
        clips holding exceptionally long bars decode as a group of their own, concurrently with the ordinary ones (plan_clip_groups;
        Engine.forward) -- the minibatch is permuted for that, which no loss term, gradient or statistic depends on.  group_plan: keyword
        overrides of plan_clip_groups' cost model (e.g. dict(step_cost=4.0) makes a 12-clip minibatch split the way a 256-clip one does with
        the default 150: tests/test_gpu_g4.py)."""
        self.model = model
        # The host side of a step is a few hundred small CPU tensor operations (the decode plan).  Above ~32 k elements torch runs each of them
        # as an OpenMP region over every core of the box (256 here): one descheduled worker stalls the region, and the thread waiting for it
        # holds the interpreter lock -- measured as 70-90 ms freezes of ALL issuing threads in one step out of ten (tools/host_stalls.py).
        # One intra-op thread is plenty for this work.  A2S_HOST_THREADS overrides (0: leave torch's setting alone).  The setting is process-wide
        # in torch, so it is applied for the duration of a step only and the caller's value is restored afterwards (__call__): CPU work the
        # user does between steps (a CPU front end, metrics) keeps its threads.
        self.host_threads = int(_os.environ.get("A2S_HOST_THREADS", "1"))
        self.sync_bn = (_os.environ.get("A2S_SYNC_BN") == "1") if sync_bn is None else bool(sync_bn)
        self.skip_finished_rows = (_os.environ.get("A2S_SKIP_FINISHED", "1") != "0") if skip_finished_rows is None else bool(skip_finished_rows)
        self.fuse_bars = (_os.environ.get("A2S_FUSE_BARS", "1") != "0") if fuse_bars is None else bool(fuse_bars)
        # True / False, or an explicit list of contiguous clip ranges [(0, n), (n, B)] (tests: no planner, no permutation)
        self.clip_groups = (_os.environ.get("A2S_CLIP_GROUPS", "1") != "0") if clip_groups is None else clip_groups
        # each clip group's decoder backward follows its forward at once (see __call__); False: forward of every group, then the objective,
        # then backward of every group
        self.pipeline_groups = True
        self.group_plan = dict(group_plan or {})
        # the ConvStack is enqueued before the host plans the decoder (see _step); False: after, as in round 4
        self.early_convstack = True
        # the note decoders' weight gradients beside the encoder's back-propagation instead of behind each call's reverse loop (engine_bwd.Backward)
        self.late_wgrads = True
        # the long-clip group cut in two by the bar segment of each clip's longest bar (split_long_group); False: one long-clip group
        self.long_subgroups = True
        self.keep_grads = False            # tests: keep the last step's gradient views (name -> tensor) in self.last_grads
        self.last_grads = None
        self._last = None
        self.flat = model.flatten_()
        self.opt = FusedAdadelta(self.flat, lr, rho, eps, max_grad_norm, layout=model.flat_layout())
        self.objective = Objective(self.flat.device)
        self.total = torch.zeros(1, dtype=torch.float32, device=self.flat.device)
        self.dropout = dropout
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.collectives = 0               # gradient all-reduces issued so far (0 without a process group)
        self.decode_steps = 0              # note-decoder steps the last call executed (the straggler term of data parallelism)
        self.time_exchange = False         # bench: measure how long the step's stream waits for the gradient all-reduce
        self.exchange_wait_ms = []

    def state(self):
        S = dict(self.model.named_parameters())
        S.update(dict(self.model.named_buffers()))
        return {k: (v.data if isinstance(v, torch.nn.Parameter) else v) for k, v in S.items()}

    def __call__(self, batch, teacher_forcing_ratio, rng=_py_random):
        """batch: the reference's 9-tuple (device tensors).  Returns the (4,2) device tensor of loss terms (col 0)."""
        prev = torch.get_num_threads()
        scoped = self.host_threads > 0 and prev != self.host_threads
        if scoped:
            torch.set_num_threads(self.host_threads)
        try:
            return self._step(batch, teacher_forcing_ratio, rng)
        finally:
            if scoped:
                torch.set_num_threads(prev)

    def _step(self, batch, teacher_forcing_ratio, rng):
        spectrogram, ts_t, key_t, up_t, up_len, lo_t, lo_len = batch[:7]
        eng = engine.Engine(self.model.cfg, sync_bn=self.sync_bn)
        eng.skip_finished_rows = self.skip_finished_rows
        eng.fuse_bars = self.skip_finished_rows and self.fuse_bars
        eng.late_wgrads = self.late_wgrads
        # the persistent kernels' abort latch: looked at here through the asynchronous read the previous step left behind (hip.post_persist_abort_read)
        if spectrogram.is_cuda:
            hip.poll_persist_abort(spectrogram.device)
            eng.abort_check = False
        gt_host, perm, host_plan = None, None, None
        S = self.state()
        conv_pre = None
        plan_groups = eng.fuse_bars and not isinstance(self.clip_groups, (list, tuple)) and self.clip_groups
        if plan_groups and self.early_convstack and spectrogram.is_cuda:
            # Round 5: the ConvStack does not depend on the decoder's plan (nor on the clip order), so it is ENQUEUED FIRST -- before the host
            # reads the targets and cuts the clip groups (4-5 ms during which the GPU used to idle at the start of every step).  The targets come
            # over on a side stream, so that the read does not wait for the ConvStack just enqueued on this one.  It DOES wait for everything the
            # caller had enqueued before the step (`uploads`, recorded ahead of the ConvStack): the real trainer uploads the batch with
            # non_blocking copies from pinned memory on this stream (recipe._to_device), behind the spectrogram's DMA or the online VQT --
            # without the wait the side stream could read the targets before they land, and the host plan would disagree with the device's.
            # The same wait bounds the host's run-ahead to one step (the event completes when the previous step's last kernel has): that is
            # what engine._PinnedPool's two alternating staging areas rely on.
            uploads = torch.cuda.Event()
            uploads.record()
            if self.sync_bn and eng.sync_bn:
                engine.Engine.check_counts(spectrogram.shape[0] * spectrogram.shape[2] * spectrogram.shape[3], spectrogram.device)
            Bc, _, Tc, _ = spectrogram.shape
            mask = (torch.rand((Bc * Tc, self.model.cfg["conv_feature_size"]), device=spectrogram.device) >= 0.2).to(torch.uint8) if self.dropout else None
            conv_out, conv_saved = eng.convstack(S, spectrogram, True, mask)
            # (one of the step's four streams -- a fifth would share a hardware queue with one of them anyway, engine.group_stream; the work the
            # previous step left on it finished before that step's ConvStack backward, i.e. before `uploads`)
            copy_stream = engine.side_streams(spectrogram.device)[0]
            srcs = (up_t, lo_t, up_len, lo_len)
            key = tuple((tuple(t.shape), t.dtype) for t in srcs)
            if getattr(self, "_gt_pinned_key", None) != key:              # pinned staging, allocated once per batch shape
                self._gt_pinned = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in srcs]
                self._gt_pinned_key = key
            copy_stream.wait_event(uploads)
            with torch.cuda.stream(copy_stream):
                for dst, t in zip(self._gt_pinned, srcs):
                    dst.copy_(t, non_blocking=True)
                    t.record_stream(copy_stream)
                done = torch.cuda.Event()
                done.record()
            done.synchronize()
            hip.poll_persist_abort(spectrogram.device)                    # (the previous step's latch read has landed by now)
            gt_host = [p.clone() for p in self._gt_pinned]
            conv_pre = [conv_out, conv_saved, None]
        if eng.fuse_bars and isinstance(self.clip_groups, (list, tuple)):
            eng.clip_groups = [tuple(r) for r in self.clip_groups]
        elif eng.fuse_bars and self.clip_groups:
            if gt_host is None:
                gt_host = [up_t.cpu(), lo_t.cpu(), up_len.cpu(), lo_len.cpu()]       # the one host sync of the step (Engine.forward reuses it)
            cfg_ = self.model.cfg
            order, group_cuts, host_plan = plan_step_groups(gt_host, cfg_["max_bars"], cfg_["max_length"], rng, teacher_forcing_ratio, self.group_plan,
                                                            self.long_subgroups)
            B = up_t.shape[0]
            n_main = group_cuts[0][1]
            if n_main < B:
                perm = torch.from_numpy(order)
                gt_host = [t[perm] for t in gt_host]
                pd = perm.to(spectrogram.device, non_blocking=True)
                if conv_pre is not None:
                    # the ConvStack ran in the caller's clip order: its output goes into group order, its backward gets the gradient back in the caller's
                    inv = torch.empty_like(perm)
                    inv[perm] = torch.arange(perm.numel())
                    conv_pre[0] = conv_pre[0].index_select(0, pd)
                    conv_pre[2] = inv.to(spectrogram.device, non_blocking=True)
                    ts_t, key_t, up_t, up_len, lo_t, lo_len = [t.index_select(0, pd) for t in (ts_t, key_t, up_t, up_len, lo_t, lo_len)]
                else:
                    spectrogram, ts_t, key_t, up_t, up_len, lo_t, lo_len = [t.index_select(0, pd) for t in (spectrogram, ts_t, key_t, up_t, up_len, lo_t, lo_len)]
                eng.clip_groups = group_cuts
        if self.sync_bn and eng.sync_bn and conv_pre is None:
            engine.Engine.check_counts(spectrogram.shape[0] * spectrogram.shape[2] * spectrogram.shape[3], spectrogram.device)
        exchange = GradientExchange(self.world)
        fwd = dict(inference=False, ground_truth=[ts_t, key_t, up_t, up_len, lo_t, lo_len], teacher_forcing_ratio=teacher_forcing_ratio, training=True,
                   rng=rng, dropout=self.dropout, gt_host=gt_host, conv_pre=tuple(conv_pre) if conv_pre is not None else None, host_plan=host_plan)
        if eng.fuse_bars and self.pipeline_groups:
            # Pipelined clip groups.  The gradient of the 4-term objective wrt a row's log-probabilities is -1/count at its target --
            # and the counts (denominators of the NLL means) are functions of the TARGETS alone.  So a clip group does not have to
            # wait for the others' forward passes before it back-propagates: its loss gradients and decoder backward are chained
            # right behind its decoder forward, on its own thread and streams (Engine.group_hook).  With [ordinary | long] clip groups
            # the long clips' ~400-step latency chains (forward AND backward) then run under the ordinary clips' forward + backward
            # instead of under their forward only.  The loss VALUES are reduced after the join.
            B, bars = ts_t.shape
            groups = getattr(eng, "clip_groups", None) or [(0, B)]
            cfg, dev = self.model.cfg, spectrogram.device
            U, Lo = cfg["max_length"]
            lay = lambda t: torch.cat([t[b0:b1].transpose(0, 1).reshape(-1) for b0, b1 in groups])
            up_lay, lo_lay = lay(up_t), lay(lo_t)
            inv = torch.zeros((4, 2), dtype=torch.float32, device=dev)
            inv[0:2, 1] = 1.0 / (B * bars)
            inv[2, 1] = 1.0 / (up_t != PAD).sum()
            inv[3, 1] = 1.0 / (lo_t != PAD).sum()
            gouts = [torch.zeros((B, bars, cfg["num_time_sig"]), device=dev), torch.zeros((B, bars, cfg["num_keys"]), device=dev),
                     torch.zeros((bars, B, U, VOCAB_SIZE), device=dev), torch.zeros((bars, B, Lo, VOCAB_SIZE), device=dev)]
            ctx = engine_bwd.Backward(eng, S, (B, spectrogram.shape[2], spectrogram.shape[3]), dev, groups, True, True)
            L = hip.lib()

            def grads_and_backward(gidx, gs):
                b0, b1 = gs["range"]
                n = b1 - b0
                d = [gouts[0][b0:b1], gouts[1][b0:b1], engine.group_views(gouts[2], groups, gidx), engine.group_views(gouts[3], groups, gidx)]
                tg = [ts_t[b0:b1], key_t[b0:b1], up_lay[bars * U * b0: bars * U * b1], lo_lay[bars * Lo * b0: bars * Lo * b1]]
                for i, (V_, ign) in enumerate(((cfg["num_time_sig"], -1), (cfg["num_keys"], -1), (VOCAB_SIZE, PAD), (VOCAB_SIZE, PAD))):
                    hip.check(L.a2s_nll_grad(hip.stream(), hip._p(d[i]), hip._p(tg[i]), C.c_void_p(inv.data_ptr() + 8 * i), hip.f32(1.0),
                                             C.c_long(d[i].numel() // V_), V_, C.c_longlong(ign)), "a2s_nll_grad")
                ctx.decoder_group(gidx, gs, *d)

            eng.group_hook = grads_and_backward
            outs = eng.forward(S, spectrogram, **fwd)
            groups = eng.clip_groups_used
            losses, _ = self.objective(outs, (ts_t, key_t, up_lay, lo_lay), want_grad=False)
            torch.sum(losses[:, 0], dim=0, keepdim=True, out=ctx.flat_full[ctx.total:])      # the loss word of the gradient buffer (see finish)
            G = ctx.finish(grad_ready=exchange.slice_ready)
        else:
            outs = eng.forward(S, spectrogram, **fwd)
            groups = eng.clip_groups_used
            if eng.bar_major:
                # fused bars: the staff outputs come bar-major, one contiguous (bars, clips, len, V) block per clip group; the loss is a
                # mean over rows, so the targets are simply laid out the same way
                lay = lambda t: torch.cat([t[b0:b1].transpose(0, 1).reshape(-1) for b0, b1 in groups])
                losses, gouts = self.objective(outs, (ts_t, key_t, lay(up_t), lay(lo_t)))
            else:
                losses, gouts = self.objective(outs, (ts_t, key_t, up_t, lo_t))
            G = engine_bwd.backward(eng, S, gouts, grad_ready=exchange.slice_ready, loss_total=losses[:, 0].sum())
        self.decode_steps = sum(seg["staff"][k][2]["steps"] for g in eng.saved["groups"] for seg in g["segments"] for k in ("up", "lo"))
        self.attn_clip_steps = sum(eng.attn_clip_steps)        # forward; the backward streams the same pairs once more
        self.attn_shared_clip_steps = sum(eng.attn_shared_clip_steps)      # ... of which: encoder outputs read by a pass shared between the staves
        if self.time_exchange and exchange.active:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            flat_g = exchange.finish(G[None])
            e1.record()
            self._exchange_events = getattr(self, "_exchange_events", []) + [(e0, e1)]
        else:
            flat_g = exchange.finish(G[None])
        self.collectives += exchange.issued
        torch.sum(losses[:, 0], dim=0, keepdim=True, out=self.total)          # total loss stays on the device
        # every replica must take the SAME skip / apply decision (reference check_gradients looks at the local loss only, which under data
        # parallelism lets one rank skip while the others apply): the gate is the sum of all ranks' losses -- finite iff every rank's is.  It
        # travelled as the last word of the first gradient slice (engine_bwd.Backward.flat_full), so it costs no collective of its own;
        # non-finite gradients reach every rank through the all-reduce and gate via the norm.
        gate = G["__loss_gate__"] if exchange.active else self.total
        self.opt.step(flat_g, gate, zero_grad=False)
        self.last_grads = {k: v for k, v in G.items() if isinstance(k, str) and not k.startswith("__")} if self.keep_grads else None
        # the engine, its group hook (a closure over the backward context) and the backward context (which holds the engine) form a reference
        # cycle: left to the cyclic collector, ~3.3 GiB of per-step gradient buffers stayed allocated for several steps and the caching
        # allocator answered with a fresh 11 GiB segment every third step -- a hipMalloc of 340 ms under load, with every other host thread
        # queued behind the allocator's lock (tools/alloc_stalls.py).  Break the cycle here.
        eng.saved = None
        eng.group_hook = None
        self._keep_alive, eng._keep_alive = getattr(eng, "_keep_alive", None), None      # (pinned staging of this step: released by the next one)
        self._last = (outs, eng.bar_major, groups, perm)
        if spectrogram.is_cuda:
            hip.post_persist_abort_read(spectrogram.device)
        return losses

    @property
    def last_outputs(self):
        """The four log-probability tensors of the last step in the reference's layout and the caller's clip order."""
        if self._last is None:
            return None
        outs, bar_major, groups, perm = self._last
        outs = list(outs)
        if bar_major:
            outs[2], outs[3] = (engine.gather_group_views(o, groups).transpose(0, 1) for o in outs[2:])
        if perm is not None:
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(perm.numel())
            inv = inv.to(outs[0].device)
            outs = [o.index_select(0, inv) for o in outs]
        return tuple(outs)

    @last_outputs.setter
    def last_outputs(self, value):
        self._last = None if value is None else (value, False, [(0, value[0].shape[0])], None)

    def report(self):
        """Host copy of the last step's [time-sig, key, upper, lower loss, applied flag] -- ONE small device-to-host read (the
        reference recipe does four .cpu() reads per step, pretrain.py:90-93)."""
        return torch.cat([self.objective.losses[:, 0], self.opt.ctl[2:3]]).tolist()
