"""Host-side orchestration of the transcription forward pass on liba2s_hip.so (device work = HIP kernels only).

Mirrors ``ScoreTranscription.forward`` of the reference (models.py:26-51) -> ConvStack (:523-543) -> Encoder
(:75-82) -> HierarchicalDecoder.decode_bars (:191-316).  PyTorch is used for device memory and the stream;
every arithmetic operation is a call into the C ABI (piano_a2s_amd.hip).  Control flow that the reference
evaluates on the host per step (teacher-forcing coin flips, EOS bookkeeping, early break) is resolved ONCE
per forward into a plan when ground truth is given, and on the device (with sparse polling) in greedy mode.
"""
import ctypes as C
import random as _py_random
import threading

import torch

from . import hip
from .spec import EOS, PAD, SOS, VOCAB_SIZE


def plan_note_steps(gt_rows, max_steps):
    """Host replay of the reference's loop bookkeeping for one (bar, staff) with ground truth
    (models.py:388-419): returns (steps executed, lengths per row).

    The loop breaks at the first t where every row has shown <eos> in gt[:, :t]; lengths[b] is overwritten by
    every <eos> seen at t < steps (so it ends as the LAST such position + 1), default max_steps.
    """
    B = gt_rows.shape[0]
    is_eos = gt_rows == EOS                                      # (B, max_steps) CPU bool
    first = torch.where(is_eos.any(1), is_eos.float().argmax(1), torch.full((B,), max_steps))
    steps = int(first.max()) + 1 if bool((first < max_steps).all()) else max_steps
    steps = min(steps, max_steps)
    idx = torch.arange(1, steps + 1, dtype=torch.long)
    last = (is_eos[:, :steps].long() * idx).amax(dim=1)          # last <eos> position + 1 among the executed steps, 0 = none
    lengths = torch.where(last > 0, last, torch.full((B,), max_steps, dtype=torch.long))
    return steps, lengths


import os as _os

# hipGraph replay of the greedy decoder (configs[4] of BASELINE.json asks for it): built and parity-tested, but measured no faster
# than eager launches on MI355X (B=8: 0.232 s vs 0.212 s for 2223 steps) -- the step is bound by the GPU-side execution of ~13 small
# kernels, not by host launch cost -- so it is opt-in.
_GREEDY_GRAPH = _os.environ.get("A2S_GREEDY_GRAPH") == "1"
# the encoder's input projections and the attention key images on the two-term fp16 split (operand ranges known: DESIGN.md section 5)
# instead of three bf16 terms (round 2's)
_ENC_TWO_TERM = True
# late steps of a large decoder call: the per-step products only on the leading clips that still have an unfinished row (False: all rows)
_TAIL_PREFIX = True
# ... and the few-row step kernels on the rows still running once those fit them (False: only calls that are small as a whole)
_TAIL_ROWS = True
# the (step, row) pairs of a decoder call that still ran, as a flat index list: the backward's weight-gradient products contract over those only
# (False: over every row of every step, three quarters of which are the exact zeros of finished rows at the bench's lengths)
_LIVE_ROWS = True
_SIDE_STREAMS = {}


def _dev_index(device):
    return torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()


def side_streams(device, group=0):
    """Two extra HIP streams per device (and clip group): the upper- and lower-staff note decoders of a bar are independent given the
    bar summary (reference models.py:261-275 runs them one after the other), so their step loops are enqueued on separate streams and
    overlap -- one staff's latency-bound kernels (skinny GEMMs, gates, epilogues) run under the other's bandwidth-bound attention."""
    key = (_dev_index(device), group)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = (torch.cuda.Stream(device=key[0]), torch.cuda.Stream(device=key[0]))
    return _SIDE_STREAMS[key]


class _PinnedPool:
    """Two pinned staging areas per device, used alternately by consecutive forward passes (the uploads of pass k have long completed when
    pass k + 2 starts: every pass ends with the caller's stream waiting for all of its streams, and the training step synchronises once
    per step)."""

    def __init__(self, nbytes=8 << 20):
        self.bufs = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.cur, self.off = 0, 0

    def next_pass(self):
        self.cur ^= 1
        self.off = 0

    def take(self, nbytes, dtype, shape):
        off = (self.off + 63) & ~63
        if nbytes == 0 or off + nbytes > self.bufs[self.cur].numel():
            return None
        self.off = off + nbytes
        return self.bufs[self.cur][off:off + nbytes].view(dtype).view(shape)


_PINNED_POOLS = {}


def _pinned_pool(device):
    key = _dev_index(device)
    if key not in _PINNED_POOLS:
        _PINNED_POOLS[key] = _PinnedPool()
    _PINNED_POOLS[key].next_pass()
    return _PINNED_POOLS[key]


_BAR_ATTN_SPLIT = True          # the bar-level decoder's attention of groups of >= 32 clips on the split-T kernels (tests switch it)
_BAR_ATTN_WS = {}
# the round-4 persistent note decoder for the last clip group also while another group decodes beside it (tests only: it owns every CU it runs on
# and stops the bulk group there -- 540 against 514 ms per step, profiles/r04_dec_persist_beside_bulk.txt)
_PERSIST_BESIDE = False
_PAIR_STAVES = True             # group 0's two staves issued by one host loop, their sweeps sharing the encoder-output reads (Engine._decode_pair)


def bar_attn_workspace(device, group, n_clips, T, H):
    """Workspace of the split-T attention kernels for the BAR-level decoder's attention of one clip group (forward and backward; one per (device,
    group, shape), zero-initialised once as the kernels require, never shared with the note decoders' calls, which may run beside it).  None: the
    one-workgroup-per-clip kernels (small groups, other widths, engine._BAR_ATTN_SPLIT = False)."""
    if not _BAR_ATTN_SPLIT or H != 256 or n_clips < 32:
        return None
    key = (_dev_index(device), group, n_clips, T)
    ws = _BAR_ATTN_WS.get(key)
    if ws is None:
        if len(_BAR_ATTN_WS) > 64:
            _BAR_ATTN_WS.clear()
        ws = _BAR_ATTN_WS[key] = hip.attn_workspace(n_clips, T, H, device)
    return ws


_HOST_TRACE = None        # tools/phase_times.py --segments: a list that receives (label, host time) at the decoder's host-side milestones


def _trace(label):
    if _HOST_TRACE is not None:
        import time
        _HOST_TRACE.append((label, time.time()))


_GROUP_STREAMS = {}
_GROUP_POOL = None


def staff_streams(device, group):
    """(upper, lower) streams of a clip group's note decoders, inside the four-stream budget (group_stream): group 0 = the caller's
    current (default) stream + side stream 1, the long-clip group = its own stream (current in its thread) + side stream 0.  The
    upper staff's step loop is issued by a worker thread INTO the stream the group's bar-level chain and heads also use -- the loop
    is the long pole (398 vs 189 steps), the bar-level kernels interleave with it -- so each group's lower staff gets a hardware queue
    of its own and the long-clip group's two staves run side by side instead of one after the other (1292 -> 876 dependent steps)."""
    cur = torch.cuda.current_stream()
    side = side_streams(device, 0)
    return (cur, side[1]) if group == 0 else (cur, side[0])


def group_stream(device, group):
    """The ONE stream everything of clip group `group` > 0 runs on (group 0: the caller's current stream + side_streams).

    Stream budget.  The HIP runtime multiplexes every stream of a process onto GPU_MAX_HW_QUEUES = 4 hardware queues (a new stream
    joins the least-shared queue) and two streams on one queue execute in order.  Measured on MI355X with the long-clip group on
    streams of its own: 308 clips/s with 4 queues, 245 with 6, 224 with 16 (more hardware queues than 4 are time-sliced), and 241 with
    high-priority streams (a second pool of queues: same oversubscription).  So the step uses exactly FOUR streams: the default one,
    the two side streams (staves of group 0; encoder directions) and this one -- the long-clip group's latency chain gets a hardware
    queue to itself instead of waiting in line behind the bulk group's bandwidth-bound kernels.  The encoder's weight-gradient
    GEMMs reuse it (engine_bwd._weight_grad_stream): the decoder is done by then."""
    if group >= 2:
        # round 5: two long-clip sub-groups (train.split_long_group), each decoding its two staves one after the other on ONE stream: the
        # first on the stream above, the second on the side stream the single long-clip group's lower staff would have used -- still four streams
        return side_streams(device, 0)[0]
    key = (_dev_index(device), 1)
    if key not in _GROUP_STREAMS:
        _GROUP_STREAMS[key] = torch.cuda.Stream(device=key[0])
    return _GROUP_STREAMS[key]


def staves_concurrent(gidx, n_groups):
    """Do the two note decoders of clip group gidx run side by side on two streams?  Group 0 always; the long-clip group when it is the only one beside
    it; with two long-clip sub-groups each of them runs upper then lower on its one stream (four streams in all, see group_stream).  Round 6
    re-measured the alternative -- the sub-groups' lower staves on two more streams: 442 -> 456 ms per step on the four hardware queues (the
    extra streams share queues with the others and wait in line), 493 with GPU_MAX_HW_QUEUES=6 or 8 (profiles/r06_long_staves_streams.txt)."""
    return gidx == 0 or (gidx == 1 and n_groups == 2)


def draw_plan(gt_cpu, bars, maxlen, rng, teacher_forcing_ratio):
    """The host plan of a forward pass with ground truth: executed steps of every (bar, staff) and EVERY coin of the reference's protocol, drawn in the
    reference's order (one per executed note step, upper then lower, then one per bar: models.py:404,289).  gt_cpu: (upper, lower, ...) host tensors
    (B, bars, len); the step counts are batch-wide maxima, so the clip order does not matter."""
    plan = []
    for bar in range(bars):
        p = {}
        for gi_idx in (0, 1):
            steps, _ = plan_note_steps(gt_cpu[gi_idx][:, bar, :], maxlen[gi_idx])
            p[gi_idx] = (steps, [rng.random() < teacher_forcing_ratio for _ in range(steps)])
        p["tf"] = rng.random() < teacher_forcing_ratio
        plan.append(p)
    return plan


def plan_segments(plan, bars, fuse, inference=False):
    """Bars decoded in one call each: consecutive bars whose predecessor's bar-level coin says "teacher-force" (at most 5 = A2S_ATTN_MAX_GROUPS)."""
    segments = [[0]]
    for bar in range(1, bars):
        if fuse and plan[bar - 1]["tf"] and not inference and len(segments[-1]) < 5:
            segments[-1].append(bar)
        else:
            segments.append([bar])
    return segments


def group_views(t, clip_groups, gidx):
    """Bar-major staff tensor (bars, B, len, V) as ONE flat buffer in which every clip group owns a contiguous bar-major block:
    returns group gidx's block as (bars, clips of the group, len, V).  With a single group that is `t` itself."""
    bars, B = t.shape[0], t.shape[1]
    b0, b1 = clip_groups[gidx]
    per_clip = t[0, 0].numel()
    return t.view(-1)[bars * per_clip * b0: bars * per_clip * b1].view((bars, b1 - b0) + tuple(t.shape[2:]))


def gather_group_views(t, clip_groups):
    """Inverse view of group_views: the flat-by-group buffer back as an ordinary (bars, B, len, V) tensor (a copy when there are groups)."""
    if len(clip_groups) == 1:
        return t
    return torch.cat([group_views(t, clip_groups, g) for g in range(len(clip_groups))], dim=1)


def run_clip_groups(device, fns):
    """fns[g]() = the decoder work of clip group g.  Group 0 runs on the calling thread and its current stream; every other group on
    a host thread of its own with group_stream(device, g) current (forked from the caller's stream, joined to it before returning) --
    its step loops are issued by that thread's own two issue threads (fork_on_streams).  Returns [fns[g]() results]."""
    global _GROUP_POOL
    if len(fns) == 1:
        return [fns[0]()]
    fork = torch.cuda.Event()
    fork.record()
    dev_index = _dev_index(device)

    def task(g, fn):
        torch.cuda.set_device(dev_index)
        st = group_stream(device, g)
        st.wait_event(fork)
        with torch.cuda.stream(st):
            r = fn()
            done = torch.cuda.Event()
            done.record()
        return r, done

    if _GROUP_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _GROUP_POOL = ThreadPoolExecutor(max_workers=3, thread_name_prefix="a2s-group")
    futures = [_GROUP_POOL.submit(task, g, fn) for g, fn in enumerate(fns[1:], start=1)]
    first = fns[0]()
    rest = [f.result() for f in futures]
    for _, done in rest:
        torch.cuda.current_stream().wait_event(done)
    return [first] + [r for r, _ in rest]


def encoder_streams(device, batch=0):
    """Streams for the two directions of an encoder GRU layer.
    batch: clips of the call.  The persistent recurrences (csrc/a2s_persist.hip) need every workgroup of a launch resident: 16 per 16 clips, two per
    CU.  Up to 256 clips both directions fit side by side (2 x 256 workgroups on 256 CUs); above that (up to 512 clips) ONE direction fits, so the two run
    one after the other on the current stream and the library is told that nothing persistent runs beside a launch ("gru_persist_alone")."""
    serial = batch > 256
    hip.check(hip.lib().a2s_debug_set(b"gru_persist_alone", 1 if batch > 256 else 0), "a2s_debug_set")
    if serial:
        cur = torch.cuda.current_stream()
        return (cur, cur)
    return side_streams(device)


_ISSUE_POOL = None


def fork_on_streams(device, streams, fns):
    """Run fns[i]() with streams[i] current, each in its own host thread, forked from the caller's current stream.  The step loops
    these functions enqueue (encoder directions, the two staves) are bound by the HOST's launch rate (~6 us per kernel, measured:
    two loops issued from one thread take exactly twice one loop), and ctypes drops the GIL inside liba2s_hip.so, so two threads
    issue two streams at the same time.  Returns a join() callable: it returns [fns[i]() results] and makes the caller's current
    stream wait for both."""
    global _ISSUE_POOL
    fork = torch.cuda.Event()
    fork.record()
    dev_index = _dev_index(device)
    caller_stream = torch.cuda.current_stream()

    def task(st, fn):
        torch.cuda.set_device(dev_index)
        if st == caller_stream:                # the caller's own stream (stream order does the fork / join): just issue from this thread
            with torch.cuda.stream(st):
                return fn(), None
        st.wait_event(fork)                    # everything the loop reads was enqueued before the fork
        with torch.cuda.stream(st):
            r = fn()
        done = torch.cuda.Event()
        done.record(st)
        return r, done

    if _ISSUE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _ISSUE_POOL = ThreadPoolExecutor(max_workers=6, thread_name_prefix="a2s-issue")      # 2 staves x up to 3 clip groups
    futures = [_ISSUE_POOL.submit(task, st, fn) for st, fn in zip(streams, fns)]

    def join(wait=True):
        """wait=False: the caller's stream is NOT made to wait; returns (results, [events to wait for later])."""
        res = [f.result() for f in futures]
        events = [done for _, done in res if done is not None]
        if not wait:
            return [r for r, _ in res], events
        for done in events:
            torch.cuda.current_stream().wait_event(done)
        return [r for r, _ in res]
    return join


def _dist_world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 0     # 0 = no process group


class Engine:
    def __init__(self, cfg, sync_bn=False):
        self.cfg = cfg
        self.attn_clip_steps = []
        self.attn_shared_clip_steps = []       # ... of which: pairs served by a pass shared between the two staves (Engine._decode_pair)
        self.poll = 16            # greedy decode: host looks at the device-side done counter every `poll` steps
        # Synchronised BatchNorm (what SpeechBrain's DDP wrapping gives the reference): batch statistics over the
        # GLOBAL minibatch -- per-channel (sum, sum of squares, count) are all-reduced between the ranks.  Off by default:
        # per-rank statistics (plain DDP semantics).  Needs an initialised process group.
        self.sync_bn = bool(sync_bn) and _dist_world() >= 1

    def _global_stats(self, partial, nblocks, C_, count):
        """(nblocks, C, 2) per-block partial sums of this rank -> (1, C, 2) sums of ALL ranks, global element count.
        The count is world x the local count, on the host: train.TrainStep has verified at the start of the step that every rank holds the
        same batch shape (Engine.check_counts), so no device-to-host read stalls the forward pass five times per step."""
        import torch.distributed as dist
        sums = torch.empty(2 * C_, dtype=torch.float32, device=partial.device)
        hip.check(hip.lib().a2s_col_sum(hip.stream(), hip._p(partial), C.c_long(2 * C_), hip._p(sums), C.c_long(nblocks), 2 * C_,
                                        hip.f32(1.0), hip.f32(0.0), C.c_void_p(0), C.c_size_t(0)), "a2s_col_sum (bn stats)")
        dist.all_reduce(sums)
        return sums, float(count) * dist.get_world_size()

    @staticmethod
    def check_counts(count, dev):
        """world x local count is the global count only when every rank holds the same number of frames (DistributedSampler pads the index
        list; a user DataLoader's uneven last batch does not).  ONE tiny collective per step, issued and read at the start of the step -- the
        step begins with a host read of the targets anyway, nothing is in flight yet -- instead of one queued collective per BatchNorm layer
        (round 3: five extra all-reduces per step and, as the advisor noted, a blocking read in the middle of the ConvStack)."""
        import torch.distributed as dist
        t = torch.tensor([float(count), -float(count)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        hi, lo = float(t[0]), -float(t[1])
        if hi != lo:
            raise RuntimeError(f"synchronised BatchNorm: the ranks hold different numbers of frames in this step ({lo:.0f} .. {hi:.0f}); "
                               "give every rank the same batch shape (DistributedSampler pads; drop or pad an uneven last batch)")

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def _empty(*shape, dev, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=dev)

    def _bn(self, S, name, partial, nblocks, C_, count, training):
        dev = S[name + ".weight"].device
        mean, invstd, scale, shift = (self._empty(C_, dev=dev) for _ in range(4))
        L = hip.lib()
        self.bn_counts = getattr(self, "bn_counts", {})
        if training and self.sync_bn:
            partial, count = self._global_stats(partial, nblocks, C_, count)
            nblocks = 1
        self.bn_counts[name] = count
        hip.check(L.a2s_bn_finalize(hip.stream(), hip._p(partial), nblocks, C_, C.c_double(count), hip._p(S[name + ".weight"]),
                                    hip._p(S[name + ".bias"]), hip._p(S[name + ".running_mean"]), hip._p(S[name + ".running_var"]),
                                    hip._p(S[name + ".num_batches_tracked"]), hip._p(mean), hip._p(invstd), hip._p(scale),
                                    hip._p(shift), hip.f32(1e-5), hip.f32(0.1), 1 if training else 0), "a2s_bn_finalize")
        return mean, invstd, scale, shift

    # ------------------------------------------------------------------ ConvStack
    def convstack(self, S, spec_in, training, drop_mask=None):
        L = hip.lib()
        B, _, T, F = spec_in.shape
        dev = spec_in.device
        x = spec_in.contiguous()
        chans = [(1, 20), (20, 20), (20, 40), (40, 40)]
        scale = shift = None
        saved = {"x0": x, "y": [], "bn": [], "yabs": []}
        yabs = None                          # per-channel max |y| of the previous layer's output, written by the launch that produced it
        for i, (ci, co) in enumerate(chans, start=1):
            y = self._empty(B, T, co, F, dev=dev)
            nblk = L.a2s_conv3x3_stat_blocks(B, T, F, ci)
            partial = self._empty(nblk, co, 2, dev=dev) if training else None
            cws = hip.conv_workspace(ci, dev)
            # the operand of layer i+1 is relu(scale_c y_c + shift_c): |scale_c| max|y_c| + |shift_c| bounds it, and the row-streaming
            # kernel scales channel c by the matching power of two (nothing is clamped whatever BatchNorm's gamma is: DESIGN.md section 5)
            yabs_out = self._empty(co, dev=dev)
            hip.conv3x3_forward(x.view(B, T, ci, F), S[f"convstack.conv{i}.weight"], y, scale, shift, partial, cws, yabs, yabs_out)
            mean, invstd, scale, shift = self._bn(S, f"convstack.bn{i}", partial, nblk, co, float(B) * T * F, training)
            saved["y"].append(y)
            saved["bn"].append((mean, invstd, scale, shift))
            saved["yabs"].append(yabs_out)
            # device scalar bounding relu(bn_i(y_i)) over the tensor: operand range of the kernels that re-form it on the fly (this layer's
            # consumers in the backward pass -- weight gradient of layer i + 1 --, the 19200 -> 256 Linear)
            saved.setdefault("abound", []).append(hip.act_bound(scale, shift, yabs_out))
            x, yabs = y, yabs_out
        # the (B*T, 40*F) operand of the 19200->256 Linear is relu(bn4(y4)): formed while the GEMM stages its A tiles (column k belongs
        # to channel k // F), never written to memory
        y4 = x.view(B * T, 40 * F)
        Cf = self.cfg["conv_feature_size"]
        a4 = None
        # two-term fp16 split: activations scaled by the power of two of their bound, the weights by that of max|W|
        saved["w_out_amax"] = hip.absmax(S["convstack.out.weight"])
        z = hip.linear_forward(y4, S["convstack.out.weight"], (scale, shift, F), saved["abound"][3], saved["w_out_amax"])
        rows = B * T
        rpb = 64
        nblk = (rows + rpb - 1) // rpb
        partial = None
        if training:
            partial = self._empty(nblk, Cf, 2, dev=dev)
            hip.check(L.a2s_col_stats(hip.stream(), hip._p(z), hip._p(partial), C.c_long(rows), Cf, rpb), "a2s_col_stats")
        mean, invstd, scale, shift = self._bn(S, "convstack.out_bn", partial, nblk, Cf, float(rows), training)
        out = self._empty(rows, Cf, dev=dev)
        hip.check(L.a2s_bn1d_relu_dropout(hip.stream(), hip._p(z), hip._p(out), hip._p(scale), hip._p(shift), hip._p(drop_mask),
                                          hip.f32(1.0 / 0.8), C.c_long(out.numel()), Cf), "a2s_bn1d_relu_dropout")
        saved.update(a4=a4, z=z, out_bn=(mean, invstd, scale, shift), drop=drop_mask)
        return out.view(B, T, Cf), saved

    # ------------------------------------------------------------------ Encoder
    def encoder(self, S, x, training):
        L = hip.lib()
        B, T, _ = x.shape
        H = self.cfg["hidden_size"]
        dev = x.device
        saved = {"layers": []}
        gws = [hip.gemm_workspace(B, dev), hip.gemm_workspace(B, dev)]
        streams = encoder_streams(dev, B)        # the two directions of a layer are independent 1201-step chains: one stream each
        inp = x.reshape(B * T, -1)
        finals = []
        for layer in (0, 1):
            out = self._empty(B, T, 2 * H, dev=dev)
            lsave = {"in": inp, "dirs": []}
            # input projections on the two-term fp16 split: layer 0 reads the ConvStack features (max measured), layer 1 GRU states (< 1)
            in_amax = (hip.absmax(inp) if layer == 0 else hip.one(dev)) if _ENC_TWO_TERM else None
            gis = [hip.linear(inp, S[f"encoder.gru.weight_ih_{sfx}"], S[f"encoder.gru.bias_ih_{sfx}"],
                              two_term=(in_amax, hip.absmax(S[f"encoder.gru.weight_ih_{sfx}"])) if _ENC_TWO_TERM else None)      # (B*T, 3H) per direction
                   for sfx in (f"l{layer}", f"l{layer}_reverse")]
            def direction(d, sfx, gi):
                hbuf = self._empty(2, B, H, dev=dev)
                gh = self._empty(B, 3 * H, dev=dev)
                hn = self._empty(B, H, dev=dev)
                gates = self._empty(T, B, 4 * H, dev=dev) if training else None
                hip.check(L.a2s_gru_seq_fwd(hip.stream(), hip._p(gi), C.c_long(T * 3 * H), C.c_long(3 * H),
                                            hip._p(S[f"encoder.gru.weight_hh_{sfx}"]), hip._p(S[f"encoder.gru.bias_hh_{sfx}"]),
                                            C.c_void_p(out.data_ptr() + 4 * d * H), C.c_long(T * 2 * H), C.c_long(2 * H),
                                            hip._p(hbuf), hip._p(gh), hip._p(gates), hip._p(hn), B, T, H, d, hip._p(gws[d]),
                                            C.c_size_t(gws[d].numel() * 4)), "a2s_gru_seq_fwd")
                return {"gi": gi, "gates": gates, "hn": hn, "scratch": (hbuf, gh)}

            join = fork_on_streams(dev, streams, [lambda d=d, sfx=sfx: direction(d, sfx, gis[d])
                                                  for d, sfx in enumerate((f"l{layer}", f"l{layer}_reverse"))])
            for rec in join():
                finals.append(rec["hn"])
                lsave["dirs"].append(rec)
            lsave["out"] = out
            saved["layers"].append(lsave)
            inp = out.view(B * T, 2 * H)
        # bridge: hidden = [tanh(fc([hf0;hr0])) | tanh(fc([hf1;hr1]))]   (models.py:78-81); fc applied as two
        # half-width contractions so the concatenation never materialises
        W, bias = S["encoder.fc.weight"], S["encoder.fc.bias"]
        hidden = self._empty(B, 2 * H, dev=dev)
        for l in (0, 1):
            hf, hr = finals[2 * l], finals[2 * l + 1]
            hip.gemm(hf, H, 1, W, 1, 2 * H, hidden, 2 * H, B, H, H, bias=bias, c_off=l * H)
            hip.gemm(hr, H, 1, W, 1, 2 * H, hidden, 2 * H, B, H, H, beta=1.0, act=2, b_off=H, c_off=l * H)
        saved["finals"] = finals
        saved["hidden"] = hidden
        return saved["layers"][1]["out"], hidden, saved

    # ------------------------------------------------------------------ decoder pieces
    def _keys(self, S, prefix, enc2d, H):
        """Key image exp(2 K), K = enc W_e^T with W_e = attn.weight[:, 2H:]  (step-invariant half of the attention Linear); the
        attention kernels form tanh(K + q) as 1 - 2 / (1 + exp(2K) exp(2q)) -- see include/a2s.h."""
        W = S[prefix + ".attn.weight"]
        K = self._empty(enc2d.shape[0], H, dev=enc2d.device)
        # (encoder outputs are GRU states, |enc| < 1; max |W| over the whole attention Linear bounds its key half)
        hip.gemm(enc2d, 2 * H, 1, W, 1, 4 * H, K, H, enc2d.shape[0], H, 2 * H, b_off=2 * H, act=3,
                 two_term=(hip.one(enc2d.device), hip.absmax(W)) if _ENC_TWO_TERM else None)
        return K

    def _staff_token(self, S, ids, lengths, len_stride, out, col0, maxlen, id_bstride, ids_are_i64, record=None):
        """record: list that receives what the backward pass needs to replay this call (training only)."""
        L = hip.lib()
        names = [f"decoder.staff_emb.{w}_{sfx}" for sfx in ("l0", "l0_reverse") for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        arr = (C.c_void_p * 8)(*[S[n].data_ptr() for n in names])
        R = out.shape[0]
        E, Sz = self.cfg["note_emb_size"], self.cfg["staff_emb_size"]
        hsave = torch.empty((R, 2, maxlen, Sz), dtype=torch.float32, device=out.device) if record is not None else None
        hip.check(L.a2s_staff_emb_fwd(hip.stream(), hip._p(S["decoder.note_emb.weight"]), arr,
                                      hip._p(ids) if ids_are_i64 else C.c_void_p(0), C.c_void_p(0) if ids_are_i64 else hip._p(ids),
                                      C.c_long(id_bstride), hip._p(lengths), C.c_long(len_stride), hip._p(out), C.c_long(out.stride(0)),
                                      col0, hip._p(hsave), R, maxlen, E, Sz), "a2s_staff_emb_fwd")
        if record is not None:
            record.append(dict(ids=ids, lengths=lengths, len_stride=len_stride, col0=col0, maxlen=maxlen, id_bstride=id_bstride,
                               i64=ids_are_i64, hsave=hsave))

    def _decode_staff(self, S, prefix, keys, enc, h0, max_steps, probs_bar, gt_bar, steps, tf_flags, training, drop_p, B, T, attn_ws=None, gemm_ws=None,
                      active=None, drop=None, flags_dev=None, persist=False, defer_launch=False):
        """One NoteDecoder.decode_notes call over B rows.  probs_bar: view (B, max_steps, V) of the output tensor (strided).
        tf_flags: per step, bit g = teacher-force the rows of group g.
        active: optional dict(until: (B,) int32, order / rank: (n_clips,) int32 device tensors; n_active: host int array per step;
        n_clips) -- the rows are B/n_clips fused bars of the same clips and row r's attention is skipped from step until[r] on (see
        Engine.forward: skip_finished_rows / fuse_bars)."""
        L = hip.lib()
        H, E, V = self.cfg["hidden_size"], self.cfg["note_emb_size"], VOCAB_SIZE
        H2, ldx = 2 * H, E + 2 * H
        dev = enc.device
        n = steps
        graph = gt_bar is None and not training and getattr(self, "greedy_graph", _GREEDY_GRAPH)     # greedy decode: replayed hipGraph
        t_base = torch.zeros(1, dtype=torch.int32, device=dev) if graph else None
        # (with row_list the tail steps write only the rows still running; a2s_note_decoder_fwd zero-fills these buffers itself -- from C,
        # where a memset that has to wait for room in a busy stream's queue does not hold the interpreter lock)
        h = self._empty(n + 1, B, H2, dev=dev)
        h[0].copy_(h0)
        x = self._empty(n + 1, B, ldx, dev=dev)
        q = self._empty(n, B, H, dev=dev)
        o = self._empty(n, B, 2 * H2, dev=dev)
        gates = self._empty(n, B, 4 * H2, dev=dev) if training else None
        attw = self._empty(n, B, T, dev=dev) if training else None
        gh, gi = self._empty(B, 3 * H2, dev=dev), self._empty(B, 3 * H2, dev=dev)
        logits = self._empty(B, V, dev=dev)
        ids = torch.zeros((B, max_steps), dtype=torch.int32, device=dev)
        eos_seen = torch.zeros(B, dtype=torch.int32, device=dev)
        lengths = torch.full((B,), max_steps, dtype=torch.long, device=dev)
        n_done = torch.zeros(1, dtype=torch.int32, device=dev)
        steps_exec = torch.zeros(1, dtype=torch.int32, device=dev)
        if drop is None and training and drop_p > 0:
            drop = (torch.rand((n + 1, B, E), device=dev) >= drop_p).to(torch.uint8)
        # SOS token embedding -> x[0][:, :E]
        hip.check(L.a2s_embed_rows(hip.stream(), hip._p(S[prefix + ".embedding.weight"]), C.c_void_p(0), C.c_void_p(0), C.c_long(0), SOS,
                                   hip._p(x), C.c_long(ldx), 0, B, E, hip._p(drop), hip.f32(1.0 / (1.0 - drop_p) if drop is not None else 1.0)),
                  "a2s_embed_rows")
        flags = (C.c_uint8 * max(n, 1))(*([int(f) for f in tf_flags] if tf_flags is not None else [0] * n))
        a = hip.NoteDecArgs()
        for name, t in (("attn_w", S[prefix + ".attn.attn.weight"]), ("attn_b", S[prefix + ".attn.attn.bias"]),
                        ("attn_v", S[prefix + ".attn.v.weight"]), ("w_ih", S[prefix + ".gru.weight_ih_l0"]),
                        ("w_hh", S[prefix + ".gru.weight_hh_l0"]), ("b_ih", S[prefix + ".gru.bias_ih_l0"]),
                        ("b_hh", S[prefix + ".gru.bias_hh_l0"]), ("out_w", S[prefix + ".out.weight"]), ("out_b", S[prefix + ".out.bias"]),
                        ("emb", S[prefix + ".embedding.weight"]), ("keys", keys), ("enc", enc), ("h", h), ("x", x), ("q", q),
                        ("gates", gates), ("attw", attw), ("o", o), ("gh", gh), ("gi", gi), ("logits", logits),
                        ("argmax_out", ids), ("eos_seen", eos_seen), ("lengths", lengths), ("n_done", n_done), ("steps_exec", steps_exec), ("drop", drop), ("attn_ws", attn_ws), ("gemm_ws", gemm_ws), ("t_base", t_base),
                        ("clip_order", active and active["order"]), ("clip_rank", active and active["rank"]),
                        ("row_until", active and active["until"])):
            setattr(a, name, t.data_ptr() if t is not None else None)
        a.gemm_ws_bytes = gemm_ws.numel() * 4 if gemm_ws is not None else 0
        step_ws = hip.step_workspace(H, E, dev)
        a.step_ws, a.step_ws_floats = step_ws.data_ptr(), step_ws.numel()
        a.n_active = C.cast(active["n_active"], C.c_void_p).value if active else None
        a.n_clips = active["n_clips"] if active else 0
        a.m_active = C.cast(active["m_active"], C.c_void_p).value if (active and active.get("m_active") is not None) else None
        if active and active.get("row_list") is not None:
            a.row_list, a.n_rows_active = active["row_list"].data_ptr(), C.cast(active["n_rows_active"], C.c_void_p).value
        else:
            a.row_list, a.n_rows_active = None, None
        a.probs, a.probs_bstride = probs_bar.data_ptr(), probs_bar.stride(0)
        if gt_bar is not None:
            a.gt, a.gt_bstride = gt_bar.data_ptr(), gt_bar.stride(0)
        else:
            a.gt, a.gt_bstride = None, 0
        a.tf_flags = C.cast(flags, C.c_void_p).value
        a.inv_keep = 1.0 / (1.0 - drop_p) if drop is not None else 1.0
        a.am_bstride = max_steps
        a.R, a.T, a.H, a.E, a.V, a.steps, a.poll, a.eos_id = B, T, H, E, V, n, (self.poll if gt_bar is None else 0), EOS
        a.use_graph = 1 if graph else 0
        # few clips: the whole call as ONE persistent launch (csrc/a2s_dec_persist.hip).  Only when the caller says so: two such launches must
        # never be in flight together (each wants every CU's LDS; see decode_group)
        persist_ws = None
        a.tf_flags_dev, a.persist_ws, a.persist_ws_bytes = None, None, 0
        if persist and n > 0 and (gt_bar is not None or not training) and (tf_flags is None or flags_dev is not None):
            nb_ws = L.a2s_note_decoder_persist_ws_bytes(active["n_clips"] if active else B, B, n)
            if nb_ws:
                persist_ws = torch.empty(nb_ws, dtype=torch.uint8, device=dev)
                a.persist_ws, a.persist_ws_bytes = persist_ws.data_ptr(), nb_ws
                a.tf_flags_dev = flags_dev.data_ptr() if flags_dev is not None else None
        def finish(launched):
            # steps the reference would have executed: known from the plan with ground truth; read back from the device
            # in greedy mode (launched steps can overshoot the early break by < poll; those were no-ops)
            executed = n if gt_bar is not None else int(steps_exec.item())
            if gt_bar is None and a.persist_ws:
                hip.check_persist_abort(dev, raise_error=True)       # (the host has just synchronised: a persistent launch that gave up returns unusable ids)
            saved = dict(h=h, x=x, q=q, o=o, gates=gates, attw=attw, drop=drop, steps=executed, launched=launched, ids=ids, flags=list(flags),
                         gt_bar=gt_bar, prefix=prefix, max_steps=max_steps, drop_p=drop_p, attn_ws=attn_ws, gemm_ws=gemm_ws, active=active, flags_dev=flags_dev,
                         step_ws=step_ws, persist_ws=persist_ws, logits=logits, gh=gh, gi=gi, eos_seen=eos_seen, n_done=n_done)
            return ids, lengths, saved
        if defer_launch:                      # _decode_pair: the arguments are ready, the caller launches both staves with one call
            return a, finish
        done = C.c_int(0)
        hip.check(L.a2s_note_decoder_fwd(hip.stream(), C.byref(a), C.byref(done)), "a2s_note_decoder_fwd")
        return finish(done.value)

    def _decode_pair(self, calls, streams, pair):
        """The two NoteDecoder calls of a segment (models.py:261-275) issued by ONE host loop: calls = the argument tuples of _decode_staff for the
        upper and the lower staff, streams = their streams, pair = dict(order, rank: device int32 (n_clips,), n_active: host int array) -- the
        clip bookkeeping of both staves together.  While both staves run, a decode step's attention sweep is one launch that reads the encoder
        outputs once (a2s_note_decoder_fwd_pair)."""
        prepared = []
        for st, args in zip(streams, calls):
            with torch.cuda.stream(st):
                prepared.append(self._decode_staff(*args, defer_launch=True))
        (au, fin_u), (al, fin_l) = prepared
        du, dl = C.c_int(0), C.c_int(0)
        hip.check(hip.lib().a2s_note_decoder_fwd_pair(C.c_void_p(streams[0].cuda_stream), C.c_void_p(streams[1].cuda_stream), C.byref(au), C.byref(al),
                                                      hip._p(pair["order"]), hip._p(pair["rank"]), pair["n_active"], C.byref(du), C.byref(dl)),
                  "a2s_note_decoder_fwd_pair")
        return [fin_u(du.value), fin_l(dl.value)]

    # ------------------------------------------------------------------ full forward
    def forward(self, S, spectrogram, inference=True, ground_truth=None, teacher_forcing_ratio=0.0, training=False,
                rng=_py_random, dropout=True, gt_host=None, conv_pre=None, host_plan=None):
        """S: dict name -> device tensor (parameters and BN buffers, reference state_dict names).
        gt_host: optional host copies (upper, lower, upper_len, lower_len) of the ground truth, when the caller already has them.
        conv_pre: optional (conv_out (B, T, Cf), saved, unperm) of a ConvStack pass the caller has ALREADY enqueued on this stream for the same clips
        (train.TrainStep launches it before it plans the decoder: the planning then runs under it); `spectrogram` is then only looked at for its
        shape.  unperm: None, or the int64 device index that maps this call's clip order back to the order the ConvStack ran in (its backward
        then gets its gradient in that order: engine_bwd).
        host_plan: optional result of draw_plan(...) for this very call (train.TrainStep draws it before it cuts the clip groups: the cut looks at the
        bar segments); the coins are then NOT drawn again here."""
        if inference:
            assert teacher_forcing_ratio == 0 and ground_truth is None     # models.py:202-204
        if not spectrogram.is_cuda:
            raise hip.A2SError("Engine.forward needs device tensors: the transcription hot path has no CPU implementation")
        L = hip.lib()
        cfg = self.cfg
        H, E, Sz = cfg["hidden_size"], cfg["note_emb_size"], cfg["staff_emb_size"]
        te, ke, bars = cfg["time_sig_emb_size"], cfg["key_emb_size"], cfg["max_bars"]
        U, Lo = cfg["max_length"]
        V = VOCAB_SIZE
        B, _, T, F = spectrogram.shape
        dev = spectrogram.device
        drop_on = training and dropout

        gt_cpu = None
        if ground_truth is not None:
            ts_gt, key_gt, up_gt, up_len_gt, lo_gt, lo_len_gt = [g.contiguous() for g in ground_truth]
            # ONE host sync per forward, before anything is enqueued
            gt_cpu = tuple(gt_host) if gt_host is not None else (up_gt.cpu(), lo_gt.cpu(), up_len_gt.cpu(), lo_len_gt.cpu())
        # did a persistent launch of the PREVIOUS pass give up a wait?  (its loss was non-finite, the update skipped; from now on launch per step)
        # (train.TrainStep polls the latch itself, without synchronising, and sets abort_check = False: a blocking read here would wait for the ConvStack it has
        # already enqueued)
        if getattr(self, "abort_check", True):
            hip.check_persist_abort(dev)

        # ConvStack + encoder are enqueued first: the host-side planning of the decoder below runs while they execute.  Its small
        # host->device uploads go through pinned memory without synchronising (a pageable upload would drain the stream each time).
        self.conv_unperm = None
        if conv_pre is not None:
            conv_out, conv_saved, self.conv_unperm = conv_pre
        else:
            mask = (torch.rand((B * T, cfg["conv_feature_size"]), device=dev) >= 0.2).to(torch.uint8) if drop_on else None
            conv_out, conv_saved = self.convstack(S, spectrogram, training, mask)
        # (the encoder is enqueued AFTER the host plan below: its ~4800 launches can block the issuing threads until the GPU has worked
        # most of them off, and a plan made after that wait reaches the decoder late -- tools/phase_times.py --segments: 195 ms into the
        # step on the host against 90 ms on the GPU)
        pinned = []                                      # keeps the staging buffers alive until the step is over
        pin_lock = threading.Lock()
        pool = _pinned_pool(dev)

        def upload(t):
            # small host -> device uploads of the plan, staged in a pinned pool that lives as long as the process: Tensor.pin_memory()
            # calls hipHostMalloc whenever a size has no cached block -- tens of milliseconds during which every thread of the step
            # stands still (tools/phase_times.py --segments showed both clip groups frozen for 85 ms in steps whose segmentation had not
            # occurred before)
            t = t.contiguous()
            nbytes = t.numel() * t.element_size()
            with pin_lock:
                p = pool.take(nbytes, t.dtype, t.shape)
                if p is None:
                    p = torch.empty(t.shape, dtype=t.dtype).pin_memory()
                    pinned.append(p)
            p.copy_(t)
            return p.to(dev, non_blocking=True)

        maxlen = (U, Lo)
        # ---- host plan.  With ground truth the number of executed steps of every (bar, staff) is known up front, so every coin of
        # the reference's protocol (one per executed note step, upper then lower, then one per bar: models.py:404,289) is drawn here,
        # in the reference's order.
        plan = None
        if gt_cpu is not None:
            plan = host_plan if host_plan is not None else draw_plan(gt_cpu, bars, maxlen, rng, teacher_forcing_ratio)
        # Fused training step only (train.TrainStep sets these; never the drop-in module path, whose output rows must equal the
        # reference's everywhere):
        #  skip_finished_rows -- a row whose remaining targets are all <pad> no longer reaches the loss (ignore_index) nor the next
        #    bar token (its staff embedding reads ids[:length] only): its attention, the HBM-bound part of a step, is skipped;
        #  fuse_bars -- when bar k's coin says "teacher-force", bar k+1's input token comes from the ground truth, so the note
        #    decoders of bar k+1 do not depend on those of bar k: consecutive such bars are decoded in ONE call over bars x B rows
        #    (fewer, fatter launches; the rows of a clip share its keys / encoder outputs, streamed once for all of them);
        #  clip_groups -- everything behind the encoder is independent per CLIP (the reference couples the clips of a minibatch only
        #    through the loop lengths and the shared coin flips, both already in the plan): the minibatch is cut into contiguous clip
        #    ranges, e.g. [clips with ordinary bars | clips holding a full-length bar], each range decodes with ITS OWN step counts on
        #    its own streams / host threads, concurrently -- the few-row, latency-bound tail of the long clips (398 dependent steps per
        #    segment) runs under the bandwidth-bound attention of the many ordinary ones instead of after it.
        skip = plan is not None and training and getattr(self, "skip_finished_rows", False)
        fuse = skip and getattr(self, "fuse_bars", False)
        segments = plan_segments(plan, bars, fuse, inference) if plan is not None else [[bar] for bar in range(bars)]
        until_all = None
        if skip:
            until_all = []
            for g in gt_cpu[:2]:
                idx = torch.arange(1, g.shape[-1] + 1, dtype=torch.int32)
                until_all.append(((g != PAD).to(torch.int32) * idx).amax(dim=-1).to(torch.int32).t().contiguous())   # (bars, B): last real target + 1
        clip_groups = getattr(self, "clip_groups", None) if fuse else None
        if not clip_groups or len(clip_groups) < 2:
            clip_groups = [(0, B)]
        assert clip_groups[0][0] == 0 and clip_groups[-1][1] == B and all(a[1] == b[0] for a, b in zip(clip_groups[:-1], clip_groups[1:]))
        # occupancy cap of the bulk group's attention launches (csrc/a2s_seq.hip a2s_attn_bulk_lds): only while another group decodes beside it
        cap_level = 0
        if len(clip_groups) > 1:
            cap_level = 1
        hip.check(hip.lib().a2s_debug_set(b"attn_bulk_cap", cap_level), "debug_set")

        bar_major = fuse
        self.bar_major = bar_major
        ts_out = torch.zeros((B, bars, cfg["num_time_sig"]), device=dev)
        key_out = torch.zeros((B, bars, cfg["num_keys"]), device=dev)
        # fused bars: staff outputs bar-major (bars, clips, len, V) so that the rows of consecutive bars are uniformly strided; with
        # clip groups every group owns a contiguous bar-major block of ONE flat buffer (group_views) -- the loss is a mean over
        # rows, whatever their order (train.TrainStep lays the targets out the same way)
        up_out = torch.zeros((bars, B, U, V) if bar_major else (B, bars, U, V), device=dev)
        lo_out = torch.zeros((bars, B, Lo, V) if bar_major else (B, bars, Lo, V), device=dev)
        self.clip_groups_used = list(clip_groups)

        tokw = 4 * Sz + te + ke
        ldxb = tokw + 2 * H
        greedy_graph = gt_cpu is None and not training and getattr(self, "greedy_graph", _GREEDY_GRAPH)
        # the two staves of a segment run on two streams, each issued by its own host thread -- also in greedy decoding, where each
        # thread polls the done counter of its own stream (the hipGraph variant captures on one created stream and stays sequential)
        concurrent = getattr(self, "concurrent_staves", True) and not greedy_graph
        enc, hidden, enc_saved = self.encoder(S, conv_out, training)
        enc2d = enc.view(B * T, 2 * H)
        keys = {p: self._keys(S, p + ".attn", enc2d, H) for p in ("decoder", "decoder.upper_decoder", "decoder.lower_decoder")}
        enc_hidden = hidden

        def decode_group(gidx, b0, b1, gen):
            """The decoder (reference HierarchicalDecoder.decode_bars, models.py:191-316) over the clips [b0, b1).  Runs on the calling
            thread's current stream plus that group's two staff streams; `gen`: the torch generator its dropout masks come from."""
            Bg = b1 - b0
            _trace(f"g{gidx} decode_group entered")
            hidden = enc_hidden[b0:b1]
            enc_g = enc[b0:b1]
            keys_g = {p: k.view(B, T, H)[b0:b1] for p, k in keys.items()}
            if ground_truth is not None:
                ts_g, key_g, up_g, upl_g, lo_g, lol_g = ts_gt[b0:b1], key_gt[b0:b1], up_gt[b0:b1], up_len_gt[b0:b1], lo_gt[b0:b1], lo_len_gt[b0:b1]
                gt_cpu_g = tuple(t[b0:b1] for t in gt_cpu)
            ts_out_g, key_out_g = ts_out[b0:b1], key_out[b0:b1]
            if bar_major:
                up_out_g, lo_out_g = group_views(up_out, clip_groups, gidx), group_views(lo_out, clip_groups, gidx)
                gt_bm = (up_g.transpose(0, 1).contiguous(), lo_g.transpose(0, 1).contiguous())
            else:
                up_out_g, lo_out_g = up_out[b0:b1], lo_out[b0:b1]
                gt_bm = None
            _trace(f"g{gidx} views made")
            # A group of at most 8 clips decodes each (segment, staff) call as ONE persistent launch (csrc/a2s_dec_persist.hip) whose workgroups
            # hold a clip's keys / encoder outputs in LDS and wait for each other: two such launches in flight at once would share the CUs and
            # starve each other.  So only the LAST group may take that path (the long clips; or the whole minibatch when it is that small), and
            # its two staves then run one after the other on the group's stream.
            # ... and only when no other clip group decodes beside it (engine._PERSIST_BESIDE lifts that, tests): a persistent launch occupies the
            # registers and LDS of every CU, so a bulk group that runs concurrently is stopped for as long as it is resident -- measured at
            # B = 256 with 8 long clips: the long-clip chain went from ~230 to ~85 ms per step, the bulk group's first segment from 280 to 825 us
            # per step, and the step from 514 to 540 ms (profiles/r04_dec_persist_beside_bulk.txt).
            alone = len(clip_groups) == 1 or _PERSIST_BESIDE
            persist_g = (Bg <= 8 and gidx == len(clip_groups) - 1 and alone and H == 256 and E == 16 and (plan is not None or (inference and not greedy_graph))
                         and _os.environ.get("A2S_DEC_PERSIST", "1") != "0")
            concurrent_g = concurrent and staves_concurrent(gidx, len(clip_groups)) and not persist_g
            streams = staff_streams(dev, gidx) if concurrent_g else None

            def rand(shape):
                return torch.rand(shape, device=dev, generator=gen)

            # steps of this group's calls: the loop of a (bar, staff) ends when every row OF THE GROUP has shown <eos> (a group with no
            # full-length row does not run to the cap because another group has one); the coins are the plan's (global step index)
            gsteps = None
            if plan is not None:
                gsteps = [{gi_idx: min(plan[bar][gi_idx][0], plan_note_steps(gt_cpu_g[gi_idx][:, bar, :], maxlen[gi_idx])[0]) for gi_idx in (0, 1)}
                          for bar in range(bars)]

            _trace(f"g{gidx} step counts planned")

            def active_rows(gi_idx, seg, n):
                """Row / clip bookkeeping of one decoder call over the bars `seg` (n steps launched)."""
                until = torch.stack([until_all[gi_idx][bar][b0:b1].clamp(max=gsteps[bar][gi_idx]) for bar in seg])   # the bar's loop ends at its own step count
                clip_until = until.amax(dim=0)
                order = torch.argsort(clip_until, descending=True, stable=True).to(torch.int32)                     # clips that finish last come first
                rank = torch.empty_like(order)
                rank[order.long()] = torch.arange(Bg, dtype=torch.int32)
                cnt = torch.bincount(clip_until.long(), minlength=n + 1)
                n_act = Bg - torch.cumsum(cnt, 0)[:n]                                                               # clips with until > t
                self.attn_clip_steps.append(int(n_act.sum()))          # (clip, step) pairs whose keys / encoder outputs this call streams: bench.py's step roofline
                # 1 + the largest clip POSITION still unfinished at step t (the per-step products of the tail run on that prefix of every
                # fused bar only: a2s_note_dec_args.m_active; train.plan_clip_groups puts the clips with the longest rows first)
                last = torch.zeros(n + 1, dtype=torch.long)
                cu = clip_until.long().clamp(max=n)
                last.scatter_reduce_(0, cu, torch.arange(1, Bg + 1), reduce="amax")                                 # last[u] = 1 + max position with until == u
                m_act = torch.flip(torch.cummax(torch.flip(last, [0]), 0)[0], [0])[1:n + 1]                          # max over until > t
                # the rows themselves, latest-finishing first: the few-row step kernels cover the rows still running (a2s_note_dec_args.row_list)
                uflat = until.reshape(-1)
                row_list = torch.argsort(uflat, descending=True, stable=True).to(torch.int32)
                rcnt = torch.bincount(uflat.long().clamp(max=n), minlength=n + 1)
                n_rows = uflat.numel() - torch.cumsum(rcnt, 0)[:n]                                                  # rows with until > t
                live_idx = None
                if _LIVE_ROWS:
                    live_idx = (torch.arange(n).unsqueeze(1) < uflat.unsqueeze(0)).reshape(-1).nonzero().squeeze(1).to(torch.int32)    # k = step * rows + row
                return dict(until=upload(uflat), order=upload(order), rank=upload(rank),
                            live_idx=upload(live_idx) if live_idx is not None else None, live_steps=n,
                            n_active=(C.c_int * max(n, 1))(*n_act.tolist()), n_clips=Bg,
                            m_active=(C.c_int * max(n, 1))(*m_act.tolist()) if _TAIL_PREFIX else None,
                            row_list=upload(row_list) if _TAIL_ROWS else None,
                            n_rows_active=(C.c_int * max(n, 1))(*n_rows.tolist()) if _TAIL_ROWS else None)

            # Every host-side decision of the decoder and every small upload happens HERE, while the GPU is busy with the ConvStack and
            # the encoder enqueued above -- not once per (segment, staff) in the middle of the decoder.
            _trace(f"g{gidx} planning starts")
            def pair_rows(seg, n):
                """Clip bookkeeping of BOTH staves of a segment together (a2s_note_decoder_fwd_pair): the clips sorted by the step their last row of
                either staff finishes at."""
                until = torch.stack([until_all[gi_idx][bar][b0:b1].clamp(max=gsteps[bar][gi_idx]) for gi_idx in (0, 1) for bar in seg])
                clip_until = until.amax(dim=0)
                order = torch.argsort(clip_until, descending=True, stable=True).to(torch.int32)
                rank = torch.empty_like(order)
                rank[order.long()] = torch.arange(Bg, dtype=torch.int32)
                n_act = Bg - torch.cumsum(torch.bincount(clip_until.long(), minlength=n + 1), 0)[:n]
                # (clip, step) pairs whose encoder outputs ONE pass serves for both staves: steps both staves run on the launch-per-step loop (more rows
                # still running than the few-row kernels take), clips with an unfinished row of each staff -- bench.py's step roofline
                nb_, fmax, t_ = len(seg), L.a2s_debug_get(b"attn_pair_fused_rows"), torch.arange(n).unsqueeze(1)
                rows_live = [(until[k * nb_:(k + 1) * nb_].reshape(1, -1) > t_).sum(1) for k in (0, 1)]
                clip_live = [until[k * nb_:(k + 1) * nb_].amax(dim=0).unsqueeze(0) > t_ for k in (0, 1)]
                joint = (rows_live[0] > fmax) & (rows_live[1] > fmax)
                self.attn_shared_clip_steps.append(int(((clip_live[0] & clip_live[1]).sum(1) * joint).sum()))
                return dict(order=upload(order), rank=upload(rank), n_active=(C.c_int * max(n, 1))(*n_act.tolist()))

            seg_plan = []
            for seg in segments:
                sp = {}
                for gi_idx in (0, 1):
                    if plan is None:
                        sp[gi_idx] = (maxlen[gi_idx], None, None, None)
                        continue
                    steps = max(gsteps[bar][gi_idx] for bar in seg)
                    flags = [sum(int(t < gsteps[bar][gi_idx] and plan[bar][gi_idx][1][t]) << j for j, bar in enumerate(seg)) for t in range(steps)]
                    # the backward pass needs the flags on the device (which token each step consumed)
                    flags_dev = upload(torch.tensor(flags[:steps], dtype=torch.int32)) if training and steps > 0 else None
                    sp[gi_idx] = (steps, flags, active_rows(gi_idx, seg, steps) if skip else None, flags_dev)
                # both staves' step loops issued as one (the sweeps of a step share the pass over the encoder outputs): training, two streams
                # (the bulk group only, and only calls over more rows than the few-row step kernels take: those loops are bound by the host's launch rate)
                if (plan is not None and skip and training and concurrent_g and gidx == 0 and _PAIR_STAVES and sp[0][0] > 0 and sp[1][0] > 0 and H == 256
                        and len(seg) * Bg > L.a2s_debug_get(b"attn_pair_fused_rows") and L.a2s_debug_get(b"attn_pair")):
                    sp["pair"] = pair_rows(seg, max(sp[0][0], sp[1][0]))
                seg_plan.append(sp)

            sos_ids = torch.full((Bg, 2), SOS, dtype=torch.long, device=dev)
            sos_ids[:, 1] = EOS
            two = torch.full((Bg,), 2, dtype=torch.long, device=dev)
            token = self._empty(Bg, tokw, dev=dev)
            sos_rec = [] if training else None
            self._staff_token(S, sos_ids, two, 1, token, 0, 2, 2, True, sos_rec)
            token[:, 2 * Sz:4 * Sz].copy_(token[:, :2 * Sz])
            hip.check(L.a2s_embed_rows(hip.stream(), hip._p(S["decoder.time_sig_emb.weight"]), C.c_void_p(0), C.c_void_p(0), C.c_long(0),
                                       cfg["num_time_sig"], hip._p(token), C.c_long(tokw), 4 * Sz, Bg, te, C.c_void_p(0), hip.f32(1.0)), "embed ts")
            hip.check(L.a2s_embed_rows(hip.stream(), hip._p(S["decoder.key_emb.weight"]), C.c_void_p(0), C.c_void_p(0), C.c_long(0),
                                       cfg["num_keys"], hip._p(token), C.c_long(tokw), 4 * Sz + te, Bg, ke, C.c_void_p(0), hip.f32(1.0)), "embed key")

            bar_saved, seg_saved = [], []
            max_rows = Bg * max(len(sg) for sg in segments)
            # per-staff scratch (split-T attention partials, split-K slabs): the two staves run concurrently on two streams
            attn_ws = [hip.attn_workspace(Bg, T, H, dev, groups=max_rows // Bg) for _ in range(2)]
            gemm_ws = [hip.gemm_workspace(max_rows, dev) for _ in range(2)]

            def bar_step(bar, token, hidden):
                """Bar-level attention + GRU step (models.py:241-247) and the two heads (models.py:281-286)."""
                xbar = self._empty(Bg, ldxb, dev=dev)
                headin = self._empty(Bg, 4 * H, dev=dev)
                if drop_on:
                    keep = (rand((Bg, tokw)) >= 0.1).to(token.dtype)
                    xbar[:, :tokw].copy_(token * keep / 0.9)
                else:
                    keep = None
                    xbar[:, :tokw].copy_(token)
                qb = self._empty(Bg, H, dev=dev)
                Wa = S["decoder.attn.attn.weight"]
                hip.gemm(hidden, 2 * H, 1, Wa, 1, 4 * H, qb, H, Bg, H, 2 * H, bias=S["decoder.attn.attn.bias"])
                attw = self._empty(Bg, T, dev=dev) if training else None
                # 5 calls per forward.  Round 5: on the split-T kernels when the group is large (their combine writes the context with 4-byte stores, so
                # the odd stride of the bar-level GRU input row [token(141) | ctx] does not matter to them): 0.85 -> ~0.2 ms per call at 248 clips, in series
                # with the group's decode; small groups keep the one-workgroup-per-clip kernel
                bar_ws = bar_attn_workspace(dev, gidx, Bg, T, H)
                hip.check(L.a2s_attn_step_fwd(hip.stream(), hip._p(keys_g["decoder"]), hip._p(enc_g), hip._p(qb), C.c_long(H),
                                              hip._p(S["decoder.attn.v.weight"]), C.c_void_p(xbar.data_ptr() + 4 * tokw), C.c_long(ldxb),
                                              C.c_void_p(headin.data_ptr() + 4 * 2 * H), C.c_long(4 * H), hip._p(attw), Bg, T, H,
                                              C.c_void_p(0), 0, hip._p(bar_ws)), "a2s_attn_step_fwd")
                gi = hip.linear(xbar, S["decoder.gru.weight_ih_l0"], S["decoder.gru.bias_ih_l0"])
                gh = hip.linear(hidden, S["decoder.gru.weight_hh_l0"], S["decoder.gru.bias_hh_l0"])
                hnew = self._empty(Bg, 2 * H, dev=dev)
                gates = self._empty(Bg, 8 * H, dev=dev) if training else None
                hip.check(L.a2s_gru_gates_fwd(hip.stream(), hip._p(gi), C.c_long(6 * H), hip._p(gh), C.c_long(6 * H), hip._p(hidden),
                                              C.c_long(2 * H), hip._p(hnew), C.c_long(2 * H), hip._p(headin), C.c_long(4 * H),
                                              hip._p(gates), Bg, 2 * H), "a2s_gru_gates_fwd")
                return dict(xbar=xbar, headin=headin, qb=qb, attw=attw, gates=gates, hprev=hidden, hnew=hnew, keep=keep)

            def bar_heads(bar, headin):
                heads = {}
                for hname, out_t, nc in (("time_sig_out", ts_out_g, cfg["num_time_sig"]), ("key_out", key_out_g, cfg["num_keys"])):
                    t1 = hip.linear(headin, S[f"decoder.{hname}.0.weight"], S[f"decoder.{hname}.0.bias"], act=1)
                    t2 = hip.linear(t1, S[f"decoder.{hname}.2.weight"], S[f"decoder.{hname}.2.bias"], act=1)
                    lg = hip.linear(t2, S[f"decoder.{hname}.4.weight"], S[f"decoder.{hname}.4.bias"])
                    am = torch.empty(Bg, dtype=torch.int32, device=dev)
                    hip.check(L.a2s_log_softmax_rows(hip.stream(), hip._p(lg), C.c_long(nc), C.c_void_p(out_t.data_ptr() + 4 * bar * nc),
                                                     C.c_long(bars * nc), hip._p(am), Bg, nc), "a2s_log_softmax_rows")
                    heads[hname] = (t1, t2, lg, am)
                return heads

            def next_token(bar, teacher_force, staff, heads):
                """Token bar `bar` hands to the next one (models.py:289-311); staff: {name: (ids, lengths)} of this bar's rows (not needed
                when teacher-forced)."""
                token = self._empty(Bg, tokw, dev=dev)
                tok_rec = [] if training else None
                if teacher_force and not inference:
                    if int(gt_cpu_g[2][:, bar].min()) <= 0 or int(gt_cpu_g[3][:, bar].min()) <= 0:
                        raise RuntimeError("Length of all samples has to be greater than 0")     # pack_padded_sequence
                    self._staff_token(S, up_g[:, bar], upl_g[:, bar], bars, token, 0, U, bars * U, True, tok_rec)
                    self._staff_token(S, lo_g[:, bar], lol_g[:, bar], bars, token, 2 * Sz, Lo, bars * Lo, True, tok_rec)
                    ts_ids, key_ids, i64, stride = ts_g[:, bar], key_g[:, bar], True, bars
                else:
                    self._staff_token(S, staff["up"][0], staff["up"][1], 1, token, 0, U, U, False, tok_rec)
                    self._staff_token(S, staff["lo"][0], staff["lo"][1], 1, token, 2 * Sz, Lo, Lo, False, tok_rec)
                    ts_ids, key_ids, i64, stride = heads["time_sig_out"][3], heads["key_out"][3], False, 1
                for table, ids_, col, width in ((S["decoder.time_sig_emb.weight"], ts_ids, 4 * Sz, te), (S["decoder.key_emb.weight"], key_ids, 4 * Sz + te, ke)):
                    hip.check(L.a2s_embed_rows(hip.stream(), hip._p(table), hip._p(ids_) if i64 else C.c_void_p(0),
                                               C.c_void_p(0) if i64 else hip._p(ids_), C.c_long(stride), 0, hip._p(token), C.c_long(tokw), col, Bg,
                                               width, C.c_void_p(0), hip.f32(1.0)), "embed next token")
                return token, tok_rec, (ts_ids, key_ids, i64, stride)

            for seg_i, seg in enumerate(segments):
                nb = len(seg)
                # (1) bar-level chain of the segment: inside a segment every next token comes from the ground truth
                for j, bar in enumerate(seg):
                    _trace(f"g{gidx} bar {bar} step")
                    rec = bar_step(bar, token, hidden)
                    rec["seg"] = (len(seg_saved), j)
                    bar_saved.append(rec)
                    hidden = rec["hnew"]
                    if j + 1 < nb:
                        token, rec["tok_rec"], rec["next_ids"] = next_token(bar, True, None, None)
                        rec["teacher_force"] = True
                h0 = bar_saved[seg[0]]["hnew"] if nb == 1 else torch.cat([bar_saved[bar]["hnew"] for bar in seg], dim=0)
                R = nb * Bg
                # (2) note decoders of the segment (models.py:261-275)
                staff = {}
                calls = []
                for name, prefix, maxs, out_t, gi_idx in (("up", "decoder.upper_decoder", U, up_out_g, 0), ("lo", "decoder.lower_decoder", Lo, lo_out_g, 1)):
                    steps, flags, active, flags_dev = seg_plan[seg_i][gi_idx]
                    if plan is not None:
                        if bar_major:
                            gt_bar = gt_bm[gi_idx][seg[0]:seg[0] + nb].view(R, maxs)
                            probs = out_t[seg[0]:seg[0] + nb].view(R, maxs, V)
                        else:
                            gt_bar = (up_g if gi_idx == 0 else lo_g)[:, seg[0], :]
                            probs = out_t[:, seg[0]]
                    else:
                        gt_bar, probs = None, out_t[:, seg[0]]
                    # the dropout masks are drawn here, on the caller's thread and stream: one deterministic draw order per seed
                    drop = (rand((steps + 1, R, E)) >= 0.1).to(torch.uint8) if drop_on else None
                    calls.append((name, (S, prefix, keys_g[prefix], enc_g, h0, maxs, probs, gt_bar, steps, flags, training, 0.1 if drop_on else 0.0, R, T,
                                         attn_ws[gi_idx], gemm_ws[gi_idx], active, drop, flags_dev, persist_g)))
                if concurrent_g and seg_plan[seg_i].get("pair") is not None:
                    _trace(f"g{gidx} seg {seg_i} staves forked (one loop)")
                    pair_join = fork_on_streams(dev, [streams[1]], [lambda cs=[args for _, args in calls], pr=seg_plan[seg_i]["pair"]: self._decode_pair(cs, streams, pr)])
                    join = lambda pj=pair_join: pj()[0]
                elif concurrent_g:
                    _trace(f"g{gidx} seg {seg_i} staves forked")
                    join = fork_on_streams(dev, streams, [lambda args=args: self._decode_staff(*args) for _, args in calls])
                elif greedy_graph:
                    # greedy decode replays a captured hipGraph: capture needs a created (non-default) stream
                    st = side_streams(dev)[0]
                    res = []
                    for _, args in calls:
                        st.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(st):
                            res.append(self._decode_staff(*args))
                        torch.cuda.current_stream().wait_stream(st)
                        for _ in range(res[-1][2]["steps"]):       # the reference draws once per executed step, also in inference
                            rng.random()
                    join = lambda res=res: res
                else:
                    res = []
                    for _, args in calls:
                        res.append(self._decode_staff(*args))
                        if gt_cpu is None:
                            for _ in range(res[-1][2]["steps"]):
                                rng.random()
                    join = lambda res=res: res
                # (3) heads do not depend on the note decoders: they overlap with them on the main stream
                for bar in seg:
                    bar_saved[bar]["heads"] = bar_heads(bar, bar_saved[bar]["headin"])
                _trace(f"g{gidx} seg {seg_i} heads issued, joining")
                for (name, _), (ids, lengths, sv) in zip(calls, join()):     # the next token / next bar may read what the staves produced
                    sv["groups"] = nb
                    staff[name] = (ids, lengths, sv)
                    if concurrent_g and gt_cpu is None:
                        for _ in range(sv["steps"]):          # the reference draws once per executed step, also in inference (upper, then lower)
                            rng.random()
                seg_saved.append(dict(bars=seg, staff=staff, pair=seg_plan[seg_i].get("pair")))
                for bar in seg:
                    bar_saved[bar]["staff"] = staff            # (shared by the bars of a fused segment)
                # (4) token for the bar after the segment: one draw per bar, after both staves (drawn in the plan with ground truth)
                last = seg[-1]
                teacher_force = plan[last]["tf"] if plan is not None else (rng.random() < teacher_forcing_ratio)
                rec = bar_saved[last]
                last_rows = {name: (staff[name][0][(nb - 1) * Bg:], staff[name][1][(nb - 1) * Bg:]) for name in staff}
                _trace(f"g{gidx} seg {seg_i} joined")
                token, rec["tok_rec"], rec["next_ids"] = next_token(last, teacher_force, last_rows, rec["heads"])
                rec["teacher_force"] = teacher_force
            gs = dict(range=(b0, b1), bars=bar_saved, segments=seg_saved, sos_rec=sos_rec, keys=keys_g, enc=enc_g,
                      outs=(ts_out_g, key_out_g, up_out_g, lo_out_g), gt=(ground_truth is not None and (up_g, lo_g)) or None)
            hook = getattr(self, "group_hook", None)
            if hook is not None:
                # train.TrainStep: the group's loss gradients and decoder backward follow its forward at once, on the group's own
                # thread and streams (the denominators of the NLL means are known from the targets, so no group waits for another)
                hook(gidx, gs)
            return gs

        if len(clip_groups) == 1:
            group_saved = [decode_group(0, 0, B, None)]
        else:
            # group 0 on the caller's thread and stream; every other group on a host thread and stream set of its own, forked from and
            # joined to the caller's stream.  Dropout masks: one generator per group, seeded from torch's (seeded) CPU generator, so
            # that a seed still fixes the run whatever the interleaving of the threads.
            _trace("decoder about to start (generators next)")
            gens = []
            for _ in clip_groups:
                g_ = torch.Generator(device=dev)
                g_.manual_seed(int(torch.randint(0, 2 ** 62, (1,)).item()))
                gens.append(g_)
            _trace("generators made")
            group_saved = run_clip_groups(dev, [lambda gi=gi, r=r: decode_group(gi, r[0], r[1], gens[gi]) for gi, r in enumerate(clip_groups)])
        self.saved = dict(conv=conv_saved, enc=enc_saved, keys=keys, groups=group_saved, enc_out=enc,
                          bars=group_saved[0]["bars"], segments=group_saved[0]["segments"], sos_rec=group_saved[0]["sos_rec"],
                          training=training, concurrent=concurrent, outs=(ts_out, key_out, up_out, lo_out), bar_major=bar_major,
                          clip_groups=list(clip_groups), gt=(ground_truth is not None and (up_gt, lo_gt)) or None, shape=(B, T, F),
                          drop_on=drop_on, pinned=pinned)
        return ts_out, key_out, up_out, lo_out
