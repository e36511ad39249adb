"""Backward pass of the transcription model on liba2s_hip.so: the manual reverse of piano_a2s_amd.engine.Engine.forward.

What torch.autograd would compute for the reference graph (models.py:26-51 and everything below it), composed from
the C-ABI backward kernels.  Sequential parts (note-decoder steps, encoder GRU steps) run in C++ loops; everything that
does not depend on the recurrence is deferred and batched over all steps of a (bar, staff):
  * output projection: ONE GEMM for dlogits_all W_out and ONE for dW_out,
  * GRU / attention-query weight gradients: ONE GEMM each over all steps,
  * attention key/value side: dEnc via a batched (T x steps)(steps x 2H) GEMM, dK via a tanh-recompute kernel.
"""
import ctypes as C
import os
import threading

import torch

from . import hip
from .spec import SOS, VOCAB_SIZE

NULL = C.c_void_p(0)


def _os_env(name, default):
    return os.environ.get(name, default)


def _ptr(t, off=0):
    return C.c_void_p(t.data_ptr() + 4 * off)


_COLSUM_WS = {}


def _colsum(x, ld, out, rows, ncol, x_off=0, out_off=0, beta=1.0):
    ws = None
    if rows >= 2048:                                  # two-stage reduction scratch, one buffer per (device, stream), reused
        # (per host thread as well: two threads may issue into ONE stream -- engine.staff_streams -- and a reduction's two launches must
        # not be interleaved with another reduction that uses the same scratch)
        key = (x.device, torch.cuda.current_stream().cuda_stream, threading.get_ident())
        ws = _COLSUM_WS.get(key)
        if ws is None or ws.numel() < 1024 * ncol:
            ws = _COLSUM_WS[key] = torch.empty(1024 * max(ncol, 2048), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().a2s_col_sum(hip.stream(), _ptr(x, x_off), C.c_long(ld), _ptr(out, out_off), C.c_long(rows), ncol,
                                    hip.f32(1.0), hip.f32(beta), hip._p(ws), C.c_size_t(ws.numel() if ws is not None else 0)), "a2s_col_sum")


def _linear_bwd(x, W, dy, G, wname, bname, dx=None, dx_beta=0.0, x_affine=None, dy_amax=None, x_bound=None):
    """y = x W^T + b  (x (M,K) contiguous, W (N,K)):  dW += dy^T x ; db += colsum(dy) ; dx (+)= dy W.
    x_affine = (scale, shift, period): the layer's input was max(0, x*scale[k // period] + shift[k // period]) formed on the fly
    (hip.linear) -- the weight gradient re-forms it the same way while staging x.  dy_amax: device scalar max|dy| -- the weight
    gradient may then run on the two-term fp16 split (x is O(1): post-BatchNorm activations)."""
    M, K = x.shape
    N = W.shape[0]
    sk = hip.lib().a2s_gemm_pick_splitk(N, K, M, 1)
    hip.gemm(dy, 1, N, x, K, 1, G[wname], K, N, K, M, beta=1.0, splitk=sk, b_affine=x_affine, two_term=(dy_amax, x_bound) if dy_amax is not None else None)
    if bname is not None:
        _colsum(dy, N, G[bname], M, N)
    if dx is not None:
        hip.gemm(dy, N, 1, W, K, 1, dx, dx.stride(0), M, K, N, beta=dx_beta)
    return dx


_ONES = {}


def _one(dev):
    """Device scalar 1.0: the range of an operand bounded by 1 (softmax weights, GRU states) for the two-term fp16 products."""
    key = torch.device(dev).index
    if key not in _ONES:
        _ONES[key] = torch.ones(1, dtype=torch.float32, device=dev)
    return _ONES[key]


def _staff_token_bwd(eng, S, G, rec, dtok):
    """Backward of one _staff_token call; dtok: (R, >= col0+2S) gradient buffer holding the token gradient."""
    L = hip.lib()
    names = [f"decoder.staff_emb.{w}_{sfx}" for sfx in ("l0", "l0_reverse") for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    warr = (C.c_void_p * 8)(*[S[n].data_ptr() for n in names])
    gptrs = G.get("__staff_emb_ptrs__")
    if gptrs is None:
        gptrs = torch.tensor([G[n].data_ptr() for n in names], dtype=torch.int64, device=dtok.device)
    E, Sz = eng.cfg["note_emb_size"], eng.cfg["staff_emb_size"]
    ids = rec["ids"]
    hip.check(L.a2s_staff_emb_bwd(hip.stream(), hip._p(S["decoder.note_emb.weight"]), warr, hip._p(gptrs), hip._p(G["decoder.note_emb.weight"]),
                                  hip._p(ids) if rec["i64"] else NULL, NULL if rec["i64"] else hip._p(ids), C.c_long(rec["id_bstride"]),
                                  hip._p(rec["lengths"]), C.c_long(rec["len_stride"]), hip._p(dtok), C.c_long(dtok.stride(0)), rec["col0"],
                                  hip._p(rec["hsave"]), dtok.shape[0], rec["maxlen"], E, Sz), "a2s_staff_emb_bwd")
    return gptrs     # keep alive until the caller returns


def _attn_deferred(eng, S, G, prefix, keys, enc, q_all, ds_all, attw_all, dctx_all, dK, dEnc, B, T, H, steps, active=None, groups=1):
    """Key/value side of `steps` attention calls of one layer: dEnc += A^T dCtx (per clip), dK += ..., dv += ...
    The per-step tensors hold groups*B rows per step, row = group*B + clip (fused bars): (step, group) is one flat reduction index."""
    L = hip.lib()
    # dEnc[b] += sum_{s,g} a_sg[b,:]^T dctx_sg[b,:]   -- batched over clips: (T x steps*groups)(steps*groups x 2H)
    # (operand ranges for the two-term fp16 product: softmax weights <= 1, max |dctx| measured; K = steps * groups < 256 runs fp32-input)
    hip.gemm(attw_all, 1, B * T, dctx_all, B * 2 * H, 1, dEnc, 2 * H, T, 2 * H, steps * groups, beta=1.0, batch=B, bsA=T, bsB=2 * H, bsC=T * 2 * H,
             two_term=(_one(enc.device), hip.absmax(dctx_all)) if dctx_all.is_contiguous() else None)
    nblk = L.a2s_attn_dk_blocks(B, T)
    dvp = torch.empty((nblk, H), dtype=torch.float32, device=enc.device)
    hip.check(L.a2s_attn_dk_accum(hip.stream(), hip._p(keys), hip._p(q_all), hip._p(ds_all), hip._p(S[prefix + ".v.weight"]), hip._p(dK),
                                  hip._p(dvp), B, T, steps, H, hip._p(active["until"] if active else None), groups), "a2s_attn_dk_accum")
    _colsum(dvp, H, G[prefix + ".v.weight"], nblk, H)


def _note_decoder_bwd(eng, S, G, sv, keys, enc, dprobs_bar, probs_bar, dK, dEnc, n_clips, T, deferred=None, enc_amax=None, late=None, defer_launch=False):
    """Reverse of Engine._decode_staff.  Returns the gradient wrt the initial hidden (rows, 2H); rows = groups * n_clips.
    late: optional list.  The call's WEIGHT gradients (output projection, GRU, attention query half, embedding rows: everything only the
    optimizer waits for) are then not computed here: a (closure, event, tensors) entry is appended and Backward.finish runs the closures on the
    weight-gradient stream beside the encoder's back-propagation -- on the staff's own stream they sit in series with the next segment's decode
    steps (8-12 ms of GEMMs per call of the bulk clip group, profiles/r05_queue_timeline.txt).  The key / encoder-output gradients stay here.
    deferred: optional HIP stream for everything that does not gate the recurrence (weight gradients, the deferred key / encoder-
    output gradients, embedding scatter): a staff's stream executes in order, so leaving them on it would put ~20 % of MFMA-bound GEMM
    time in series with the HBM-bound attention steps of the next segment; returns (dh0, event after the deferred work or None)."""
    L = hip.lib()
    groups = sv.get("groups", 1)
    B = groups * n_clips
    cfg = eng.cfg
    H, E, V = cfg["hidden_size"], cfg["note_emb_size"], VOCAB_SIZE
    H2, ldx = 2 * H, E + 2 * H
    n, prefix, dev = sv["steps"], sv["prefix"], enc.device
    R = n * B
    # (a) dlogits for all executed steps, step-major (n, B, V)
    dlog = torch.empty((n, B, V), dtype=torch.float32, device=dev)
    hip.check(L.a2s_log_softmax_bwd_rows(hip.stream(), hip._p(dprobs_bar), hip._p(probs_bar), C.c_long(probs_bar.stride(0)), n, hip._p(dlog),
                                         R, V, B, 1), "a2s_log_softmax_bwd_rows")
    # (b) output projection, all steps at once: do_all = dlog W_out ; dW_out += dlog^T o ; db_out += colsum
    o2d, dlog2d = sv["o"].view(R, 2 * H2), dlog.view(R, V)
    do_all = torch.empty((n, B, 2 * H2), dtype=torch.float32, device=dev)
    Wo = S[prefix + ".out.weight"]
    dlog_amax = hip.absmax(dlog) if enc_amax is not None else None
    hip.gemm(dlog2d, V, 1, Wo, 2 * H2, 1, do_all.view(R, 2 * H2), 2 * H2, R, 2 * H2, V,          # do_all = dlog W_out  (gates the recurrence)
             two_term=(dlog_amax, hip.absmax(Wo)) if enc_amax is not None else None)
    # (c) reverse recurrence
    dgi_all = torch.empty((n, B, 3 * H2), dtype=torch.float32, device=dev)
    dgh_all = torch.empty((n, B, 3 * H2), dtype=torch.float32, device=dev)
    dq_all = torch.empty((n, B, H), dtype=torch.float32, device=dev)
    ds_all = torch.empty((n, B, T), dtype=torch.float32, device=dev)
    dctx_all = torch.empty((n, B, H2), dtype=torch.float32, device=dev)
    m_active = sv["active"].get("m_active") if sv.get("active") else None
    # (with m_active the per-step products skip the rows of finished clips: their dx rows are never written and must read as zero)
    dx = torch.empty((n, B, ldx), dtype=torch.float32, device=dev)          # (zero-filled by a2s_note_decoder_bwd when rows are skipped)
    dh = torch.empty((2, B, H2), dtype=torch.float32, device=dev)
    a = hip.NoteDecBwdArgs()
    for name, t in (("attn_w", S[prefix + ".attn.attn.weight"]), ("attn_v", S[prefix + ".attn.v.weight"]), ("w_ih", S[prefix + ".gru.weight_ih_l0"]),
                    ("w_hh", S[prefix + ".gru.weight_hh_l0"]), ("keys", keys), ("enc", enc), ("h", sv["h"]), ("x", sv["x"]), ("q", sv["q"]),
                    ("gates", sv["gates"]), ("attw", sv["attw"]), ("do_all", do_all), ("dgi_all", dgi_all), ("dgh_all", dgh_all),
                    ("dq_all", dq_all), ("ds_all", ds_all), ("dctx_all", dctx_all), ("dx", dx), ("dh", dh), ("attn_ws", sv["attn_ws"]), ("gemm_ws", sv["gemm_ws"]),
                    ("clip_order", sv.get("active") and sv["active"]["order"]), ("clip_rank", sv.get("active") and sv["active"]["rank"]),
                    ("row_until", sv.get("active") and sv["active"]["until"])):
        setattr(a, name, t.data_ptr() if t is not None else None)
    a.gemm_ws_bytes = sv["gemm_ws"].numel() * 4 if sv["gemm_ws"] is not None else 0
    if sv.get("step_ws") is not None:
        a.step_ws, a.step_ws_floats = sv["step_ws"].data_ptr(), sv["step_ws"].numel()
    a.n_active = C.cast(sv["active"]["n_active"], C.c_void_p).value if sv.get("active") else None
    a.n_clips = sv["active"]["n_clips"] if sv.get("active") else 0
    a.m_active = C.cast(m_active, C.c_void_p).value if m_active is not None else None
    if sv.get("active") and sv["active"].get("row_list") is not None:
        a.row_list, a.n_rows_active = sv["active"]["row_list"].data_ptr(), C.cast(sv["active"]["n_rows_active"], C.c_void_p).value
    else:
        a.row_list, a.n_rows_active = None, None
    a.R, a.T, a.H, a.E, a.steps = B, T, H, E, n
    # the forward ran this call as one persistent launch (at most 8 clips): so does the reverse loop (csrc/a2s_dec_persist.hip)
    a.persist_ws, a.persist_ws_bytes, a.w_ih_full = None, 0, None
    persist_ws = None
    if sv.get("persist_ws") is not None:
        nb_ws = L.a2s_note_decoder_bwd_persist_ws_bytes(n_clips)
        if nb_ws:
            persist_ws = torch.empty(nb_ws, dtype=torch.uint8, device=dev)
            a.persist_ws, a.persist_ws_bytes = persist_ws.data_ptr(), nb_ws
    if not defer_launch:
        hip.check(L.a2s_note_decoder_bwd(hip.stream(), C.byref(a)), "a2s_note_decoder_bwd")
    # (d) everything nobody in the recurrence waits for
    def deferred_work(part="all"):
        if part == "attn":          # the key / encoder-output gradients only (the weight gradients follow in Backward.finish)
            _attn_deferred(eng, S, G, prefix + ".attn", keys, enc, sv["q"], ds_all, sv["attw"], dctx_all, dK, dEnc, n_clips, T, H, n, sv.get("active"), groups)
            return
        # Operand ranges of the weight-gradient products (two-term fp16 split): the GRU state lies in (-1, 1), the
        # context is a convex combination of encoder rows, the token half of x an embedding row; the gradients' max magnitudes are measured.
        one = _one(dev)
        xb = torch.maximum(hip.absmax(S[prefix + ".embedding.weight"]), enc_amax) if enc_amax is not None else None
        ob = torch.maximum(one, enc_amax) if enc_amax is not None else None
        am = (lambda t: hip.absmax(t)) if enc_amax is not None else (lambda t: None)
        x2d, h2d = sv["x"][:n].view(R, ldx), sv["h"][:n].view(R, H2)
        # Every product below contracts over the (step, row) pairs of the call.  Rows whose targets had run out were skipped by the forward pass and
        # carry EXACT zero gradients from then on (dlog, dgi, dgh, dq: zeros): with the bench's lengths that is three quarters of the pairs.  The
        # pairs that ran (active["live_idx"], planned on the host with everything else) are gathered into dense operands first -- 22 KB per pair,
        # read once -- and the products, bias sums and operand ranges run over those only.
        live = sv["active"].get("live_idx") if (sv.get("active") and sv["active"].get("live_steps") == n) else None
        dlog_w, o_w, dgi_w, dgh_w, dq_w, Rw = dlog2d, o2d, dgi_all.view(R, 3 * H2), dgh_all.view(R, 3 * H2), dq_all.view(R, H), R
        if live is not None and live.numel() < 0.8 * R:
            Rw = live.numel()
            if Rw == 0:
                return                                    # nothing ran: every gradient of this call is zero
            sel = lambda t2d: t2d.index_select(0, live)
            dlog_w, o_w, dgi_w, dgh_w, dq_w, x2d, h2d = sel(dlog_w), sel(o_w), sel(dgi_w), sel(dgh_w), sel(dq_w), sel(x2d), sel(h2d)
        else:
            live = None
        _linear_bwd(o_w, Wo, dlog_w, G, prefix + ".out.weight", prefix + ".out.bias", dy_amax=dlog_amax, x_bound=ob)       # dW_out += dlog^T o ; db_out += colsum
        _linear_bwd(x2d, S[prefix + ".gru.weight_ih_l0"], dgi_w, G, prefix + ".gru.weight_ih_l0", prefix + ".gru.bias_ih_l0",
                    dy_amax=am(dgi_w), x_bound=xb)
        _linear_bwd(h2d, S[prefix + ".gru.weight_hh_l0"], dgh_w, G, prefix + ".gru.weight_hh_l0", prefix + ".gru.bias_hh_l0",
                    dy_amax=am(dgh_w), x_bound=one)
        # attention query half: dW[:, :2H] += dq^T h ; db += colsum(dq)
        Gw = G[prefix + ".attn.attn.weight"]
        sk = L.a2s_gemm_pick_splitk(H, H2, Rw, 1)
        hip.gemm(dq_w, 1, H, h2d, H2, 1, Gw, 4 * H, H, H2, Rw, beta=1.0, splitk=sk, two_term=(hip.absmax(dq_w), one) if enc_amax is not None else None)
        _colsum(dq_w, H, G[prefix + ".attn.attn.bias"], Rw, H)
        if part == "all":
            _attn_deferred(eng, S, G, prefix + ".attn", keys, enc, sv["q"], ds_all, sv["attw"], dctx_all, dK, dEnc, n_clips, T, H, n, sv.get("active"), groups)
        # embedding rows of the tokens consumed at each step: <sos> at step 0, then gt or argmax of the previous step
        tok = torch.full((n, B), SOS, dtype=torch.int32, device=dev)
        if n > 1:
            prev = sv["ids"][:, :n - 1].t()
            if sv["gt_bar"] is not None:
                bits = sv.get("flags_dev")                                                                            # bit g: group g teacher-forced
                if bits is None:
                    bits = torch.tensor(sv["flags"][:n - 1], dtype=torch.int32, device=dev)
                bits = bits[:n - 1].unsqueeze(1)
                grp = (torch.arange(B, dtype=torch.int32, device=dev) // n_clips).unsqueeze(0)
                flags = ((bits >> grp) & 1).bool()
                prev = torch.where(flags, sv["gt_bar"][:, :n - 1].t().to(torch.int32), prev)
            tok[1:] = prev
        drop = sv["drop"]
        dx_w, ld_w = dx, ldx
        if live is not None:        # (the token-embedding columns of dx at the pairs that ran; their tokens; their dropout masks)
            tok = tok.view(-1).index_select(0, live)
            dx_w, ld_w = dx.view(R, ldx)[:, :E].index_select(0, live), E
            if drop is not None:
                drop = drop.reshape(-1, E)[:R].index_select(0, live)
        hip.check(L.a2s_embed_scatter_add(hip.stream(), hip._p(G[prefix + ".embedding.weight"]), NULL, hip._p(tok), C.c_long(1), 0, hip._p(dx_w),
                                          C.c_long(ld_w), 0, Rw, E, hip._p(drop), hip.f32(1.0 / (1.0 - sv["drop_p"]) if drop is not None else 1.0)),
                  "a2s_embed_scatter_add")

    def after_launch():
        if late is not None:
            deferred_work("attn")
            ev = torch.cuda.Event()
            ev.record()
            # everything the closure reads that was allocated on this (staff / group) stream: the weight-gradient stream records them all
            act = sv.get("active") or {}
            late.append((lambda: deferred_work("weights"), ev, (dlog, do_all, dgi_all, dgh_all, dq_all, dx, sv["x"], sv["h"], sv["o"], sv["ids"], sv["drop"], sv["gt_bar"],
                                                                sv.get("flags_dev"), act.get("live_idx"), dlog_amax, enc_amax)))
            return dh[0], None
        if deferred is None:
            deferred_work()
            return dh[0], None
        ev = torch.cuda.Event()
        ev.record()
        deferred.wait_event(ev)
        for t in (dlog, do_all, dgi_all, dgh_all, dq_all, ds_all, dctx_all, dx):
            t.record_stream(deferred)                       # keep the allocator from handing them back to the staff stream too early
        with torch.cuda.stream(deferred):
            deferred_work()
            done = torch.cuda.Event()
            done.record()
        return dh[0], done

    if defer_launch:                  # _note_decoder_bwd_pair: the arguments are ready (kept alive by the closure), the caller launches both staves with one call
        return a, after_launch, (persist_ws,)
    return after_launch()


def _note_decoder_bwd_pair(calls, streams, pair):
    """The reverse loops of a segment's two note decoders issued by ONE host loop on their two streams (Engine._decode_pair in reverse): while both staves
    step, the attention sweep of a step is one launch that reads the encoder outputs once for both (a2s_note_decoder_bwd_pair).  calls: the argument
    tuples of _note_decoder_bwd (upper, lower); returns their results."""
    prepared = []
    for st, args in zip(streams, calls):
        with torch.cuda.stream(st):
            prepared.append(_note_decoder_bwd(*args, defer_launch=True))
    hip.check(hip.lib().a2s_note_decoder_bwd_pair(C.c_void_p(streams[0].cuda_stream), C.c_void_p(streams[1].cuda_stream), C.byref(prepared[0][0]),
                                                  C.byref(prepared[1][0]), hip._p(pair["order"]), hip._p(pair["rank"]), pair["n_active"]), "a2s_note_decoder_bwd_pair")
    res = []
    for st, (_, after, _keep) in zip(streams, prepared):
        with torch.cuda.stream(st):
            res.append(after())
    return res

class Backward:
    """The backward pass in three phases, so that the decoder part of each clip group can be started by whoever has that group's loss
    gradients first (train.TrainStep chains it right behind the group's forward, on the group's own streams):
        ctx = Backward(eng, S, (B, T, F), device, clip_groups, concurrent, bar_major)     allocations, before any group starts
        ctx.decoder_group(gidx, gs, dts_g, dkey_g, dup_g, dlo_g)                        one group: note decoders, bar chain, heads
        G = ctx.finish(grad_ready)                                                      keys, encoder, ConvStack
    What the groups write per CLIP (dEnc, dK, d_hidden) they write to disjoint slices; what they ACCUMULATE over clips (every weight
    gradient) goes to a flat gradient buffer of the group's own (16.4 M floats), added to `flat` once after the join -- no two streams
    ever accumulate into the same memory."""

    check_fold = False          # tests set it: finish() then verifies (with a host sync) what the late fold of the group buffers assumes

    def __init__(self, eng, S, shape, dev, clip_groups, concurrent, bar_major):
        from .spec import flat_layout, is_buffer
        self.eng, self.S, self.dev = eng, S, dev
        self.B, self.T, self.F = shape
        B, T = self.B, self.T
        H = eng.cfg["hidden_size"]
        H2 = 2 * H
        self.names = [k for k in S if not is_buffer(k)]
        self.offs, self.total = flat_layout([S[k].numel() for k in self.names])          # same layout as models.ScoreTranscription.flatten_()
        # one word behind the gradients: the step's loss (train.TrainStep writes it before finish()).  It rides in the FIRST slice that is
        # all-reduced, so the data-parallel skip / apply decision ("is every rank's loss finite?") needs no collective of its own.
        self.flat_full = torch.zeros(self.total + 1, dtype=torch.float32, device=dev)
        self.flat = self.flat_full[:self.total]
        self.G = {k: self.flat[off:off + S[k].numel()].view(S[k].shape) for k, off in zip(self.names, self.offs)}
        # device table of the staff-embedding gradient pointers: uploaded once, before the first kernel of the backward pass
        ptrs_host = torch.tensor([self.G[f"decoder.staff_emb.{w}_{sfx}"].data_ptr() for sfx in ("l0", "l0_reverse")
                                  for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")], dtype=torch.int64).pin_memory()
        self.G["__staff_emb_ptrs__"] = ptrs_host.to(dev, non_blocking=True)          # pinned: no stream synchronisation (the host keeps running ahead)
        self.keep_alive = [ptrs_host]
        self.dEnc = torch.zeros((B, T, H2), dtype=torch.float32, device=dev)
        self.dK = {p: torch.zeros((B, T, H), dtype=torch.float32, device=dev) for p in ("decoder", "decoder.upper_decoder", "decoder.lower_decoder")}
        # the two note decoders of a bar back-propagate concurrently on two side streams (see engine.side_streams); each accumulates its
        # encoder-output gradient in its own buffer (summed once at the end), everything else they write is per-staff already
        self.concurrent = bool(concurrent)
        self.dEnc_staff = [torch.zeros_like(self.dEnc), torch.zeros_like(self.dEnc)] if self.concurrent else [self.dEnc, self.dEnc]
        self.bar_major = bool(bar_major)
        # weight gradients etc. of each staff off its recurrence stream: measured +0.7 % in round 1, but two more streams than the four
        # hardware queues the runtime has (engine.group_stream) -- off since the clip groups need a queue
        self.use_deferred = False
        # round 5: the note decoders' weight gradients run in finish(), on the weight-gradient stream beside the encoder's back-propagation
        # (eng.late_wgrads = False: where they were, behind each call's reverse loop on the staff's stream)
        lw = getattr(eng, "late_wgrads", None)
        self.late_wgrads = True if lw is None else bool(lw)
        self.late = []                                     # (closure, event on the issuing stream, tensors it reads): appended by any group's host thread
        self.clip_groups = list(clip_groups) if clip_groups else [(0, B)]
        # the deferred products of the decoder backward (weight gradients, key / encoder-output gradients) with measured operand ranges on
        # the two-term fp16 split instead of three bf16 terms (round 2's)
        self.two_term = True
        self.d_hidden = torch.empty((B, H2), dtype=torch.float32, device=dev)       # gradient wrt the encoder's bridge output (initial bar-level hidden)
        self.group_flat = [self.flat] + [torch.zeros_like(self.flat) for _ in self.clip_groups[1:]]
        self.group_ptrs = [self.G["__staff_emb_ptrs__"]]
        for gf in self.group_flat[1:]:
            ph = (ptrs_host + (gf.data_ptr() - self.flat.data_ptr())).pin_memory()
            self.keep_alive.append(ph)
            self.group_ptrs.append(ph.to(dev, non_blocking=True))

    def decoder_group(self, gidx, gs, dts_g, dkey_g, dup_g, dlo_g):
        """Decoder backward of clip group gidx (gs: what Engine.forward saved for it) on the calling thread's current stream (+ the two
        side streams for group 0).  d*_g: gradients wrt the group's four output views (gs["outs"])."""
        from .engine import fork_on_streams, staff_streams
        eng, S, G, dev, T = self.eng, self.S, self.G, self.dev, self.T
        names, offs, group_flat, group_ptrs = self.names, self.offs, self.group_flat, self.group_ptrs
        dEnc, dK, dEnc_staff, d_hidden = self.dEnc, self.dK, self.dEnc_staff, self.d_hidden
        concurrent, bar_major, use_deferred = self.concurrent, self.bar_major, self.use_deferred
        L = hip.lib()
        cfg = eng.cfg
        H, Sz = cfg["hidden_size"], cfg["staff_emb_size"]
        te, ke, bars = cfg["time_sig_emb_size"], cfg["key_emb_size"], cfg["max_bars"]
        tokw = 4 * Sz + te + ke
        H2 = 2 * H
        b0, b1 = gs["range"]
        Bg = b1 - b0
        Gg = G if gidx == 0 else {k: group_flat[gidx][off:off + S[k].numel()].view(S[k].shape) for k, off in zip(names, offs)}
        Gg["__staff_emb_ptrs__"] = group_ptrs[gidx]
        enc_g, keys_g = gs["enc"], gs["keys"]
        enc_amax_g = hip.absmax(enc_g) if (self.two_term and enc_g.is_contiguous()) else None      # max |enc| of the group's clips
        dEnc_g = dEnc[b0:b1]
        dK_g = {p: t[b0:b1] for p, t in dK.items()}
        dEnc_staff_g = [t[b0:b1] for t in dEnc_staff]
        ts_out_g, key_out_g, up_out_g, lo_out_g = gs["outs"]
        # as in Engine.forward (engine.staff_streams); a group whose calls ran as persistent launches back-propagates its staves one after
        # the other (two persistent launches must never be in flight together)
        persist_g = any(seg["staff"][k][2].get("persist_ws") is not None for seg in gs["segments"] for k in ("up", "lo"))
        from .engine import staves_concurrent
        concurrent_g = concurrent and staves_concurrent(gidx, len(self.clip_groups)) and not persist_g
        streams = staff_streams(dev, gidx) if concurrent_g else None
        use_deferred_g = use_deferred and gidx == 0
        deferred_streams = _deferred_streams(dev, gidx) if use_deferred_g else None
        deferred_done = []
        seg_dh0 = {}                # segment index -> [dh0 of the upper call, dh0 of the lower call], rows = (bar in segment, clip)

        def segment_decoders_bwd(si_seg):
            """Both note decoders of a segment (bars decoded in one call each): they only need the loss gradients."""
            seg = gs["segments"][si_seg]
            bar0, nb = seg["bars"][0], len(seg["bars"])
            calls = []
            for si, (name, prefix, dout, out_t) in enumerate((("up", "decoder.upper_decoder", dup_g, up_out_g), ("lo", "decoder.lower_decoder", dlo_g, lo_out_g))):
                if bar_major:
                    maxs = out_t.shape[2]
                    dpr, pr = dout[bar0:bar0 + nb].view(nb * Bg, maxs, -1), out_t[bar0:bar0 + nb].view(nb * Bg, maxs, -1)
                else:
                    dpr, pr = dout[:, bar0], out_t[:, bar0]
                calls.append((eng, S, Gg, seg["staff"][name][2], keys_g[prefix], enc_g, dpr, pr, dK_g[prefix], dEnc_staff_g[si], Bg, T,
                              deferred_streams[si] if use_deferred_g else None, enc_amax_g, self.late if self.late_wgrads else None))
            if concurrent_g and seg.get("pair") is not None and L.a2s_debug_get(b"attn_pair"):
                # both staves' reverse loops from one host thread, their sweeps as one launch per step (as the forward ran them)
                (res,), events = fork_on_streams(dev, [streams[1]], [lambda: _note_decoder_bwd_pair(calls, streams, seg["pair"])])(wait=False)
                seg_dh0[si_seg] = ([r[0] for r in res], events)
                deferred_done.extend(r[1] for r in res if r[1] is not None)
            elif concurrent_g:    # one host thread per staff (engine.fork_on_streams); the current stream waits when the result is consumed
                res, events = fork_on_streams(dev, streams, [lambda args=args: _note_decoder_bwd(*args) for args in calls])(wait=False)
                seg_dh0[si_seg] = ([r[0] for r in res], events)
                deferred_done.extend(r[1] for r in res if r[1] is not None)
            else:
                seg_dh0[si_seg] = ([_note_decoder_bwd(*args)[0] for args in calls], [])

        # The note decoders' backward passes only need the loss gradients: all segments are enqueued up front (last segment first, as the
        # bar chain below consumes them), so the two staff streams run through every segment back to back instead of draining at each
        # segment boundary while the bar-level chain catches up.
        for si_seg in reversed(range(len(gs["segments"]))):
            segment_decoders_bwd(si_seg)

        bar_attn = []               # (q, ds, attention weights, dctx) of every bar: their key / encoder-output gradients in ONE call below
        d_hid_carry = None          # gradient wrt the bar-level hidden after bar k, coming from bar k+1
        d_token_next = None         # gradient wrt the (pre-dropout) token that bar k produced for bar k+1
        for bar in reversed(range(bars)):
            b = gs["bars"][bar]
            # ---- (1) the token this bar produced for the next one
            if d_token_next is not None:
                for rec in b["tok_rec"]:
                    _staff_token_bwd(eng, S, Gg, rec, d_token_next)
                ts_ids, key_ids, i64, stride = b["next_ids"]
                for table, ids_, col, width in (("decoder.time_sig_emb.weight", ts_ids, 4 * Sz, te), ("decoder.key_emb.weight", key_ids, 4 * Sz + te, ke)):
                    hip.check(L.a2s_embed_scatter_add(hip.stream(), hip._p(Gg[table]), hip._p(ids_) if i64 else NULL, NULL if i64 else hip._p(ids_),
                                                      C.c_long(stride), 0, hip._p(d_token_next), C.c_long(tokw), col, Bg, width, NULL, hip.f32(1.0)), "scatter ts/key")
            # ---- (2) heads: log_softmax + 3-layer MLP on headin = [bar_summary | ctx]
            d_headin = torch.zeros((Bg, 4 * H), dtype=torch.float32, device=dev)
            for hname, dout, out_t, nc in (("time_sig_out", dts_g, ts_out_g, cfg["num_time_sig"]), ("key_out", dkey_g, key_out_g, cfg["num_keys"])):
                t1, t2, lg, _ = b["heads"][hname]
                dlg = torch.empty((Bg, nc), dtype=torch.float32, device=dev)
                hip.check(L.a2s_log_softmax_bwd_rows(hip.stream(), _ptr(dout, bar * nc), _ptr(out_t, bar * nc), C.c_long(bars * nc), 1, hip._p(dlg),
                                                     Bg, nc, Bg, 0), "lsm bwd head")
                p = f"decoder.{hname}"
                dt2 = torch.empty_like(t2)
                _linear_bwd(t2, S[p + ".4.weight"], dlg, Gg, p + ".4.weight", p + ".4.bias", dx=dt2)
                hip.check(L.a2s_ew_act_bwd(hip.stream(), hip._p(dt2), hip._p(t2), hip._p(dt2), C.c_long(dt2.numel()), 1), "relu bwd")
                dt1 = torch.empty_like(t1)
                _linear_bwd(t1, S[p + ".2.weight"], dt2, Gg, p + ".2.weight", p + ".2.bias", dx=dt1)
                hip.check(L.a2s_ew_act_bwd(hip.stream(), hip._p(dt1), hip._p(t1), hip._p(dt1), C.c_long(dt1.numel()), 1), "relu bwd")
                _linear_bwd(b["headin"], S[p + ".0.weight"], dt1, Gg, p + ".0.weight", p + ".0.bias", dx=d_headin, dx_beta=1.0)
            # ---- (3) note decoders: both start from bar_summary
            d_hnew = torch.zeros((Bg, H2), dtype=torch.float32, device=dev)
            si_seg, j = b["seg"]
            dh0s, events = seg_dh0[si_seg]
            for ev in events:
                torch.cuda.current_stream().wait_event(ev)
            for dh0 in dh0s:
                d_hnew.add_(dh0[j * Bg:(j + 1) * Bg])
            d_hnew.add_(d_headin[:, :H2])
            if d_hid_carry is not None:
                d_hnew.add_(d_hid_carry)
            # ---- (4) bar-level GRU step + attention
            ldxb = tokw + H2
            dgi, dgh = torch.empty((Bg, 3 * H2), device=dev), torch.empty((Bg, 3 * H2), device=dev)
            dhp = torch.empty((Bg, H2), device=dev)
            hip.check(L.a2s_gru_gates_bwd(hip.stream(), hip._p(d_hnew), C.c_long(H2), NULL, C.c_long(0), hip._p(b["gates"]), hip._p(b["hprev"]), C.c_long(H2),
                                          hip._p(dgi), C.c_long(3 * H2), hip._p(dgh), C.c_long(3 * H2), NULL, C.c_long(0), hip._p(dhp), C.c_long(H2), Bg, H2),
                      "gates bwd bar")
            d_xbar = torch.empty((Bg, ldxb), device=dev)
            _linear_bwd(b["xbar"], S["decoder.gru.weight_ih_l0"], dgi, Gg, "decoder.gru.weight_ih_l0", "decoder.gru.bias_ih_l0", dx=d_xbar)
            _linear_bwd(b["hprev"], S["decoder.gru.weight_hh_l0"], dgh, Gg, "decoder.gru.weight_hh_l0", "decoder.gru.bias_hh_l0", dx=dhp, dx_beta=1.0)
            dq, ds = torch.empty((Bg, H), device=dev), torch.empty((1, Bg, T), device=dev)
            dctx = torch.empty((1, Bg, H2), device=dev)
            from .engine import bar_attn_workspace
            bar_ws = bar_attn_workspace(dev, gidx, Bg, T, H)
            if bar_ws is not None:
                # split-T kernels (round 5: 1.3 -> ~0.25 ms per bar at 248 clips): they load the saved context and its gradient 16 bytes at a time, so
                # the two odd-stride column blocks of the bar-level GRU input row are copied out first (0.5 MB each)
                ctx_c, dctx_c = b["xbar"][:, tokw:].contiguous(), d_xbar[:, tokw:].contiguous()
                hip.check(L.a2s_attn_step_bwd(hip.stream(), hip._p(keys_g["decoder"]), hip._p(enc_g), hip._p(b["qb"]), C.c_long(H), hip._p(S["decoder.attn.v.weight"]),
                                              hip._p(b["attw"]), hip._p(ctx_c), C.c_long(H2), hip._p(dctx_c), C.c_long(H2), _ptr(d_headin, H2),
                                              C.c_long(4 * H), hip._p(dctx), C.c_long(H2), hip._p(dq), C.c_long(H), hip._p(ds), Bg, T, H, hip._p(bar_ws)), "attn bwd bar")
            else:
                hip.check(L.a2s_attn_step_bwd(hip.stream(), hip._p(keys_g["decoder"]), hip._p(enc_g), hip._p(b["qb"]), C.c_long(H), hip._p(S["decoder.attn.v.weight"]),
                                              hip._p(b["attw"]), _ptr(b["xbar"], tokw), C.c_long(ldxb), _ptr(d_xbar, tokw), C.c_long(ldxb), _ptr(d_headin, H2),
                                              C.c_long(4 * H), hip._p(dctx), C.c_long(H2), hip._p(dq), C.c_long(H), hip._p(ds), Bg, T, H, NULL), "attn bwd bar")
            Wa = S["decoder.attn.attn.weight"]
            hip.gemm(dq, H, 1, Wa, 4 * H, 1, dhp, H2, Bg, H2, H, beta=1.0)                              # d hprev += dq W_h
            hip.gemm(dq, 1, H, b["hprev"], H2, 1, Gg["decoder.attn.attn.weight"], 4 * H, H, H2, Bg, beta=1.0)   # dW_h += dq^T hprev
            _colsum(dq, H, Gg["decoder.attn.attn.bias"], Bg, H)
            bar_attn.append((b["qb"].view(1, Bg, H), ds, b["attw"].view(1, Bg, T), dctx))
            d_hid_carry = dhp
            d_token = d_xbar[:, :tokw].contiguous()
            if b["keep"] is not None:
                d_token = d_token * b["keep"] / 0.9
            d_token_next = d_token
        # the bars' key / encoder-output gradients: one K = bars product per clip instead of `bars` rank-1 updates of the (T, 2H) gradient
        qs, dss, aws, dcs = [torch.cat(t, 0) for t in zip(*bar_attn)]
        _attn_deferred(eng, S, Gg, "decoder.attn", keys_g["decoder"], enc_g, qs, dss, aws, dcs, dK_g["decoder"], dEnc_g, Bg, T, H, len(bar_attn))
        # ---- initial token: <sos>/<eos> staff token (used for both staves) + time-signature / key <sos> rows
        d_sos = d_token_next.clone()
        d_sos[:, :2 * Sz] += d_sos[:, 2 * Sz:4 * Sz]
        _staff_token_bwd(eng, S, Gg, gs["sos_rec"][0], d_sos)
        for table, cid, col, width in (("decoder.time_sig_emb.weight", cfg["num_time_sig"], 4 * Sz, te), ("decoder.key_emb.weight", cfg["num_keys"], 4 * Sz + te, ke)):
            hip.check(L.a2s_embed_scatter_add(hip.stream(), hip._p(Gg[table]), NULL, NULL, C.c_long(0), cid, hip._p(d_token_next), C.c_long(tokw), col, Bg, width,
                                              NULL, hip.f32(1.0)), "scatter sos ts/key")
        d_hidden[b0:b1].copy_(d_hid_carry)
        for ev in deferred_done:                      # the staves' weight / key / encoder-output gradients are complete past this point
            torch.cuda.current_stream().wait_event(ev)
        return None


    def finish(self, grad_ready=None):
        eng, S, G, dev = self.eng, self.S, self.G, self.dev
        B, T, F = self.B, self.T, self.F
        names, offs, total, flat = self.names, self.offs, self.total, self.flat
        dEnc, dK = self.dEnc, self.dK
        sv = eng.saved
        assert sv["training"], "backward needs a forward run with training=True (batch statistics / saved activations)"
        L = hip.lib()
        H = eng.cfg["hidden_size"]
        H2 = 2 * H
        enc = sv["enc_out"]
        G["__staff_emb_ptrs__"] = self.group_ptrs[0]
        late_join = None
        if self.late:
            # The note decoders' weight gradients, on the weight-gradient stream: they start now (every closure waits for the event its call
            # recorded behind its reverse loop) and run beside the key products and the encoder's back-propagation below.  They accumulate
            # into their clip group's flat buffer; the groups' buffers are folded into `flat` on the same stream afterwards, over the decoder's
            # parameters only (nothing else is non-zero in them, and the main stream accumulates the encoder's gradients meanwhile), once the
            # key products below -- which write the other half of the decoders' attention matrices -- are done.
            dec_lo = next(off for k, off in zip(names, offs) if k.startswith("decoder."))
            assert all(k.startswith("decoder.") for k, off in zip(names, offs) if off >= dec_lo), "the decoder's parameters must be the tail of the flat layout"
            wg = _weight_grad_stream(dev)
            wg.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(wg):
                for fn, ev, tensors in self.late:
                    wg.wait_event(ev)
                    for t in tensors:
                        if t is not None:
                            t.record_stream(wg)
                    fn()
            late_join = dec_lo
            self.late = []
        else:
            for gf in self.group_flat[1:]:
                flat.add_(gf)
        d_hid_carry = self.d_hidden
        if self.concurrent:
            dEnc.add_(self.dEnc_staff[0]).add_(self.dEnc_staff[1])
        # ---- attention keys: K = enc W_e^T  ->  dW_e += dK^T enc ; dEnc += dK W_e
        enc2d = enc.view(B * T, H2)
        enc_amax = hip.absmax(enc) if self.two_term else None
        for p, dKp in dK.items():
            Wn = p + ".attn.attn.weight"
            sk = L.a2s_gemm_pick_splitk(H, H2, B * T, 1)
            dk_amax = hip.absmax(dKp) if self.two_term else None
            hip.gemm(dKp, 1, H, enc2d, H2, 1, G[Wn], 4 * H, H, H2, B * T, beta=1.0, splitk=sk, c_off=H2,
                     two_term=(dk_amax, enc_amax) if self.two_term else None)
            hip.gemm(dKp, H, 1, S[Wn], 4 * H, 1, dEnc, H2, B * T, H2, H, beta=1.0, b_off=H2,
                     two_term=(dk_amax, hip.absmax(S[Wn])) if self.two_term else None)
        if late_join is not None:
            wg = _weight_grad_stream(dev)
            wg.wait_stream(torch.cuda.current_stream())            # (the key products above)
            with torch.cuda.stream(wg):
                for gf in self.group_flat[1:]:
                    flat[late_join:].add_(gf[late_join:])
                if Backward.check_fold:                      # tests: nothing but decoder gradients may sit in the groups' own buffers
                    for gi, gf in enumerate(self.group_flat[1:], 1):
                        assert not bool(gf[:late_join].any()), f"clip group {gi}: non-decoder gradients in the group buffer would be lost by the late fold"
        # The encoder's last weight gradients (layer 0: ~5 ms of GEMMs on the weight-gradient stream, nothing of the encoder left to run beside
        # them) overlap with the START of the ConvStack backward: the caller's stream does not wait for them here.  The decoder + encoder slice
        # is announced from that stream (a collective issued there is ordered behind its work, which itself waited for everything the main
        # stream had enqueued when those GEMMs were launched), and the main stream joins it after the ConvStack backward.
        defer = True
        d_conv = _encoder_bwd(eng, S, G, sv["enc"], dEnc, d_hid_carry, B, T, wait_weight_grads=not defer)
        n_conv = next(off for k, off in zip(names, offs) if not k.startswith("convstack."))      # state_dict order: convstack first
        if grad_ready is not None:
            if defer:
                with torch.cuda.stream(_weight_grad_stream(dev)):
                    grad_ready(self.flat_full, n_conv, total + 1)
            else:
                grad_ready(self.flat_full, n_conv, total + 1)           # (+ the loss word)
        if getattr(eng, "conv_unperm", None) is not None:       # the ConvStack ran before the clips were permuted into their groups (train.TrainStep)
            d_conv = d_conv.reshape(B, T, -1).index_select(0, eng.conv_unperm)
        _convstack_bwd(eng, S, G, sv["conv"], d_conv, B, T, F)
        if defer:
            torch.cuda.current_stream().wait_stream(_weight_grad_stream(dev))
        if grad_ready is not None:
            grad_ready(flat, 0, n_conv)
        G[None] = flat
        G["__loss_gate__"] = self.flat_full[total:]
        eng._keep_alive = self.keep_alive
        return G




def backward(eng, S, grad_outputs, grad_ready=None, loss_total=None):
    """Gradients of sum_i <out_i, grad_outputs_i> wrt every parameter.  Returns dict name -> tensor (views of ONE flat buffer,
    also returned as `flat` under key None) in state_dict parameter order.
    grad_ready(flat, start, end): optional callback, called when flat[start:end] is final (everything that writes it has been
    enqueued on the current stream) -- first the encoder + decoder slice, then the ConvStack slice; train.GradientExchange starts
    the data-parallel all-reduce of a slice there, under the rest of the backward pass."""
    from .engine import group_views, run_clip_groups
    sv = eng.saved
    assert sv["training"], "backward needs a forward run with training=True (batch statistics / saved activations)"
    dev = sv["enc_out"].device
    clip_groups = sv.get("clip_groups") or [(0, sv["shape"][0])]
    ctx = Backward(eng, S, sv["shape"], dev, clip_groups, sv.get("concurrent"), sv.get("bar_major"))
    dts, dkey, dup, dlo = [g.contiguous() for g in grad_outputs]

    def group(gidx):
        b0, b1 = clip_groups[gidx]
        if ctx.bar_major:
            dup_g, dlo_g = group_views(dup, clip_groups, gidx), group_views(dlo, clip_groups, gidx)
        else:
            dup_g, dlo_g = dup[b0:b1], dlo[b0:b1]
        ctx.decoder_group(gidx, sv["groups"][gidx], dts[b0:b1], dkey[b0:b1], dup_g, dlo_g)

    run_clip_groups(dev, [lambda gi=gi: group(gi) for gi in range(len(clip_groups))])
    if loss_total is not None:
        ctx.flat_full[ctx.total:].copy_(loss_total.reshape(1))
    return ctx.finish(grad_ready)


_WG_STREAMS = {}
_DEFERRED_STREAMS = {}


def _deferred_streams(dev, group=0):
    idx = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    key = (idx, group)
    if key not in _DEFERRED_STREAMS:
        _DEFERRED_STREAMS[key] = (torch.cuda.Stream(device=idx), torch.cuda.Stream(device=idx))
    return _DEFERRED_STREAMS[key]



def _weight_grad_stream(dev):
    from .engine import group_stream
    return group_stream(dev, 1)             # the fourth stream of the budget: idle while the encoder back-propagates


def _encoder_bwd(eng, S, G, es, dEnc, d_hidden, B, T, wait_weight_grads=True):
    """wait_weight_grads=False: the caller's stream is NOT made to wait for the weight gradients on _weight_grad_stream(dev) (layer 0's run
    after its recurrence, with nothing left to hide them under): the caller waits for that stream before anything reads the encoder's
    gradient slice (Backward.finish announces the slice FROM that stream and joins it after the ConvStack backward)."""
    L = hip.lib()
    H = eng.cfg["hidden_size"]
    dev = dEnc.device
    W = S["encoder.fc.weight"]
    # bridge: hidden = tanh(pre); pre_l = fc([hf_l ; hr_l])
    dpre = torch.empty_like(d_hidden)
    hip.check(L.a2s_ew_act_bwd(hip.stream(), hip._p(d_hidden), hip._p(es["hidden"]), hip._p(dpre), C.c_long(dpre.numel()), 2), "tanh bwd")
    dhn = []
    for l in (0, 1):
        for half, hfin in enumerate((es["finals"][2 * l], es["finals"][2 * l + 1])):
            d = torch.empty((B, H), device=dev)
            hip.gemm(dpre, 2 * H, 1, W, 2 * H, 1, d, H, B, H, H, a_off=l * H, b_off=half * H)                       # d h_fin = dpre_l W[:, half]
            hip.gemm(dpre, 1, 2 * H, hfin, H, 1, G["encoder.fc.weight"], 2 * H, H, H, B, beta=1.0, a_off=l * H, c_off=half * H)   # dW += dpre_l^T h_fin
            dhn.append(d)
        _colsum(dpre, 2 * H, G["encoder.fc.bias"], B, H, x_off=l * H)
    gws = [hip.gemm_workspace(B, dev), hip.gemm_workspace(B, dev)]
    two_term = True
    in_amax = hip.absmax(es["layers"][0]["in"]) if two_term else None        # max of the ConvStack features (layer 0's input)
    dout = dEnc                                              # gradient wrt layer-1 outputs (B,T,2H)
    for layer in (1, 0):
        ls = es["layers"][layer]
        inp = ls["in"]                                       # (B*T, I)
        I = inp.shape[1]
        out = ls["out"]
        dX = torch.empty((B * T, I), device=dev)
        from .engine import encoder_streams, fork_on_streams
        streams = encoder_streams(dev, B)                 # the two directions' BPTT chains are independent: one stream (and host thread) each

        def direction(d, sfx):
            dgi = torch.empty((B, T, 3 * H), device=dev)
            dghs = torch.empty((B, T, 3 * H), device=dev)
            dgh_first, dhbuf, dgh_tmp = torch.empty((B, 3 * H), device=dev), torch.empty((2, B, H), device=dev), torch.empty((B, 3 * H), device=dev)
            hip.check(L.a2s_gru_seq_bwd(hip.stream(), _ptr(dout, d * H), C.c_long(T * 2 * H), C.c_long(2 * H), _ptr(out, d * H), C.c_long(T * 2 * H),
                                        C.c_long(2 * H), hip._p(ls["dirs"][d]["gates"]), hip._p(S[f"encoder.gru.weight_hh_{sfx}"]), hip._p(dhn[2 * layer + d]),
                                        hip._p(dgi), hip._p(dghs), hip._p(dgh_first), hip._p(dhbuf), hip._p(dgh_tmp), B, T, H, d, hip._p(gws[d]),
                                        C.c_size_t(gws[d].numel() * 4)), "a2s_gru_seq_bwd")
            return (dgi, dghs, dgh_first, dhbuf, dgh_tmp)

        res = fork_on_streams(dev, streams, [lambda d=d, sfx=sfx: direction(d, sfx) for d, sfx in enumerate((f"l{layer}", f"l{layer}_reverse"))])()
        dgi_amax = []
        # input gradient (what the next recurrence / the ConvStack needs) on the current stream ...
        for d, sfx in enumerate((f"l{layer}", f"l{layer}_reverse")):
            dgi2 = res[d][0].view(B * T, 3 * H)
            Wih = S[f"encoder.gru.weight_ih_{sfx}"]
            tt = (hip.absmax(dgi2), hip.absmax(Wih)) if two_term else None        # measured ranges: the two-term fp16 split
            dgi_amax.append(tt[0] if tt else None)
            if L.a2s_debug_get(b"gemm_bf16x3") > 0:      # k-contiguous weight copy (<= 1.5 MB): both operands on the GEMM's split-operand path
                hip.gemm(dgi2, 3 * H, 1, Wih.t().contiguous(), 1, 3 * H, dX, I, B * T, I, 3 * H, beta=0.0 if d == 0 else 1.0, two_term=tt)
            else:
                hip.gemm(dgi2, 3 * H, 1, Wih, I, 1, dX, I, B * T, I, 3 * H, beta=0.0 if d == 0 else 1.0, two_term=tt)
        # ... the weight gradients (MFMA-bound, nobody waits for them) on a third stream, under the next layer's latency-bound recurrence
        wg = _weight_grad_stream(dev)
        ev = torch.cuda.Event()
        ev.record()
        wg.wait_event(ev)
        for d, sfx in enumerate((f"l{layer}", f"l{layer}_reverse")):
            dgi, dghs, dgh_first = res[d][:3]
            # (everything the weight-gradient stream reads that was allocated on another stream -- the operand-range scalars included: freed by
            # this function's return, their blocks would be handed to the next small allocation of the main stream while these GEMMs still run)
            for t in (dgi, dghs, dgh_first, dgi_amax[d], in_amax):
                if t is not None:
                    t.record_stream(wg)
            with torch.cuda.stream(wg):
                dgi2, dghs2 = dgi.view(B * T, 3 * H), dghs.view(B * T, 3 * H)
                _linear_bwd(inp, S[f"encoder.gru.weight_ih_{sfx}"], dgi2, G, f"encoder.gru.weight_ih_{sfx}", f"encoder.gru.bias_ih_{sfx}",
                            dy_amax=dgi_amax[d], x_bound=(in_amax if layer == 0 else hip.one(dev)) if two_term else None)
                sk = L.a2s_gemm_pick_splitk(3 * H, H, B * T, 1)
                hip.gemm(dghs2, 1, 3 * H, out, 2 * H, 1, G[f"encoder.gru.weight_hh_{sfx}"], H, 3 * H, H, B * T, beta=1.0, splitk=sk, b_off=d * H,
                         two_term=(hip.absmax(dghs2), hip.one(dev)) if two_term else None)
                _colsum(dghs2, 3 * H, G[f"encoder.gru.bias_hh_{sfx}"], B * T, 3 * H)
                _colsum(dgh_first, 3 * H, G[f"encoder.gru.bias_hh_{sfx}"], B, 3 * H)
        dout = dX.view(B, T, I)
    if wait_weight_grads:
        torch.cuda.current_stream().wait_stream(_weight_grad_stream(dev))     # the encoder gradients are complete past this point
    return dout                                              # (B, T, conv_feature_size)


# How the ConvStack's backward is laid out.  Module constants (tests set them directly); the one environment switch is the documented fallback
# A2S_FUSE_BN_ROWS=0: the BatchNorm-backward apply as a tensor pass of its own instead of inside the row-streaming weight gradient.
# _FUSE_BN_APPLY: the round-2 form of that fusion (inside the tiled weight gradient a2s_conv3x3_wgrad_bn) -- parity-tested, slower (the extra operand
# pushes conv3x3_wgrad<40> from 202 to 281 registers); kept for conv1 only (_FUSE_BN_APPLY_L1), whose weight gradient is its own kernel.
_FUSE_BN_APPLY = False
_FUSE_BN_APPLY_L1 = True
_DGRAD_BNSTATS = True           # BatchNorm-backward statistics in the data-gradient conv's epilogue
# synchronised BatchNorm keeps the fused backward paths (statistics from the producers' partials, ONE small all-reduce per layer, the input gradient
# formed inside the weight-gradient kernel): round 5; False = round 4's separate statistics + apply passes
_SYNC_FUSED = True
# Where the 19200 -> 256 Linear's weight gradient runs (see _convstack_bwd).  Round 6 A/Bs (profiles/r06_ab_switches.txt, r06_ab_lin_wgrad_placement.txt): in
# front of the data gradient on the main stream 454.6 ms per step, beside it on the weight-gradient stream (round 5) 454.5, beside conv4's / conv3's /
# conv2's weight gradient 456.9 / 456.3 / 463.3 -- no placement of this 11 ms launch moves the step; the simplest is the default.
_LIN_WGRAD_AT = "before"
_FUSE_BN_ROWS = os.environ.get("A2S_FUSE_BN_ROWS", "1") != "0"        # BatchNorm-backward apply inside the row-streaming weight gradient's staging


def _linear_dgrad_generic(L, dev, rows, F, Cf, dz, Wout, da, y4, bn4, dz_amax, w_amax):
    """Data gradient of the 19200 -> Cf Linear + layer-4 BatchNorm-backward statistics on the generic two-term GEMM tiles (shapes the kernel of
    csrc/a2s_linear.hip does not take, or hip.LINEAR_KERNELS off).  Returns (partials, blocks)."""
    nblk = L.a2s_gemm_bnstats_blocks(rows, F)
    part = torch.empty((nblk, 40, 2), dtype=torch.float32, device=dev)
    if L.a2s_debug_get(b"gemm_bf16x3") > 0:
        # a k-contiguous copy of the weight (19.7 MB) puts this product on the split-operand path of the GEMM (both operands k-contiguous)
        Wt = Wout.t().contiguous()
        wb, s_bk, s_bn = Wt, 1, Cf
    else:
        wb, s_bk, s_bn = Wout, 40 * F, 1
    hip.check(L.a2s_gemm_f32_bnstats_scaled(hip.stream(), rows, 40 * F, Cf, hip._p(dz), C.c_long(Cf), C.c_long(1), hip._p(wb), C.c_long(s_bk), C.c_long(s_bn),
                                            hip._p(da), C.c_long(40 * F), hip._p(y4), hip._p(bn4[0]), hip._p(bn4[1]), hip._p(bn4[2]), hip._p(bn4[3]), F,
                                            hip._p(part), hip._p(dz_amax), hip._p(w_amax)), "a2s_gemm_f32_bnstats_scaled")
    return part, nblk


def _convstack_bwd(eng, S, G, cs, d_out, B, T, F):
    L = hip.lib()
    dev = d_out.device
    Cf = eng.cfg["conv_feature_size"]
    rows = B * T

    def bn_bwd(g, x, bn, name, mask, n_rows, C_, F_, stats_only=False, partial=None, amax=None):
        """stats_only: dgamma / dbeta and the two per-channel means (returned) only -- the input gradient is then formed inside the
        weight-gradient kernel (a2s_conv3x3_wgrad_bn).  partial = (tensor, nblocks): the statistics partials were already produced by
        the data-gradient convolution that wrote g (a2s_conv3x3_dgrad_bnstats): no statistics pass."""
        mean, invstd, scale, shift = bn
        if partial is not None and eng.sync_bn and stats_only:
            # synchronised statistics on the fused path (round 5): the producer's partials -> this rank's sums -> ONE small all-reduce ->
            # c12 from the global sums; dgamma / dbeta from the local ones (torch.nn.SyncBatchNorm's rule).  No pass over (g, x).
            import torch.distributed as dist
            local = torch.empty(2 * C_, dtype=torch.float32, device=dev)
            hip.check(L.a2s_bn_bwd_sums_from_partial(hip.stream(), hip._p(partial[0]), partial[1], C_, hip._p(local)), "a2s_bn_bwd_sums_from_partial")
            glob = local.clone()
            dist.all_reduce(glob)
            c12 = torch.empty(2 * C_, dtype=torch.float32, device=dev)
            hip.check(L.a2s_bn_bwd_c12_from_sums(hip.stream(), hip._p(local), hip._p(glob), C.c_double(eng.bn_counts[name]), hip._p(G[name + ".weight"]),
                                                 hip._p(G[name + ".bias"]), hip._p(c12), C_), "a2s_bn_bwd_c12_from_sums")
            return c12
        if partial is not None and not eng.sync_bn:
            c12 = torch.empty(2 * C_, dtype=torch.float32, device=dev)
            hip.check(L.a2s_bn_bwd_from_partial_amax(hip.stream(), hip._p(g), hip._p(x), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift),
                                                     hip._p(G[name + ".weight"]), hip._p(G[name + ".bias"]), NULL if stats_only else hip._p(g),
                                                     hip._p(partial[0]), partial[1], hip._p(c12), C.c_long(n_rows), C_, F_, hip._p(amax)),
                      "a2s_bn_bwd_from_partial")
            return c12 if stats_only else g
        part = torch.empty(L.a2s_bn_bwd_partial_floats(C.c_long(n_rows), C_, F_), dtype=torch.float32, device=dev)
        c12 = torch.empty(2 * C_, dtype=torch.float32, device=dev)
        if eng.sync_bn:                                       # statistics of the global minibatch (see Engine.__init__)
            import torch.distributed as dist
            local = torch.empty(2 * C_, dtype=torch.float32, device=dev)
            hip.check(L.a2s_bn_bwd_stats(hip.stream(), hip._p(g), hip._p(x), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift), hip._p(mask),
                                         hip.f32(1.0 / 0.8), hip._p(part), hip._p(local), C.c_long(n_rows), C_, F_), "a2s_bn_bwd_stats")
            glob = local.clone()
            dist.all_reduce(glob)
            hip.check(L.a2s_bn_bwd_apply(hip.stream(), hip._p(g), hip._p(x), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift), hip._p(mask),
                                         hip.f32(1.0 / 0.8), hip._p(local), hip._p(glob), C.c_double(eng.bn_counts[name]), hip._p(G[name + ".weight"]),
                                         hip._p(G[name + ".bias"]), hip._p(g), hip._p(c12), C.c_long(n_rows), C_, F_), "a2s_bn_bwd_apply")
            return g
        hip.check(L.a2s_bn_bwd_amax(hip.stream(), hip._p(g), hip._p(x), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift), hip._p(mask), hip.f32(1.0 / 0.8),
                                    hip._p(G[name + ".weight"]), hip._p(G[name + ".bias"]), NULL if stats_only else hip._p(g), hip._p(part), hip._p(c12),
                                    C.c_long(n_rows), C_, F_, hip._p(amax)), "a2s_bn_bwd")
        return c12 if stats_only else g                       # in place: g now holds dx

    # dropout + ReLU + BatchNorm1d over the (B*T, Cf) Linear output
    g = d_out.reshape(rows, Cf).contiguous()
    dz_amax = torch.zeros(1, dtype=torch.float32, device=dev)
    dz = bn_bwd(g, cs["z"], cs["out_bn"], "convstack.out_bn", cs["drop"], rows, Cf, 1, amax=dz_amax)
    if eng.sync_bn:
        hip.absmax(dz, dz_amax)
    w_amax = cs.get("w_out_amax")
    # Linear 19200 -> Cf, no bias:  dW += dz^T a4 ; da4 = dz W
    a4 = cs["a4"]
    g_partial = None                     # BatchNorm-backward statistics partials of g, when the kernel that produced g also reduced them
    g_amax = None                        # max |g| (device scalar), when the kernel that produced g also reduced it
    Wout = S["convstack.out.weight"]
    if a4 is None:                         # the Linear read relu(bn4(y4)) on the fly: so does its weight gradient
        y4 = cs["y"][3].view(rows, 40 * F)
        da = torch.empty_like(y4)
        bn4 = cs["bn"][3]
        def lin_wgrad():
            if not hip.linear_wgrad(dz, y4, (bn4[2], bn4[3], F), dz_amax, cs["abound"][3], G["convstack.out.weight"]):       # round 4: csrc/a2s_linear.hip
                _linear_bwd(y4, Wout, dz, G, "convstack.out.weight", None, x_affine=(bn4[2], bn4[3], F), dy_amax=dz_amax, x_bound=cs["abound"][3])
        # Where the Linear's weight gradient (reads y4 once, 10-11 ms alone; nothing but the optimizer waits for it) runs -- _LIN_WGRAD_AT:
        #   "before"  on the main stream in front of the data gradient (on the backward's critical chain);
        #   "beside"  round 5: on the weight-gradient stream beside the data gradient;
        #   4 / 3 / 2 on the weight-gradient stream from the moment layer i's weight gradient is enqueued on the main stream (beside the later, smaller
        #             launches of the chain); the main stream joins that stream at the end of the ConvStack backward.
        lin_at = _LIN_WGRAD_AT
        lin_overlap = lin_at != "before"

        def lin_wgrad_side():
            wg_stream = _weight_grad_stream(dev)
            ev = torch.cuda.Event()
            ev.record()
            wg_stream.wait_event(ev)
            with torch.cuda.stream(wg_stream):
                lin_wgrad()
        if lin_at == "beside":
            lin_wgrad_side()
        elif lin_at == "before":
            lin_wgrad()
        if _DGRAD_BNSTATS and (not eng.sync_bn or _SYNC_FUSED) and F >= 128 and F % 4 == 0 and rows > 64:      # (the epilogue lives in the 128-row GEMM tile)
            # data gradient of the Linear with the layer-4 BatchNorm-backward statistics accumulated in the GEMM's epilogue
            if w_amax is None:
                w_amax = hip.absmax(Wout)
            if hip.LINEAR_KERNELS and L.a2s_linear_dgrad_eligible(rows, 40 * F, Cf, F):
                # round 4: the kernel of its own (csrc/a2s_linear.hip): the weight pre-split once per launch, a workgroup's rows of dz resident in LDS
                # for all column tiles, no barrier in the sweep
                nblk = L.a2s_linear_dgrad_blocks(rows)
                part = torch.empty((nblk, 40, 2), dtype=torch.float32, device=dev)
                Wt = Wout.t().contiguous()
                nb = L.a2s_linear_dgrad_ws_bytes(40 * F, Cf)
                lws = torch.empty(nb // 4, dtype=torch.float32, device=dev)
                g_amax = torch.zeros(1, dtype=torch.float32, device=dev)          # max |da|: the range of the BatchNorm backward fused into conv4's weight gradient
                hip.check(L.a2s_linear_dgrad_bnstats(hip.stream(), rows, 40 * F, Cf, hip._p(dz), C.c_long(Cf), hip._p(Wt), hip._p(da), C.c_long(40 * F), hip._p(y4),
                                                     hip._p(bn4[0]), hip._p(bn4[1]), hip._p(bn4[2]), hip._p(bn4[3]), F, hip._p(part), hip._p(dz_amax), hip._p(w_amax),
                                                     hip._p(lws), C.c_size_t(nb), hip._p(g_amax)), "a2s_linear_dgrad_bnstats")
                g_partial = (part, nblk)
            else:
                g_partial = _linear_dgrad_generic(L, dev, rows, F, Cf, dz, Wout, da, y4, bn4, dz_amax, w_amax)
        else:
            hip.gemm(dz, Cf, 1, Wout, 40 * F, 1, da, 40 * F, rows, 40 * F, Cf)
    else:
        da = torch.empty_like(a4)
        _linear_bwd(a4, Wout, dz, G, "convstack.out.weight", None, dx=da)
    chans = [(1, 20), (20, 20), (20, 40), (40, 40)]
    g = da.view(B, T, 40, F)
    for i in (4, 3, 2, 1):
        ci, co = chans[i - 1]
        y = cs["y"][i - 1]                                    # pre-BN conv output of this layer
        x_in = cs["y"][i - 2] if i > 1 else cs["x0"]
        in_bn = cs["bn"][i - 2] if i > 1 else None
        nb = L.a2s_conv3x3_wgrad_workspace_bytes(ci, co)
        ws = torch.empty(nb // 4, dtype=torch.float32, device=dev)
        # the first layer has no data-gradient consumer: its BatchNorm input gradient is only read by the (streaming, HBM-bound)
        # weight-gradient kernel, which forms it on the fly -- one pass over (g, y) instead of apply (read 2, write 1) + read 1
        fuse_here = _FUSE_BN_APPLY or (i == 1 and _FUSE_BN_APPLY_L1)
        # max |dy| of this layer's output gradient, reduced by the kernel that writes dy: the two-term fp16 data-gradient convolution
        # below scales its operand by the matching power of two (gradients would otherwise sit in fp16's subnormal range)
        dy_amax = torch.zeros(1, dtype=torch.float32, device=dev) if (i > 1 and (not eng.sync_bn or _SYNC_FUSED)) else None
        fuse_rows = (_FUSE_BN_ROWS and (not eng.sync_bn or _SYNC_FUSED) and i > 1 and g_amax is not None and g_partial is not None and in_bn is not None
                     and L.a2s_conv3x3_wgrad_bn_ranged_eligible(F, ci, co))
        if a4 is None and lin_at == i:
            lin_wgrad_side()
        if fuse_rows:
            # round 4: BatchNorm backward as statistics only; dy is formed by the STAGING waves of the row-streaming weight-gradient kernel (they
            # wait 40-57 % of their time for the multiply waves), written once for the data-gradient convolution: one pass over (g, y) less
            bn_i = cs["bn"][i - 1]
            c12 = bn_bwd(g, y, bn_i, f"convstack.bn{i}", None, rows, co, F, stats_only=True, partial=g_partial)
            dy = g                                                 # in place: every element is read and written once, by the same thread
            hip.check(L.a2s_conv3x3_wgrad_bn_ranged(hip.stream(), hip._p(g), hip._p(y), hip._p(bn_i[0]), hip._p(bn_i[1]), hip._p(bn_i[2]), hip._p(bn_i[3]),
                                                    hip._p(c12), hip._p(g_amax), g_amax.numel(), hip._p(cs["yabs"][i - 1]), hip._p(dy), hip._p(dy_amax), hip._p(x_in),
                                                    hip._p(in_bn[2]), hip._p(in_bn[3]), hip._p(G[f"convstack.conv{i}.weight"]), hip._p(ws), C.c_size_t(nb),
                                                    B, T, F, ci, co, hip._p(cs["abound"][i - 2])), "a2s_conv3x3_wgrad_bn_ranged")
        elif eng.sync_bn or not fuse_here:
            dy = bn_bwd(g, y, cs["bn"][i - 1], f"convstack.bn{i}", None, rows, co, F, partial=g_partial, amax=dy_amax)
            if eng.sync_bn and dy_amax is not None:            # (the synchronised apply pass does not reduce the range of what it writes)
                hip.absmax(dy, dy_amax)
            hip.conv3x3_wgrad(dy, x_in.view(B, T, ci, F), in_bn[2] if in_bn else None, in_bn[3] if in_bn else None, G[f"convstack.conv{i}.weight"], ws,
                              dy_amax, cs["abound"][i - 2] if in_bn else None)
        else:
            # BatchNorm backward: statistics pass only; dy = scale (g' - c1 - xhat c2) is formed by the weight-gradient kernel while
            # it stages its dy operand (MFMA-bound, HBM to spare) and written out for the data-gradient convolution below
            bn_i = cs["bn"][i - 1]
            c12 = bn_bwd(g, y, bn_i, f"convstack.bn{i}", None, rows, co, F, stats_only=True, partial=g_partial)
            dy = torch.empty_like(g) if i > 1 else None
            hip.check(L.a2s_conv3x3_wgrad_bn(hip.stream(), hip._p(g), hip._p(y), hip._p(bn_i[0]), hip._p(bn_i[1]), hip._p(bn_i[2]), hip._p(bn_i[3]),
                                             hip._p(c12), hip._p(dy), hip._p(x_in), hip._p(in_bn[2]) if in_bn else NULL, hip._p(in_bn[3]) if in_bn else NULL,
                                             hip._p(G[f"convstack.conv{i}.weight"]), hip._p(ws), C.c_size_t(nb), B, T, F, ci, co), "a2s_conv3x3_wgrad_bn")
        if i > 1:
            gprev = torch.empty((B, T, ci, F), dtype=torch.float32, device=dev)
            cws = hip.conv_workspace(co, dev)
            if _DGRAD_BNSTATS and (not eng.sync_bn or _SYNC_FUSED):
                # the data-gradient convolution also accumulates the BatchNorm-backward statistics of the layer below in its epilogue
                bn_l = cs["bn"][i - 2]
                nblk = L.a2s_conv3x3_stat_blocks(B, T, F, co)
                part = torch.empty((nblk, ci, 2), dtype=torch.float32, device=dev)
                g_amax_next = torch.empty(ci, dtype=torch.float32, device=dev) if (_FUSE_BN_ROWS and i > 2) else None      # max |g| per channel (zeroed by the launch)
                hip.check(L.a2s_conv3x3_dgrad_bnstats_ranged(hip.stream(), hip._p(dy), hip._p(S[f"convstack.conv{i}.weight"]), hip._p(gprev),
                                                             hip._p(cs["y"][i - 2]), hip._p(bn_l[0]), hip._p(bn_l[1]), hip._p(bn_l[2]), hip._p(bn_l[3]),
                                                             hip._p(part), B, T, F, co, ci, hip._p(cws),
                                                             hip._p(dy_amax) if not fuse_here else NULL, hip._p(g_amax_next)), "a2s_conv3x3_dgrad_bnstats")
                g_partial = (part, nblk)
            else:
                hip.check(L.a2s_conv3x3(hip.stream(), hip._p(dy), hip._p(S[f"convstack.conv{i}.weight"]), hip._p(gprev), NULL, NULL, NULL, B, T, F, co, ci, 1,
                                        hip._p(cws)), "a2s_conv3x3 dgrad")
                g_partial = None
            g = gprev
            g_amax = g_amax_next if (_DGRAD_BNSTATS and (not eng.sync_bn or _SYNC_FUSED)) else None
    if a4 is None and lin_overlap:
        torch.cuda.current_stream().wait_stream(_weight_grad_stream(dev))      # the Linear's weight gradient (issued beside the data gradient above)
