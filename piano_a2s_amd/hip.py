"""ctypes binding of liba2s_hip.so (include/a2s.h).  There is NO fallback: if the library is missing or a
call fails, this raises -- the product path never silently runs anything else."""
import ctypes as C
import os

import torch

from .build import LIB

LIB = os.environ.get("A2S_LIB", LIB)          # A/B measurements against an older build of the library

_lib = None


class A2SError(RuntimeError):
    pass


class NoteDecArgs(C.Structure):
    """Mirror of `a2s_note_dec_args` (include/a2s.h) -- same members, same order."""
    _fields_ = [(n, C.c_void_p) for n in (
        "attn_w", "attn_b", "attn_v", "w_ih", "w_hh", "b_ih", "b_hh", "out_w", "out_b", "emb", "keys", "enc",
        "h", "x", "q", "gates", "attw", "o", "gh", "gi", "logits")] + [
        ("probs", C.c_void_p), ("probs_bstride", C.c_long),
        ("gt", C.c_void_p), ("gt_bstride", C.c_long),
        ("tf_flags", C.c_void_p),
        ("drop", C.c_void_p), ("inv_keep", C.c_float),
        ("argmax_out", C.c_void_p), ("am_bstride", C.c_long),
        ("eos_seen", C.c_void_p), ("lengths", C.c_void_p), ("n_done", C.c_void_p), ("steps_exec", C.c_void_p), ("attn_ws", C.c_void_p),
        ("gemm_ws", C.c_void_p), ("gemm_ws_bytes", C.c_size_t), ("t_base", C.c_void_p), ("clip_order", C.c_void_p), ("clip_rank", C.c_void_p),
        ("row_until", C.c_void_p), ("n_active", C.c_void_p), ("n_clips", C.c_int), ("m_active", C.c_void_p),
        ("row_list", C.c_void_p), ("n_rows_active", C.c_void_p),
        ("R", C.c_int), ("T", C.c_int), ("H", C.c_int), ("E", C.c_int), ("V", C.c_int),
        ("steps", C.c_int), ("poll", C.c_int), ("eos_id", C.c_int), ("use_graph", C.c_int),
        ("step_ws", C.c_void_p), ("step_ws_floats", C.c_size_t),
        ("tf_flags_dev", C.c_void_p), ("persist_ws", C.c_void_p), ("persist_ws_bytes", C.c_size_t)]


class NoteDecBwdArgs(C.Structure):
    """Mirror of `a2s_note_dec_bwd_args` (include/a2s.h) -- same members, same order."""
    _fields_ = [(n, C.c_void_p) for n in (
        "attn_w", "attn_v", "w_ih", "w_hh", "keys", "enc", "h", "x", "q", "gates", "attw", "do_all",
        "dgi_all", "dgh_all", "dq_all", "ds_all", "dctx_all", "dx", "dh", "attn_ws", "gemm_ws")] + [
        ("gemm_ws_bytes", C.c_size_t), ("clip_order", C.c_void_p), ("clip_rank", C.c_void_p), ("row_until", C.c_void_p),
        ("n_active", C.c_void_p), ("n_clips", C.c_int), ("m_active", C.c_void_p), ("row_list", C.c_void_p), ("n_rows_active", C.c_void_p),
        ("R", C.c_int), ("T", C.c_int), ("H", C.c_int), ("E", C.c_int), ("steps", C.c_int),
        ("step_ws", C.c_void_p), ("step_ws_floats", C.c_size_t), ("persist_ws", C.c_void_p), ("persist_ws_bytes", C.c_size_t), ("w_ih_full", C.c_void_p)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise A2SError(f"HIP extension not built: {LIB} is missing (run `python __graft_entry__.py build`). "
                           "There is no CPU fallback for the transcription hot path.")
        _lib = C.CDLL(LIB)
        _lib.a2s_last_error.restype = C.c_char_p
        _lib.a2s_launch_count.restype = C.c_longlong
        for fn in ("a2s_note_step_workspace_floats", "a2s_note_decoder_persist_ws_bytes", "a2s_note_decoder_bwd_persist_ws_bytes", "a2s_linear_dgrad_ws_bytes", "a2s_linear_wgrad_ws_bytes", "a2s_gemm_workspace_bytes", "a2s_bn_bwd_partial_floats", "a2s_conv3x3_wgrad_workspace_bytes", "a2s_attn_workspace_floats", "a2s_attn_workspace_floats_fused",
                   "a2s_conv3x3_workspace_floats"):
            getattr(_lib, fn).restype = C.c_size_t
        # A2S_ARITH: the arithmetic of the dense contractions -- "f16x2" (default: two exact fp16 terms per fp32 operand, three
        # products), "bf16x3" (rounds 1-2: three bf16 terms, six products) or "f32" (fp32-input matrix instructions / vector FMAs); per-kernel keys:
        # a2s_debug_set("conv_f16x2" | "wgrad_f16x2" | "gemm_f16x2" | "conv_bf16x3" | "gemm_bf16x3" | "wgrad_bf16x3", n)
        arith = os.environ.get("A2S_ARITH", "f16x2")
        if arith not in ("f16x2", "bf16x3", "f32"):
            raise A2SError(f"A2S_ARITH={arith!r}: expected f16x2, bf16x3 or f32")
        if arith != "f16x2":
            for key in (b"conv_f16x2", b"wgrad_f16x2", b"gemm_f16x2"):
                _lib.a2s_debug_set(key, 0)
        if arith == "f32":
            for key in (b"conv_bf16x3", b"gemm_bf16x3", b"wgrad_bf16x3"):
                _lib.a2s_debug_set(key, 0)
        # documented fallbacks: the row-streaming convolutions, the few-row decoder path, the persistent note decoder
        for env, key in (("A2S_CONV_ROWS", b"conv_rows"), ("A2S_DEC_FUSED", b"dec_fused"), ("A2S_DEC_PERSIST", b"dec_persist")):
            if os.environ.get(env):
                _lib.a2s_debug_set(key, int(os.environ[env]))
    return _lib


# the 19200 -> 256 Linear on the kernels of csrc/a2s_linear.hip (A2S_LINEAR_KERNELS=0: the generic two-term GEMM tiles; bench.py times both)
LINEAR_KERNELS = os.environ.get("A2S_LINEAR_KERNELS", "1") != "0"

_ABORT_LATCH = {}
PERSIST_ABORTS = 0          # persistent launches of this process that gave up a bounded wait (observed through the latch)


def abort_latch(device):
    """The device word the persistent kernels OR a bit into when one of their bounded waits gives up (a2s_persist_abort_latch): allocated
    and registered on first use; one process drives one GPU, so the library keeps one pointer."""
    idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    t = _ABORT_LATCH.get(idx)
    if t is None:
        t = _ABORT_LATCH[idx] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
        _ABORT_LATCH["registered"] = None
    if _ABORT_LATCH.get("registered") != idx:
        check(lib().a2s_persist_abort_latch(C.c_void_p(t.data_ptr())), "a2s_persist_abort_latch")
        _ABORT_LATCH["registered"] = idx
    return t


_LATCH_READS = {}         # device index -> [pinned host word, event of the copy in flight or None]


def post_persist_abort_read(device):
    """Enqueue an asynchronous read of the abort latch behind everything the current stream holds (train.TrainStep: at the end of a step).  The
    NEXT step looks at the host copy with poll_persist_abort -- no synchronisation: a blocking read at the start of Engine.forward made the host
    wait for the ConvStack the fused step had already enqueued (48 ms) before it planned the decoder and enqueued the encoder (round 5)."""
    idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    e = _LATCH_READS.get(idx)
    if e is None:
        e = _LATCH_READS[idx] = [torch.zeros(1, dtype=torch.int32).pin_memory(), None]
    if e[1] is not None:
        # the previous step's read was never consumed (a loop without a host synchronisation per step: the poll at the start of this step came
        # before that copy had completed).  It is a whole step old -- waiting for it costs nothing -- and must not be overwritten unseen: an
        # abort would otherwise never be noticed and every following update would be skipped for a NaN loss.
        e[1].synchronize()
        _consume_latch_read(e, device, False)
    e[0].copy_(abort_latch(device), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    e[1] = ev


def poll_persist_abort(device, raise_error=False):
    """Non-blocking form of check_persist_abort: acts on the latch value read by the last post_persist_abort_read once that copy has completed
    (an abort is then noticed one step later at worst; the step it happened in skipped its update on the device anyway).  Before the first
    posted read it falls back to the blocking check (the stream is idle then)."""
    idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    e = _LATCH_READS.get(idx)
    if e is None:
        return check_persist_abort(device, raise_error)
    if e[1] is None or not e[1].query():
        return 0                          # (still in flight: it stays pending -- the next poll or the next post consumes it)
    return _consume_latch_read(e, device, raise_error)


def _consume_latch_read(e, device, raise_error):
    e[1] = None
    bits = int(e[0][0])
    if not bits:
        return 0
    return _persist_abort_seen(bits, abort_latch(device), raise_error)


def check_persist_abort(device, raise_error=False):
    """Call where the host has just synchronised with the device anyway (a 4-byte read).  If a persistent launch gave up since the last call:
    its outputs were poisoned (NaN loss -> the update was skipped) -- switch the persistent paths off for the rest of the process (the chip is
    evidently shared or partitioned in a way the residency check cannot see) and warn, or raise (greedy decoding: the ids are unusable)."""
    t = abort_latch(device)
    bits = int(t.item())
    if not bits:
        return 0
    return _persist_abort_seen(bits, t, raise_error)


def _persist_abort_seen(bits, t, raise_error):
    global PERSIST_ABORTS
    t.zero_()
    PERSIST_ABORTS += 1
    L = lib()
    L.a2s_debug_set(b"gru_persist", 0)
    L.a2s_debug_set(b"dec_persist", 0)
    os.environ["A2S_DEC_PERSIST"] = "0"
    os.environ["A2S_GRU_PERSIST"] = "0"
    msg = (f"a persistent kernel gave up a bounded wait (latch bits {bits:#x}: 1/2 encoder fwd/bwd, 4/8 note decoder fwd/bwd); its outputs were "
           "poisoned with NaN and the persistent paths are switched off for the rest of this process (launch-per-step kernels from now on)")
    if raise_error:
        raise A2SError(msg)
    import warnings
    warnings.warn(msg, RuntimeWarning)
    return bits


def _p(t):
    """device pointer of a tensor (None -> NULL); tensors must be CUDA(HIP) and of the dtype the C side expects."""
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, int):
        return C.c_void_p(t)
    if not t.is_cuda:
        raise A2SError("liba2s_hip operates on device memory only: got a CPU tensor (no CPU fallback exists)")
    return C.c_void_p(t.data_ptr())


def attn_workspace(B, T, H, device, groups=1):
    """Scratch for the split-T attention kernels (None when the one-workgroup-per-clip kernels are used); groups: fused bars per call."""
    if H != 256:
        return None
    # zero-initialised: the head of the workspace holds the arrival counters of the fused combine (left at zero by every launch)
    return torch.zeros(lib().a2s_attn_workspace_floats_fused(B, T, H, groups), dtype=torch.float32, device=device)


def conv_workspace(cin, device):
    n = lib().a2s_conv3x3_workspace_floats(cin)
    return torch.empty(n, dtype=torch.float32, device=device) if n else None


def gemm_workspace(rows, device):
    """Split-K scratch for the per-step skinny GEMMs of the decoder loops (16 slabs of rows x 2048 floats)."""
    return torch.empty(16 * max(rows, 1) * 2048, dtype=torch.float32, device=device)


def step_workspace(H, E, device):
    """Scratch of the fused few-row decoder step kernels (csrc/a2s_step.hip): flags, ticket counters (must start at zero), logits,
    transposed weight copies."""
    return torch.zeros(lib().a2s_note_step_workspace_floats(H, E), dtype=torch.float32, device=device)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise A2SError(f"{what} failed ({rc}): {lib().a2s_last_error().decode()}")


def f32(x):
    return C.c_float(float(x))


def gemm(A, sAm, sAk, B, sBk, sBn, Cout, ldc, M, N, K, alpha=1.0, beta=0.0, bias=None, act=0,
         batch=1, bsA=0, bsB=0, bsC=0, splitk=1, a_off=0, b_off=0, c_off=0, a_affine=None, b_affine=None, two_term=None):
    """C[m,n] = act(alpha*sum_k A(m,k)B(k,n) + beta*C + bias[n]); *_off are element offsets into the tensors.
    a_affine / b_affine = (scale, shift, period): BatchNorm+ReLU of that operand applied while it is staged (a2s_gemm_f32_affine).
    two_term = (a_absmax, b_absmax): device scalars max|A| / max|B| (None: the operand is O(1)) -- the product may run on the two-term
    fp16 split (a2s_gemm_f32_affine_scaled)."""
    L = lib()
    ws, ws_bytes = None, 0
    if splitk > 1:
        ws_bytes = L.a2s_gemm_workspace_bytes(M, N, batch, splitk)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=Cout.device)
    pa = C.c_void_p(A.data_ptr() + 4 * a_off)
    pb = C.c_void_p(B.data_ptr() + 4 * b_off)
    pc = C.c_void_p(Cout.data_ptr() + 4 * c_off)
    if a_affine is None and b_affine is None and two_term is None:
        check(L.a2s_gemm_f32(stream(), M, N, K, f32(alpha), pa, C.c_long(sAm), C.c_long(sAk), pb, C.c_long(sBk), C.c_long(sBn),
                             f32(beta), pc, C.c_long(ldc), _p(bias), act, batch, C.c_long(bsA), C.c_long(bsB), C.c_long(bsC),
                             splitk, _p(ws), C.c_size_t(ws_bytes)), "a2s_gemm_f32")
        return
    asc, ash, ap = a_affine if a_affine is not None else (None, None, 0)
    bsc, bsh, bp = b_affine if b_affine is not None else (None, None, 0)
    if two_term is not None:
        check(L.a2s_gemm_f32_affine_scaled(stream(), M, N, K, f32(alpha), pa, C.c_long(sAm), C.c_long(sAk), pb, C.c_long(sBk), C.c_long(sBn),
                                           f32(beta), pc, C.c_long(ldc), _p(bias), act, batch, C.c_long(bsA), C.c_long(bsB), C.c_long(bsC),
                                           splitk, _p(ws), C.c_size_t(ws_bytes), _p(asc), _p(ash), ap, _p(bsc), _p(bsh), bp,
                                           _p(two_term[0]), _p(two_term[1])), "a2s_gemm_f32_affine_scaled")
        return
    check(L.a2s_gemm_f32_affine(stream(), M, N, K, f32(alpha), pa, C.c_long(sAm), C.c_long(sAk), pb, C.c_long(sBk), C.c_long(sBn),
                                f32(beta), pc, C.c_long(ldc), _p(bias), act, batch, C.c_long(bsA), C.c_long(bsB), C.c_long(bsC),
                                splitk, _p(ws), C.c_size_t(ws_bytes), _p(asc), _p(ash), ap, _p(bsc), _p(bsh), bp), "a2s_gemm_f32_affine")


_ONES = {}


def one(device):
    """Device scalar 1.0: the range of an operand bounded by 1 (GRU states, softmax weights) for the two-term fp16 products."""
    key = torch.device(device).index
    if key not in _ONES:
        _ONES[key] = torch.ones(1, dtype=torch.float32, device=device)
    return _ONES[key]


def absmax(x, out=None):
    """max |x| of a contiguous float32 tensor as a device scalar (operand scale of the two-term fp16 kernels)."""
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=x.device)
    check(lib().a2s_absmax(stream(), _p(x), C.c_long(x.numel()), _p(out)), "a2s_absmax")
    return out


def linear(x2d, weight, bias=None, act=0, out=None, beta=0.0, x_affine=None, two_term=None):
    """y = act(x @ weight.T + bias) for row-major contiguous x (M,K) and weight (N,K); x_affine = (scale, shift, period): the input
    is max(0, x*scale[k // period] + shift[k // period]) formed on the fly."""
    M, K = x2d.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x2d.device)
    gemm(x2d, x2d.stride(0), x2d.stride(1), weight, weight.stride(1), weight.stride(0), out, out.stride(0), M, N, K,
         bias=bias, act=act, beta=beta, a_affine=x_affine, two_term=two_term)
    return out


# ---- the ConvStack's launches as the engine issues them (one place: engine.py / engine_bwd.py and the robustness tests share these)
def conv3x3_forward(x, w, y, scale, shift, partial, cws, in_absmax=None, out_absmax=None):
    """y = conv3x3(relu(x * scale[c] + shift[c])) (scale None: x as it is), (B, T, C, F) tensors; batch-statistics partials in `partial`;
    in_absmax / out_absmax: per-channel max |x| (from the launch that produced x) / max |y| (written here) -- a2s_conv3x3_ranged."""
    B, T, Cin, F = x.shape
    check(lib().a2s_conv3x3_ranged(stream(), _p(x), _p(w), _p(y), _p(scale), _p(shift), _p(in_absmax), _p(partial), _p(out_absmax),
                                   B, T, F, Cin, y.shape[2], _p(cws)), "a2s_conv3x3_ranged")


def conv3x3_dgrad_for_test(dy, w, yl):
    """Data gradient of a (Cin -> Cout) layer as engine_bwd issues it: dy (B, T, Cout, F) -> dx (B, T, Cin, F), operand scaled by the
    power of two derived from max|dy|, BatchNorm-backward statistics epilogue against `yl` (identity BatchNorm parameters here)."""
    L = lib()
    B, T, Cout, F = dy.shape
    Cin = w.shape[1]
    dev = dy.device
    dx = torch.full((B, T, Cin, F), float("nan"), device=dev)
    amax = absmax(dy)
    part = torch.zeros(L.a2s_conv3x3_stat_blocks(B, T, F, Cout), Cin, 2, device=dev)
    cws = conv_workspace(Cout, dev)
    zeros, ones = torch.zeros(Cin, device=dev), torch.ones(Cin, device=dev)
    check(L.a2s_conv3x3_dgrad_bnstats_scaled(stream(), _p(dy), _p(w), _p(dx), _p(yl), _p(zeros), _p(ones), _p(ones), _p(zeros), _p(part),
                                             B, T, F, Cout, Cin, _p(cws), _p(amax)), "a2s_conv3x3_dgrad_bnstats_scaled")
    torch.cuda.synchronize()
    return dx


def act_bound(scale, shift, x_absmax):
    """Device scalar bounding relu(x * scale[c] + shift[c]) over the tensor, from the per-channel max |x| its producer wrote."""
    out = torch.empty(1, dtype=torch.float32, device=scale.device)
    check(lib().a2s_act_bound(stream(), _p(scale), _p(shift), _p(x_absmax), scale.numel(), _p(out)), "a2s_act_bound")
    return out


def conv3x3_wgrad(dy, x, scale, shift, dW, ws, dy_absmax, act_absmax):
    """dW += weight gradient of a 3x3 layer from dy (B, T, Cout, F) and the layer input relu(x * scale + shift) (scale None: x as it is);
    dy_absmax / act_absmax: device scalars max |dy| / bound of the activated input (None: unknown -> that operand unscaled)."""
    B, T, Cout, F = dy.shape
    Cin = x.shape[2]
    check(lib().a2s_conv3x3_wgrad_ranged(stream(), _p(dy), _p(x), _p(scale), _p(shift), _p(dW), _p(ws), C.c_size_t(ws.numel() * 4), B, T, F, Cin, Cout,
                                          _p(dy_absmax), _p(act_absmax)), "a2s_conv3x3_wgrad_ranged")


def conv3x3_wgrad_for_test(dy, x, scale, shift):
    """Weight gradient as engine_bwd issues it: dW (Cout, Cin, 3, 3) from dy (B, T, Cout, F) and the layer input relu(x * scale + shift),
    with both operand ranges (max |dy|; the activation bound from the input's per-channel max)."""
    L = lib()
    B, T, Cout, F = dy.shape
    Cin = x.shape[2]
    dev = dy.device
    dW = torch.zeros(Cout, Cin, 3, 3, device=dev)
    ws = torch.empty(L.a2s_conv3x3_wgrad_workspace_bytes(Cin, Cout) // 4, device=dev)
    bound = act_bound(scale, shift, x.abs().amax(dim=(0, 1, 3)).contiguous())
    conv3x3_wgrad(dy, x, scale, shift, dW, ws, absmax(dy), bound)
    torch.cuda.synchronize()
    return dW


def linear_forward(x2d, weight, x_affine, x_bound, w_absmax, out=None):
    """The ConvStack's 19200 -> 256 Linear forward as the engine issues it: the kernel of its own (csrc/a2s_linear.hip: weight pre-split once,
    activations through a four-stage LDS ring) where the shape qualifies (hip.LINEAR_KERNELS / A2S_LINEAR_KERNELS=0: never), the generic two-term GEMM tile otherwise."""
    M, K = x2d.shape
    N = weight.shape[0]
    L = lib()
    period = x_affine[2]
    if (LINEAR_KERNELS and x2d.is_contiguous() and weight.is_contiguous() and x_bound is not None
            and L.a2s_linear_fwd_eligible(M, N, K, period)):
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=x2d.device)
        nb = L.a2s_linear_dgrad_ws_bytes(N, K)
        ws = torch.empty(nb // 4, dtype=torch.float32, device=x2d.device)
        check(L.a2s_linear_fwd(stream(), M, N, K, _p(x2d), C.c_long(K), _p(weight), _p(out), C.c_long(N), _p(x_affine[0]), _p(x_affine[1]), period,
                               _p(x_bound), _p(w_absmax), _p(ws), C.c_size_t(nb)), "a2s_linear_fwd")
        return out
    return linear(x2d, weight, out=out, x_affine=x_affine, two_term=(x_bound, w_absmax))


def linear_wgrad(dz, x2d, x_affine, dz_absmax, x_bound, G):
    """G (N, K) += dz^T relu(bn(x)) for the ConvStack's 19200 -> 256 Linear on the kernel of csrc/a2s_linear.hip; returns False when the shape
    does not qualify (hip.LINEAR_KERNELS off: never) -- the caller then runs the generic split-K GEMM."""
    M, K = x2d.shape
    N = dz.shape[1]
    L = lib()
    if (not LINEAR_KERNELS or not (x2d.is_contiguous() and dz.is_contiguous() and G.is_contiguous()) or dz_absmax is None
            or x_bound is None or not L.a2s_linear_wgrad_eligible(M, N, K, x_affine[2])):
        return False
    nb = L.a2s_linear_wgrad_ws_bytes(M, K)
    ws = torch.empty(nb // 4, dtype=torch.float32, device=x2d.device)
    check(L.a2s_linear_wgrad(stream(), M, N, K, _p(dz), C.c_long(N), _p(x2d), C.c_long(K), _p(G), C.c_long(K), _p(x_affine[0]), _p(x_affine[1]), x_affine[2],
                             _p(dz_absmax), _p(x_bound), _p(ws), C.c_size_t(nb)), "a2s_linear_wgrad")
    return True


def linear_forward_for_test(x, w, aff):
    """The 19200 -> 256 Linear's forward as engine.convstack issues it (operand BatchNorm+ReLU while staging, two-term split with the
    activation bound and max |w| as operand ranges)."""
    period = aff[2]
    xmax = x.view(x.shape[0], -1, period).abs().amax(dim=(0, 2)).contiguous()
    return linear_forward(x, w, aff, act_bound(aff[0], aff[1], xmax), absmax(w))
