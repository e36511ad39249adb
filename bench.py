#!/usr/bin/env python3
"""Headline benchmark: training clips/s of the piano-a2s hot path on MI355X (BASELINE.json metric).

A "step" = one full optimizer step of the recipe on one minibatch of synthetic 12 s / 5-bar clips with the
hparams/pretrain.yaml model (16.36 M parameters, 1201 x 480 spectrogram frames): forward, 4-term NLL objective,
backward, (gradient all-reduce,) clip_grad_norm_(5.0) + Adadelta -- piano_a2s_amd.train.TrainStep, all on liba2s_hip.so.

  python bench.py --gpus 1 --steps K --warmup W                      (single GPU)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W   (one rank per GPU, RCCL)

Rank 0 prints ONE JSON line.  Workload (config.workload): BASELINE.json configs[1], the pretrain.yaml model on synthetic clips as
SURVEY.md 8(d) specifies them -- including the 1 % tail of full-length (398 / 189 token, no <eos>) bars; `tail_off` repeats the
measurement without that tail (round 1's default workload).  `roofline` describes the 40-channel 3x3 convolution (two-term fp16 MFMA), measured
live on conv4's forward launch; `roofline_phases` (round 5) EVERY phase of the step -- ConvStack forward / backward, encoder forward / backward,
decoder -- against both roofs, from HIP events in this very run; `roofline_attention` the HBM-bound additive-attention
step (streams a clip's keys and encoder outputs once per decode step); `loss_parity` checks the full-size model's loss against the
reference's own CPU numbers in this very run; `cpu_baseline` is the oracle's as-written CPU restatement of the same training step timed
on this box's host cores on a bounded sample (a reported baseline, not the target); with N > 1 `data_parallel` lists each rank's decode
steps and all-reduce wait (the straggler terms of SURVEY 8e).
"""
import argparse
import ctypes as C
import json
import os
import random
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFS = 2500.0    # dense bf16 MFMA peak (same guide, "Peak BF16/FP16 MFMA ~2.5 PF dense")
MFMA_F32_PEAK_TFS = 157.3      # dense fp32-input MFMA peak (same guide: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD)
TF_RATIO = 0.7                 # hparams/pretrain.yaml teacher_forcing_ratio at epoch 0
N_MINIBATCHES = 8              # distinct synthetic minibatches the timed steps cycle over
# Teacher-forcing coin stream across data-parallel ranks: "rank_offset" = random.seed(1234 + rank) (the reference's
# processes draw independently), "shared" = every rank seeds Python's random alike.  dp_straggler_simulation measures
# both: the spread between ranks comes from the data, not from the coins (0.90 either way), so the reference's behaviour stays.
COIN_POLICY = os.environ.get("A2S_COIN_POLICY", "rank_offset")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-reserve", action="store_true", help="do not reserve the caching allocator's per-stream pools before the first step")
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU per step")
    ap.add_argument("--full-tail", type=float, default=0.01, help="probability of a full-length (no <eos>) row per (clip,bar,staff); SURVEY 8d: 1 %%")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the tail-off secondary measurement and the loss-parity check")
    ap.add_argument("--cpu-clips", type=int, default=4)
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-launch", action="store_true", help="with --gpus N > 1 and no torchrun environment: print the launcher command as JSON and exit")
    ap.add_argument("--launcher", action="store_true", help="start the ranks through torch.distributed.run even for --gpus 1 (RCCL world of one)")
    ap.add_argument("--master-port", type=int, default=int(os.environ.get("MASTER_PORT", "29533")))
    ap.add_argument("--no-straggler-sim", action="store_true", help="skip the 1-GPU data-parallel straggler simulation (predicted_dp_efficiency)")
    ap.add_argument("--no-inference", action="store_true", help="skip the greedy-decode (config 5) block")
    return ap.parse_args(argv)


def launcher_command(args, argv):
    """The command a bare `python bench.py --gpus N` (N > 1, no RANK in the environment) starts as a CHILD process: one rank per GPU
    under torch.distributed.run, the reference's own launch shape (reference README.md:119-122, pretrain.py:257)."""
    rest = [a for a in argv if a not in ("--dry-launch", "--launcher")]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(args.master_port), os.path.abspath(__file__)] + rest


def self_launch(args, argv):
    """Start the ranks, relay their output (rank 0 prints the JSON line), exit non-zero if any rank fails.  Nothing in this process has
    touched the GPU (no HIP call, no torch.cuda.is_available()): the ranks are children, never an exec of this process."""
    import subprocess
    cmd = launcher_command(args, argv)
    if args.dry_launch:
        print(json.dumps({"launch": cmd, "n_gpus": args.gpus}), flush=True)
        return 0
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(args.master_port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    line = None
    for ln in p.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    if rc == 0 and line is None:
        sys.stderr.write("bench.py: the ranks exited without printing a result line\n")
        rc = 1
    return rc


def conv_roofline(B, T, F, iters=6):
    """Average launch duration of the dominant kernel by GPU time, the 40-channel row-streaming 3x3 convolution (conv3x3_rows16, two fp16
    terms per fp32 operand), on conv4's forward launch at the step's own shapes (40 -> 40 channels, BatchNorm + ReLU of the producer
    folded into the staging, batch statistics and per-channel output ranges written; the weight-packing pre-kernel is inside the
    average), HIP events on the launch stream.  Put against BOTH roofs: HBM -- algorithmic bytes = input + output tensor once --, and
    the matrix pipes -- algorithmic fp32 flops 2 * 9 * Cin * Cout per output element against the dense fp16 MFMA peak / 3 term products."""
    from piano_a2s_amd import hip
    L = hip.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    ci = co = 40
    x = torch.randn(B, T, ci, F, device=dev)
    y = torch.empty(B, T, co, F, device=dev)
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    scale, shift = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
    partial = torch.empty(L.a2s_conv3x3_stat_blocks(B, T, F, ci), co, 2, device=dev)
    cws = hip.conv_workspace(ci, dev)
    in_absmax = x.abs().amax(dim=(0, 1, 3)).contiguous()      # (in the step: written by the launch that produced x)
    out_absmax = torch.empty(co, device=dev)

    def launch():
        hip.conv3x3_forward(x, w, y, scale, shift, partial, cws, in_absmax, out_absmax)
    for _ in range(2):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    avg_s = e0.elapsed_time(e1) / iters / 1e3
    rows = L.a2s_debug_get(b"conv_rows")
    kernel = ("conv3x3_rows16<40, 40, affine> (conv4 forward launch, incl. its weight-packing pre-kernel)" if rows & 2 else
              "conv3x3_rows<40, 40, affine>" if rows & 1 else "conv3x3_split<40, false, 2> (tiled kernel of round 2)")
    flops = 2.0 * 9 * ci * co * B * T * F
    achieved = flops / avg_s / 1e12
    peak = MFMA_BF16_PEAK_TFS / 3
    mfma = {"achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "peak_note": "fp32-equivalent: 2.5 PFLOP/s dense fp16 MFMA / 3 term products per fp32 product (two exact fp16 terms per operand)",
            "frac_of_fp32_mfma_peak": round(achieved / MFMA_F32_PEAK_TFS, 4), "algorithmic_flops_per_launch": int(flops)}
    nbytes = 4.0 * B * T * F * (ci + co)
    gbs = nbytes / avg_s / 1e9
    hbm = {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(nbytes)}
    # measured HBM traffic of this very kernel: only from a committed rocprofv3 --pmc measurement that names the kernel and batch
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "conv_traffic.json")) as f:
            m = json.load(f)
        if m.get("per_gpu_batch") == B and m.get("kernel", "").split("<")[0] == kernel.split("<")[0]:
            traffic, traffic_src = m["traffic_bytes_per_launch"], m["source"]
    except Exception:  # noqa: BLE001
        pass
    first, other, bound = (hbm, mfma, "hbm") if hbm["frac"] >= mfma["frac"] else (mfma, hbm, "mfma")
    out = {"bound": bound, "bound_note": "the roof the launch is closer to; both fractions are given", "kernel": kernel}
    out.update(first)
    out.update({"traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": round(avg_s * 1e6, 1), ("mfma_view" if bound == "hbm" else "hbm_view"): other})
    return out


def step_roofline(B, T, F, H, clip_steps, ms_per_step, shared_steps=0.0):
    """Whole-step HBM roofline: ALGORITHMIC bytes of one optimizer step (derivation: DESIGN.md section 6) / measured step time / 8 TB/s.
      ConvStack, per clip, in units u = T x F x 4 bytes (one channel plane): forward = input (1) + every activation tensor written once and
        read once by its consumer (2 x 120 channels); backward = every activation read once more (120) + every activation gradient written
        once and read once (2 x 120) + the layer input of each weight gradient (80)  ->  681 u;
      encoder: features, the four input projections and the two layers' outputs, forward + backward  ~ 60 MB per clip;
      decoder: every (clip, decode step) pair that is still running streams the clip's key image and encoder outputs once in the forward
        and once in the backward pass: 2 x T x 3H x 4 bytes per pair (pairs counted by the step's own plan) -- minus the encoder outputs (2H of the 3H)
        of the pairs of the lower staff that one pass serves together with the upper staff's (round 6: `shared_steps`, also from the plan)."""
    u = T * F * 4.0
    conv = B * 681.0 * u
    enc = B * 60e6
    attn = 2.0 * (clip_steps * 3 - shared_steps * 2) * T * H * 4.0
    total = conv + enc + attn
    gbs = total / (ms_per_step * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
            "algorithmic_bytes_per_step": int(total), "parts_GB": {"convstack": round(conv / 1e9, 1), "encoder": round(enc / 1e9, 1), "decoder_attention": round(attn / 1e9, 1)},
            "attention_clip_steps_per_step": int(clip_steps), "attention_shared_enc_clip_steps_per_step": int(shared_steps), "ms_per_step": ms_per_step,
            "what": "algorithmic HBM bytes of one optimizer step / the timed step / HBM peak (the profile is flat: no single kernel is more than 5 % of the step)"}



def phase_roofline(step, batches, B, T, F, H, n_steps=6):
    """Every phase of the optimizer step against its roofs, FROM THIS RUN: HIP events on the step's main stream at the phase boundaries (the main
    stream joins every side stream where a phase ends), averaged over n_steps steps on the bench's own minibatches and coins.  Algorithmic
    work per phase: ConvStack in units u = T x F x 4 bytes per clip -- forward 241 u (input + every activation written
    once and read once), backward 440 u (every activation read once more, every activation gradient written and read once, the layer inputs
    of the weight gradients); convolution + Linear flops 41.07 GFLOP per clip forward, twice that backward, against the two-term matrix
    roof (2.5 PFLOP/s fp16 dense / 3 products); encoder 30 MB per clip each way, 4.72 + 0.94 GFLOP forward (recurrences + key images);
    decoder = the window from the first decoder launch to the start of the encoder backward (forward AND pipelined backward of all clip
    groups): 2 x (clip, step) pairs x T x 3H x 4 bytes."""
    from piano_a2s_amd import engine, engine_bwd
    marks = []

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append((name, e))
    saved = []

    def wrap(obj, attr, name):
        orig = getattr(obj, attr)
        saved.append((obj, attr, orig))

        def f(*args, **kw):
            mark("before " + name)
            r = orig(*args, **kw)
            mark(name)
            return r
        setattr(obj, attr, f)
    wrap(engine.Engine, "convstack", "convstack_fwd")
    wrap(engine.Engine, "encoder", "encoder_fwd")
    wrap(engine_bwd, "_encoder_bwd", "encoder_bwd")
    wrap(engine_bwd, "_convstack_bwd", "convstack_bwd")
    tot, clip_steps, shared_steps, walls = {}, [], [], []
    try:
        for k in range(n_steps + 1):
            marks.clear()
            mark("start")
            step(batches[k % len(batches)], TF_RATIO)
            mark("end")
            torch.cuda.synchronize()
            if k == 0:
                continue
            clip_steps.append(getattr(step, "attn_clip_steps", 0))
            shared_steps.append(getattr(step, "attn_shared_clip_steps", 0))
            walls.append(marks[0][1].elapsed_time(marks[-1][1]))
            for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
                name = {"before encoder_bwd": "decoder", "end": "clip_adadelta", "before convstack_fwd": "host_plan_gap"}.get(n1, n1)
                if name.startswith("before "):
                    name = "gaps"
                tot[name] = tot.get(name, 0.0) + e0.elapsed_time(e1)
    finally:
        for obj, attr, orig in saved:
            setattr(obj, attr, orig)
    ms = {k: v / n_steps for k, v in tot.items()}
    u = T * F * 4.0
    cs = sum(clip_steps) / max(len(clip_steps), 1)
    sh = sum(shared_steps) / max(len(shared_steps), 1)
    flop_fwd = B * 41.07e9
    work = {"convstack_fwd": (B * 241.0 * u, flop_fwd), "convstack_bwd": (B * 440.0 * u, 2 * flop_fwd),
            "encoder_fwd": (B * 30e6, B * 5.66e9), "encoder_bwd": (B * 30e6, B * 2 * 5.66e9), "decoder": (2.0 * (cs * 3 - sh * 2) * T * H * 4.0, None)}
    peak_tf = MFMA_BF16_PEAK_TFS / 3
    out = {"steps": n_steps, "ms_per_step": round(sum(walls) / len(walls), 2), "attention_clip_steps_per_step": int(cs),
           "peak_GBs": HBM_PEAK_GBS, "peak_TFLOPs_two_term": round(peak_tf, 1),
           "what": "per phase: measured ms (HIP events on the main stream, this run), algorithmic bytes and flops (DESIGN.md section 6), fractions of the "
                   "HBM roof and of the two-term matrix roof; the decoder phase holds the forward and the pipelined backward of all clip groups"}
    for name, (nbytes, flops) in work.items():
        t = ms.get(name)
        if not t:
            continue
        ph = {"ms": round(t, 2), "algorithmic_GB": round(nbytes / 1e9, 1), "GBs": round(nbytes / t / 1e6, 1), "frac_hbm": round(nbytes / t / 1e6 / HBM_PEAK_GBS, 4)}
        if flops:
            ph.update({"algorithmic_TFLOP": round(flops / 1e12, 2), "TFLOPs": round(flops / t / 1e9, 1), "frac_mfma": round(flops / t / 1e9 / peak_tf, 4)})
        out[name] = ph
    out["other_ms"] = {k: round(v, 2) for k, v in ms.items() if k not in work}
    return out

def linear_roofline(B, T, F, Cf=256, iters=5):
    """The three products of the 19200 -> 256 Linear (reference models.py:504,537-539) at the step's shapes, as the step issues them (two
    exact fp16 terms per fp32 operand, BatchNorm+ReLU of the operand folded into the staging, BatchNorm-backward statistics in the data
    gradient's epilogue): time, fraction of the two-term matrix roof (2.5 PF / 3 products) and of the HBM roof (operand tensor once)."""
    from piano_a2s_amd import hip
    L = hip.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    rows, K = B * T, 40 * F
    y4 = torch.randn(rows, K, device=dev)
    W = torch.randn(Cf, K, device=dev) * 0.007
    Wt = W.t().contiguous()
    dz = torch.randn(rows, Cf, device=dev) * 1e-5
    scale, shift = torch.rand(40, device=dev) + 0.5, torch.randn(40, device=dev) * 0.1
    mean, invstd = torch.zeros(40, device=dev), torch.ones(40, device=dev)
    wmax, dmax = hip.absmax(W), hip.absmax(dz)
    z = torch.empty(rows, Cf, device=dev)
    da = torch.empty(rows, K, device=dev)
    G = torch.zeros(Cf, K, device=dev)
    part = torch.empty((L.a2s_gemm_bnstats_blocks(rows, F), 40, 2), device=dev)
    sk = L.a2s_gemm_pick_splitk(Cf, K, rows, 1)

    def timed(fn):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    own = bool(L.a2s_linear_dgrad_eligible(rows, K, Cf, F)) and hip.LINEAR_KERNELS      # what engine_bwd runs
    if own:
        nb = L.a2s_linear_dgrad_ws_bytes(K, Cf)
        lws = torch.empty(nb // 4, dtype=torch.float32, device=dev)
        part2 = torch.empty((L.a2s_linear_dgrad_blocks(rows), 40, 2), device=dev)
        damax = torch.zeros(1, device=dev)

        def dgrad():
            hip.check(L.a2s_linear_dgrad_bnstats(hip.stream(), rows, K, Cf, hip._p(dz), C.c_long(Cf), hip._p(Wt), hip._p(da), C.c_long(K), hip._p(y4), hip._p(mean),
                                                 hip._p(invstd), hip._p(scale), hip._p(shift), F, hip._p(part2), hip._p(dmax), hip._p(wmax), hip._p(lws),
                                                 C.c_size_t(nb), hip._p(damax)), "linear_dgrad")
    else:
        def dgrad():
            hip.check(L.a2s_gemm_f32_bnstats_scaled(hip.stream(), rows, K, Cf, hip._p(dz), C.c_long(Cf), C.c_long(1), hip._p(Wt), C.c_long(1), C.c_long(Cf), hip._p(da),
                                                    C.c_long(K), hip._p(y4), hip._p(mean), hip._p(invstd), hip._p(scale), hip._p(shift), F, hip._p(part), hip._p(dmax),
                                                    hip._p(wmax)), "bnstats")
    bound = hip.act_bound(scale, shift, y4.view(rows, 40, F).abs().amax(dim=(0, 2)).contiguous())
    own_fwd = bool(L.a2s_linear_fwd_eligible(rows, Cf, K, F)) and hip.LINEAR_KERNELS
    ms = {"forward": timed(lambda: hip.linear_forward(y4, W, (scale, shift, F), bound, wmax, out=z)),
          "data_gradient": timed(dgrad),
          "weight_gradient": timed(lambda: hip.linear_wgrad(dz, y4, (scale, shift, F), dmax, bound, G)
                                   or hip.gemm(dz, 1, Cf, y4, K, 1, G, K, Cf, K, rows, beta=1.0, splitk=sk, b_affine=(scale, shift, F), two_term=(dmax, bound)))}
    flops = 2.0 * rows * K * Cf
    big = 4.0 * rows * K                                   # the (rows, 19200) operand / result: read (written) once
    byts = {"forward": big, "data_gradient": 2 * big, "weight_gradient": big}        # (the data gradient writes da and reads y4 for the statistics)
    own_wg = bool(L.a2s_linear_wgrad_eligible(rows, Cf, K, F)) and hip.LINEAR_KERNELS
    generic = "gemm_f32_kernel<256, 256, 4, 2, ..., 2> (two-term fp16 tiles)"
    out = {"kernel": "csrc/a2s_linear.hip where the shape qualifies, each incl. its operand-plane pre-pass: forward " + ("lin_fwd" if own_fwd else generic)
                     + "; data gradient " + ("lin_dgrad_bnstats" if own else generic) + "; weight gradient " + ("lin_wgrad (+ lin_wgrad_reduce)" if own_wg else generic),
           "peak_TFLOPs": round(MFMA_BF16_PEAK_TFS / 3, 1), "peak_GBs": HBM_PEAK_GBS}
    for k, t in ms.items():
        out[k] = {"ms": round(t, 2), "TFLOPs": round(flops / t / 1e9, 1), "frac_mfma": round(flops / t / 1e9 / (MFMA_BF16_PEAK_TFS / 3), 4),
                  "GBs": round(byts[k] / t / 1e6, 1), "frac_hbm": round(byts[k] / t / 1e6 / HBM_PEAK_GBS, 4)}
    return out


def vqt_block(cfg, dev, B=64, iters=3):
    """The VQT front end (reference utilities.py:240-254: librosa.vqt + amplitude_to_db, offline) as framed GEMMs on the GPU: clips/s and the
    algorithmic rates (waveform in, (1201, 480) features out; direct kernel-bank flops).  Parity with librosa itself is UNPINNED (DESIGN.md
    section 9); B = 64 here (the front end is not in the training step)."""
    from piano_a2s_amd import synthetic, vqt
    wave = synthetic.make_waveforms(B, 5, device=dev)
    front = vqt.VQT(dev)
    front(wave)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = front(wave)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = sum(2.0 * 2 * (o["hi"] - o["lo"]) * o["n_fft"] for o in front.octaves) * out.shape[2] * B        # complex bank: 2 real GEMMs per octave
    nbytes = wave.numel() * 4.0 + out.numel() * 4.0
    return {"clips_per_s": round(B / ms * 1e3, 1), "ms_per_batch": round(ms, 2), "batch": B, "frames": int(out.shape[2]), "bins": int(out.shape[3]),
            "algorithmic_TFLOPs": round(flops / ms / 1e9, 2), "frac_of_fp32_mfma_peak": round(flops / ms / 1e9 / MFMA_F32_PEAK_TFS, 4),
            "algorithmic_GBs": round(nbytes / ms / 1e6, 1), "frac_hbm": round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 5),
            "parity": "vs oracle/vqt_ref.py <= 0.012 dB (tests/test_gpu_vqt.py); vs librosa 0.10.1: UNPINNED (librosa / numba / soxr absent, no network)"}


def dp_settings_block(model, cfg, B, dev, full_tail, steps=3):
    """Collectives per optimizer step and step time under a ONE-rank process group (RCCL through the launcher-less path) for both BatchNorm
    settings: per-rank statistics (default, plain DDP semantics) and synchronised statistics (A2S_SYNC_BN=1: what SpeechBrain's
    SyncBatchNorm conversion gives the reference when it wraps the model in DDP)."""
    from piano_a2s_amd import synthetic, train
    created = False
    if not (dist.is_available() and dist.is_initialized()):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group(backend="nccl", rank=0, world_size=1)
        created = True
    out = {}
    try:
        b = synthetic.make_batch(B, cfg, 1234, full_tail=full_tail)
        b = [t.to(dev) if torch.is_tensor(t) else t for t in b]
        for name, sync in (("per_rank_batchnorm", False), ("synchronised_batchnorm", True)):
            st = train.TrainStep(model, dropout=True, sync_bn=sync)
            calls = {"n": 0}
            orig = dist.all_reduce

            def counting(*a, **k):
                calls["n"] += 1
                return orig(*a, **k)
            dist.all_reduce = counting
            try:
                rng = random.Random(7)
                st(b, TF_RATIO, rng=rng)
                torch.cuda.synchronize()
                calls["n"] = 0
                t0 = time.time()
                for _ in range(steps):
                    st(b, TF_RATIO, rng=rng)
                torch.cuda.synchronize()
                dt = (time.time() - t0) / steps
            finally:
                dist.all_reduce = orig
            out[name] = {"all_reduces_per_step": round(calls["n"] / steps, 1), "ms_per_step_one_rank": round(dt * 1e3, 1)}
            del st
        out["note"] = ("gradient exchange: 2 overlapped all-reduces of the flat gradient buffer (the loss gate rides in the first); synchronised BatchNorm adds "
                       "1 batch-shape check + 5 statistics exchanges forward + 5 backward, each a few hundred floats")
    finally:
        if created:
            dist.destroy_process_group()
    return out



def attention_roofline(step, batch_dev, B, T, H, iters=50):
    """Average launch duration of the dominant kernel (attn_step_fwd) at the step's own shapes, HIP events on the launch stream."""
    from piano_a2s_amd import hip
    L = hip.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    keys = torch.exp(2 * torch.randn(B, T, H, device=dev) * 0.5)      # the kernels take the key image exp(2K)
    enc = torch.randn(B, T, 2 * H, device=dev)
    q = torch.randn(B, H, device=dev) * 0.5
    v = torch.randn(H, device=dev) * 0.3
    ctx = torch.empty(B, 2 * H, device=dev)
    attw = torch.empty(B, T, device=dev)
    ws = hip.attn_workspace(B, T, H, dev)

    def launch():     # the launch the decoder loop issues every step: split kernel + combine kernel (both inside the timed average)
        hip.check(L.a2s_attn_step_fwd(hip.stream(), hip._p(keys), hip._p(enc), hip._p(q), C.c_long(H), hip._p(v), hip._p(ctx), C.c_long(2 * H),
                                      C.c_void_p(0), C.c_long(0), hip._p(attw), B, T, H, C.c_void_p(0), 0, hip._p(ws)), "attn_step_fwd")
    for _ in range(5):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    avg_s = e0.elapsed_time(e1) / iters / 1e3
    algo_bytes = B * (T * H + T * 2 * H) * 4.0          # keys + encoder outputs of every clip, read once per step (fp32)
    achieved = algo_bytes / avg_s / 1e9
    traffic, traffic_src = None, None                    # PMC counters cannot be collected from inside the bench: use the committed
    try:                                                 # rocprofv3 --pmc measurement when it was taken at this batch size
        with open(os.path.join(ROOT, "profiles", "attn_traffic.json")) as f:
            m = json.load(f)
        if m.get("per_gpu_batch") == B:
            traffic, traffic_src = m["traffic_bytes_per_launch"], m["source"]
    except Exception:  # noqa: BLE001
        pass
    return {"bound": "hbm", "kernel": "attn_fwd_split256 (+attn_fwd_combine256)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": round(avg_s * 1e6, 1),
            "algorithmic_bytes_per_launch": int(algo_bytes)}


# CPU baseline per BASELINE.md section 3: the oracle's as-written training step, B = 4, 1 warm-up + 2 timed steps, os.cpu_count()
# threads, in a child process under a hard timeout.  The oracle's per-step ops are small, so a host with hundreds of hardware threads
# can be SLOWER with all of them (round 1: 256 threads did not finish one step in 15 min); the attempts below fall back -- all threads
# -> 16 threads -> 16 threads on one clip -- and the line says which one ran and which ones timed out.
CPU_BASELINE_ATTEMPTS = ((4, 0, 75), (4, 16, 150), (1, 16, 60))        # (clips, threads (0 = os.cpu_count()), timeout s)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(full_tail):
    """Run _cpu_baseline_child in a subprocess with a hard timeout (a slow or oversubscribed host must not stall the bench)."""
    import subprocess
    notes = []
    for clips, threads, timeout in CPU_BASELINE_ATTEMPTS:
        threads = threads or (os.cpu_count() or 1)
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--cpu-clips", str(clips), "--cpu-threads", str(threads),
               "--full-tail", str(full_tail)]
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads))
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
            for line in reversed(r.stdout.strip().splitlines()):
                if line.startswith("{"):
                    out = json.loads(line)
                    if notes:                      # first, so that a truncated line still says it
                        out["sample"] = "FELL BACK after: " + "; ".join(notes) + " -- " + out["sample"]
                    return out
            notes.append(f"B={clips}/{threads} threads produced no result ({r.stderr[-120:]!r})")
        except subprocess.TimeoutExpired:
            notes.append(f"B={clips}/{threads} threads did not finish 1 warm-up + 2 timed steps within {timeout} s")
    return {"value": None, "unit": "clips/s", "cores": 0, "kind": "port", "sample": "; ".join(notes)}


def _cpu_baseline_child(cfg, n_clips, seed, threads, full_tail):
    """Training steps of the oracle (as-written CPU restatement of the reference path) on the host cores: 1 warm-up + 2 timed."""
    from oracle import model_ref, recipe_ref
    from piano_a2s_amd import spec, synthetic
    torch.set_num_threads(threads)
    st = spec.procedural_state(cfg, 1)
    P, Bf = spec.split_state(st)
    P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    state = {}
    rng = random.Random(1234)
    times = []
    for i in range(3):
        batch = synthetic.make_batch(n_clips, cfg, seed + i, full_tail=full_tail)
        gt = [batch[j] for j in range(1, 7)]
        t0 = time.time()
        outs = model_ref.forward(P, Bf, cfg, batch[0], inference=False, ground_truth=gt, teacher_forcing_ratio=TF_RATIO, training=True, rng=rng, dropout=True)
        losses = recipe_ref.objectives(outs, (batch[1], batch[2], batch[3], batch[5]))
        losses[0].backward()
        grads = {k: p.grad for k, p in P.items()}
        with torch.no_grad():
            recipe_ref.train_step({k: p.data for k, p in P.items()}, grads, state, float(losses[0]))
        for p in P.values():
            p.grad = None
        times.append(time.time() - t0)
    dt = sum(times[1:])
    return {"value": round(2 * n_clips / dt, 5), "unit": "clips/s", "cores": threads, "kind": "port", "cpu": _cpu_model(), "host_threads": os.cpu_count(),
            "sample": f"{threads} of {os.cpu_count()} host threads, {n_clips} clips per step: 2 timed training steps after 1 warm-up (fwd+loss+bwd+clip+Adadelta) of the "
                      f"oracle on synthetic 12 s clips, fp32, full-length tail {full_tail}: {times[1]:.1f} + {times[2]:.1f} s (warm-up {times[0]:.1f} s)"}


def loss_parity(dev):
    """Loss of the FULL-size model against the reference's own numbers, in this very run (BASELINE.json metric: "+ CPU-ref loss
    parity"): tests/golden/g2_full*.npz hold the four loss terms + total the reference computed on CPU for B = 2 clips with the
    procedural weights, train mode, dropout neutralised, tf = 1.0 and seeded tf = 0.7; the same batch goes through TrainStep here."""
    import numpy as np
    import models
    from piano_a2s_amd import spec, synthetic, train
    gd = os.path.join(ROOT, "tests", "golden")
    meta = json.load(open(os.path.join(gd, "g2_full.json")))
    meta_tf = json.load(open(os.path.join(gd, "g2_full_tf07.json")))
    ref = {"tf1.0": (np.load(os.path.join(gd, "g2_full.npz"))["train_tf1.losses"], 1.0, None),
           "tf0.7_seeded": (np.load(os.path.join(gd, "g2_full_tf07.npz"))["losses"], meta_tf["tf"], meta_tf["random_seed"])}
    cfg = spec.default_cfg()
    st = spec.procedural_state(cfg, meta["weights_seed"], eos_bias=meta["eos_bias"], lively=meta["lively"])
    kw = dict(meta["batch_kwargs"])
    kw["upper_range"], kw["lower_range"] = tuple(kw["upper_range"]), tuple(kw["lower_range"])
    batch = [t.to(dev) if torch.is_tensor(t) else t for t in synthetic.make_batch(2, cfg, meta["batch_seed"], **kw)]
    worst, detail = 0.0, {}
    for name, (want, tf, rseed) in ref.items():
        m = models.ScoreTranscription(**cfg)
        m.load_state_dict(st)
        m = m.to(dev).train()
        step = train.TrainStep(m, dropout=False)
        step(batch, tf, rng=random.Random(rseed))
        got = step.report()[:4]
        terms = [sum(got)] + got
        err = max(abs(g - float(w)) / abs(float(w)) for g, w in zip(terms, want))
        detail[name] = float(f"{err:.3e}")
        worst = max(worst, err)
    return {"rel_err": float(f"{worst:.3e}"), "bar": 1e-4, "ok": bool(worst <= 1e-4), "per_case": detail,
            "config": "full-size model (16.36M params), B=2, 1201 frames, train mode (batch-statistics BatchNorm), dropout off, through the fused "
                      "TrainStep; total + 4 loss terms vs the reference's CPU values (tests/golden/g2_full.npz, g2_full_tf07.npz)"}


def straggler_simulation(step, cfg, B, dev, full_tail, ranks=8, steps=20, steps_shared=3):
    """Data-parallel straggler term WITHOUT an 8-GPU box: every rank of an N-rank job meets the others at the gradient
    all-reduce, so a job step lasts as long as its slowest rank's.  A rank's step time is a function of its minibatch (target lengths) and
    of the teacher-forcing coins it draws (how bars fuse, reference models.py:289,404) -- not of the weights -- so the 8 ranks' step
    sequences are run here one after another on this GPU and  predicted_dp_efficiency = mean(step time) / mean_k(max_rank step_k time).
    Two coin policies: `rank_offset_coins` (random.seed(1234 + rank), round 2's default) and `shared_coins` (every rank seeds Python's
    random alike, data still differs per rank: the N-rank job then flips ONE coin per step for the whole global batch, which is what the
    reference does on one GPU with the N-fold batch).  Communication is not in this number (65.4 MB over xGMI, overlapped)."""
    from piano_a2s_amd import synthetic

    def run(rank, coin_seed, steps=steps):
        rng = random.Random(coin_seed)
        ms = []
        for k in range(steps + 1):
            b = synthetic.make_batch(B, cfg, 1234 + 1000 * rank + (k % N_MINIBATCHES), full_tail=full_tail)
            b = [t.to(dev) if torch.is_tensor(t) else t for t in b]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            step(b, TF_RATIO, rng=rng)
            e1.record()
            torch.cuda.synchronize()
            if k:                                      # first step of a sequence: new shapes / plans, untimed
                ms.append(e0.elapsed_time(e1))
        return ms

    def efficiency(table):
        k_ = len(table[0])
        mean = sum(sum(r) for r in table) / (len(table) * k_)
        worst = sum(max(r[k] for r in table) for k in range(k_)) / k_
        return mean / worst

    offset = [run(r, 1234 + r) for r in range(ranks)]
    shared = [offset[0][:steps_shared]] + [run(r, 1234, steps_shared) for r in range(1, ranks)]
    flat = [t for r in offset for t in r]
    slowest = [max(r[k] for r in offset) for k in range(steps)]
    out = {"ranks": ranks, "steps_per_rank": steps, "steps_per_rank_shared_coins": steps_shared,
           "step_ms_min_mean_max": [round(min(flat), 1), round(sum(flat) / len(flat), 1), round(max(flat), 1)],
           "slowest_rank_step_ms_min_mean_max": [round(min(slowest), 1), round(sum(slowest) / len(slowest), 1), round(max(slowest), 1)],
           "rank_offset_coins": {"predicted_dp_efficiency": round(efficiency(offset), 4), "step_ms_by_rank": [[round(t, 1) for t in r] for r in offset]},
           "shared_coins": {"predicted_dp_efficiency": round(efficiency(shared), 4), "step_ms_by_rank": [[round(t, 1) for t in r] for r in shared]},
           "what": "8 ranks' step sequences run one after another on this GPU; efficiency = mean step time / mean over steps of the slowest "
                   "rank's step time (straggler term only, no communication cost)"}
    return out


def inference_block(cfg, dev, batches=(256, 8)):
    """BASELINE.json configs[4]: greedy decode, eval mode, procedural weights with an <eos> bias (decoding ends at data-dependent
    steps), reference pretrain.py:131-136 / models.py:408.  clips/s and decoded tokens/s, kernel launches per executed decode step."""
    from piano_a2s_amd import engine, hip, spec, synthetic
    L = hip.lib()
    S = {k: v.to(dev) for k, v in spec.procedural_state(cfg, 2032, eos_bias=2.5, lively="token").items()}
    out = {"weights": "procedural (seed 2032, <eos> bias 2.5)",
           "mode": "eval, greedy; B > 8: 4-5 launches per decode step (attention sweep + combine, fused GRU step, fused projection / log-softmax / argmax / "
                   "embedding / next query), device-side <eos> bookkeeping polled every 16 steps; B <= 8: ONE persistent launch per (bar, staff) call "
                   "(csrc/a2s_dec_persist.hip), the end of the decode decided on the device; `hipgraph_replay`: the captured-chunk variant "
                   "BASELINE.json configs[4] names (A2S_GREEDY_GRAPH=1), measured beside it -- slower, hence opt-in"}
    for B, graph in [(b, False) for b in batches] + [(max(batches), True)]:
        x = synthetic.make_batch(B, cfg, 77, spectrogram="ridges", full_tail=0.0)[0].to(dev)
        eng = engine.Engine(cfg)
        eng.greedy_graph = graph
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            n0 = L.a2s_launch_count() if hasattr(L, "a2s_launch_count") else None
            t0 = time.time()
            with torch.no_grad():
                ts, key, up, lo = eng.forward(S, x, inference=True)
            torch.cuda.synchronize()
            dt = time.time() - t0
            n1 = L.a2s_launch_count() if hasattr(L, "a2s_launch_count") else None
            best = dt if best is None else min(best, dt)
        steps = sum(b["staff"][k][2]["steps"] for b in eng.saved["bars"] for k in ("up", "lo"))
        launched = sum(b["staff"][k][2]["launched"] for b in eng.saved["bars"] for k in ("up", "lo"))
        tokens = int((up.abs().sum(-1) > 0).sum() + (lo.abs().sum(-1) > 0).sum())
        out[f"B{B}" + ("_hipgraph_replay" if graph else "")] = {"seconds": round(best, 4), "clips_per_s": round(B / best, 2), "tokens_per_s": round(tokens / best),
                         "executed_decode_steps": steps, "launched_decode_steps": launched,
                         "kernel_launches_per_decode_step": (round((n1 - n0) / max(launched, 1), 2) if n0 is not None else None)}
        del eng, x
        torch.cuda.empty_cache()
    return out


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.cpu_baseline_child:                            # CPU-only helper process: never touches the GPU
        from piano_a2s_amd import spec
        print(json.dumps(_cpu_baseline_child(spec.default_cfg(), args.cpu_clips, 1234, args.cpu_threads or (os.cpu_count() or 1), args.full_tail)), flush=True)
        return
    if (args.gpus > 1 or args.launcher) and "RANK" not in os.environ:     # bare `python bench.py --gpus N`: start one rank per GPU as child processes
        sys.exit(self_launch(args, argv))
    if args.dry_launch:
        print(json.dumps({"launch": None, "n_gpus": args.gpus}), flush=True)
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or "RANK" in os.environ or os.environ.get("A2S_FORCE_DIST") == "1"     # under a launcher: the RCCL path even with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl")           # nccl == RCCL on ROCm
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs the MI355X: the transcription hot path has no CPU implementation"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.cuda.set_per_process_memory_fraction(0.96, local)     # an over-sized batch raises in torch instead of taking the box down
    torch.manual_seed(1234)
    random.seed(1234 + (rank if COIN_POLICY == "rank_offset" else 0))      # teacher-forcing coins (COIN_POLICY above)
    from piano_a2s_amd import build as a2s_build
    build_info = dict(a2s_build.info(), lib=os.path.relpath(a2s_build.LIB, ROOT), stale_vs_sources=bool(a2s_build.needs_build()),
                      note="last piano_a2s_amd.build.build() of the shipped .so; __graft_entry__.build() forces a from-scratch compile of every .hip source")

    import models
    from piano_a2s_amd import spec, synthetic, train
    cfg = spec.default_cfg()
    model = models.ScoreTranscription(**cfg).to(dev)
    model.train()
    if use_dist:                                           # identical replicas: broadcast rank 0's initial parameters
        train.broadcast_parameters(model.flatten_(), src=0)
    step = train.TrainStep(model, lr=1.0, rho=0.95, eps=1e-8, max_grad_norm=5.0, dropout=True)
    step.time_exchange = use_dist
    B = args.batch

    def make_batches(full_tail):
        out = []
        for i in range(min(N_MINIBATCHES, args.steps + args.warmup)):      # distinct minibatches (0.6 GB each), resident in HBM before timing
            b = synthetic.make_batch(B, cfg, 1234 + 1000 * rank + i, full_tail=full_tail)
            out.append([t.to(dev) if torch.is_tensor(t) else t for t in b])
        return out

    def timed(batches, warmup, steps):
        """W untimed + K timed steps, bracketed by barrier + synchronize on both sides; max over ranks; decode steps executed per step."""
        decode_steps = []
        timed.clip_steps = []
        timed.shared_steps = []
        for i in range(warmup):
            step(batches[i % len(batches)], TF_RATIO)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        step._exchange_events = []
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]      # per-step GPU time (diagnostic only: no extra sync)
        timed.launches0 = a2s_hip.lib().a2s_launch_count()
        t0 = time.time()
        marks[0].record()
        segs = [torch.cuda.memory_stats().get("segment.all.allocated", 0)]
        for i in range(steps):
            step(batches[i % len(batches)], TF_RATIO)
            decode_steps.append(step.decode_steps)
            timed.clip_steps.append(getattr(step, "attn_clip_steps", 0))
            timed.shared_steps.append(getattr(step, "attn_shared_clip_steps", 0))
            marks[i + 1].record()
            segs.append(torch.cuda.memory_stats().get("segment.all.allocated", 0))
        timed.new_segments = [b - a for a, b in zip(segs[:-1], segs[1:])]
        torch.cuda.synchronize()
        timed.step_ms = [round(a.elapsed_time(b), 1) for a, b in zip(marks[:-1], marks[1:])]
        if use_dist:
            dist.barrier()
        elapsed = time.time() - t0
        if use_dist:
            t = torch.tensor([elapsed], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t)
        return elapsed, decode_steps

    batches = make_batches(args.full_tail)
    from piano_a2s_amd import hip as a2s_hip
    # The caching allocator's pools, reserved before anything is timed (train.reserve_pools: one block per stream the step allocates on): a step's
    # tensor shapes follow its minibatch and its coins (how many bars fuse), so new block sizes keep turning up for ~10 steps, and each one a pool
    # cannot serve is a hipMalloc of several GiB -- 50-130 ms during which the step stands still (`hipMalloc_segments_per_step` in the JSON line).
    # A trainer that knows its memory budget reserves it up front; so does the bench (--no-reserve: off).
    pool_gib = 0.0

    def reserve_pool():
        nonlocal pool_gib
        if not args.no_reserve:
            pool_gib = round(train.reserve_pools(dev, B), 1)
    reserve_pool()
    elapsed, decode_steps = timed(batches, args.warmup, args.steps)
    launches_per_step = round((a2s_hip.lib().a2s_launch_count() - timed.launches0) / args.steps)
    clip_steps_per_step = sum(timed.clip_steps) / max(len(timed.clip_steps), 1)
    main_step_ms = list(timed.step_ms)
    main_new_segments = list(timed.new_segments)
    phases = phase_roofline(step, batches, B, 1201, cfg["freq_bins"], cfg["hidden_size"]) if (rank == 0 and world == 1 and not args.no_secondary) else None
    loss = float(step.total)
    groups = step._last[2] if step._last else None
    # data-parallel straggler terms: decode steps each rank executed per optimizer step, and how long each rank's stream
    # sat waiting for the gradient all-reduce (the wait of a fast rank IS the imbalance)
    dp = None
    if use_dist:
        waits = [e0.elapsed_time(e1) for e0, e1 in getattr(step, "_exchange_events", [])]
        mine = torch.tensor([sum(decode_steps) / max(len(decode_steps), 1), sum(waits) / max(len(waits), 1)], device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        dp = {"decode_steps_per_step_by_rank": [round(float(t[0]), 1) for t in allr],
              "allreduce_wait_ms_per_step_by_rank": [round(float(t[1]), 2) for t in allr]}
    secondary = None
    if not args.no_secondary and args.full_tail > 0:
        batches = None
        step._last = None
        torch.cuda.empty_cache()
        reserve_pool()
        e2, _ = timed(make_batches(0.0), 1, max(2, min(args.steps, 4)))
        k2 = max(2, min(args.steps, 4))
        secondary = {"value": round(B * world * k2 / e2, 3), "unit": "clips/s", "ms_per_step": round(e2 / k2 * 1e3, 2), "steps": k2, "warmup": 1,
                     "what": "the same step on minibatches WITHOUT the 1 % full-length tail (round 1's default workload)"}
    if rank == 0:
        clips = B * world * args.steps
        out = {"metric": "training clips/sec (12 s, 5-bar)", "value": round(clips / elapsed, 3), "unit": "clips/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "dtype_note": "fp32 data, fp32 accumulation and fp32 results; the large contractions multiply on the fp16 matrix pipes with every fp32 operand "
                             "EMULATED as two exact fp16 terms (22 of 24 significand bits, element error measured equal to the fp32-input MFMA kernels')",
               "data": "synthetic",
               "config": {"workload": "pretrain.yaml model (16.36M params), 12 s clips = 1201x480 frames, 5 bars, max 398/189 tokens; "
                                      "random-init weights; tf_ratio 0.7; dropout on; fwd+loss+bwd+clip+Adadelta",
                          "per_gpu_batch": B, "global_batch": B * world, "upper_len": "U{20..120}", "lower_len": "U{10..80}",
                          "full_length_tail": args.full_tail, "distinct_minibatches": min(N_MINIBATCHES, args.steps + args.warmup),
                          "parallelism": f"dp{world}", "batchnorm": "per-rank statistics",
                          "coin_policy": COIN_POLICY, "kernel_launches_per_step": launches_per_step,
                          "decoder": "rows whose remaining targets are all <pad> skipped; teacher-forced bars decoded in one call; clips "
                                     "holding full-length bars decoded as a concurrent clip group (loss, gradients and update identical "
                                     "to the per-bar loop over the whole minibatch)",
                          "clip_groups": groups, "decode_steps_per_step": round(sum(decode_steps) / max(len(decode_steps), 1), 1),
                          "step_ms": main_step_ms, "hipMalloc_segments_per_step": main_new_segments, "allocator_pool_reserved_GiB": pool_gib,
                          "allocator": {"hipMalloc_calls": torch.cuda.memory_stats().get("segment.all.allocated", 0),
                                        "alloc_retries": torch.cuda.memory_stats().get("num_alloc_retries", 0),
                                        "reserved_peak_GiB": round(torch.cuda.memory_stats().get("reserved_bytes.all.peak", 0) / 2 ** 30, 1)},
                          "arithmetic": "fp32 data and fp32 accumulation everywhere; the 3x3 convolutions (forward, data gradient: row-streaming "
                                        "kernels; conv3/conv4 weight gradient) and the 19200->256 Linear multiply on the fp16 matrix pipes with every "
                                        "fp32 operand carried as TWO exact fp16 terms under power-of-two scales derived from operand ranges the producing "
                                        "kernels write (three term products per fp32 product), the other 128x128 GEMM tiles on the bf16 pipes with three "
                                        "bf16 terms (six products); element error vs float64 equal to the fp32-input MFMA kernels' (DESIGN.md section 5); "
                                        "A2S_CONV_ROWS=0 selects round 2's tiled convolutions, A2S_ARITH=bf16x3 the three-term "
                                        "kernels, A2S_ARITH=f32 the fp32-input ones",
                          "final_loss": round(loss, 4), "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}}
        if secondary is not None:
            out["tail_off"] = secondary
        if dp is not None:
            out["data_parallel"] = dp
        step._last = None
        batches = None
        torch.cuda.empty_cache()
        out["roofline"] = conv_roofline(B, 1201, cfg["freq_bins"])
        out["roofline_step"] = step_roofline(B, 1201, cfg["freq_bins"], cfg["hidden_size"], clip_steps_per_step, out["ms_per_step"],
                                             sum(timed.shared_steps) / max(len(timed.shared_steps), 1))
        if phases is not None:
            out["roofline_phases"] = phases
        out["roofline_linear"] = linear_roofline(B, 1201, cfg["freq_bins"])
        out["roofline_attention"] = attention_roofline(step, None, B, 1201, cfg["hidden_size"])
        if not args.no_secondary and not use_dist:
            # (single-process runs only: with a process group every TrainStep call takes part in the gradient all-reduce, and rank 0 is
            # alone here -- the N = 1 line of the same commit carries the check)
            out["loss_parity"] = loss_parity(dev)
        if world == 1 and not args.no_inference:
            out["inference"] = inference_block(cfg, dev)
            out["inference"]["attention_roofline_frac"] = out["roofline_attention"]["frac"]
        if world == 1 and not args.no_inference:
            out["vqt"] = vqt_block(cfg, dev)
        if world == 1 and not use_dist and not args.no_straggler_sim:
            out["data_parallel_settings"] = dp_settings_block(model, cfg, B, dev, args.full_tail)
        if world == 1 and not args.no_straggler_sim:
            out["dp_straggler_simulation"] = straggler_simulation(step, cfg, B, dev, args.full_tail)
            out["predicted_dp_efficiency"] = out["dp_straggler_simulation"][COIN_POLICY + "_coins"]["predicted_dp_efficiency"]
        out["build"] = build_info
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.full_tail)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
