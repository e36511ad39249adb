"""From a rocprofv3 --kernel-trace csv of tools/step_trace.py: the timeline of every hardware queue over the LAST traced optimizer step, compressed
into runs (consecutive kernels of one class: the per-step decoder chain forward / backward, attention sweeps, GEMMs, convolutions, ...):
start, end, kernels, busy time.  Shows what the long-clip group's queue does between its chains (the deferred products) and what it waits for.
usage: python tools/queue_timeline.py <dir-with-*_kernel_trace.csv> [min_run_ms]"""
import csv, glob, os, sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = list(csv.DictReader(open(f)))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
K = sorted(((name(r), int(r["Start_Timestamp"]) / 1e3, int(r["End_Timestamp"]) / 1e3, r.get("Queue_Id", "?")) for r in rows), key=lambda k: k[1])
# the last step starts at the last first-layer convolution forward launch (conv3x3_c1 runs once per step, first thing)
starts = [k[1] for k in K if k[0].startswith("conv3x3_c1")]
t0 = starts[-1] if starts else K[0][1]
K = [(n, a - t0, b - t0, q) for n, a, b, q in K if a >= t0]


def cls(n):
    if n.startswith(("dec_gru_step", "dec_out_step", "attn_fwd", "note_step", "embed_rows", "gru_gates_fwd")): return "decoder fwd step chain"
    if n.startswith(("dec_bwd", "attn_bwd", "gru_gates_bwd")): return "decoder bwd step chain"
    if n.startswith("gemm_f32_kernel") or n.startswith("gemm_"): return "gemm"
    if n.startswith(("conv3x3", "bn_", "lin_")): return "convstack/linear"
    if n.startswith("gru_seq"): return "encoder recurrence"
    return n[:28]


by = {}
for k in K:
    by.setdefault(k[3], []).append(k)
for q, v in sorted(by.items()):
    print(f"=== queue {q}: {len(v)} kernels, busy {sum(b - a for _, a, b, _ in v) / 1e3:.1f} ms, last ends at {max(b for _, a, b, _ in v) / 1e3:.1f} ms")
    runs = []
    for n, a, b, _ in v:
        c = cls(n)
        # gemms / small kernels inside a step chain belong to the chain: merge a short foreign kernel into the current run
        if runs and (runs[-1][0] == c or (a - runs[-1][2] < 200 and (c == "gemm" or len(c) <= 28) and runs[-1][0].startswith("decoder") and b - a < 200)):
            r = runs[-1]
            r[2] = b; r[3] += 1; r[4] += b - a
        else:
            runs.append([c, a, b, 1, b - a])
    for c, a, b, n, busy in runs:
        if (b - a) / 1e3 >= min_ms:
            print(f"   {a / 1e3:8.1f} -> {b / 1e3:8.1f} ms  ({(b - a) / 1e3:7.1f} ms, {n:6d} kernels, busy {busy / 1e3:7.1f} ms)  {c}")
