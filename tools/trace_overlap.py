"""From a rocprofv3 --kernel-trace csv: how the long-clip group's per-step chain (dec_gru_step / dec_out_step / dec_bwd_* launches,
which only that group issues) sits in time relative to the bulk group's kernels on the other queues.
usage: python tools/trace_overlap.py <dir-with-*_kernel_trace.csv>"""
import csv, glob, os, sys
import numpy as np

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
t0 = min(int(r["Start_Timestamp"]) for r in rows)
K = [(name(r), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
K.sort(key=lambda k: k[1])
queues = {}
for k in K:
    queues.setdefault(k[3], []).append(k)
print("queues:", {q: len(v) for q, v in queues.items()})
chain = [k for k in K if k[0].startswith(("dec_gru_step", "dec_out_step", "dec_bwd_products", "dec_bwd_query"))]
if not chain:
    sys.exit("no fused-step kernels in the trace")
q1 = max(set(k[3] for k in chain), key=lambda q: sum(1 for k in chain if k[3] == q))
print("long-clip group's queue:", q1, " kernels on it:", len(queues[q1]))
g1 = queues[q1]
# per-step period of the chain: time between consecutive dec_gru_step launches
gs = [k for k in g1 if k[0].startswith("dec_gru_step")]
per = np.diff([k[1] for k in gs])
per = per[per < 1000]
print(f"forward step period on that queue: median {np.median(per):.1f} us, p10 {np.quantile(per, .1):.1f}, p90 {np.quantile(per, .9):.1f}  ({len(per)} steps)")
bs = [k for k in g1 if k[0].startswith("dec_bwd_products")]
per = np.diff([k[1] for k in bs])
per = per[per < 1000]
if len(per):
    print(f"backward step period: median {np.median(per):.1f} us, p10 {np.quantile(per, .1):.1f}, p90 {np.quantile(per, .9):.1f}  ({len(per)} steps)")
# kernel durations on that queue, and the gaps between them
byname = {}
for a, b in zip(g1[:-1], g1[1:]):
    byname.setdefault(a[0], []).append((a[2] - a[1], b[1] - a[2]))
print("kernel                          n     dur_us(median)   gap_after_us(median)")
for n, v in sorted(byname.items(), key=lambda kv: -len(kv[1]))[:14]:
    d = np.array(v)
    print(f"{n[:30]:30s} {len(v):6d} {np.median(d[:, 0]):10.1f} {np.median(d[:, 1]):14.1f}")
# how busy are the other queues while the chain runs?
lo, hi = g1[0][1], g1[-1][2]
for q, v in queues.items():
    if q == q1:
        continue
    busy = sum(min(k[2], hi) - max(k[1], lo) for k in v if k[2] > lo and k[1] < hi)
    print(f"queue {q}: busy {100 * busy / (hi - lo):.0f} % of the chain's span ({(hi - lo) / 1e3:.0f} ms)")

# ---- every queue: busy share of the traced span, idle time by gap size, top kernels by time
span_lo, span_hi = K[0][1], max(k[2] for k in K)
print(f"\ntraced span {(span_hi - span_lo) / 1e3:.0f} ms")
for q, v in sorted(queues.items()):
    v = sorted(v, key=lambda k: k[1])
    busy = sum(k[2] - k[1] for k in v)
    gaps = np.array([b[1] - a[2] for a, b in zip(v[:-1], v[1:])])
    gaps = gaps[gaps > 0]
    small, mid, big = gaps[gaps < 20].sum(), gaps[(gaps >= 20) & (gaps < 1000)].sum(), gaps[gaps >= 1000].sum()
    print(f"queue {q}: {len(v)} kernels, busy {busy / 1e3:.0f} ms ({100 * busy / (span_hi - span_lo):.0f} %), idle in gaps < 20 us {small / 1e3:.0f} ms, 20 us - 1 ms {mid / 1e3:.0f} ms, > 1 ms {big / 1e3:.0f} ms")
    tot = {}
    for k in v:
        tot[k[0]] = tot.get(k[0], 0.0) + k[2] - k[1]
    print("    " + ", ".join(f"{n[:28]} {t / 1e3:.0f}" for n, t in sorted(tot.items(), key=lambda kv: -kv[1])[:8]))

# ---- a window of the long-clip queue in the middle of its backward chain and of its forward chain (what runs between two steps?)
for label, seq in (("backward", bs), ("forward", gs)):
    if len(seq) < 200:
        continue
    mid = seq[len(seq) // 2]
    i0 = g1.index(mid)
    print(f"\n{label} chain, 36 consecutive kernels on queue {q1} (start us relative to the first, duration, stream, kernel):")
    for k in g1[i0:i0 + 36]:
        print(f"  {k[1] - mid[1]:9.1f} {k[2] - k[1]:7.1f}  s{k[4]}  {k[0][:60]}")

# ---- the bulk group's chains (note_step_finalize is issued by the library-path decode step only): a window of each queue carrying them
fin = [k for k in K if k[0].startswith("note_step_finalize")]
for q in sorted(set(k[3] for k in fin)):
    v = queues[q]
    fq = [k for k in fin if k[3] == q]
    if len(fq) < 100:
        continue
    mid = fq[len(fq) // 3]
    i0 = v.index(mid)
    print(f"\nbulk group, forward, queue {q}: 30 consecutive kernels (start us relative to the first, duration, stream, kernel):")
    for k in v[i0:i0 + 30]:
        print(f"  {k[1] - mid[1]:9.1f} {k[2] - k[1]:7.1f}  s{k[4]}  {k[0][:70]}")
gb = [k for k in K if k[0].startswith("gru_gates_bwd") and k[3] != q1]
for q in sorted(set(k[3] for k in gb)):
    v = queues[q]
    fq = [k for k in gb if k[3] == q]
    if len(fq) < 100:
        continue
    mid = fq[len(fq) // 3]
    i0 = v.index(mid)
    print(f"\nbulk group, backward, queue {q}: 30 consecutive kernels:")
    for k in v[i0:i0 + 30]:
        print(f"  {k[1] - mid[1]:9.1f} {k[2] - k[1]:7.1f}  s{k[4]}  {k[0][:70]}")

# ---- the bar-level chain: what the main stream runs between the end of the encoder and the first note-decoder step of a training step
enc = [i for i, k in enumerate(K) if k[0].startswith("gru_step_fwd_fused")]
if enc:
    # last encoder step kernel of the LAST training step in the trace
    last = enc[-1]
    stream0 = K[last][4]
    print(f"\nafter the encoder of the last traced step: the next 70 kernels of every stream (start us relative to the encoder's last step, duration, queue, stream, kernel):")
    t_ref = K[last][2]
    n = 0
    for k in K[last + 1:]:
        print(f"  {k[1] - t_ref:9.1f} {k[2] - k[1]:7.1f}  q{k[3]} s{k[4]}  {k[0][:70]}")
        n += 1
        if n >= 70:
            break

# ---- (round 5) where the long-clip chain's queue idles: gaps between consecutive kernels of that queue, by the kernel in front of the gap, and what the
# other queues run during the long ones
g1s = sorted(g1, key=lambda k: k[1])
gap_by = {}
long_gaps = []
for a, b in zip(g1s[:-1], g1s[1:]):
    gap = b[1] - a[2]
    if gap <= 0 or gap > 2000:
        continue
    gap_by.setdefault((a[0][:28], b[0][:28]), []).append(gap)
    if gap > 30:
        long_gaps.append((a[2], b[1]))
print("\ngaps on the long-clip group's queue (kernel in front -> kernel behind): n, mean us, p90 us, total ms")
for (ka, kb), v in sorted(gap_by.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v = np.array(v)
    print(f"  {ka:28s} -> {kb:28s} {len(v):6d} {v.mean():8.1f} {np.quantile(v, .9):8.1f} {v.sum() / 1e3:8.1f}")
tot_long = sum(b - a for a, b in long_gaps)
print(f"gaps > 30 us: {len(long_gaps)}, {tot_long / 1e3:.1f} ms in all; what the other queues run meanwhile (share of that time, per queue):")
others = [k for k in K if k[3] != q1]
starts = np.array([k[1] for k in others])
ends = np.array([k[2] for k in others])
cover = {}
for a, b in long_gaps[:20000]:
    lo_i = np.searchsorted(starts, a - 3000)
    hi_i = np.searchsorted(starts, b)
    for i in range(lo_i, hi_i):
        o = min(ends[i], b) - max(starts[i], a)
        if o > 0:
            key = (others[i][3], others[i][0][:34])
            cover[key] = cover.get(key, 0.0) + o
for (q, n), t in sorted(cover.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  queue {q}  {n:34s} {100 * t / max(tot_long, 1):5.1f} %")
