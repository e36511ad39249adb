"""Condense rocprofv3 `--kernel-trace --stats --output-format csv` output (…_kernel_stats.csv) into the table kept under profiles/.
usage: python tools/kernel_stats.py <dir-or-csv> [top_n]"""
import csv
import glob
import os
import re
import sys


def main():
    src = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
    rows = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*$", "", r["Name"]).strip()
            name = re.sub(r"^void ", "", name)
            c, tot = int(r["Calls"]), float(r["TotalDurationNs"])
            mn, mx = float(r["MinNs"]), float(r["MaxNs"])
            a = rows.setdefault(name, [0, 0.0, mn, mx])
            a[0] += c
            a[1] += tot
            a[2], a[3] = min(a[2], mn), max(a[3], mx)
    total = sum(a[1] for a in rows.values())
    print(f"# total kernel time {total / 1e6:.1f} ms over {sum(a[0] for a in rows.values())} launches")
    print("# pct   total_ms   calls   avg_us   min_us   max_us   kernel")
    for name, a in sorted(rows.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{100 * a[1] / total:5.1f} {a[1] / 1e6:10.2f} {a[0]:7d} {a[1] / a[0] / 1e3:9.1f} {a[2] / 1e3:8.1f} {a[3] / 1e3:8.1f}  {name}")


if __name__ == "__main__":
    main()
