"""Per-kernel averages of the counters in a rocprofv3 `--pmc ... --output-format csv` run (…_counter_collection.csv).
usage: python tools/pmc_summary.py <dir> [kernel-name regex]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    files = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*$", "", r["Kernel_Name"]).strip().replace("void ", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    for name, ctrs in sorted(acc.items()):
        if pat and not pat.search(name):
            continue
        print(name)
        for c, v in sorted(ctrs.items()):
            print(f"    {c:32s} n={len(v):4d}  mean {sum(v) / len(v):16.2f}")


if __name__ == "__main__":
    main()
