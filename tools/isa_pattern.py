"""Instruction pattern of the MFMA-carrying basic blocks of one kernel in a hipcc -save-temps .s file: one letter per instruction
(M mfma, v VALU, r/w LDS read/write, G global/buffer, | s_waitcnt, B barrier, s SALU).  usage: isa_pattern.py file.s kernel-substring"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r"^(\S*%s\S*):" % re.escape(key), s, re.M)
i = m.start()
k = s[i:s.index(".Lfunc_end", i)]
blocks, cur = [], None
for ln in k.split("\n"):
    b = re.match(r"^(\.LBB\d+_\d+):", ln)
    if b:
        cur = [b.group(1), []]
        blocks.append(cur)
    elif cur is not None and ln.strip() and not ln.strip().startswith((".", ";")):
        cur[1].append(ln.strip())
for name, ins in blocks:
    if sum("mfma" in x for x in ins) < 8:
        continue
    out, c = "", Counter()
    for x in ins:
        op = x.split()[0]
        ch = ("M" if "mfma" in op else "r" if op.startswith("ds_read") else "w" if op.startswith("ds_write") else "G" if op.startswith(("buffer", "global")) else
              "|" if op.startswith("s_waitcnt") else "B" if op.startswith("s_barrier") else "a" if op.startswith("v_accvgpr") else "v" if op.startswith("v_") else "s")
        out += ch
        c[ch] += 1
    print(name, len(ins), dict(c))
    for q in range(0, len(out), 160):
        print("  " + out[q:q + 160])
