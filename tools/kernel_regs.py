"""Registers / spills / LDS of the kernels in a compiled object of the product build (one line per kernel, fields matched by name).
usage: python tools/kernel_regs.py a2s_step [name-filter]"""
import re
import subprocess
import sys

B = "/opt/rocm/lib/llvm/bin"
name = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
obj = f"/root/repo/piano_a2s_amd/csrc/_obj/{name}.o"
subprocess.run([f"{B}/llvm-objcopy", f"--dump-section=.hip_fatbin=/tmp/{name}.fatbin", obj], check=True)
t = [l for l in subprocess.run([f"{B}/clang-offload-bundler", "--list", "--type=o", f"--input=/tmp/{name}.fatbin"], capture_output=True, text=True).stdout.split() if "gfx950" in l][0]
subprocess.run([f"{B}/clang-offload-bundler", "--type=o", f"--targets={t}", f"--input=/tmp/{name}.fatbin", f"--output=/tmp/{name}.co", "--unbundle"], check=True)
notes = subprocess.run([f"{B}/llvm-readelf", "--notes", f"/tmp/{name}.co"], capture_output=True, text=True).stdout
for blk in re.split(r"\n\s+- \.", notes):
    m = re.search(r"\.name:\s+(\S+)", blk)
    if not m or ".vgpr_count" not in blk:
        continue
    g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [0, "?"])[1]
    dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt in dem:
        print(f"{dem[:70]:70s} vgpr {g('vgpr_count'):>3s} agpr {g('agpr_count'):>3s} spill {g('vgpr_spill_count'):>3s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s}")
