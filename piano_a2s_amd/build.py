"""Build liba2s_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so sits next to its sources so it
travels with the repository snapshot to the GPU box."""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "liba2s_hip.so")
SOURCES = ["a2s_api.hip", "a2s_gemm.hip", "a2s_conv.hip", "a2s_seq.hip", "a2s_bwd.hip", "a2s_opt.hip", "a2s_vqt.hip"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "a2s.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"] + SOURCES + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
