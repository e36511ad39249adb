"""Build liba2s_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so sits next to its sources so it
travels with the repository snapshot to the GPU box.  Every .hip file is compiled to an object of its own (in parallel, only when it or
a header changed), then linked."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "liba2s_hip.so")
OBJ = os.path.join(CSRC, "_obj")
SOURCES = ["a2s_api.hip", "a2s_gemm.hip", "a2s_conv.hip", "a2s_conv_rows.hip", "a2s_conv_wrows.hip", "a2s_seq.hip", "a2s_bwd.hip", "a2s_opt.hip", "a2s_vqt.hip", "a2s_step.hip", "a2s_persist.hip", "a2s_dec_persist.hip", "a2s_linear.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _headers():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "a2s.h"))
    return deps


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    srcs = [os.path.join(CSRC, f) for f in SOURCES if os.path.exists(os.path.join(CSRC, f))]
    return _stale(LIB, srcs + _headers())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    headers = _headers()
    jobs, objs = [], []
    for f in SOURCES:
        src = os.path.join(CSRC, f)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJ, f[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc] + FLAGS + ["-c", f, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, cwd=CSRC, check=True)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as pool:
        list(pool.map(run, jobs))
    # link next to the target and rename: a process that has the old library mapped keeps its (unlinked) file instead of seeing it rewritten
    tmp = LIB + f".tmp{os.getpid()}"
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp])
    os.replace(tmp, LIB)
    import json
    import time
    with open(os.path.join(OBJ, "build_info.json"), "w") as f:
        json.dump({"forced": bool(force), "sources_compiled": len(jobs), "sources_total": len(objs), "unix_time": int(time.time())}, f)
    return LIB


def build_variant(name, extra_flags, sources=None):
    """A measurement build next to the product library: csrc/_obj/liba2s_hip_<name>.so with extra compiler flags (e.g. -DRW_TRACE); load it with
    A2S_LIB=<path>.  Only `sources` (default: all) are recompiled with the flags, the rest reuse the product objects."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    build()
    vdir = os.path.join(OBJ, name)
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for f in SOURCES:
        if sources is None or f in sources:
            obj = os.path.join(vdir, f[:-4] + ".o")
            subprocess.run([hipcc] + FLAGS + list(extra_flags) + ["-c", f, "-o", obj], cwd=CSRC, check=True)
        else:
            obj = os.path.join(OBJ, f[:-4] + ".o")
        objs.append(obj)
    out = os.path.join(OBJ, f"liba2s_hip_{name}.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out], check=True)
    return out


def info():
    """What the last build() of the shipped .so did (forced from-scratch compile or incremental)."""
    import json
    try:
        with open(os.path.join(OBJ, "build_info.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {"forced": None}


if __name__ == "__main__":
    print(build(force=True, verbose=True))
