"""CPU-side checks of the C-ABI boundary: the library builds, loads, exports every symbol include/a2s.h declares
(no compute is launched without a GPU), the ctypes mirror of the argument block matches the C layout, and the
product path refuses to run without device memory instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from piano_a2s_amd import build
    return build.build()


def test_exports_every_declared_symbol(libpath):
    header = open(os.path.join(ROOT, "include", "a2s.h")).read()
    declared = sorted(set(re.findall(r"\b(a2s_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 18
    lib = C.CDLL(libpath)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"declared in include/a2s.h but not exported with C linkage: {missing}"
    lib.a2s_last_error.restype = C.c_char_p
    assert lib.a2s_version() >= 1 and isinstance(lib.a2s_last_error(), bytes)


@pytest.mark.parametrize("cname,pyname", [("a2s_note_dec_args", "NoteDecArgs"), ("a2s_note_dec_bwd_args", "NoteDecBwdArgs")])
def test_arg_block_layout_matches_c(libpath, tmp_path, cname, pyname):
    """sizeof / offsetof of the argument blocks as the C compiler sees them == the ctypes mirrors."""
    from piano_a2s_amd import hip
    cls = getattr(hip, pyname)
    fields = [f[0] for f in cls._fields_]
    src = tmp_path / "layout.c"
    body = "\n".join(f'printf("{f} %zu\\n", offsetof({cname}, {f}));' for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "a2s.h"\nint main(){printf("sizeof %zu\\n", sizeof(' + cname + '));\n' + body + "\nreturn 0;}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    assert int(out["sizeof"]) == C.sizeof(cls)
    for f in fields:
        assert int(out[f]) == getattr(cls, f).offset, f


def test_no_cpu_fallback():
    from piano_a2s_amd import engine, hip, spec
    cfg = spec.default_cfg(freq_bins=24, conv_feature_size=32, hidden_size=32, max_length=(12, 8))
    S = spec.procedural_state(cfg, 1)
    with pytest.raises(hip.A2SError):
        engine.Engine(cfg).forward(S, torch.zeros(1, 1, 9, 24), inference=True)


def _literal_loop(gt, max_steps, eos):
    """The reference's host loop, step by step (models.py:386-419 with ground truth): independent of plan_note_steps."""
    B = gt.shape[0]
    seen = [0] * B
    lengths = [max_steps] * B
    steps = 0
    for t in range(max_steps):
        if sum(seen) == B:
            break
        steps += 1
        for b in range(B):
            if int(gt[b, t]) == eos:
                seen[b] = 1
                lengths[b] = t + 1
    return steps, lengths


def test_training_plan_matches_reference_bookkeeping():
    """plan_note_steps (vectorised, used by the engine) == the reference's literal per-step loop."""
    from piano_a2s_amd.engine import plan_note_steps
    from piano_a2s_amd.spec import EOS, PAD
    g = torch.Generator().manual_seed(0)
    for trial in range(200):
        B, M = int(torch.randint(1, 6, (1,), generator=g)), int(torch.randint(1, 14, (1,), generator=g))
        rnd = torch.randint(0, 150, (B, M), generator=g)
        rnd[torch.rand(B, M, generator=g) < 0.2] = EOS          # arbitrary (also repeated / missing) <eos> positions
        steps, lengths = plan_note_steps(rnd, M)
        assert (steps, lengths.tolist()) == _literal_loop(rnd, M, EOS), (trial, rnd.tolist())
    gt = torch.full((4, 10), PAD, dtype=torch.long)
    gt[0, :3] = 5; gt[0, 3] = EOS
    gt[1, :6] = 7; gt[1, 6] = EOS
    gt[2, 0] = EOS; gt[2, 2] = EOS          # <eos> twice before the break: length is overwritten
    gt[3, :2] = 9; gt[3, 2] = EOS
    steps, lengths = plan_note_steps(gt, 10)
    assert steps == 7 and lengths.tolist() == [4, 7, 3, 3]
    gt[1] = 7                                 # a full row without <eos>: never breaks
    steps, lengths = plan_note_steps(gt, 10)
    assert steps == 10 and lengths.tolist() == [4, 10, 3, 3]
