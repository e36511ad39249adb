"""Two data-parallel ranks on the one GPU of the test box (gloo backend, both processes on cuda:0): the fused training step with
synchronised BatchNorm and the overlapped gradient exchange must leave both ranks with identical parameters, equal to one process
stepping on the concatenated minibatch (tools/dp_check.py, run as a child process so that its ranks start from fresh interpreters)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(330)
def test_two_ranks_equal_one_process_on_the_whole_minibatch():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.empty_cache()          # the ranks are separate processes on the same GPU: hand the parent's cached blocks back first
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_check.py")], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ranks identical: max |p0 - p1| = 0.000e+00" in r.stdout, r.stdout[-1000:]


@pytest.mark.timeout(330)
def test_two_ranks_with_per_rank_batchnorm_equal_averaged_gradients():
    """The DEFAULT data-parallel setting (per-rank BatchNorm statistics) through the whole fused step on two ranks: identical parameters on both
    ranks, equal to one process averaging the two halves' gradients before the fused clip + Adadelta (DDP's all-reduce-mean)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.empty_cache()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_check.py"), "--local-bn"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ranks identical: max |p0 - p1| = 0.000e+00" in r.stdout and "per-rank BatchNorm" in r.stdout, r.stdout[-1000:]


def test_bench_launches_its_own_rank_through_torchrun():
    """`python bench.py --launcher` (the N = 1 form of what `--gpus N` does without a launcher environment): torch.distributed.run child,
    RCCL world of one, rank 0's JSON line relayed."""
    import json
    import os
    import subprocess
    import sys
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--launcher", "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "8", "--master-port", "29641",
                        "--no-cpu-baseline", "--no-secondary", "--no-inference", "--no-straggler-sim"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0 and "data_parallel" in out
