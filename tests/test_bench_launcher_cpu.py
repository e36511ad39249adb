"""bench.py --gpus N without a launcher environment starts its own ranks (reference README.md:119-122: `torchrun --nproc_per_node=N`).
CPU box: the launcher command it would start (--dry-launch), nothing touches a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_dry_launch_builds_the_torchrun_command():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = _run("--gpus", "4", "--steps", "7", "--warmup", "2", "--dry-launch", env=env)
    cmd = out["launch"]
    assert out["n_gpus"] == 4 and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "7", "--warmup", "2"], tail          # the ranks get the same flags, minus the launcher's own


def test_single_gpu_and_ranks_do_not_relaunch():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    assert _run("--dry-launch", env=env)["launch"] is None                         # --gpus 1: runs in this process
    assert _run("--gpus", "2", "--dry-launch", env=dict(env, RANK="1", WORLD_SIZE="2", LOCAL_RANK="1"))["launch"] is None      # already a rank
    assert _run("--launcher", "--dry-launch", env=env)["launch"] is not None       # forced: one rank through torch.distributed.run
