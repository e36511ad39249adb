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
