"""N > 1 with REAL processes on the one GPU a test box has: `torch.distributed.run` starts W ranks of
tests/two_rank_gpu_check.py (all on cuda:0, process group gloo -- RCCL refuses two ranks per device), which
check the sharded layers / stack / training step against the unsharded ones.  See that file's docstring."""
import os
import re
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_paths_over_real_process_group(world):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "two_rank_gpu_check.py")]
    env = dict(os.environ, OMP_NUM_THREADS="4")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    tail = (r.stdout + "\n" + r.stderr)[-6000:]
    assert r.returncode == 0, tail
    # 8 layer cases + 1 stack + 2 training cases + 2 graph-boundary cases per rank (ranks may interleave their lines)
    assert len(re.findall(r"rank \d+ ok ", r.stdout)) == 13 * world, tail
