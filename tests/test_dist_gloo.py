"""CPU: the N > 1 path (sharding + gather + histogram reduce) with world_size 2 over gloo."""
import os
import subprocess
import sys

from conftest import ROOT


def test_two_rank_sharded_run_equals_single_process():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29511", os.path.join(ROOT, "tests", "_dist_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "DIST OK 2" in out.stdout
