"""GPU: the sharded `Fast.run()` path end to end with two ranks on one device (gloo rendezvous;
the in-library RCCL communicator refuses two ranks on one GPU, so the exchange falls back to
torch.distributed -- the sharding, seeding and reassembly logic is what is under test)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("noseed,coherent", [("", ""), ("1", ""), ("", "1")])
def test_sharded_run_two_ranks_one_gpu(noseed, coherent):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", NOSEED=noseed, COHERENT=coherent)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29577", os.path.join(ROOT, "tests", "_dist_gpu_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "GPU DIST OK 2" in out.stdout
