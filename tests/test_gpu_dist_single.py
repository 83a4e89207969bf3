"""The RCCL code path of the multi-GPU runs on ONE GPU: a single rank launched through torch.distributed.run with
R3DET_FORCE_DIST=1 builds the nccl process group, gathers detections, wraps the detector in DDP and takes one
training step (tests/dist_single_worker.py).  What world size 2 adds on top -- the actual exchange -- is covered
with gloo on the CPU (tests/test_dist_cpu.py, tests/test_dist_train_cpu.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_one_rank_over_rccl():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, R3DET_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "dist_single_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["backend"] == "nccl" and rec["world"] == 1 and rec["loss"] > 0
