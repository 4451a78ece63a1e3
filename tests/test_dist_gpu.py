"""GPU: the template-sharded forward with the real HIP model on two ranks (one GPU, gloo) equals the single-process
forward bit for bit — top-k ids, stage-2 poses, key-point lists (the RCCL run itself needs a multi-GPU node).
The ranks take turns on the shared card for their local compute, the collectives run between them (dist_worker_gpu.py:
two busy processes on ONE MI355X corrupted SGPR lane masks of the warp kernel until it was rewritten)."""
import os
import socket
import subprocess
import sys

import pytest

gpu = pytest.mark.gpu


@gpu
def test_sharded_forward_two_ranks_one_gpu_equals_single_process():
    here = os.path.dirname(os.path.abspath(__file__))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(here, "dist_worker_gpu.py")],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert "RANK0 OK" in out and "RANK1 OK" in out, out[-3000:]
