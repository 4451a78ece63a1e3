"""world_size-2 gloo test of the template-sharded stage 1 (CPU; the oracle stands in for the
HIP scorer, which is exactly what the N>1 host logic is parameterised on)."""
import os
import socket

import subprocess
import sys

from picopose_amd.dist import shard_bounds


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds_cover_and_balance():
    for n, w in [(162, 8), (42, 4), (7, 2), (5, 8), (162, 1)]:
        sizes, pos = [], 0
        for r in range(w):
            lo, hi = shard_bounds(n, w, r)
            assert lo == pos
            pos = hi
            sizes.append(hi - lo)
        assert pos == n and max(sizes) - min(sizes) <= 1
    assert [shard_bounds(162, 8, r)[1] - shard_bounds(162, 8, r)[0] for r in range(8)] == [21, 21, 20, 20, 20, 20, 20, 20]


def test_sharded_matching_world2_gloo():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
         os.path.join(here, "dist_worker.py")],
        env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-2000:]
    assert "RANK0 OK" in out and "RANK1 OK" in out, out[-2000:]
