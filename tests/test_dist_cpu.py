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


def test_local_world_eight_ranks_at_the_configs3_shard_layout():
    """picopose_amd.dist.LocalWorld (every rank a thread of ONE process, collectives real exchanges) at the shard layout of BASELINE
    configs[3]: world 8, global batch 32, 162 templates -> 4 crops + 21 / 20 templates per rank (narrow C: the oracle scores on CPU).
    Every rank gets the full-bank result of its own crops from `sharded_forward`, and `sharded_matching_templates` the global one."""
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import picopose_amd.dist as pd
    from oracle import matching as om

    torch.set_num_threads(4)
    world, bl, N, C, hyp = 8, 4, 162, 8, 5
    g = torch.Generator().manual_seed(11)
    Bt = world * bl
    bank = torch.randn(Bt, N, C, 16, 16, generator=g)
    rgb = torch.randn(Bt, C, 16, 16, generator=g)
    mask = (torch.rand(Bt, 224, 224, generator=g) < 0.7).float()
    score = lambda b, q, m: om.template_scores(b, q, m)            # noqa: E731
    topk = lambda sc, k: torch.topk(sc, k, dim=1)                  # noqa: E731

    def body(rank):
        lo, hi = pd.shard_bounds(N, world, rank)
        own = slice(rank * bl, (rank + 1) * bl)
        ids = pd.sharded_forward(None, {"real_rgb": rgb[own], "real_mask": mask[own]}, bank[:, lo:hi].contiguous(), N, hyp=hyp,
                                 features_fn=lambda x: ("state", x), scores_fn=score, topk_fn=topk, tail_fn=lambda e, i, real: i)
        s, i = pd.sharded_matching_templates(bank[:, lo:hi].contiguous(), rgb, mask, N, topk=hyp, score_fn=score, topk_fn=topk)
        return ids, s, i

    lw = pd.LocalWorld(world)
    with lw.installed():
        outs = lw.run(body)
    assert pd.dist is torch.distributed
    ws, wi = om.matching_templates(bank, rgb, None, mask, topk=hyp)
    for rank, (ids, s, i) in enumerate(outs):
        assert torch.equal(ids, wi[rank * bl:(rank + 1) * bl]), rank
        assert torch.equal(i, wi) and float((s - ws).abs().max()) <= 1e-6, rank


def test_local_world_deposits_are_copies_and_single_process_buckets_allow_accumulation():
    """ LocalWorld.all_gather_into_tensor deposits a COPY of a rank's contribution: the rank that resumes first may overwrite
    its input buffer (the model's per-shape scratch) before the others have read it.  GradientBuckets in a single process: the hooks do
    nothing — several backward() calls per finish() (gradient accumulation) work and accumulate."""
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import picopose_amd.dist as pd

    world = 4
    shared = torch.zeros(3)                       # ONE buffer every rank thread fills with its own value, like a per-shape scratch

    def body(rank):
        out = torch.empty(world, 3)
        shared.fill_(float(rank + 1))
        pd.dist.all_gather_into_tensor(out, shared)
        shared.fill_(-1.0)                        # legal after a synchronous collective
        return out

    lw = pd.LocalWorld(world)
    with lw.installed():
        outs = lw.run(body)
    for o in outs:
        assert torch.equal(o, torch.arange(1.0, world + 1)[:, None].expand(world, 3)), o

    net = torch.nn.Linear(5, 3)
    gb = pd.GradientBuckets(list(net.parameters()))
    x = torch.randn(4, 5)
    for _ in range(3):                            # three micro-batches, one finish(): no error in a single process
        net(x).sum().backward()
    assert gb.finish() == 0
    one = torch.autograd.grad(net(x).sum(), list(net.parameters()))
    for p, g1 in zip(net.parameters(), one):
        assert torch.allclose(p.grad, 3 * g1)
    gb.remove()
