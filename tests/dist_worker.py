"""Rank body of tests/test_dist_cpu.py (launched with torch.distributed.run, backend gloo, CPU).
The CPU oracle stands in for the HIP scorer: the N>1 host logic is parameterised on it."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import matching as om  # noqa: E402
from picopose_amd.dist import shard_bounds, sharded_matching_templates  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.set_num_threads(2)
    g = torch.Generator().manual_seed(5)
    B, N, C = 3, 7, 32  # 7 templates over 2 ranks: uneven slices 4 + 3
    bank = torch.randn(B, N, C, 16, 16, generator=g)
    query = torch.randn(B, C, 16, 16, generator=g)
    mask = (torch.rand(B, 224, 224, generator=g) < 0.7).float()
    lo, hi = shard_bounds(N, world, rank)
    s, i = sharded_matching_templates(
        bank[:, lo:hi].contiguous(), query, mask, N, topk=4,
        score_fn=lambda b, qq, m: om.template_scores(b, qq, m),
        topk_fn=lambda sc, k: torch.topk(sc, k, dim=1))
    rs, ri = om.matching_templates(bank, query, None, mask, topk=4)
    ok = torch.equal(i, ri) and float((s - rs).abs().max()) <= 1e-6

    # full-path orchestration (crops data-parallel, feature bank template-sharded): CPU stand-ins for the stages
    from picopose_amd.dist import sharded_forward

    bl = 2                                            # crops per rank
    g2 = torch.Generator().manual_seed(9)
    Bt = bl * world
    bank_all = torch.randn(Bt, N, C, 16, 16, generator=g2)
    rgb_all = torch.randn(Bt, C, 16, 16, generator=g2)     # stand-in "image": the feature extractor is the identity
    mask_all = (torch.rand(Bt, 224, 224, generator=g2) < 0.7).float()
    own = slice(rank * bl, (rank + 1) * bl)
    ep = {"real_rgb": rgb_all[own], "real_mask": mask_all[own]}
    outs = sharded_forward(None, ep, bank_all[:, lo:hi].contiguous(), N, hyp=3,
                           features_fn=lambda x: ("state", x), scores_fn=lambda b, qq, m: om.template_scores(b, qq, m),
                           topk_fn=lambda sc, k: torch.topk(sc, k, dim=1), tail_fn=lambda e, ids, real: ids)
    _, want = om.matching_templates(bank_all[own], rgb_all[own], None, mask_all[own], topk=3)
    ok = ok and torch.equal(outs, want)
    # DDP gradient averaging of the backward slice: rank r holds (r + 1) x the REFERENCE's gradients of the slice's parameters
    # (tests/golden/train_grads.npz); after allreduce_gradients every rank holds their mean, 1.5 x — in several buckets
    import numpy as np

    from picopose_amd.dist import allreduce_gradients

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_grads.npz"))
    names = sorted(k for k in z.files if k.startswith("grad/"))
    params = []
    for k in names:
        p = torch.nn.Parameter(torch.zeros(z[k].shape))
        p.grad = torch.from_numpy(z[k]).clone() * (rank + 1)
        params.append(p)
    frozen = torch.nn.Parameter(torch.zeros(3))             # a parameter outside the slice: no .grad, skipped
    nb = allreduce_gradients(params + [frozen], bucket_bytes=1 << 18)
    ok = ok and nb >= 3 and frozen.grad is None
    for k, p in zip(names, params):
        ok = ok and torch.allclose(p.grad, torch.from_numpy(z[k]) * 1.5, rtol=1e-6, atol=1e-12)
    # ... and the OVERLAPPED form (GradientBuckets): a small network under real autograd, rank-dependent inputs; every bucket's
    # all-reduce is issued from a gradient hook during backward(), the result is the mean of the two ranks' gradients
    from picopose_amd.dist import GradientBuckets

    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 8))
    xs = [torch.randn(16, 40, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    want = []
    for r in range(world):
        net.zero_grad(set_to_none=True)
        net(xs[r]).square().sum().backward()
        want.append([p.grad.clone() for p in net.parameters()])
    mean = [sum(g) / world for g in zip(*want)]
    net.zero_grad(set_to_none=True)
    gb = GradientBuckets(list(net.parameters()), bucket_bytes=1 << 13)
    for step in range(2):                       # twice: the buckets re-arm
        net.zero_grad(set_to_none=True)
        net(xs[rank]).square().sum().backward()
        launched = gb.launched_in_backward
        nb2 = gb.finish()
        ok = ok and nb2 == len(gb.buckets) >= 3 and launched == (step + 1) * len(gb.buckets)      # every bucket went out DURING backward
        for p, m in zip(net.parameters(), mean):
            ok = ok and torch.allclose(p.grad, m, rtol=1e-5, atol=1e-7)
    # a second backward() before finish() (gradient accumulation) is refused instead of silently averaging the first micro-batch only
    net.zero_grad(set_to_none=True)
    net(xs[rank]).square().sum().backward()
    try:
        net(xs[rank]).square().sum().backward()
        ok = False
    except RuntimeError as e:
        ok = ok and "one backward() per finish()" in str(e)
    gb.finish()
    gb.remove()
    print(f"RANK{rank} {'OK' if ok else 'MISMATCH'}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
