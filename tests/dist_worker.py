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
    print(f"RANK{rank} {'OK' if ok else 'MISMATCH'}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
