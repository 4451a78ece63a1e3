"""Rank body of tests/test_dist_gpu.py: the REAL template-sharded forward (HIP model, picopose_amd.dist.sharded_forward)
on two ranks that share one GPU (gloo rendezvous, every rank on cuda:0 — a one-GPU box cannot run RCCL between ranks),
against the single-process forward of the same crops."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    from netcfg import make_end_points, small_cfg

    from picopose_amd.dist import shard_bounds, sharded_forward, sharded_matching_templates
    from picopose_amd.picopose import Net
    from picopose_amd.utils import matching as hm
    from picopose_amd.utils.seeding import calibrated_state_dict

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    net = Net(small_cfg())
    net.load_state_dict(calibrated_state_dict(net.state_dict(), 4, "dinov2_vits14"))
    net = net.cuda().eval()
    bl, N, hyp = 2, 7, 3                                  # 2 crops per rank, 7 templates: uneven shards 4 + 3
    ep_all = {k: v.cuda() for k, v in make_end_points(bl * world, N, 55, dome=True).items()}
    with torch.no_grad():
        bank_all = torch.stack([net.feature_extractor(ep_all["tem_rgb"][b])[-1] for b in range(bl * world)])
    ep_all["template_feature"] = bank_all
    own = slice(rank * bl, (rank + 1) * bl)
    lo, hi = shard_bounds(N, world, rank)
    ep = {k: v[own].contiguous() for k, v in ep_all.items() if k != "template_feature"}
    got = sharded_forward(net, ep, bank_all[:, lo:hi].contiguous(), N, hyp=hyp)
    # single-process reference on this rank's crops with the whole bank
    ref_in = dict(ep)
    ref_in["template_feature"] = bank_all[own].contiguous()
    want = net(ref_in, hyp)
    ok = len(got) == len(want) == hyp
    for g, w in zip(got, want):
        for key in w:
            ok = ok and torch.equal(g[key], w[key])
    # and the stage-1-only entry point
    q = torch.randn(bl * world, 384, 16, 16, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    s, i = sharded_matching_templates(bank_all[:, lo:hi].contiguous(), q, ep_all["real_mask"], N, topk=4)
    ws, wi = hm.matching_templates(bank_all, q, None, ep_all["real_mask"], topk=4)
    ok = ok and torch.equal(i, wi) and torch.equal(s, ws)
    torch.cuda.synchronize()
    print(f"RANK{rank} {'OK' if ok else 'MISMATCH'}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
