"""Rank body of tests/test_dist_gpu.py: the REAL template-sharded forward (HIP model, picopose_amd.dist.sharded_forward)
on two ranks that share one GPU (gloo rendezvous, every rank on cuda:0 — a one-GPU box cannot run RCCL between ranks),
against the single-process forward of the same crops.  This IS the strong-scaling form of BASELINE configs[3] in small:
a global batch of 4 crops and 7 templates -> 2 crops + 4 / 3 templates per rank.

Environment: PP_DIST_BACKEND = gloo (default) | nccl (RCCL; on a one-GPU box only world size 1 can run — it proves that the
RCCL path loads, takes the device tensors of dist.py and returns what gloo returns); PP_DIST_TURNS = 1 (default: the ranks take
turns on the shared card for their local compute) | 0 (fully concurrent: the stress form, see below).

The ranks take TURNS on the GPU for their local compute phases (the collectives still run between all ranks): two processes
computing on one MI355X at the same time is not a configuration of the product (one process per GPU), and on this platform
lane masks a kernel holds in SGPRs were seen corrupted under that load (the warp kernel lost taps in lanes 48-63 of
single waves until it was rewritten without lane-masked branches; tests/stress_pc.py, tests/stress_two.sh)."""
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

    backend = os.environ.get("PP_DIST_BACKEND", "gloo")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))       # exactly bench.py's call
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    turns = os.environ.get("PP_DIST_TURNS", "1") == "1" and world > 1

    def in_turn(fn):
        """fn runs on one rank at a time (every rank calls the wrapped hooks in the same order, so the barriers line up)."""
        if not turns:
            return fn

        def wrapped(*a, **k):
            out = None
            for r in range(world):
                if r == rank:
                    out = fn(*a, **k)
                    torch.cuda.synchronize()
                dist.barrier()
            return out
        return wrapped

    net = Net(small_cfg())
    net.load_state_dict(calibrated_state_dict(net.state_dict(), 4, "dinov2_vits14"))
    net = net.cuda().eval()
    bl, N, hyp = 2, 7, 3                                  # 2 crops per rank, 7 templates: uneven shards 4 + 3
    ep_all = {k: v.cuda() for k, v in make_end_points(bl * world, N, 55, dome=True).items()}
    with torch.no_grad():
        bank_all = in_turn(lambda: torch.stack([net.feature_extractor(ep_all["tem_rgb"][b])[-1] for b in range(bl * world)]))()
    ep_all["template_feature"] = bank_all
    own = slice(rank * bl, (rank + 1) * bl)
    lo, hi = shard_bounds(N, world, rank)
    ep = {k: v[own].contiguous() for k, v in ep_all.items() if k != "template_feature"}

    from picopose_amd import ops

    def features_fn(x):                                   # the defaults of sharded_forward, taking turns
        toks, (h0, w0) = net.feature_extractor.forward_tokens(x)
        return (toks, (h0, w0), None), ops.tokens_to_nchw(toks[-1], 1, h0, w0)

    def overlap_fn(real):
        toks, (h0, w0), _ = real
        return toks, (h0, w0), net.offset_regressor.dpt_head.forward_nhwc([t[:, 1:].unflatten(1, (h0, w0)) for t in toks])

    scores_fn = lambda b, q, m: hm.template_scores(b, q, m, mode=net.match_mode)  # noqa: E731
    torch.cuda.synchronize()
    dist.barrier()
    got = sharded_forward(net, ep, bank_all[:, lo:hi].contiguous(), N, hyp=hyp, features_fn=in_turn(features_fn),
                          overlap_fn=in_turn(overlap_fn), scores_fn=in_turn(scores_fn), topk_fn=hm.topk_templates,
                          tail_fn=in_turn(net.forward_hypotheses))
    # single-process reference on this rank's crops with the whole bank
    ref_in = dict(ep)
    ref_in["template_feature"] = bank_all[own].contiguous()
    want = in_turn(lambda: net(ref_in, hyp))()
    ok = len(got) == len(want) == hyp
    for h, (g, w) in enumerate(zip(got, want)):
        for key in w:
            same = torch.equal(g[key], w[key])
            if not same:   # (diagnostics: which output, how far)
                diff = (g[key].double() - w[key].double()).abs()
                print(f"RANK{rank} hyp {h} {key}: {int((diff > 0).sum())} of {diff.numel()} entries differ, max {float(diff.max()):.3e}", flush=True)
            ok = ok and same
    # and the stage-1-only entry point
    q = torch.randn(bl * world, 384, 16, 16, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    s, i = sharded_matching_templates(bank_all[:, lo:hi].contiguous(), q, ep_all["real_mask"], N, topk=4,
                                      score_fn=in_turn(lambda b, qq, m: hm.template_scores(b, qq, m, mode=None)), topk_fn=hm.topk_templates)
    ws, wi = in_turn(lambda: hm.matching_templates(bank_all, q, None, ep_all["real_mask"], topk=4))()
    if not (torch.equal(i, wi) and torch.equal(s, ws)):
        print(f"RANK{rank} stage-1 entry point: ids equal {torch.equal(i, wi)}, max score diff {float((s - ws).abs().max()):.3e}", flush=True)
    ok = ok and torch.equal(i, wi) and torch.equal(s, ws)
    torch.cuda.synchronize()
    # gather_scores on its own: uneven slices, -inf padding, rank-major reassembly
    from picopose_amd.dist import gather_scores

    full = torch.arange(bl * world * N, device="cuda", dtype=torch.float32).view(bl * world, N)
    ok = ok and torch.equal(gather_scores(full[:, lo:hi].contiguous(), N), full)
    torch.cuda.synchronize()
    print(f"RANK{rank} {'OK' if ok else 'MISMATCH'} backend={dist.get_backend()} world={world} turns={int(turns)}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
