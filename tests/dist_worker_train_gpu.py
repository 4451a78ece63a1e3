"""Rank body of tests/test_dist_gpu.py::test_data_parallel_training_step_*: the DDP training step of run_train.py:109-130 with the
real HIP model — every rank runs forward_train + Loss + backward on ITS OWN batch of pairs, picopose_amd.dist.allreduce_gradients
averages the gradients (several buckets), then every rank must hold the mean of the per-rank gradients, which rank-independent
recomputation checks: each rank also computes every other rank's gradients locally (fresh model copies: training-mode BatchNorm
moves the running buffers) and averages them itself.  The scatter adjoints (warp, correlation lookup) use fp32 atomics, so two
computations of the same gradient agree to rounding, not bit for bit: 2e-5 of a tensor's largest gradient.
Two ranks share one GPU (gloo rendezvous; PP_DIST_BACKEND=nccl at world size 1 pushes the same buckets through RCCL); the ranks take
turns on the card for their compute, as in dist_worker_gpu.py."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    from netcfg import make_train_end_points, small_cfg

    from picopose_amd.dist import allreduce_gradients
    from picopose_amd.picopose import Net
    from picopose_amd.utils.loss_utils import Loss
    from picopose_amd.utils.seeding import calibrated_state_dict

    if os.environ.get("PP_DETERMINISTIC") == "1":     # fixed-point scatter adjoints: two computations of a gradient give the same bits
        from picopose_amd import autograd

        autograd.DETERMINISTIC = True
    backend = os.environ.get("PP_DIST_BACKEND", "gloo")
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    def grads_of(r):
        """a fresh copy of the model (same weights on every rank) holding the gradients of rank r's batch"""
        import numpy as np

        net = Net(small_cfg())
        net.load_state_dict(calibrated_state_dict(net.state_dict(), 4, "dinov2_vits14"))
        net = net.cuda().train()
        ep = {k: v.cuda() for k, v in make_train_end_points(2, 100 + r).items()}
        np.random.seed(700 + r)                     # the stage-3 noise of utils/augment.aug_gtM_noise
        torch.manual_seed(900 + r)
        Loss()(net(ep))["loss"].backward()
        torch.cuda.synchronize()
        return net

    turn = dist.new_group(backend="gloo") if world > 1 else None    # its own queue: independent of the gradient collectives' order

    def in_turn(fn):
        out = None
        for q in range(world):                      # one rank at a time on the card
            if q == rank:
                out = fn()
            if turn is not None:
                dist.barrier(group=turn)
        return out

    local = in_turn(lambda: {r: grads_of(r) for r in range(world)})          # every rank's gradients, recomputed here
    if os.environ.get("PP_DDP") == "torch":
        # torch's own DistributedDataParallel around the HIP model (what LightningLite's strategy='ddp' builds, utils/lite.py):
        # its reducer all-reduces gradient buckets INSIDE backward, so the ranks compute concurrently on the shared card here
        # (the stress form); find_unused_parameters for the checkpoint's dead layers (dpt.py:241-249, dinov2.norm, mask_token)
        import numpy as np
        from torch.nn.parallel import DistributedDataParallel as DDP

        mine = Net(small_cfg())
        mine.load_state_dict(calibrated_state_dict(mine.state_dict(), 4, "dinov2_vits14"))
        mine = mine.cuda().train()
        ddp = DDP(mine, device_ids=[0], find_unused_parameters=True, bucket_cap_mb=4)
        ep = {k: v.cuda() for k, v in make_train_end_points(2, 100 + rank).items()}
        np.random.seed(700 + rank)
        torch.manual_seed(900 + rank)
        Loss()(ddp(ep))["loss"].backward()
        torch.cuda.synchronize()
        nb = 3
    elif os.environ.get("PP_DDP") == "buckets":
        # the OVERLAPPED form (picopose_amd.dist.GradientBuckets): every bucket's all-reduce is issued by a gradient hook while
        # backward is still running; the ranks still take turns on the card (a rank's buckets wait for its peer's), then finish()
        import numpy as np

        from picopose_amd.dist import GradientBuckets

        mine = Net(small_cfg())
        mine.load_state_dict(calibrated_state_dict(mine.state_dict(), 4, "dinov2_vits14"))
        mine = mine.cuda().train()
        trains = {n for n, p in local[rank].named_parameters() if p.grad is not None}
        gb = GradientBuckets([p for n, p in mine.named_parameters() if n in trains], bucket_bytes=4 << 20)

        def step():
            ep = {k: v.cuda() for k, v in make_train_end_points(2, 100 + rank).items()}
            np.random.seed(700 + rank)
            torch.manual_seed(900 + rank)
            Loss()(mine(ep))["loss"].backward()
            torch.cuda.synchronize()

        in_turn(step)
        launched = gb.launched_in_backward
        nb = gb.finish()
        assert launched == nb == len(gb.buckets), (launched, nb, len(gb.buckets))
    else:
        mine = in_turn(lambda: grads_of(rank))                               # the copy that is all-reduced
        nb = allreduce_gradients(list(mine.parameters()), bucket_bytes=4 << 20)
    torch.cuda.synchronize()
    ok, worst, n, worst_name = nb >= 3, 0.0, 0, ""
    for (name, p), *cols in zip(mine.named_parameters(), *[local[r].parameters() for r in range(world)]):
        if p.grad is None or all(c.grad is None for c in cols):        # (DDP's reducer leaves zeros in the unused parameters)
            ok = ok and all(c.grad is None for c in cols) and (p.grad is None or float(p.grad.abs().max()) == 0.0)
            continue
        want = sum(c.grad for c in cols) / world
        top = float(want.abs().max())
        n += 1
        if top > 1e-7:                               # (analytically zero gradients hold rounding noise)
            e = float((p.grad - want).abs().max()) / top
            if e > worst:
                worst, worst_name = e, name
            if e > 2e-5 and os.environ.get("PP_DDP_VERBOSE") == "1":
                d = (p.grad - want).abs()
                print(f"RANK{rank} off: {name} e={e:.2e} elements off by > 1e-6 top: {int((d > 1e-6 * top).sum())} of {d.numel()}", flush=True)
    ok = ok and n >= 300 and worst <= 2e-5
    print(f"RANK{rank} {'OK' if ok else 'MISMATCH'} backend={dist.get_backend()} world={world} buckets={nb} tensors={n} worst={worst:.2e} ({worst_name})", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
