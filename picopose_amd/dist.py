"""Template-bank sharding of stage 1 across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI).  The
template axis N is cut into contiguous slices; each rank scores its slice of
every crop's bank with the fused HIP kernel, then ONE all-gather of the
(B, ceil(N/G)) fp32 score slices gives every rank the full (B, N) sim_avg and an
identical local top-k.  The message is a few KB per rank: latency-bound, a single
hop on the fully connected xGMI mesh; nothing else on the path is exchanged.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_total, world, rank):
    """Contiguous slice [lo, hi) of rank `rank`: sizes ceil(N/G) for the first N%G ranks, floor after."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_scores(local_scores, n_total, group=None):
    """All-gather (B, n_local) slices into the full (B, n_total) matrix on every rank."""
    world = dist.get_world_size(group)
    B = local_scores.shape[0]
    n_max = -(-n_total // world)
    padded = local_scores.new_full((B, n_max), float("-inf"))
    padded[:, : local_scores.shape[1]] = local_scores
    out = local_scores.new_empty((world * B, n_max))  # rank-major concatenation along dim 0
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    out = out.view(world, B, n_max)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, world, r)
        parts.append(out[r, :, : hi - lo])
    return torch.cat(parts, dim=1)


def sharded_matching_templates(local_bank, tar_feat, tar_mask, n_total, topk=5, group=None,
                               score_fn=None, topk_fn=None, mode=None):
    """matching_templates (reference utils/matching.py:29-69) with the bank sharded over ranks.

    local_bank: (B, n_local, C, 16, 16) — this rank's slice [shard_bounds) of every crop's bank.
    Returns (pred_score_src, pred_id_src) with GLOBAL template ids, identical on every rank.
    score_fn/topk_fn default to the HIP kernels; tests inject the CPU oracle to run under gloo.
    """
    if score_fn is None:
        from .utils import matching as hm

        score_fn = lambda b, q, m: hm.template_scores(b, q, m, mode=mode)  # noqa: E731
        topk_fn = hm.topk_templates
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_bounds(n_total, world, rank)
    assert local_bank.shape[1] == hi - lo, (local_bank.shape, lo, hi)
    local = score_fn(local_bank, tar_feat, tar_mask)
    full = gather_scores(local, n_total, group=group)
    return topk_fn(full, topk)
