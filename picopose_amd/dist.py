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


def sharded_forward(net, local_end_points, local_bank, n_total, hyp=5, group=None, features_fn=None, scores_fn=None,
                    topk_fn=None, tail_fn=None, overlap_fn=None, mark=None):
    """Net.forward (eval) with the template FEATURE bank sharded over the ranks and the crops data-parallel.

    Every rank owns `b_local` crops (`local_end_points`, with the raw template data of ITS crops) and the slice
    [shard_bounds) of the feature bank of ALL crops: local_bank (world*b_local, n_local, C, 16, 16), crops ordered
    rank-major.  Exchange steps: all-gather of the query features and masks (every rank scores its template slice
    against all crops) and the all-gather of the (B_total, n_local) score slices; stages 2-3 run on the rank's own
    crops.  The query-side DPT head (overlap_fn: state -> state) does not feed stage 1, so it runs while the all-gathers
    of the query features are in flight.  *_fn default to the HIP model; tests inject CPU stand-ins to run under gloo.
    mark(name): optional hook called on the compute stream at the phase boundaries "start", "features" (query ViT + DPT done),
    "exchange_q" (query / mask all-gathers waited for), "stage1" (local score slices), "exchange_s" (score all-gather + top-k),
    "tail" (stages 2-3 of the own crops) — bench.py records a HIP event at each to decompose a step.
    Returns the list (hyp) of output dicts for the rank's own crops."""
    mark = mark or (lambda name: None)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if features_fn is None:
        from .utils import matching as hm

        from . import ops

        def features_fn(x):
            toks, (h0, w0) = net.feature_extractor.forward_tokens(x)
            return (toks, (h0, w0), None), ops.tokens_to_nchw(toks[-1], 1, h0, w0)

        def overlap_fn(real):   # query-side DPT maps, computed once for all hypotheses (picopose.py forward_test)
            toks, (h0, w0), _ = real
            return toks, (h0, w0), net.offset_regressor.dpt_head.forward_nhwc([t[:, 1:].unflatten(1, (h0, w0)) for t in toks])

        scores_fn = lambda b, q, m: hm.template_scores(b, q, m, mode=net.match_mode)  # noqa: E731
        topk_fn = hm.topk_templates
        tail_fn = net.forward_hypotheses
    with torch.no_grad():
        mark("start")
        real, q_local = features_fn(local_end_points["real_rgb"])          # (state for stages 2-3, (b,C,16,16))
        b_local = q_local.shape[0]
        q_all = q_local.new_empty((world * b_local,) + tuple(q_local.shape[1:]))
        pending = [dist.all_gather_into_tensor(q_all, q_local.contiguous(), group=group, async_op=True)]
        # stage 1 reads the query mask only at the nearest-sampled 16 x 16 patch grid (utils/matching.py:41-43,
        # F.interpolate(mode="nearest") = rows/columns floor(i * H / 16)): exchange that kilobyte, not the full mask
        m_full = local_end_points["real_mask"]
        ih = (torch.arange(16, device=m_full.device) * m_full.shape[1]) // 16
        iw = (torch.arange(16, device=m_full.device) * m_full.shape[2]) // 16
        m_local = m_full[:, ih][:, :, iw].contiguous()
        m_all = m_local.new_empty((world * b_local,) + tuple(m_local.shape[1:]))
        pending.append(dist.all_gather_into_tensor(m_all, m_local, group=group, async_op=True))
        if overlap_fn is not None:
            real = overlap_fn(real)
        mark("features")
        for wk in pending:
            wk.wait()
        mark("exchange_q")
        lo, hi = shard_bounds(n_total, world, rank)
        assert local_bank.shape[0] == world * b_local and local_bank.shape[1] == hi - lo
        local = scores_fn(local_bank, q_all, m_all)
        mark("stage1")
        full = gather_scores(local, n_total, group=group)                                    # (B_total, N)
        _, ids = topk_fn(full[rank * b_local:(rank + 1) * b_local].contiguous(), hyp)
        mark("exchange_s")
        outs = tail_fn(local_end_points, ids, real)
        mark("tail")
        return outs


class LocalWorld:
    """A one-process stand-in for `torch.distributed` that runs ALL ranks of a sharded job: every rank is a thread, and a collective is
    a REAL exchange between them (an all-gather returns every rank's contribution, not copies of the caller's).  It exists so that the
    real shard shapes of BASELINE configs[3] — 8 ranks, 4 crops and 21 / 20 templates each — run through `sharded_forward` on a one-GPU
    box (tests/test_dist_gpu.py); the RCCL hop itself is the one thing it does not exercise.  (`bench.py --emulate-world` is a different,
    cheaper stand-in: ONE rank's share with every all-gather replaced by copies of that rank's own contribution.)

    Ranks take TURNS: a thread computes only while it holds the run token and hands it over when it waits for a collective, so between two
    collectives a rank's launches are never interleaved with another rank's (the model keeps per-shape scratch buffers, and one process per
    GPU is the product's configuration).  All ranks enqueue on the same HIP stream, whose order makes a producer's writes visible to the
    ranks that copy them after the rendezvous.

        world = LocalWorld(8)
        with world.installed():                       # picopose_amd.dist.dist -> this object
            outs = world.run(lambda rank: sharded_forward(net, ep[rank], bank[rank], N))
    """
    ReduceOp = dist.ReduceOp

    class _Work:
        def __init__(self, world, seq, out, shape):
            self.world, self.seq, self.out, self.shape, self.done = world, seq, out, shape, False

        def wait(self):
            if self.done:
                return None
            w = self.world
            w._token.release()               # hand the card to the next rank while this one waits for the others
            try:
                w._barrier.wait()
            finally:
                w._token.acquire()
            parts = w._slots[self.seq]
            self.out.view(w.world, *self.shape).copy_(torch.stack([parts[r] for r in range(w.world)]))
            w._taken[self.seq] = w._taken.get(self.seq, 0) + 1
            if w._taken[self.seq] == w.world:    # every rank has its copy: drop the contributions
                del w._slots[self.seq], w._taken[self.seq]
            self.done = True
            return None

    def __init__(self, world):
        import threading

        self.world = int(world)
        self._tls = threading.local()
        self._token = threading.Lock()
        self._barrier = threading.Barrier(self.world)
        self._slots, self._taken = {}, {}

    # ---- the subset of torch.distributed this module uses
    def get_rank(self, group=None):
        return self._tls.rank

    def get_world_size(self, group=None):
        return self.world

    def is_available(self):
        return True

    def is_initialized(self):
        return True

    def all_gather_into_tensor(self, out, inp, group=None, async_op=False):
        seq = self._tls.seq                  # the n-th collective of a rank meets the n-th collective of every other rank
        self._tls.seq += 1
        assert out.numel() == self.world * inp.numel(), (tuple(out.shape), tuple(inp.shape))
        # deposited as a COPY (stream-ordered, a few KB .. MB): the depositing rank resumes before the others have read its contribution and
        # may legally overwrite `inp` (the model's scratch buffers are per shape and shared by all rank threads of this process)
        self._slots.setdefault(seq, {})[self._tls.rank] = inp.clone()
        work = LocalWorld._Work(self, seq, out, tuple(inp.shape))
        if async_op:
            return work
        work.wait()
        return None

    def barrier(self, group=None):
        self._token.release()
        try:
            self._barrier.wait()
        finally:
            self._token.acquire()

    # ---- driver
    def installed(self):
        """Context manager: this module's `dist` is the stand-in inside, torch.distributed again outside."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            g = globals()
            old = g["dist"]
            g["dist"] = self
            try:
                yield self
            finally:
                g["dist"] = old
        return cm()

    def run(self, fn):
        """fn(rank) on every rank (one thread each, taking turns); returns [fn(0), ..., fn(world - 1)] or re-raises a rank's exception."""
        import threading

        results, errors = [None] * self.world, [None] * self.world

        def body(rank):
            self._tls.rank, self._tls.seq = rank, 0
            self._token.acquire()
            try:
                results[rank] = fn(rank)
            except BaseException as e:      # noqa: BLE001  (re-raised in the caller; the others must not wait for this rank)
                errors[rank] = e
                self._barrier.abort()
            finally:
                self._token.release()

        threads = [threading.Thread(target=body, args=(r,), name=f"rank{r}") for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None) or \
            next((e for e in errors if e is not None), None)
        if first is not None:
            raise first
        return results


def allreduce_gradients(parameters, group=None, bucket_bytes=25 << 20):
    """The gradient averaging of DistributedDataParallel (utils/lite.py / run_train.py:109-130 train with strategy='ddp') for the
    parameters that carry a `.grad` — with the default scope of picopose_amd/autograd.py ("full") every parameter the reference
    trains: the ViT, the affine regressor, the DPT head and the flow decoder (338 tensors at ViT-S).  Gradients are packed into flat buckets of <= bucket_bytes in parameter order (every rank builds the same buckets: the
    set of parameters with a gradient is the same on all ranks), each bucket is ONE all-reduce (RCCL over xGMI: a ring per bucket,
    per-link bound — 25 MB buckets keep the ring's latency term below 1 % of its transfer time at ~50 GB/s per link), then divided
    by the world size and copied back.  Returns the number of buckets.  Call after loss.backward(), before optimizer.step()."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0                                           # a single process: nothing to average
    world = dist.get_world_size(group)
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return 0
    buckets, cur, size = [], [], 0
    for g in grads:
        nb = g.numel() * g.element_size()
        if cur and size + nb > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append(g)
        size += nb
    buckets.append(cur)
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for g in b:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    return len(buckets)


class GradientBuckets:
    """The gradient averaging of DistributedDataParallel WITH its overlap (utils/lite.py:33-49 / run_train.py:109-118 train through
    Lightning DDP, which all-reduces a bucket as soon as its gradients exist, while backward continues): `allreduce_gradients` above
    runs after `backward()` has returned; this class issues every bucket's all-reduce from autograd's own hooks.

        buckets = GradientBuckets([p for p in net.parameters() if p.requires_grad and <p trains>])   # once
        loss.backward()            # hooks: a bucket whose last gradient has just been accumulated is packed and all-reduced (async)
        buckets.finish()           # wait, divide by the world size, copy back into the .grad tensors; re-arm for the next step
        optimizer.step()

    Buckets are filled in REVERSE parameter order (gradients arrive roughly output-to-input), <= bucket_bytes each (25 MB: the
    ring's latency term stays below 1 % of its transfer time at ~50 GB/s per xGMI link).  Every parameter handed in must receive a
    gradient in every step (a bucket waits for all of its members: pass the parameters that train — INTEGRATION.md section 6 — as
    a trainer passes them to its optimizer).  Single process / no process group: the hooks do nothing and finish() returns 0.
    Contract, as for DistributedDataParallel without find_unused_parameters: the SET of parameters that receive a gradient is the same
    on every rank in a step.  A rank-dependent set makes the ranks issue different collectives from their hooks — by the time finish()
    runs they are already queued, so it cannot be turned into an error after the fact (gloo aborts on the size mismatch, RCCL hangs);
    use `allreduce_gradients` after backward() for models with data-dependent unused parameters.  What IS detected: a second
    backward() before finish() (gradient accumulation), which would silently average only the first micro-batch — it raises."""

    def __init__(self, parameters, group=None, bucket_bytes=25 << 20):
        self.group = group
        self.params = [p for p in parameters if p.requires_grad]
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            nb = p.numel() * p.element_size()
            if cur and size + nb > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nb
        if cur:
            self.buckets.append(cur)
        self._where = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._left = [len(b) for b in self.buckets]
        self._inflight = {}
        self.launched_in_backward = 0
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    def _active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _hook(self, p):
        if not self._active():
            return       # a single process: nothing is counted, nothing raises — gradient accumulation over several backward() calls works
        i = self._where[id(p)]
        self._left[i] -= 1
        if self._left[i] < 0:
            # a SECOND backward() before finish() (gradient accumulation over micro-batches): the bucket was already reduced with the first
            # micro-batch's gradients and finish() would write that mean over the accumulated .grad — silently wrong gradients
            raise RuntimeError("GradientBuckets: a parameter received a second gradient before finish() — one backward() per finish(); "
                               "for gradient accumulation call allreduce_gradients() after the last micro-batch instead")
        if self._left[i] == 0:
            flat = torch.cat([q.grad.reshape(-1) for q in self.buckets[i]])
            self._inflight[i] = (flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.launched_in_backward += 1

    def finish(self):
        """Call after backward(): waits for the buckets in flight (and reduces, synchronously, any bucket whose hooks did not all
        fire — a parameter without a gradient this step), averages and writes the gradients back.  Returns the number of buckets."""
        n = 0
        if self._active():
            world = dist.get_world_size(self.group)
            for i, b in enumerate(self.buckets):
                if i in self._inflight:
                    flat, work = self._inflight.pop(i)
                    work.wait()
                else:
                    have = [q for q in b if q.grad is not None]
                    if not have:
                        continue
                    flat = torch.cat([q.grad.reshape(-1) for q in have])
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                    b = have
                flat.div_(world)
                off = 0
                for q in b:
                    q.grad.copy_(flat[off:off + q.numel()].view_as(q.grad))
                    off += q.numel()
                n += 1
        self._left = [len(b) for b in self.buckets]
        self._inflight.clear()
        return n

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
