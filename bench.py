#!/usr/bin/env python3
"""bench.py — throughput of the PicoPose hot path on MI355X (driver contract: one JSON line).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--mode fast|exact]

A "step" is one pass of stage-1 template matching (reference utils/matching.py:29-69:
normalise, 256x256xC similarity per template, masked best-match mean, top-k) over one batch of
synthetic crops with a per-crop fp32 template bank resident in HBM.  Default workload =
BASELINE.json configs[2]'s stage-1 shape (batch 32, 162 templates, ViT-B width 768: a 4.08 GB
bank, larger than the 256 MiB Infinity Cache so the HBM roofline is honest).  Stages 2-3 are
not yet on the HIP path ("scope of this round"), so `config.workload` says stage 1.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the template axis is
sharded over the ranks and the global batch grows with N (32*N crops per step), so per-GPU
bank bytes are fixed ("weak"); the only exchange is one all-gather of the (B, N/G) scores.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (crops per GPU-step, templates, channels, BASELINE config it is the stage-1 shape of)
    "stage1_b32_n162_c768": (32, 162, 768, "configs[2] stage-1 (batch 32, 162 templates, ViT-B/14)"),
    "stage1_b8_n42_c384": (8, 42, 384, "configs[1] (batch 8, 42 templates, ViT-S/14, stage-1 only)"),
    "stage1_b32_n162_c1024": (32, 162, 1024, "base.yaml shape (ViT-L/14) stage-1"),
}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 measured achievable


def disk_mask(B, device):
    yy, xx = torch.meshgrid(torch.arange(224.0), torch.arange(224.0), indexing="ij")
    m = (((yy - 111.5) ** 2 + (xx - 111.5) ** 2) < (0.4 * 224) ** 2).float()
    return m[None].repeat(B, 1, 1).to(device)


def make_inputs(B, N, C, device, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    bank = torch.randn(B, N, C, 16, 16, device=device, generator=g)
    query = torch.randn(B, C, 16, 16, device=device, generator=g)
    return bank, query, disk_mask(B, device)


def algorithmic_bytes(B, N, C):
    # per crop N*C*256*4 (bank, read once) + C*256*4 (query) + 256*4 (mask) + k*12
    return B * (N * C * 256 * 4 + C * 256 * 4 + 256 * 4 + 5 * 12)


def cpu_baseline(N, C, seconds_budget=20.0):
    """The CPU oracle (a port of the reference's torch-CPU path) on a bounded sample."""
    from oracle import matching as om

    # the GPU box hands one GPU's job a 16-core share of the host; never oversubscribe it
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    Bs = 4
    g = torch.Generator().manual_seed(1)
    bank = torch.randn(Bs, N, C, 16, 16, generator=g)
    query = torch.randn(Bs, C, 16, 16, generator=g)
    mask = disk_mask(Bs, "cpu")
    om.matching_templates(bank, query, None, mask, topk=5)  # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        om.matching_templates(bank, query, None, mask, topk=5)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > seconds_budget or reps >= 8:
            break
    return {"value": Bs * reps / dt, "unit": "crops/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x matching_templates on {Bs} crops x {N} templates x C={C} (same shape, "
                      f"smaller batch), torch CPU fp32, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="stage1_b32_n162_c768", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="fast", choices=["fast", "exact"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with {a.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from picopose_amd import _lib
    from picopose_amd.dist import shard_bounds, sharded_matching_templates
    from picopose_amd.utils import matching as hm

    Bg, N, C, cfg_name = WORKLOADS[a.workload]
    B = Bg * world  # global batch: every rank scores its template slice of ALL crops
    lo, hi = shard_bounds(N, world, rank)
    n_local = hi - lo
    # identical query/mask on every rank (seed 0); this rank's slice of every crop's bank (seed 1+rank)
    _, query, mask = make_inputs(B, 1, C, dev, 0)
    gb = torch.Generator(device=dev).manual_seed(1 + rank)
    bank = torch.randn(B, n_local, C, 16, 16, device=dev, generator=gb)

    def step():
        if distributed:
            return sharded_matching_templates(bank, query, mask, N, topk=5, mode=a.mode)
        return hm.matching_templates(bank, query, None, mask, topk=5, mode=a.mode)

    for _ in range(a.warmup):
        out = step()
    torch.cuda.synchronize()
    L = _lib.lib()
    _lib.check(L.pp_prof_enable(a.steps), "pp_prof_enable")
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    buf = (ctypes.c_float * a.steps)()
    cnt = ctypes.c_int()
    _lib.check(L.pp_prof_collect(buf, a.steps, ctypes.byref(cnt)), "pp_prof_collect")
    _lib.check(L.pp_prof_enable(0), "pp_prof_enable")
    kern_ms = sum(buf[i] for i in range(cnt.value)) / max(cnt.value, 1)

    if distributed:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        assert out[1].shape == (B, 5) and out[1].dtype == torch.int64
        ms = dt / a.steps * 1e3
        kbytes = algorithmic_bytes(B, n_local, C)  # bytes the roofline kernel launch streams on this rank
        traffic = None  # HBM bytes per launch from the PMC passes of the same command (profiles/, tools/pmc.sh)
        pmc = os.path.join(ROOT, "profiles", "r01", "pmc_traffic_stage1.json")
        if world == 1 and a.mode == "fast" and os.path.exists(pmc):
            rec = json.load(open(pmc))
            if rec.get("workload") == a.workload:
                traffic = rec["hbm_bytes_per_launch"]
        achieved = kbytes / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "image-crops/sec (224x224, 162 templates), stage-1 template matching",
            "value": B / (dt / a.steps), "unit": "crops/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 in / f16 MFMA operands / f32 accumulate" if a.mode == "fast" else "f32",
            "data": "synthetic",
            "config": {"workload": f"{a.workload}: {cfg_name}; matching_templates only (stages 2-3 not on the HIP path yet)",
                       "global_batch": B, "templates": N, "channels": C, "mode": a.mode,
                       "parallelism": "single GPU" if world == 1 else f"template-shard x{world} + 1 RCCL all-gather of (B,N/G) scores"},
            "roofline": {"bound": "hbm", "kernel": f"s1_main<{a.mode}>", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": kbytes},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, C)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
