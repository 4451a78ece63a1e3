#!/usr/bin/env python3
"""bench.py — throughput of the PicoPose correspondence hot path on MI355X (driver contract: ONE JSON line).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--mode fast|exact]

Default workload `full_b32_n162_vitb` = BASELINE.json configs[2] (the configuration the metric is quoted
on): batch 32 synthetic 224x224 crops, 162 templates per crop, DINOv2 ViT-B/14, hyp 5 — one step is
`Net.forward` (query ViT, stage-1 template matching over the 4.08 GB per-crop feature bank, 5 x [template
ViT, stage-2 affine regression, stage-3 DPT + flow decoder, keypoint selection]) followed by the batched
PnP/RANSAC of all 160 (crop, hypothesis) pairs and the copy of the poses to the host.  All inputs and the
bank are resident in HBM before the timed region (the bank is precomputed as run_test.py:120-134 does).
`stage1_*` workloads time reference utils/matching.py:29-69 alone (configs[1] and the stage-1 shape of
configs[2]).

`python bench.py --gpus N` (N > 1) from a plain shell starts its own ranks: before anything touches the GPU it
spawns `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process and relays its output
and return code (a process that has initialised the GPU is never re-exec'ed).  Under torch.distributed.run
(WORLD_SIZE set) it is a rank.

N > 1 (one rank per GPU, RCCL), two modes:
  --scaling weak (default; the 1/2/4/8-GPU curve of BASELINE's metric): every rank owns 32 crops (global batch 32*N) AND
    their banks — crops are independent units of the path (a crop is matched against its OWN template bank,
    run_test.py:157-162), so they are sharded across the ranks with NO data-path collective; the ranks meet only at the
    barrier around the timed region and in the max-over-ranks of its duration.  (`--shard templates` runs the weak batch
    on the template-sharded bank instead: every rank then scores (32 N) x (162 / N) pairs — the work of the unsharded
    32 x 162 — plus a 201 MB query all-gather at N = 8: overhead without benefit, kept for comparison.)
  --scaling strong (BASELINE configs[3] as written: "batch = 32 ... sharded 8-ways"): the GLOBAL batch stays at the
    workload's 32 crops (or --global-batch), every rank owns 32/N of them, and the template FEATURE bank is sharded over
    the ranks along the template axis (each rank scores its ceil(162/N)-template slice of ALL crops of the global batch),
    exchanged with all-gathers of the query features / masks and of the (B, 162/N) score slices; stages 2-3 and PnP run
    data-parallel on the rank's own crops.  The line carries `phases_ms` (HIP events at the phase boundaries of one extra
    untimed step: features, exchange, stage1, tail, pnp).

JSON extras: `roofline` = the dominant kernel of the workload.  Full path: the pre-split f16x3 GEMM/conv kernel
(75 % of the step; MFMA-bound) — executed MFMA flops of all its launches in one step / their summed durations,
measured live with HIP events around every launch (an extra, untimed step after the timed region, so the
events do not perturb `value`).  Stage-1 workloads: the fused similarity kernel vs the HBM roofline (events
around exactly that launch inside the timed steps); on the full path the same object is reported as
`roofline_stage1`.  `mfma` = whole-step FLOP rate; `cpu_baseline` = the CPU oracle (a port of the reference's
torch-CPU path) on a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
ns = types.SimpleNamespace

VIT = {  # name: (channels, heads, interaction_indexes, GFLOP per image)  — SURVEY.md §6.2
    "dinov2_vits14": (384, 6, [[0, 2], [3, 5], [6, 8], [9, 11]], 12.2),
    "dinov2_vitb14": (768, 12, [[0, 2], [3, 5], [6, 8], [9, 11]], 46.3),
    "dinov2_vitl14": (1024, 16, [[0, 5], [6, 11], [12, 17], [18, 23]], 162.0),
}
WORKLOADS = {
    # name: (kind, crops per GPU-step, templates, vit, description)
    "full_b32_n162_vitb": ("full", 32, 162, "dinov2_vitb14", "configs[2]: batch 32, 162 templates, ViT-B/14, stage1+2+3 + PnP/RANSAC, hyp 5"),
    "full_b8_n42_vits": ("full", 8, 42, "dinov2_vits14", "batch 8, 42 templates, ViT-S/14, stage1+2+3 + PnP/RANSAC, hyp 5"),
    # SURVEY.md §8(f) row 1 (NOT the headline: the reference recomputes the template ViT / DPT inside the step)
    "full_cached_b32_n162_vitb": ("full_cached", 32, 162, "dinov2_vitb14",
                                  "configs[2] with the EXTENDED template bank (template-side DPT maps precomputed too, SURVEY 8f row 1)"),
    "full_b8_n42_vitl": ("full", 8, 42, "dinov2_vitl14", "batch 8, 42 templates, ViT-L/14 (config/base.yaml backbone), stage1+2+3 + PnP/RANSAC, hyp 5"),
    # config/base.yaml, the reference's ONLY configuration: ViT-L/14, 162 templates, hyp 5, test batch 4 — and the same network at batch 32
    "full_b4_n162_vitl": ("full", 4, 162, "dinov2_vitl14", "config/base.yaml: ViT-L/14, 162 templates, hyp 5, test batch 4 (run_test.py:102), stage1+2+3 + PnP/RANSAC"),
    "full_b32_n162_vitl": ("full", 32, 162, "dinov2_vitl14", "config/base.yaml's network at batch 32: ViT-L/14, 162 templates, hyp 5, stage1+2+3 + PnP/RANSAC"),
    "stage1_b32_n162_c768": ("stage1", 32, 162, "dinov2_vitb14", "configs[2] stage-1 shape: matching_templates only"),
    "stage1_b8_n42_c384": ("stage1", 8, 42, "dinov2_vits14", "configs[1]: batch 8, 42 templates, ViT-S/14, stage-1 matching only"),
    "stage1_b32_n162_c1024": ("stage1", 32, 162, "dinov2_vitl14", "base.yaml shape (ViT-L/14), stage-1 matching only"),
    # BASELINE configs[4] groundwork (batch 64 / 512 templates / ViT-L / fp16 on 8 GPUs): what ONE rank of that job runs —
    # its 64-template slice of the fp16-stored feature bank for all 64 crops, and stages 2-3 + PnP on its own crops
    "stage1_b64_n64_c1024_f16bank": ("stage1", 64, 64, "dinov2_vitl14",
                                     "one rank's share of configs[4], stage 1 only: 64 crops x 64 of the 512 templates, ViT-L/14 width, bank stored fp16"),
    "full_b64_n64_vitl": ("full", 64, 64, "dinov2_vitl14",
                          "one rank's share of configs[4]: batch 64, 64 of the 512 templates (fp16 bank), ViT-L/14, stage1+2+3 + PnP/RANSAC, hyp 5, 224x224 "
                          "(256x256 is not a multiple of the 14-pixel patch: the reference asserts, patch_embed.py:73-74)"),
    # ... and configs[4] at its full size on ONE GPU (the 17 GB fp16 bank of all 512 templates x 64 crops is resident: everything
    # of that configuration except the 8-way sharding)
    "stage1_b64_n512_c1024_f16bank": ("stage1", 64, 512, "dinov2_vitl14",
                                      "configs[4] stage-1 shape on one GPU: 64 crops x 512 templates, ViT-L/14 width, bank stored fp16 (17.2 GB)"),
    "full_b64_n512_vitl": ("full", 64, 512, "dinov2_vitl14",
                           "configs[4] unsharded on one GPU: batch 64, 512 templates (fp16 bank), ViT-L/14, stage1+2+3 + PnP/RANSAC, hyp 5, 224x224"),
}
FP16_BANK = {"stage1_b64_n64_c1024_f16bank", "full_b64_n64_vitl", "stage1_b64_n512_c1024_f16bank", "full_b64_n512_vitl"}   # workloads whose template feature bank is stored fp16
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); a read-only probe reaches 6.2-6.4 TB/s
MFMA_F32_PEAK_TF = 157.3   # dense fp32-input MFMA peak (v_mfma_f32_32x32x2_f32)
MFMA_F16_PEAK_TF = 2500.0  # dense fp16 MFMA peak (v_mfma_f32_32x32x16_f16), MI355X_MICROARCH.md


def disk_mask(B, device):
    yy, xx = torch.meshgrid(torch.arange(224.0), torch.arange(224.0), indexing="ij")
    m = (((yy - 111.5) ** 2 + (xx - 111.5) ** 2) < (0.4 * 224) ** 2).float()
    return m[None].repeat(B, 1, 1).to(device)


def cpu_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))  # one GPU's job owns a 16-core share of the host


def stage1_bytes(B, N, C, bank_elem_bytes=4):
    # per crop N*C*256*4 (bank, read once; 2 B/elem when the bank is stored fp16) + C*256*4 (query)
    # + 256*4 (mask) + k*12
    return B * (N * C * 256 * bank_elem_bytes + C * 256 * 4 + 256 * 4 + 5 * 12)


def full_gflop_per_crop(N, vit, hyp=5, cached=False):
    C = VIT[vit][0]
    dpt = {"dinov2_vits14": 18.5, "dinov2_vitb14": 19.0, "dinov2_vitl14": 19.4}[vit]
    # executed work: the query-side DPT head runs once per forward, not once per hypothesis (picopose_amd/picopose.py);
    # with the extended template bank the template-side ViT and DPT are not executed in the step at all
    nvit = 1 if cached else 1 + hyp
    return nvit * VIT[vit][3] + nvit * dpt + hyp * 108.4 + hyp * 0.14 + N * 2 * 256 * 256 * C / 1e9


def make_cfg(vit):
    C, _, idx, _ = VIT[vit]
    return ns(hypothesis=5, stage1=ns(vit_type=vit, pretrained=False, interaction_indexes=idx),
              stage2=ns(in_channel=256, hidden_dim=256),
              stage3=ns(nclass=1, in_channels=C, use_bn=True, out_channels=[256, 512, 1024, 1024], num_levels=3, radius=4))


def seeded_weights(net, seed, vit):
    """Random-init weights of the architecture (no network for checkpoints) with the last layer of every prediction
    head calibrated (picopose_amd/utils/seeding.py) so that ~half of the 4096 key-point slots survive stage 3 and
    PnP/RANSAC receives thousands of correspondences per problem, as with a trained network."""
    from picopose_amd.utils.seeding import calibrated_state_dict

    sd = calibrated_state_dict(net.state_dict(), seed, vit)
    net.load_state_dict(sd)
    return sd


def dome_points(K, M, device, size=64, crop=224, z0=0.8, relief=0.08, radius=0.4):
    """(size,size,3) template-camera-frame points of a dome (8 cm of relief on a 17 cm wide object at 0.8 m) seen through crop
    affine M with intrinsics K: a geometrically consistent synthetic object, so that the key-points of stage 3 describe a pose
    PnP/RANSAC can find.  Same construction as tests/netcfg.py (DOME_M, DOME_RELIEF)."""
    c = torch.arange(size, dtype=torch.float32, device=device) * (crop / size) + crop / (2 * size)
    cy, cx = torch.meshgrid(c, c, indexing="ij")
    u, v = (cx - M[0, 2]) / M[0, 0], (cy - M[1, 2]) / M[1, 1]
    r2 = ((cx - (crop - 1) / 2) ** 2 + (cy - (crop - 1) / 2) ** 2) / (radius * crop) ** 2
    z = z0 - relief * (1 - r2).clamp_min(0)
    return torch.stack([(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z], dim=-1)


def make_end_points(B, N, device, seed):
    """Synthetic eval inputs of SURVEY.md §8(d), generated on the device (N(0,1) crops/templates, disk masks, BOP K).
    Deviations from §8(d), all for the PnP/RANSAC leg to do the work it does on real data: `tem_pts3d` is the dome of
    dome_points() instead of U(-0.1, 0.1) noise (random 3-D points admit no pose), both crop affines are centred on the
    principal point at scale 2 and the template mask is full (tests/netcfg.make_end_points explains why)."""
    g = torch.Generator(device=device).manual_seed(seed)
    K = torch.tensor([[572.4114, 0, 320], [0, 573.57043, 240], [0, 0, 1.0]], device=device)
    # both crops: scale 2, centred on the principal point (tests/netcfg.DOME_M): the object sits on the optical axis, so the
    # crop-to-crop similarity stage 2 predicts is a rigid motion for any relief (up to sub-pixel parallax)
    M = torch.tensor([[2.0, 0, 112.0 - 640.0], [0, 2.0, 112.0 - 480.0], [0, 0, 1.0]], device=device)
    c = torch.arange(64, device=device).float() * 3.5 + 1.75
    cx, cy = torch.meshgrid(c, c, indexing="ij")     # the dataset's layout: entry [i][j] = image point of crop pixel (x = c[i], y = c[j])
    pts = (torch.stack([cx, cy], dim=-1) - M[:2, 2]) / M[0, 0]
    ang = torch.rand(B, N, generator=g, device=device) * 6.2831853
    pose = torch.eye(4, device=device)[None, None].repeat(B, N, 1, 1)
    pose[..., 0, 0], pose[..., 0, 1], pose[..., 1, 0], pose[..., 1, 1] = ang.cos(), -ang.sin(), ang.sin(), ang.cos()
    pose[..., 2, 3] = 0.8
    return {
        "real_rgb": torch.randn(B, 3, 224, 224, device=device, generator=g),
        "real_mask": disk_mask(B, device), "real_K": K[None].repeat(B, 1, 1),
        "real_M": M[None].repeat(B, 1, 1),
        "real_pose": torch.eye(4, device=device)[None].repeat(B, 1, 1), "real_pts2d": pts[None].repeat(B, 1, 1, 1),
        "tem_rgb": torch.randn(B, N, 3, 224, 224, device=device, generator=g),
        # (the object fills the template crop: a disk mask would put half of the valid key-points into the band where the
        # up-sampled masked init flow is garbage — tests/netcfg.make_end_points)
        "tem_mask": torch.ones(B, N, 224, 224, device=device),
        "tem_pts3d": dome_points(K, M, device)[None, None].repeat(B, N, 1, 1, 1), "tem_pose": pose,
        "tem_K": K[None, None].repeat(B, N, 1, 1), "tem_M": M[None, None].repeat(B, N, 1, 1),
    }


def cpu_baseline_stage1(N, C, budget=20.0):
    from oracle import matching as om

    cores = cpu_cores()
    torch.set_num_threads(cores)
    Bs = 4
    g = torch.Generator().manual_seed(1)
    bank, query = torch.randn(Bs, N, C, 16, 16, generator=g), torch.randn(Bs, C, 16, 16, generator=g)
    mask = disk_mask(Bs, "cpu")
    om.matching_templates(bank, query, None, mask, topk=5)
    t0, reps = time.perf_counter(), 0
    while True:
        om.matching_templates(bank, query, None, mask, topk=5)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > budget or reps >= 8:
            break
    return {"value": Bs * reps / dt, "unit": "crops/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x matching_templates on {Bs} crops x {N} templates x C={C}, torch CPU fp32, {cores} threads"}


def cpu_baseline_full(N, vit, sd, budget=25.0):
    """Oracle Net.forward on ONE crop of the same shape (N templates, hyp 5); the oracle PnP is timed beside it."""
    from oracle import nets as on
    from oracle import pnp as opnp

    cores = cpu_cores()
    torch.set_num_threads(cores)
    C, heads, idx, _ = VIT[vit]
    take = [b[-1] for b in idx]
    ep = {k: v.cpu() for k, v in make_end_points(1, N, "cpu", 7).items()}
    with torch.no_grad():
        # the bank is precomputed outside the measure (as on the GPU); its values do not change the timing
        ep["template_feature"] = torch.randn(1, N, C, 16, 16, generator=torch.Generator().manual_seed(8))
        t0, reps = time.perf_counter(), 0
        while True:
            outs, _ = on.net_forward_test(sd, ep, 5, heads, take)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > budget or (reps >= 3 and dt > 10.0):      # a bounded sample: about 10 s of CPU work, at least three forwards
                break
        # the PnP/RANSAC oracle is numpy restating OpenCV's C++ solver: timed beside the forward, not inside `value`
        # (it would make the reference's CPU path look slower than it is)
        t1 = time.perf_counter()
        npts = []
        for k, o in enumerate(outs):      # run_test.py:168-184: one PnP problem per (instance, hypothesis)
            opnp.pose_recovery_ransac_pnp(o["tar_pts_2d"][0].numpy(), o["src_pts_3d"][0].numpy(), ep["real_K"][0].numpy(),
                                          o["tem_pose"][0].numpy(), o["pred_tar_pts"][0].numpy(),
                                          o["pred_src_pts"][0].numpy(), prob=k)
            npts.append(int((o["pred_tar_pts"][0, :, 0] >= 0).sum()))
        pnp_s = time.perf_counter() - t1
    return {"value": reps / dt, "unit": "crops/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x oracle Net.forward (stage1->stage3, hyp 5) on 1 crop x {N} templates, {vit}, torch CPU fp32, "
                      f"{cores} threads (feature bank precomputed, as on the GPU); PnP/RANSAC not in `value`: cv2 is absent and "
                      f"the numpy oracle of it took {pnp_s:.2f} s for the crop's 5 problems ({sum(npts) // 5} correspondences "
                      f"each on average, 1 thread)"}


CFG_KERNEL = {0: "pp_gemm_u_kernel<128x128, 4 waves, 2-stage ring, 2 workgroups/CU>", 2: "pp_gemm_u_kernel<128x64, 4 waves, 3-stage ring, 2 workgroups/CU>",
              4: "pp_gemm_u_kernel<256x128, 8 waves, 3-stage ring>", 5: "pp_gemm_u_kernel<256x256, 8 waves, 2-stage ring>",
              6: "pp_gemm_uh_kernel (256x256, row-shared A delivery of 3x3 convolutions)"}


def latency_leg(dev, s1_mode, images=10, n_det=8, bs=4, N=162, vit="dinov2_vitl14", n_obj=2, hyp=5):
    """What the reference itself measures (run_test.py:141-216: `end = time.time()` ... `image_time = time.time() - end`, mean seconds per
    test image) at its own configuration (config/base.yaml: ViT-L/14, 162 templates, hyp 5, test batch 4): one test image = `n_det`
    detections walked in chunks of `bs` by picopose_amd.pipeline.infer_image — per chunk the gather of the detections' object banks
    (`templates_data[key][obj_idx].contiguous()`), Net.forward, PnP/RANSAC of its bs * hyp problems, D2H, hypothesis ranking on the host.
    Wall clock per image after a device synchronisation, like the reference; synthetic detections of `n_obj` objects (seg_time = 0).
    The template feature bank is computed once outside, in chunks of `bs` renders (run_test.py:120-134)."""
    from picopose_amd.picopose import Net
    from picopose_amd.pipeline import infer_image

    net = Net(make_cfg(vit))
    seeded_weights(net, 4, vit)
    net = net.to(dev).eval()
    net.match_mode = s1_mode
    tem = make_end_points(n_obj, N, dev, 300)
    templates_data = {k: v for k, v in tem.items() if k.startswith("tem_")}
    with torch.no_grad():
        templates_data["template_feature"] = torch.stack([torch.cat([net.feature_extractor(tem["tem_rgb"][o, s0:s0 + bs])[-1] for s0 in range(0, N, bs)])
                                                          for o in range(n_obj)])
    g = torch.Generator().manual_seed(7)
    times, times_la, times_one = [], [], []

    def image(i):
        det = make_end_points(n_det, 1, dev, 400 + i)
        data = {k: v[None] for k, v in det.items() if k.startswith("real_")}
        data["obj_idx"] = torch.randint(0, n_obj, (1, n_det, 1), generator=g).to(dev)
        data["score"] = torch.rand(1, n_det, generator=g).to(dev)
        return data

    datas = [image(i) for i in range(images + 3)]
    # pass 1: the reference's walk.  pass 2: the loader hands over the NEXT image too (infer_image(next_data=): its first chunk's query crops
    # ride in this image's last forward).  pass 3: the whole image as ONE chunk (test batch = its detections; results do not depend on the batch).
    for sink, chunk, look_ahead in ((times, bs, False), (times_la, bs, True), (times_one, n_det, False)):
        for i in range(images + 2):
            data = datas[i]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                preds = infer_image(net, data, templates_data, hyp=hyp, bs=chunk, next_data=datas[i + 1] if look_ahead else None)
            dt = time.perf_counter() - t0
            assert len(preds) == n_det and len(preds[0]) == hyp
            if i >= 2:            # (two warm-up images: autotuner, allocator)
                sink.append(dt * 1e3)
    times_one.sort()
    times.sort()
    times_la.sort()
    med = times[len(times) // 2]
    del net, templates_data, tem
    torch.cuda.empty_cache()
    return {"what": "wall-clock per test image as run_test.py:142-188 takes it (synchronise, then chunks of bs detections: bank gather + forward + "
                    "PnP/RANSAC + D2H + ranking), seg_time excluded",
            "config": f"config/base.yaml: {vit}, {N} templates, hyp {hyp}, test batch {bs}; {n_det} detections per image of {n_obj} objects",
            "ms_per_image": med, "ms_per_image_min": times[0], "ms_per_image_max": times[-1], "images": len(times),
            "detections_per_image": n_det, "chunk": bs, "ms_per_chunk": med / (-(-n_det // bs)), "crops_per_s": n_det / (med * 1e-3),
            "seconds_per_image": med * 1e-3,
            "ms_per_image_with_next_image_look_ahead": times_la[len(times_la) // 2],
            "ms_per_image_as_one_chunk": times_one[len(times_one) // 2],
            "one_chunk_note": f"the same images with test batch = {n_det} (all detections of an image in one forward): the reference's test batch is a "
                              "memory knob, results do not depend on it (tests/test_e2e.py batch-independence)",
            "look_ahead_note": "the same images with infer_image(next_data=): the loader's next image is known, so its first chunk's query ViT "
                               "rides in this image's last forward (the first chunk of an image otherwise has no predecessor to carry it)"}


def train_step_leg(dev, pairs=32, vit="dinov2_vitb14", steps=5):
    """SURVEY 8f row 4 in the driver's line: the FULL training step (Net.forward in train mode under autograd — key-point ground truth,
    both ViT passes, ten losses, BatchNorm on batch statistics — Loss, backward of every parameter the reference trains, SGD) at `pairs`
    real/template pairs, ViT-B/14, f16x3 engine; after the headline's timed region, `steps` steps timed one by one behind two warm-up
    steps.  grad_norm_checksum = sum over the parameter tensors of the L2 norm of their gradient after the FIRST step (float64 on the
    host): it moves when a gradient moves; tests/test_train_gpu.py compares all 338 tensors with the reference's autograd at ViT-S."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from netcfg import make_train_end_points

    from picopose_amd import ops
    from picopose_amd.picopose import Net
    from picopose_amd.utils.loss_utils import Loss
    from picopose_amd.utils.seeding import calibrated_state_dict

    old = ops.PRECISION
    ops.PRECISION = "f16x3"
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    base_bytes = torch.cuda.memory_allocated(dev)       # (the headline's model and inputs are still resident: reported memory is the step's own)
    try:
        net = Net(make_cfg(vit))
        net.load_state_dict(calibrated_state_dict(net.state_dict(), 4, vit))
        net = net.to(dev).train()
        ep = {k: v.to(dev) for k, v in make_train_end_points(pairs, 11).items()}
        np.random.seed(0)
        torch.manual_seed(0)
        opt, times, checksum, n_grads, loss0 = None, [], None, 0, None
        for i in range(steps + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tot = Loss()(net(dict(ep)))["loss"]
            tot.backward()
            if opt is None:
                grads = [p.grad for p in net.parameters() if p.grad is not None]
                checksum, n_grads, loss0 = float(sum(g.double().norm() for g in grads)), len(grads), float(tot.detach())
                opt = torch.optim.SGD([p for p in net.parameters() if p.grad is not None], lr=1e-6)
                t0 = None
            opt.step()
            opt.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            if t0 is not None and i >= 2:
                times.append((time.perf_counter() - t0) * 1e3)
        times.sort()
        return {"what": "full training step (forward_train under autograd + Loss + backward of every trained parameter + SGD), wall clock per "
                        "step with a synchronisation, after the headline's timed region",
                "config": f"{vit}, {pairs} real/template pairs, f16x3 engine, scope full", "ms_per_step": times[len(times) // 2],
                "ms_per_step_min": times[0], "steps": len(times), "pairs_per_s": pairs / (times[len(times) // 2] * 1e-3),
                "peak_memory_gib": (torch.cuda.max_memory_allocated(dev) - base_bytes) / 2 ** 30, "gradient_tensors": n_grads,
                "grad_norm_checksum": checksum, "loss_first_step": loss0}
    finally:
        ops.PRECISION = old
        torch.cuda.empty_cache()


F_CFG_KERNEL = {3: "pp_gemm_f_kernel<128x128, 4 waves, fp32 MFMA, 2 workgroups/CU>", 4: "pp_gemm_f_kernel<256x128, 8 waves, fp32 MFMA>",
                5: "pp_gemm_f_kernel<256x256, 8 waves, fp32 MFMA>", 6: "pp_gemm_f_kernel<128x64, 4 waves, fp32 MFMA, 2 workgroups/CU>",
                7: "pp_gemm_f_kernel<256x192, 8 waves, fp32 MFMA>"}


def gemm_per_kernel(L, cap=8192):
    """Per-kernel view of the GEMM launches the event pass recorded: [{kernel, cfg, conv, launches, ms, algorithmic_flops,
    useful_tflops}] — every per-kernel roofline fraction can be recomputed from it (executed MFMA flops = 3 x algorithmic in
    the f16x3 kernels, 1 x in the fp32-MFMA ones)."""
    from picopose_amd import _lib

    shape, ms, fl, cnt = (ctypes.c_int * (8 * cap))(), (ctypes.c_float * cap)(), (ctypes.c_double * cap)(), ctypes.c_int()
    by = (ctypes.c_double * cap)()
    _lib.check(L.pp_prof_gemm_records2(cap, shape, ms, fl, by, ctypes.byref(cnt)), "pp_prof_gemm_records2")
    agg = {}
    for i in range(cnt.value):
        M, N, K, ck, cfg, kind, amode = (shape[8 * i + k] for k in range(7))
        name = CFG_KERNEL.get(cfg, f"cfg {cfg}") if kind == 0 else "gemm_f16x3_kernel / gemm_kernel (operands split on the fly or fp32 MFMA)"
        key = (name, ("dense", "conv, channel-slice-major K", "conv, natural K order")[amode] if kind == 0 else ("conv" if ck else "dense"))
        # rocprof_key: what tools/profile_set.py derives from a rocprofv3 kernel name (tile + A-delivery MODE template argument), so
        # the PMC passes join this table without guessing layer shapes
        rkey = f"u{cfg}:m{amode}"
        if kind != 0 and amode >= 16:     # the fp32 engine (pp_gemm_f.hip): pp_gemm_f_kernel<FTile<BM, BN, ..>, MODE> on v_mfma_f32_32x32x2_f32
            fm = amode - 16
            name = F_CFG_KERNEL.get(cfg, f"pp_gemm_f_kernel cfg {cfg}")
            key = (name, ("dense", "conv, channel-slice-major K", "conv, natural K order")[fm])
            rkey = f"f{cfg}:m{fm}"
        elif kind != 0:     # the kernels on fp32 operands: gemm_f16x3_kernel<NJ, OCC, .> (split on the fly) / gemm_kernel<VEC4, NJ, OCC> (fp32 MFMA)
            vec4, onfly = bool(amode & 1), bool(amode & 2)
            nj_occ = ((1, 4) if cfg == 2 else (2, 2) if cfg == 0 else (2, 3)) if (onfly or vec4) else ((1, 2) if cfg == 2 else (2, 2))
            rkey = f"g{'x' if onfly else 'f'}:{nj_occ[0]}:{nj_occ[1]}"
            name = f"gemm_f16x3_kernel<{nj_occ[0]}, {nj_occ[1]}, .> (fp32 operands split on the fly)" if onfly else f"gemm_kernel<{str(vec4).lower()}, {nj_occ[0]}, {nj_occ[1]}> (v_mfma_f32_32x32x2_f32)"
            key = (name, "conv" if ck else "dense")
        a_ = agg.setdefault(key, {"kernel": name, "a_operand": key[1], "rocprof_key": rkey,
                                  "launches": 0, "ms": 0.0, "algorithmic_flops": 0.0, "algorithmic_bytes": 0.0})
        a_["launches"] += 1
        a_["ms"] += ms[i]
        a_["algorithmic_flops"] += fl[i]
        a_["algorithmic_bytes"] += by[i]
    out = sorted(agg.values(), key=lambda r: -r["ms"])
    for r in out:
        r["useful_tflops"] = r["algorithmic_flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else None
        r["algorithmic_gbs"] = r["algorithmic_bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else None
    return out


def batch_plan(workload_crops, world, scaling="weak", global_batch=None):
    """(global batch B, crops per rank, scaling) of a run on `world` ranks.  weak: every rank owns the workload's crops
    (B grows with the ranks — the 1/2/4/8-GPU throughput curve); strong: the workload's crops (or --global-batch) ARE the
    global batch and every rank owns B / world of them (BASELINE configs[3]: "batch = 32 ... sharded 8-ways")."""
    if global_batch is not None:
        scaling = "strong"
    if scaling == "strong":
        B = global_batch if global_batch is not None else workload_crops
        if B <= 0 or B % world != 0:
            raise SystemExit(f"--scaling strong: the global batch {B} is not a positive multiple of the {world} ranks")
        return B, B // world, "strong"
    return workload_crops * world, workload_crops, "weak"


def sharded_leg_plan(workload_crops, n_templates, world, rank):
    """The default N > 1 run's SHARDED leg = BASELINE configs[3] as written: the workload's crops as the GLOBAL batch, cut evenly over the
    ranks, and the template feature bank cut along the template axis (picopose_amd.dist.shard_bounds).  Returns (crops of this rank,
    first template, one past the last) or None when the ranks do not divide the batch (the leg is skipped and the line says so)."""
    from picopose_amd.dist import shard_bounds

    if world < 2 or workload_crops % world != 0 or n_templates < world:
        return None
    lo, hi = shard_bounds(n_templates, world, rank)
    return workload_crops // world, lo, hi


def input_batches(prefetch_query, single_batch, sharded, cached):
    """How many different input batches the timed loop alternates.  Two when every step looks ahead to the NEXT batch's query crops (a serving
    loop: the look-ahead must not be the crops the step is working on).  One for the template-sharded forward and for the extended template
    bank: neither has a template-side ViT pass for the next queries to ride in — and a second batch of the extended-bank workload would
    have to carry its own cache (round 6: a second batch WITHOUT one made every other step an uncached forward, 601 instead of 862 crops/s)."""
    return 2 if prefetch_query and not single_batch and not sharded and not cached else 1


def launch_ranks(n, argv):
    """`python bench.py --gpus N` from a plain shell: run the N ranks as a child torch.distributed.run job (nothing in
    this process has touched the GPU) and hand back its return code; rank 0 of the child prints the JSON line."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, cpu_cores() // n)))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="full_b32_n162_vitb", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="fast", choices=["fast", "exact", "fp16"],
                    help="fast: f16x3 networks (fp32-grade: 2 fp16 terms per operand) + fp16-MFMA stage 1 with exact re-evaluation of near-ties; "
                         "exact: fp32 MFMA everywhere; fp16: plain fp16 operands, ONE MFMA per product (BASELINE configs[4]'s arithmetic)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = the workload's crops PER RANK (global batch grows with N); strong = the workload's crops as "
                         "the GLOBAL batch, split over the ranks (BASELINE configs[3])")
    ap.add_argument("--global-batch", type=int, default=None, help="strong scaling with this global batch (implies --scaling strong)")
    ap.add_argument("--sync-loop", action="store_true", help="the host reads every step's poses before it launches the next step "
                                                             "(rounds 1-2; default now: a serving loop, one step of read latency)")
    ap.add_argument("--pnp-stream", default="side", choices=["side", "same"],
                    help="serving loop: a batch's PnP/RANSAC launch + D2H on a side stream (it waits for the batch's forward, then runs beside "
                         "the NEXT batch's forward) or on the forward's own stream (rounds 1-3)")
    ap.add_argument("--shard", default="auto", choices=["auto", "crops", "templates"],
                    help="N > 1: what is sharded over the ranks. auto = crops for weak scaling (independent replicas, no data-path "
                         "collective), templates for strong scaling (configs[3] / [4]: template-sharded bank + all-gathers)")
    ap.add_argument("--tune-file", default=None,
                    help="the GEMM autotuner's table (pp_gemm_tune_load / _save): loaded before the first step if the file exists, "
                         "written after the run otherwise — every pass of a profiling set then runs identical launches")
    ap.add_argument("--emulate-world", type=int, default=None,
                    help="8-GPU preflight on ONE GPU in ONE process: run rank 0's share of a --scaling strong job of this many ranks "
                         "(its crops, its template slice, every launch at the real shard shapes) with the collectives replaced by local "
                         "copies of the same sizes; reports phases_ms and the rank's peak memory.  Not a measurement of the exchange.")
    ap.add_argument("--single-batch", action="store_true",
                    help="every timed step runs the SAME input batch (rounds 1-5); default: two distinct synthetic batches alternate, so the look-ahead "
                         "(the next batch's query crops inside this batch's template-side ViT pass) prefetches a batch that really differs")
    ap.add_argument("--no-sharded-leg", action="store_true",
                    help="N > 1, default (weak) run: skip the extra leg that runs BASELINE configs[3] — global batch = the workload's crops, "
                         "template-sharded bank, all-gathers over RCCL — after the timed region")
    ap.add_argument("--no-prefetch-query", dest="prefetch_query", action="store_false",
                    help="the query ViT of every batch as its own pass (rounds 1-4) instead of inside the previous batch's template-side pass")
    ap.add_argument("--graph", action="store_true",
                    help="stage-1 workloads: replay the five kernels of a matching call as ONE HIP graph (picopose_amd.utils.matching.MatchingGraph) "
                         "instead of launching them one by one — measured no faster on this runtime (profiles/r05/stage1_small.txt): off by default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-leg", action="store_true", help="skip the untimed --mode exact comparison leg")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the training-step extra (default workload, fast mode only)")
    ap.add_argument("--no-latency-leg", action="store_true",
                    help="skip the per-image latency leg (the reference's own measurement: ViT-L/14, 162 templates, hyp 5, chunks of 4 detections)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("PP_BENCH_REHEARSE") != "1":
        # (counting devices does not initialise the GPU on this image; a rank that asked for a GPU the node lacks would otherwise
        # fail later with an opaque HIP error — or worse, several ranks would silently share one card)
        have = torch.cuda.device_count()
        if have < a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but this node shows {have} GPU(s).  One process per GPU is the product's "
                             f"configuration; to REHEARSE the {a.gpus}-rank code path on fewer GPUs set PP_BENCH_REHEARSE=1 (all ranks on "
                             f"cuda:0 over gloo, at most 6 ranks per card on this pool) or use --emulate-world {a.gpus} (one process).")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))      # child job; this process never touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    emulate = a.emulate_world
    if emulate is not None:
        if a.gpus != 1 or world != 1 or emulate < 2:
            raise SystemExit("--emulate-world G (G >= 2) runs in ONE process: use it with --gpus 1")
        a.scaling = "strong"
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: start the ranks with torch.distributed.run --nproc-per-node {a.gpus} "
                         f"(or run plain `python bench.py --gpus {a.gpus}`, which does that itself)")
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py needs an MI355X (rank {rank} of {world}): the HIP path has no CPU fallback")
    # PP_BENCH_REHEARSE=1: rehearse the N > 1 code path on a ONE-GPU box (every rank on cuda:0, gloo instead of RCCL)
    rehearse = os.environ.get("PP_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
        if rank == 0:
            print("bench.py: PP_BENCH_REHEARSE=1 puts every rank on cuda:0 — a rehearsal of the N > 1 code path, NOT a supported "
                  "configuration of the product (one process per GPU; DESIGN.md section 6) and not a measurement", file=sys.stderr, flush=True)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if emulate is not None:
        # rank 0 of a world of `emulate` ranks, in this one process: picopose_amd.dist sees a stand-in for torch.distributed whose
        # collectives are local copies of the same shapes (an all-gather of x = x written world times)
        import picopose_amd.dist as pdist

        class _Work:
            def wait(self):
                return None

        class _LocalDist:
            ReduceOp = torch.distributed.ReduceOp

            @staticmethod
            def get_rank(group=None):
                return 0

            @staticmethod
            def get_world_size(group=None):
                return emulate

            @staticmethod
            def is_available():
                return True

            @staticmethod
            def is_initialized():
                return True

            @staticmethod
            def all_gather_into_tensor(out, inp, group=None, async_op=False):
                out.view(emulate, *inp.shape).copy_(inp.unsqueeze(0).expand(emulate, *inp.shape))
                return _Work() if async_op else None

        pdist.dist = _LocalDist
        world = emulate
    distributed = world > 1 and emulate is None
    backend = "none"
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        backend = f"{dist.get_backend()} ({'gloo rehearsal on one GPU' if rehearse else 'RCCL over xGMI'}), world_size {dist.get_world_size()}"

    from picopose_amd import _lib, ops
    from picopose_amd.dist import shard_bounds, sharded_forward, sharded_matching_templates
    from picopose_amd.model.stage3 import COMPUTE_DEAD_LAYER1 as dead_layer1

    ops.PRECISION = {"fast": "f16x3", "exact": "f32", "fp16": "f16"}[a.mode]   # --mode exact: fp32 MFMA in every kernel
    s1_mode = "exact" if a.mode == "exact" else "fast"       # stage 1's own switch (fp16 MFMA + exact re-evaluation of near-ties | fp32 MFMA)
    from picopose_amd.utils import matching as hm

    tune = None
    if a.tune_file:
        tune = {"file": os.path.relpath(os.path.abspath(a.tune_file), ROOT), "loaded": 0}
        if os.path.exists(a.tune_file):
            tune["loaded"] = _lib.lib().pp_gemm_tune_load(os.path.abspath(a.tune_file).encode())
            if tune["loaded"] < 0:
                raise SystemExit(f"--tune-file {a.tune_file}: unreadable")
    kind, Bl, N, vit, desc = WORKLOADS[a.workload]
    C = VIT[vit][0]
    B, Bl, a.scaling = batch_plan(Bl, world, a.scaling, a.global_batch)
    if a.shard == "auto":
        a.shard = "templates" if a.scaling == "strong" else "crops"
    if a.scaling == "strong" and a.shard != "templates":
        raise SystemExit("--scaling strong shards the template bank (configs[3] / [4]): --shard templates")
    sharded = (distributed or emulate is not None) and a.shard == "templates"       # the template-sharded forward of picopose_amd/dist.py
    lo, hi = shard_bounds(N, world, rank) if sharded else (0, N)
    n_local = hi - lo
    Bq = B if sharded else Bl                               # crops whose scores this rank computes
    sd = None
    bank_dtype = torch.float16 if a.workload in FP16_BANK else torch.float32

    cached = kind == "full_cached"
    if cached:
        kind = "full"
    if kind == "stage1":
        g0 = torch.Generator(device=dev).manual_seed(0)
        query = torch.randn(Bq, C, 16, 16, device=dev, generator=g0)    # (template-sharded: identical on every rank)
        mask = disk_mask(Bq, dev)
        bank = torch.randn(Bq, n_local, C, 16, 16, device=dev, generator=torch.Generator(device=dev).manual_seed(1 + rank)).to(bank_dtype)

        graphed = hm.MatchingGraph(bank, query, mask, topk=5, mode=s1_mode) if (a.graph and not sharded) else None

        def step():
            if sharded:
                return sharded_matching_templates(bank, query, mask, N, topk=5, mode=s1_mode)
            if graphed is not None:     # the call's five launches replayed as one HIP graph (fixed buffers: a resident bank)
                return graphed()
            return hm.matching_templates(bank, query, None, mask, topk=5, mode=s1_mode)
    else:
        from picopose_amd.picopose import Net
        from picopose_amd.pipeline import pnp_for_outputs, pnp_inputs
        from picopose_amd.utils.pose_recovery import pnp_launch

        net = Net(make_cfg(vit))
        sd = seeded_weights(net, 4, vit)
        net = net.to(dev).eval()
        net.match_mode = s1_mode
        ep = make_end_points(Bl, N, dev, 100 + rank)                     # this rank's crops + their raw templates
        # a second, different batch: the timed loop alternates the two, so every step's look-ahead prefetches crops that are NOT the
        # ones it is working on (--single-batch: one batch, as in rounds 1-5)
        two_batches = input_batches(a.prefetch_query, a.single_batch, sharded, cached) == 2
        ep_b = make_end_points(Bl, N, dev, 200 + rank) if two_batches else None
        fe = net.feature_extractor
        # feature bank (outside the timed region, run_test.py:120-134): this rank's template slice of ALL crops.
        # Synthetic stand-in for the other ranks' crops: features of this rank's own renders (same shapes/bytes).
        with torch.no_grad():
            if cached:   # extended bank (single GPU): last-level features + template-side DPT maps of every template
                assert not sharded, "the extended-bank workload does not take the template-sharded bank"
                banks = [net.precompute_templates(ep["tem_rgb"][b]) for b in range(Bl)]
                feats = torch.stack([bk["feature"] for bk in banks])
                ep["template_cache"] = {"obj_index": torch.arange(Bl, device=dev),
                                        "dpt": [torch.stack([bk["dpt"][k] for bk in banks]) for k in range(3)]}
                del banks
            else:
                feats = torch.stack([torch.cat([fe(ep["tem_rgb"][b, s:min(s + 54, hi)])[-1] for s in range(lo, hi, 54)])
                                     for b in range(Bl)])
        feats = feats.to(bank_dtype)
        if sharded:
            bank = feats.repeat(world, 1, 1, 1, 1).contiguous()
        else:
            ep["template_feature"] = feats
            if ep_b is not None:
                with torch.no_grad():
                    ep_b["template_feature"] = torch.stack([torch.cat([fe(ep_b["tem_rgb"][b, s:min(s + 54, hi)])[-1] for s in range(lo, hi, 54)])
                                                            for b in range(Bl)]).to(bank_dtype)
        eps = [ep, ep_b] if ep_b is not None else [ep, ep]

        seq = [0]       # index of the next forward: batch seq % 2, look-ahead = batch (seq + 1) % 2 — EVERY call continues the alternation, so the
        #               extra (untimed, profiled) steps after the timed region find their query levels stashed like the timed ones

        def forward(mark=None, i=None):
            if sharded:
                return sharded_forward(net, ep, bank, N, hyp=5, mark=mark)
            i = seq[0] if i is None else i
            seq[0] = i + 1
            # a serving loop knows its next batch: its query crops ride in this batch's template-side ViT pass (Net.forward_test)
            cur, nxt = eps[i % 2], eps[(i + 1) % 2]
            return net(cur, 5, next_real_rgb=nxt["real_rgb"]) if a.prefetch_query else net(cur, 5)

        def step(i=None):
            i = seq[0] if i is None else i
            outs = forward(i=i)
            return outs, pnp_for_outputs(outs, eps[i % 2]["real_K"])            # PnP/RANSAC + D2H of the poses

        # the timed loop is a serving loop: step i + 1 is launched before the host reads step i's poses (asynchronous D2H into
        # two alternating pinned buffers on the same stream), so the GPU does not wait for the host between batches; every
        # step's poses are read, the last ones before the closing synchronisation
        from picopose_amd.pipeline import pnp_collect, pnp_for_outputs_async
        pinned = [torch.empty(5 * Bl + (1 if ops.SATURATION_FLAG else 0), 15, dtype=torch.float64, pin_memory=True) for _ in range(2)]   # (+ the saturation row)

        pnp_stream = torch.cuda.Stream(device=dev) if a.pnp_stream == "side" else None

        def step_launch(i):
            outs = forward(i=i)
            return pnp_for_outputs_async(outs, eps[i % 2]["real_K"], host=pinned[i % 2], stream=pnp_stream)

    if kind == "stage1":   # a step is ~1 ms: without ~0.3 s of load first, the timed steps run while the clocks still ramp
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < 0.3:
            for _ in range(20):
                out = step()
            torch.cuda.synchronize()
    for w_ in range(a.warmup):
        out = step() if kind == "stage1" else step(w_ + a.warmup % 2)     # (the warm-up ends on batch 1: the timed loop starts on batch 0 with its look-ahead in place)
    if kind != "stage1" and a.warmup == 0 and not sharded:
        seq[0] = 0
    torch.cuda.synchronize()
    L = _lib.lib()
    # stage-1 workloads: the roofline kernel IS the step, so its launch is bracketed by HIP events inside the timed steps
    # (on the launch stream); full path: the event brackets go into extra, untimed steps after the timed region
    s1_graphed = kind == "stage1" and a.graph and not sharded      # (a replayed graph does not pass the library's event hooks)
    if kind == "stage1" and not s1_graphed:
        _lib.check(L.pp_prof_enable(a.steps), "pp_prof_enable")
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    pending = None
    for i in range(a.steps):
        if kind == "stage1":
            out = step()
        elif a.sync_loop:
            out = step(i)
        else:
            h = step_launch(i)
            if pending is not None:
                poses = pnp_collect(pending, 5, Bl)       # the previous step's poses, while this step runs
            pending = h
        marks[i + 1].record()
    if pending is not None:
        poses = pnp_collect(pending, 5, Bl)               # noqa: F841
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))

    def collect_stage1(n):
        buf, cnt = (ctypes.c_float * n)(), ctypes.c_int()
        _lib.check(L.pp_prof_collect(buf, n, ctypes.byref(cnt)), "pp_prof_collect")
        _lib.check(L.pp_prof_enable(0), "pp_prof_enable")
        return sum(buf[i] for i in range(cnt.value)) / max(cnt.value, 1)

    gemm = pnp = exact = phases = sharded_leg = dead_leg = None
    if kind == "stage1" and s1_graphed:     # the kernel's own time: the same launches outside the graph, after the timed region
        n_ev = min(a.steps, 64)
        _lib.check(L.pp_prof_enable(n_ev), "pp_prof_enable")
        for _ in range(n_ev):
            hm.matching_templates(bank, query, None, mask, topk=5, mode=s1_mode)
        torch.cuda.synchronize()
        kern_ms = collect_stage1(n_ev)
    elif kind == "stage1":
        kern_ms = collect_stage1(a.steps)
    else:
        _lib.check(L.pp_prof_enable(3), "pp_prof_enable")       # 3 untimed steps with events around the stage-1 launch
        for _ in range(3):
            out = step()
        torch.cuda.synchronize()
        kern_ms = collect_stage1(3)
        # one more untimed step with an event pair around every GEMM launch, and around the PnP launch
        _lib.check(L.pp_prof_gemm_enable(8192), "pp_prof_gemm_enable")
        net.keep_stage3 = True
        fast_idx = seq[0]
        outs = forward()
        net.keep_stage3 = False
        torch.cuda.synchronize()
        per_kernel = gemm_per_kernel(L)
        g_ms, g_fl, g_n = (ctypes.c_double * 2)(), (ctypes.c_double * 2)(), (ctypes.c_int * 2)()
        _lib.check(L.pp_prof_gemm_collect(g_ms, g_fl, g_n), "pp_prof_gemm_collect")
        _lib.check(L.pp_prof_gemm_enable(0), "pp_prof_gemm_enable")
        gemm = {"ms": list(g_ms), "flops": list(g_fl), "launches": list(g_n), "per_kernel": per_kernel}
        phases = None
        if sharded:   # one more untimed step with a HIP event at every phase boundary of the sharded forward
            evs = []

            def mark(name):
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append((name, e))

            po = forward(mark)
            pnp_for_outputs(po, ep["real_K"])
            mark("pnp")
            torch.cuda.synchronize()
            d_ = {evs[i + 1][0]: evs[i][1].elapsed_time(evs[i + 1][1]) for i in range(len(evs) - 1)}
            phases = {"features": d_["features"], "exchange": d_["exchange_q"] + d_["exchange_s"], "exchange_query_wait": d_["exchange_q"],
                      "exchange_scores_topk": d_["exchange_s"], "stage1": d_["stage1"], "tail": d_["tail"], "pnp": d_["pnp"],
                      "note": "HIP events on the compute stream of rank 0 at the phase boundaries of ONE extra untimed step; features = "
                              "query ViT + query-side DPT head (the query all-gathers are in flight beside the DPT head), exchange = what the "
                              "compute stream still waits for the query / mask all-gathers + the score all-gather and top-k, stage1 = this "
                              "rank's score slices of ALL crops, tail = stages 2-3 of the own crops, pnp = batched PnP/RANSAC + D2H"}
        # The same serving loop with the DPT head's dead layer_1 branch COMPUTED (dpt.py:252-272 computes projects[0], resize_layers[0] and
        # layer1_rn and reads only the shape of the result; this build skips them — DESIGN section 4 "Dead layer"): what the step costs when
        # the reference's dead code is executed too, for a reader who wants the layer count of the reference.  Untimed leg, world 1 only.
        dead_leg = None
        if world == 1 and emulate is None and not sharded and not cached and not dead_layer1:
            from picopose_amd.model import stage3 as _st3

            _st3.COMPUTE_DEAD_LAYER1 = True
            try:
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                d_steps, pend = 6, None
                td = time.perf_counter()
                for _ in range(d_steps):
                    i_ = seq[0]
                    h_ = step_launch(i_)
                    if pend is not None:
                        pnp_collect(pend, 5, Bl)
                    pend = h_
                pnp_collect(pend, 5, Bl)
                torch.cuda.synchronize()
                d_dt = (time.perf_counter() - td) / d_steps
                dead_leg = {"value": Bl / d_dt, "ms_per_step": d_dt * 1e3, "steps": d_steps,
                            "note": "the headline loop with projects[0] / resize_layers[0] / layer1_rn of the DPT head executed (their result is unused)"}
            finally:
                _st3.COMPUTE_DEAD_LAYER1 = False
            step()      # (back to the default path before the legs below)
        sat_checked = None
        if a.mode in ("fast", "fp16"):    # one more untimed forward that verifies every operand buffer it produces (raises on saturation)
            n0, ops.CHECK_SATURATION = ops.saturation_checks, True
            forward()
            ops.CHECK_SATURATION, sat_checked = False, ops.saturation_checks - n0
        args = pnp_inputs(outs, eps[fast_idx % 2]["real_K"] if not sharded else ep["real_K"])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pnp_launch(*args)
        torch.cuda.synchronize()
        e0.record()
        pnp_launch(*args)
        e1.record()
        torch.cuda.synchronize()
        rot, tvec, ratio, ok, npts = pnp_for_outputs(outs, eps[fast_idx % 2]["real_K"] if not sharded else ep["real_K"], return_npts=True)
        pnp = {"problems": int(npts.size), "valid_points_mean": float(npts.mean()), "valid_points_min": int(npts.min()),
               "valid_points_max": int(npts.max()), "kernel_ms": e0.elapsed_time(e1), "success_rate": float(ok.mean()),
               "inliers_ratio_mean": float(ratio.mean()), "iterations": 150, "reprojection_px": 2.0,
               "note": "batched EPnP-RANSAC of all (crop, hypothesis) problems in one launch (csrc/pp_pnp.hip), inside every timed step"}
        if npts.min() < 5:
            raise SystemExit(f"bench: a PnP problem received {int(npts.min())} correspondences (< the 5-point sample): "
                             "the key-point -> PnP chain is not loaded")
        sharded_leg = None
        if distributed and not sharded and not cached and not a.no_sharded_leg:
            # BASELINE configs[3] as written, as an extra leg of the default N > 1 run: the workload's crops as the GLOBAL batch,
            # this rank's share of them, the feature bank cut along the template axis (utils/matching.py:29-69 sharded, picopose_amd/dist.py:
            # all-gathers of query features / masks and of the score slices over RCCL), stages 2-3 + PnP data-parallel on the own crops
            plan = sharded_leg_plan(WORKLOADS[a.workload][1], N, world, rank)
            if plan is None:
                sharded_leg = {"skipped": f"{world} ranks do not divide the workload's {WORKLOADS[a.workload][1]} crops"}
            else:
                bl_s, lo_s, hi_s = plan
                ep_s = {k_: v_[:bl_s] for k_, v_ in ep.items() if k_ != "template_feature"}
                # (stand-in for the other ranks' crops: this rank's own crops repeated — same shapes and bytes)
                bank_s = ep["template_feature"][:bl_s, lo_s:hi_s].repeat(world, 1, 1, 1, 1).contiguous()

                def s_forward(mark=None):
                    return sharded_forward(net, ep_s, bank_s, N, hyp=5, mark=mark)

                for _ in range(2):
                    pnp_for_outputs(s_forward(), ep_s["real_K"])
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                s_steps = max(3, min(a.steps, 10))
                ts = time.perf_counter()
                for _ in range(s_steps):
                    pnp_for_outputs(s_forward(), ep_s["real_K"])     # forward + PnP/RANSAC + D2H, the poses read every step
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
                s_dt = torch.tensor([time.perf_counter() - ts], device=dev, dtype=torch.float64)
                dist.all_reduce(s_dt, op=dist.ReduceOp.MAX)
                s_dt = float(s_dt.item()) / s_steps
                evs = []

                def s_mark(name):
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    evs.append((name, e))

                po = s_forward(s_mark)
                pnp_for_outputs(po, ep_s["real_K"])
                s_mark("pnp")
                torch.cuda.synchronize()
                d_ = {evs[j + 1][0]: evs[j][1].elapsed_time(evs[j + 1][1]) for j in range(len(evs) - 1)}
                sharded_leg = {"crops_per_s": WORKLOADS[a.workload][1] / s_dt, "ms_per_step": s_dt * 1e3, "steps": s_steps,
                               "global_batch": WORKLOADS[a.workload][1], "crops_per_rank": bl_s, "templates_of_rank0": hi_s - lo_s,
                               "world_size": world, "backend": backend,
                               "phases_ms": {"features": d_["features"], "exchange_q": d_["exchange_q"], "stage1": d_["stage1"],
                                             "exchange_s": d_["exchange_s"], "tail": d_["tail"], "pnp": d_["pnp"]},
                               "note": "configs[3]: global batch = the workload's crops, template-sharded bank + all-gathers; strong-scaling leg "
                                       "after the weak-scaling timed region; max over ranks; phases = HIP events of rank 0, one extra step"}
                del bank_s
        if a.mode == "fast" and world == 1 and not a.no_exact_leg:
            # the same step in --mode exact (fp32 MFMA everywhere, exact-fp32 stage 1), untimed leg: its rate and the
            # deviation of the default (f16x3) arithmetic from it on these very inputs
            fast = {"poses": torch.stack([o["pred_poses"] for o in outs]), "tar": torch.stack([o["pred_tar_pts"] for o in outs]),
                    "ids": torch.stack([o["tem_pose"] for o in outs]), "flow": net.last_stage3[0].clone(),
                    "cert": net.last_stage3[1].clone(), "tvec": tvec, "rot": rot}
            ops.PRECISION, net.match_mode = "f32", "exact"
            net.keep_stage3 = True
            for j_ in range(2):
                xo = step(fast_idx + 1 + j_)
            torch.cuda.synchronize()
            x_steps = 6     # (even, starting one past `fast_idx`: the last step runs the batch `fast` was computed on)
            xm = [torch.cuda.Event(enable_timing=True) for _ in range(x_steps + 1)]
            t1 = time.perf_counter()
            xm[0].record()
            for i in range(x_steps):
                xo = step(fast_idx + 1 + i)
                xm[i + 1].record()
            torch.cuda.synchronize()
            x_dt = (time.perf_counter() - t1) / x_steps
            x_ms = sorted(xm[i].elapsed_time(xm[i + 1]) for i in range(x_steps))
            x_flow, x_cert = net.last_stage3[0].clone(), net.last_stage3[1].clone()     # (of the LAST timed exact step: the batch `fast` ran on)
            # ... and with every convolution DIRECT (the reference's operation count, ops.WINOGRAD off): 2 warm-up steps, 3 timed
            xd_dt = None
            if ops.WINOGRAD:
                from picopose_amd.model import stage3 as _s3
                ops.WINOGRAD, ocf, _s3.OUT_CONV_FIRST = False, _s3.OUT_CONV_FIRST, False     # (... and the reference's order out_conv(interpolate(.)))
                try:
                    for j_ in range(2):
                        step()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for j_ in range(4):
                        step()
                    torch.cuda.synchronize()
                    xd_dt = (time.perf_counter() - t1) / 4
                finally:
                    ops.WINOGRAD, _s3.OUT_CONV_FIRST = True, ocf
            # its own roofline: HIP events around every GEMM launch of one more (untimed) exact step, fp32-MFMA peak
            _lib.check(L.pp_prof_gemm_enable(8192), "pp_prof_gemm_enable")
            forward()
            torch.cuda.synchronize()
            x_per_kernel = gemm_per_kernel(L)
            xg_ms, xg_fl, xg_n = (ctypes.c_double * 2)(), (ctypes.c_double * 2)(), (ctypes.c_int * 2)()
            _lib.check(L.pp_prof_gemm_collect(xg_ms, xg_fl, xg_n), "pp_prof_gemm_collect")
            _lib.check(L.pp_prof_gemm_enable(0), "pp_prof_gemm_enable")
            net.keep_stage3 = False
            ops.PRECISION, net.match_mode = "f16x3", s1_mode
            xouts, (xrot, xtvec, _, xok) = xo
            import numpy as np

            # stage 1 ranks 162 random templates whose scores differ in the 4th digit: the two arithmetic modes may order a
            # near-tie differently, and a different template is a different problem — compare the (crop, hypothesis) pairs
            # that chose the same template, and report how many did
            same = (torch.stack([o["tem_pose"] for o in xouts]) == fast["ids"]).flatten(2).all(-1)          # (hyp, B)
            xtar = torch.stack([o["pred_tar_pts"] for o in xouts])
            sm = same.cpu().numpy()
            flat = same.reshape(-1)
            both = ok & xok & sm
            dpose = (torch.stack([o["pred_poses"] for o in xouts]) - fast["poses"]).abs()[same]
            any_same = bool(same.any())   # (no pair picked the same template in both modes: nothing to compare, report None)
            xk_ms, xk_fl = xg_ms[0] + xg_ms[1], xg_fl[0] + xg_fl[1]
            exact = {"value": Bl / x_dt, "unit": "crops/s", "ms_per_step": x_dt * 1e3, "steps": x_steps,
                     "ms_per_step_median_hip_events": x_ms[len(x_ms) // 2],
                     "dtype": "f32 (v_mfma_f32_32x32x2_f32 in every kernel, exact-fp32 stage 1; PnP f64)",
                     "winograd": {"on": bool(ops.WINOGRAD), "min_output_pixels": int(ops.WINOGRAD_MIN_PIXELS),
                                  "note": "3x3 / stride 1 convolutions of at least min_output_pixels run as Winograd F(2x2, 3x3): fp32 transforms around 16 fp32 "
                                          "products per 2x2 output tile instead of 36 (error within 3x the direct convolution's against float64); "
                                          "PP_WINOGRAD=0 runs every convolution direct — the like-for-like operation count (227 ms, 140.7 crops/s, profiles/r05)",
                                  "direct_convolution_equivalent_tflops": Bl * full_gflop_per_crop(N, vit, cached=cached) / x_dt / 1e3,
                                  "value_with_every_convolution_direct": None if xd_dt is None else Bl / xd_dt,
                                  "ms_per_step_with_every_convolution_direct": None if xd_dt is None else xd_dt * 1e3},
                     "roofline": {"bound": "mfma", "kernel": "pp_gemm_f_kernel (fp32 engine, v_mfma_f32_32x32x2_f32) + round-1 gemm_kernel, all GEMM / conv launches of one step",
                                  "flops_note": "EXECUTED fp32 MFMA flops: a Winograd convolution counts its sixteen products, not the direct convolution it replaces",
                                  "achieved": xk_fl / (xk_ms * 1e-3) / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                                  "frac": xk_fl / (xk_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, "launches_per_step": xg_n[0] + xg_n[1],
                                  "kernel_ms_per_step": xk_ms, "algorithmic_flops_per_step": xk_fl, "per_kernel": x_per_kernel,
                                  "timing": "HIP events on the launch stream around every launch of one extra, untimed exact-mode step"},
                     "f16x3_vs_exact": {
                         "pairs": int(same.numel()), "pairs_with_same_template": int(same.sum()),
                         "pred_poses_max_abs": float(dpose.max()) if any_same else None,
                         "flow_max_abs_px": float((x_flow - fast["flow"]).abs()[flat].max()) if any_same else None,
                         "flow_max_abs_value": float(fast["flow"].abs().max()),
                         "certainty_logit_max_abs": float((x_cert - fast["cert"]).abs()[flat].max()) if any_same else None,
                         "keypoint_slots_equal": float((xtar == fast["tar"]).all(-1)[same].float().mean()) if any_same else None,
                         "pnp_translation_max_abs_m": float(abs(xtvec - tvec)[both].max()) if both.any() else None,
                         "pnp_translation_median_abs_m": float(np.median(abs(xtvec - tvec)[both])) if both.any() else None,
                         "note": "compared on the (crop, hypothesis) pairs for which both modes picked the same template"}}

    peak_mem = torch.cuda.max_memory_allocated(dev)
    if distributed:
        t = torch.tensor([dt, float(peak_mem)], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, peak_mem = float(t[0].item()), int(t[1].item())

    if rank == 0:
        ms = dt / a.steps * 1e3
        bpe = 2 if bank_dtype == torch.float16 else 4
        kbytes = stage1_bytes(Bq, n_local, C, bpe)  # bytes one launch of the stage-1 kernel streams on this rank
        achieved = kbytes / (kern_ms * 1e-3) / 1e9
        prof_dir = next((d for d in ("r03", "r02", "r01") if os.path.exists(os.path.join(ROOT, "profiles", d, "pmc_traffic_stage1.json"))), "r01")
        traffic = traffic_src = None  # HBM bytes per launch from a separate PMC pass (tools/pmc.sh): same kernel, same shape
        # the committed measurement set of this mode and workload, if any (tools/profile_set.sh: kernel trace, MFMA-busy, FETCH_SIZE and
        # WRITE_SIZE passes on identical launches — the autotuner table is pinned — joined per kernel in per_kernel.json)
        pset = pset_src = None
        for rdir in ("r06", "r05", "r04"):
            f = os.path.join(ROOT, "profiles", rdir, a.mode, "per_kernel.json")
            if pset is None and world == 1 and emulate is None and not cached and os.path.exists(f):
                cand = json.load(open(f))
                if cand.get("summary", {}).get("config", {}).get("workload", "").split(":")[0] == a.workload:
                    pset, pset_src = cand, os.path.relpath(f, ROOT)
        pmc = os.path.join(ROOT, "profiles", prof_dir, "pmc_traffic_stage1.json")
        if pset is not None and any("s1_main" in k["kernel"] for k in pset["kernels"]):
            k1 = next(k for k in pset["kernels"] if "s1_main" in k["kernel"])
            traffic, traffic_src = (k1["fetch_bytes"] + k1["write_bytes"]) / k1["launches"], pset_src
        elif world == 1 and a.mode == "fast" and (B, N, C, bpe) == (32, 162, 768, 4) and os.path.exists(pmc):
            traffic, traffic_src = json.load(open(pmc))["hbm_bytes_per_launch"], os.path.relpath(pmc, ROOT)
        line = {
            "metric": f"image-crops/sec (224x224, {N} templates)" + ("" if kind == "full" else ", stage-1 template matching only")
                      + (", extended template bank (SURVEY 8f row 1: template ViT/DPT precomputed)" if cached else ""),
            "value": (Bl if emulate is not None else B) / (dt / a.steps), "unit": "crops/s", "n_gpus": 1 if emulate is not None else world,
            "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms, "ms_per_step_median_hip_events": step_ms[len(step_ms) // 2],
            "higher_is_better": True, "scaling": a.scaling if world > 1 else "weak", "vs_baseline": None,
            "peak_memory_gib_per_rank": peak_mem / 2 ** 30,
            "dtype": {"fast": "f32 tensors; networks: f32 operands split into 2 f16 terms (22 bits) on f16 MFMA with f32 accumulate; stage-1 "
                              "contraction: f16 MFMA operands, f32 accumulate, exact f32 re-evaluation of near-ties; PnP f64",
                      "exact": "f32 (fp32 MFMA everywhere; PnP f64)",
                      "fp16": "fp16 operands (activations and weights stored f16(4 x), ONE v_mfma_f32_16x16x32_f16 per product), f32 accumulate "
                              "and f32 epilogues / residual stream; stage-1 contraction: f16 MFMA operands, exact f32 re-evaluation of near-ties; "
                              "PnP f64"}[a.mode],
            "data": "synthetic",
            "loop": ("every timed step = forward + PnP/RANSAC launch + asynchronous D2H of its poses; the host reads step i's poses after "
                     "launching step i + 1 (serving loop), the last step's before the closing synchronisation"
                     + ("; PnP + D2H of a step on a side stream (after that step's forward, beside the next step's)" if a.pnp_stream == "side" else "")
                     if kind != "stage1" and not a.sync_loop else
                     "every timed step = forward + PnP/RANSAC + D2H of its poses, read by the host before the next step is launched" if kind != "stage1" else
                     "every timed step = one matching call, results left on the device"
                     + ("; the call's five launches replayed as ONE HIP graph (MatchingGraph: fixed buffers)" if (kind == "stage1" and a.graph and not sharded) else "")),
            "config": {"workload": f"{a.workload}: {desc}", "global_batch": B, "crops_per_rank": Bl, "templates": N,
                       "templates_per_rank": n_local, "backbone": vit, "channels": C,
                       "hypotheses": 5, "mode": a.mode, "bank_dtype": "f16" if bpe == 2 else "f32",
                       "weights": "seeded random init, prediction heads calibrated (picopose_amd/utils/seeding.py)",
                       "shard": "none" if world == 1 else a.shard,
                       "prefetch_query": (bool(a.prefetch_query) and not cached) if kind != "stage1" else None,
                       "input_batches": None if kind == "stage1" else (2 if (kind != "stage1" and ep_b is not None) else 1),
                       "winograd_f4x4_heads": bool(ops.WINOGRAD4) if a.mode == "fast" else False,
                       "dpt_layer1_branch": None if kind == "stage1" else
                       ("computed (PP_DPT_DEAD_LAYER1=1)" if dead_layer1 else "not computed: only its shape is read (dpt.py:263-271)"),
                       "parallelism": "single GPU" if world == 1 else
                       (f"{a.scaling} scaling: {Bl} crops per rank x{world} (global batch {B}); feature bank template-sharded x{world} "
                        f"({n_local} of {N} templates on rank 0) + all-gathers (query features, sampled masks, scores); {backend}" if sharded else
                        f"weak scaling: {Bl} crops per rank x{world} (global batch {B}), each rank its own crops and their banks — independent "
                        f"units, no data-path collective (barrier + max-over-ranks of the timed region only); {backend}")},
        }
        s1_roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                   "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": kbytes, "traffic_measured_in_this_run": False,
                   "kernel": f"s1_main<{s1_mode}> (stage-1 fused similarity)", "traffic_source": traffic_src,
                   "timing": "HIP events on the launch stream around this launch, " +
                             ("inside the timed steps" if kind == "stage1" and not s1_graphed else
                              "the same launches outside the graph, after the timed region" if kind == "stage1" else "3 extra untimed steps after the timed region")}
        # the other roof of the same launch: the 256 x 256 x C contraction per (crop, template) on the matrix cores
        # (fast mode: one fp16 MFMA term; exact mode: fp32 MFMA).  With an fp16-stored bank the intensity is 2 C 256^2 / (2 C 256)
        # = 256 flop/B against a ridge of 2500 / 8 = 312: the kernel then sits between both roofs and this reading is the binding one.
        s1_flops = 2.0 * Bq * n_local * 256 * 256 * C
        s1_peak = MFMA_F32_PEAK_TF if a.mode == "exact" else MFMA_F16_PEAK_TF
        s1_roof["mfma"] = {"achieved": s1_flops / (kern_ms * 1e-3) / 1e12, "peak": s1_peak, "unit": "TFLOP/s",
                           "frac": s1_flops / (kern_ms * 1e-3) / 1e12 / s1_peak, "flops_per_launch": s1_flops,
                           "intensity_flop_per_byte": s1_flops / kbytes}
        k = 1 if a.mode == "exact" else 0     # fast / fp16: the pre-split kernel; exact: the fp32-MFMA kernel
        if kind == "full" and gemm and gemm["launches"][k] > 0:
            mult, peak = {"fast": (3, MFMA_F16_PEAK_TF), "fp16": (1, MFMA_F16_PEAK_TF), "exact": (1, MFMA_F32_PEAK_TF)}[a.mode]
            n, msum, fl = gemm["launches"][k], gemm["ms"][k], gemm["flops"][k]
            ach = mult * fl / (msum * 1e-3) / 1e12
            g_traffic = g_src = None   # HBM bytes of these kernels over one step, from separate PMC passes (tools/pmc_step.sh)
            pmc_step = os.path.join(ROOT, "profiles", prof_dir, "pmc_step.json")
            if pset is not None:      # FETCH + WRITE of the kernels this object describes, summed over one step
                tags = ("gemm_kernel<", "pp_gemm_f_kernel") if a.mode == "exact" else ("pp_gemm_u",)
                sel = [kk for kk in pset["kernels"] if any(t in kk["kernel"] for t in tags) and kk.get("fetch_bytes") is not None]
                if sel:
                    g_traffic, g_src = sum(kk["fetch_bytes"] + kk["write_bytes"] for kk in sel), pset_src
            elif world == 1 and a.mode == "fast" and a.workload == "full_b32_n162_vitb" and os.path.exists(pmc_step):
                g_traffic, g_src = json.load(open(pmc_step)).get("gemm_f16x3_hbm_bytes_per_step"), os.path.relpath(pmc_step, ROOT)
            pk = gemm.get("per_kernel") or []
            dom = pk[0] if pk else None
            direct_tf = Bl * full_gflop_per_crop(N, vit, cached=cached) / (dt / a.steps) / 1e3     # direct-convolution-equivalent, whole step
            # Scalars FIRST (the driver's record keeps the first ~23 keys of this object and drops unknown top-level keys): the contract's six,
            # then what a reader needs without the rest of the line; every string after them, each <= 100 characters.
            line["roofline"] = {
                "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                "traffic": None if g_traffic is None else g_traffic / n,
                "frac_algorithmic": fl / (msum * 1e-3) / 1e12 / peak,
                "dominant_ms": dom["ms"] if dom else None,
                "dominant_frac_algorithmic": dom["useful_tflops"] / peak if dom else None,
                "dominant_frac_executed": mult * dom["useful_tflops"] / peak if dom else None,
                "stage1_hbm_frac": s1_roof["frac"], "stage1_kernel_ms": kern_ms,
                # world 1: the like-for-like legs of the same run; world > 1: the template-sharded (configs[3]) leg
                **({"exact_value": None, "exact_direct_value": None, "latency_ms_per_image": None, "train_step_ms": None} if world == 1 else
                   {"sharded_crops_per_s": (sharded_leg or {}).get("crops_per_s"), "sharded_ms_per_step": (sharded_leg or {}).get("ms_per_step"),
                    "sharded_exchange_q_ms": ((sharded_leg or {}).get("phases_ms") or {}).get("exchange_q"),
                    "sharded_exchange_s_ms": ((sharded_leg or {}).get("phases_ms") or {}).get("exchange_s"),
                    "rccl_world_size": world}),
                "kernel_ms_per_step": msum, "share_of_step": msum / ms, "launches_per_step": n,
                "value_with_dead_layer1_computed": (dead_leg or {}).get("value"),
                "useful_tflops": fl / (msum * 1e-3) / 1e12, "step_direct_conv_equiv_tflops": direct_tf,
                "dominant_launches": dom["launches"] if dom else None,
                "algorithmic_flops_per_step": fl, "mfma_flops_per_step": mult * fl,
                "avg_launch_ms": msum / n,
                "traffic_bytes_per_step": g_traffic, "traffic_measured_in_this_run": False, "prefetch_query": bool(a.prefetch_query) and not cached,
                "dominant_kernel": ((dom["kernel"] + " | " + dom["a_operand"])[:100] if dom else None),
                "kernel": {"fast": "pp_gemm_u_kernel, all tiles: operands pre-split in 2 f16 terms, 3 MFMAs per product",
                           "fp16": "pp_gemm_u_kernel, all tiles: plain f16 operands, 1 MFMA per product",
                           "exact": "pp_gemm_f_kernel (fp32 operands, v_mfma_f32_32x32x2_f32) + round-1 gemm_kernel"}[a.mode],
                "frac_note": ("frac: executed MFMA flops (3 per product) / time / peak; frac_algorithmic: 2MNK / time / peak" if a.mode == "fast"
                              else "one MFMA per product: executed = algorithmic"),
                "flops_note": "2MNK of every launch as executed: a Winograd convolution counts its dense products",
                "traffic_source": g_src,
                "timing": "HIP events on the launch stream around every launch of one extra, untimed step",
                "per_kernel_note": "per kernel: ms, 2MNK, algorithmic bytes (operands + results once); profiles/r06/*/per_kernel.txt",
                "per_kernel": pk,
            }
            line["roofline_stage1"] = s1_roof
        else:
            line["roofline"] = s1_roof
        if kind == "full":
            tf = Bl * full_gflop_per_crop(N, vit, cached=cached) / (dt / a.steps) / 1e3   # useful (fp32-equivalent) TFLOP/s per GPU
            if a.mode == "fp16":
                line["mfma"] = {"bound": "mfma", "scope": "whole step, all kernels (the GEMM/conv engine is >85 % of it)",
                                "engine": "f16: plain fp16 operands, 1 x v_mfma_f32_16x16x32_f16 per product, fp32 accumulate", "achieved": tf,
                                "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s", "frac": tf / MFMA_F16_PEAK_TF,
                                "gflop_per_crop": full_gflop_per_crop(N, vit, cached=cached)}
            elif a.mode == "fast":   # every product = 3 fp16 MFMA products
                line["mfma"] = {"bound": "mfma", "scope": "whole step, all kernels (the GEMM/conv engine is >85 % of it)",
                                "engine": "f16x3: operands split into 2 fp16 terms, 3 x v_mfma_f32_16x16x32_f16, fp32 accumulate",
                                "useful_tflops": tf, "achieved": 3 * tf, "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                                "frac": 3 * tf / MFMA_F16_PEAK_TF, "frac_algorithmic": tf / MFMA_F16_PEAK_TF,
                                "frac_of_fp32_mfma_peak_equivalent": tf / MFMA_F32_PEAK_TF,
                                "gflop_per_crop": full_gflop_per_crop(N, vit, cached=cached)}
            else:
                line["mfma"] = {"bound": "mfma", "scope": "whole step, all kernels (the GEMM/conv engine is >85 % of it)",
                                "engine": "f32: v_mfma_f32_32x32x2_f32", "achieved": tf, "peak": MFMA_F32_PEAK_TF,
                                "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TF, "gflop_per_crop": full_gflop_per_crop(N, vit, cached=cached),
                                "note": "DIRECT-convolution-equivalent flops of the step: with Winograd on (ops.WINOGRAD) the large 3x3 convolutions execute "
                                        "2.25x fewer, so this fraction is not an MFMA utilisation and may exceed 1; `roofline` counts executed flops",
                                "winograd": bool(ops.WINOGRAD)}
            line["pnp"] = pnp
            if phases is not None:
                line["phases_ms"] = phases
            if sat_checked is not None:
                line["operand_range"] = {"operands_verified": sat_checked, "saturated": 0,
                                               "note": "every operand buffer of one untimed forward checked against the fp16 clamp "
                                                       "(|activation| < 16376, picopose_amd/ops.py CHECK_SATURATION); a hit aborts the bench"}
            if exact is not None:
                line["exact_mode"] = exact
                line["exact_value"] = exact["value"]                 # scalars: the strictly like-for-like (all-fp32) number
                line["exact_ms_per_step"] = exact["ms_per_step"]
                line["exact_roofline_frac"] = exact["roofline"]["frac"]
                line["exact_direct_value"] = exact["winograd"]["value_with_every_convolution_direct"]   # (the same leg with ops.WINOGRAD off)
                if "exact_value" in line["roofline"]:
                    line["roofline"]["exact_value"] = exact["value"]
                    line["roofline"]["exact_direct_value"] = line["exact_direct_value"]
            if dead_leg is not None:
                line["dead_layer1_computed"] = dead_leg
            if sharded_leg is not None:
                line["sharded_leg"] = sharded_leg
                for k_ in ("crops_per_s", "ms_per_step"):
                    line["sharded_" + k_] = sharded_leg.get(k_)
        if emulate is not None:
            line["emulated_world"] = {
                "world": emulate, "rank": 0, "crops_of_this_rank": Bl, "templates_of_this_rank": n_local, "global_batch": B,
                "note": f"8-GPU preflight: rank 0's share of `bench.py --gpus {emulate} --scaling strong` run in ONE process on ONE GPU — every "
                        "kernel launch at the real shard shapes, the collectives replaced by local copies of the same sizes (no RCCL, no "
                        "xGMI: `exchange` in phases_ms is NOT a measurement).  `value` = this ONE rank's crops per second, not the job's."}
        if tune is not None:
            tune["entries"] = L.pp_gemm_tune_entries()
            if not tune["loaded"]:
                tune["saved"] = L.pp_gemm_tune_save(os.path.abspath(a.tune_file).encode())
            tune["note"] = ("problem shape -> tile configuration of the GEMM engine; loaded: every launch of this run used the file's "
                            "choice (shapes missing from it were tuned in this run)" if tune["loaded"] else
                            "tuned in this run and written for the other passes of the measurement set")
            line["autotune"] = tune
        if world == 1 and emulate is None and kind == "full" and not a.no_latency_leg:
            ops.PRECISION = {"fast": "f16x3", "exact": "f32", "fp16": "f16"}[a.mode]
            line["latency"] = latency_leg(dev, s1_mode)
            line["latency_ms_per_image"] = line["latency"]["ms_per_image"]
            if "latency_ms_per_image" in line.get("roofline", {}):
                line["roofline"]["latency_ms_per_image"] = line["latency_ms_per_image"]
        if world == 1 and emulate is None and kind == "full" and a.workload == "full_b32_n162_vitb" and a.mode == "fast" and not a.no_train_leg:
            line["train_step"] = train_step_leg(dev)
            line["train_step_ms"] = line["train_step"]["ms_per_step"]
            if "train_step_ms" in line.get("roofline", {}):
                line["roofline"]["train_step_ms"] = line["train_step_ms"]
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_full(N, vit, sd) if kind == "full" else cpu_baseline_stage1(N, C)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
