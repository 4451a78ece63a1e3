"""Per-batch inference as the reference's evaluator runs it (run_test.py:151-186): network forward, then
PnP/RANSAC for every (instance, hypothesis), hypotheses ranked by inlier ratio, stage-2 pose as the
fallback when PnP fails.  One batched PnP launch and one device->host copy per batch instead of the
reference's B*hyp host round trips."""
import numpy as np
import torch

from .utils.pose_recovery import pose_recovery_ransac_pnp_batched


def pnp_inputs(outputs, real_K):
    """The (hyp*B) PnP problems of a forward, hypothesis-major: arguments of pose_recovery_ransac_pnp_batched."""
    hyp = len(outputs)
    cat = lambda k: torch.cat([o[k] for o in outputs], dim=0)  # noqa: E731
    return (cat("tar_pts_2d"), cat("src_pts_3d"), real_K.repeat(hyp, 1, 1), cat("tem_pose"), cat("pred_tar_pts"),
            cat("pred_src_pts"))


def pnp_for_outputs(outputs, real_K, return_npts=False):
    """outputs: list (hyp) of Net.forward dicts; real_K (B,3,3) -> rot (hyp,B,3,3), tvec (hyp,B,3,1), ratio, ok
    [+ npts (hyp,B): correspondences each problem received]."""
    hyp, B = len(outputs), outputs[0]["pred_poses"].shape[0]
    res = pose_recovery_ransac_pnp_batched(*pnp_inputs(outputs, real_K), return_npts=return_npts)
    rot, tvec, ratio, ok = res[:4]
    out = (rot.reshape(hyp, B, 3, 3), tvec.reshape(hyp, B, 3, 1), ratio.reshape(hyp, B), ok.reshape(hyp, B))
    return out + (res[4].reshape(hyp, B),) if return_npts else out


def pnp_for_outputs_async(outputs, real_K, host=None, stream=None):
    """pnp_for_outputs without the host wait -> handle; `pnp_collect(handle, hyp, B)` reads it (one batch later in a serving loop).
    stream: run the PnP launch and the copy on this side stream, beside the next batch's forward (pose_recovery_ransac_pnp_batched_async)."""
    from .utils.pose_recovery import pose_recovery_ransac_pnp_batched_async

    return pose_recovery_ransac_pnp_batched_async(*pnp_inputs(outputs, real_K), host=host, stream=stream)


def pnp_collect(handle, hyp, B):
    rot, tvec, ratio, ok = handle.result()
    return rot.reshape(hyp, B, 3, 3), tvec.reshape(hyp, B, 3, 1), ratio.reshape(hyp, B), ok.reshape(hyp, B)


def _rank_hypotheses(stage2, rot, tvec, ratio, ok, hyp):
    """run_test.py:168-186 for one mini-batch: per instance the hypotheses sorted by inlier ratio, stage-2 pose where PnP failed."""
    B = stage2.shape[1]
    results = []
    for b in range(B):
        hyps = []
        for k in range(hyp):
            if ok[k, b]:
                hyps.append(dict(R=rot[k, b], t=tvec[k, b, :, 0], inliers_ratio=float(ratio[k, b]), pnp_success=True))
            else:  # run_test.py:177-179: fall back to the stage-2 pose — float32 as the network returned it (the csv row of
                # such an instance prints float32 values), with the inlier ratio PnP reported
                hyps.append(dict(R=stage2[k, b, :3, :3], t=stage2[k, b, :3, 3], inliers_ratio=float(ratio[k, b]), pnp_success=False))
        hyps.sort(key=lambda h: h["inliers_ratio"], reverse=True)                   # run_test.py:186 (stable, like sorted())
        results.append(hyps)
    return results


def infer_batch(net, end_points, hyp=5, pnp_fn=None):
    """-> per-instance pose hypotheses sorted by inlier ratio (run_test.py:168-186):
    list over instances of list over hypotheses of dict(R (3,3), t (3,), inliers_ratio, pnp_success).
    pnp_fn(outputs, real_K) -> (rot (hyp,B,3,3), tvec (hyp,B,3,1), ratio (hyp,B), ok (hyp,B)) replaces the batched HIP
    PnP (tests of the loop semantics inject canned answers)."""
    outputs = net(end_points, hyp)
    rot, tvec, ratio, ok = (pnp_fn or pnp_for_outputs)(outputs, end_points["real_K"])
    stage2 = np.stack([o["pred_poses"].cpu().numpy() for o in outputs])            # (hyp,B,4,4) float32
    return _rank_hypotheses(stage2, rot, tvec, ratio, ok, hyp)


def infer_image(net, data, templates_data, hyp=5, bs=16, pnp_fn=None, pipelined=True, next_data=None):
    """One test image exactly as run_test.py:141-188 walks it: `data` holds the image's instances on dim 1
    (data[key][0] = (n_instance, ...), plus 'obj_idx'), `templates_data[key]` the per-object template bank
    ((n_objects, N, ...), including 'template_feature' and, optionally, an extended bank under 'template_cache').
    Instances are processed in mini-batches of `bs`; returns preds_image: per instance the hypotheses sorted by
    inlier ratio, each {'R_stage_3' (9,), 't_stage_3' (3,) in mm, 'inliers_ratio'} (run_test.py:181-186).
    pipelined (default; HIP PnP only): the mini-batches of an image are independent, so mini-batch j + 1's forward is launched
    BEFORE the host waits for mini-batch j's PnP results (its PnP launch + device->host copy are already enqueued behind its forward) —
    the card does not idle while the host ranks hypotheses; same results, same order.  pipelined=False: the reference's strictly
    sequential walk (every mini-batch ends with a host wait).
    next_data: the NEXT test image's `data` (an evaluator's loader has it one iteration ahead): the query crops of its first mini-batch
    ride in this image's last forward, as the mini-batches of one image do among themselves — same results."""
    n_instance = data["score"].shape[1]
    preds_image = []

    def inputs_of(start, end):
        obj_idx = data["obj_idx"][0][start:end].reshape(-1)
        inputs = {k: v[0][start:end].contiguous() for k, v in data.items() if v[0].dim() > 0}
        for k, v in templates_data.items():
            if k == "template_cache":   # extended bank: maps stay per object, instances carry their object index
                inputs[k] = {"obj_index": obj_idx, "dpt": v["dpt"]}
            else:
                inputs[k] = v[obj_idx].contiguous()
        return inputs

    def emit(batch_results):
        for hyps in batch_results:
            preds_image.append([{"R_stage_3": np.asarray(h["R"]).reshape(9), "t_stage_3": np.asarray(h["t"]).reshape(3) * 1000,
                                 "inliers_ratio": h["inliers_ratio"]} for h in hyps])

    if pnp_fn is not None or not pipelined:
        for start in range(0, n_instance, bs):
            emit(infer_batch(net, inputs_of(start, min(start + bs, n_instance)), hyp, pnp_fn=pnp_fn))
        return preds_image
    pending = None      # (PnP handle, pinned stage-2 poses, their event, batch size) of the mini-batch in flight
    starts = list(range(0, n_instance, bs))
    for j, start in enumerate(starts):
        inputs = inputs_of(start, min(start + bs, n_instance))
        # the next mini-batch's query crops ride in this one's template-side ViT pass (Net.forward_test): same bits, fuller launches
        if j + 1 < len(starts):
            nxt = data["real_rgb"][0][starts[j + 1]:min(starts[j + 1] + bs, n_instance)].contiguous()
        elif next_data is not None and next_data["score"].shape[1] > 0:
            nxt = next_data["real_rgb"][0][0:min(bs, next_data["score"].shape[1])].contiguous()
        else:
            nxt = None
        outputs = net(inputs, hyp, next_real_rgb=nxt) if nxt is not None else net(inputs, hyp)
        handle = pnp_for_outputs_async(outputs, inputs["real_K"])
        s2 = torch.stack([o["pred_poses"] for o in outputs])                        # (hyp,B,4,4) float32
        s2_host = torch.empty(s2.shape, dtype=s2.dtype, pin_memory=True)
        s2_host.copy_(s2, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        if pending is not None:
            emit(_collect(pending, hyp))
        pending = (handle, s2_host, ev, s2.shape[1])
    if pending is not None:
        emit(_collect(pending, hyp))
    return preds_image


def _collect(pending, hyp):
    handle, s2_host, ev, B = pending
    rot, tvec, ratio, ok = pnp_collect(handle, hyp, B)
    ev.synchronize()
    return _rank_hypotheses(s2_host.numpy(), rot, tvec, ratio, ok, hyp)


def bop_csv_lines(scene_id, img_id, obj_ids, scores, preds_image, image_time):
    """The BOP results rows of run_test.py:191-206: one line per instance, best hypothesis, t in millimetres."""
    lines = []
    for k, preds in enumerate(preds_image):
        lines.append(",".join((str(scene_id), str(img_id), str(obj_ids[k]), str(scores[k]),
                               " ".join(str(v) for v in preds[0]["R_stage_3"]),
                               " ".join(str(v) for v in preds[0]["t_stage_3"]), f"{image_time}\n")))
    return lines
