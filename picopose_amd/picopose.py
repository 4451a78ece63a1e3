"""Top-level model — drop-in for the reference's model/picopose.py (`Net`).

Same constructor (`Net(cfg)` with cfg.stage1/2/3), same sub-module names (identical state_dict, so
`Lite.load_from_checkpoint(..., network=model)` loads the authors' checkpoint), same eval call
`model(end_points, hyp)` -> list of `hyp` dicts (model/picopose.py:97-112, 72-95), and
`model.feature_extractor(x)` usable on its own (run_test.py:130).  All arithmetic runs in
libpicopose_hip.so.  In training mode `model(end_points)` is the reference's forward_train (:114-137): it returns
`end_points` with the ten `loss*` entries, under autograd (picopose_amd/autograd.py: `loss.backward()` is the reference's training
step — SURVEY.md 8f rank 4; `Net.train_backward` narrows or switches off the graph)."""
import os

import torch
import torch.nn as nn

from . import ops
from .model.stage1 import FeatureExtractor
from .model.stage2 import AffineRegressor
from .model.stage3 import OffsetRegressor
from .utils.augment import aug_gtM_noise
from .utils.correspondence import compute_init_correspondences, compute_stage3_correspondences
from .utils.keypoints import KeypointInput, KeyPointSampler
from .utils.loss_utils import compute_stage_two_loss, flow_level_losses, infonce_rows
from .utils.matching import matching_features_similarity, matching_templates
from .utils.pose_recovery import pose_recovery_2d_prediction
from .utils.torch_utils import calc_pred_Ms


# One pass of the DPT head over [selected templates ; query crops] instead of one pass each (PP_BATCH_DPT=0: two passes, A/B)
BATCH_VIT_TRAIN = os.environ.get("PP_BATCH_VIT_TRAIN", "1") != "0"
BATCH_DPT = os.environ.get("PP_BATCH_DPT", "1") != "0"
# The NEXT batch's query ViT inside this batch's template-side ViT pass (Net.forward(..., next_real_rgb=...); PP_PREFETCH_QUERY=0: off, A/B)
PREFETCH_QUERY = os.environ.get("PP_PREFETCH_QUERY", "1") != "0"


def _tensor_key(t, fe=None):
    """Identity of a tensor's contents as far as the host can tell: storage address, shape, version counter — and, for the query stash of
    Net.forward_test, what its levels were computed WITH: the engine's arithmetic mode and the feature extractor's weights (the
    signature model/common.Packed watches; a write through `.data` needs Net.invalidate_packed(), which also drops the stash).
    The address identifies the contents only while the tensor is ALIVE (the stash keeps it), and the version counter only sees writes made
    through torch: a buffer refilled in place by a custom kernel or through DLPack must be passed as a new tensor, or refilled with
    `copy_`."""
    key = (t.data_ptr(), tuple(t.shape), t._version, ops.PRECISION)
    return key if fe is None else key + (fe._signatures()[0],)


def _image_rows(p, b0, b1):
    """Images b0 .. b1 of an NHWC map, with its attached operand form (`._hl`, model/stage3.py) cut the same way."""
    q = p[b0:b1]
    hl = getattr(p, "_hl", None)
    if hl is not None:
        hw = p.shape[1] * p.shape[2]
        s = ops.Split(hl.hl[b0 * hw:b1 * hw], hl.terms)
        s.image = (b1 - b0, p.shape[1], p.shape[2])
        q._hl = s
    return q


class Net(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.keypoint_sampler = KeyPointSampler()
        self.feature_extractor = FeatureExtractor(cfg.stage1)
        self.affine_regressor = AffineRegressor(cfg.stage2)
        self.offset_regressor = OffsetRegressor(cfg.stage3)
        self.match_mode = None  # None -> picopose_amd.utils.matching.DEFAULT_MODE ("fast")
        self.batch_hypotheses = True  # run the hyp candidates of all crops as one batch (same values, larger launches)
        # measurement aid (bench.py's f16x3-vs-exact deviation, tests): keep the last level's flow / certainty maps
        # (NHWC, hypothesis-major when batched) of the latest forward in self.last_stage3
        self.keep_stage3 = False
        self.last_stage3 = None
        self.train_backward = True   # True / "full" | "vit+stage2" | "slice1" | False: what trains under autograd (picopose_amd/autograd.py)

    def invalidate_packed(self):
        """Drop every derived copy of the weights (packed / BatchNorm-folded / pre-split) in every sub-module.  Needed only after
        a write THROUGH `param.data` (EMA / weight-surgery code): such a write moves neither the parameter's version counter nor
        its address, which is what the caches watch (model/common.py Packed._signatures)."""
        from .model.common import Packed

        for m in self.modules():
            if isinstance(m, Packed):
                m.invalidate_packed()
        self._query_stash = None

    # model/picopose.py:52-70 — pick hypothesis k's template for every crop (pure indexing)
    def select_template_data(self, end_points, pred_id_src, k):
        idx = pred_id_src[:, k]
        rows = torch.arange(idx.shape[0], device=idx.device)
        sel = {key: end_points[key][rows, idx] for key in ("tem_pose", "tem_K", "tem_M", "tem_mask", "tem_rgb", "tem_pts3d")}
        for key in ("real_pts2d", "real_K", "real_M", "real_mask", "real_pose"):
            sel[key] = end_points[key]
        return sel

    # ---- template bank extended to the template-side DPT outputs ---------------------------
    def precompute_templates(self, tem_rgb, chunk=54):
        """The bank precompute of run_test.py:120-134 for one object, extended: tem_rgb (N,3,224,224) ->
        {"feature": (N,C,16,16) — what the reference keeps — and "dpt": [path_4 (N,16,16,256), path_3 (N,32,32,256),
        path_2 (N,64,64,256)] NHWC}.  Eval mode is deterministic and every op is per sample, so a forward that reads
        these maps (end_points["template_cache"]) returns exactly what recomputing the selected templates' ViT and DPT
        head returns; it trades 5.5 MB of HBM per template for 5 of the 6 ViT forwards and 5 of the 6 DPT calls."""
        fe, dpt = self.feature_extractor, self.offset_regressor.dpt_head
        feats, maps = [], [[], [], []]
        with torch.no_grad():
            for s0 in range(0, tem_rgb.shape[0], chunk):
                toks, (h0, w0) = fe.forward_tokens(tem_rgb[s0:s0 + chunk])
                feats.append(ops.tokens_to_nchw(toks[-1], 1, h0, w0))
                for k, m in enumerate(dpt.forward_nhwc([t[:, 1:].unflatten(1, (h0, w0)) for t in toks])):
                    maps[k].append(m)
        return {"feature": torch.cat(feats), "dpt": [torch.cat(m) for m in maps]}

    # model/picopose.py:72-95
    def forward_test_hyp(self, end_points, real, tem_cached=None):
        """`real` = (query token maps at the 4 taken blocks, (h0, w0), query-side DPT maps or None).
        `tem_cached` = (selected templates' last-level features (B,C,16,16), their DPT maps) from the extended bank
        instead of recomputing the template ViT + DPT head (precompute_templates).

        The reference recomputes dpt_head(features_real) for every hypothesis (offset_regressor.py:17 inside the
        loop of picopose.py:107); the query features do not depend on the hypothesis, so forward_test computes
        them once — bit-identical outputs (eval mode is deterministic), 4 of the 10 DPT calls saved."""
        fe = self.feature_extractor
        real_tok, (h0, w0) = real[0], real[1]
        real_dpt = real[2] if len(real) > 2 else None
        levels = real[3] if len(real) > 3 else None     # shared level buffers: templates first, then the query crops (forward_test)
        output = {"tem_pose": end_points["tem_pose"]}
        output["tar_pts_2d"] = end_points["real_pts2d"].permute(0, 3, 2, 1)
        output["src_pts_3d"] = end_points["tem_pts3d"].permute(0, 3, 1, 2)
        # stage 1: template features
        pre = real[4] if len(real) > 4 else None        # (level buffers with room in front, the NEXT batch's query crops): forward_test
        if tem_cached is None and pre is not None and levels is not None:
            # the next batch's query ViT rides in this batch's template pass: rows [next queries ; templates] of one launch per layer
            # (a GEMM row does not depend on the other rows: every token keeps its bits); its four levels stay behind as the stash
            alloc, nxt = pre
            Bn, T = nxt.shape[0], h0 * w0 + 1
            both, _ = fe.forward_tokens(torch.cat([nxt, end_points["tem_rgb"]]), level_out=(alloc, 0))
            tem_tok = [t[Bn:] for t in both]
            # the stash: the key of the crops it was computed FROM, copies of its Bn * T rows per level (views would pin the whole `alloc` of
            # this batch until the next call), and `nxt` itself — held so that its storage cannot be freed and its address handed to another
            # batch of the same shape, whose key would then match
            self._query_stash = (_tensor_key(nxt, fe), [a[:Bn * T].clone() for a in alloc], (h0, w0), nxt)
            tem_last = ops.tokens_to_nchw(tem_tok[-1], 1, h0, w0)
        elif tem_cached is None:
            tem_tok, _ = fe.forward_tokens(end_points["tem_rgb"], level_out=None if levels is None else (levels, 0))
            tem_last = ops.tokens_to_nchw(tem_tok[-1], 1, h0, w0)
        else:
            tem_last, tem_dpt = tem_cached
        # stage 2
        sim = matching_features_similarity(tem_last, ops.tokens_to_nchw(real_tok[-1], 1, h0, w0),
                                           end_points["tem_mask"], end_points["real_mask"])
        pred_translation, pred_scale, pred_inplane = self.affine_regressor(sim)
        pred_Ms = calc_pred_Ms(pred_scale, pred_inplane, pred_translation, end_points["tem_pose"], end_points["tem_K"],
                               end_points["tem_M"])
        output["pred_poses"] = pose_recovery_2d_prediction(end_points["real_M"], end_points["real_K"], pred_Ms,
                                                           end_points["tem_K"], end_points["tem_M"], end_points["tem_pose"])
        # stage 3 (NHWC inside; the token maps are read in place, cls row skipped)
        init_flow, init_certainty = compute_init_correspondences(pred_Ms, end_points["tem_mask"])
        as_img = lambda t: t[:, 1:].unflatten(1, (h0, w0))  # noqa: E731  (B,h0,w0,C) view, batch stride (1+h0*w0)*C
        orr = self.offset_regressor
        if levels is not None and real_dpt is None and tem_cached is None:
            # ONE pass of the DPT head over [templates ; query crops] (the same weights; a row of a GEMM does not depend on the
            # other rows, so every map keeps the bits two passes give it): larger launches, half as many of them
            T = h0 * w0 + 1
            nT, nR = tem_tok[0].shape[0], levels[0].shape[0] // T - tem_tok[0].shape[0]
            both = orr.dpt_head.forward_nhwc([as_img(lv.view(-1, T, lv.shape[1])) for lv in levels])
            tem_dpt, real_dpt = [_image_rows(p_, 0, nT) for p_ in both], [_image_rows(p_, nT, nT + nR) for p_ in both]
        else:
            if real_dpt is None:
                real_dpt = orr.dpt_head.forward_nhwc([as_img(t) for t in real_tok])
            if tem_cached is None:
                tem_dpt = orr.dpt_head.forward_nhwc([as_img(t) for t in tem_tok])
        flows, certs = orr.flow_decoder.forward_nhwc(tem_dpt, real_dpt, ops.to_nhwc(init_flow), ops.to_nhwc(init_certainty))
        if self.keep_stage3:
            self.last_stage3 = (flows[-1], certs[-1])
        output["pred_tar_pts"], output["pred_src_pts"] = compute_stage3_correspondences(
            ops.to_nchw(flows[-1]), ops.to_nchw(certs[-1]), threshold=0.5)
        return output

    def forward_hypotheses(self, end_points, pred_id_src, real):
        """The loop of model/picopose.py:106-108 over the top-k templates of every crop -> list of k output dicts."""
        hyp = pred_id_src.shape[1]
        cache = end_points.get("template_cache")  # {"obj_index": (B,) long, "dpt": 3 x (O,N,h,w,256)} (precompute_templates)

        def cached(rows, idx):
            if cache is None:
                return None
            obj = cache["obj_index"][rows]
            return end_points["template_feature"][rows, idx], [m[obj, idx] for m in cache["dpt"]]

        if not self.batch_hypotheses:
            rows = torch.arange(pred_id_src.shape[0], device=pred_id_src.device)
            return [self.forward_test_hyp(self.select_template_data(end_points, pred_id_src, k), real,
                                          cached(rows, pred_id_src[:, k])) for k in range(hyp)]
        # All hypotheses as ONE batch of hyp*B samples (hypothesis-major).  Every op on this path is per-sample
        # (eval-mode BatchNorm is folded, GroupNorm is per sample) and a GEMM row does not depend on the other
        # rows, so each hypothesis gets exactly the values the per-hypothesis loop gives it — with 5x larger
        # launches that fill the 256 CUs evenly.
        real_tok, hw = real[0], real[1]
        real_dpt = real[2] if len(real) > 2 and real[2] is not None else None
        levels = real[3] if len(real) > 3 else None
        B = pred_id_src.shape[0]
        idx = pred_id_src.t().reshape(-1)
        rows = torch.arange(B, device=idx.device).repeat(hyp)
        rep = lambda t: t.repeat(hyp, *([1] * (t.dim() - 1)))  # noqa: E731
        N = end_points["tem_pose"].shape[1]
        flat = rows * N + idx                                            # row of (b, template) in the (B*N, ...) view
        sel = {key: (ops.gather_rows(end_points[key].flatten(0, 1), flat) if end_points[key].is_contiguous()
                     else end_points[key][rows, idx])
               for key in ("tem_pose", "tem_K", "tem_M", "tem_mask", "tem_rgb", "tem_pts3d")}
        for key in ("real_pts2d", "real_K", "real_M", "real_mask", "real_pose"):
            sel[key] = rep(end_points[key])
        # (with the query-side DPT maps at hand only the last token level is read again: for the stage-2 similarity)
        toks = ([rep(t) for t in real_tok] if real_dpt is None and levels is None
                else [None] * (len(real_tok) - 1) + [rep(real_tok[-1])])
        # (the query-side DPT maps go in un-repeated: the flow decoder projects them once and tiles the projection)
        out = self.forward_test_hyp(sel, (toks, hw, real_dpt, levels if cache is None else None, real[4] if len(real) > 4 and cache is None else None),
                                    cached(rows, idx))
        return [{key: v[k * B:(k + 1) * B] for key, v in out.items()} for k in range(hyp)]

    # model/picopose.py:97-112
    def forward_test(self, end_points, hyp=5, next_real_rgb=None):
        """next_real_rgb: the query crops of the batch this forward will be called with NEXT (a serving loop, the mini-batches of a test
        image: picopose_amd/pipeline.infer_image).  Their ViT pass does not depend on this batch, so it runs as extra rows of this batch's
        template-side ViT launches ((hyp + 1) B images per launch instead of hyp B and, separately, B at a fifth of the rows and two thirds
        of the tile fill), and the next call finds its query levels stashed — same bits with or without (tests/test_e2e.py)."""
        with torch.no_grad():
            fe = self.feature_extractor
            levels = None
            ops.saturation_word(end_points["real_rgb"].device)      # (registered before the first producer kernel of this device runs)
            stash, self._query_stash = getattr(self, "_query_stash", None), None
            if BATCH_DPT and self.batch_hypotheses and end_points.get("template_cache") is None:
                # the four feature levels of the hyp * B selected templates and of the B query crops share one buffer per level
                # (templates first): the DPT head then runs ONCE over both (forward_test_hyp)
                B, _, H, W = end_points["real_rgb"].shape
                T = (H // fe.patch_size) * (W // fe.patch_size) + 1
                pre_ok = PREFETCH_QUERY and next_real_rgb is not None and tuple(next_real_rgb.shape[1:]) == (3, H, W) and next_real_rgb.is_contiguous()
                Bn = next_real_rgb.shape[0] if pre_ok else 0
                alloc = [torch.empty((Bn + (hyp + 1) * B) * T, fe.num_features, dtype=torch.float32, device=end_points["real_rgb"].device)
                         for _ in fe.blocks_to_take]
                levels = [a_[Bn * T:] for a_ in alloc]           # [templates ; query crops]; the Bn images in front: the next batch's queries
                if stash is not None and stash[0] == _tensor_key(end_points["real_rgb"], fe) and stash[1][0].shape[0] == B * T:
                    (h0, w0) = stash[2]
                    for lv, st in zip(levels, stash[1]):        # computed inside the previous batch's template pass
                        lv[hyp * B * T:].copy_(st)
                    real_tok = [lv[hyp * B * T:].view(B, T, -1) for lv in levels]
                else:
                    real_tok, (h0, w0) = fe.forward_tokens(end_points["real_rgb"], level_out=(levels, hyp * B))
                real = (real_tok, (h0, w0), None, levels, (alloc, next_real_rgb) if Bn else None)
            else:
                real_tok, (h0, w0) = fe.forward_tokens(end_points["real_rgb"])
                real_dpt = self.offset_regressor.dpt_head.forward_nhwc([t[:, 1:].unflatten(1, (h0, w0)) for t in real_tok])
                real = (real_tok, (h0, w0), real_dpt)
            # matching.py normalises the bank itself; the reference's extra F.normalize of the whole bank
            # (picopose.py:99) is idempotent up to rounding and is not materialised here
            pred_score_src, pred_id_src = matching_templates(
                end_points["template_feature"], ops.tokens_to_nchw(real_tok[-1], 1, h0, w0), end_points["tem_mask"],
                end_points["real_mask"], topk=hyp, mode=self.match_mode)
            return self.forward_hypotheses(end_points, pred_id_src, real)

    # model/picopose.py:29-50
    def compute_keypoint_data(self, end_points):
        rel_pose = end_points["tem_pose"] @ torch.inverse(end_points["real_pose"])
        tar = KeypointInput(full_depth=end_points["real_full_depth"], K=end_points["real_K"], M=end_points["real_M"], mask=end_points["real_mask"])
        src = KeypointInput(full_depth=end_points["tem_full_depth"], K=end_points["tem_K"], M=end_points["tem_M"], mask=end_points["tem_mask"])
        return self.keypoint_sampler.sample_pts(tar_data=tar, src_data=src, T_src2target=torch.inverse(rel_pose), T_tar2source=rel_pose)

    # model/picopose.py:114-137
    def forward_train(self, end_points, pred_Ms=None):
        """The training forward: key-point ground truth, both ViT passes, the InfoNCE / stage-2 / flow + certainty losses,
        BatchNorm layers on batch statistics (running buffers updated).  Returns `end_points` with the `loss*` entries the
        reference adds (utils/loss_utils.Loss sums them).  With autograd enabled the backward of picopose_amd/autograd.py is live,
        scope by `self.train_backward`: True / "full" (default) — all ten losses carry a graph: `Loss()(end_points)["loss"].
        backward()` is the reference's training step (every parameter the reference trains receives its gradient: the ViT, the
        affine regressor, the DPT head, the flow decoder); "vit+stage2" — only `loss_info` and the three stage-2 losses (through
        the similarity volume, every ViT block, the embeddings, the affine regressor), stage 3 forward-only; "slice1" — the first
        slice (affine regressor from the stage-2 losses, last ViT block from InfoNCE); False — forward values only.  pred_Ms: the noisy ground-truth affines of stage 3;
        drawn by utils/augment.aug_gtM_noise when not given (tests pass the ones a reference run drew)."""
        from . import autograd as ag
        from .utils.loss_utils import infonce_index_rows

        live = torch.is_grad_enabled() and bool(self.train_backward)
        wide = live and self.train_backward != "slice1"          # the whole ViT + stage 2 from the stage-1/2 losses
        full = wide and self.train_backward != "vit+stage2"      # True / "full": stage 3 under autograd too — the reference's training step
        fe, orr = self.feature_extractor, self.offset_regressor
        with torch.no_grad():
            kp = self.compute_keypoint_data(end_points)
        with torch.set_grad_enabled(live):
            kw = dict(last_block_fn=ag.last_block_forward if live else None, all_blocks=wide, embed_fn=ag.embed_tokens if wide else None)
            if BATCH_VIT_TRAIN and end_points["real_rgb"].shape == end_points["tem_rgb"].shape:
                # ONE pass over [real ; template] crops: the ViT has no batch statistics and a GEMM row does not depend on the other rows,
                # so every token keeps its bits; twice the rows per launch fill the engine's tile rounds better (8 224 rows are 1.55
                # rounds of 256 x 256 tiles on the N = 3072 layers, 16 448 are 3.05) and the weight gradients are one product each
                nb = end_points["real_rgb"].shape[0]
                both, (h0, w0) = fe.forward_tokens(torch.cat([end_points["real_rgb"], end_points["tem_rgb"]]), **kw)
                real_tok, tem_tok = [t[:nb] for t in both], [t[nb:] for t in both]
            else:
                real_tok, (h0, w0) = fe.forward_tokens(end_points["real_rgb"], **kw)
                tem_tok, _ = fe.forward_tokens(end_points["tem_rgb"], **kw)
            if live:
                s_rows, t_rows = infonce_index_rows(tem_tok[-1].shape, kp["src_pts"], kp["tar_pts"])
                end_points["loss_info"] = (ag.infonce(tem_tok[-1], real_tok[-1], s_rows, t_rows) if s_rows.numel() else
                                           torch.full((), float("nan"), device=s_rows.device))
            else:
                end_points["loss_info"] = infonce_rows(tem_tok[-1], real_tok[-1], kp["src_pts"], kp["tar_pts"])
            if wide:
                sim = ag.similarity_volume(tem_tok[-1], real_tok[-1], end_points["tem_mask"])
            else:
                with torch.no_grad():
                    sim = matching_features_similarity(ops.tokens_to_nchw(tem_tok[-1].detach(), 1, h0, w0), ops.tokens_to_nchw(real_tok[-1].detach(), 1, h0, w0),
                                                       end_points["tem_mask"], end_points["real_mask"])
            if not full:
                real_tok = [t.detach() for t in real_tok]        # stage 3 forward-only: its inputs carry no graph
                tem_tok = [t.detach() for t in tem_tok]
            if live:
                pred_translation, pred_scale, pred_inplane = ag.affine_regressor_forward(self.affine_regressor, sim)
            else:
                pred_translation, pred_scale, pred_inplane = self.affine_regressor(sim)
            end_points["loss_2d_trans"], end_points["loss_scale"], end_points["loss_inplane"] = compute_stage_two_loss(
                end_points, pred_translation, pred_scale, pred_inplane)
        with torch.no_grad():
            if pred_Ms is None:
                pred_Ms = aug_gtM_noise(end_points)
            init_flow, init_certainty = compute_init_correspondences(pred_Ms, end_points["tem_mask"])
        as_img = lambda t: t[:, 1:].unflatten(1, (h0, w0))  # noqa: E731
        if full:
            flows, certs = ag.offset_regressor_forward(orr, [as_img(t) for t in tem_tok], [as_img(t) for t in real_tok],
                                                       ops.to_nhwc(init_flow), ops.to_nhwc(init_certainty))
            for idx, (fl, ce) in enumerate(zip(flows, certs)):
                end_points[f"loss_flow{idx}"], end_points[f"loss_certainty{idx}"] = ag.flow_level_losses(fl, ce, kp["tar_pts"])
            if self.keep_stage3:
                self.last_stage3 = (flows[-1].detach(), certs[-1].detach())
            return end_points
        with torch.no_grad():
            flows, certs = orr.forward_nhwc([as_img(t) for t in tem_tok], [as_img(t) for t in real_tok], ops.to_nhwc(init_flow),
                                            ops.to_nhwc(init_certainty), train=True)
            for idx, (fl, ce) in enumerate(zip(flows, certs)):
                end_points[f"loss_flow{idx}"], end_points[f"loss_certainty{idx}"] = flow_level_losses(fl, ce, kp["tar_pts"])
            if self.keep_stage3:
                self.last_stage3 = (flows[-1], certs[-1])
        return end_points

    def forward(self, end_points, hyp=5, next_real_rgb=None):
        if self.training:
            return self.forward_train(end_points)
        return self.forward_test(end_points, hyp, next_real_rgb=next_real_rgb)
