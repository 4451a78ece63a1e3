"""Oracle for the networks of the path (TEST INFRASTRUCTURE only): functional, plain-torch CPU
restatements driven by a state_dict with the reference's names.

  vit_features        model/stage1/feature_extractor.py:93-109 over vision_transformer.py:179-228 and
                      layers/{patch_embed.py:69-82, block.py:82-107, attention.py:49-62, mlp.py:35-41,
                      layer_scale.py:27-28}
  affine_regressor    model/stage2/affine_regressor.py:72-84
  dpt_head            model/stage3/dpt.py:252-272 (FeatureFusionBlock :129-156, ResidualConvUnit :72-95)
  flow_decoder        model/stage3/flow_decoder.py:74-94 (forward_flow :58-72, feature_sample :49-56) with
                      raft_decoder.py:30-53 (CorrelationPyramid), :147-161 (MotionEncoder), :287-289 (XHead)
                      and utils/corr_lookup.py:100-134 (CorrLookup)
Pinned by tests/golden/nets_*.npz (outputs of the reference modules, oracle/gen_golden.py).
"""
import math

import torch
import torch.nn.functional as F


def _g(sd, prefix, name):
    return sd[prefix + name]


# ------------------------------------------------------------------------------------------ stage 1
def _pos_embed(pos, w0, h0, offset=0.1):
    N = pos.shape[1] - 1
    if w0 * h0 == N:
        return pos
    sq = int(math.sqrt(N))
    dim = pos.shape[-1]
    sx, sy = float(w0 + offset) / math.sqrt(N), float(h0 + offset) / math.sqrt(N)
    grid = F.interpolate(pos[:, 1:].reshape(1, sq, sq, dim).permute(0, 3, 1, 2), scale_factor=(sx, sy), mode="bicubic")
    return torch.cat([pos[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)], dim=1)


def vit_features(sd, x, heads, blocks_to_take, prefix="feature_extractor.dinov2.", patch=14):
    B, _, H, W = x.shape
    h0, w0 = H // patch, W // patch
    t = F.conv2d(x, _g(sd, prefix, "patch_embed.proj.weight"), _g(sd, prefix, "patch_embed.proj.bias"), stride=patch)
    C = t.shape[1]
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat([_g(sd, prefix, "cls_token").expand(B, -1, -1), t], dim=1)
    t = t + _pos_embed(_g(sd, prefix, "pos_embed"), h0, w0)
    hd = C // heads
    depth = 1 + max(int(k[len(prefix) + 7:].split(".")[0]) for k in sd if k.startswith(prefix + "blocks."))
    outs = []
    for i in range(depth):
        p = f"{prefix}blocks.{i}."
        h = F.layer_norm(t, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
        qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, -1, 3, heads, hd)
        q, k, v = (qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))
        a = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
        o = (a @ v).transpose(1, 2).reshape(B, -1, C)
        t = t + sd[p + "ls1.gamma"] * F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        h = F.layer_norm(t, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
        f = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        t = t + sd[p + "ls2.gamma"] * F.linear(f, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        if i in blocks_to_take:
            outs.append(t[:, 1:].permute(0, 2, 1).reshape(B, C, h0, w0).contiguous())
    return outs


# ------------------------------------------------------------------------------------------ stage 2
def _mlp3(sd, p, x, last=None):
    x = F.relu(F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"]))
    x = F.relu(F.linear(x, sd[p + "2.weight"], sd[p + "2.bias"]))
    x = F.linear(x, sd[p + "4.weight"], sd[p + "4.bias"])
    return last(x) if last else x


def affine_regressor(sd, x, prefix="affine_regressor."):
    p = prefix
    x = F.relu(F.group_norm(F.conv2d(x, sd[p + "features.0.weight"], sd[p + "features.0.bias"]), 32,
                            sd[p + "features.1.weight"], sd[p + "features.1.bias"]))
    x = F.relu(F.group_norm(F.conv2d(x, sd[p + "features.3.weight"], None, stride=2, padding=1), 32,
                            sd[p + "features.4.weight"], sd[p + "features.4.bias"]))
    x = x.flatten(1)
    x = F.leaky_relu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]), 0.1)
    x = F.leaky_relu(F.linear(x, sd[p + "fc2.weight"], sd[p + "fc2.bias"]), 0.1)
    t = _mlp3(sd, p + "translation_predictor.", x)
    s = _mlp3(sd, p + "scale_predictor.", x)
    ip = F.normalize(_mlp3(sd, p + "inplane_predictor.", x, torch.tanh), dim=1)
    return t, s.squeeze(1), ip


# ------------------------------------------------------------------------------------------ stage 3
def _bn(sd, p, x, train=False):
    """nn.BatchNorm2d(eps=1e-5, momentum=0.1).  train: normalise with the statistics of this batch and update the running
    buffers of `sd` in place (unbiased variance, num_batches_tracked + 1), as the module does in training mode."""
    if train:
        sd[p + "num_batches_tracked"] += 1
        return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], True, 0.1, 1e-5)
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5)


def _rcu(sd, p, x, train=False):
    h = _bn(sd, p + "bn1.", F.conv2d(F.relu(x), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1), train)
    h = _bn(sd, p + "bn2.", F.conv2d(F.relu(h), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1), train)
    return h + x


def _fusion(sd, p, size, x0, x1=None, train=False):
    out = x0 if x1 is None else x0 + _rcu(sd, p + "resConfUnit1.", x1, train)
    out = _rcu(sd, p + "resConfUnit2.", out, train)
    out = F.interpolate(out, size=size, mode="bilinear", align_corners=True)
    return F.conv2d(out, sd[p + "out_conv.weight"], sd[p + "out_conv.bias"])


def dpt_head(sd, feats, prefix="offset_regressor.dpt_head.", train=False):
    p = prefix
    x = [F.conv2d(f, sd[f"{p}projects.{i}.weight"], sd[f"{p}projects.{i}.bias"]) for i, f in enumerate(feats)]
    l1 = F.conv_transpose2d(x[0], sd[p + "resize_layers.0.weight"], sd[p + "resize_layers.0.bias"], stride=4)
    l2 = F.conv_transpose2d(x[1], sd[p + "resize_layers.1.weight"], sd[p + "resize_layers.1.bias"], stride=2)
    l3 = x[2]
    l4 = F.conv2d(x[3], sd[p + "resize_layers.3.weight"], sd[p + "resize_layers.3.bias"], stride=2, padding=1)
    rn = [F.conv2d(l, sd[f"{p}scratch.layer{i + 1}_rn.weight"], None, padding=1) for i, l in enumerate((l1, l2, l3, l4))]
    p4 = _fusion(sd, p + "scratch.refinenet4.", rn[2].shape[2:], rn[3], train=train)
    p3 = _fusion(sd, p + "scratch.refinenet3.", rn[1].shape[2:], p4, rn[2], train)
    p2 = _fusion(sd, p + "scratch.refinenet2.", rn[0].shape[2:], p3, rn[1], train)
    return [p4, p3, p2]


def _pixel_grid(B, H, W):
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(B, 1, 1, 1)


def _sample(feat, coords_xy):
    """bilinear_sample (corr_lookup.py:29-65): pixel coords (..., 2) -> grid_sample(align_corners=True, zeros)."""
    H, W = feat.shape[-2:]
    gx = coords_xy[..., 0] * 2.0 / max(W - 1, 1) - 1.0
    gy = coords_xy[..., 1] * 2.0 / max(H - 1, 1) - 1.0
    return F.grid_sample(feat, torch.stack([gx, gy], dim=-1), "bilinear", "zeros", True)


def corr_lookup(f1, f2, flow, levels, r):
    """CorrelationPyramid + CorrLookup: (B,C,H,W) x2, flow (B,2,H,W) -> (B, levels*(2r+1)^2, H, W)."""
    B, C, H, W = f1.shape
    corr = (f1.reshape(B, C, -1).permute(0, 2, 1) @ f2.reshape(B, C, -1)).reshape(B * H * W, 1, H, W)
    corr = corr / torch.sqrt(torch.tensor(C).float())
    pyr = [corr]
    for _ in range(levels - 1):
        pyr.append(F.avg_pool2d(pyr[-1], 2, 2))
    grid = (_pixel_grid(B, H, W) + flow).permute(0, 2, 3, 1).reshape(B * H * W, 1, 1, 2)
    d = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).reshape(1, 2 * r + 1, 2 * r + 1, 2)
    out = []
    for i, c in enumerate(pyr):
        out.append(_sample(c, grid / 2 ** i + delta).reshape(B, H, W, -1))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous()


def _cm(sd, p, x, pad):
    return F.relu(F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=pad))


def flow_decoder(sd, feat_render_list, feat_real_list, init_flow, init_cert, num_levels=3, radius=4,
                 prefix="offset_regressor.flow_decoder.", train=False):
    r = int(radius / 2)
    flow, cert = init_flow, init_cert
    flows, certs = [], []
    for l in range(num_levels):
        p = f"{prefix}proj.{l}."
        proj = lambda t: _bn(sd, p + "1.", F.conv2d(t, sd[p + "0.weight"], sd[p + "0.bias"]), train)  # noqa: E731
        fr, fq = proj(feat_render_list[l]), proj(feat_real_list[l])
        B, _, H, W = fr.shape
        corr = corr_lookup(fr, fq, flow, l + 1, r)
        e = f"{prefix}encoder.{l}."
        cf = _cm(sd, e + "corr_net.1.", _cm(sd, e + "corr_net.0.", corr, 0), 1)
        ff = _cm(sd, e + "flow_net.1.", _cm(sd, e + "flow_net.0.", flow, 3), 1)
        motion = torch.cat([_cm(sd, e + "out_net.0.", torch.cat([cf, ff], dim=1), 1), flow], dim=1)
        warped = _sample(fq, (_pixel_grid(B, H, W) + flow).permute(0, 2, 3, 1))
        x = torch.cat([fr, warped, motion], dim=1)
        fp, mp = f"{prefix}flow_pred.{l}.", f"{prefix}mask_pred.{l}."
        h = _cm(sd, fp + "layers.1.", _cm(sd, fp + "layers.0.", x, 1), 1)
        flow = flow + F.conv2d(h, sd[fp + "predict_layer.weight"], sd[fp + "predict_layer.bias"], padding=1)
        h = _cm(sd, mp + "layers.1.", _cm(sd, mp + "layers.0.", x, 1), 1)
        cert = cert + F.conv2d(h, sd[mp + "predict_layer.weight"], sd[mp + "predict_layer.bias"])
        flows.append(flow)
        certs.append(cert)
        if l != num_levels - 1:
            flow = 2 * F.interpolate(flow, scale_factor=(2, 2), mode="bilinear", align_corners=True)
            cert = F.interpolate(cert, scale_factor=(2, 2), mode="bilinear", align_corners=True)
    return flows, certs


# ------------------------------------------------------------------------------------------ whole path
def net_forward_test(sd, end_points, hyp, heads, blocks_to_take, num_levels=3, radius=4):
    """model/picopose.py:97-112 + :72-95 restated on the oracle pieces.  Returns (outputs, aux) where aux
    keeps the tensors a parity test needs to reason about discontinuous outputs (scores, final flow/logits)."""
    from . import geometry as og
    from . import matching as om

    feats_real = vit_features(sd, end_points["real_rgb"], heads, blocks_to_take)
    bank = end_points["template_feature"]
    bank = bank / bank.norm(dim=2, keepdim=True).clamp_min(1e-12)           # picopose.py:99
    sim_avg = om.template_scores(bank, feats_real[-1], end_points["real_mask"])
    score, ids = torch.topk(sim_avg, hyp, dim=1)
    B = ids.shape[0]
    rows = torch.arange(B)
    outputs, aux = [], {"sim_avg": sim_avg, "ids": ids, "flow": [], "cert": []}
    for k in range(hyp):
        sel = {key: end_points[key][rows, ids[:, k]] for key in ("tem_pose", "tem_K", "tem_M", "tem_mask", "tem_rgb", "tem_pts3d")}
        out = {"tem_pose": sel["tem_pose"], "tar_pts_2d": end_points["real_pts2d"].permute(0, 3, 2, 1),
               "src_pts_3d": sel["tem_pts3d"].permute(0, 3, 1, 2)}
        feats_tem = vit_features(sd, sel["tem_rgb"], heads, blocks_to_take)
        sim = om.matching_features_similarity(feats_tem[-1], feats_real[-1], sel["tem_mask"], None)
        t, s, ip = affine_regressor(sd, sim)
        Ms = og.calc_pred_Ms(s, ip, t, sel["tem_pose"], sel["tem_K"], sel["tem_M"])
        out["pred_poses"] = og.pose_recovery_2d_prediction(end_points["real_M"], end_points["real_K"], Ms, sel["tem_K"],
                                                           sel["tem_M"], sel["tem_pose"])
        f0, c0 = og.compute_init_correspondences(Ms, sel["tem_mask"])
        fl, ce = flow_decoder(sd, dpt_head(sd, feats_tem), dpt_head(sd, feats_real), f0, c0, num_levels, radius)
        out["pred_tar_pts"], out["pred_src_pts"] = og.compute_stage3_correspondences(fl[-1], ce[-1], 0.5)
        aux["flow"].append(fl[-1])
        aux["cert"].append(ce[-1])
        outputs.append(out)
    return outputs, aux
