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
