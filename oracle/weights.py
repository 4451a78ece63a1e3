"""Seeded weights for the parity fixtures (TEST INFRASTRUCTURE).  There is no network for DINOv2 or the
authors' checkpoint, so fixtures use seeded random weights; BatchNorm running stats and LayerScale gammas are
randomised too (their default init is identity-like and would hide bugs — SURVEY.md §8c).  The same
function fills the reference model (oracle/gen_golden.py) and ours (tests), by state_dict name."""
import torch


def seeded_state_dict(template, seed):
    """template: ordered {name: tensor}; returns {name: tensor} with values drawn by tensor kind."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, t in template.items():
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            v = torch.zeros(shape, dtype=torch.long)
        elif name.endswith("running_var"):
            v = torch.rand(shape, generator=g) + 0.5
        elif name.endswith("running_mean"):
            v = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".gamma"):
            v = 0.3 + 0.7 * torch.rand(shape, generator=g)
        elif name.endswith(("cls_token", "pos_embed", "mask_token")):
            v = 0.1 * torch.randn(shape, generator=g)
        elif len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            v = torch.randn(shape, generator=g) * (1.5 / fan_in) ** 0.5
        elif name.endswith("weight"):  # norm scales
            v = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:  # biases
            v = 0.05 * torch.randn(shape, generator=g)
        out[name] = v
    return out


# ---- head calibration: realistic key-point occupancy from random weights (oracle/calibrate_heads.py) ---------------
# last layer of each stage-2 head: (gain on weight and bias, value added to the bias)
AFFINE_CALIBRATION = {"translation": (1.0, (0.0, 0.0)), "scale": (0.2, (1.0,)), "inplane": (0.3, (1.0, 0.0))}
# eval BatchNorm of the flow decoder's feature projections (proj.l.1): scale and shift times this.  With the plain draw
# the projected maps have std ~20, their correlation ~400 and the decoder's hidden maps reach 5e4 — a trained decoder
# works on O(1) maps, and 5e4 is outside the f16x3 engine's operand range (|x| < 16376, picopose_amd/ops.py)
PROJ_BN_GAIN = 0.05
# last layer of the stage-3 heads per level: (gain, bias shift).  Round 3: the round-2 table (oracle/calibrate_heads.py: flow
# updates of std 0.15 / 0.2 / 0.4 grid px per level, certainty updates of std 1 around 0) made the VALID key-points a random half
# of the 4096 slots — independent of the template mask, so half of them carried the out-of-mask init flow (-grid index: tens of
# cells) — and gave the in-mask ones 1.5-2 image px of noise against PnP's 2 px threshold: inlier ratios 0.3-0.5 and a RANSAC
# winner that depended on the eigen-solver.  Now flow gains and shifts x 0.2 (0.03 / 0.04 / 0.08 grid px per level: ~0.3 image px in
# total) and certainty x 0.15 with -0.5 on the first level, so that the final logit is (template mask - 0.5) + noise of std ~0.25:
# the valid key-points are the object's, a few per cent flip at the threshold (which keeps that logic exercised).
HEAD_CALIBRATION = {
    "dinov2_vits14": {"flow": [(0.04368, 0.01751), (0.01191, 0.006022), (0.01297, -0.1217)],
                      "cert": [(0.2551, -0.5993), (0.1193, -0.2897), (0.04053, -0.375)]},
    "dinov2_vitb14": {"flow": [(0.03342, 0.02432), (0.02138, -0.008208), (0.02636, 0.04276)],
                      "cert": [(0.2928, -0.7361), (0.1324, 0.07571), (0.0511, 0.3864)]},
    "dinov2_vitl14": {"flow": [(0.0257, 0.00569), (0.008892, 0.05142), (0.007686, 0.1025)],
                      "cert": [(0.1732, -0.8416), (0.03891, 0.1852), (0.01735, -0.2792)]},
}


def apply_head_calibration(sd, cal):
    """-> copy of `sd` with the LAST layer of every prediction head rescaled: w' = g*w, b' = g*b + shift.
    cal = {"affine": AFFINE_CALIBRATION-like, "proj_bn": gain, "flow": [(g, shift)]*levels, "cert": [(g, shift)]*levels}."""
    out = dict(sd)

    def rescale(prefix, g, shift):
        out[prefix + "weight"] = sd[prefix + "weight"] * g
        out[prefix + "bias"] = sd[prefix + "bias"] * g + torch.as_tensor(shift, dtype=sd[prefix + "bias"].dtype)

    for head, (g, shift) in cal.get("affine", {}).items():
        rescale(f"affine_regressor.{head}_predictor.4.", g, shift)
    if "proj_bn" in cal:
        for l in range(len(cal.get("flow", ()))):
            rescale(f"offset_regressor.flow_decoder.proj.{l}.1.", cal["proj_bn"], 0.0)
    for key, name in (("flow", "flow_pred"), ("cert", "mask_pred")):
        for l, (g, shift) in enumerate(cal.get(key, ())):
            rescale(f"offset_regressor.flow_decoder.{name}.{l}.predict_layer.", g, shift)
    return out


def calibrated_state_dict(template, seed, vit_type):
    """Seeded weights with the committed head calibration of the architecture."""
    return apply_head_calibration(seeded_state_dict(template, seed), dict(HEAD_CALIBRATION[vit_type], affine=AFFINE_CALIBRATION, proj_bn=PROJ_BN_GAIN))
