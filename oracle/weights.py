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
# last layer of the stage-3 heads per level, measured with weight seed 4 and PROJ_BN_GAIN on the synthetic inputs: (gain, bias shift)
HEAD_CALIBRATION = {
    "dinov2_vits14": {"flow": [(0.2184, 0.08754), (0.05955, 0.03011), (0.06486, -0.6083)],
                      "cert": [(1.701, -0.6621), (0.7953, -1.931), (0.2702, -2.5)]},
    "dinov2_vitb14": {"flow": [(0.1671, 0.1216), (0.1069, -0.04104), (0.1318, 0.2138)],
                      "cert": [(1.952, -1.574), (0.883, 0.5047), (0.3407, 2.576)]},
    "dinov2_vitl14": {"flow": [(0.1285, 0.02845), (0.04446, 0.2571), (0.03843, 0.5124)],
                      "cert": [(1.155, -2.277), (0.2594, 1.235), (0.1157, -1.861)]},
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
