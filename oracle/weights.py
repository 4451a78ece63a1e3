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
# last layer of the stage-3 heads per level, measured with weight seed 4 on the synthetic inputs: (gain, bias shift)
HEAD_CALIBRATION = {
    "dinov2_vits14": {"flow": [(0.0008216, 0.06904), (0.0002099, 0.03091), (0.0001567, -0.5886)],
                      "cert": [(0.01325, -0.4792), (0.003205, -1.727), (0.0006062, -1.996)]},
    "dinov2_vitb14": {"flow": [(0.0006585, 0.1665), (0.0003779, 0.07594), (0.0006391, 0.313)],
                      "cert": [(0.00871, -1.248), (0.003779, 1.673), (0.0009429, 2.101)]},
    "dinov2_vitl14": {"flow": [(0.0003526, 0.05194), (0.0001661, 0.2838), (9.149e-05, 0.3788)],
                      "cert": [(0.003864, -1.477), (0.0007211, 1.431), (0.0002781, -1.573)]},
}


def apply_head_calibration(sd, cal):
    """-> copy of `sd` with the LAST layer of every prediction head rescaled: w' = g*w, b' = g*b + shift.
    cal = {"affine": AFFINE_CALIBRATION-like, "flow": [(g, shift)]*levels, "cert": [(g, shift)]*levels}."""
    out = dict(sd)

    def rescale(prefix, g, shift):
        out[prefix + "weight"] = sd[prefix + "weight"] * g
        out[prefix + "bias"] = sd[prefix + "bias"] * g + torch.as_tensor(shift, dtype=sd[prefix + "bias"].dtype)

    for head, (g, shift) in cal.get("affine", {}).items():
        rescale(f"affine_regressor.{head}_predictor.4.", g, shift)
    for key, name in (("flow", "flow_pred"), ("cert", "mask_pred")):
        for l, (g, shift) in enumerate(cal.get(key, ())):
            rescale(f"offset_regressor.flow_decoder.{name}.{l}.predict_layer.", g, shift)
    return out


def calibrated_state_dict(template, seed, vit_type):
    """Seeded weights with the committed head calibration of the architecture."""
    return apply_head_calibration(seeded_state_dict(template, seed), dict(HEAD_CALIBRATION[vit_type], affine=AFFINE_CALIBRATION))
