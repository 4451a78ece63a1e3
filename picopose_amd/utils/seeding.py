"""Seeded random-init weights of an architecture, by state_dict name (there is no network for DINOv2 or the authors'
checkpoint, so bench.py and smoke() run on random weights; BatchNorm running stats and LayerScale gammas are
randomised too).  tests/ check that this recipe equals the one the golden fixtures were generated with."""
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
