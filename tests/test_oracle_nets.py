"""CPU: the functional network oracle reproduces the REFERENCE modules' outputs (tests/golden/nets.npz),
with the weights regenerated from the stored seed by state_dict name (our modules provide the names)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from netcfg import HEADS, TAKE, small_cfg  # noqa: E402

from oracle import nets as on  # noqa: E402
from oracle.weights import seeded_state_dict  # noqa: E402


def _z(golden_dir):
    z = np.load(os.path.join(golden_dir, "nets.npz"))
    return z, {k: torch.from_numpy(z[k]) for k in z.files if z[k].dtype == np.float32}


def _close(a, ref, tol):
    err = float(np.abs(a.numpy() - ref).max())
    assert err <= tol * max(1.0, float(np.abs(ref).max())), err


def test_vit_oracle_vs_reference(golden_dir):
    from picopose_amd.model.stage1 import FeatureExtractor

    z, t = _z(golden_dir)
    torch.set_num_threads(4)
    sd = seeded_state_dict(FeatureExtractor(small_cfg().stage1).state_dict(), int(z["vit/seed"]))
    feats = on.vit_features({"feature_extractor." + k: v for k, v in sd.items()}, t["vit/x"], HEADS, TAKE)
    _close(feats[-1], z["vit/feat_last"], 1e-5)
    _close(torch.stack([f[0, :, 3, 5] for f in feats]), z["vit/feat_probe"], 1e-5)


def test_affine_regressor_oracle_vs_reference(golden_dir):
    from picopose_amd.model.stage2 import AffineRegressor

    z, t = _z(golden_dir)
    sd = seeded_state_dict(AffineRegressor(small_cfg().stage2).state_dict(), int(z["aff/seed"]))
    tr, sc, ip = on.affine_regressor({"affine_regressor." + k: v for k, v in sd.items()}, t["aff/sim"])
    _close(tr, z["aff/translation"], 1e-5)
    _close(sc, z["aff/scale"], 1e-5)
    _close(ip, z["aff/inplane"], 1e-5)


def test_stage3_oracle_vs_reference(golden_dir):
    from picopose_amd.model.stage3 import OffsetRegressor

    z, t = _z(golden_dir)
    torch.set_num_threads(4)
    sd = seeded_state_dict(OffsetRegressor(small_cfg().stage3).state_dict(), int(z["s3/seed"]))
    sd = {"offset_regressor." + k: v for k, v in sd.items()}
    dt = on.dpt_head(sd, [t[f"s3/ft{i}"] for i in range(4)])
    dr = on.dpt_head(sd, [t[f"s3/fr{i}"] for i in range(4)])
    _close(dt[0], z["s3/dpt_t_path4"], 1e-5)
    _close(dt[1][0, :, ::8, ::8], z["s3/dpt_t_path3_probe"], 1e-5)
    _close(dt[2][0, :, ::16, ::16], z["s3/dpt_t_path2_probe"], 1e-5)
    fl, ce = on.flow_decoder(sd, dt, dr, t["s3/init_flow"], t["s3/init_cert"])
    for i in range(3):
        _close(fl[i], z[f"s3/flow{i}"], 2e-5)
        _close(ce[i], z[f"s3/cert{i}"], 2e-5)


def test_corr_lookup_oracle_vs_reference(golden_dir):
    z, t = _z(golden_dir)
    out = on.corr_lookup(t["corr/f1"], t["corr/f2"], t["corr/flow"], 3, 2)
    assert out.shape == (2, 75, 16, 16)
    _close(out, z["corr/out"], 1e-6)


def _wide_cfg(vit):
    import types

    C, heads, idx = {"dinov2_vitb14": (768, 12, [[0, 2], [3, 5], [6, 8], [9, 11]]),
                     "dinov2_vitl14": (1024, 16, [[0, 5], [6, 11], [12, 17], [18, 23]])}[vit]
    return types.SimpleNamespace(vit_type=vit, pretrained=False, interaction_indexes=idx), heads, [b[-1] for b in idx]


def test_vit_oracle_vs_reference_at_vitb_and_vitl(golden_dir):
    """The widths the bench and config/base.yaml run (ViT-B/14: 12 heads, K = 768/3072; ViT-L/14: 24 blocks, 16 heads)."""
    from picopose_amd.model.stage1 import FeatureExtractor

    z = np.load(os.path.join(golden_dir, "vit_wide.npz"))
    torch.set_num_threads(8)
    for vit in ("dinov2_vitb14", "dinov2_vitl14"):
        s1, heads, take = _wide_cfg(vit)
        wseed, xseed = (int(v) for v in z[f"{vit}/seeds"])
        sd = seeded_state_dict(FeatureExtractor(s1).state_dict(), wseed)
        x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(xseed))
        feats = on.vit_features({"feature_extractor." + k: v for k, v in sd.items()}, x, heads, take)
        _close(torch.stack([f[0, :, 3, 5] for f in feats]), z[f"{vit}/pixel_probe"], 1e-5)
        _close(torch.stack([f[0, ::32] for f in feats]), z[f"{vit}/channel_probe"], 1e-5)
