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
