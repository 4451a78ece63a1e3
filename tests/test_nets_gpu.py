"""GPU parity of the network modules (HIP engine) against the CPU oracle, seeded weights shared by name."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from netcfg import HEADS, TAKE, small_cfg  # noqa: E402

from oracle import nets as on  # noqa: E402
from oracle.weights import seeded_state_dict  # noqa: E402

gpu = pytest.mark.gpu


def _rel(a, b):
    a, b = a.cpu(), b.cpu()
    return float((a - b).abs().max()) / max(1e-6, float(b.abs().max()))


@gpu
def test_feature_extractor_matches_oracle():
    from picopose_amd.model.stage1 import FeatureExtractor

    cfg = small_cfg()
    fe = FeatureExtractor(cfg.stage1)
    sd = seeded_state_dict(fe.state_dict(), 11)
    fe.load_state_dict(sd)
    fe = fe.cuda().eval()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 224, 224, generator=g)
    ref = on.vit_features({"feature_extractor.dinov2." + k[len("dinov2."):]: v for k, v in sd.items()}, x, HEADS, TAKE)
    got = fe(x.cuda())
    assert len(got) == 4 and got[0].shape == (2, 384, 16, 16)
    for r, o in zip(ref, got):
        assert _rel(o, r) <= 2e-4, _rel(o, r)  # fp32 both sides; 12 blocks of reassociation noise
