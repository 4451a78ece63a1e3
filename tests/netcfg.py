"""Shared test configuration of the network fixtures (ViT-S/14: the smallest architecture the reference's
FeatureExtractor can build, feature_extractor.py:12-18)."""
import types

ns = types.SimpleNamespace


def small_cfg():
    return ns(hypothesis=5,
              stage1=ns(vit_type="dinov2_vits14", pretrained=False, interaction_indexes=[[0, 2], [3, 5], [6, 8], [9, 11]]),
              stage2=ns(in_channel=256, hidden_dim=256),
              stage3=ns(nclass=1, in_channels=384, use_bn=True, out_channels=[256, 512, 1024, 1024], num_levels=3,
                        radius=4))


HEADS = 6
TAKE = [2, 5, 8, 11]
