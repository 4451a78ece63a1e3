"""Stage 2: AffineRegressor on the HIP engine — mirrors model/stage2/affine_regressor.py:6-84.

conv1x1(256->256) + GroupNorm(32) + ReLU, conv3x3 stride 2 (no bias) + GroupNorm(32) + ReLU, flatten,
fc1 16384->1024, LeakyReLU(0.1), fc2 1024->256, LeakyReLU(0.1), three 3-layer MLP heads:
translation (B,2), scale (B,), in-plane cos/sin tanh -> L2-normalised (B,2)."""
import torch

from .. import ops
from .common import Packed, conv_p, linear_p, norm_p, seq


def _head(hidden, out):
    return seq((0, linear_p(hidden, hidden)), (2, linear_p(hidden, hidden)), (4, linear_p(hidden, out)))


class AffineRegressor(Packed):
    def __init__(self, cfg, use_tanh_act=True, normalize_output=True):
        super().__init__()
        self.in_channel, self.hidden_dim = cfg.in_channel, cfg.hidden_dim
        self.feat_size = 8
        hd = self.hidden_dim
        self.features = seq((0, conv_p(self.in_channel, hd, 1)), (1, norm_p(hd)), (3, conv_p(hd, hd, 3, bias=False)),
                            (4, norm_p(hd)))
        self.fc1 = linear_p(hd * self.feat_size * self.feat_size, 1024)
        self.fc2 = linear_p(1024, 256)
        self.translation_predictor = _head(hd, 2)
        self.scale_predictor = _head(hd, 1)
        self.inplane_predictor = _head(hd, 2)

    def _pack(self):
        f = self.features
        hd, fs = self.hidden_dim, self.feat_size
        # x.flatten(1) of the NCHW map indexes (c, h, w); the engine's map is (h, w, c)
        fc1 = self.fc1.weight.float().view(-1, hd, fs, fs).permute(0, 2, 3, 1).reshape(-1, fs * fs * hd).contiguous()
        return {"c0": ops.pack_conv_weight(getattr(f, "0").weight.float()),
                "c3": ops.pack_conv_weight(getattr(f, "3").weight.float()), "fc1": fc1}

    def _mlp(self, head, x, last_act=None):
        l0, l2, l4 = getattr(head, "0"), getattr(head, "2"), getattr(head, "4")
        x = ops.linear(x, l0.weight, l0.bias, act="relu")
        x = ops.linear(x, l2.weight, l2.bias, act="relu")
        return ops.linear(x, l4.weight, l4.bias, act=last_act)

    def forward(self, x):
        """(B,256,16,16) -> translation (B,2), scale (B,), inplane (B,2)."""
        with torch.no_grad():
            pk, f = self.packed(), self.features
            B = x.shape[0]
            h = ops.conv2d(ops.to_nhwc(x), pk["c0"], getattr(f, "0").bias, 1)
            h = ops.groupnorm(h, getattr(f, "1").weight, getattr(f, "1").bias, 32, relu=True)
            h = ops.conv2d(h, pk["c3"], None, 3, stride=2, pad=1)
            h = ops.groupnorm(h, getattr(f, "4").weight, getattr(f, "4").bias, 32, relu=True)
            # fc1: B rows x 16384 x 1024 — split along K into 32 slices that run as one batched GEMM (ops.linear_splitk)
            h = ops.linear_splitk(h.view(B, -1), pk["fc1"], self.fc1.bias, act="leaky01")
            h = ops.linear(h, self.fc2.weight, self.fc2.bias, act="leaky01")
            translation = self._mlp(self.translation_predictor, h)
            scale = self._mlp(self.scale_predictor, h)
            inplane = ops.normalize_rows(self._mlp(self.inplane_predictor, h, last_act="tanh"))
            return translation, scale.squeeze(1), inplane
