"""In-memory stand-ins for two third-party modules the reference imports but this image lacks,
used ONLY by oracle/gen_golden.py when it runs the reference on CPU:
  * mmcv.cnn.ConvModule (mmcv==2.0.0, requirements.txt:9) with norm_cfg=None is Conv2d(bias=True)
    stored as `.conv` followed by the activation stored as `.activate` (default ReLU);
  * cv2 (utils/pose_recovery.py:1) is only needed at import time for the functions used here.
"""
import sys
import types

import torch.nn as nn


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), **kw):
        super().__init__()
        assert norm_cfg is None and conv_cfg is None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=True)
        kind = None if act_cfg is None else act_cfg["type"]
        self.activate = {None: None, "ReLU": nn.ReLU(inplace=True), "Sigmoid": nn.Sigmoid(), "Tanh": nn.Tanh()}[kind]

    def forward(self, x):
        x = self.conv(x)
        return x if self.activate is None else self.activate(x)


def install():
    if "mmcv" not in sys.modules:
        mmcv = types.ModuleType("mmcv")
        cnn = types.ModuleType("mmcv.cnn")
        cnn.ConvModule = ConvModule
        mmcv.cnn = cnn
        sys.modules["mmcv"] = mmcv
        sys.modules["mmcv.cnn"] = cnn
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
