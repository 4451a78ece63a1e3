"""The per-image latency leg of bench.py (bench.latency_leg: the reference's own measurement — run_test.py:141-216 — at its own configuration:
ViT-L/14, 162 templates, hyp 5, chunks of 4 detections) as a stand-alone program for rocprofv3 --kernel-trace: prints ms per image.
usage: [IMAGES=6] [DETS=8] [BS=4] [MODE=f16x3|f32|f16] python tools/latency_image.py   (tools/latency_profile.sh wraps it)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from picopose_amd import ops  # noqa: E402

ops.PRECISION = os.environ.get("MODE", "f16x3")
dev = torch.device("cuda", 0)
r = bench.latency_leg(dev, "exact" if ops.PRECISION == "f32" else "fast", images=int(os.environ.get("IMAGES", "6")), n_det=int(os.environ.get("DETS", "8")),
                      bs=int(os.environ.get("BS", "4")))
print(json.dumps(r))
