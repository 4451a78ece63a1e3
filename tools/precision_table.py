"""Merge the PRECISION_STUDY lines of tests/precision_study.py runs (one per library build) into a markdown table."""
import json
import sys

runs = []
for path in sys.argv[1:]:
    for line in open(path):
        if line.startswith("PRECISION_STUDY "):
            runs.append(json.loads(line[len("PRECISION_STUDY "):]))
name = {"3-term (product)": "3 terms: hi.hi + hi.lo + lo.hi (product build)", "_a1": "2 terms: weights split, activations plain fp16",
        "_a1w1": "1 term: plain fp16 x fp16"}
f = lambda v: "%.1e" % v  # noqa: E731
print("| product form | ViT-B levels (max err / max) | ViT-L levels | affine regressor | DPT head | flow decoder flow (3 levels) | certainty | e2e ViT-B: same templates / stage-2 pose err / key-point slots differing of 4096 |")
print("|---|---|---|---|---|---|---|---|")
for r in runs:
    e = r["e2e_vitb"]
    print(f"| {name.get(r['lib'], r['lib'])} | {' '.join(f(v) for v in r['dinov2_vitb14_levels'])} | {' '.join(f(v) for v in r['dinov2_vitl14_levels'])} | "
          f"{f(r['affine_regressor'])} | {f(r['dpt_head'])} | {' '.join(f(v) for v in r['flow_decoder_flow'])} | {' '.join(f(v) for v in r['flow_decoder_cert'])} | "
          f"{e['same_templates']} / {f(e['pred_poses_max_abs'])} / {e['keypoint_slots_differing']} |")
