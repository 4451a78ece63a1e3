"""Provenance of tests/golden/: the committed fixtures are what oracle/gen_golden.py writes AT HEAD.

Round 3 ended with two fixtures (train_forward*.npz) written under an older head calibration than the one in
oracle/weights.py — still genuine reference outputs (the calibration travels inside the fixture), but no longer what the
generator produces.  Two guards:
  * every `cal_*` array of every fixture equals the table in oracle/weights.py (runs everywhere, no reference needed);
  * the fast generators are re-run against /root/reference into a temp dir and compared array by array, bytes and dtype
    (build container only: skipped where the reference is absent, i.e. on the GPU box).  PP_PROVENANCE_ALL=1 adds the slow
    ones (e2e_calibrated, train_grads: minutes of CPU).
An .npz is compared per array, not per file: numpy stamps the zip members with the wall clock.
"""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PICOPOSE_REFERENCE", "/root/reference")

# generator name -> files it writes
FAST = {"stage1": ["stage1_matching_templates.npz", "stage2_similarity.npz"], "geometry": ["geometry.npz"], "nets": ["nets.npz"],
        "state_dict": ["state_dict_names.json"], "preprocess": ["preprocess_boxes.npz"], "run_test": ["run_test_rows.json"],
        "train_forward": ["train_forward.npz"], "train_forward_edge": ["train_forward_edge.npz"], "vit_wide": ["vit_wide.npz"],
        "e2e": ["e2e.npz"], "train_grads_dup": ["train_grads_dup.npz"]}
SLOW = {"e2e_calibrated": ["e2e_calibrated.npz"], "train_grads": ["train_grads.npz"], "train_grads_f64": ["train_grads_f64.npz"]}


def _cal_table(vit):
    from oracle.weights import AFFINE_CALIBRATION, HEAD_CALIBRATION, PROJ_BN_GAIN

    cal = HEAD_CALIBRATION[vit]
    return {"cal_flow": np.array(cal["flow"], np.float64), "cal_cert": np.array(cal["cert"], np.float64),
            "cal_proj_bn": np.float64(PROJ_BN_GAIN),
            **{f"cal_affine_{h}": np.array([g, *shift], np.float64) for h, (g, shift) in AFFINE_CALIBRATION.items()}}


def test_every_stored_calibration_is_the_committed_table(golden_dir):
    checked = 0
    for path in sorted(glob.glob(os.path.join(golden_dir, "*.npz"))):
        z = np.load(path)
        for key in z.files:
            leaf = key.split("/")[-1]
            if not leaf.startswith("cal_"):
                continue
            prefix = key[: -len(leaf)]
            vit = str(z[prefix + "vit"])
            want = _cal_table(vit)[leaf]
            assert np.array_equal(z[key], want), (os.path.basename(path), key, z[key], want)
            checked += 1
    assert checked >= 30, checked   # 6 arrays x (3 e2e cases + 2 train_forward + train_grads)


def _same_npz(a, b):
    za, zb = np.load(a), np.load(b)
    assert sorted(za.files) == sorted(zb.files), (a, set(za.files) ^ set(zb.files))
    bad = [k for k in za.files if za[k].dtype != zb[k].dtype or za[k].shape != zb[k].shape or za[k].tobytes() != zb[k].tobytes()]
    return bad, len(za.files)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "model")), reason="the reference tree is only present in the build container")
def test_fixtures_regenerate_from_the_reference_at_head(golden_dir, tmp_path):
    gens = dict(FAST, **(SLOW if os.environ.get("PP_PROVENANCE_ALL") == "1" else {}))
    env = dict(os.environ, PICOPOSE_GOLDEN_OUT=str(tmp_path), PICOPOSE_REFERENCE=REF)
    procs = {}
    names = list(gens)
    arrays = 0
    for i in range(0, len(names), 4):       # four generators at a time (each runs torch on 4 threads at most)
        for name in names[i:i + 4]:
            procs[name] = subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), "--only", name],
                                           env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        for name in names[i:i + 4]:
            out, _ = procs[name].communicate(timeout=1800)
            assert procs[name].returncode == 0, (name, out.decode()[-2000:])
    for name, files in gens.items():
        for f in files:
            new, old = os.path.join(str(tmp_path), f), os.path.join(golden_dir, f)
            assert os.path.exists(new), (name, f)
            if f.endswith(".json"):
                assert json.load(open(new)) == json.load(open(old)), f
                assert open(new, "rb").read() == open(old, "rb").read(), f
            else:
                bad, n = _same_npz(new, old)
                assert not bad, (f, f"{len(bad)} of {n} arrays differ from what oracle/gen_golden.py writes at HEAD", bad[:8])
                arrays += n
    assert arrays >= 200
