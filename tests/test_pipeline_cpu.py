"""CPU: evaluator-side formatting — no GPU, no library calls."""
import numpy as np

from picopose_amd.pipeline import bop_csv_lines


def test_bop_csv_lines_format_matches_run_test():
    # run_test.py:191-206: scene,img,obj,score,"R (9 values)","t in mm (3 values)",time\n — best hypothesis first
    preds_image = [[{"R_stage_3": np.eye(3).reshape(9), "t_stage_3": np.array([0.01, -0.02, 0.8]) * 1000, "inliers_ratio": 0.9},
                    {"R_stage_3": np.zeros(9), "t_stage_3": np.zeros(3), "inliers_ratio": 0.1}]]
    (line,) = bop_csv_lines(3, 17, [5], [0.75], preds_image, 0.25)
    f = line.split(",")
    assert f[:4] == ["3", "17", "5", "0.75"] and line.endswith("0.25\n")
    assert [float(v) for v in f[4].split(" ")] == list(np.eye(3).reshape(9))
    assert np.allclose([float(v) for v in f[5].split(" ")], [10.0, -20.0, 800.0])


def test_package_seeding_recipe_equals_the_fixture_recipe():
    import torch

    from oracle.weights import seeded_state_dict as ref
    from picopose_amd.utils.seeding import seeded_state_dict as got

    tmpl = {"a.weight": torch.zeros(4, 3, 2, 2), "a.bias": torch.zeros(4), "bn.running_var": torch.zeros(4),
            "bn.running_mean": torch.zeros(4), "bn.num_batches_tracked": torch.zeros((), dtype=torch.long),
            "ls1.gamma": torch.zeros(4), "cls_token": torch.zeros(1, 1, 4), "n.weight": torch.zeros(4)}
    a, b = ref(tmpl, 7), got(tmpl, 7)
    assert all(torch.equal(a[k], b[k]) for k in tmpl)


def test_compat_module_name_resolves_like_the_reference_loader():
    """run_test.py:17-20,234-235: `sys.path.append(<model dir>); MODEL = importlib.import_module("picopose")`."""
    import importlib
    import os
    import sys

    import picopose_amd

    sys.path.insert(0, os.path.join(os.path.dirname(picopose_amd.__file__), "compat"))
    try:
        mod = importlib.import_module("picopose")
        from picopose_amd.picopose import Net

        assert mod.Net is Net
    finally:
        sys.path.pop(0)
        sys.modules.pop("picopose", None)
