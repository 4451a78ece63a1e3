"""Import shim: `importlib.import_module("picopose")` (run_test.py:234 with the reference's `model_name: picopose`)
resolves here when this directory replaces the reference's `model/` on sys.path (run_test.py:17-20)."""
from picopose_amd.picopose import Net  # noqa: F401
