import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "stress: load / repetition tests outside the product's configuration (PP_RUN_STRESS=1 to run)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _fresh_saturation_word(request):
    """The operand-saturation word (picopose_amd/ops.py) is STICKY and process-wide: a test that drives an engine with out-of-range values
    (plain random weights, deliberately huge inputs) and never reads its poses would leave it set for the next test that does.  GPU tests
    start with a clear word; nothing here touches the GPU in the CPU suite."""
    if request.node.get_closest_marker("gpu") is not None:
        import torch

        if torch.cuda.is_available():
            from picopose_amd import ops

            if ops._sat_words:
                ops.saturation_raised()
    yield
