"""CPU: the C-ABI library builds, loads and exports every symbol include/picopose_hip.h declares.
No compute calls here (no GPU)."""
import ctypes

from picopose_amd import _lib
from picopose_amd.build import LIB, build_lib


def test_library_exports_every_declared_symbol():
    build_lib()
    lib = ctypes.CDLL(LIB)
    names = _lib.declared_symbols()
    assert "pp_stage1_match" in names and "pp_topk" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/picopose_hip.h but not exported"


def test_error_strings_and_workspace_query():
    L = _lib.lib()
    assert L.pp_version() >= 100
    assert L.pp_strerror(0) == b"ok"
    for code in (-1, -2, -3):
        assert len(L.pp_strerror(code)) > 5
    need = ctypes.c_size_t()
    assert L.pp_stage1_workspace_bytes(32, 162, 768, ctypes.byref(need)) == 0
    assert 20e6 < need.value < 200e6
    assert L.pp_stage1_workspace_bytes(0, 162, 768, ctypes.byref(need)) == -1  # PP_EINVAL


def test_argument_validation_needs_no_gpu():
    L = _lib.lib()
    # null pointers / unsupported channel count are rejected before any HIP call
    assert L.pp_stage1_scores(None, None, None, 224, 224, 1, 1, 64, 1, 0.0, None, 0, None, None, None) == -1
    assert L.pp_topk(None, 1, 4, 2, None, None, None) == -1
