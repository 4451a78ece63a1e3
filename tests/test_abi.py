"""CPU: the C-ABI library builds, loads and exports every symbol include/picopose_hip.h declares.
No compute calls here (no GPU)."""
import ctypes

import pytest

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
    # the entry points added for the training forward, the narrow predict layers and the operand-side helpers
    assert L.pp_train_keypoints_workspace_bytes(0) == 0 and L.pp_train_keypoints_workspace_bytes(2) > 2 * 4096 * 24
    assert L.pp_train_keypoints(*([None] * 2), 224, 224, *([None] * 2), 480, 640, *([None] * 10), 2, None, None, None, 0, None) == -1
    assert L.pp_batchnorm_train_workspace_bytes(1000, 256) >= 4 * 256 * 16
    assert L.pp_batchnorm_train(None, None, None, 10, 256, 1e-5, 0.1, None, None, 0, None, None, None, None, 0, None) == -1
    assert L.pp_gather_normalize_rows(None, 64, None, 4, 64, 1e-12, None, None) == -1
    assert L.pp_xent_diag_rows(None, 4, 4, 10.0, None, None) == -1
    assert L.pp_flow_loss_blocks() > 0 and L.pp_flow_loss_sums(None, None, None, 2, 16, 16, 400.0, None, None) == -1
    assert L.pp_conv_narrow_hl(None, 256, 1, 64, 64, 256, None, None, 3, 2, None, None, None) == -1
    assert L.pp_conv_narrow_f32(None, 256, 1, 64, 64, 256, None, None, 3, 2, None, None, None) == -1
    assert L.pp_corr_lookup_nhwc_hl(None, 256, None, None, None, 1, None, 1, 64, 64, 256, 3, 2, 2, None, 80, None) == -1
    assert L.pp_hl_patch_columns(None, 2, 2, 10, None, 640, 638, None) == -1
    assert L.pp_sum_slices(None, 32, 160, 1024, None, 3, None, None) == -1
    # round 6: the Winograd F(4x4, 3x3) entries reject null operands, maps that are no multiple of the 4x4 tile, channel counts off
    # the operand's 8-column groups, a tile count above the padded row count and byte extents beyond the 32-bit buffer offsets
    import ctypes as c
    buf = (c.c_char * 64)()
    p = c.addressof(buf) + (-c.addressof(buf)) % 16
    assert L.pp_winograd4_input_hl(None, 64, 64 * 64 * 64, 1, 64, 64, 64, 0, None, 256, None) == -1
    assert L.pp_winograd4_input_hl(p, 64, 62 * 64 * 64, 1, 62, 64, 64, 0, p, 256, None) == -1          # H % 4
    assert L.pp_winograd4_input_hl(p, 60, 64 * 64 * 60, 1, 64, 64, 60, 0, p, 256, None) == -1          # C % 8
    assert L.pp_winograd4_input_hl(p, 64, 64 * 64 * 64, 1, 64, 64, 64, 0, p, 255, None) == -1          # P_pad < tiles
    assert L.pp_winograd4_input_hl(p, 1024, 64 * 64 * 1024, 400, 64, 64, 1024, 0, p, 102400, None) == -1   # operand beyond 4 GiB of offsets
    assert L.pp_winograd4_weight_f32(None, 64, 64, 576, None, None) == -1
    assert L.pp_winograd4_weight_f32(p, 64, 64, 575, p, None) == -1                                    # ldw < 9 Cin
    assert L.pp_winograd4_output(None, 64, 1, 64, 64, 64, None, 0, None, None, None, 64, None, 64, 0, 256, None) == -1
    assert L.pp_winograd4_output(p, 64, 1, 64, 64, 64, None, 0, None, None, None, 64, None, 64, 0, 256, None) == -1   # neither output
    assert L.pp_winograd4_output(p, 64, 1, 64, 64, 64, None, 7, None, None, p, 64, None, 64, 0, 256, None) == -1      # activation code
    assert L.pp_winograd4_output(p, 64, 1, 64, 64, 64, None, 0, p, None, None, 64, p, 64, 0, 256, None) == -1         # residual without fp32 out
    assert L.pp_winograd4_output(p, 32, 1, 64, 64, 64, None, 0, None, None, p, 64, None, 64, 0, 256, None) == -1      # ld_y < Cout
    assert L.pp_winograd4_chain(None, 64, 1, 64, 64, 64, None, 0, 0, None, 256, None) == -1
    assert L.pp_winograd4_chain(p, 64, 1, 64, 48, 64, None, 0, 0, p, 256, None) == -1                  # widths 16 / 32 / 64 only
    assert L.pp_winograd4_chain(p, 48, 1, 64, 64, 48, None, 0, 0, p, 256, None) == -1                  # C % 32
    assert L.pp_winograd_chain_f32(None, 1, 64, 64, 64, None, 0, 0, None, None) == -1
    assert L.pp_winograd_chain_f32(p, 1, 63, 64, 64, None, 0, 0, p, None) == -1                        # H % 2
    assert L.pp_winograd_chain_f32(p, 1, 64, 24, 64, None, 0, 0, p, None) == -1
    # ... the resize with both outputs (channel count off the operand's 8-column groups, a missing output, operand format other than 1 / 2 terms)
    assert L.pp_resize_bilinear_nhwc_dual(None, 1, 16, 16, 64, 32, 32, 1.0, None, None, 2, None) == -1
    assert L.pp_resize_bilinear_nhwc_dual(p, 1, 16, 16, 60, 32, 32, 1.0, p, p, 2, None) == -1
    assert L.pp_resize_bilinear_nhwc_dual(p, 1, 16, 16, 64, 32, 32, 1.0, None, p, 2, None) == -1
    assert L.pp_resize_bilinear_nhwc_dual(p, 1, 16, 16, 64, 32, 32, 1.0, p, p, 3, None) == -1
    # ... and the correlation lookup's arithmetic argument: PP_PREC_F32 / F16X3 / F16 (0 / 1 / 2), nothing else
    assert L.pp_corr_lookup_nhwc_ex(p, 64, p, None, None, 1, p, 1, 16, 16, 64, 1, 2, 2, 3, p, 32, None) == -1


def test_state_dict_names_shapes_equal_the_reference(golden_dir):
    """`Net(cfg).state_dict()` must carry the reference Net's tensors — same names, same ORDER, same shapes and
    dtypes — for ViT-S/B/L, so that Lite.load_from_checkpoint(..., network=model) loads the authors' checkpoint
    (run_test.py:272).  Fixture: tests/golden/state_dict_names.json, dumped from the reference (oracle/gen_golden.py)."""
    import json
    import os
    import types

    from picopose_amd.picopose import Net

    ns = types.SimpleNamespace
    table = json.load(open(os.path.join(golden_dir, "state_dict_names.json")))
    widths = {"dinov2_vits14": (384, [[0, 2], [3, 5], [6, 8], [9, 11]]), "dinov2_vitb14": (768, [[0, 2], [3, 5], [6, 8], [9, 11]]),
              "dinov2_vitl14": (1024, [[0, 5], [6, 11], [12, 17], [18, 23]])}
    assert set(table) == set(widths)
    for vit, (C, idx) in widths.items():
        cfg = ns(hypothesis=5, stage1=ns(vit_type=vit, pretrained=False, interaction_indexes=idx), stage2=ns(in_channel=256, hidden_dim=256),
                 stage3=ns(nclass=1, in_channels=C, use_bn=True, out_channels=[256, 512, 1024, 1024], num_levels=3, radius=4))
        got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in Net(cfg).state_dict().items()]
        assert got == table[vit], next((a, b) for a, b in zip(got, table[vit]) if a != b)
    assert len(table["dinov2_vitl14"]) == 603


def test_pack_cache_is_dropped_when_a_checkpoint_loads_through_the_top_level_net():
    """nn.Module.load_state_dict never calls a child's load_state_dict, so a pack-once cache keyed on
    nothing survived `Net.load_state_dict` and later forwards mixed old packed conv weights with new linears."""
    import sys
    import os

    import torch

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from netcfg import small_cfg

    from picopose_amd.model.common import Packed
    from picopose_amd.picopose import Net
    from picopose_amd.utils.seeding import seeded_state_dict

    net = Net(small_cfg())
    net.load_state_dict(seeded_state_dict(net.state_dict(), 1))
    packed = [m for m in net.modules() if isinstance(m, Packed)]
    assert len(packed) >= 4
    fd = net.offset_regressor.flow_decoder
    old = fd.packed()["fp0_p"].clone()                       # packs on first use (pure re-layout: runs on the CPU too)
    for m in packed:
        m._pack_cache = m._pack_cache or {"stale": True}
    net.load_state_dict(seeded_state_dict(net.state_dict(), 2))
    assert all(m._pack_cache is None for m in packed)
    assert not torch.equal(fd.packed()["fp0_p"], old)
    # loading through a wrapper one level further up (Lightning's `network.` prefix) drops it as well
    wrapper = torch.nn.Module()
    wrapper.network = net
    fd.packed()
    wrapper.load_state_dict({"network." + k: v for k, v in seeded_state_dict(net.state_dict(), 3).items()})
    assert fd._pack_cache is None


def test_pack_cache_follows_in_place_parameter_updates():
    """the packed / BN-folded copies are derived from the parameters; an in-place update between two forwards
    (an optimizer step, `with torch.no_grad(): p.mul_()`) must not leave them stale.  The cache records every source tensor's
    (address, version) and rebuilds on a mismatch; a BatchNorm buffer update re-folds the eval packing only.
    ADVICE r03: a write through `p.data` moves NEITHER number (`.data` is a view with its own version counter) — that case is
    the caller's: `Net.invalidate_packed()` after it.  Both behaviours are pinned here."""
    import os
    import sys

    import torch

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from netcfg import small_cfg

    from picopose_amd.picopose import Net
    from picopose_amd.utils.seeding import seeded_state_dict

    net = Net(small_cfg())
    net.load_state_dict(seeded_state_dict(net.state_dict(), 1))
    fd = net.offset_regressor.flow_decoder
    first = fd.packed()
    assert fd.packed() is first                                              # unchanged sources: the cache is reused
    old = first["proj0"].clone()
    with torch.no_grad():
        getattr(fd.proj[0], "0").weight.mul_(2.0)                                       # an optimizer-style in-place step
    assert not torch.equal(fd.packed()["proj0"], old)
    second = fd.packed()
    with torch.no_grad():
        getattr(fd.proj[0], "1").running_var.add_(1.0)                                  # a BatchNorm buffer moved: the fold is stale
    third = fd.packed()
    assert third is not second and not torch.equal(third["proj0"], second["proj0"])
    fd.invalidate_packed()
    assert fd._pack_cache is None and torch.equal(fd.packed()["proj0"], third["proj0"])
    # a write through .data: invisible to (address, version) — documented — and picked up after Net.invalidate_packed()
    w = getattr(fd.proj[0], "0").weight
    v0, a0 = w._version, w.data_ptr()
    w.data.mul_(0.5)
    assert (w._version, w.data_ptr()) == (v0, a0)
    assert torch.equal(fd.packed()["proj0"], third["proj0"])                  # stale, as documented
    net.invalidate_packed()
    assert all(m._pack_cache is None for m in net.modules() if hasattr(m, "_pack_cache"))
    assert not torch.equal(fd.packed()["proj0"], third["proj0"])


@pytest.mark.gpu
def test_library_loaded_before_torch_still_launches():
    """`__graft_entry__.build()` opens the library before anything has imported torch.  PyTorch-ROCm bundles its own HIP
    runtime; if the library bound to /opt/rocm's copy instead, the process would hold two runtimes and every launch on a
    torch stream would fail (seen on the GPU box: build() followed by smoke() in one process).  _lib.lib() therefore
    imports torch first — checked here in a fresh interpreter."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import __graft_entry__ as g; g.build(); import torch; from picopose_amd.utils import matching as hm; "
            "g0 = torch.Generator().manual_seed(0); bank = torch.randn(1, 4, 384, 16, 16, generator=g0).cuda(); "
            "q = torch.randn(1, 384, 16, 16, generator=g0).cuda(); s, i = hm.matching_templates(bank, q, None, torch.ones(1, 224, 224).cuda(), topk=4); "
            "torch.cuda.synchronize(); print('LAUNCH OK', i.tolist())")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0 and "LAUNCH OK" in r.stdout.decode(), r.stdout.decode()[-2000:]
