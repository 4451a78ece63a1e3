"""Thin host wrappers over the network engine of libpicopose_hip.so (pp_gemm.hip, pp_sample.hip).

All activations are token-major / NHWC fp32 device tensors: (rows, C) or (B, H, W, C).  Nothing
here computes with torch ops — torch only allocates the outputs."""
import ctypes
import os

import torch

from . import _lib
from ._lib import PpGemmDesc

ACT = {None: 0, "none": 0, "relu": 1, "gelu": 2, "leaky01": 3, "tanh": 4}

# Arithmetic of the GEMM engine (include/picopose_hip.h PP_PREC_*):
#   "f32"   v_mfma_f32_32x32x2_f32 — exact fp32 products;
#   "f16x3" every operand split into two fp16 terms (22 bits), 3 fp16 MFMAs, fp32 accumulate: fp32-grade results
#           (operand rounding 2^-22 instead of 2^-24) at a fraction of the matrix-pipe time.
#   "f16"   plain fp16 operands f16(4 x) ("h" format), ONE fp16 MFMA per product, fp32 accumulate — the arithmetic BASELINE
#           configs[4] names ("fp16 storage / MFMA with fp32 accumulate"); half the operand bytes and a third of the MFMAs
#           of f16x3, at fp16-grade results (error ~5e-4 .. 5e-3 of a tensor's maximum, profiles/r02/precision_study.md).
PRECISION = "f16x3"
_PREC = {"f32": 0, "f16x3": 1, "f16": 2}
_split_cache = {}


def terms():
    """fp16 terms per operand element of the engine's current operand format: 2 = hl (f16x3), 1 = h (f16); 0 = no operands (f32)."""
    return {"f32": 0, "f16x3": 2, "f16": 1}[PRECISION]


def presplit():
    """The engine runs on pre-split operands (Split objects between producers and consumers)."""
    return PRECISION != "f32"

# The f16x3 engine stores an activation operand as hi = f16(4 x), lo = f16(4 x - hi): |x| must stay below
# 65504 / 4 = 16376, beyond which the split SATURATES (finite, but wrong) — fp32 has no such limit.  Trained networks
# stay orders of magnitude below it; badly scaled random weights need not.  CHECK_SATURATION = True makes every producer
# of an operand verify its buffer (a host sync per operand: a debugging / test / bench-verification mode, never on in a
# timed region) and raise instead of returning silently clipped values.
CHECK_SATURATION = False
PP_A_SCALE = 4.0              # csrc/pp_common.h: activation operand scale
saturation_checks = 0        # operands verified since import (bench.py reports it)


# The default guard (round 6): a STICKY device word.  Every kernel that writes operand terms ORs bit 0 into it when a term hit the fp16 clamp
# (csrc/pp_common.h pp_sat_flag: an atomic only from a wave that saw one — nothing in a healthy forward); the host reads the word together
# with a batch's poses (picopose_amd/utils/pose_recovery.py: one more row of the packed device->host copy — no extra synchronisation) and
# raises instead of returning poses computed from clipped operands.  A caller of the bare `Net.forward` asks with `saturation_raised()`
# (a 4-byte copy + wait) or reads `saturation_word()` itself.  PP_SAT_FLAG=0 switches the reporting off.
SATURATION_FLAG = os.environ.get("PP_SAT_FLAG", "1") != "0"
_sat_words = {}            # device index -> registered int32 tensor (1,)


def saturation_word(device=None):
    """The registered saturation word of `device` (default: the current one): int32 (1,), 0 = no operand term was clamped since the last
    reset.  Registered with the library on first use; None when the reporting is switched off."""
    if not SATURATION_FLAG:
        return None
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    w = _sat_words.get(idx)
    if w is None:
        with torch.cuda.device(idx):
            w = torch.zeros(1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()      # (the zero is in place before the library's kernels may write the word from another stream)
            _lib.check(_lib.lib().pp_set_saturation_word(_p(w)), "pp_set_saturation_word")
        _sat_words[idx] = w
    return w


def saturation_error(what="a forward since the last check"):
    return _lib.PicoPoseHipError(
        f"an f16x3 / f16 operand saturated in {what}: an activation reached the fp16 range of the engine's operand format (|x| >= 16376; "
        "Winograd-transformed maps: >= 10480) and was clamped — the results of this batch are wrong.  Run this network with "
        "ops.PRECISION = 'f32' (bench.py --mode exact); ops.CHECK_SATURATION = True names the layer")


def saturation_raised(reset=True, device=None):
    """True if an operand term was clamped since the last reset (waits for the device: a 4-byte copy)."""
    w = saturation_word(device)
    if w is None:
        return False
    hit = bool(w.item())
    if hit and reset:
        w.zero_()
    return hit


def _chk(hl, what):
    """hl: fp16 operand buffer (or None).  Raises if any element sits at the fp16 clamp (|v| >= 65504) or is not finite."""
    global saturation_checks
    if CHECK_SATURATION and hl is not None:
        saturation_checks += 1
        bad = (hl.view(torch.int16) & 0x7FFF) >= 0x7BFF
        if bool(bad.any()):
            raise _lib.PicoPoseHipError(
                f"f16x3 operand saturated in {what}: {int(bad.sum())} of {hl.numel()} fp16 terms at the clamp "
                "(|activation| >= 16376); run with ops.PRECISION = 'f32' (bench.py --mode exact) for this network")
    return hl


def split_weight(w, cache=True):
    """(hl, scale) — the f16x3 "hl" operand (fp16 (N, 2K): per 8 k the hi then the lo terms) and power-of-two scale
    of a weight matrix; split once per tensor version (one host sync to read the scale back).  cache=False: a transient matrix
    (the training graph's re-packed convolution weights): split, not remembered, not kept alive."""
    t = terms()
    if not cache:
        hl = torch.empty(w.shape[0], t * w.shape[1], dtype=torch.float16, device=w.device)
        scale = torch.empty(1, dtype=torch.float32, device=w.device)
        _lib.check(_lib.lib().pp_split_weights_t(_p(w), w.numel(), t, _p(hl), _p(scale), _lib.stream_ptr()), "pp_split_weights_t")
        return hl, float(scale.item())
    # keyed by the tensor's ADDRESS (+ shape, format); the version lives in the value: an optimizer step bumps the version, and the entry
    # of the old version — which can never be hit again — is overwritten in place instead of piling up (~4 bytes per
    # parameter per step of dead device memory until the wholesale clear)
    key = (w.data_ptr(), tuple(w.shape), t)
    hit = _split_cache.get(key)
    if hit is None or hit[3] != w._version:
        assert w.is_contiguous()
        hl = hit[0] if hit is not None and hit[0].shape == (w.shape[0], t * w.shape[1]) else \
            torch.empty(w.shape[0], t * w.shape[1], dtype=torch.float16, device=w.device)
        scale = torch.empty(1, dtype=torch.float32, device=w.device)
        _lib.check(_lib.lib().pp_split_weights_t(_p(w), w.numel(), t, _p(hl), _p(scale), _lib.stream_ptr()), "pp_split_weights_t")
        if len(_split_cache) > 4096:       # (last resort: thousands of distinct live weight tensors)
            _split_cache.clear()
        hit = _split_cache[key] = (hl, float(scale.item()), w, w._version)  # keep w alive so its address is not reused
    return hit[0], hit[1]


def split_weight_dev(w):
    """(hl, scale2) of a trained PARAMETER with its power-of-two scale left on the device (scale2 = [2^k, 2^-k]): no host wait.  The
    training step re-splits every parameter after every optimizer step; `split_weight` reads each scale back (197 host waits per step at
    ViT-B, the GPU idling while the host catches up) — here the launch takes 2^-k through PpGemmDesc.alpha_dev with b_scale = 1
    (exact: powers of two).  Remembered per (address, version) like `split_weight`."""
    t = terms()
    key = (w.data_ptr(), tuple(w.shape), t, "dev")
    hit = _split_cache.get(key)
    if hit is None or hit[3] != w._version:      # (a new version re-splits INTO the previous version's buffers: nothing accumulates)
        assert w.is_contiguous()
        if hit is not None and hit[0].shape == (w.shape[0], t * w.shape[1]):
            hl, buf = hit[0], hit[4]
        else:
            hl = torch.empty(w.shape[0], t * w.shape[1], dtype=torch.float16, device=w.device)
            buf = torch.empty(2 + 1024, dtype=torch.float32, device=w.device)
        _lib.check(_lib.lib().pp_split_weights_ws(_p(w), w.numel(), t, _p(hl), _p(buf), _p(buf[2:]), _lib.stream_ptr()), "pp_split_weights_ws")
        if len(_split_cache) > 4096:
            _split_cache.clear()
        hit = _split_cache[key] = (hl, buf[:2], w, w._version, buf)
    return hit[0], hit[1]


def drop_split_cache():
    """Forget every cached operand copy of a weight (Packed.invalidate_packed: after a write through `param.data`, which the
    (address, version) keys cannot see)."""
    _split_cache.clear()
    _wino_cache.clear()
    _tracked_scale.clear()


_tracked_scale = {}   # (id(parameter), tag) -> [scale (host float), pending (pinned tensor, event) | None, weakref to the parameter]


def split_weight_tracked(w, owner, tag=""):
    """(hl, scale) of a matrix DERIVED from the parameter `owner` (a re-packed / flipped / reshaped copy made by the training graph
    every step) without a host wait: the split uses the power-of-two scale the host read back one step EARLIER (any power of two
    near max|w| serves — the weights move by far less than a factor of two per step), and this call schedules the next read
    (pp_pow2_scale + an asynchronous copy into pinned memory).  Only the first use of a (parameter, tag) waits."""
    import weakref

    key = (id(owner), tag)
    ent = _tracked_scale.get(key)
    if ent is None or ent[2]() is not owner:
        hl, scale = split_weight(w, cache=False)
        if len(_tracked_scale) > 8192:
            _tracked_scale.clear()
        _tracked_scale[key] = [scale, None, weakref.ref(owner)]
        return hl, scale
    if ent[1] is not None and ent[1][1].query():
        v = float(ent[1][0][0])
        if v > 0.0 and v == v and v != float("inf"):
            ent[0] = v
        ent[1] = None
    t = terms()
    hl = torch.empty(w.shape[0], t * w.shape[1], dtype=torch.float16, device=w.device)
    sdev = torch.full((1,), ent[0], dtype=torch.float32, device=w.device)
    _lib.check(_lib.lib().pp_split_with_scale_t(_p(w), w.numel(), t, _p(sdev), _p(hl), _lib.stream_ptr()), "pp_split_with_scale_t")
    if ent[1] is None:
        s2 = torch.empty(2, dtype=torch.float32, device=w.device)
        _lib.check(_lib.lib().pp_pow2_scale(_p(w), w.numel(), _p(s2), _lib.stream_ptr()), "pp_pow2_scale")
        host = torch.empty(2, dtype=torch.float32, pin_memory=True)
        host.copy_(s2, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ent[1] = (host, ev)
    return hl, ent[0]


class Split:
    """An activation that exists only as the f16x3 engine's "hl" operand: fp16 (rows, 2C), per 8 channels the 8 hi
    terms then the 8 lo terms (include/picopose_hip.h).  Producers (layernorm, attention, a GEMM epilogue) write it
    directly, so the consuming GEMM needs no split pass and the fp32 tensor is never stored."""
    __slots__ = ("hl", "image", "terms")

    def __init__(self, hl, t=None):
        self.hl = hl
        self.image = None  # (B, H, W) when the rows are the pixels of NHWC images (operand of a convolution)
        self.terms = t or terms()   # 2: hl format (rows, 2C); 1: h format (rows, C) — ops.PRECISION = "f16"

    @property
    def shape(self):
        return torch.Size((self.hl.shape[0], self.hl.shape[1] // self.terms))

    @property
    def device(self):
        return self.hl.device

    @staticmethod
    def empty(rows, C, device):
        return Split(torch.empty(rows, terms() * C, dtype=torch.float16, device=device))

    def cols(self, col0, c):
        """The fp16 columns of the buffer that hold operand columns col0 .. col0 + c (col0, c multiples of 8)."""
        return self.hl[:, self.terms * col0:self.terms * (col0 + c)]

    def col_ptr(self, col0):
        """Device address of operand column col0's group (col0 % 8 == 0)."""
        return self.hl.data_ptr() + 2 * self.terms * col0


def _split_ok(C):
    return presplit() and C % 8 == 0


def split_activation(x, B, P, C, batch_stride, row_stride, relu=False, into=None):
    """Pre-split an activation operand (B, P, C) into a contiguous hl buffer (B*P, 2C) for the f16x3 engine.
    into = (Split over (B*P, Ctot), col0): write columns col0 .. col0 + C of that wider operand instead (col0 % 8 == 0)."""
    t = terms()
    if into is not None:
        tgt, col0 = into
        ctot = tgt.shape[1]
        assert tgt.shape[0] == B * P and col0 % 8 == 0 and col0 + C <= ctot
        _lib.check(_lib.lib().pp_split_activation_t(_p(x), batch_stride, B, P, row_stride, C, int(relu),
                                                    tgt.col_ptr(col0), ctot, t, _lib.stream_ptr()), "pp_split_activation_t")
        if CHECK_SATURATION:
            _chk(tgt.cols(col0, C), "pp_split_activation_t")
        return None
    hl = torch.empty(B * P, t * C, dtype=torch.float16, device=x.device)
    _lib.check(_lib.lib().pp_split_activation_t(_p(x), batch_stride, B, P, row_stride, C, int(relu), _p(hl), C, t,
                                                _lib.stream_ptr()), "pp_split_activation_t")
    return _chk(hl, "pp_split_activation_t")


def hl_patch_columns(x, tgt, col0):
    """Columns col0 .. col0 + c - 1 of the Split `tgt` <- x (..., c) fp32 (contiguous): the narrow member of an operand concat."""
    c = x.shape[-1]
    rows = x.numel() // c
    assert x.is_contiguous() and rows == tgt.shape[0] and col0 + c <= tgt.shape[1]
    _lib.check(_lib.lib().pp_hl_patch_columns_t(_p(x), c, c, rows, _p(tgt.hl), tgt.shape[1], col0, tgt.terms, _lib.stream_ptr()),
               "pp_hl_patch_columns_t")
    if CHECK_SATURATION:   # (the 8-channel groups holding the patched columns; their other members are already written)
        _chk(tgt.cols(col0 // 8 * 8, -(-(col0 + c) // 8) * 8 - col0 // 8 * 8), "pp_hl_patch_columns_t")


def _can_presplit(x, K, C, *strides):
    return (presplit() and K % 8 == 0 and C % 8 == 0 and x.data_ptr() % 16 == 0
            and all(st % 4 == 0 for st in strides))


def _weight_args(w, K, cache=True):
    """desc fields for a weight operand under the current precision (pre-split operand when it is aligned).
    cache: True (a parameter: remembered per version, scale read back once) | "dev" (a TRAINED parameter: remembered per version, scale
    left on the device) | False (transient: split now, one host wait for its scale) | (owner, tag) (derived from the parameter `owner`
    every step: split_weight_tracked, no host wait)."""
    if isinstance(cache, tuple) and os.environ.get("PP_TRACK_SCALE", "1") == "0":
        cache = False                      # (A/B switch: the scale of every derived weight read back at once, one host wait each)
    if presplit() and K % 8 == 0 and w.data_ptr() % 16 == 0:
        if cache == "dev":                 # a trained parameter: scale on the device, taken by the launch through alpha_dev
            hl, s2 = split_weight_dev(w)
            return dict(prec=_PREC[PRECISION], B_hl=_p(hl), b_scale=1.0, alpha_dev=_p(s2[1:2]), _hl=(hl, s2))
        hl, scale = split_weight_tracked(w, *cache) if isinstance(cache, tuple) else split_weight(w, cache)
        return dict(prec=_PREC[PRECISION], B_hl=_p(hl), b_scale=scale, _hl=hl)
    return dict(prec=_fly_prec())


def _fly_prec():
    """Arithmetic of the kernels that take fp32 operands (tiny / unaligned layers, batched products): exact fp32 MFMA, or the
    f16x3 split on the fly — also in "f16" mode, whose single-term kernels exist for pre-split operands only."""
    return 0 if PRECISION == "f32" else 1


def _fly_args(wargs):
    """`wargs` for a launch whose A operand is NOT pre-split: the on-the-fly kernels read fp32 weights (f16 mode: the h-format
    weights cannot be used there) or hl weights (f16x3)."""
    if PRECISION == "f16":
        return dict(prec=1)
    if "alpha_dev" in wargs:     # device-scaled weights (cache="dev") belong to the pre-split kernels: these launches read the fp32 weights
        return dict(prec=_fly_prec())
    return wargs


def _p(t):
    return t.data_ptr() if t is not None else None


def _run(d, what="pp_gemm", written=None):
    """written: the Split this launch writes operand columns of (verified under CHECK_SATURATION)."""
    _lib.check(_lib.lib().pp_gemm(ctypes.byref(d), _lib.stream_ptr()), what)
    if written is not None:
        _chk(written.hl, f"{what} epilogue (M={d.M}, N={d.N}, K={d.K})")


def _desc(**kw):
    d = PpGemmDesc()
    d.batch0 = d.batch1 = 1
    d.alpha = 1.0
    for k, v in kw.items():
        if not k.startswith("_"):     # (_hl: the operand tensor itself, carried so that it outlives the enqueue)
            setattr(d, k, v)
    return d


def linear(x, weight, bias=None, act=None, gamma=None, residual=None, out=None, relu_in=False, out_split=False, cache_weight=True):
    """y = residual + gamma * act(x @ weight.T + bias); x (M,K) fp32 with row stride or a Split, weight (N,K).
    out_split (f16x3 engine only; ignored otherwise): return y as a Split for the next linear instead of fp32.
    cache_weight=False: `weight` is a transient matrix — its operand form is not remembered."""
    M, K = x.shape
    N = weight.shape[0]
    assert weight.shape[1] == K and weight.is_contiguous()
    wargs = _weight_args(weight, K, cache_weight)
    sargs, ret = {}, None
    if out_split and "B_hl" in wargs and _split_ok(N) and out is None:
        ret = Split.empty(M, N, x.device)
        sargs = dict(C_hl=_p(ret.hl), ldc_h=N)
    elif out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    ldc = N if out is None else out.stride(0)
    if out is not None:
        assert out.stride(1) == 1
        ret = out
    if residual is not None:
        assert tuple(residual.shape) == (M, N) and residual.stride(1) == 1 and residual.stride(0) == ldc
    if isinstance(x, Split):
        assert "B_hl" in wargs
        _run(_desc(A_hl=_p(x.hl), B=_p(weight), C=_p(out), bias=_p(bias), gamma=_p(gamma), residual=_p(residual),
                   M=M, N=N, K=K, lda=K, ldb=K, ldc=ldc, act=ACT[act], **wargs, **sargs), written=ret if sargs else None)
        return ret
    assert x.stride(1) == 1
    if "B_hl" in wargs and N > 64 and _can_presplit(x, K, K, x.stride(0)) and M * K < 2 ** 31:
        hl = split_activation(x, 1, M, K, 0, x.stride(0), relu=relu_in)          # every column tile reuses the split
        _run(_desc(A_hl=_p(hl), B=_p(weight), C=_p(out), bias=_p(bias), gamma=_p(gamma), residual=_p(residual),
                   M=M, N=N, K=K, lda=K, ldb=K, ldc=ldc, act=ACT[act], **wargs, **sargs), written=ret if sargs else None)
        return ret
    if sargs:  # the fp32-operand kernels write planes too, but keep this rare path simple: fp32 out + split pass
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
        ret, ldc, sargs = out, N, {}
    _run(_desc(A=_p(x), B=_p(weight), C=_p(out), bias=_p(bias), gamma=_p(gamma), residual=_p(residual), M=M, N=N, K=K,
               lda=x.stride(0), ldb=K, ldc=ldc, act=ACT[act], relu_in=int(relu_in), **_fly_args(wargs)))
    return ret


def split_scaled(x2d, s2):
    """x2d (rows, C) fp32 (row pitch free, C % 8 == 0) -> Split of s2[0] * x2d: the range-normalised operand of a backward product in ONE
    pass (the scale is a device scalar from autograd._pow2_scale; no scaled fp32 copy is made)."""
    rows, C = x2d.shape
    assert x2d.stride(1) == 1
    sp = Split.empty(rows, C, x2d.device)
    _lib.check(_lib.lib().pp_split_scaled_t(_p(x2d), rows, x2d.stride(0), C, _p(s2), _p(sp.hl), sp.terms, _lib.stream_ptr()), "pp_split_scaled_t")
    return sp


def split_transposed(x2d, s2=None, k_pad=None):
    """x2d (rows, cols) fp32 (row pitch free; rows % 8 == 0) -> Split (cols, rows) of s2[0] * x2d^T (s2 None: 1): the K-major operands of
    dW = dz^T x without a transposed fp32 copy.  k_pad >= rows (% 8 == 0): Split (cols, k_pad), the k past `rows` zero (K slices)."""
    rows, cols = x2d.shape
    assert x2d.stride(1) == 1 and rows % 8 == 0
    k_pad = rows if k_pad is None else k_pad
    sp = Split.empty(cols, k_pad, x2d.device)
    _lib.check(_lib.lib().pp_split_transpose_ld(_p(x2d), rows, cols, x2d.stride(0), _p(s2), _p(sp.hl), k_pad, sp.terms, _lib.stream_ptr()),
               "pp_split_transpose_ld")
    return sp


KSPLIT = os.environ.get("PP_KSPLIT", "1") != "0"


def ksplit_choice(M, N, K, can_pad=True):
    """(S, K_pad) for a product of two K-major operands with few output tiles over a long K (weight gradients): S K-slices run as S x tiles
    work items of ONE engine launch (PpGemmDesc.ksplit), K padded with zeros to a multiple of 64 S when the producer can
    (can_pad).  Chosen from the shape only (a cost model in units of k per work item: rounds over the 256 CUs x (slice length + a
    tile's fixed cost)), so the summation order is a function of the shape.  S = 1: no slices."""
    if not KSPLIT or M % 256 != 0 or N % 8 != 0 or PRECISION != "f16x3":     # (the sliced kernel exists for the hl format and the vector epilogue)
        return 1, K
    t = -(-M // 256) * -(-N // 256)
    base = -(-t // 256) * (K + 512)
    best = (base, 1, K)
    for S in range(2, 65):
        kp = -(-K // (64 * S)) * 64 * S
        if (kp != K and not can_pad) or kp // S < 256 or S * M * N * 4 > (1 << 28):
            continue
        cost = -(-S * t // 256) * (kp // S + 512)
        if cost < best[0]:
            best = (cost, S, kp)
    return (best[1], best[2]) if best[0] <= 0.8 * base else (1, K)      # the partial sums have to be paid for: 20 % at least


def operands_ok(M, N, K):
    """Shapes the pre-split engine takes for a product of two transient operands (32-bit byte offsets into either operand)."""
    return presplit() and PRECISION == "f16x3" and M >= 64 and N >= 64 and K % 8 == 0 and M * K < 2 ** 30 and N * K < 2 ** 30


def matmul_operands(A, Bt, alpha_dev=(), out=None, ksplit=1):
    """A (M,K) @ Bt (N,K)^T for two Splits (activation scale): fp32 (M,N).  alpha_dev: up to two device scalars the result is
    multiplied by inside the launch (PpGemmDesc.alpha_dev: the inverse range scales of the operands).  ksplit = S > 1: the S K-slices
    as one launch into (S, M, N), added in index order (K % (64 S) == 0, M % 256 == 0: ksplit_choice)."""
    (M, K), (N, K2) = A.shape, Bt.shape
    assert K == K2
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    assert out.stride(1) == 1
    ad = list(alpha_dev) + [None, None]
    if ksplit > 1:
        part = torch.empty(ksplit, M, N, dtype=torch.float32, device=A.device)
        _run(_desc(A_hl=_p(A.hl), B_hl=_p(Bt.hl), b_scale=4.0, C=_p(part), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, prec=_PREC[PRECISION],
                   alpha_dev=_p(ad[0]), alpha_dev2=_p(ad[1]), ksplit=ksplit, _keep=(A.hl, Bt.hl, part)))
        tgt = out if out.is_contiguous() else torch.empty(M, N, dtype=torch.float32, device=A.device)
        _lib.check(_lib.lib().pp_sum_slices(_p(part), ksplit, M, N, None, 0, _p(tgt), _lib.stream_ptr()), "pp_sum_slices")
        if tgt is not out:
            out.copy_(tgt)
        return out
    _run(_desc(A_hl=_p(A.hl), B_hl=_p(Bt.hl), b_scale=4.0, C=_p(out), M=M, N=N, K=K, lda=K, ldb=K, ldc=out.stride(0), prec=_PREC[PRECISION],
               alpha_dev=_p(ad[0]), alpha_dev2=_p(ad[1]), _keep=(A.hl, Bt.hl)))
    return out


def matmul_nt_presplit(a, bt):
    """a (M,K) @ bt (N,K)^T with BOTH operands split as activations (scale PP_A_SCALE) — the pre-split engine for products of two
    transient matrices (the backward GEMMs of picopose_amd/autograd.py: their operands are range-normalised first).  Unlike
    `linear`, nothing is cached and nothing is read back.  Falls to the on-the-fly kernel when the shapes do not qualify."""
    M, K = a.shape
    N = bt.shape[0]
    assert bt.shape[1] == K and a.is_contiguous() and bt.is_contiguous()
    out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    if presplit() and PRECISION == "f16x3" and N >= 64 and M >= 64 and _can_presplit(a, K, K, K) and bt.data_ptr() % 16 == 0 \
            and M * K < 2 ** 30 and N * K < 2 ** 30:
        ah = split_activation(a, 1, M, K, 0, K)
        bh = split_activation(bt, 1, N, K, 0, K)
        _run(_desc(A_hl=_p(ah), B=_p(bt), B_hl=_p(bh), b_scale=4.0, C=_p(out), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, prec=_PREC[PRECISION]))
        return out
    _run(_desc(A=_p(a), B=_p(bt), C=_p(out), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, prec=_fly_prec()))
    return out


def linear_splitk(x, weight, bias=None, act=None, slices=32):
    """act(x @ weight.T + bias) for a FEW rows against a long K (stage 2's fc1: 160 x 16384 x 1024 is 8 tiles as one GEMM):
    the K axis is cut into `slices` equal parts that run as ONE batched GEMM (slices x more workgroups) and are then added in
    index order.  The same slicing whatever the number of rows, so a row's result does not depend on the batch it is in."""
    M, K = x.shape
    N = weight.shape[0]
    assert x.is_contiguous() and weight.is_contiguous() and weight.shape[1] == K
    if K % slices != 0 or (K // slices) % 8 != 0:
        return linear(x, weight, bias, act=act)
    kc = K // slices
    part = bmm_nt(x.view(M, slices, kc).permute(1, 0, 2)[None], weight.view(N, slices, kc).permute(1, 0, 2)[None])
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().pp_sum_slices(_p(part), slices, M, N, _p(bias), ACT[act], _p(out), _lib.stream_ptr()), "pp_sum_slices")
    return out


def bmm_nt(a, b, alpha=1.0, out=None):
    """out[z0,z1] = alpha * a[z0,z1] @ b[z0,z1].T for 4-D strided views (Z0,Z1,M,K) x (Z0,Z1,N,K)."""
    Z0, Z1, M, K = a.shape
    N = b.shape[2]
    assert a.stride(3) == 1 and b.stride(3) == 1
    if out is None:
        out = torch.empty(Z0, Z1, M, N, dtype=torch.float32, device=a.device)
    _run(_desc(A=_p(a), B=_p(b), C=_p(out), M=M, N=N, K=K, lda=a.stride(2), ldb=b.stride(2), ldc=out.stride(2),
               batch0=Z0, batch1=Z1, a_bs0=a.stride(0), a_bs1=a.stride(1), b_bs0=b.stride(0), b_bs1=b.stride(1),
               c_bs0=out.stride(0), c_bs1=out.stride(1), alpha=float(alpha), prec=_fly_prec()))
    return out


def bmm_nn(a, b, out, alpha=1.0):
    """out[z0,z1] = alpha * a[z0,z1] @ b[z0,z1] for (Z0,Z1,M,K) x (Z0,Z1,K,N) strided views (b: n contiguous)."""
    Z0, Z1, M, K = a.shape
    N = b.shape[3]
    assert a.stride(3) == 1 and b.stride(3) == 1 and out.stride(3) == 1
    _run(_desc(A=_p(a), B=_p(b), C=_p(out), M=M, N=N, K=K, lda=a.stride(2), ldb=b.stride(2), ldc=out.stride(2), b_kn=1,
               batch0=Z0, batch1=Z1, a_bs0=a.stride(0), a_bs1=a.stride(1), b_bs0=b.stride(0), b_bs1=b.stride(1),
               c_bs0=out.stride(0), c_bs1=out.stride(1), alpha=float(alpha), prec=_fly_prec()))
    return out


def pack_conv_weight(w, cin_pad=None):
    """(Cout, Cin, KH, KW) torch layout -> (Cout, KH*KW*Cin) rows in the engine's k order.  cin_pad > Cin appends
    zero input channels (the matching activation buffer carries zero-filled pad channels), which keeps layers with
    odd channel counts (3, 2, 25, 50, 75) on the vector-load / split-precision path."""
    co, ci, kh, kw = w.shape
    w = w.permute(0, 2, 3, 1)
    if cin_pad is not None and cin_pad > ci:
        w = torch.cat([w, w.new_zeros(co, kh, kw, cin_pad - ci)], dim=3)
        ci = cin_pad
    return w.reshape(co, kh * kw * ci).contiguous()


def pack_convT_weight(w, bias):
    """ConvTranspose2d(kernel = stride = r) weight (Cin, Cout, r, r) -> ((r*r*Cout, Cin), bias tiled r*r)."""
    ci, co, r, r2 = w.shape
    assert r == r2
    wp = w.permute(2, 3, 1, 0).reshape(r * r * co, ci).contiguous()
    return wp, (bias.repeat(r * r).contiguous() if bias is not None else None)


def split_image(x, relu=False):
    """NHWC image (B,H,W,C) (C % 8 == 0, channel-contiguous rows, free batch stride) -> Split with .image = (B,H,W):
    the operand of several convolutions split once (f16x3 engine only; otherwise x itself)."""
    B, H, W, C = x.shape
    if not (_split_ok(C) and x.stride(3) == 1 and x.stride(1) == W * x.stride(2) and x.data_ptr() % 16 == 0
            and x.stride(2) % 4 == 0 and x.stride(0) % 4 == 0 and B * H * W * C < 2 ** 30):
        return x
    sp = Split(split_activation(x, B, H * W, C, x.stride(0), x.stride(2), relu=relu))
    sp.image = (B, H, W)
    return sp


def conv2d(x, wp, bias, ksize, stride=1, pad=0, act=None, relu_in=False, residual=None, out=None, cin=None,
           residual2=None, out_split=False, split_relu=False, also_split=None, hl_into=None, cache_weight=True, in_cols=None,
           alpha_dev=(), wino=False, wino_next=False):
    """NHWC convolution. x (B,H,W,Cx) (channel-contiguous, may be a channel slice: cin <= Cx stride) or a Split
    carrying .image (a pre-split operand: no split pass, the producer has already applied any input ReLU),
    wp (Cout, k*k*cin) from pack_conv_weight.  out may be a channel slice of a wider NHWC buffer.
    out_split (f16x3 engine, no `out`): return the result as a Split (with split_relu: of max(result, 0), the next
    layer's input ReLU folded in) instead of an fp32 tensor.
    also_split ("relu" | "plain"; f16x3 engine): return the fp32 tensor AND attach its operand form as `._hl` /
    `._hl_relu` (a Split) for a following convolution, both written by this epilogue.
    hl_into = (Split over (B*Ho*Wo, Ctot), col0): write the result as operand columns col0.. of that shared Split
    (channel concatenation of operands; Cout % 8 == 0, col0 % 8 == 0) — and, if `out` is given, as fp32 there too;
    returns `out` (None without it).
    in_cols = (col0, c) with a Split input: the convolution reads operand columns col0 .. col0 + c of the wider Split (a channel
    slice of a shared operand: two layers fused along N hand their halves to their successors without a copy).
    alpha_dev: up to two device scalars the accumulated product is multiplied by inside the launch (PpGemmDesc.alpha_dev: the inverse
    range scale of a backward operand) — pre-split path only.
    wino (f16x3 engine, 3x3 / stride 1 / pad 1 on an operand image): run the layer by Winograd F(4x4, 3x3) (`_conv3x3_winograd4`).
    wino_next (True | "relu"): the result feeds ONLY another 3x3 / stride 1 / pad 1 convolution of the same kind ("relu": through that layer's
    input ReLU) — where both run by Winograd, return the NEXT layer's Winograd input (f16x3 engine, with wino and out_split: a WinoInput4;
    strict-fp32 mode: a WinoInput) with the output transform chained into its input transform (csrc/pp_winograd.hip wino4_chain_kernel /
    wino2_chain_kernel: the hidden map is never stored); ignored where the chain does not apply (then the usual result comes back, which the
    next conv2d takes just the same)."""
    if isinstance(x, WinoInput):      # the shared / chained Winograd input of 3x3 convolutions (strict-fp32 mode)
        B, H, W, Cx = x.geom
        assert ksize == 3 and stride == 1 and pad == 1 and cin in (None, Cx) and wp.shape[1] == 9 * Cx and hl_into is None
        assert not (relu_in and not x.relu), "the shared / chained Winograd input was made without the input ReLU this layer asks for"
        return _conv3x3_winograd(x, wp, bias, B, H, W, Cx, Cx, wp.shape[0], act, x.relu, residual, residual2, out, chain_next=wino_next)
    if isinstance(x, WinoInput4):     # the shared F(4x4, 3x3) input of several 3x3 convolutions (f16x3 engine)
        B, H, W, Cx = x.geom
        assert ksize == 3 and stride == 1 and pad == 1 and cin in (None, Cx) and wp.shape[1] == 9 * Cx and hl_into is None and not relu_in
        return _conv3x3_winograd4(x, wp, bias, B, H, W, Cx, wp.shape[0], act, residual, residual2, out, out_split, split_relu, also_split, wino_next)
    xs = x if isinstance(x, Split) else None
    a_ptr = None
    if xs is not None and in_cols is not None:
        col0, cin = in_cols
        (B, H, W), Cx = xs.image, xs.shape[1]
        assert not relu_in and col0 % 8 == 0 and cin % 8 == 0 and col0 + cin <= Cx and hl_into is None
        a_ptr, ld_in = xs.col_ptr(col0), Cx
    elif xs is not None:
        (B, H, W), Cx = xs.image, xs.shape[1]
        assert not relu_in and cin in (None, Cx)
        ld_in = Cx
    else:
        B, H, W, Cx = x.shape
        assert x.stride(3) == 1
        ld_in = x.stride(2)
        assert x.stride(1) == W * ld_in  # images may be spaced apart (tokens with a cls row): x.stride(0) is free
    cin = cin or Cx
    Cout = wp.shape[0]
    assert wp.shape[1] == ksize * ksize * cin
    Ho = (H + 2 * pad - ksize) // stride + 1
    Wo = (W + 2 * pad - ksize) // stride + 1
    dev = xs.device if xs is not None else x.device
    if (xs is not None and a_ptr is None and xs.terms == 2 and Cout <= 2 and ksize in (1, 3) and stride == 1 and pad == ksize // 2 and act is None and out is None
            and not out_split and also_split is None and hl_into is None and residual2 is None and Cx % 32 == 0 and W in (16, 32, 64)
            and H % (256 // W) == 0 and wp.dtype == torch.float32 and wp.is_contiguous()
            and (residual is None or (residual.is_contiguous() and tuple(residual.shape) == (B, H, W, Cout)))
            and os.environ.get("PP_CONV_NARROW", "1") != "0"):
        # one- / two-channel predict layers of the decoder heads: a direct convolution on the operand (pp_conv_narrow_hl)
        # instead of a GEMM tile padded from 2 to 64 columns
        out = torch.empty(B, H, W, Cout, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().pp_conv_narrow_hl(_p(xs.hl), Cx, B, H, W, Cx, _p(wp), _p(bias), ksize, Cout, _p(residual), _p(out),
                                                _lib.stream_ptr()), "pp_conv_narrow_hl")
        return out
    if (xs is None and PRECISION == "f32" and Cout <= 2 and ksize in (1, 3) and stride == 1 and pad == ksize // 2 and act is None and out is None
            and not relu_in and residual2 is None and cin == Cx and Cx % 32 == 0 and W in (16, 32, 64) and H % (256 // W) == 0
            and ld_in % 4 == 0 and x.data_ptr() % 16 == 0 and x.stride(0) == H * W * ld_in and wp.dtype == torch.float32 and wp.is_contiguous()
            and (residual is None or (residual.is_contiguous() and tuple(residual.shape) == (B, H, W, Cout)))
            and os.environ.get("PP_CONV_NARROW", "1") != "0"):
        # the same predict layers in the strict-fp32 mode: a direct fp32 convolution on the NHWC map (pp_conv_narrow_f32)
        out = torch.empty(B, H, W, Cout, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().pp_conv_narrow_f32(_p(x), ld_in, B, H, W, Cx, _p(wp), _p(bias), ksize, Cout, _p(residual), _p(out),
                                                 _lib.stream_ptr()), "pp_conv_narrow_f32")
        return out
    if (xs is None and ksize == 3 and stride == 1 and pad == 1 and Cout >= 32 and act in (None, "none", "relu", "leaky01")
            and _winograd_ok(B, H, W, cin, ld_in, x) and wp.dtype == torch.float32 and wp.is_contiguous()
            and hl_into is None and cache_weight is True):     # (out_split / also_split: f16x3-engine hints, ignored in this mode)
        return _conv3x3_winograd(x, wp, bias, B, H, W, cin, ld_in, Cout, act, relu_in, residual, residual2, out, chain_next=wino_next)
    if (wino and xs is not None and ksize == 3 and stride == 1 and pad == 1 and xs.terms == 2 and hl_into is None and not alpha_dev and cache_weight is True
            and act in (None, "none", "relu", "leaky01") and wp.dtype == torch.float32 and wp.is_contiguous()
            and _winograd4_ok(B, H, W, cin, Cout)):
        src = xs if a_ptr is None else (xs, in_cols[0])
        return _conv3x3_winograd4(src, wp, bias, B, H, W, cin, Cout, act, residual, residual2, out, out_split, split_relu, also_split, wino_next)
    wargs = _weight_args(wp, ksize * ksize * cin, cache_weight)   # (cache_weight=False: a transient packed weight of the training graph)
    presplit = xs is not None or ("B_hl" in wargs and (Cout > 64 or ksize > 1)
                                  and _can_presplit(x, ksize * ksize * cin, cin, ld_in, x.stride(0)) and B * H * W * cin < 2 ** 30)
    sargs, ret = {}, None
    if hl_into is not None:
        tgt, col0 = hl_into
        assert presplit and Cout % 8 == 0 and col0 % 8 == 0 and residual is None and residual2 is None
        ctot = tgt.shape[1]
        assert tgt.shape[0] == B * Ho * Wo and col0 + Cout <= ctot
        ldc = Cout
        if out is not None:
            ldc = out.stride(2)
            assert out.stride(3) == 1 and out.stride(1) == Wo * ldc and out.stride(0) == Ho * Wo * ldc
        hl = xs.hl if xs is not None else split_activation(x, B, H * W, cin, x.stride(0), ld_in, relu=relu_in)
        _run(_desc(A_hl=_p(hl), B=_p(wp), C=_p(out), bias=_p(bias), conv_bstride=H * W * cin, M=B * Ho * Wo, N=Cout,
                   K=ksize * ksize * cin, lda=cin, ldb=wp.shape[1], ldc=ldc, act=ACT[act], conv_kh=ksize, conv_kw=ksize,
                   conv_cin=cin, conv_stride=stride, conv_pad=pad, conv_h=H, conv_w=W, conv_ho=Ho, conv_wo=Wo,
                   C_hl=tgt.col_ptr(col0), ldc_h=ctot, c_relu=int(split_relu), **wargs),
             written=Split(tgt.cols(col0, Cout), tgt.terms) if CHECK_SATURATION else None)
        return out
    # (operand-only output with residuals: the epilogue reads them with the row pitch ldc = Cout — contiguous (B, Ho, Wo, Cout) maps)
    if (out_split and out is None and presplit and _split_ok(Cout)
            and all(r_ is None or (r_.is_contiguous() and tuple(r_.shape) == (B, Ho, Wo, Cout)) for r_ in (residual, residual2))):
        ret = Split.empty(B * Ho * Wo, Cout, dev)
        ret.image = (B, Ho, Wo)
        sargs = dict(C_hl=_p(ret.hl), ldc_h=Cout, c_relu=int(split_relu))
        ldc = Cout
    else:
        if out is None:
            out = torch.empty(B, Ho, Wo, Cout, dtype=torch.float32, device=dev)
        ldc = out.stride(2)
        assert out.stride(3) == 1 and out.stride(1) == Wo * ldc and out.stride(0) == Ho * Wo * ldc
        ret = out
    for r_ in (residual, residual2):
        if r_ is not None and out is not None:
            assert r_.stride() == out.stride()
    extra = None
    if also_split is not None and presplit and _split_ok(Cout) and not sargs and out.is_contiguous():
        extra = Split.empty(B * Ho * Wo, Cout, dev)
        extra.image = (B, Ho, Wo)
        sargs = dict(C_hl=_p(extra.hl), ldc_h=Cout, c_relu=int(also_split == "relu"))
        setattr(out, "_hl_relu" if also_split == "relu" else "_hl", extra)
    if presplit:
        assert "B_hl" in wargs
        hl = xs.hl if xs is not None else split_activation(x, B, H * W, cin, x.stride(0), ld_in, relu=relu_in)  # once, not per tap / column tile
        ld_a = ld_in if a_ptr is not None else cin
        ad = list(alpha_dev) + [None, None]
        aargs = dict(alpha_dev=_p(ad[0]), alpha_dev2=_p(ad[1]))
        _run(_desc(A_hl=a_ptr if a_ptr is not None else _p(hl), B=_p(wp), C=_p(out), bias=_p(bias), residual=_p(residual),
                   residual2=_p(residual2), conv_bstride=H * W * ld_a, M=B * Ho * Wo, N=Cout, K=ksize * ksize * cin, lda=ld_a,
                   ldb=wp.shape[1], ldc=ldc, act=ACT[act], conv_kh=ksize, conv_kw=ksize, conv_cin=cin, conv_stride=stride,
                   conv_pad=pad, conv_h=H, conv_w=W, conv_ho=Ho, conv_wo=Wo, **wargs, **sargs, **aargs),
             written=(extra if extra is not None else ret) if sargs else None)
        return ret
    assert not alpha_dev
    _run(_desc(A=_p(x), B=_p(wp), C=_p(out), bias=_p(bias), residual=_p(residual), residual2=_p(residual2),
               conv_bstride=x.stride(0), M=B * Ho * Wo, N=Cout,
               K=ksize * ksize * cin, lda=ld_in, ldb=wp.shape[1], ldc=ldc, act=ACT[act], relu_in=int(relu_in),
               conv_kh=ksize, conv_kw=ksize, conv_cin=cin, conv_stride=stride, conv_pad=pad, conv_h=H, conv_w=W,
               conv_ho=Ho, conv_wo=Wo, **_fly_args(wargs)))
    return ret


# Winograd F(2x2, 3x3) for the large 3x3 convolutions of the strict-fp32 mode (csrc/pp_winograd.hip): 2.25 x fewer fp32 multiplications on
# matrix-bound layers.  PP_WINOGRAD=0 keeps the direct implicit-GEMM convolution; layers below WINOGRAD_MIN_PIXELS output pixels stay direct
# (the sixteen products run as one or two grouped launches of the engine, so 32 K pixels — 2 048 row tiles — already fill the chip; measured at
# configs[2], ms per exact-mode step: threshold 128 K 181.3, 32 K 177.8, 8 K 177.4).
WINOGRAD = os.environ.get("PP_WINOGRAD", "1") != "0"
WINOGRAD_MIN_PIXELS = int(os.environ.get("PP_WINOGRAD_MIN_PIXELS", str(32 * 1024)))
_wino_cache = {}


def winograd_weight(wp, cin):
    """V (16, Cout, Cin) = G g G^T of a packed 3x3 weight (Cout, 9 Cin): once per tensor version (keyed like the split caches)."""
    key = (wp.data_ptr(), tuple(wp.shape), "wino")
    hit = _wino_cache.get(key)
    if hit is None or hit[2] != wp._version:
        Cout = wp.shape[0]
        V = hit[0] if hit is not None else torch.empty(16, Cout, cin, dtype=torch.float32, device=wp.device)
        _lib.check(_lib.lib().pp_winograd_weight_f32(_p(wp), Cout, cin, wp.shape[1], _p(V), _lib.stream_ptr()), "pp_winograd_weight_f32")
        if len(_wino_cache) > 1024:
            _wino_cache.clear()
        hit = _wino_cache[key] = (V, wp, wp._version)
    return hit[0]


class WinoInput:
    """The Winograd input transform U (16, P, C) of an fp32 NHWC map, made once for SEVERAL 3x3 convolutions that read the same map (the
    two heads of the flow decoder read one 640-channel input: raft_decoder.py:251-289) — `ops.winograd_shared(x)`; `conv2d` takes it in
    place of `x`.  The caller guarantees that the map is not rewritten between the transform and its last use."""
    __slots__ = ("U", "geom", "relu")

    def __init__(self, U, geom, relu):
        self.U, self.geom, self.relu = U, geom, relu


def _winograd_ok(B, H, W, cin, ld_in, x):
    P = B * (H // 2) * (W // 2)      # (P % 256 != 0: the products cannot share a launch — sixteen small launches pay only on larger maps)
    return (PRECISION == "f32" and WINOGRAD and H % 2 == 0 and W % 2 == 0 and cin % 4 == 0
            and B * H * W >= (WINOGRAD_MIN_PIXELS if P % 256 == 0 else 4 * WINOGRAD_MIN_PIXELS) and ld_in % 4 == 0 and x.data_ptr() % 16 == 0 and x.stride(0) % 4 == 0 and not torch.is_grad_enabled())


def winograd_shared(x, relu=False, cout=None):
    """x (B,H,W,C) fp32 NHWC (channel-contiguous) -> WinoInput when the strict-fp32 mode would run its 3x3 convolutions by Winograd
    (otherwise x itself): the input transform of a map that several convolutions read, done once.  f16x3 engine: x a Split with .image and
    cout = the output channels of the convolutions that will read it -> WinoInput4 when F(4x4, 3x3) applies to them (else x itself)."""
    if isinstance(x, Split) and cout is not None and x.image is not None and x.terms == 2:
        B, H, W = x.image
        if not relu and _winograd4_ok(B, H, W, x.shape[1], cout):
            return WinoInput4(_winograd4_input(x.hl.data_ptr(), x.shape[1], B, H, W, x.shape[1], x.device), (B, H, W, x.shape[1]), x)
        return x
    if isinstance(x, (Split, WinoInput, WinoInput4)) or x.dim() != 4:
        return x
    B, H, W, C = x.shape
    if not (x.stride(3) == 1 and x.stride(1) == W * x.stride(2) and _winograd_ok(B, H, W, C, x.stride(2), x)):
        return x
    U = torch.empty(16, B * (H // 2) * (W // 2), C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().pp_winograd_input_f32(_p(x), x.stride(2), x.stride(0), B, H, W, C, int(relu), _p(U), _lib.stream_ptr()),
               "pp_winograd_input_f32")
    return WinoInput(U, (B, H, W, C), bool(relu))


WINO2_CHAIN = os.environ.get("PP_WINOGRAD_CHAIN", "1") != "0"


def _conv3x3_winograd(x, wp, bias, B, H, W, cin, ld_in, Cout, act, relu_in, residual, residual2, out, chain_next=False):
    """3x3 / stride 1 / pad 1 on an fp32 NHWC map (a channel slice is fine) by Winograd F(2x2, 3x3): input transform, 16 dense fp32
    products on the engine, output transform with bias / activation / residuals.  Everything fp32.  x: the map, or its WinoInput.
    chain_next (True | "relu"): return the WinoInput of the next 3x3 layer (Cout -> .) instead of the map (pp_winograd_chain_f32)."""
    P = B * (H // 2) * (W // 2)
    L = _lib.lib()
    V = winograd_weight(wp, cin)
    if isinstance(x, WinoInput):
        assert x.geom == (B, H, W, cin) and x.relu == bool(relu_in)
        U, dev = x.U, x.U.device
    else:
        dev = x.device
        U = torch.empty(16, P, cin, dtype=torch.float32, device=dev)
        _lib.check(L.pp_winograd_input_f32(_p(x), ld_in, x.stride(0), B, H, W, cin, int(relu_in), _p(U), _lib.stream_ptr()), "pp_winograd_input_f32")
    Y = torch.empty(16, P, Cout, dtype=torch.float32, device=dev)
    # the sixteen products as batches of one pp_gemm call each: the engine runs a batch as ONE persistent launch (row tiles of all its
    # products, include/picopose_hip.h grp_rows) while the stacked operand / result blocks stay inside its 32-bit byte offsets
    per = max(1, min(16, (0xF0000000 // (4 * P * max(cin, Cout))))) if P % 256 == 0 else 1    # (a row tile must lie inside one product)
    for x0 in range(0, 16, per):
        n = min(per, 16 - x0)
        _run(_desc(A=_p(U[x0]), B=_p(V[x0]), C=_p(Y[x0]), M=P, N=Cout, K=cin, lda=cin, ldb=cin, ldc=Cout, prec=0, batch0=n,
                   a_bs0=P * cin, b_bs0=Cout * cin, c_bs0=P * Cout))
    if (chain_next and WINO2_CHAIN and WINOGRAD and out is None and residual is None and residual2 is None and W in (16, 32, 64) and Cout % 32 == 0
            and B * H * W >= (WINOGRAD_MIN_PIXELS if P % 256 == 0 else 4 * WINOGRAD_MIN_PIXELS)):
        # the next layer's Winograd input straight from this layer's products: h = act(A^T Y A + bias) lives in LDS only
        U1 = torch.empty(16, P, Cout, dtype=torch.float32, device=dev)
        _lib.check(L.pp_winograd_chain_f32(_p(Y), B, H, W, Cout, _p(bias), ACT[act], int(chain_next == "relu"), _p(U1), _lib.stream_ptr()),
                   "pp_winograd_chain_f32")
        return WinoInput(U1, (B, H, W, Cout), chain_next == "relu")
    if out is None:
        out = torch.empty(B, H, W, Cout, dtype=torch.float32, device=dev)
    ldc = out.stride(2)
    assert out.stride(3) == 1 and out.stride(1) == W * ldc and out.stride(0) == H * W * ldc
    for r_ in (residual, residual2):
        if r_ is not None:
            assert r_.stride() == out.stride()
    _lib.check(L.pp_winograd_output_f32(_p(Y), B, H, W, Cout, _p(bias), ACT[act], _p(residual), _p(residual2), _p(out), ldc, _lib.stream_ptr()),
               "pp_winograd_output_f32")
    return out


# Winograd F(4x4, 3x3) on the f16x3 engine (round 6; csrc/pp_winograd.hip): four times fewer products on the wide 3x3 convolutions of the flow
# decoder's heads (640 -> 512, 512 -> 256: raft_decoder.py:251-289), transformed operands 2.25 x the map.  PP_WINOGRAD4=0 keeps them direct.
# Which layers: the caller says so (`conv2d(..., wino=True)` / `winograd_shared(.., cout=)`: the heads' two hidden layers, where the saved
# MFMAs outweigh the transform passes at every level — per-layer A/B in profiles/r06/wino4_layers.txt); the decision never looks at the
# batch, so a crop's result does not depend on the batch it is in (test_full_size_forward_is_batch_independent).
WINOGRAD4 = os.environ.get("PP_WINOGRAD4", "1") != "0"
WINO4_GROUPS_PER_LAUNCH = 36
WINO4_U_SCALE = 1.0 / 16.0        # operand scale of U = B^T d B (|U| <= 100 |d|): alpha = PP_A_SCALE / WINO4_U_SCALE = 64 undoes it


class WinoInput4:
    """The F(4x4, 3x3) input transform of an operand image — a Split (36 P, C) — made once for SEVERAL convolutions that read the same
    map (`ops.winograd_shared(Xs, cout=...)`); `conv2d` takes it in place of the Split.  `.src` keeps the source operand alive."""
    __slots__ = ("U", "geom", "src")

    def __init__(self, U, geom, src):
        self.U, self.geom, self.src = U, geom, src


def _winograd4_ok(B, H, W, cin, cout):
    P = B * (H // 4) * (W // 4)
    return (PRECISION == "f16x3" and WINOGRAD4 and H % 4 == 0 and W % 4 == 0 and cin % 8 == 0 and cout % 8 == 0
            and 36 * (P + 255) < 2 ** 31 and not torch.is_grad_enabled())


def _wino4_rows(P):
    """Rows of one frequency block: P tiles rounded up to the engine's row tile (a row tile of the grouped launch lies inside one block)."""
    return -(-P // 256) * 256


def winograd4_weight(wp, cin):
    """(V (36 Cout, Cin) fp32, its hl operand, power-of-two scale) = G g G^T of a packed 3x3 weight (Cout, 9 Cin), transformed and split
    once per tensor version (a new version re-uses the buffers; one host sync to read the scale back)."""
    key = (wp.data_ptr(), tuple(wp.shape), "wino4")
    hit = _wino_cache.get(key)
    if hit is None or hit[2] != wp._version:
        Cout = wp.shape[0]
        V = hit[0] if hit is not None else torch.empty(36 * Cout, cin, dtype=torch.float32, device=wp.device)
        hl = hit[3] if hit is not None else torch.empty(36 * Cout, 2 * cin, dtype=torch.float16, device=wp.device)
        _lib.check(_lib.lib().pp_winograd4_weight_f32(_p(wp), Cout, cin, wp.shape[1], _p(V), _lib.stream_ptr()), "pp_winograd4_weight_f32")
        scale = torch.empty(1, dtype=torch.float32, device=wp.device)
        _lib.check(_lib.lib().pp_split_weights_t(_p(V), V.numel(), 2, _p(hl), _p(scale), _lib.stream_ptr()), "pp_split_weights_t")
        if len(_wino_cache) > 1024:
            _wino_cache.clear()
        hit = _wino_cache[key] = (V, wp, wp._version, hl, float(scale.item()))
    return hit[0], hit[3], hit[4]


def _winograd4_input(src_ptr, ld_x, B, H, W, cin, dev):
    P = B * (H // 4) * (W // 4)
    Pp = _wino4_rows(P)
    # (pad rows: zeros, so that the products of the pad rows are finite — they are never read)
    U = Split((torch.empty if Pp == P else torch.zeros)(36 * Pp, 2 * cin, dtype=torch.float16, device=dev), 2)
    _lib.check(_lib.lib().pp_winograd4_input_hl(src_ptr, ld_x, H * W * ld_x, B, H, W, cin, 0, _p(U.hl), Pp, _lib.stream_ptr()), "pp_winograd4_input_hl")
    _chk(U.hl, "pp_winograd4_input_hl")
    return U


WINO4_CHAIN = os.environ.get("PP_WINOGRAD4_CHAIN", "1") != "0"
WINO4_CHAIN_WIDTHS = (32, 64)


def _wino4_products(U, V, vhl, vscale, Pp, cin, N):
    """Y (36, Pp, N) fp32 = U_xi (Pp, cin) V_xi (N, cin)^T for the 36 frequencies: grouped launches of the pre-split engine (as many
    frequencies per launch as keep the stacked operand / result blocks inside 32-bit byte offsets)."""
    Y = torch.empty(36, Pp, N, dtype=torch.float32, device=U.device)
    per = max(1, min(WINO4_GROUPS_PER_LAUNCH, 0xF0000000 // (4 * Pp * max(cin, N))))
    for x0 in range(0, 36, per):
        n = min(per, 36 - x0)
        _run(_desc(A_hl=U.hl.data_ptr() + x0 * Pp * cin * 4, B=_p(V), B_hl=vhl.data_ptr() + x0 * N * cin * 4, b_scale=vscale, C=_p(Y[x0]),
                   M=Pp, N=N, K=cin, lda=cin, ldb=cin, ldc=N, prec=_PREC["f16x3"], batch0=n, a_bs0=Pp * cin, b_bs0=N * cin,
                   c_bs0=Pp * N, alpha=PP_A_SCALE / WINO4_U_SCALE, _keep=(U, vhl, V)))
    return Y


def _wino4_finish(Y, col0, ldy, bias, B, H, W, Cout, act, residual, residual2, out, out_split, split_relu, also_split, chain_next):
    """Output side of a Winograd F(4x4) layer whose products are columns col0 .. col0 + Cout of Y (36, Pp, ldy): the chained transform into
    the next layer's WinoInput4, or the output transform into an operand and / or an fp32 map."""
    P = B * (H // 4) * (W // 4)
    Pp, dev, L = _wino4_rows(P), Y.device, _lib.lib()
    yp = Y.data_ptr() + 4 * col0
    ret, hl_t, ldc = None, None, 0
    # (W = 16: the chain's one workgroup per (image, slice) is four tile rows long and loses to the two separate kernels, 0.144 vs 0.108 ms)
    if (chain_next and WINO4_CHAIN and out_split and out is None and residual is None and residual2 is None and W in WINO4_CHAIN_WIDTHS and Cout % 32 == 0
            and _winograd4_ok(B, H, W, Cout, Cout)):
        # the next layer's Winograd input straight from this layer's products: h = act(A^T Y A + bias) lives in LDS only
        U1 = Split((torch.empty if Pp == P else torch.zeros)(36 * Pp, 2 * Cout, dtype=torch.float16, device=dev), 2)
        _lib.check(L.pp_winograd4_chain(yp, ldy, B, H, W, Cout, _p(bias), ACT[act], int(split_relu), _p(U1.hl), Pp, _lib.stream_ptr()), "pp_winograd4_chain")
        _chk(U1.hl, "pp_winograd4_chain")
        return WinoInput4(U1, (B, H, W, Cout), None)
    if out_split and out is None and residual is None and residual2 is None:
        hl_t = Split.empty(B * H * W, Cout, dev)
        hl_t.image = (B, H, W)
        ret, c_relu = hl_t, int(split_relu)
    else:
        if out is None:
            out = torch.empty(B, H, W, Cout, dtype=torch.float32, device=dev)
        ldc = out.stride(2)
        assert out.stride(3) == 1 and out.stride(1) == W * ldc and out.stride(0) == H * W * ldc
        for r_ in (residual, residual2):
            if r_ is not None:
                assert r_.stride() == out.stride()
        ret, c_relu = out, 0
        if also_split is not None and out.is_contiguous():
            hl_t = Split.empty(B * H * W, Cout, dev)
            hl_t.image = (B, H, W)
            c_relu = int(also_split == "relu")
            setattr(out, "_hl_relu" if also_split == "relu" else "_hl", hl_t)
    _lib.check(L.pp_winograd4_output(yp, ldy, B, H, W, Cout, _p(bias), ACT[act], _p(residual), _p(residual2), _p(out), ldc,
                                     _p(hl_t.hl) if hl_t is not None else None, Cout, c_relu, Pp, _lib.stream_ptr()), "pp_winograd4_output")
    if hl_t is not None:
        _chk(hl_t.hl, "pp_winograd4_output")
    return ret


def _conv3x3_winograd4(x, wp, bias, B, H, W, cin, Cout, act, residual, residual2, out, out_split, split_relu, also_split, chain_next=False):
    """3x3 / stride 1 / pad 1 on an operand image by Winograd F(4x4, 3x3): input transform (operand -> operand), 36 dense products on the
    pre-split engine as grouped launches, output transform with bias / activation (/ residuals) into an fp32 map and / or the next
    layer's operand.  x: a Split with .image, (Split, col0) for a channel slice of it, or its WinoInput4."""
    P = B * (H // 4) * (W // 4)
    V, vhl, vscale = winograd4_weight(wp, cin)
    if isinstance(x, WinoInput4):
        assert x.geom == (B, H, W, cin)
        U = x.U
    elif isinstance(x, tuple):
        U = _winograd4_input(x[0].col_ptr(x[1]), x[0].shape[1], B, H, W, cin, x[0].device)
    else:
        U = _winograd4_input(x.hl.data_ptr(), x.shape[1], B, H, W, cin, x.device)
    Pp = _wino4_rows(P)
    assert U.hl.shape[0] == 36 * Pp
    Y = _wino4_products(U, V, vhl, vscale, Pp, cin, Cout)
    return _wino4_finish(Y, 0, Cout, bias, B, H, W, Cout, act, residual, residual2, out, out_split, split_relu, also_split, chain_next)


WINO4_PAIR = os.environ.get("PP_WINOGRAD4_PAIR", "1") != "0"


def winograd4_weight_pair(wp_a, wp_b, cin):
    """(V (36 (Ca + Cb), Cin), its hl operand, scale) for TWO layers that read the same Winograd input: per frequency the rows of layer a, then
    those of layer b — one product with N = Ca + Cb serves both (the activation operand U is then read once).  Cached per version pair."""
    key = (wp_a.data_ptr(), wp_b.data_ptr(), tuple(wp_a.shape), tuple(wp_b.shape), "wino4pair")
    hit = _wino_cache.get(key)
    if hit is None or hit[2] != (wp_a._version, wp_b._version):
        Va, Vb = winograd4_weight(wp_a, cin)[0], winograd4_weight(wp_b, cin)[0]
        Ca, Cb = wp_a.shape[0], wp_b.shape[0]
        V = torch.cat([Va.view(36, Ca, cin), Vb.view(36, Cb, cin)], dim=1).reshape(36 * (Ca + Cb), cin).contiguous()
        hl = torch.empty(V.shape[0], 2 * cin, dtype=torch.float16, device=V.device)
        scale = torch.empty(1, dtype=torch.float32, device=V.device)
        _lib.check(_lib.lib().pp_split_weights_t(_p(V), V.numel(), 2, _p(hl), _p(scale), _lib.stream_ptr()), "pp_split_weights_t")
        hit = _wino_cache[key] = (V, (wp_a, wp_b), (wp_a._version, wp_b._version), hl, float(scale.item()))
    return hit[0], hit[3], hit[4]


def conv2d_wino_pair(x, layer_a, layer_b, act=None, out_split=False, split_relu=False, wino_next=False):
    """Two 3x3 / stride 1 / pad 1 convolutions (wp, bias) of the SAME operand image (the flow and certainty heads' first layers,
    flow_decoder.py:58-72) by Winograd F(4x4, 3x3) with ONE product per frequency over the concatenated filters (N = Ca + Cb): the input
    operand U is read once instead of twice (5.67 -> 5.48 ms for the 640 -> 512 pair at 64 x 64 x 160).  Every output element is the sum the
    separate products make (the engine's accumulation order does not depend on N) — unless the two filters' power-of-two operand scales
    differ from the concatenated tensor's (then within fp16 subnormals of the lo terms, as for stage3.FUSE_XHEADS).  Returns the two
    results (Splits, or WinoInput4s with wino_next); falls back to two conv2d calls where F(4x4) does not apply."""
    (wa, ba), (wb, bb) = layer_a, layer_b
    kw = dict(act=act, out_split=out_split, split_relu=split_relu, wino=True, wino_next=wino_next)
    src = x.src if isinstance(x, WinoInput4) else x
    if not (WINO4_PAIR and isinstance(x, (WinoInput4, Split)) and out_split):
        return conv2d(x, wa, ba, 3, pad=1, **kw), conv2d(x, wb, bb, 3, pad=1, **kw)
    if isinstance(x, WinoInput4):
        B, H, W, cin = x.geom
    else:
        (B, H, W), cin = src.image, src.shape[1]
    Ca, Cb = wa.shape[0], wb.shape[0]
    if not (_winograd4_ok(B, H, W, cin, Ca) and _winograd4_ok(B, H, W, cin, Cb) and wa.shape[1] == 9 * cin == wb.shape[1] and Ca % 8 == 0):
        return conv2d(x, wa, ba, 3, pad=1, **kw), conv2d(x, wb, bb, 3, pad=1, **kw)
    P = B * (H // 4) * (W // 4)
    Pp = _wino4_rows(P)
    U = x.U if isinstance(x, WinoInput4) else _winograd4_input(src.hl.data_ptr(), cin, B, H, W, cin, src.device)
    V, vhl, vscale = winograd4_weight_pair(wa, wb, cin)
    Y = _wino4_products(U, V, vhl, vscale, Pp, cin, Ca + Cb)
    fin = lambda col0, C, bias: _wino4_finish(Y, col0, Ca + Cb, bias, B, H, W, C, act, None, None, None, out_split, split_relu, None, wino_next)  # noqa: E731
    return fin(0, Ca, ba), fin(Ca, Cb, bb)


def conv_transpose2d(x, wp, bias_tiled, r, out_split=False):
    """ConvTranspose2d(kernel = stride = r, padding 0) on NHWC: (B,H,W,Cin) -> (B,H*r,W*r,Cout).
    out_split (f16x3 engine): return the result only as a Split with .image (the pixel-shuffle store of the epilogue
    writes operand groups) for the convolution that follows — the fp32 map is never stored."""
    xs = x if isinstance(x, Split) else None
    if xs is not None:
        (B, H, W), Cin = xs.image, xs.shape[1]
    else:
        B, H, W, Cin = x.shape
        assert x.is_contiguous()
    Cout = wp.shape[0] // (r * r)
    wargs = _weight_args(wp, Cin)
    dev = xs.device if xs is not None else x.device
    if xs is not None or ("B_hl" in wargs and _can_presplit(x, Cin, Cin) and x.numel() < 2 ** 31):
        assert "B_hl" in wargs
        hl = xs.hl if xs is not None else split_activation(x, 1, B * H * W, Cin, 0, Cin)
        if out_split and _split_ok(Cout):
            sp = Split.empty(B * H * r * W * r, Cout, dev)
            sp.image = (B, H * r, W * r)
            _run(_desc(A_hl=_p(hl), B=_p(wp), C=None, bias=_p(bias_tiled), M=B * H * W, N=r * r * Cout, K=Cin,
                       lda=Cin, ldb=Cin, ldc=Cout, shuffle_r=r, shuffle_h=H, shuffle_w=W, C_hl=_p(sp.hl), ldc_h=Cout, **wargs),
                 written=sp)
            return sp
        out = torch.empty(B, H * r, W * r, Cout, dtype=torch.float32, device=dev)
        _run(_desc(A_hl=_p(hl), B=_p(wp), C=_p(out), bias=_p(bias_tiled), M=B * H * W, N=r * r * Cout, K=Cin,
                   lda=Cin, ldb=Cin, ldc=Cout, shuffle_r=r, shuffle_h=H, shuffle_w=W, **wargs))
        return out
    out = torch.empty(B, H * r, W * r, Cout, dtype=torch.float32, device=x.device)
    _run(_desc(A=_p(x), B=_p(wp), C=_p(out), bias=_p(bias_tiled), M=B * H * W, N=r * r * Cout, K=Cin, lda=Cin, ldb=Cin,
               ldc=Cout, shuffle_r=r, shuffle_h=H, shuffle_w=W, **_fly_args(wargs)))
    return out


def attention(qkv, B, T, heads, hd, out_split=False):
    """qkv (B*T, 3*heads*hd) from the qkv linear (fp32, or a Split written by its epilogue) -> (B*T, heads*hd):
    softmax((q hd^-1/2) k^T) v per head, fused, in the engine's current arithmetic.  out_split: return the result as
    a Split (f16x3 engine only)."""
    sp = out = None
    dev = qkv.device
    if out_split and _split_ok(heads * hd):
        sp = Split.empty(B * T, heads * hd, dev)
    else:
        out = torch.empty(B * T, heads * hd, dtype=torch.float32, device=dev)
    if isinstance(qkv, Split):
        _lib.check(_lib.lib().pp_attention_t(_p(qkv.hl), qkv.terms, B, T, heads, hd, float(hd) ** -0.5, _p(out),
                                             _p(sp.hl) if sp else None, _lib.stream_ptr()), "pp_attention_t")
    else:
        assert qkv.is_contiguous()
        if sp is not None and sp.terms == 1:    # (f16 mode with an fp32 qkv: the fp32-input kernel writes hl operands only)
            sp, out = None, torch.empty(B * T, heads * hd, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().pp_attention_ex(_p(qkv), B, T, heads, hd, float(hd) ** -0.5, _fly_prec(), _p(out),
                                              _p(sp.hl) if sp else None, _lib.stream_ptr()), "pp_attention_ex")
    if sp is not None:
        _chk(sp.hl, "pp_attention")
    return sp if sp is not None else out


def layernorm(x, weight, bias, eps, out_split=False):
    """nn.LayerNorm over the last dim of (rows, C); out_split: return a Split (f16x3 engine only)."""
    rows, C = x.shape
    assert x.is_contiguous()
    if out_split and _split_ok(C):
        sp = Split.empty(rows, C, x.device)
        _lib.check(_lib.lib().pp_layernorm_t(_p(x), _p(weight), _p(bias), rows, C, float(eps), None, _p(sp.hl), sp.terms,
                                             _lib.stream_ptr()), "pp_layernorm_t")
        _chk(sp.hl, "pp_layernorm_t")
        return sp
    y = torch.empty_like(x)
    _lib.check(_lib.lib().pp_layernorm(_p(x), _p(weight), _p(bias), rows, C, float(eps), _p(y), _lib.stream_ptr()),
               "pp_layernorm")
    return y


def softmax_rows_(x):
    """In-place softmax over the last dim of a contiguous (..., n) tensor."""
    assert x.is_contiguous()
    n = x.shape[-1]
    _lib.check(_lib.lib().pp_softmax_rows(_p(x), x.numel() // n, n, n, _lib.stream_ptr()), "pp_softmax_rows")
    return x


def groupnorm(x, weight, bias, groups, eps=1e-5, relu=False):
    B, H, W, C = x.shape
    assert x.is_contiguous()
    y = torch.empty_like(x)
    _lib.check(_lib.lib().pp_groupnorm_nhwc(_p(x), _p(weight), _p(bias), B, H * W, C, groups, float(eps), int(relu),
                                            _p(y), _lib.stream_ptr()), "pp_groupnorm_nhwc")
    return y


def batchnorm_train(x, bn, relu=False, residual=None, residual2=None, eps=1e-5, momentum=0.1):
    """nn.BatchNorm2d in TRAINING mode on an NHWC map (training forward only): normalises with the statistics of this batch
    and updates bn.running_mean / running_var / num_batches_tracked in place, as the module does.  `bn`: a parameter holder
    with weight, bias and the three buffers (model/common.bn_p).  y = relu?(bn(x)) + residual + residual2."""
    assert x.is_contiguous() and x.dtype == torch.float32
    C = x.shape[-1]
    rows = x.numel() // C
    L = _lib.lib()
    nbytes = L.pp_batchnorm_train_workspace_bytes(rows, C)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    y = torch.empty_like(x)
    for r in (residual, residual2):
        assert r is None or (r.is_contiguous() and r.shape == x.shape)
    _lib.check(L.pp_batchnorm_train(_p(x), _p(bn.weight), _p(bn.bias), rows, C, float(eps), float(momentum), _p(bn.running_mean),
                                    _p(bn.running_var), int(relu), _p(residual) if residual is not None else None,
                                    _p(residual2) if residual2 is not None else None, _p(y), _p(ws), nbytes, _lib.stream_ptr()),
               "pp_batchnorm_train")
    bn.num_batches_tracked += 1
    return y


def to_nhwc(x, c_pad=None):
    """(B,C,H,W) -> (B,H,W,C); with c_pad > C the result has c_pad channels, the extra ones zero."""
    B, C, H, W = x.shape
    x = x.contiguous().float()
    Cp = c_pad if c_pad and c_pad > C else C
    out = (torch.zeros if Cp > C else torch.empty)(B, H, W, Cp, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().pp_transpose_batched(_p(x), 0, B, C, H * W, _p(out), 0, Cp, 0, _lib.stream_ptr()),
               "pp_transpose_batched")
    return out


def to_nchw(x):
    """(B,H,W,C) -> (B,C,H,W)."""
    B, H, W, C = x.shape
    assert x.is_contiguous()
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().pp_transpose_batched(_p(x), 0, B, H * W, C, _p(out), 0, H * W, 0, _lib.stream_ptr()),
               "pp_transpose_batched")
    return out


def resize_bilinear(x, Ho, Wo, mul=1.0, out_split=False, also_split=False):
    """F.interpolate(bilinear, align_corners=True) on NHWC, result times `mul`.  out_split (f16x3 engine): return the
    result as a Split with .image for the convolution that follows, instead of an fp32 tensor; also_split: the fp32 tensor with
    its operand form attached as `._hl` (one pass writes both)."""
    B, H, W, C = x.shape
    assert x.is_contiguous()
    if also_split and _split_ok(C) and x.data_ptr() % 16 == 0:
        out = torch.empty(B, Ho, Wo, C, dtype=torch.float32, device=x.device)
        sp = Split.empty(B * Ho * Wo, C, x.device)
        sp.image = (B, Ho, Wo)
        _lib.check(_lib.lib().pp_resize_bilinear_nhwc_dual(_p(x), B, H, W, C, Ho, Wo, float(mul), _p(out), _p(sp.hl), sp.terms,
                                                            _lib.stream_ptr()), "pp_resize_bilinear_nhwc_dual")
        _chk(sp.hl, "pp_resize_bilinear_nhwc_dual")
        out._hl = sp
        return out
    if out_split and _split_ok(C) and x.data_ptr() % 16 == 0:
        sp = Split.empty(B * Ho * Wo, C, x.device)
        sp.image = (B, Ho, Wo)
        _lib.check(_lib.lib().pp_resize_bilinear_nhwc_t(_p(x), B, H, W, C, Ho, Wo, float(mul), _p(sp.hl), sp.terms, _lib.stream_ptr()),
                   "pp_resize_bilinear_nhwc_t")
        _chk(sp.hl, "pp_resize_bilinear_nhwc_t")
        return sp
    out = torch.empty(B, Ho, Wo, C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().pp_resize_bilinear_nhwc(_p(x), B, H, W, C, Ho, Wo, float(mul), _p(out), _lib.stream_ptr()),
               "pp_resize_bilinear_nhwc")
    return out


def warp(feat, flow, out=None, hl_into=None):
    """grid_sample warp of NHWC `feat` by NHWC `flow` (B,H,W,>=2 with x,y first); out may be a channel slice.
    feat may hold B / k images: image b of the (hypothesis-major) batch then samples feat[b % (B / k)].
    hl_into = (Split over (B*H*W, Ctot), col0): write the result only as operand columns of that Split (f16x3 engine)."""
    B, H, W = flow.shape[:3]
    Bf, Hf, Wf, C = feat.shape
    assert (Hf, Wf) == (H, W) and B % Bf == 0
    assert feat.is_contiguous() and flow.stride(3) == 1 and flow.stride(2) * W == flow.stride(1)
    if hl_into is not None:
        tgt, col0 = hl_into
        ctot = tgt.shape[1]
        assert tgt.shape[0] == B * H * W and col0 % 8 == 0 and col0 + C <= ctot and C % 8 == 0
        _lib.check(_lib.lib().pp_warp_nhwc_t(_p(feat), Bf, _p(flow), B, H, W, C, flow.stride(2), tgt.col_ptr(col0),
                                             ctot, tgt.terms, _lib.stream_ptr()), "pp_warp_nhwc_t")
        if CHECK_SATURATION:
            _chk(tgt.cols(col0, C), "pp_warp_nhwc_t")
        return None
    if out is None:
        out = torch.empty(B, H, W, C, dtype=torch.float32, device=feat.device)
    _lib.check(_lib.lib().pp_warp_nhwc(_p(feat), Bf, _p(flow), B, H, W, C, flow.stride(2), _p(out), out.stride(2),
                                       _lib.stream_ptr()), "pp_warp_nhwc")
    return out


def avgpool2(x):
    B, H, W, C = x.shape
    assert x.is_contiguous()
    out = torch.empty(B, H // 2, W // 2, C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().pp_avgpool2_nhwc(_p(x), B, H, W, C, _p(out), _lib.stream_ptr()), "pp_avgpool2_nhwc")
    return out


def corr_lookup(f1, f2, flow, levels, radius, c_pad=None, f1_hl=None, f2_hl=None):
    """Correlation pyramid + lookup, NHWC: (B,H,W,C) x2, flow (B,H,W,>=2) -> (B,H,W,levels*(2r+1)^2)
    (c_pad: channel count of the result, zero-filled beyond the levels*(2r+1)^2 real channels).
    f1 may be a channel slice of a wider buffer; f2 may hold B / k images (image b reads f2[b % (B / k)]).
    f1_hl = (Split over (B*H*W, Ctot), col0), f2_hl = Split over (Bf*H*W, C): the same maps as operands the producers have
    already written (f16x3 engine, tileable shapes): the kernel then stages 16-byte copies instead of splitting every chunk."""
    B, H, W, C = f1.shape
    assert f2.is_contiguous() and flow.stride(3) == 1 and B % f2.shape[0] == 0 and tuple(f2.shape[1:]) == (H, W, C)
    assert f1.stride(3) == 1 and f1.stride(1) == W * f1.stride(2) and f1.stride(0) == H * f1.stride(1)
    pyr = [f2]
    for _ in range(levels - 1):
        pyr.append(avgpool2(pyr[-1]))
    n = levels * (2 * radius + 1) ** 2
    np_ = c_pad if c_pad and c_pad > n else n
    out = (torch.zeros if np_ > n else torch.empty)(B, H, W, np_, dtype=torch.float32, device=f1.device)
    if (f1_hl is not None and f2_hl is not None and PRECISION == "f16x3" and f2_hl.terms == 2 and H % 8 == 0 and W % 8 == 0 and C % 32 == 0
            and os.environ.get("PP_CORR_TILED", "1") != "0" and os.environ.get("PP_CORR_HL", "1") != "0"):
        tgt, col0 = f1_hl
        assert tgt.shape[0] == B * H * W and col0 % 8 == 0 and col0 + C <= tgt.shape[1] and f2_hl.shape == (f2.shape[0] * H * W, C)
        hls = [f2_hl.hl] + [split_activation(p, p.shape[0], p.shape[1] * p.shape[2], C, p.shape[1] * p.shape[2] * C, C) for p in pyr[1:]]
        _lib.check(_lib.lib().pp_corr_lookup_nhwc_hl(tgt.hl.data_ptr() + 4 * col0, tgt.shape[1], _p(hls[0]), _p(hls[1]) if levels > 1 else None,
                                                     _p(hls[2]) if levels > 2 else None, f2.shape[0], _p(flow), B, H, W, C, levels, radius,
                                                     flow.stride(2), _p(out), np_, _lib.stream_ptr()), "pp_corr_lookup_nhwc_hl")
        return out
    _lib.check(_lib.lib().pp_corr_lookup_nhwc_ex(_p(f1), f1.stride(2), _p(pyr[0]), _p(pyr[1]) if levels > 1 else None,
                                                 _p(pyr[2]) if levels > 2 else None, f2.shape[0], _p(flow), B, H, W, C,
                                                 levels, radius, flow.stride(2), 2 if PRECISION == "f16" else _fly_prec(), _p(out), np_,
                                                 _lib.stream_ptr()), "pp_corr_lookup_nhwc_ex")
    return out


def tokens_to_nchw(tokens, skip, H, W):
    """(B,T,C) token rows [skip: skip+H*W] -> (B,C,H,W)  (FeatureExtractor output, feature_extractor.py:105-107)."""
    B, T, C = tokens.shape
    assert tokens.is_contiguous()
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=tokens.device)
    src = tokens[:, skip:]
    _lib.check(_lib.lib().pp_transpose_batched(_p(src), T * C, B, H * W, C, _p(out), 0, H * W, 0, _lib.stream_ptr()),
               "pp_transpose_batched")
    return out


def assemble_tokens(patches, cls_token, pos):
    """patches (B,T,C), cls (C), pos (T+1,C) -> tokens (B,T+1,C)."""
    B, T, C = patches.shape
    assert patches.is_contiguous() and pos.is_contiguous() and cls_token.is_contiguous()
    out = torch.empty(B, T + 1, C, dtype=torch.float32, device=patches.device)
    _lib.check(_lib.lib().pp_assemble_tokens(_p(patches), _p(cls_token), _p(pos), B, T, C, _p(out), _lib.stream_ptr()),
               "pp_assemble_tokens")
    return out


def normalize_rows(x, eps=1e-12):
    rows, n = x.shape
    assert x.is_contiguous()
    y = torch.empty_like(x)
    _lib.check(_lib.lib().pp_normalize_rows(_p(x), rows, n, float(eps), _p(y), _lib.stream_ptr()), "pp_normalize_rows")
    return y


def gather_rows(src, index):
    """dst[i] = src.flatten(0, k)[index[i]]: src (R, ...) contiguous fp32 with rows of a multiple of 4 floats, index (n,)
    int64 -> (n, ...).  Small rows (poses, intrinsics) that are not a multiple of 4 floats fall to torch indexing."""
    R = src.shape[0]
    row = src[0].numel()
    if row % 4 != 0 or not src.is_contiguous() or src.dtype != torch.float32 or src.data_ptr() % 16 != 0:
        return src[index]
    index = index.contiguous()
    dst = torch.empty((index.numel(),) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    _lib.check(_lib.lib().pp_gather_rows(_p(src), _p(index), R, row, index.numel(), _p(dst), _lib.stream_ptr()),
               "pp_gather_rows")
    return dst
