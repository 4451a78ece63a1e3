"""The backward of the training path: the reference's whole training step under autograd.

`torch.autograd.Function` wrappers whose forward AND backward run on libpicopose_hip.so: the matrix products are pp_gemm
launches on the pre-split engine (dgrad = dz @ W; wgrad = dz^T @ x with both operands produced K-major in one pass and, when the
output has few tiles, the K slices of PpGemmDesc.ksplit), attention is fused in both directions (csrc/pp_attn_bwd.hip), the
row-wise and sampling adjoints are the kernels of csrc/pp_backward.hip / pp_backward3.hip.  torch keeps the graph, allocates
tensors and sums the gradients of a tensor that is used twice (the residual stream, the taken ViT levels: autograd's own
accumulation) — nothing else on the gradient path is torch arithmetic.

Scope (exactly the parameters that receive a gradient from `Net.forward_train`), by `Net.train_backward`:
  * "full" (= True, the default): the reference's training step — all ten losses under autograd, every parameter the reference
      trains receives its gradient: the ViT (blocks, patch / cls / position embeddings), `affine_regressor`, the DPT head and the
      flow decoder (stage 3 in training mode, layer by layer: implicit-GEMM convolutions in all three directions, training-mode BatchNorm, align_corners
      resize, ConvTranspose, feature warp, the fused correlation pyramid + lookup, flow / certainty losses — adjoints in
      csrc/pp_backward3.hip);
  * "vit+stage2": only the stage-1 and stage-2 losses (InfoNCE, utils/loss_utils.py:144-175; the three stage-2 losses, :177-186,
      through the similarity volume of utils/matching.py:6-26) -> every ViT block, the embeddings, `affine_regressor`; stage 3
      forward-only;
  * "slice1": the first slice — stage-2 losses -> `affine_regressor` (the gradient stops at the similarity volume), InfoNCE ->
      the LAST ViT block (stops at its input).
Backward products run on range-normalised operands (`_ranged`: gradients span 1e-9 .. 1, the f16x3 operand format is full precision
only above ~3e-5): each operand is scaled by a power of two chosen on the device from its max, the product is scaled back — exact.
Parity: tests/test_train_gpu.py compares these gradients with the reference's own autograd on CPU (tests/golden/train_grads.npz).
"""
import os

import torch

from . import _lib, ops

ACT = ops.ACT
# True: the two scatter adjoints (feature warp, correlation lookup) accumulate in 64-bit fixed point with integer atomics — gradients
# repeat bit for bit from run to run (the reference offers deterministic training through seeding); False: fp32 atomics, faster,
# last-bit differences between runs (tests/test_train_gpu.py::test_deterministic_option_repeats_bit_for_bit).
DETERMINISTIC = False


def _p(t):
    return t.data_ptr() if t is not None else None


def _f32c(t):
    return t.contiguous().float()


def colsum(x2d):
    """Column sums of a contiguous (rows, cols) matrix (fixed summation order)."""
    rows, cols = x2d.shape
    L = _lib.lib()
    nbytes = L.pp_colsum_workspace_bytes(rows, cols)
    ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=x2d.device)
    out = torch.empty(cols, dtype=torch.float32, device=x2d.device)
    _lib.check(L.pp_colsum(_p(x2d), rows, cols, x2d.stride(0), _p(out), _p(ws), nbytes, _lib.stream_ptr()), "pp_colsum")
    return out


def _ew(op, a, b, cols=0):
    out = torch.empty_like(a)
    _lib.check(_lib.lib().pp_elementwise(op, _p(a), _p(b), a.numel(), cols, _p(out), _lib.stream_ptr()), "pp_elementwise")
    return out


def _pow2_scale(t):
    """Device scalars (2^k, 2^-k) with max|t| 2^k in [512, 1024): gradient tensors span 1e-9 .. 1, the engine's operand format holds
    4 x as two fp16 terms (full 22 bits only above ~3e-5) — backward products are taken on range-normalised copies and scaled
    back, exactly (powers of two).  One reduction on the device (pp_pow2_scale), no host sync; the scale never enters a value
    except through rounding."""
    buf = torch.empty(2 + 1024, dtype=torch.float32, device=t.device)      # (scale, inverse | per-workgroup maxima)
    _lib.check(_lib.lib().pp_pow2_scale_ws(_p(t), t.numel(), _p(buf), _p(buf[2:]), _lib.stream_ptr()), "pp_pow2_scale_ws")
    return buf[:2]


def _ranged(t, on=True):
    """(range-normalised contiguous copy, its (scale, 1 / scale) pair) — identity in fp32 engine mode and for operands that are
    forward quantities (activations, weights, probabilities: in the operand format's range as they are; on=False)."""
    if ops.PRECISION == "f32" or not on:
        return t, None
    t = t.contiguous()
    s2 = _pow2_scale(t)
    return _ew(1, t, s2[0:1], 1), s2


def _unscale(out, sa, sb, target=None):
    """out / (sa sb) (exact: powers of two), into `target` (a possibly strided view) when given."""
    for s2 in (sa, sb):
        if s2 is not None:
            out = _ew(1, out, s2[1:2], 1)
    if target is not None:
        target.copy_(out)
        return target
    return out


def _scale_of(t, on=True):
    """Device scale pair of a backward operand WITHOUT a scaled copy (None: fp32 engine, or a forward quantity that is in range as
    it is) — for the products whose operands are split with the scale applied (ops.split_scaled / split_transposed)."""
    return None if (ops.PRECISION == "f32" or not on) else _pow2_scale(t.contiguous() if not t.is_contiguous() else t)


def _inv(*scales):
    return [s[1:2] for s in scales if s is not None]


def _mm_tn(a, b, ra=True, rb=True, sa=None):
    """a^T @ b for a (R, M), b (R, N) — the weight-gradient form (R = the rows of the batch is the K axis).  On the pre-split engine both
    operands are produced K-major in ONE pass each (scale applied, transposed, split: ops.split_transposed) and the inverse scales ride
    in the launch (alpha_dev): no scaled copy, no transposed copy, no unscale pass.  sa: the scale pair of `a` if the caller has it."""
    R, M = a.shape
    N = b.shape[1]
    tiles = -(-M // 128) * -(-N // 128)
    if ops.operands_ok(M, N, R) and a.stride(1) == 1 and b.stride(1) == 1:
        # few output tiles over the long K: K slices inside ONE engine launch (ops.ksplit_choice: a function of the shape; the slice's
        # rows must be whole tiles — M % 256 — so a product with only N % 256 == 0 is taken transposed)
        swap = M % 256 != 0 and N % 256 == 0
        S, kp = ops.ksplit_choice(N, M, R) if swap else ops.ksplit_choice(M, N, R)
        if (tiles > 24 or S > 1) and max(M, N) * kp < 2 ** 30:
            sa = sa if sa is not None else _scale_of(a, ra)
            sb = _scale_of(b, rb)
            A, Bt = ops.split_transposed(a, sa, k_pad=kp), ops.split_transposed(b, sb, k_pad=kp)
            if swap:
                return ops.matmul_operands(Bt, A, alpha_dev=_inv(sa, sb), ksplit=S).t().contiguous()
            return ops.matmul_operands(A, Bt, alpha_dev=_inv(sa, sb), ksplit=S)
    return _mm(a.t().contiguous(), b, ra=ra, rb=rb)


def bmm_nn_b(a, b, out=None, alpha=1.0, ra=True, rb=True):
    """Backward product out = alpha a @ b for 4-D views (Z0,Z1,M,K) x (Z0,Z1,K,N); ra / rb: range-normalise that operand (a
    gradient) or take it as it is (a forward quantity)."""
    a2, sa = _ranged(a, ra)
    b2, sb = _ranged(b, rb)
    tmp = torch.empty(a.shape[0], a.shape[1], a.shape[2], b.shape[3], dtype=torch.float32, device=a.device)
    ops.bmm_nn(a2, b2, tmp, alpha=alpha)
    return _unscale(tmp, sa, sb, out)


def bmm_nt_b(a, b, alpha=1.0, ra=True, rb=True):
    a2, sa = _ranged(a, ra)
    b2, sb = _ranged(b, rb)
    return _unscale(ops.bmm_nt(a2, b2, alpha=alpha), sa, sb)


def _mm(a, b, alpha=1.0, out=None, ra=True, rb=True):
    """a (M,K) @ b (K,N) on the GEMM engine (b rows contiguous), operands range-normalised (backward products).  Large products
    take the pre-split engine (both operands split as activations, b through a transposed copy); small ones the batched kernel."""
    M, K = a.shape
    N = b.shape[1]
    tiles = -(-M // 128) * -(-N // 128)
    if tiles <= 24 and K >= 8192 and K % 64 == 0 and alpha == 1.0:     # (above ~24 tiles the pre-split engine at tiles / 256 of its rate beats the batched kernel)
        # few output tiles over a long K (weight gradients: the rows of the batch are the K axis, e.g. 256 x 256 x 131072 or the
        # 2 x 131072 x 2304 of a predict layer): a handful of tiles walking all of K leave the chip idle — cut K into S slices
        # that run as one batched launch, then add them in index order
        S = 64 if tiles <= 8 else 16
        while K % S or (K // S) % 8:
            S //= 2
        kc = K // S
        a2, sa = _ranged(a, ra)
        b2, sb = _ranged(b, rb)
        part = torch.empty(1, S, M, N, dtype=torch.float32, device=a.device)
        ops.bmm_nn(a2.contiguous().view(M, S, kc).permute(1, 0, 2)[None], b2.contiguous().view(S, kc, N)[None], part)
        r = torch.empty(M, N, dtype=torch.float32, device=a.device)
        _lib.check(_lib.lib().pp_sum_slices(_p(part), S, M, N, None, 0, _p(r), _lib.stream_ptr()), "pp_sum_slices")
        return _unscale(r, sa, sb, out)
    if alpha == 1.0 and ops.operands_ok(M, N, K) and M * N * K >= 1 << 24 and a.stride(1) == 1 and b.stride(1) == 1:
        # both operands split with their range scale applied (b: transposed in the same pass), the inverse scales in the launch
        sa, sb = _scale_of(a, ra), _scale_of(b, rb)
        A = ops.split_scaled(a, sa) if sa is not None else ops.Split(ops.split_activation(a, 1, M, K, 0, a.stride(0)))
        return ops.matmul_operands(A, ops.split_transposed(b, sb), alpha_dev=_inv(sa, sb), out=out)
    r = bmm_nn_b(a[None, None], b[None, None], None if out is None else out[None, None], alpha=alpha, ra=ra, rb=rb)
    return r[0, 0]


class _Linear(torch.autograd.Function):
    """y = act(x @ w.T + b) with x (M,K), w (N,K): forward on the engine (activation as its own pass, so that the
    pre-activation z is kept), backward dz = dy act'(z), dx = dz @ w, dw = dz^T @ x, db = column sums of dz."""

    @staticmethod
    def forward(ctx, x, w, b, act, owner=None):
        # a parameter itself: its operand form is cached per (address, version); a matrix derived from the parameter `owner`
        # (reshaped, re-packed): split every step with the scale read back one step earlier; otherwise transient (one host wait).
        # `is_leaf` is not the test: a matrix derived from a FROZEN parameter is a leaf too, and caching a fresh tensor every step
        # would pin it and its operand copy in the cache for good
        cache = ("dev" if (w.requires_grad and DEVICE_WEIGHT_SCALE) else True) if isinstance(w, torch.nn.Parameter) else ((owner, "lin") if owner is not None else False)
        x, w = _f32c(x), _f32c(w)
        z = ops.linear(x, w, b, cache_weight=cache)
        ctx.act = ACT[act]
        if ctx.act:
            y = torch.empty_like(z)
            _lib.check(_lib.lib().pp_act_forward(_p(z), z.numel(), ctx.act, _p(y), _lib.stream_ptr()), "pp_act_forward")
        else:
            y = z
        ctx.save_for_backward(x, w, z if ctx.act else None)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, z = ctx.saved_tensors
        dy = _f32c(dy)
        if ctx.act:
            dz = torch.empty_like(dy)
            _lib.check(_lib.lib().pp_act_backward(_p(z), _p(dy), dy.numel(), ctx.act, _p(dz), _lib.stream_ptr()), "pp_act_backward")
        else:
            dz = dy
        dx = _mm(dz, w, rb=False) if ctx.needs_input_grad[0] else None                       # (weights / activations: forward quantities)
        dw = _mm_tn(dz, x, rb=False) if ctx.needs_input_grad[1] else None
        db = colsum(dz) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None, None


def linear(x, w, b=None, act=None, owner=None):
    return _Linear.apply(x, w, b, act, owner)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = _f32c(x)
        ctx.save_for_backward(x, w)
        ctx.eps = eps
        return ops.layernorm(x, w, b, eps)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _f32c(dy)
        dx, gx = torch.empty_like(x), torch.empty_like(x)
        _lib.check(_lib.lib().pp_layernorm_backward(_p(x), _p(w), _p(dy), x.shape[0], x.shape[1], float(ctx.eps), _p(dx), _p(gx), _lib.stream_ptr()),
                   "pp_layernorm_backward")
        return dx, colsum(gx), colsum(dy), None


def layernorm(x, w, b, eps):
    return _LayerNorm.apply(x, w, b, eps)


class _ScaleResidual(torch.autograd.Function):
    """y = res + gamma * t (LayerScale + residual, layers/block.py:92-106)."""

    @staticmethod
    def forward(ctx, t, gamma, res):
        t, res = _f32c(t), _f32c(res)
        ctx.save_for_backward(t, gamma)
        return _ew(2, res, _ew(1, t, gamma, t.shape[1]))

    @staticmethod
    def backward(ctx, dy):
        t, gamma = ctx.saved_tensors
        dy = _f32c(dy)
        return _ew(1, dy, gamma, t.shape[1]), colsum(_ew(0, dy, t)), dy


FUSED_ATTENTION = os.environ.get("PP_FUSED_ATTENTION", "1") != "0"
# trained parameters are re-split after every optimizer step: their power-of-two scale stays on the device (ops.split_weight_dev) instead of
# being read back by the host (one wait per parameter and step)
DEVICE_WEIGHT_SCALE = os.environ.get("PP_DEVICE_WEIGHT_SCALE", "1") != "0"


class _Attention(torch.autograd.Function):
    """softmax((q hd^-1/2) k^T) v per head on the qkv rows (B*T, 3*heads*hd) (layers/attention.py:49-62).  f16x3 engine: the fused
    kernel forward (pp_attention_train keeps the log-sum-exp per query) and the recomputing adjoint (pp_attention_backward): the T x T
    probabilities are never stored.  fp32 engine (or FUSED_ATTENTION off, head_dim != 64): unfused — S and the three products are batched
    pp_gemm launches, soft-max rows / their adjoint row kernels, the probabilities kept."""

    @staticmethod
    def forward(ctx, qkv, B, T, heads, hd):
        qkv = _f32c(qkv)
        ctx.dims = (B, T, heads, hd)
        ctx.fused = FUSED_ATTENTION and ops.PRECISION == "f16x3" and hd == 64
        if ctx.fused:
            out = torch.empty(B * T, heads * hd, dtype=torch.float32, device=qkv.device)
            lse = torch.empty(B * heads, T, dtype=torch.float32, device=qkv.device)
            _lib.check(_lib.lib().pp_attention_train(_p(qkv), B, T, heads, hd, float(hd) ** -0.5, _p(out), _p(lse), _lib.stream_ptr()),
                       "pp_attention_train")
            ctx.save_for_backward(qkv, out, lse)
            return out
        v5 = qkv.view(B, T, 3, heads, hd)
        q, k, v = (v5[:, :, i].permute(0, 2, 1, 3) for i in range(3))                # (B, heads, T, hd) strided views
        P = ops.softmax_rows_(ops.bmm_nt(q, k, alpha=float(hd) ** -0.5))             # (B, heads, T, T)
        out = torch.empty(B, T, heads, hd, dtype=torch.float32, device=qkv.device)
        ops.bmm_nn(P, v, out.permute(0, 2, 1, 3))
        ctx.save_for_backward(qkv, P)
        return out.view(B * T, heads * hd)

    @staticmethod
    def backward(ctx, dout):
        B, T, heads, hd = ctx.dims
        if ctx.fused:
            qkv, out, lse = ctx.saved_tensors
            dout = _f32c(dout)
            g = _pow2_scale(dout)
            dqkv = torch.empty_like(qkv)
            ws = torch.empty(B * heads * T, dtype=torch.float32, device=qkv.device)
            _lib.check(_lib.lib().pp_attention_backward(_p(qkv), _p(out), _p(dout), _p(lse), _p(g), B, T, heads, hd, float(hd) ** -0.5,
                                                        _p(ws), _p(dqkv), _lib.stream_ptr()), "pp_attention_backward")
            return dqkv, None, None, None, None
        qkv, P = ctx.saved_tensors
        v5 = qkv.view(B, T, 3, heads, hd)
        q, k, v = (v5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        dO = _f32c(dout).view(B, T, heads, hd).permute(0, 2, 1, 3)
        dP = bmm_nt_b(dO, v, rb=False)                                                # dO v^T
        dS = torch.empty_like(P)
        _lib.check(_lib.lib().pp_softmax_backward_rows(_p(P), _p(dP), B * heads * T, T, _p(dS), _lib.stream_ptr()), "pp_softmax_backward_rows")
        dqkv = torch.empty_like(qkv)
        d5 = dqkv.view(B, T, 3, heads, hd)
        s = float(hd) ** -0.5
        bmm_nn_b(dS, k.contiguous(), d5[:, :, 0].permute(0, 2, 1, 3), alpha=s, rb=False)                # dq = dS k / sqrt(hd)
        bmm_nn_b(dS.transpose(2, 3).contiguous(), q.contiguous(), d5[:, :, 1].permute(0, 2, 1, 3), alpha=s, rb=False)   # dk = dS^T q / sqrt(hd)
        bmm_nn_b(P.transpose(2, 3).contiguous(), dO.contiguous(), d5[:, :, 2].permute(0, 2, 1, 3), ra=False)      # dv = P^T dO
        return dqkv, None, None, None, None


class _GroupNormRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, groups, relu):
        x = _f32c(x)
        ctx.save_for_backward(x, w, b)
        ctx.groups, ctx.relu = groups, relu
        return ops.groupnorm(x, w, b, groups, relu=relu)

    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        B, H, W, C = x.shape
        dy = _f32c(dy)
        dx, gx, gy = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        _lib.check(_lib.lib().pp_groupnorm_backward_nhwc(_p(x), _p(w), _p(b), _p(dy), B, H * W, C, ctx.groups, 1e-5, int(ctx.relu), _p(dx), _p(gx),
                                                         _p(gy), _lib.stream_ptr()), "pp_groupnorm_backward_nhwc")
        return dx, colsum(gx.view(-1, C)), colsum(gy.view(-1, C)), None, None


class _Im2col(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        x = _f32c(x)
        B, H, W, C = x.shape
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        col = torch.empty(B * Ho * Wo, k * k * C, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().pp_im2col_nhwc(_p(x), B, H, W, C, k, s, p, _p(col), _lib.stream_ptr()), "pp_im2col_nhwc")
        ctx.geom = (B, H, W, C, k, s, p)
        return col

    @staticmethod
    def backward(ctx, dcol):
        B, H, W, C, k, s, p = ctx.geom
        dx = torch.empty(B, H, W, C, dtype=torch.float32, device=dcol.device)
        _lib.check(_lib.lib().pp_col2im_nhwc(_p(_f32c(dcol)), B, H, W, C, k, s, p, _p(dx), _lib.stream_ptr()), "pp_col2im_nhwc")
        return dx, None, None, None


class _NormalizeRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        x = _f32c(x)
        ctx.save_for_backward(x)
        ctx.eps = eps
        return ops.normalize_rows(x, eps)

    @staticmethod
    def backward(ctx, dq):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().pp_normalize_rows_backward(_p(x), x.shape[1], None, _p(_f32c(dq)), x.shape[0], x.shape[1], float(ctx.eps), _p(dx),
                                                         _lib.stream_ptr()), "pp_normalize_rows_backward")
        return dx, None


class _InfoNCE(torch.autograd.Function):
    """utils/loss_utils.py:163-175 on token-major features: gather + normalise the key-point rows, logits = q r^T on the engine,
    mean of the diagonal cross-entropy rows; backward down to the two token tensors."""

    @staticmethod
    def forward(ctx, tok_s, tok_t, s_rows, t_rows, tau):
        ts, tt = _f32c(tok_s), _f32c(tok_t)
        L = _lib.lib()
        n, C = s_rows.numel(), ts.shape[-1]
        q = torch.empty(n, C, dtype=torch.float32, device=ts.device)
        r = torch.empty_like(q)
        _lib.check(L.pp_gather_normalize_rows(_p(ts), C, _p(s_rows), n, C, 1e-12, _p(q), _lib.stream_ptr()), "pp_gather_normalize_rows")
        _lib.check(L.pp_gather_normalize_rows(_p(tt), C, _p(t_rows), n, C, 1e-12, _p(r), _lib.stream_ptr()), "pp_gather_normalize_rows")
        logits = ops.bmm_nt(q[None, None], r[None, None])[0, 0]
        rows = torch.empty(n, dtype=torch.float32, device=ts.device)
        _lib.check(L.pp_xent_diag_rows(_p(logits), n, logits.stride(0), 1.0 / tau, _p(rows), _lib.stream_ptr()), "pp_xent_diag_rows")
        ctx.save_for_backward(ts, tt, s_rows, t_rows, q, r, logits)
        ctx.tau = tau
        return rows.mean()

    @staticmethod
    def backward(ctx, up):
        ts, tt, s_rows, t_rows, q, r, logits = ctx.saved_tensors
        L = _lib.lib()
        n, C = q.shape
        dl = torch.empty(n, n, dtype=torch.float32, device=q.device)
        _lib.check(L.pp_xent_diag_backward(_p(logits), n, logits.stride(0), 1.0 / ctx.tau, _p(_f32c(up).reshape(1)), _p(dl), _lib.stream_ptr()),
                   "pp_xent_diag_backward")
        dq, dr = _mm(dl, r, rb=False), _mm(dl.t().contiguous(), q, rb=False)
        out = []
        for tok, rows_, dn in ((ts, s_rows, dq), (tt, t_rows, dr)):
            dx = torch.empty(n, C, dtype=torch.float32, device=q.device)
            _lib.check(L.pp_normalize_rows_backward(_p(tok), C, _p(rows_), _p(dn), n, C, 1e-12, _p(dx), _lib.stream_ptr()),
                       "pp_normalize_rows_backward")
            # the backward of the row gather is a scatter-ADD: the template-side rows are distinct grid cells, but the real-image
            # rows are re-projected points quantised to the 16x16 feature grid (utils/loss_utils.py:150-160) and several key-points
            # can share a cell (more than half of them when the real crop is the smaller view).  Fixed summation order, no atomics.
            g = torch.zeros(tok.numel() // C, C, dtype=torch.float32, device=q.device)
            _lib.check(L.pp_scatter_add_rows(_p(dx), _p(rows_), n, C, _p(g), _lib.stream_ptr()), "pp_scatter_add_rows")
            out.append(g.view_as(tok))
        return out[0], out[1], None, None, None


class _AssembleTokens(torch.autograd.Function):
    """tokens = [cls ; patches] + pos (vision_transformer.py:209-228) on rows: patches (B*P, C), cls (1,1,C), pos (P+1, C)."""

    @staticmethod
    def forward(ctx, patches, cls, pos, B):
        patches = _f32c(patches)
        P, C = patches.shape[0] // B, patches.shape[1]
        ctx.dims = (B, P, C)
        out = ops.assemble_tokens(patches.view(B, P, C), _f32c(cls).reshape(C), _f32c(pos))
        return out.view(B * (P + 1), C)

    @staticmethod
    def backward(ctx, dtok):
        B, P, C = ctx.dims
        d3 = _f32c(dtok).view(B, P + 1, C)
        dpatches = d3[:, 1:].reshape(B * P, C)
        dcls = colsum(d3[:, 0]).view(1, 1, C)                         # rows b, stride (P + 1) C
        dpos = colsum(d3.view(B, (P + 1) * C)).view(P + 1, C)
        return dpatches, dcls, dpos, None


class _InterpPos(torch.autograd.Function):
    """interpolate_pos_encoding (vision_transformer.py:179-207) as a function of pos_embed: the value is the table the module
    resamples at pack time (`pos`), the gradient goes through the resampling MATRIX `wt` (N, P) = W^T, W[p][n] = the bicubic
    weight of source cell n in output cell p (FeatureExtractor._pos_wt): dpos_embed[1:] = W^T dpos[1:] on the engine."""

    @staticmethod
    def forward(ctx, pos_embed, pos, wt):
        ctx.save_for_backward(wt)
        ctx.shape = pos_embed.shape
        return pos.clone()

    @staticmethod
    def backward(ctx, dpos):
        (wt,) = ctx.saved_tensors
        dpos = _f32c(dpos)
        out = torch.empty(ctx.shape, dtype=torch.float32, device=dpos.device)
        out[0, 0].copy_(dpos[0])
        if wt is None:                                                  # the grid is the stored one: no resampling
            out[0, 1:].copy_(dpos[1:])
        else:
            _mm(wt, dpos[1:], out=out[0, 1:], ra=False)
        return out, None, None


class _SimilarityVolume(torch.autograd.Function):
    """matching_features_similarity (utils/matching.py:6-26) of token tensors (B, 1 + 256, C) [cls row first]: the value is
    pp_similarity_volume's (the forward-only path's bits); backward through mask / clamp / layout (pp_simvol_backward), the two
    products dS src_hat and dS^T tar_hat on the engine, and F.normalize's adjoint per patch row."""

    @staticmethod
    def forward(ctx, tok_src, tok_tar, src_mask):
        from .utils.matching import matching_features_similarity

        ts, tt = _f32c(tok_src), _f32c(tok_tar)
        out = matching_features_similarity(ops.tokens_to_nchw(ts, 1, 16, 16), ops.tokens_to_nchw(tt, 1, 16, 16), src_mask, None)
        ctx.save_for_backward(ts, tt, src_mask, out)
        return out

    @staticmethod
    def backward(ctx, dout):
        ts, tt, mask, out = ctx.saved_tensors
        L = _lib.lib()
        B, T, C = ts.shape
        P = T - 1
        rows = (torch.arange(B, device=ts.device)[:, None] * T + 1 + torch.arange(P, device=ts.device)[None]).reshape(-1)
        q = torch.empty(B * P, C, dtype=torch.float32, device=ts.device)
        r = torch.empty_like(q)
        _lib.check(L.pp_gather_normalize_rows(_p(ts), C, _p(rows), B * P, C, 1e-12, _p(q), _lib.stream_ptr()), "pp_gather_normalize_rows")
        _lib.check(L.pp_gather_normalize_rows(_p(tt), C, _p(rows), B * P, C, 1e-12, _p(r), _lib.stream_ptr()), "pp_gather_normalize_rows")
        mask = _f32c(mask)
        dS = torch.empty(B, P, P, dtype=torch.float32, device=ts.device)           # [b][t][s]
        _lib.check(L.pp_simvol_backward(_p(out), _p(_f32c(dout)), _p(mask), mask.shape[1], mask.shape[2], B, _p(dS), _lib.stream_ptr()),
                   "pp_simvol_backward")
        dr = torch.empty(1, B, P, C, dtype=torch.float32, device=ts.device)
        dq = torch.empty_like(dr)
        bmm_nn_b(dS[None], q.view(1, B, P, C), dr, rb=False)                        # d tar_hat[t] = sum_s dS[t][s] src_hat[s]
        bmm_nn_b(dS.transpose(1, 2).contiguous()[None], r.view(1, B, P, C), dq, rb=False)   # d src_hat[s] = sum_t dS[t][s] tar_hat[t]
        grads = []
        for tok, dn in ((ts, dq), (tt, dr)):
            dx = torch.empty(B * P, C, dtype=torch.float32, device=ts.device)
            _lib.check(L.pp_normalize_rows_backward(_p(tok), C, _p(rows), _p(dn), B * P, C, 1e-12, _p(dx), _lib.stream_ptr()),
                       "pp_normalize_rows_backward")
            g = torch.zeros(B * T, C, dtype=torch.float32, device=ts.device)
            g.index_copy_(0, rows, dx)
            grads.append(g.view(B, T, C))
        return grads[0], grads[1], None


def similarity_volume(tok_src, tok_tar, src_mask):
    return _SimilarityVolume.apply(tok_src, tok_tar, src_mask)


def embed_tokens(fe, x):
    """prepare_tokens_with_masks (vision_transformer.py:209-228) under autograd: the 14x14 / stride 14 patch embedding as im2col +
    linear (the image needs no gradient), cls token, resampled position embedding.  Returns token rows (B*T, C)."""
    v = fe.dinov2
    B, _, H, W = x.shape
    p = v.patch_size
    h0, w0 = H // p, W // p
    img = ops.to_nhwc(x, c_pad=8)
    col = torch.empty(B * h0 * w0, p * p * 8, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().pp_im2col_nhwc(_p(img), B, H, W, 8, p, p, 0, _p(col), _lib.stream_ptr()), "pp_im2col_nhwc")
    wp = ops.pack_conv_weight(v.patch_embed.proj.weight.float(), cin_pad=8)            # views + a zero pad: torch differentiates them
    patches = linear(col, wp, v.patch_embed.proj.bias)
    pos = _InterpPos.apply(v.pos_embed, fe._pos(h0, w0), fe._pos_wt(h0, w0))
    return _AssembleTokens.apply(patches, v.cls_token, pos, B)


# ---- the two sub-graphs of Net.forward_train that carry gradients -------------------------------------------------------------
def last_block_forward(blk, xs, B, T, heads, hd):
    """One pre-norm ViT block (layers/block.py:82-107) on token rows xs (B*T, C) under autograd."""
    h = layernorm(xs, blk.norm1.weight, blk.norm1.bias, 1e-6)
    qkv = linear(h, blk.attn.qkv.weight, blk.attn.qkv.bias)
    o = _Attention.apply(qkv, B, T, heads, hd)
    x1 = _ScaleResidual.apply(linear(o, blk.attn.proj.weight, blk.attn.proj.bias), blk.ls1.gamma, xs)
    h = layernorm(x1, blk.norm2.weight, blk.norm2.bias, 1e-6)
    f = linear(h, blk.mlp.fc1.weight, blk.mlp.fc1.bias, act="gelu")
    return _ScaleResidual.apply(linear(f, blk.mlp.fc2.weight, blk.mlp.fc2.bias), blk.ls2.gamma, x1)


def affine_regressor_forward(reg, sim):
    """AffineRegressor.forward (model/stage2/affine_regressor.py:72-84) under autograd; sim (B,256,16,16) is a constant or the
    output of similarity_volume (then the gradient continues into the ViT)."""
    f = reg.features
    B = sim.shape[0]
    hd, fs = reg.hidden_dim, reg.feat_size
    x = sim.permute(0, 2, 3, 1).contiguous() if sim.requires_grad else ops.to_nhwc(sim)   # (B,16,16,256)
    c0, c3 = getattr(f, "0"), getattr(f, "3")
    w0 = c0.weight.reshape(c0.weight.shape[0], -1)                                   # 1x1 convolution = a linear layer on the pixels
    h = linear(x.view(-1, x.shape[-1]), w0, c0.bias).view(B, 16, 16, hd)
    h = _GroupNormRelu.apply(h, getattr(f, "1").weight, getattr(f, "1").bias, 32, True)
    w3 = c3.weight.permute(0, 2, 3, 1).reshape(c3.weight.shape[0], -1)               # (Cout, ky, kx, Cin): the engine's k order
    h = linear(_Im2col.apply(h, 3, 2, 1), w3, None).view(B, fs, fs, hd)
    h = _GroupNormRelu.apply(h, getattr(f, "4").weight, getattr(f, "4").bias, 32, True)
    # x.flatten(1) of the NCHW map indexes (c, h, w); the engine's map is (h, w, c)
    w1 = reg.fc1.weight.view(-1, hd, fs, fs).permute(0, 2, 3, 1).reshape(-1, fs * fs * hd)
    h = linear(h.reshape(B, -1), w1, reg.fc1.bias, act="leaky01")
    h = linear(h, reg.fc2.weight, reg.fc2.bias, act="leaky01")

    def mlp(head, last_act=None):
        l0, l2, l4 = getattr(head, "0"), getattr(head, "2"), getattr(head, "4")
        y = linear(linear(h, l0.weight, l0.bias, act="relu"), l2.weight, l2.bias, act="relu")
        return linear(y, l4.weight, l4.bias, act=last_act)

    translation = mlp(reg.translation_predictor)
    scale = mlp(reg.scale_predictor).squeeze(1)
    inplane = _NormalizeRows.apply(mlp(reg.inplane_predictor, "tanh"), 1e-12)
    return translation, scale, inplane


def infonce(tokens_src, tokens_tar, s_rows, t_rows, tau=0.1):
    return _InfoNCE.apply(tokens_src, tokens_tar, s_rows, t_rows, tau)


# ---- stage 3 under autograd (DPT head + flow decoder in training mode, unfused) ---------------------------------------------------
class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        x = _f32c(x)
        ctx.act = ACT[act]
        ctx.save_for_backward(x)
        y = torch.empty_like(x)
        _lib.check(_lib.lib().pp_act_forward(_p(x), x.numel(), ctx.act, _p(y), _lib.stream_ptr()), "pp_act_forward")
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        _lib.check(_lib.lib().pp_act_backward(_p(x), _p(_f32c(dy)), x.numel(), ctx.act, _p(dx), _lib.stream_ptr()), "pp_act_backward")
        return dx, None


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return _ew(2, _f32c(a), _f32c(b))

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _Add.apply(a, b)


class _BatchNormTrain(torch.autograd.Function):
    """nn.BatchNorm2d in training mode on an NHWC map (+ ReLU): the forward is ops.batchnorm_train (running buffers updated)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, relu):
        x = _f32c(x)
        ctx.save_for_backward(x, gamma, beta)
        ctx.relu = relu
        return ops.batchnorm_train(x, bn, relu=relu)

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta = ctx.saved_tensors
        C = x.shape[-1]
        rows = x.numel() // C
        L = _lib.lib()
        nbytes = L.pp_batchnorm_train_backward_workspace_bytes(rows, C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        dx = torch.empty_like(x)
        dg, db = torch.empty(C, dtype=torch.float32, device=x.device), torch.empty(C, dtype=torch.float32, device=x.device)
        _lib.check(L.pp_batchnorm_train_backward(_p(x), _p(gamma), _p(beta), _p(_f32c(dy)), rows, C, 1e-5, int(ctx.relu), _p(dx), _p(dg), _p(db),
                                                 _p(ws), nbytes, _lib.stream_ptr()), "pp_batchnorm_train_backward")
        return dx, dg, db, None, None


def batchnorm_train(x, bn, relu=False):
    return _BatchNormTrain.apply(x, bn.weight, bn.bias, bn, relu)


class _Resize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, mul):
        x = _f32c(x)
        ctx.geom = (x.shape, Ho, Wo, mul)
        return ops.resize_bilinear(x, Ho, Wo, mul=mul)

    @staticmethod
    def backward(ctx, dy):
        (B, H, W, C), Ho, Wo, mul = ctx.geom
        dx = torch.empty(B, H, W, C, dtype=torch.float32, device=dy.device)
        _lib.check(_lib.lib().pp_resize_bilinear_backward_nhwc(_p(_f32c(dy)), B, H, W, C, Ho, Wo, float(mul), _p(dx), _lib.stream_ptr()),
                   "pp_resize_bilinear_backward_nhwc")
        return dx, None, None, None


def resize(x, Ho, Wo, mul=1.0):
    return _Resize.apply(x, Ho, Wo, mul)


def _fixed_to_float(acc):
    out = torch.empty(acc.shape, dtype=torch.float32, device=acc.device)
    _lib.check(_lib.lib().pp_fixed_to_float(_p(acc), acc.numel(), _p(out), _lib.stream_ptr()), "pp_fixed_to_float")
    return out


class _Warp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, flow):
        feat, flow = _f32c(feat), _f32c(flow)
        ctx.save_for_backward(feat, flow)
        return ops.warp(feat, flow)

    @staticmethod
    def backward(ctx, dy):
        feat, flow = ctx.saved_tensors
        B, H, W, C = feat.shape
        dflow = torch.empty(B, H, W, 2, dtype=torch.float32, device=feat.device)
        if DETERMINISTIC:
            acc = torch.zeros(feat.shape, dtype=torch.int64, device=feat.device)
            _lib.check(_lib.lib().pp_warp_backward_nhwc_fixed(_p(feat), _p(flow), _p(_f32c(dy)), B, H, W, C, flow.shape[-1], _p(acc), _p(dflow),
                                                              _lib.stream_ptr()), "pp_warp_backward_nhwc_fixed")
            return _fixed_to_float(acc), dflow
        dfeat = torch.zeros_like(feat)
        _lib.check(_lib.lib().pp_warp_backward_nhwc(_p(feat), _p(flow), _p(_f32c(dy)), B, H, W, C, flow.shape[-1], _p(dfeat), _p(dflow),
                                                    _lib.stream_ptr()), "pp_warp_backward_nhwc")
        return dfeat, dflow


class _CorrLookup(torch.autograd.Function):
    """CorrelationPyramid + CorrLookup (raft_decoder.py:30-53, utils/corr_lookup.py:100-134), fused in both directions: the
    pyramid is never stored — level l is the correlation with the query map pooled l times."""

    @staticmethod
    def forward(ctx, f1, f2, flow, levels, r, c_pad):
        f1, f2, flow = _f32c(f1), _f32c(f2), _f32c(flow)
        ctx.save_for_backward(f1, f2, flow)
        ctx.cfg = (levels, r)
        return ops.corr_lookup(f1, f2, flow, levels, r, c_pad=c_pad)

    @staticmethod
    def backward(ctx, dout):
        import ctypes

        f1, f2, flow = ctx.saved_tensors
        levels, r = ctx.cfg
        B, H, W, C = f1.shape
        dout = _f32c(dout)
        pyr = [f2]
        for _ in range(levels - 1):
            pyr.append(ops.avgpool2(pyr[-1]))
        df1 = torch.empty_like(f1)
        dflow = torch.empty(B, H, W, 2, dtype=torch.float32, device=f1.device)
        arr = ctypes.c_void_p * 3
        fl = arr(*[_p(t) for t in pyr] + [None] * (3 - levels))
        if DETERMINISTIC:
            accs = [torch.zeros(t.shape, dtype=torch.int64, device=t.device) for t in pyr]
            dl = arr(*[_p(t) for t in accs] + [None] * (3 - levels))
            _lib.check(_lib.lib().pp_corr_lookup_backward_nhwc_fixed(_p(f1), fl, _p(flow), _p(dout), B, H, W, C, levels, r, flow.shape[-1],
                                                                     dout.shape[-1], _p(df1), dl, _p(dflow), _lib.stream_ptr()),
                       "pp_corr_lookup_backward_nhwc_fixed")
            dpyr = [_fixed_to_float(a) for a in accs]
        else:
            dpyr = [torch.zeros_like(t) for t in pyr]
            dl = arr(*[_p(t) for t in dpyr] + [None] * (3 - levels))
            _lib.check(_lib.lib().pp_corr_lookup_backward_nhwc(_p(f1), fl, _p(flow), _p(dout), B, H, W, C, levels, r, flow.shape[-1], dout.shape[-1],
                                                               _p(df1), dl, _p(dflow), _lib.stream_ptr()), "pp_corr_lookup_backward_nhwc")
        for l in range(levels - 1, 0, -1):       # the pooling chain's adjoint, coarse to fine
            t = dpyr[l - 1]
            _lib.check(_lib.lib().pp_avgpool2_backward_nhwc(_p(dpyr[l]), B, t.shape[1], t.shape[2], C, 1, _p(t), _lib.stream_ptr()),
                       "pp_avgpool2_backward_nhwc")
        return df1, dpyr[0], dflow, None, None, None


class _ConvTranspose(torch.autograd.Function):
    """ConvTranspose2d(kernel = stride = r) on NHWC (dpt.py resize_layers 0 / 1): a GEMM whose store is a pixel shuffle."""

    @staticmethod
    def forward(ctx, x, weight, bias, r):
        x = _f32c(x)
        wp, bp = ops.pack_convT_weight(weight.detach().float(), bias.detach().float())
        ctx.save_for_backward(x, wp)
        ctx.r, ctx.wshape = r, weight.shape
        return ops.conv_transpose2d(x, wp, bp, r)

    @staticmethod
    def backward(ctx, dy):
        x, wp = ctx.saved_tensors
        r = ctx.r
        B, H, W, Cin = x.shape
        Cout = wp.shape[0] // (r * r)
        d = _f32c(dy).view(B, H, r, W, r, Cout).permute(0, 1, 3, 2, 4, 5).reshape(B * H * W, r * r * Cout)   # un-shuffle (a copy)
        dx = _mm(d, wp, rb=False).view(B, H, W, Cin)
        dwp = _mm(d.t().contiguous(), x.view(-1, Cin), rb=False)                                             # (r r Cout, Cin)
        dw = dwp.view(r, r, Cout, Cin).permute(3, 2, 0, 1).contiguous()                                      # -> (Cin, Cout, r, r)
        db = colsum(d.view(-1, Cout))
        return dx, dw, db, None


class _FlowLoss(torch.autograd.Function):
    """One level of compute_stage_three_loss (utils/loss_utils.py:188-202): (loss_flow, loss_certainty) of NHWC maps."""

    @staticmethod
    def forward(ctx, flow, cert, tar_pts, flow_weight, mask_weight, max_flow, eps):
        fl, ce, tp = _f32c(flow), _f32c(cert), _f32c(tar_pts)
        B, H, W, _ = fl.shape
        L = _lib.lib()
        part = torch.empty(L.pp_flow_loss_blocks(), 3, dtype=torch.float64, device=fl.device)
        _lib.check(L.pp_flow_loss_sums(_p(fl), _p(ce), _p(tp), B, H, W, float(max_flow), _p(part), _lib.stream_ptr()), "pp_flow_loss_sums")
        bce, l1, cnt = part.sum(0)
        ctx.save_for_backward(fl, ce, tp, cnt)
        ctx.cfg = (flow_weight, mask_weight, max_flow, eps)
        return (flow_weight * l1 / (cnt + eps)).float(), (mask_weight * bce / (B * H * W)).float()

    @staticmethod
    def backward(ctx, up_f, up_c):
        fl, ce, tp, cnt = ctx.saved_tensors
        fw, mw, max_flow, eps = ctx.cfg
        B, H, W, _ = fl.shape
        gf = (up_f.double() * fw / (cnt + eps)).float().reshape(1).contiguous()      # two device scalars (no host sync)
        gc = (up_c.double() * mw / (B * H * W)).float().reshape(1).contiguous()
        dfl, dce = torch.empty_like(fl), torch.empty_like(ce)
        _lib.check(_lib.lib().pp_flow_loss_backward(_p(fl), _p(ce), _p(tp), B, H, W, float(max_flow), _p(gf), _p(gc), _p(dfl), _p(dce),
                                                    _lib.stream_ptr()), "pp_flow_loss_backward")
        return dfl, dce, None, None, None, None, None


def flow_level_losses(flow, cert, tar_pts, mask_weight=1.0, flow_weight=0.1, max_flow=400.0, eps=1e-10):
    return _FlowLoss.apply(flow, cert, tar_pts, flow_weight, mask_weight, max_flow, eps)


class _Conv2d(torch.autograd.Function):
    """Stride-1 'same' convolution (k odd, pad = k // 2) on NHWC with optional ReLU, implicit GEMM in all three directions:
    forward = the inference engine's convolution (no im2col matrix); dgrad = the same kernel on dz with the spatially flipped,
    in/out-swapped weights; wgrad = dz^T colT with the K-major im2col written only for the duration of the product."""

    @staticmethod
    def forward(ctx, x, weight, bias, k, act):
        x = _f32c(x)
        Cx = x.shape[-1]
        w = weight.detach().float()
        wp = ops.pack_conv_weight(w, cin_pad=Cx if Cx > w.shape[1] else None)
        y = ops.conv2d(x, wp, bias, k, pad=k // 2, act=act, cache_weight=(weight, "conv"))
        ctx.save_for_backward(x, w, y if act else None)
        ctx.cfg = (k, act, bias is not None)
        ctx.owner = weight
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        k, act, has_bias = ctx.cfg
        B, H, W, Cx = x.shape
        Cout, Cin = w.shape[0], w.shape[1]
        dz = _f32c(dy)
        if act:
            g = torch.empty_like(dz)
            _lib.check(_lib.lib().pp_act_backward(_p(y), _p(dz), dz.numel(), ACT[act], _p(g), _lib.stream_ptr()), "pp_act_backward")
            dz = g                                                        # relu'(z) = [y > 0]
        rows = B * H * W
        db = colsum(dz.view(-1, Cout)) if has_bias and ctx.needs_input_grad[2] else None
        Cout0 = Cout
        if Cout % 8 != 0 and Cout >= 64 and ops.PRECISION == "f16x3":
            # (the motion encoder's 126-channel convolution: its dz and weights zero-padded to 128 channels take the engine in both
            # backward products instead of the on-the-fly kernel; the pad rows of dW are dropped)
            padc = -Cout % 8
            dz = torch.nn.functional.pad(dz, (0, padc))
            w = torch.cat([w, w.new_zeros(padc, Cin, k, k)], dim=0)
            Cout += padc
        s = _scale_of(dz)                                                 # the gradient operand of both products: ONE scale
        fused = s is not None and Cout % 8 == 0 and rows % 8 == 0 and rows * Cout < 2 ** 30
        dzs = None
        if not fused:
            dzs, s = _ranged(dz)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wf = w.flip(2, 3).permute(1, 0, 2, 3)                         # (Cin, Cout, k, k): dx = conv(dz, flipped weights)
            if Cx > Cin:
                wf = torch.cat([wf, wf.new_zeros(Cx - Cin, Cout, k, k)], dim=0)
            wfp = ops.pack_conv_weight(wf.contiguous())
            if fused:     # dz split with its range scale applied (no scaled copy), the inverse scale inside the launch (no unscale pass)
                dz_op = ops.split_scaled(dz.view(rows, Cout), s)
                dz_op.image = (B, H, W)
                dx = ops.conv2d(dz_op, wfp, None, k, pad=k // 2, cache_weight=(ctx.owner, "flip"), alpha_dev=_inv(s))
            else:
                dx = ops.conv2d(dzs, wfp, None, k, pad=k // 2, cache_weight=(ctx.owner, "flip"))
                dx = _unscale(dx, s, None)
        if ctx.needs_input_grad[1]:
            tiles = -(-Cout // 128) * -(-(k * k * Cx) // 128)
            kk = k * k * Cx
            swap = Cout % 256 != 0 and kk % 256 == 0      # K slices need whole tile rows per slice: take the product transposed
            S = ops.ksplit_choice(kk, Cout, rows, can_pad=False)[0] if swap else ops.ksplit_choice(Cout, kk, rows, can_pad=False)[0]
            if fused and (tiles > 24 or S > 1) and ops.operands_ok(Cout, kk, rows) and Cx % 4 == 0:
                # dz^T as the K-major A operand in one pass (scale, transpose, split); the K-major im2col of x straight as the B
                # operand (x is a forward activation: in range as it is) — no fp32 im2col matrix, no split pass over it
                A = ops.split_transposed(dz.view(rows, Cout), s)
                Bt = ops.Split.empty(kk, rows, x.device)
                _lib.check(_lib.lib().pp_im2col_t_operand(_p(x), B, H, W, Cx, k, 1, k // 2, _p(Bt.hl), Bt.terms, _lib.stream_ptr()), "pp_im2col_t_operand")
                if swap:
                    dwp = ops.matmul_operands(Bt, A, alpha_dev=_inv(s), ksplit=S).t().contiguous()
                else:
                    dwp = ops.matmul_operands(A, Bt, alpha_dev=_inv(s), ksplit=S)
            else:
                colT = torch.empty(k * k * Cx, rows, dtype=torch.float32, device=x.device)
                _lib.check(_lib.lib().pp_im2col_t_nhwc(_p(x), B, H, W, Cx, k, 1, k // 2, _p(colT), _lib.stream_ptr()), "pp_im2col_t_nhwc")
                if dzs is None:
                    dzs = _ew(1, dz, s[0:1], 1)
                dzt = dzs.view(rows, Cout).t().contiguous()                   # (Cout, rows)
                if ops.PRECISION == "f16x3" and Cout >= 64 and tiles > 24 and rows % 8 == 0:
                    dwp = ops.matmul_nt_presplit(dzt, colT)                   # x is a forward activation: in range as it is
                else:
                    dwp = _mm_kmajor(dzt, colT)
                dwp = _unscale(dwp, s, None)                                  # (Cout, k k Cx)
            dw = dwp.view(Cout, k, k, Cx)[:Cout0, ..., :Cin].permute(0, 3, 1, 2).contiguous()
        return dx, dw, db, None, None


def _mm_kmajor(a, bt):
    """a (M,K) @ bt (N,K)^T for products with few output tiles over a long K: K cut into slices that run as one batched launch."""
    M, K = a.shape
    N = bt.shape[0]
    tiles = -(-M // 128) * -(-N // 128)
    S = 64 if (tiles <= 8 or (tiles <= 24 and K >= 65536)) else 16      # (a function of the shape: the order of the sums is fixed)
    while S > 1 and (K % S or (K // S) % 8):
        S //= 2
    kc = K // S
    part = ops.bmm_nt(a.view(M, S, kc).permute(1, 0, 2)[None], bt.view(N, S, kc).permute(1, 0, 2)[None])     # (1,S,M,N)
    r = torch.empty(M, N, dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().pp_sum_slices(_p(part), S, M, N, None, 0, _p(r), _lib.stream_ptr()), "pp_sum_slices")
    return r


def conv2d(x, weight, bias, k, stride=1, pad=0, act=None, cin_pad=None):
    """NHWC convolution under autograd: x (B,H,W,Cx), weight in torch layout (Cout, Cin, k, k) (Cx = cin_pad >= Cin: the map carries
    zero pad channels).  1x1: a linear layer on the pixels; otherwise im2col + linear (dgrad through col2im)."""
    B, H, W, Cx = x.shape
    Cout = weight.shape[0]
    if k == 1 and stride == 1 and pad == 0:
        w2 = weight.reshape(Cout, -1)
        if Cx > w2.shape[1]:
            w2 = torch.cat([w2, w2.new_zeros(Cout, Cx - w2.shape[1])], dim=1)
        return linear(x.reshape(-1, Cx), w2.contiguous(), bias, act=act, owner=weight).view(B, H, W, Cout)
    if stride == 1 and k % 2 == 1 and pad == k // 2 and act in (None, "relu") and Cx % 8 == 0:
        return _Conv2d.apply(x, weight, bias, k, act)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    w2 = ops.pack_conv_weight(weight.float(), cin_pad=Cx if Cx > weight.shape[1] else None)
    return linear(_Im2col.apply(x, k, stride, pad), w2, bias, act=act, owner=weight).view(B, Ho, Wo, Cout)


def _rcu(u, x, extra=None):
    """ResidualConvUnit in training mode (dpt.py:72-95): bn2(conv2(relu(bn1(conv1(relu(x)))))) + x (+ the fusion block's other input)."""
    h = conv2d(_Act.apply(x, "relu"), u.conv1.weight, u.conv1.bias, 3, pad=1)
    h = batchnorm_train(h, u.bn1, relu=True)
    h = conv2d(h, u.conv2.weight, u.conv2.bias, 3, pad=1)
    h = add(batchnorm_train(h, u.bn2), x)
    return h if extra is None else add(h, extra)


def dpt_head_forward(dpt, feats):
    """DPTHead.forward (dpt.py:252-272) in training mode under autograd: feats = 4 NHWC maps (B,16,16,C) -> [path_4, path_3, path_2]."""
    r, s = dpt.resize_layers, dpt.scratch
    from .model.stage3 import COMPUTE_DEAD_LAYER1

    # (layer_1 / layer_1_rn: only the SHAPE of layer_1_rn is read — dpt.py:263, 270 — and no gradient reaches projects[0], resize_layers[0]
    # or layer1_rn in the reference either (their .grad stays None): not computed, see DPTHead.forward_nhwc)
    x = {i: conv2d(feats[i], dpt.projects[i].weight, dpt.projects[i].bias, 1) for i in (range(4) if COMPUTE_DEAD_LAYER1 else range(1, 4))}
    size1 = (4 * feats[0].shape[1], 4 * feats[0].shape[2])
    l2 = _ConvTranspose.apply(x[1], getattr(r, "1").weight, getattr(r, "1").bias, 2)
    l3 = x[2]
    l4 = conv2d(x[3], getattr(r, "3").weight, getattr(r, "3").bias, 3, stride=2, pad=1)
    rn = {i: conv2d(l, getattr(s, f"layer{i + 1}_rn").weight, None, 3, pad=1) for i, l in ((1, l2), (2, l3), (3, l4))}
    if COMPUTE_DEAD_LAYER1:
        l1 = _ConvTranspose.apply(x[0], getattr(r, "0").weight, getattr(r, "0").bias, 4)
        rn[0] = conv2d(l1, s.layer1_rn.weight, None, 3, pad=1)

    def fuse(i, size, x0, x1=None):
        f = getattr(s, f"refinenet{i}")
        out = x0 if x1 is None else _rcu(f.resConfUnit1, x1, extra=x0)
        out = _rcu(f.resConfUnit2, out)
        out = resize(out, size[0], size[1])
        return conv2d(out, f.out_conv.weight, f.out_conv.bias, 1)

    p4 = fuse(4, rn[2].shape[1:3], rn[3])
    p3 = fuse(3, rn[1].shape[1:3], p4, rn[2])
    p2 = fuse(2, size1, p3, rn[1])
    dpt.bn_moved()
    return [p4, p3, p2]


def flow_decoder_forward(fd, feat_render_list, feat_real_list, flow, cert):
    """FlowDecoder.forward (flow_decoder.py:74-94, forward_flow :58-72) in training mode under autograd, NHWC."""
    flows, certs = [], []
    for l in range(fd.num_levels):
        pj, e, fp, mp = fd.proj[l], fd.encoder[l], fd.flow_pred[l], fd.mask_pred[l]
        c0, bn = getattr(pj, "0"), getattr(pj, "1")
        fr = batchnorm_train(conv2d(feat_render_list[l], c0.weight, c0.bias, 1), bn)      # render maps first (flow_decoder.py:78)
        fq = batchnorm_train(conv2d(feat_real_list[l], c0.weight, c0.bias, 1), bn)
        B, H, W, _ = fr.shape
        ncorr = (l + 1) * (2 * fd.r + 1) ** 2
        corr = _CorrLookup.apply(fr, fq, flow, l + 1, fd.r, -(-ncorr // 8) * 8)
        cn0, cn1 = getattr(e.corr_net, "0").conv, getattr(e.corr_net, "1").conv
        c = conv2d(conv2d(corr, cn0.weight, cn0.bias, 1, act="relu"), cn1.weight, cn1.bias, 3, pad=1, act="relu")
        fn0, fn1 = getattr(e.flow_net, "0").conv, getattr(e.flow_net, "1").conv
        flow8 = torch.cat([flow, flow.new_zeros(B, H, W, 6)], dim=-1)
        f = conv2d(conv2d(flow8, fn0.weight, fn0.bias, 7, pad=3, act="relu"), fn1.weight, fn1.bias, 3, pad=1, act="relu")
        on = getattr(e.out_net, "0").conv
        out = conv2d(torch.cat([c, f], dim=-1), on.weight, on.bias, 3, pad=1, act="relu")
        X = torch.cat([fr, _Warp.apply(fq, flow), out, flow], dim=-1)                       # [render | warped real | motion (126 + flow)]

        def head(hd, k_last):
            h = conv2d(X, getattr(hd.layers, "0").conv.weight, getattr(hd.layers, "0").conv.bias, 3, pad=1, act="relu")
            h = conv2d(h, getattr(hd.layers, "1").conv.weight, getattr(hd.layers, "1").conv.bias, 3, pad=1, act="relu")
            return conv2d(h, hd.predict_layer.weight, hd.predict_layer.bias, k_last, pad=k_last // 2)

        flow = add(flow, head(fp, 3))
        cert = add(cert, head(mp, 1))
        flows.append(flow)
        certs.append(cert)
        if l != fd.num_levels - 1:
            flow = resize(flow, 2 * H, 2 * W, mul=2.0)
            cert = resize(cert, 2 * H, 2 * W)
    fd.bn_moved()
    return flows, certs


def offset_regressor_forward(orr, feats_tem, feats_real, init_flow, init_cert):
    """OffsetRegressor.forward (offset_regressor.py:16-19) in training mode under autograd (template maps first)."""
    tem = dpt_head_forward(orr.dpt_head, feats_tem)
    real = dpt_head_forward(orr.dpt_head, feats_real)
    return flow_decoder_forward(orr.flow_decoder, tem, real, init_flow, init_cert)
