"""First backward slice of the training path.

`torch.autograd.Function` wrappers whose forward AND backward run on libpicopose_hip.so: the matrix products are pp_gemm
launches (dgrad = dz @ W, wgrad = dz^T @ x through ops.bmm_nn), the row-wise adjoints are the kernels of csrc/pp_backward.hip.
torch keeps the graph, allocates tensors, makes the transposed / gathered copies and sums the gradients of a tensor that is used
twice (the residual stream, the two ViT passes over shared weights: autograd's own accumulation) — nothing else on the gradient
path is torch arithmetic.

Scope (exactly the parameters that receive a gradient from `Net.forward_train`), by `Net.train_backward`:
  * "vit+stage2" (= True, the default): the stage-1 and stage-2 losses train what the reference trains with them —
      InfoNCE (utils/loss_utils.py:144-175) and the three stage-2 losses (:177-186, through the similarity volume of
      utils/matching.py:6-26)  ->  EVERY ViT block, the patch embedding, cls token and position embedding (through the bicubic
      resampling of vision_transformer.py:179-207); the stage-2 losses  ->  every parameter of `affine_regressor`;
  * "slice1": the first slice only — stage-2 losses -> `affine_regressor` (the gradient stops at the similarity volume), InfoNCE ->
      the LAST ViT block (stops at its input).
The stage-3 losses (DPT head, flow decoder) still run forward-only, so the ViT misses their contribution.
Parity: tests/test_train_gpu.py compares these gradients with the reference's own autograd on CPU (tests/golden/train_grads.npz).
"""
import torch

from . import _lib, ops

ACT = ops.ACT


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


def _mm(a, b, alpha=1.0, out=None):
    """a (M,K) @ b (K,N) on the GEMM engine (b rows contiguous)."""
    o = ops.bmm_nn(a[None, None], b[None, None], torch.empty(1, 1, a.shape[0], b.shape[1], dtype=torch.float32, device=a.device) if out is None
                   else out[None, None], alpha=alpha)
    return o[0, 0]


class _Linear(torch.autograd.Function):
    """y = act(x @ w.T + b) with x (M,K), w (N,K): forward on the engine (activation as its own pass, so that the
    pre-activation z is kept), backward dz = dy act'(z), dx = dz @ w, dw = dz^T @ x, db = column sums of dz."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        x, w = _f32c(x), _f32c(w)
        z = ops.linear(x, w, b)
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
        dx = _mm(dz, w) if ctx.needs_input_grad[0] else None
        dw = _mm(dz.t().contiguous(), x) if ctx.needs_input_grad[1] else None
        db = colsum(dz) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db, None


def linear(x, w, b=None, act=None):
    return _Linear.apply(x, w, b, act)


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


class _Attention(torch.autograd.Function):
    """softmax((q hd^-1/2) k^T) v per head on the qkv rows (B*T, 3*heads*hd) (layers/attention.py:49-62), unfused so that the
    probabilities are kept: S and the three products are batched pp_gemm launches, soft-max rows / their adjoint row kernels."""

    @staticmethod
    def forward(ctx, qkv, B, T, heads, hd):
        qkv = _f32c(qkv)
        v5 = qkv.view(B, T, 3, heads, hd)
        q, k, v = (v5[:, :, i].permute(0, 2, 1, 3) for i in range(3))                # (B, heads, T, hd) strided views
        P = ops.softmax_rows_(ops.bmm_nt(q, k, alpha=float(hd) ** -0.5))             # (B, heads, T, T)
        out = torch.empty(B, T, heads, hd, dtype=torch.float32, device=qkv.device)
        ops.bmm_nn(P, v, out.permute(0, 2, 1, 3))
        ctx.save_for_backward(qkv, P)
        ctx.dims = (B, T, heads, hd)
        return out.view(B * T, heads * hd)

    @staticmethod
    def backward(ctx, dout):
        qkv, P = ctx.saved_tensors
        B, T, heads, hd = ctx.dims
        v5 = qkv.view(B, T, 3, heads, hd)
        q, k, v = (v5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        dO = _f32c(dout).view(B, T, heads, hd).permute(0, 2, 1, 3)
        dP = ops.bmm_nt(dO, v)                                                        # dO v^T
        dS = torch.empty_like(P)
        _lib.check(_lib.lib().pp_softmax_backward_rows(_p(P), _p(dP), B * heads * T, T, _p(dS), _lib.stream_ptr()), "pp_softmax_backward_rows")
        dqkv = torch.empty_like(qkv)
        d5 = dqkv.view(B, T, 3, heads, hd)
        s = float(hd) ** -0.5
        ops.bmm_nn(dS, k.contiguous(), d5[:, :, 0].permute(0, 2, 1, 3), alpha=s)                          # dq = dS k / sqrt(hd)
        ops.bmm_nn(dS.transpose(2, 3).contiguous(), q.contiguous(), d5[:, :, 1].permute(0, 2, 1, 3), alpha=s)   # dk = dS^T q / sqrt(hd)
        ops.bmm_nn(P.transpose(2, 3).contiguous(), dO.contiguous(), d5[:, :, 2].permute(0, 2, 1, 3))      # dv = P^T dO
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
        dq, dr = _mm(dl, r), _mm(dl.t().contiguous(), q)
        out = []
        for tok, rows_, dn in ((ts, s_rows, dq), (tt, t_rows, dr)):
            dx = torch.empty(n, C, dtype=torch.float32, device=q.device)
            _lib.check(L.pp_normalize_rows_backward(_p(tok), C, _p(rows_), _p(dn), n, C, 1e-12, _p(dx), _lib.stream_ptr()),
                       "pp_normalize_rows_backward")
            g = torch.zeros(tok.numel() // C, C, dtype=torch.float32, device=q.device)
            g.index_copy_(0, rows_, dx)        # the key-point rows are distinct (one per grid cell): a copy, not a sum
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
            _mm(wt, dpos[1:], out=out[0, 1:])
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
        ops.bmm_nn(dS[None], q.view(1, B, P, C), dr)                                # d tar_hat[t] = sum_s dS[t][s] src_hat[s]
        ops.bmm_nn(dS.transpose(1, 2).contiguous()[None], r.view(1, B, P, C), dq)   # d src_hat[s] = sum_t dS[t][s] tar_hat[t]
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
