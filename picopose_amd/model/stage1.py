"""Stage 1 backbone: DINOv2 ViT feature extractor on the HIP engine.

Mirrors model/stage1/feature_extractor.py:82-109 (FeatureExtractor) over
model/stage1/vision_transformer.py:44-228 and layers/{attention,block,mlp,layer_scale,patch_embed}.py:
patch-embed conv 14x14/s14, cls token, bicubic-resampled position embedding, `depth` pre-norm
blocks  x += ls1 * proj(softmax(q k^T / sqrt(d)) v);  x += ls2 * fc2(gelu(fc1(LN(x)))),
returning the PRE-norm outputs of the blocks listed in cfg.interaction_indexes as (B,C,16,16)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .common import Holder, Packed, conv_p, linear_p, norm_p

ARCH = {  # vision_transformer.py:336-392
    "vit_small": (384, 12, 6),
    "vit_base": (768, 12, 12),
    "vit_large": (1024, 24, 16),
    "vit_giant2": (1536, 40, 24),
}
DESCRIPTOR_MAP = {  # feature_extractor.py:12-18
    "dinov2_vits14": "vit_small",
    "dinov2_vitb14": "vit_base",
    "dinov2_vitl14": "vit_large",
    "dinov2_vitg14": "vit_giant2",
    "gigapose_dinov2": "vit_large",
}


def _block(dim):
    b = Holder()
    b.norm1 = norm_p(dim)
    b.attn = Holder()
    b.attn.qkv = linear_p(dim, 3 * dim)
    b.attn.proj = linear_p(dim, dim)
    b.ls1 = Holder()
    b.ls1.gamma = nn.Parameter(torch.ones(dim))
    b.norm2 = norm_p(dim)
    b.mlp = Holder()
    b.mlp.fc1 = linear_p(dim, 4 * dim)
    b.mlp.fc2 = linear_p(4 * dim, dim)
    b.ls2 = Holder()
    b.ls2.gamma = nn.Parameter(torch.ones(dim))
    return b


class DinoViT(Holder):
    def __init__(self, dim, depth, heads, img_size=518, patch=14):
        super().__init__()
        self.embed_dim = self.num_features = dim
        self.num_heads = heads
        self.patch_size = patch
        self.interpolate_offset = 0.1
        self.patch_embed = Holder()
        self.patch_embed.proj = conv_p(3, dim, patch)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, (img_size // patch) ** 2 + 1, dim))
        self.blocks = nn.ModuleList([_block(dim) for _ in range(depth)])
        self.norm = norm_p(dim)  # unused on this path (pre-norm outputs) but part of the checkpoint
        self.mask_token = nn.Parameter(torch.zeros(1, dim))


class FeatureExtractor(Packed):
    def __init__(self, cfg, freeze=False):
        super().__init__()
        self.cfg = cfg
        self.blocks_to_take = [blocks[-1] for blocks in cfg.interaction_indexes]
        dim, depth, heads = ARCH[DESCRIPTOR_MAP[cfg.vit_type]]
        self.dinov2 = DinoViT(dim, depth, heads)
        self.num_features = dim
        self.patch_size = self.dinov2.patch_size

    # ---- weights -------------------------------------------------------------------------
    def _pos_embed(self, w0, h0):
        """interpolate_pos_encoding (vision_transformer.py:179-207): bicubic, scale_factor=(w0+0.1)/sqrt(N).
        A constant of the weights and the input size: resampled once at pack time."""
        v = self.dinov2
        pos = v.pos_embed.float()
        N = pos.shape[1] - 1
        if w0 * h0 == N:
            return pos[0]
        sq = int(math.sqrt(N))
        sx, sy = float(w0 + v.interpolate_offset) / math.sqrt(N), float(h0 + v.interpolate_offset) / math.sqrt(N)
        grid = F.interpolate(pos[:, 1:].reshape(1, sq, sq, -1).permute(0, 3, 1, 2), scale_factor=(sx, sy),
                             mode="bicubic", antialias=False)
        assert grid.shape[-2] == w0 and grid.shape[-1] == h0
        return torch.cat([pos[0, :1], grid.permute(0, 2, 3, 1).reshape(w0 * h0, -1)], dim=0)

    def _pack(self):
        v = self.dinov2
        return {
            "patch_w": ops.pack_conv_weight(v.patch_embed.proj.weight.float(), cin_pad=8),  # RGB + 5 zero channels
            "pos": {},
        }

    def _pos_wt(self, w0, h0):
        """Transposed matrix (N, w0 h0) of the bicubic resampling above (None when the grid is the stored one): the same
        F.interpolate call applied to the N basis images — a constant of the input size, built once; the backward of the position
        embedding is one GEMM with it (picopose_amd/autograd._InterpPos)."""
        # geometry only — it does not depend on the weights, so it lives outside the packing (which every optimizer step
        # drops): keyed on the grid and the device
        v = self.dinov2
        cache = self.__dict__.setdefault("_pos_wt_cache", {})
        key = (v.pos_embed.shape[1] - 1, w0, h0, str(v.pos_embed.device))
        pk = {"pos": cache}
        if key not in pk["pos"]:
            N = v.pos_embed.shape[1] - 1
            if w0 * h0 == N:
                pk["pos"][key] = None
            else:
                sq = int(math.sqrt(N))
                sx, sy = float(w0 + v.interpolate_offset) / math.sqrt(N), float(h0 + v.interpolate_offset) / math.sqrt(N)
                with torch.no_grad():
                    basis = torch.eye(N, dtype=torch.float32, device=v.pos_embed.device).reshape(1, N, sq, sq)
                    grid = F.interpolate(basis, scale_factor=(sx, sy), mode="bicubic", antialias=False)
                pk["pos"][key] = grid.reshape(N, w0 * h0).contiguous()
        return pk["pos"][key]

    def _pos(self, w0, h0):
        pk = self.packed()
        key = (w0, h0)
        if key not in pk["pos"]:
            with torch.no_grad():
                wt = self._pos_wt(w0, h0) if self.training else None
                if wt is not None:
                    # training: every optimizer step drops the packing, and torch's bicubic kernel takes 1.7 ms for this small map —
                    # the same resampling as ONE product with the cached matrix (16 non-zero weights per output cell: the values
                    # differ from F.interpolate's by the summation order of those 16 terms)
                    pe = self.dinov2.pos_embed.float()[0]
                    pk["pos"][key] = torch.cat([pe[:1], wt.t() @ pe[1:]], dim=0).contiguous()
                else:
                    pk["pos"][key] = self._pos_embed(w0, h0).contiguous()
        return pk["pos"][key]

    # ---- forward -------------------------------------------------------------------------
    def forward_tokens(self, x, last_block_fn=None, all_blocks=False, embed_fn=None, level_out=None):
        """(B,3,H,W) -> list of token tensors (B, 1+hw, C) at the taken blocks (cls row first).
        level_out = (buffers, row0): the taken blocks' outputs are written as images row0 .. row0 + B of the caller's
        (Btot * T, C) fp32 buffers (one per taken level) — the query pass and the template pass of a forward then leave their
        levels in ONE tensor per level, which the DPT head reads as one batch (Net.forward_test).
        last_block_fn(block, xs, B, T, heads, hd) -> xs': computes the LAST block (all_blocks: every block) instead of the fused
        engine path — the training slices run them under autograd (picopose_amd/autograd.last_block_forward); embed_fn(self, x) ->
        token rows (B*T, C): the embedding under autograd (autograd.embed_tokens)."""
        v = self.dinov2
        B, _, H, W = x.shape
        p = v.patch_size
        assert H % p == 0 and W % p == 0  # patch_embed.py:73-74
        h0, w0 = H // p, W // p
        C, heads = v.embed_dim, v.num_heads
        hd = C // heads
        pk = self.packed()
        T = h0 * w0 + 1
        if embed_fn is not None:
            xs = embed_fn(self, x)
        else:
            img = ops.to_nhwc(x, c_pad=8)
            patches = ops.conv2d(img, pk["patch_w"], v.patch_embed.proj.bias, p, stride=p)        # (B,h0,w0,C)
            tok = ops.assemble_tokens(patches.view(B, h0 * w0, C), v.cls_token.reshape(C), self._pos(h0, w0))
            xs = tok.view(B * T, C)
        outs = []
        for i, blk in enumerate(v.blocks):
            if last_block_fn is not None and (all_blocks or i == len(v.blocks) - 1):
                xs = last_block_fn(blk, xs, B, T, heads, hd)
                if i in self.blocks_to_take:
                    outs.append(xs.view(B, T, C))
                continue
            # out_split: on the f16x3 engine each producer writes the next linear's operand planes directly
            h = ops.layernorm(xs, blk.norm1.weight, blk.norm1.bias, 1e-6, out_split=True)
            qkv = ops.linear(h, blk.attn.qkv.weight, blk.attn.qkv.bias, out_split=True)       # (B*T, 3*heads*hd)
            o = ops.attention(qkv, B, T, heads, hd, out_split=True)                           # fused QK^T/softmax/PV
            xs = ops.linear(o, blk.attn.proj.weight, blk.attn.proj.bias, gamma=blk.ls1.gamma, residual=xs)
            h = ops.layernorm(xs, blk.norm2.weight, blk.norm2.bias, 1e-6, out_split=True)
            f = ops.linear(h, blk.mlp.fc1.weight, blk.mlp.fc1.bias, act="gelu", out_split=True)
            dst = None
            if level_out is not None and i in self.blocks_to_take:
                dst = level_out[0][self.blocks_to_take.index(i)][level_out[1] * T:(level_out[1] + B) * T]
            xs = ops.linear(f, blk.mlp.fc2.weight, blk.mlp.fc2.bias, gamma=blk.ls2.gamma, residual=xs, out=dst)
            if i in self.blocks_to_take:
                outs.append(xs.view(B, T, C))
        return outs, (h0, w0)

    def forward(self, x):
        """Drop-in for feature_extractor.py:93-109: (B,3,224,224) -> 4 x (B,C,16,16)."""
        with torch.no_grad():
            toks, (h0, w0) = self.forward_tokens(x)
            return [ops.tokens_to_nchw(t, 1, h0, w0) for t in toks]
