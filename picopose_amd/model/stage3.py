"""Stage 3: OffsetRegressor = DPTHead + FlowDecoder on the HIP engine (NHWC inside).

Mirrors model/stage3/offset_regressor.py:9-19, dpt.py:171-272 (FeatureFusionBlock :98-156,
ResidualConvUnit :40-95), flow_decoder.py:9-94, raft_decoder.py:14-53,56-161,251-289 and
utils/corr_lookup.py:69-134.  Eval BatchNorms are folded into their convolutions; the correlation
pyramid is never materialised (ops.corr_lookup); channel concatenations are written in place into
one (B,H,W,640) buffer by the producing kernels."""
import os

import torch
import torch.nn as nn

from .. import ops
from .common import Holder, Packed, bn_p, conv_p, convT_p, fold_bn, seq


# ------------------------------------------------------------------------------------------ DPT head
COMPUTE_DEAD_LAYER1 = os.environ.get("PP_DPT_DEAD_LAYER1", "0") == "1"    # see DPTHead.forward_nhwc
# FeatureFusionBlock (dpt.py:150-155): out_conv(interpolate(x)).  Both are linear and the bilinear weights of a pixel sum to one, so
# interpolate(out_conv(x)) is the same map (the bias passes through the interpolation unchanged): the 1x1 convolution then runs on a
# quarter of the pixels.  Inference only; "0": the reference's order (bench.py's all-direct exact leg keeps the reference's operation count).
OUT_CONV_FIRST = os.environ.get("PP_DPT_OUT_CONV_FIRST", "1") != "0"


def _rcu(c):
    m = Holder()
    m.conv1, m.conv2 = conv_p(c, c, 3), conv_p(c, c, 3)
    m.bn1, m.bn2 = bn_p(c), bn_p(c)
    return m


def _fusion(c):
    m = Holder()
    m.out_conv = conv_p(c, c, 1)
    m.resConfUnit1, m.resConfUnit2 = _rcu(c), _rcu(c)
    return m


class DPTHead(Packed):
    def __init__(self, nclass, in_channels, features=256, use_bn=False, out_channels=(256, 512, 1024, 1024),
                 use_clstoken=False):
        super().__init__()
        oc = list(out_channels)
        self.projects = nn.ModuleList([conv_p(in_channels, c, 1) for c in oc])
        self.resize_layers = seq((0, convT_p(oc[0], oc[0], 4)), (1, convT_p(oc[1], oc[1], 2)), (3, conv_p(oc[3], oc[3], 3)))
        s = Holder()
        for i in range(4):
            setattr(s, f"layer{i + 1}_rn", conv_p(oc[i], features, 3, bias=False))
        for i in range(4):
            setattr(s, f"refinenet{i + 1}", _fusion(features))
        s.output_conv1 = conv_p(features, features // 2, 3)        # dead weights of the checkpoint (dpt.py:241-249)
        s.output_conv2 = seq((0, conv_p(features // 2, 32, 3)), (2, conv_p(32, 1, 1)))
        self.scratch = s

    def _pack(self):
        pk = {}
        for i, p in enumerate(self.projects):
            pk[f"proj{i}"] = ops.pack_conv_weight(p.weight.float())
        r = self.resize_layers
        pk["up0"], pk["up0_b"] = ops.pack_convT_weight(getattr(r, "0").weight.float(), getattr(r, "0").bias.float())
        pk["up1"], pk["up1_b"] = ops.pack_convT_weight(getattr(r, "1").weight.float(), getattr(r, "1").bias.float())
        pk["down3"] = ops.pack_conv_weight(getattr(r, "3").weight.float())
        for i in range(4):
            pk[f"rn{i + 1}"] = ops.pack_conv_weight(getattr(self.scratch, f"layer{i + 1}_rn").weight.float())
        for i in (2, 3, 4):
            f = getattr(self.scratch, f"refinenet{i}")
            pk[f"f{i}_out"] = ops.pack_conv_weight(f.out_conv.weight.float())
            for j in (1, 2):
                u = getattr(f, f"resConfUnit{j}")
                for c in (1, 2):
                    w, b = fold_bn(getattr(u, f"conv{c}").weight.float(), getattr(u, f"conv{c}").bias.float(),
                                   getattr(u, f"bn{c}"))
                    pk[f"f{i}_u{j}_c{c}"], pk[f"f{i}_u{j}_b{c}"] = ops.pack_conv_weight(w), b.contiguous()
        return pk

    def _pack_train(self):
        pk = {}
        for i in (2, 3, 4):
            f = getattr(self.scratch, f"refinenet{i}")
            for j in (1, 2):
                u = getattr(f, f"resConfUnit{j}")
                for c in (1, 2):
                    pk[f"f{i}_u{j}_c{c}"] = ops.pack_conv_weight(getattr(u, f"conv{c}").weight.float())
        return pk

    def _rcu_train(self, i, j, x, extra=None):
        """ResidualConvUnit in training mode (dpt.py:72-95): both BatchNorms normalise with the statistics of this batch and
        update their running buffers (ops.batchnorm_train).  The forward-only training step; under autograd the same layers run through
        picopose_amd/autograd.dpt_head_forward."""
        u, pkt = getattr(getattr(self.scratch, f"refinenet{i}"), f"resConfUnit{j}"), self.packed_train()
        h = ops.conv2d(x, pkt[f"f{i}_u{j}_c1"], u.conv1.bias, 3, pad=1, relu_in=True)
        h = ops.batchnorm_train(h, u.bn1, relu=True)                        # relu(bn1(.)): the input of conv2
        h = ops.conv2d(h, pkt[f"f{i}_u{j}_c2"], u.conv2.bias, 3, pad=1)
        return ops.batchnorm_train(h, u.bn2, residual=x, residual2=extra)   # bn2(.) + x (+ the fusion block's other input)

    def _rcu(self, pk, key, x, extra=None, more=False, operand_only=False):
        """ResidualConvUnit (dpt.py:72-95): bn2(conv2(relu(bn1(conv1(relu(x)))))) + x (+ extra).
        operand_only: the unit's output feeds one 1x1 convolution and nothing else — it leaves as operand planes (f16x3 engine)."""
        # conv1 hands relu(h) to conv2 as operand planes (out_split + split_relu): h itself is never stored.  The
        # unit's fp32 output also carries its relu'd operand form (also_split) for the next unit's conv1.
        xin = getattr(x, "_hl_relu", None)
        # (strict-fp32 mode: both convolutions by Winograd F(2x2) with conv1's output transform chained into conv2's input transform —
        # wino_next="relu": through conv2's input ReLU)
        h = ops.conv2d(x if xin is None else xin, pk[key + "_c1"], pk[key + "_b1"], 3, pad=1, relu_in=xin is None,
                       out_split=True, split_relu=True, wino_next="relu")
        return ops.conv2d(h, pk[key + "_c2"], pk[key + "_b2"], 3, pad=1, relu_in=not isinstance(h, (ops.Split, ops.WinoInput)), residual=x,
                          residual2=extra, also_split="relu" if more else None, out_split=operand_only)

    def _fuse(self, pk, i, size, x0, x1=None, train=False):
        """FeatureFusionBlock (dpt.py:129-156)."""
        if train:
            out = x0 if x1 is None else self._rcu_train(i, 1, x1, extra=x0)
            out = self._rcu_train(i, 2, out)
        else:
            out = x0 if x1 is None else self._rcu(pk, f"f{i}_u1", x1, extra=x0, more=True)   # feeds resConfUnit2
            if OUT_CONV_FIRST:
                # out_conv at the block's own resolution, then the interpolation (the same map, a quarter of the rows in the GEMM);
                # the path map leaves the resize as fp32 and as the operand of the flow decoder's 1x1 projection
                out = self._rcu(pk, f"f{i}_u2", out, operand_only=True)
                out = ops.conv2d(out, pk[f"f{i}_out"], getattr(self.scratch, f"refinenet{i}").out_conv.bias, 1)
                return ops.resize_bilinear(out, size[0], size[1], also_split=True)
            out = self._rcu(pk, f"f{i}_u2", out)
        out = ops.resize_bilinear(out, size[0], size[1], out_split=True)   # feeds only the 1x1 out_conv
        # the path map is an output (fp32) AND the input of the flow decoder's 1x1 projection: its operand form rides along
        return ops.conv2d(out, pk[f"f{i}_out"], getattr(self.scratch, f"refinenet{i}").out_conv.bias, 1, also_split="plain")

    def forward_nhwc(self, feats, train=False):
        """feats: 4 NHWC maps (B,16,16,C) (may be views with a free batch stride) -> [path_4, path_3, path_2] NHWC.
        train: the ResidualConvUnits' BatchNorms run in training mode (one statistics update per call)."""
        pk, r = self.packed(for_training=train), self.resize_layers
        # (the projected maps feed only their resize layer, the resized maps only their layerK_rn convolution: operand-only
        # outputs on the f16x3 engine — no fp32 store, no split pass)
        # The reference computes layer_1 = resize_layers[0](projects[0](.)) and layer_1_rn = layer1_rn(layer_1) but reads only the SHAPE of
        # layer_1_rn (dpt.py:263, 270; refinenet1, its one consumer, is commented out at :271): three layers — 0.93 TFLOP of the 3x3
        # convolution at 64 x 64 x 192 images — whose values reach no output, no loss and no BatchNorm buffer.  They are not computed
        # (COMPUTE_DEAD_LAYER1 = True runs them anyway: the bench's A/B, test_dead_layer1_branch_does_not_reach_any_output).
        live = range(4) if COMPUTE_DEAD_LAYER1 else range(1, 4)
        x = {i: ops.conv2d(feats[i], pk[f"proj{i}"], self.projects[i].bias, 1, out_split=True) for i in live}
        _, H0, W0, _ = feats[0].shape
        size1 = (4 * H0, 4 * W0)                               # layer_1_rn.shape[2:]: ConvTranspose2d(kernel = stride = 4), then 3x3 / pad 1
        l2 = ops.conv_transpose2d(x[1], pk["up1"], pk["up1_b"], 2, out_split=True)
        l3 = x[2]
        l4 = ops.conv2d(x[3], pk["down3"], getattr(r, "3").bias, 3, stride=2, pad=1, out_split=True)
        # every layerK_rn output is the input of a ResidualConvUnit (fp32 for its skip, relu'd operand for its conv1)
        # (layer2_rn 512 -> 256 at 32 x 32 and layer3_rn 1024 -> 256 at 16 x 16: wide enough for F(4x4, 3x3) to pay — 0.93 -> 0.76 ms and
        # 0.54 -> 0.36 ms at 192 images, profiles/r06/wino4_layers.txt)
        rn = {i: ops.conv2d(l, pk[f"rn{i + 1}"], None, 3, pad=1, also_split="relu", wino=i in (1, 2)) for i, l in ((1, l2), (2, l3), (3, l4))}
        if COMPUTE_DEAD_LAYER1:
            l1 = ops.conv_transpose2d(x[0], pk["up0"], pk["up0_b"], 4, out_split=True)
            rn[0] = ops.conv2d(l1, pk["rn1"], None, 3, pad=1, also_split="relu")
            assert tuple(rn[0].shape[1:3]) == size1
        p4 = self._fuse(pk, 4, rn[2].shape[1:3], rn[3], train=train)
        p3 = self._fuse(pk, 3, rn[1].shape[1:3], p4, rn[2], train)
        p2 = self._fuse(pk, 2, size1, p3, rn[1], train)
        if train:
            self.bn_moved()              # eval re-folds the running buffers on its next call
        return [p4, p3, p2]

    def forward(self, out_features):
        """Drop-in for dpt.py:252-272: 4 x (B,C,16,16) -> [(B,256,16,16), (B,256,32,32), (B,256,64,64)]."""
        with torch.no_grad():
            return [ops.to_nchw(p) for p in self.forward_nhwc([ops.to_nhwc(f) for f in out_features])]


# ------------------------------------------------------------------------------------------ flow decoder
def _cm(cin, cout, k):
    m = Holder()  # mmcv ConvModule(norm_cfg=None): `.conv` (bias) + ReLU
    m.conv = conv_p(cin, cout, k)
    return m


def _motion_encoder(levels, radius):
    m = Holder()
    m.corr_net = seq((0, _cm(levels * (2 * radius + 1) ** 2, 256, 1)), (1, _cm(256, 192, 3)))
    m.flow_net = seq((0, _cm(2, 128, 7)), (1, _cm(128, 64, 3)))
    m.out_net = seq((0, _cm(256, 126, 3)))
    return m


# One launch for the first layers of the flow and certainty heads: PP_FUSE_XHEADS=1.  Off by default: measured
# neutral (A/B on one box, 3 + 3 runs: 398.3 / 398.4 / 398.1 crops/s with two launches, 398.0 / 397.5 / 397.6 fused —
# profiles/r04/ab_xheads.txt: the 256x256 convolution kernel is power-limited, its A tiles already come from L2), and the shared
# 1024-column hidden operand doubles the row pitch the successors address with 32-bit byte offsets (4 GB: 160 images of 64x64 fit,
# the 320 of a configs[4] share do not — the fused path then falls back to two launches).
FUSE_XHEADS = os.environ.get("PP_FUSE_XHEADS", "0") == "1"


def _xhead(cin, kind):
    m = Holder()
    m.layers = seq((0, _cm(cin, 512, 3)), (1, _cm(512, 256, 3)))
    m.predict_layer = conv_p(256, 2, 3) if kind == "flow" else conv_p(256, 1, 1)
    return m


class FlowDecoder(Packed):
    def __init__(self, num_levels, radius):
        super().__init__()
        self.num_levels, self.radius = num_levels, radius
        self.r = int(radius / 2)  # flow_decoder.py:24
        self.proj = nn.ModuleList([seq((0, conv_p(256, 256, 1)), (1, bn_p(256))) for _ in range(num_levels)])
        self.encoder = nn.ModuleList([_motion_encoder(l + 1, self.r) for l in range(num_levels)])
        self.flow_pred = nn.ModuleList([_xhead(640, "flow") for _ in range(num_levels)])
        self.mask_pred = nn.ModuleList([_xhead(640, "mask") for _ in range(num_levels)])

    def _pack(self):
        pk = {}
        for l in range(self.num_levels):
            w, b = fold_bn(getattr(self.proj[l], "0").weight.float(), getattr(self.proj[l], "0").bias.float(),
                           getattr(self.proj[l], "1"))
            pk[f"proj{l}"], pk[f"proj{l}_b"] = ops.pack_conv_weight(w), b.contiguous()
            e = self.encoder[l]
            for name, net in (("corr", e.corr_net), ("flow", e.flow_net), ("out", e.out_net)):
                for idx, sub in net.named_children():
                    w = sub.conv.weight.float()
                    # 25(l+1)-channel correlation / 2-channel flow inputs are zero-padded to a multiple of 8 channels
                    pad = -(-w.shape[1] // 8) * 8 if w.shape[1] % 8 else None
                    pk[f"e{l}_{name}{idx}"] = ops.pack_conv_weight(w, cin_pad=pad)
            # [out_net (126) | flow (2)] as operand columns: the convolution writes 128 columns (two zero filters), the flow
            # then replaces the last two
            w126 = pk[f"e{l}_out0"]
            pk[f"e{l}_out0_128"] = torch.cat([w126, w126.new_zeros(2, w126.shape[1])]).contiguous()
            b126 = getattr(e.out_net, "0").conv.bias.float()
            pk[f"e{l}_out0_b128"] = torch.cat([b126, b126.new_zeros(2)]).contiguous()
            for name, head in (("fp", self.flow_pred[l]), ("mp", self.mask_pred[l])):
                for idx, sub in head.layers.named_children():
                    pk[f"{name}{l}_{idx}"] = ops.pack_conv_weight(sub.conv.weight.float())
                pk[f"{name}{l}_p"] = ops.pack_conv_weight(head.predict_layer.weight.float())
            # the two heads' first layers read the SAME 640-channel operand (flow_decoder.py:58-72): one launch with the filters
            # concatenated along N (1024 columns) stages every A tile once for both; their successors read their halves of the
            # hidden operand as channel slices (FUSE_XHEADS)
            pk[f"x{l}_0"] = torch.cat([pk[f"fp{l}_0"], pk[f"mp{l}_0"]]).contiguous()
            pk[f"x{l}_0_b"] = torch.cat([getattr(self.flow_pred[l].layers, "0").conv.bias.float(),
                                         getattr(self.mask_pred[l].layers, "0").conv.bias.float()]).contiguous()
        return pk

    def _pack_train(self):
        return {f"proj{l}": ops.pack_conv_weight(getattr(self.proj[l], "0").weight.float()) for l in range(self.num_levels)}

    def forward_nhwc(self, feat_render_list, feat_real_list, flow, cert, train=False):
        """NHWC everywhere: lists of (B,H,W,256); flow (B,16,16,2), cert (B,16,16,1) -> lists of per-level flow/cert.
        train: the projections' BatchNorms run in training mode (render maps first, then real maps, as flow_decoder.py:78)."""
        pkt = self.packed_train() if train else None
        pk = self.packed(for_training=train)
        flows, certs = [], []
        for l in range(self.num_levels):
            fr_in, fq_in = feat_render_list[l], feat_real_list[l]
            B, H, W, _ = fr_in.shape
            # decoder input [render | warped real | motion] (640 channels).  f32 engine: an fp32 NHWC buffer X the producers
            # write their slices of.  f16x3 engine: X never exists — the producers write their columns of the OPERAND Xs the
            # two heads read (the 1x1 projection also keeps its fp32 map for the correlation lookup), which removes the
            # read + write of the whole concat by a separate split pass.
            opcat = ops.presplit()
            dev = fr_in.device
            if opcat:
                Xs = ops.Split.empty(B * H * W, 640, dev)
                Xs.image = (B, H, W)
                X = None
            else:
                X = torch.empty(B, H, W, 640, dtype=torch.float32, device=dev)   # [render | warped real | motion]
            # query maps given once for all hypotheses (hypothesis-major batch): projected once; the lookup and the
            # warp read image b % (B / hyp) of them
            e = self.encoder[l]
            fr_src = getattr(fr_in, "_hl", fr_in)
            if train:   # conv -> BatchNorm on batch statistics, render maps first (flow_decoder.py:78)
                pj = self.proj[l]
                fr = ops.batchnorm_train(ops.conv2d(fr_src, pkt[f"proj{l}"], getattr(pj, "0").bias, 1), getattr(pj, "1"))
                fq = ops.batchnorm_train(ops.conv2d(getattr(fq_in, "_hl", fq_in), pkt[f"proj{l}"], getattr(pj, "0").bias, 1), getattr(pj, "1"))
                if opcat:
                    ops.split_activation(fr, B, H * W, 256, H * W * 256, 256, into=(Xs, 0))
                else:
                    X[..., 0:256] = fr
            else:
                fq = ops.conv2d(getattr(fq_in, "_hl", fq_in), pk[f"proj{l}"], pk[f"proj{l}_b"], 1, also_split="plain" if opcat else None)
                if opcat:
                    fr = ops.conv2d(fr_src, pk[f"proj{l}"], pk[f"proj{l}_b"], 1, hl_into=(Xs, 0),
                                    out=torch.empty(B, H, W, 256, dtype=torch.float32, device=dev))
                else:
                    fr = ops.conv2d(fr_src, pk[f"proj{l}"], pk[f"proj{l}_b"], 1, out=X[..., 0:256])   # straight into its slice of X
            ncorr = (l + 1) * (2 * self.r + 1) ** 2
            # (f16x3 engine: both maps already exist as operands — the render map as columns 0..255 of Xs, the real map from
            # its projection's epilogue — so the lookup stages copies instead of splitting every chunk)
            hl_maps = dict(f1_hl=(Xs, 0), f2_hl=getattr(fq, "_hl", None)) if opcat and not train and ops.terms() == 2 else {}
            corr = ops.corr_lookup(fr, fq, flow, l + 1, self.r, c_pad=-(-ncorr // 8) * 8, **hl_maps)
            # [corr feat 192 | flow feat 64]: on the f16x3 engine the concat exists only as the operand of out_net
            hl_cat = opcat
            if hl_cat:
                cf = ops.Split.empty(B * H * W, 256, dev)
                cf.image = (B, H, W)
            else:
                cf = torch.empty(B, H, W, 256, dtype=torch.float32, device=dev)
            c1 = ops.conv2d(corr, pk[f"e{l}_corr0"], getattr(e.corr_net, "0").conv.bias, 1, act="relu", out_split=True)
            ops.conv2d(c1, pk[f"e{l}_corr1"], getattr(e.corr_net, "1").conv.bias, 3, pad=1, act="relu",
                       **(dict(hl_into=(cf, 0)) if hl_cat else dict(out=cf[..., 0:192])))
            flow8 = torch.zeros(B, H, W, 8, dtype=torch.float32, device=dev)
            flow8[..., 0:2] = flow
            f1 = ops.conv2d(flow8, pk[f"e{l}_flow0"], getattr(e.flow_net, "0").conv.bias, 7, pad=3, act="relu", out_split=True)
            ops.conv2d(f1, pk[f"e{l}_flow1"], getattr(e.flow_net, "1").conv.bias, 3, pad=1, act="relu",
                       **(dict(hl_into=(cf, 192)) if hl_cat else dict(out=cf[..., 192:256])))
            if opcat:
                # motion = cat([out_net (126), flow (2)]) (raft_decoder.py:161), written straight into its operand columns
                ops.conv2d(cf, pk[f"e{l}_out0_128"], pk[f"e{l}_out0_b128"], 3, pad=1, act="relu", hl_into=(Xs, 512))
                ops.hl_patch_columns(flow[..., 0:2].contiguous() if flow.shape[-1] != 2 else flow, Xs, 638)
                ops.warp(fq, flow, hl_into=(Xs, 256))               # feature_sample (flow_decoder.py:49-56)
            else:
                ops.conv2d(cf, pk[f"e{l}_out0"], getattr(e.out_net, "0").conv.bias, 3, pad=1, act="relu", out=X[..., 512:638])
                X[..., 638:640] = flow                                  # cat([out, flow]) (raft_decoder.py:161)
                ops.warp(fq, flow, out=X[..., 256:512])                 # feature_sample (flow_decoder.py:49-56)
                Xs = ops.split_image(X)    # both heads read the same operand: split once; hidden maps stay operand-only
            fp, mp = self.flow_pred[l], self.mask_pred[l]
            if FUSE_XHEADS and isinstance(Xs, ops.Split) and B * H * W * 1024 * 2 * Xs.terms < 0xFFFFFF00:
                hx = ops.conv2d(Xs, pk[f"x{l}_0"], pk[f"x{l}_0_b"], 3, pad=1, act="relu", out_split=True)      # (rows, 512 | 512)
                h = ops.conv2d(hx, pk[f"fp{l}_1"], getattr(fp.layers, "1").conv.bias, 3, pad=1, act="relu", out_split=True, in_cols=(0, 512))
                flow = ops.conv2d(h, pk[f"fp{l}_p"], fp.predict_layer.bias, 3, pad=1, residual=flow)      # flow + delta
                h = ops.conv2d(hx, pk[f"mp{l}_1"], getattr(mp.layers, "1").conv.bias, 3, pad=1, act="relu", out_split=True, in_cols=(512, 512))
                cert = ops.conv2d(h, pk[f"mp{l}_p"], mp.predict_layer.bias, 1, residual=cert)             # certainty + delta
            else:
                # both heads' first 3x3 convolutions share ONE Winograd input transform (strict-fp32 mode: F(2x2); f16x3 engine: F(4x4))
                Xs = ops.winograd_shared(Xs, cout=512)
                if isinstance(Xs, ops.WinoInput4):
                    # f16x3 engine: the two heads' first layers as ONE product per Winograd frequency (N = 512 + 512: U read once), each
                    # head's output transform chained into its second layer's input transform
                    hf, hm = ops.conv2d_wino_pair(Xs, (pk[f"fp{l}_0"], getattr(fp.layers, "0").conv.bias), (pk[f"mp{l}_0"], getattr(mp.layers, "0").conv.bias),
                                                  act="relu", out_split=True, wino_next=True)
                else:
                    hf = ops.conv2d(Xs, pk[f"fp{l}_0"], getattr(fp.layers, "0").conv.bias, 3, pad=1, act="relu", out_split=True, wino=True, wino_next=True)
                    hm = ops.conv2d(Xs, pk[f"mp{l}_0"], getattr(mp.layers, "0").conv.bias, 3, pad=1, act="relu", out_split=True, wino=True, wino_next=True)
                h = ops.conv2d(hf, pk[f"fp{l}_1"], getattr(fp.layers, "1").conv.bias, 3, pad=1, act="relu", out_split=True, wino=True)
                flow = ops.conv2d(h, pk[f"fp{l}_p"], fp.predict_layer.bias, 3, pad=1, residual=flow)      # flow + delta
                h = ops.conv2d(hm, pk[f"mp{l}_1"], getattr(mp.layers, "1").conv.bias, 3, pad=1, act="relu", out_split=True, wino=True)
                cert = ops.conv2d(h, pk[f"mp{l}_p"], mp.predict_layer.bias, 1, residual=cert)             # certainty + delta
            flows.append(flow)
            certs.append(cert)
            if l != self.num_levels - 1:
                flow = ops.resize_bilinear(flow, 2 * H, 2 * W, mul=2.0)
                cert = ops.resize_bilinear(cert, 2 * H, 2 * W)
        if train:
            self.bn_moved()              # eval re-folds the running buffers on its next call
        return flows, certs

    def forward(self, feat_render_list, feat_real_list, init_flow, init_certainty, iters=1):
        """Drop-in for flow_decoder.py:74-94 (NCHW in / out)."""
        assert iters == 1
        with torch.no_grad():
            fl, ce = self.forward_nhwc([ops.to_nhwc(f) for f in feat_render_list], [ops.to_nhwc(f) for f in feat_real_list],
                                       ops.to_nhwc(init_flow), ops.to_nhwc(init_certainty))
            return [ops.to_nchw(f) for f in fl], [ops.to_nchw(c) for c in ce]


class OffsetRegressor(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dpt_head = DPTHead(cfg.nclass, cfg.in_channels, features=256, use_bn=True,
                                out_channels=[256, 512, 1024, 1024], use_clstoken=False)  # offset_regressor.py:13
        self.flow_decoder = FlowDecoder(cfg.num_levels, cfg.radius)

    def forward_nhwc(self, feats_tem, feats_real, init_flow, init_cert, train=False):
        # (template maps first, then real maps: offset_regressor.py:17 — the order of the BatchNorm updates in training)
        tem = self.dpt_head.forward_nhwc(feats_tem, train)
        real = self.dpt_head.forward_nhwc(feats_real, train)
        return self.flow_decoder.forward_nhwc(tem, real, init_flow, init_cert, train)

    def forward(self, features_tem, features_real, init_flow, init_certainty):
        """Drop-in for offset_regressor.py:16-19."""
        return self.flow_decoder(self.dpt_head(features_tem), self.dpt_head(features_real), init_flow, init_certainty)
