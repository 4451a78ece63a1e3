"""GPU numerics of the network engine (GEMM / implicit-GEMM conv / norms) against plain PyTorch
fp32 CPU references of the same ops."""
import pytest
import torch
import torch.nn.functional as F

gpu = pytest.mark.gpu
TOL = 2e-4  # fp32 products/accumulation on both sides; differences are summation order only
# "f16" (plain fp16 operands, one MFMA per product — BASELINE configs[4]'s arithmetic): operands carry 11 bits, 2^-11 = 4.9e-4
# relative per element; a contraction's error stays below 4e-3 of the tensor's maximum on these inputs
TOL_F16 = 4e-3
_mode = ["f32"]


@pytest.fixture(autouse=True, params=["f32", "f16x3", "f16"])
def engine_precision(request):
    """Every engine test runs in the arithmetic modes of pp_gemm: f32 and f16x3 with the same tolerance (f16x3 keeps 22
    operand bits), f16 with the tolerance of its 11 operand bits."""
    from picopose_amd import ops

    old = ops.PRECISION
    ops.PRECISION = _mode[0] = request.param
    yield request.param
    ops.PRECISION = _mode[0] = old


def _close(a, b, tol=TOL):
    if _mode[0] == "f16":
        tol = max(tol, TOL_F16)
    a, b = a.cpu(), b.cpu()
    scale = max(1.0, float(b.abs().max()))
    assert a.shape == b.shape, (a.shape, b.shape)
    err = float((a - b).abs().max())
    assert err <= tol * scale, (err, scale)


@gpu
@pytest.mark.parametrize("M,K,N", [(257, 384, 1152), (8224 // 8, 768, 768), (5, 16384, 1024), (130, 75, 256), (64, 256, 2), (1, 256, 1)])
@pytest.mark.parametrize("act", [None, "relu", "gelu", "leaky01", "tanh"])
def test_linear(M, K, N, act):
    from picopose_amd import ops

    g = torch.Generator().manual_seed(M + K + N)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    ref = F.linear(x, w, b)
    ref = {None: lambda t: t, "relu": F.relu, "gelu": F.gelu, "leaky01": lambda t: F.leaky_relu(t, 0.1),
           "tanh": torch.tanh}[act](ref)
    _close(ops.linear(x.cuda(), w.cuda(), b.cuda(), act=act), ref)


@gpu
def test_linear_layerscale_residual_and_strided_rows():
    from picopose_amd import ops

    g = torch.Generator().manual_seed(1)
    x = torch.randn(300, 3 * 64, generator=g)
    w, b = torch.randn(96, 64, generator=g) / 8, torch.randn(96, generator=g)
    gamma, res = torch.randn(96, generator=g), torch.randn(300, 96, generator=g)
    xs = x[:, 64:128]  # a column slice: row stride 192
    ref = res + gamma * F.linear(xs, w, b)
    _close(ops.linear(x.cuda()[:, 64:128], w.cuda(), b.cuda(), gamma=gamma.cuda(), residual=res.cuda()), ref)


@gpu
@pytest.mark.parametrize("cin,cout,k,s,p,hw", [(256, 256, 1, 1, 0, 16), (256, 256, 3, 2, 1, 16), (640, 512, 3, 1, 1, 32),
                                              (2, 128, 7, 1, 3, 32), (75, 256, 1, 1, 0, 16), (3, 64, 14, 14, 0, 28),
                                              (256, 2, 3, 1, 1, 16), (256, 1, 1, 1, 0, 64), (192 + 64, 126, 3, 1, 1, 16)])
def test_conv2d_vs_torch(cin, cout, k, s, p, hw):
    from picopose_amd import ops

    g = torch.Generator().manual_seed(cin * 7 + cout + k)
    B = 2
    x = torch.randn(B, cin, hw, hw, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x, w, b, stride=s, padding=p))
    out = ops.conv2d(ops.to_nhwc(x.cuda()), ops.pack_conv_weight(w.cuda()), b.cuda(), k, s, p, act="relu")
    _close(ops.to_nchw(out), ref)


@gpu
def test_conv2d_relu_in_residual_and_concat_slices():
    from picopose_amd import ops

    g = torch.Generator().manual_seed(3)
    B, C, H = 2, 64, 16
    x = torch.randn(B, C, H, H, generator=g)
    w, b = torch.randn(C, C, 3, 3, generator=g) / 24, torch.randn(C, generator=g)
    ref = F.conv2d(F.relu(x), w, b, padding=1) + x  # ResidualConvUnit pattern (dpt.py:82-95)
    xh = ops.to_nhwc(x.cuda())
    out = ops.conv2d(xh, ops.pack_conv_weight(w.cuda()), b.cuda(), 3, 1, 1, relu_in=True, residual=xh)
    _close(ops.to_nchw(out), ref)
    if ops.presplit():
        # the same unit with an OPERAND-ONLY output (it feeds one 1x1 convolution: stage3.OUT_CONV_FIRST): residuals are added in the
        # epilogue, the fp32 map is never stored — the operand is the split of the fp32 result
        r2 = torch.randn(B, H, H, C, generator=g).cuda()
        wp = ops.pack_conv_weight(w.cuda())
        full = ops.conv2d(xh, wp, b.cuda(), 3, 1, 1, relu_in=True, residual=xh, residual2=r2)
        sp = ops.conv2d(xh, wp, b.cuda(), 3, 1, 1, relu_in=True, residual=xh, residual2=r2, out_split=True)
        assert isinstance(sp, ops.Split) and sp.image == (B, H, H)
        assert torch.equal(sp.hl, ops.split_activation(full, 1, B * H * H, C, 0, C))
    # read from / write into channel slices of wider NHWC buffers (the 640-channel concat of the flow decoder)
    wide_in = torch.randn(B, H, H, 3 * C, generator=g).cuda()
    wide_out = torch.zeros(B, H, H, 2 * C).cuda()
    ops.conv2d(wide_in[..., C:2 * C], ops.pack_conv_weight(w.cuda()), b.cuda(), 3, 1, 1, out=wide_out[..., C:], cin=C)
    ref2 = F.conv2d(wide_in[..., C:2 * C].permute(0, 3, 1, 2).cpu(), w, b, padding=1)
    _close(wide_out[..., C:].permute(0, 3, 1, 2), ref2)
    assert float(wide_out[..., :C].abs().max()) == 0.0


@gpu
@pytest.mark.parametrize("r,cin,cout", [(4, 256, 256), (2, 512, 512), (2, 64, 96)])
def test_conv_transpose(r, cin, cout):
    from picopose_amd import ops

    g = torch.Generator().manual_seed(r)
    x = torch.randn(2, cin, 16, 16, generator=g)
    w, b = torch.randn(cin, cout, r, r, generator=g) / cin ** 0.5, torch.randn(cout, generator=g)
    ref = F.conv_transpose2d(x, w, b, stride=r)
    wp, bp = ops.pack_convT_weight(w.cuda(), b.cuda())
    _close(ops.to_nchw(ops.conv_transpose2d(ops.to_nhwc(x.cuda()), wp, bp, r)), ref)


@gpu
def test_attention_products_softmax_layernorm_groupnorm():
    from picopose_amd import ops

    g = torch.Generator().manual_seed(5)
    B, T, h, hd = 3, 257, 6, 64
    qkv = torch.randn(B, T, 3, h, hd, generator=g)
    q, k, v = qkv[:, :, 0].permute(0, 2, 1, 3), qkv[:, :, 1].permute(0, 2, 1, 3), qkv[:, :, 2].permute(0, 2, 1, 3)
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    ref = (att @ v).transpose(1, 2).reshape(B, T, h * hd)
    d = qkv.cuda()
    qd, kd, vd = d[:, :, 0].permute(0, 2, 1, 3), d[:, :, 1].permute(0, 2, 1, 3), d[:, :, 2].permute(0, 2, 1, 3)
    s = ops.bmm_nt(qd, kd, alpha=hd ** -0.5)
    _close(s, (q * hd ** -0.5) @ k.transpose(-2, -1))
    ops.softmax_rows_(s)
    _close(s, att, 1e-5)
    o = torch.empty(B, T, h, hd, device="cuda")
    ops.bmm_nn(s, vd, o.permute(0, 2, 1, 3))
    _close(o.reshape(B, T, h * hd), ref)
    fused = ops.attention(d.reshape(B * T, 3 * h * hd), B, T, h, hd)       # the fused kernel the ViT uses
    _close(fused.reshape(B, T, h * hd), ref, 1e-5)
    x = torch.randn(500, 384, generator=g) * 3 + 1
    w, b = torch.randn(384, generator=g), torch.randn(384, generator=g)
    _close(ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-6), F.layer_norm(x, (384,), w, b, 1e-6), 1e-5)
    xi = torch.randn(2, 256, 16, 16, generator=g)
    wg, bg = torch.randn(256, generator=g), torch.randn(256, generator=g)
    _close(ops.to_nchw(ops.groupnorm(ops.to_nhwc(xi.cuda()), wg.cuda(), bg.cuda(), 32, relu=True)),
           F.relu(F.group_norm(xi, 32, wg, bg)), 1e-5)


@gpu
def test_lds_dma_kernel_pinned(monkeypatch, engine_precision):
    """The 256x128 LDS-DMA kernel (normally chosen by the autotuner for chip-filling problems only) pinned on
    shapes with row / column / K tails, padded and strided taps, a Cin that is not a multiple of the K tile
    and the pixel-shuffle store."""
    if engine_precision == "f32":
        pytest.skip("pre-split operands exist in f16x3 mode only")
    from picopose_amd import ops

    _pinned_big_kernel_cases(monkeypatch, "3")


@gpu
def test_persistent_lds_dma_kernel_pinned(monkeypatch, engine_precision):
    """Same cases on the persistent form (configuration 4; shapes it does not cover fall back to configuration 3)."""
    if engine_precision == "f32":
        pytest.skip("pre-split operands exist in f16x3 mode only")
    _pinned_big_kernel_cases(monkeypatch, "4")


@gpu
@pytest.mark.parametrize("cfg", ["7", "8"])
def test_multi_workgroup_per_cu_kernels_pinned(monkeypatch, engine_precision, cfg):
    """Same cases on the two- (256x128 tiles, configuration 7) and three-workgroups-per-CU (128x128, configuration 8) kernels:
    dense layers and the convolutions they cover (Cin a multiple of 32: padded / strided taps, 1x1), tails, fall-backs."""
    if engine_precision == "f32":
        pytest.skip("pre-split operands exist in f16x3 mode only")
    _pinned_big_kernel_cases(monkeypatch, cfg)


@gpu
def test_persistent_256x256_kernel_pinned(monkeypatch, engine_precision):
    """Same cases on the 256x256-tile persistent kernel (configuration 5; falls back where it does not apply)."""
    if engine_precision == "f32":
        pytest.skip("pre-split operands exist in f16x3 mode only")
    _pinned_big_kernel_cases(monkeypatch, "5")


@gpu
def test_row_shared_conv3x3_kernel_pinned(monkeypatch, engine_precision):
    """Configuration 6: 3x3 / stride 1 / pad 1 convolutions on the persistent 256x256 kernel with row-shared A delivery
    (one LDS copy of the pixels per filter ROW, the three taps read it at shifted rows, edge lanes zeroed).  Against
    torch's conv2d for W = 64 / 32 / 16 / 8, batches that leave a partial last tile, Cout with a column tail; and
    bit-identical to configuration 5 (same K order, same MFMA order)."""
    if engine_precision == "f32":
        pytest.skip("pre-split operands exist in f16x3 mode only")
    from picopose_amd import ops

    g = torch.Generator().manual_seed(78)
    for B, cin, cout, hw in [(2, 64, 256, 64), (3, 640, 512, 32), (5, 256, 192, 16), (1, 32, 520, 64), (7, 96, 256, 8), (1, 128, 256, 16)]:
        x = torch.randn(B, cin, hw, hw, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
        b = torch.randn(cout, generator=g)
        ref = F.relu(F.conv2d(x, w, b, padding=1))
        xn, wp = ops.to_nhwc(x.cuda()), ops.pack_conv_weight(w.cuda())
        monkeypatch.setenv("PP_GEMM_FORCE_CFG", "6")
        out6 = ops.conv2d(xn, wp, b.cuda(), 3, 1, 1, act="relu")
        monkeypatch.setenv("PP_GEMM_FORCE_CFG", "5")
        out5 = ops.conv2d(xn, wp, b.cuda(), 3, 1, 1, act="relu")
        _close(ops.to_nchw(out6), ref)
        assert torch.equal(out6, out5), (B, cin, cout, hw, float((out6 - out5).abs().max()))
    _pinned_big_kernel_cases(monkeypatch, "6")     # shapes it does not apply to fall back to the other kernels


def _pinned_big_kernel_cases(monkeypatch, cfg):
    from picopose_amd import ops

    monkeypatch.setenv("PP_GEMM_FORCE_CFG", cfg)
    g = torch.Generator().manual_seed(77)
    for M, K, N in [(257, 384, 1152), (1000, 768, 768), (300, 72, 130), (5, 4096, 64), (513, 32, 129), (70000, 96, 256)]:
        x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
        _close(ops.linear(x.cuda(), w.cuda(), b.cuda(), act="gelu"), F.gelu(F.linear(x, w, b)))
    for cin, cout, k, s, p, hw in [(640, 512, 3, 1, 1, 32), (256, 256, 3, 2, 1, 16), (256, 256, 1, 1, 0, 16), (72, 136, 3, 1, 1, 20),
                                   (8, 64, 7, 1, 3, 32)]:
        x = torch.randn(3, cin, hw, hw, generator=g)
        w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
        b = torch.randn(cout, generator=g)
        res = torch.randn(3, cout, (hw + 2 * p - k) // s + 1, (hw + 2 * p - k) // s + 1, generator=g)
        ref = res + F.relu(F.conv2d(x, w, b, stride=s, padding=p))
        out = ops.conv2d(ops.to_nhwc(x.cuda()), ops.pack_conv_weight(w.cuda()), b.cuda(), k, s, p, act="relu",
                         residual=ops.to_nhwc(res.cuda()))
        _close(ops.to_nchw(out), ref)
    x, w, b = torch.randn(2, 64, 16, 16, generator=g), torch.randn(64, 96, 2, 2, generator=g) / 16, torch.randn(96, generator=g)
    wp, bp = ops.pack_convT_weight(w.cuda(), b.cuda())
    _close(ops.to_nchw(ops.conv_transpose2d(ops.to_nhwc(x.cuda()), wp, bp, 2)), F.conv_transpose2d(x, w, b, stride=2))


@gpu
@pytest.mark.parametrize("rows,C", [(501, 384), (1030, 768), (257, 1024), (7, 40), (66, 100), (9, 2048)])
def test_layernorm_widths_and_operand_output(rows, C, engine_precision):
    """Register-resident rows (C a multiple of 8 up to 1024: partial and full lane groups) and the any-width path;
    the operand written by the kernel equals the split of its own fp32 output."""
    from picopose_amd import ops

    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g) * 3 + 1.5
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    y = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-6)
    _close(y, F.layer_norm(x, (C,), w, b, 1e-6), 1e-5)
    if engine_precision != "f32" and C % 8 == 0:
        sp = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-6, out_split=True)
        assert torch.equal(sp.hl, ops.split_activation(y, 1, rows, C, 0, C))


@gpu
def test_fused_operand_planes_equal_separate_split(engine_precision):
    """layernorm / attention / GEMM epilogues that write the next GEMM's f16x3 operand planes directly give
    bit-identical results to the fp32 tensor + separate split pass."""
    if engine_precision == "f32":
        pytest.skip("operands exist in the f16x3 / f16 modes only")
    from picopose_amd import ops

    g = torch.Generator().manual_seed(5)
    B, T, heads, hd = 3, 257, 6, 64
    C = heads * hd
    x = torch.randn(B * T, C, generator=g).cuda()
    lw, lb = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    wqkv, bqkv = (torch.randn(3 * C, C, generator=g) / C ** 0.5).cuda(), torch.randn(3 * C, generator=g).cuda()
    wp = (torch.randn(C, C, generator=g) / C ** 0.5).cuda()
    w1, b1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).cuda(), torch.randn(4 * C, generator=g).cuda()
    w2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).cuda()

    def chain(fused):
        h = ops.layernorm(x, lw, lb, 1e-6, out_split=fused)
        assert isinstance(h, ops.Split) == fused
        qkv = ops.linear(h, wqkv, bqkv, out_split=fused)
        o = ops.attention(qkv, B, T, heads, hd, out_split=fused)
        y = ops.linear(o, wp, None, residual=x)
        f = ops.linear(ops.layernorm(y, lw, lb, 1e-6, out_split=fused), w1, b1, act="gelu", out_split=fused)
        assert isinstance(f, ops.Split) == fused
        return ops.linear(f, w2, None, residual=y)

    a, b = chain(True), chain(False)
    if engine_precision == "f16":   # (the unfused chain's attention reads the fp32 qkv: the 3-term kernel; the fused one the h operand)
        _close(a, b)
    else:
        assert torch.equal(a, b)


@gpu
@pytest.mark.parametrize("cfg", ["3", "4", "5", "6", "7"])
def test_fp32_engine_kernels_pinned(monkeypatch, engine_precision, cfg):
    """The fp32 engine (csrc/pp_gemm_f.hip: LDS-DMA ring + v_mfma_f32_32x32x2_f32; configurations 3 = 128x128, 4 = 256x128,
    5 = 256x256, 6 = 128x64) pinned on shapes with row / column / K tails, padded and strided taps, Cin that is not a multiple of
    the K tile, residuals and the pixel-shuffle store — the cases of the pre-split kernels' pinned tests, in `f32` mode."""
    if engine_precision != "f32":
        pytest.skip("the fp32 engine serves ops.PRECISION = 'f32'")
    _pinned_big_kernel_cases(monkeypatch, cfg)


@gpu
def test_fp32_engine_agrees_bitwise_across_tile_configurations_and_with_the_round1_kernel(monkeypatch, engine_precision):
    """Every tile configuration of the fp32 engine accumulates an output element in the same order, so its value does not depend on
    the autotuner's choice (what lets `Net` batch hypotheses in exact mode too).  Dense products also reproduce the round-1
    gemm_kernel (configuration 0: same K tiles, same pairs per MFMA) bit for bit; convolutions with Cin % 32 == 0 walk K
    channel-slice-major here and are compared with a tolerance.  Also: ReLU on the A operand, two residuals, LayerScale, strided rows."""
    if engine_precision != "f32":
        pytest.skip("the fp32 engine serves ops.PRECISION = 'f32'")
    from picopose_amd import ops

    g = torch.Generator().manual_seed(10)
    x, w, b = torch.randn(700, 768, generator=g).cuda(), (torch.randn(384, 768, generator=g) / 27).cuda(), torch.randn(384, generator=g).cuda()
    gam, res = torch.randn(384, generator=g).cuda(), torch.randn(700, 384, generator=g).cuda()
    xi = torch.randn(2, 24, 24, 64, generator=g).cuda()
    wc = ops.pack_conv_weight((torch.randn(256, 64, 3, 3, generator=g) / 24).cuda())
    xo = torch.randn(3, 20, 20, 72, generator=g).cuda()                                       # Cin % 32 != 0: natural K order
    wo = ops.pack_conv_weight((torch.randn(136, 72, 3, 3, generator=g) / 25).cuda())
    r1, r2 = torch.randn(2, 24, 24, 256, generator=g).cuda(), torch.randn(2, 24, 24, 256, generator=g).cuda()
    x2, w2 = torch.randn(1300, 396, generator=g).cuda(), (torch.randn(200, 396, generator=g) / 20).cuda()   # K % 32 == 12, row / column tails
    wide = torch.randn(700, 1000, generator=g).cuda()
    xs = wide[:, 8:8 + 768]                                                                    # strided rows (16-byte aligned start)

    def run():
        return (ops.linear(x, w, b, act="gelu"), ops.linear(x, w, b, gamma=gam, residual=res), ops.linear(x2, w2, None),
                ops.linear(xs, w, b, act="leaky01"),
                ops.conv2d(xi, wc, None, 3, pad=1, act="relu", relu_in=True, residual=r1, residual2=r2),
                ops.conv2d(xo, wo, None, 3, pad=1))

    outs = {}
    for cfg in ("0", "3", "4", "5", "6", "7"):
        monkeypatch.setenv("PP_GEMM_FORCE_CFG", cfg)
        outs[cfg] = run()
    for cfg in ("4", "5", "6", "7"):
        for a, b_ in zip(outs[cfg], outs["3"]):
            assert torch.equal(a, b_), cfg
    for k in (0, 2, 3):                      # dense: the round-1 kernel's bits
        assert torch.equal(outs["3"][k], outs["0"][k]), k
    _close(outs["3"][1], outs["0"][1], 1e-6)  # (LayerScale + residual: one fma here, a product and a sum there)
    assert torch.equal(outs["3"][5], outs["0"][5])       # natural K order: the same chain as the round-1 kernel
    _close(outs["3"][4], outs["0"][4], 2e-6)
    ref = F.gelu(F.linear(x.cpu(), w.cpu(), b.cpu()))
    _close(outs["3"][0], ref)
    _close(outs["3"][1], res.cpu() + gam.cpu() * F.linear(x.cpu(), w.cpu(), b.cpu()))
    _close(outs["3"][2], x2.cpu() @ w2.cpu().t())
    _close(outs["3"][3], F.leaky_relu(F.linear(xs.cpu(), w.cpu(), b.cpu()), 0.1))


@gpu
@pytest.mark.parametrize("nb,M,N,K", [(16, 256, 96, 64), (5, 512, 130, 36), (3, 768, 512, 640)])
def test_fp32_batch_as_one_grouped_launch_equals_the_products_one_by_one(nb, M, N, K):
    """pp_gemm, prec = PP_PREC_F32, batch0 = nb with the A and C blocks one behind the other (the sixteen products of a Winograd
    convolution): ONE persistent launch of the fp32 engine whose row tiles read the weights of their own product (include/picopose_hip.h
    grp_rows / grp_b_bytes) — the same bits as nb separate calls, with a bias and an activation shared by the products, and against float64."""
    from picopose_amd import ops

    g = torch.Generator().manual_seed(nb * 100 + N)
    A = torch.randn(nb, M, K, generator=g).cuda()
    W = (torch.randn(nb, N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    one, many = torch.empty(nb, M, N, device="cuda"), torch.full((nb, M, N), float("nan"), device="cuda")
    for z in range(nb):
        ops._run(ops._desc(A=ops._p(A[z]), B=ops._p(W[z]), C=ops._p(one[z]), bias=ops._p(b), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, prec=0, act=ops.ACT["relu"]))
    ops._run(ops._desc(A=ops._p(A), B=ops._p(W), C=ops._p(many), bias=ops._p(b), M=M, N=N, K=K, lda=K, ldb=K, ldc=N, prec=0, act=ops.ACT["relu"],
                       batch0=nb, a_bs0=M * K, b_bs0=N * K, c_bs0=M * N))
    ref = F.relu(torch.einsum("zmk,znk->zmn", A.cpu().double(), W.cpu().double()) + b.cpu().double())
    assert float((many.cpu().double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    assert torch.equal(one, many)


@gpu
@pytest.mark.parametrize("B,cin,cout,hw", [(3, 64, 96, 32), (2, 640, 512, 16), (1, 256, 126, 64), (5, 32, 32, 8)])
@torch.no_grad()      # (the Winograd paths serve inference: ops._winograd_ok / _winograd4_ok are off under autograd)
def test_winograd_3x3_of_the_strict_fp32_mode(monkeypatch, engine_precision, B, cin, cout, hw):
    """ops.PRECISION = "f32": the large 3x3 / stride 1 / pad 1 convolutions run as Winograd F(2x2, 3x3) (csrc/pp_winograd.hip: input
    transform, 16 dense fp32 products on the engine, output transform) — against float64 torch and against the direct implicit-GEMM
    convolution it replaces (ops.WINOGRAD = False), with bias, ReLU, ReLU on the input, two residuals, a channel-slice input, a
    channel-slice output and a channel count that is not a multiple of 4.  Error against float64 within 3 x the direct convolution's."""
    if engine_precision != "f32":
        pytest.skip("Winograd serves ops.PRECISION = 'f32'")
    from picopose_amd import ops

    monkeypatch.setattr(ops, "WINOGRAD_MIN_PIXELS", 0)
    g = torch.Generator().manual_seed(B * 1000 + cin + cout)
    x = torch.randn(B, hw, hw, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    r1, r2 = torch.randn(B, hw, hw, cout, generator=g), torch.randn(B, hw, hw, cout, generator=g)
    wp = ops.pack_conv_weight(w.cuda())
    wide_in = torch.zeros(B, hw, hw, cin + 24)
    wide_in[..., 8:8 + cin] = x
    wide_in = wide_in.cuda()
    xd, wd = x.permute(0, 3, 1, 2).double(), w.double()

    def both(fn):
        monkeypatch.setattr(ops, "WINOGRAD", True)
        a = fn()
        monkeypatch.setattr(ops, "WINOGRAD", False)
        d = fn()
        monkeypatch.setattr(ops, "WINOGRAD", True)
        return a, d

    cases = [
        (lambda: ops.conv2d(x.cuda(), wp, b.cuda(), 3, pad=1, act="relu"),
         F.relu(F.conv2d(xd, wd, b.double(), padding=1))),
        (lambda: ops.conv2d(wide_in[..., 8:8 + cin], wp, None, 3, pad=1, relu_in=True, residual=r1.cuda(), residual2=r2.cuda()),
         F.conv2d(F.relu(xd), wd, None, padding=1) + r1.permute(0, 3, 1, 2).double() + r2.permute(0, 3, 1, 2).double()),
    ]
    for fn, ref in cases:
        got, direct = both(fn)
        ref = ref.permute(0, 2, 3, 1)
        scale = float(ref.abs().max())
        e_w, e_d = float((got.cpu().double() - ref).abs().max()) / scale, float((direct.cpu().double() - ref).abs().max()) / scale
        assert not torch.equal(got, direct), "the Winograd path did not run"
        assert e_w <= max(3 * e_d, 2e-6), (e_w, e_d)
        # a pinned round-1 configuration (tests / tools pin 0 .. 2) must not send the GROUPED batch to the round-1 kernel, which knows
        # nothing of groups and would multiply every frequency by frequency 0's weights: same bits as unpinned
        monkeypatch.setenv("PP_GEMM_FORCE_CFG", "0")
        pinned = fn()
        monkeypatch.delenv("PP_GEMM_FORCE_CFG")
        assert torch.equal(pinned, got)
        assert e_w <= 2e-5
    # conv -> (ReLU) -> conv with the first output transform chained into the second input transform (pp_winograd_chain_f32: the hidden
    # map never stored): the same bits as the separate kernels, with and without the second layer's input ReLU
    w2p = ops.pack_conv_weight((torch.randn(64, cout, 3, 3, generator=g) / (cout * 9) ** 0.5).cuda())
    for nxt, act in (("relu", None), (True, "relu"), (True, "leaky01")):
        monkeypatch.setattr(ops, "WINO2_CHAIN", True)
        ch = ops.conv2d(x.cuda(), wp, b.cuda(), 3, pad=1, act=act, wino_next=nxt)
        assert isinstance(ch, ops.WinoInput) == (hw in (16, 32, 64) and cout % 32 == 0)
        a_ = ops.conv2d(ch, w2p, None, 3, pad=1, relu_in=(nxt == "relu") and not isinstance(ch, ops.WinoInput))
        monkeypatch.setattr(ops, "WINO2_CHAIN", False)
        sep = ops.conv2d(x.cuda(), wp, b.cuda(), 3, pad=1, act=act, wino_next=nxt)
        assert torch.is_tensor(sep)
        assert torch.equal(a_, ops.conv2d(sep, w2p, None, 3, pad=1, relu_in=nxt == "relu"))
    monkeypatch.setattr(ops, "WINO2_CHAIN", True)
    # a channel-slice output of a wider NHWC buffer
    if cout % 4 == 0:
        buf = torch.zeros(B, hw, hw, cout + 40, device="cuda")
        ops.conv2d(x.cuda(), wp, b.cuda(), 3, pad=1, out=buf[..., 16:16 + cout])
        ref = F.conv2d(xd, wd, b.double(), padding=1).permute(0, 2, 3, 1)
        assert float((buf[..., 16:16 + cout].cpu().double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
        assert not buf[..., :16].any() and not buf[..., 16 + cout:].any()


@gpu
@pytest.mark.parametrize("B,cin,cout,hw", [(16, 64, 64, 16), (4, 640, 512, 32), (1, 512, 256, 64), (64, 32, 40, 8), (3, 64, 48, 20)])
@torch.no_grad()
def test_winograd4_3x3_of_the_f16x3_engine(monkeypatch, engine_precision, B, cin, cout, hw):
    """ops.PRECISION = "f16x3": wide 3x3 / stride 1 / pad 1 convolutions on an operand image run as Winograd F(4x4, 3x3)
    (csrc/pp_winograd.hip: operand -> operand input transform, 36 dense products on the pre-split engine as grouped launches, output
    transform) — against float64 torch and against the direct implicit-GEMM convolution it replaces (ops.WINOGRAD4 = False): operand
    output with the consumer's ReLU folded in, fp32 output with residuals and an attached operand, a channel slice of a wider operand as
    input, and one input transform shared by two convolutions.  F(4x4)'s fp32 transforms cost ~1e-5 of the maximum per layer
    (tools/wino_precision_study.py); the bar here is 4e-5, the f16x3 mode's network bars are 2e-4 / 5e-4."""
    if engine_precision != "f16x3":
        pytest.skip("F(4x4, 3x3) serves ops.PRECISION = 'f16x3'")
    from picopose_amd import ops

    monkeypatch.setattr(ops, "CHECK_SATURATION", True)
    g = torch.Generator().manual_seed(B * 1000 + cin + cout)
    x = torch.randn(B, hw, hw, cin, generator=g) * 3.0
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    b = torch.randn(cout, generator=g)
    r1, r2 = torch.randn(B, hw, hw, cout, generator=g), torch.randn(B, hw, hw, cout, generator=g)
    wp = ops.pack_conv_weight(w.cuda())
    xs = ops.split_image(x.cuda())
    assert isinstance(xs, ops.Split)
    wide = torch.zeros(B, hw, hw, cin + 24)
    wide[..., 8:8 + cin] = x
    wide_s = ops.split_image(wide.cuda())
    xq = (xs.hl.view(-1, cin // 8, 2, 8).float().sum(2).view(B, hw, hw, cin) / 4).cpu()    # the operand's values (22 bits of x)
    xd, wd = xq.permute(0, 3, 1, 2).double(), w.double()

    def both(fn):
        monkeypatch.setattr(ops, "WINOGRAD4", True)
        a = fn()
        monkeypatch.setattr(ops, "WINOGRAD4", False)
        d = fn()
        monkeypatch.setattr(ops, "WINOGRAD4", True)
        return a, d

    def as_f32(t):
        if isinstance(t, ops.Split):
            return (t.hl.view(-1, t.shape[1] // 8, 2, 8).float().sum(2).view(B, hw, hw, -1) / 4).cpu().double()
        return t.cpu().double()

    ref_plain = F.conv2d(xd, wd, b.double(), padding=1)
    cases = [
        (lambda: ops.conv2d(xs, wp, b.cuda(), 3, pad=1, act="relu", out_split=True, split_relu=True, wino=True), F.relu(ref_plain)),
        (lambda: ops.conv2d(xs, wp, b.cuda(), 3, pad=1, act="leaky01", residual=r1.cuda(), residual2=r2.cuda(), wino=True),
         F.leaky_relu(ref_plain, 0.1) + r1.permute(0, 3, 1, 2).double() + r2.permute(0, 3, 1, 2).double()),
        (lambda: ops.conv2d(wide_s, wp, None, 3, pad=1, in_cols=(8, cin), wino=True), F.conv2d(xd, wd, None, padding=1)),
    ]
    for fn, ref in cases:
        got, direct = both(fn)
        ref = ref.permute(0, 2, 3, 1)
        scale = float(ref.abs().max())
        e_w, e_d = float((as_f32(got) - ref).abs().max()) / scale, float((as_f32(direct) - ref).abs().max()) / scale
        assert e_d <= 4e-6, e_d
        assert e_w <= 4e-5, (e_w, e_d)
        assert float((as_f32(got) - as_f32(direct)).abs().max()) > 0, "the Winograd path did not run"
    # fp32 output that also carries its relu'd operand form
    o = ops.conv2d(xs, wp, b.cuda(), 3, pad=1, also_split="relu", wino=True)
    ref = ref_plain.permute(0, 2, 3, 1)
    assert float((o.cpu().double() - ref).abs().max()) <= 4e-5 * float(ref.abs().max())
    assert float((as_f32(o._hl_relu) - F.relu(ref)).abs().max()) <= 4e-5 * float(ref.abs().max())
    # one input transform shared by two convolutions: the same bits as two separate ones
    w2p = ops.pack_conv_weight((torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).cuda())
    sh = ops.winograd_shared(xs, cout=cout)
    assert isinstance(sh, ops.WinoInput4)
    for wq in (wp, w2p):
        assert torch.equal(ops.conv2d(sh, wq, b.cuda(), 3, pad=1, act="relu"), ops.conv2d(xs, wq, b.cuda(), 3, pad=1, act="relu", wino=True))
    # conv -> ReLU -> conv with the first output transform CHAINED into the second input transform (the hidden map never stored): the
    # same bits as the two separate kernels, with the consumer's ReLU folded in or not
    monkeypatch.setattr(ops, "WINO4_CHAIN_WIDTHS", (16, 32, 64))       # (the product chains at 32 and 64 only: at 16 it is slower)
    w3p = ops.pack_conv_weight((torch.randn(64, cout, 3, 3, generator=g) / (cout * 9) ** 0.5).cuda())
    for kw in (dict(act="relu"), dict(act="leaky01", split_relu=True), dict(act=None)):
        ch = ops.conv2d(xs, wp, b.cuda(), 3, pad=1, out_split=True, wino=True, wino_next=True, **kw)
        assert isinstance(ch, ops.WinoInput4) == (hw in (16, 32, 64) and cout % 32 == 0)
        sep = ops.conv2d(xs, wp, b.cuda(), 3, pad=1, out_split=True, wino=True, **kw)
        assert isinstance(sep, ops.Split)
        assert torch.equal(ops.conv2d(ch, w3p, None, 3, pad=1, wino=True), ops.conv2d(sep, w3p, None, 3, pad=1, wino=True))
    # two layers that read the same operand as ONE product per frequency (filters concatenated along N): each layer's result equals its
    # own product's — bit for bit when the operand scales coincide, else within fp16 subnormals of the weights' lo terms (1e-6)
    for nxt in (False, True):
        pa, pb = ops.conv2d_wino_pair(sh, (wp, b.cuda()), (w2p, None), act="relu", out_split=True, wino_next=nxt)
        sa = ops.conv2d(sh, wp, b.cuda(), 3, pad=1, act="relu", out_split=True, wino=True, wino_next=nxt)
        sb = ops.conv2d(sh, w2p, None, 3, pad=1, act="relu", out_split=True, wino=True, wino_next=nxt)
        for got_, want_ in ((pa, sa), (pb, sb)):
            assert type(got_) is type(want_)
            g_, w_ = (got_.U.hl, want_.U.hl) if isinstance(got_, ops.WinoInput4) else (got_.hl, want_.hl)
            assert float((g_.float() - w_.float()).abs().max()) <= 1e-6 * float(w_.float().abs().max()) + 1e-7
    # a grouped launch cut into several (32-bit byte offsets of the stacked blocks): the same bits
    one = ops.conv2d(xs, wp, b.cuda(), 3, pad=1, wino=True)
    monkeypatch.setattr(ops, "WINO4_GROUPS_PER_LAUNCH", 7)
    assert torch.equal(ops.conv2d(xs, wp, b.cuda(), 3, pad=1, wino=True), one)
    # the result of a crop does not depend on the batch it is in (other row-tile counts, other pad rows)
    if B > 1:
        xs1 = ops.split_image(x[B - 1:].cuda())
        assert torch.equal(ops.conv2d(xs1, wp, b.cuda(), 3, pad=1, wino=True), one[B - 1:])


@gpu
def test_tail_split_launches_equal_the_single_launch_bitwise(monkeypatch, engine_precision):
    """Configurations 9 / 10 of pp_gemm (round 6): the rows of the whole rounds of a persistent launch on the 256x256 / 256x128 tile, the
    remaining rows as a second launch on 128x128 tiles (the ViT linears at M = 49 344 leave their last round of 256-row tiles 5-50 % filled).
    Same bits as the single launch — fp32 output with residual and LayerScale, operand output, both engines."""
    from picopose_amd import ops

    g = torch.Generator().manual_seed(21)
    M, K, N = 256 * 100 + 77, 256, 768          # 101 x 3 tiles of 256 x 256: one whole round of 256 slots + 47 tiles
    x = torch.randn(M, K, generator=g).cuda()
    w, b = (torch.randn(N, K, generator=g) / 16).cuda(), torch.randn(N, generator=g).cuda()
    gam, res = torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    outs = []
    for cfg in ("5", "9", "10", "4"):
        monkeypatch.setenv("PP_GEMM_FORCE_CFG", cfg)
        o = [ops.linear(x, w, b, act="gelu", gamma=gam, residual=res)]
        if engine_precision != "f32":
            sp = ops.linear(ops.Split(ops.split_activation(x, 1, M, K, 0, K)), w, b, act="relu", out_split=True)
            o.append(sp.hl)
        outs.append(o)
    for o in outs[1:]:
        for a_, b_ in zip(o, outs[0]):
            assert torch.equal(a_, b_)
    ref = res.cpu() + gam.cpu() * F.gelu(F.linear(x.cpu(), w.cpu(), b.cpu()))
    _close(outs[0][0], ref)


@gpu
def test_presplit_kernels_agree_bitwise_across_tile_configurations(monkeypatch, engine_precision):
    """The three pre-split kernels (128x128, 128x64, 256x128 LDS-DMA) walk K in the same order and accumulate the
    same way, so the value of an output element does not depend on which one the autotuner picks for a shape —
    which is what makes "all hypotheses as one batch" give exactly the per-hypothesis values."""
    if engine_precision == "f32":
        pytest.skip("pre-split operands exist in the f16x3 / f16 modes only")
    from picopose_amd import ops

    g = torch.Generator().manual_seed(9)
    x, w, b = torch.randn(700, 768, generator=g).cuda(), (torch.randn(384, 768, generator=g) / 27).cuda(), torch.randn(384, generator=g).cuda()
    xi = torch.randn(2, 24, 24, 64, generator=g).cuda()
    wc = ops.pack_conv_weight((torch.randn(256, 64, 3, 3, generator=g) / 24).cuda())
    x2, w2 = torch.randn(1300, 392, generator=g).cuda(), (torch.randn(200, 392, generator=g) / 20).cuda()   # K % 16 == 8, row / column tails
    outs = []
    for cfg in ("0", "2", "3", "4", "5", "6", "7", "8"):
        monkeypatch.setenv("PP_GEMM_FORCE_CFG", cfg)
        outs.append((ops.linear(x, w, b, act="gelu"), ops.conv2d(xi, wc, None, 3, pad=1, act="relu"), ops.linear(x2, w2, None, residual=None)))
    for lin, conv, lin2 in outs[1:]:
        assert torch.equal(lin, outs[0][0])
        assert torch.equal(conv, outs[0][1])
        assert torch.equal(lin2, outs[0][2])
    if engine_precision == "f16x3":
        assert torch.allclose(outs[0][2].cpu(), x2.cpu() @ w2.cpu().t(), atol=2e-5, rtol=1e-5)
    else:
        _close(outs[0][2], x2.cpu() @ w2.cpu().t())


@gpu
@pytest.mark.parametrize("k,nout,hw,B", [(3, 2, 64, 3), (1, 1, 64, 2), (3, 2, 32, 5), (1, 1, 16, 4), (3, 1, 16, 2), (1, 2, 32, 1)])
def test_narrow_output_convolution(monkeypatch, k, nout, hw, B, engine_precision):
    """The one- / two-channel predict layers on an operand input (pp_conv_narrow_hl) against torch and against the GEMM
    route they replace (PP_CONV_NARROW=0), with bias and a residual, image borders and every band of rows."""
    from picopose_amd import ops

    g = torch.Generator().manual_seed(k * 100 + hw + nout)
    C = 256
    x = torch.randn(B, hw, hw, C, generator=g)
    w = torch.randn(nout, C, k, k, generator=g) / (C * k * k) ** 0.5
    b, res = torch.randn(nout, generator=g), torch.randn(B, hw, hw, nout, generator=g)
    # f32 mode: the fp32 map itself (pp_conv_narrow_f32); f16x3 / f16: the operand its producer wrote (pp_conv_narrow_hl)
    xs = x.cuda() if engine_precision == "f32" else ops.split_image(x.cuda())
    wp = ops.pack_conv_weight(w.cuda())
    got = ops.conv2d(xs, wp, b.cuda(), k, pad=k // 2, residual=res.cuda())
    monkeypatch.setenv("PP_CONV_NARROW", "0")
    gemm = ops.conv2d(xs, wp, b.cuda(), k, pad=k // 2, residual=res.cuda())
    monkeypatch.delenv("PP_CONV_NARROW")
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=k // 2).permute(0, 2, 3, 1) + res.double()
    assert got.shape == (B, hw, hw, nout)
    _close(got, ref.float(), 1e-5)
    _close(gemm, ref.float(), 1e-5)
    assert float((got - gemm).abs().max()) <= 2e-6 * float(ref.abs().max())


@gpu
def test_gelu_epilogue_accuracy():
    """The GELU of the GEMM epilogue (rational erf) against torch's erf GELU: <= 2e-6 absolute on [-10, 10]."""
    from picopose_amd import ops

    x = torch.linspace(-10, 10, 4096 * 8).reshape(4096, 8)
    eye = torch.eye(8)
    got = ops.linear(x.cuda(), eye.cuda(), None, act="gelu").cpu()
    ref = F.gelu(x.double()).float()
    assert (got - ref).abs().max().item() <= 2e-6 + 2e-7 * 10   # + the f16x3 operand rounding of x (2^-22 relative)


@gpu
def test_resize_with_both_outputs_equals_the_two_single_output_passes(engine_precision):
    """pp_resize_bilinear_nhwc_dual: the fp32 map and the operand of pp_resize_bilinear_nhwc_t from one pass."""
    from picopose_amd import ops

    if engine_precision == "f32":
        pytest.skip("no operand form in strict-fp32 mode")
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(3, 16, 16, 64, device="cuda", generator=g)
    for (ho, wo) in ((32, 32), (20, 36)):
        both = ops.resize_bilinear(x, ho, wo, also_split=True)
        # (the fp32-only kernel branch contracts the blend's multiply-adds differently: the last bit may differ from it)
        assert (both - ops.resize_bilinear(x, ho, wo)).abs().max().item() <= 2.0 ** -22 * x.abs().max().item()
        assert torch.equal(both._hl.hl, ops.resize_bilinear(x, ho, wo, out_split=True).hl) and both._hl.image == (3, ho, wo)
        # the two outputs are the same values: the operand is the split of the fp32 map
        assert torch.equal(both._hl.hl, ops.split_activation(both, 1, 3 * ho * wo, 64, 0, 64))


@gpu
@pytest.mark.parametrize("B,T,heads", [(1, 1, 1), (1, 3, 2), (2, 33, 2), (2, 96, 3), (3, 100, 6), (2, 257, 12), (1, 300, 3), (1, 1025, 2)])
def test_attention_shapes_and_operand_input(B, T, heads, engine_precision):
    """Fused attention for ragged sequence lengths (query tiles / key chunks with tails), from the fp32 qkv tensor
    and — f16x3 engine — from the hl operand the qkv GEMM epilogue writes (same bits)."""
    from picopose_amd import ops

    hd = 64
    g = torch.Generator().manual_seed(B * 1000 + T)
    qkv = torch.randn(B * T, 3 * heads * hd, generator=g)
    q, k, v = qkv.view(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
    ref = F.scaled_dot_product_attention(q.double(), k.double(), v.double()).permute(0, 2, 1, 3).reshape(B * T, heads * hd).float()
    got = ops.attention(qkv.cuda(), B, T, heads, hd)
    _close(got, ref, 2e-5)
    if engine_precision != "f32":
        sp = ops.Split(ops.split_activation(qkv.cuda(), 1, B * T, 3 * heads * hd, 0, 3 * heads * hd))
        got_sp = ops.attention(sp, B, T, heads, hd)
        if engine_precision == "f16x3":
            assert torch.equal(got_sp, got)
        else:   # h operands: q, k, v and the probabilities carry 11 bits
            _close(got_sp, ref)
        out_sp = ops.attention(sp, B, T, heads, hd, out_split=True)
        assert torch.equal(out_sp.hl, ops.split_activation(got_sp, 1, B * T, heads * hd, 0, heads * hd))


@gpu
@pytest.mark.parametrize("B,T,heads", [(2, 257, 12), (3, 100, 6), (1, 300, 3), (2, 33, 2), (1, 1025, 2), (5, 1, 1)])
def test_attention_lds_dma_ring_equals_the_register_staged_kernel(monkeypatch, B, T, heads, engine_precision):
    """The fused attention on operand input stages its K / V chunks through an LDS-DMA ring of three (PP_ATTN_RING=3, default) or two
    stages; the register-staged kernel of rounds 2-4 (PP_ATTN_RING=0) computes the same products in the same order: bit-identical
    outputs in both operand formats, fp32 and operand outputs, for token counts with full chunks, a masked last chunk and VALU tail keys."""
    if engine_precision == "f32":
        pytest.skip("operand input exists in the f16x3 / f16 modes only")
    from picopose_amd import ops

    g = torch.Generator().manual_seed(31)
    hd = 64
    qkv = (torch.randn(B * T, 3 * heads * hd, generator=g) * 2).cuda()
    w = torch.eye(3 * heads * hd).cuda()
    sp = ops.linear(qkv, w, None, out_split=True)        # the operand the qkv GEMM epilogue writes
    assert isinstance(sp, ops.Split)
    outs = {}
    for ring in ("0", "3", "2", "res"):
        monkeypatch.setenv("PP_ATTN_RING", "2" if ring == "res" else ring)
        if ring == "res":        # T = 257 .. 260: the nine-wave variant with all of K / V resident in LDS (opt-in; other T: the ring)
            monkeypatch.setenv("PP_ATTN_RESIDENT", "1")
        o32 = ops.attention(sp, B, T, heads, hd)
        osp = ops.attention(sp, B, T, heads, hd, out_split=True)
        outs[ring] = (o32, osp.hl)
    monkeypatch.delenv("PP_ATTN_RESIDENT")
    for ring in ("3", "2", "res"):
        assert torch.equal(outs[ring][0], outs["0"][0]), ring
        assert torch.equal(outs[ring][1], outs["0"][1]), ring
    q, k, v = (qkv.cpu().view(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)[i] for i in range(3))
    ref = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v
    _close(outs["3"][0].view(B, T, heads, hd).permute(0, 2, 1, 3), ref, 5e-4 if engine_precision == "f16x3" else TOL_F16)


@gpu
def test_saturation_check_raises_instead_of_returning_clipped_operands():
    """The f16x3 operand format holds |x| < 16376; ops.CHECK_SATURATION turns a silent clip into an error."""
    from picopose_amd import _lib, ops

    if ops.PRECISION != "f16x3":
        pytest.skip("f16x3 engine only")
    x = torch.randn(64, 256, device="cuda")
    w = torch.randn(128, 256, device="cuda") * 0.05
    ops.CHECK_SATURATION = True
    try:
        ops.linear(x, w, out_split=True)                           # in range: fine
        x[3, 7] = 4.0e4
        with pytest.raises(_lib.PicoPoseHipError, match="saturated"):
            ops.linear(x, w)                                       # the pre-split of the activation clips
        big = torch.full((128, 256), 30.0, device="cuda")
        with pytest.raises(_lib.PicoPoseHipError, match="saturated"):
            ops.linear(torch.full((64, 256), 30.0, device="cuda"), big, out_split=True)   # 256 * 900 = 230400 in the epilogue
    finally:
        ops.CHECK_SATURATION = False


@gpu
@pytest.mark.parametrize("r,cin,cout,hw", [(4, 256, 256, 16), (2, 512, 512, 16), (2, 64, 96, 12)])
def test_conv_transpose_operand_output_equals_split_of_the_fp32_result(r, cin, cout, hw, engine_precision):
    """ConvTranspose2d(kernel = stride = r) whose pixel-shuffle epilogue writes the NEXT convolution's operand directly
    (no fp32 map): bit-identical to splitting the fp32 result, and correct against torch."""
    if engine_precision == "f32":
        pytest.skip("operand outputs exist in f16x3 mode only")
    from picopose_amd import ops

    g = torch.Generator().manual_seed(r * 100 + cin)
    x = torch.randn(3, cin, hw, hw, generator=g)
    w, b = torch.randn(cin, cout, r, r, generator=g) / cin ** 0.5, torch.randn(cout, generator=g)
    wp, bp = ops.pack_convT_weight(w.cuda(), b.cuda())
    xn = ops.to_nhwc(x.cuda())
    full = ops.conv_transpose2d(xn, wp, bp, r)
    _close(ops.to_nchw(full), F.conv_transpose2d(x, w, b, stride=r))
    sp = ops.conv_transpose2d(xn, wp, bp, r, out_split=True)
    assert isinstance(sp, ops.Split) and sp.image == (3, hw * r, hw * r)
    want = ops.split_image(full)
    assert torch.equal(sp.hl, want.hl)
