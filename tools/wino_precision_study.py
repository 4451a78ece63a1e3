"""CPU study (no GPU): rounding error of the Winograd forms of a 3x3 / stride 1 / pad 1 convolution on the f16x3 operand format
(22-bit operands hi + lo, fp32 accumulation) against a float64 direct convolution — what decided round 6's F(4x4, 3x3).

    python tools/wino_precision_study.py

Emulation: operands are rounded exactly as pp_split_f16 does (hi = f16(s x), lo = f16(s x - hi)); the three MFMA terms
hi hi + hi lo + lo hi are three fp32 matmuls (CPU sgemm: fp32 accumulation in blocked order).  Error = max |y - y64| / max |y64|."""
import torch

torch.manual_seed(0)


def split22(x, s):
    hi = (x * s).half()
    lo = (x * s - hi.float()).half()
    return hi.float(), lo.float()


def mm3(a, b, sa, sb):
    """(rows, K) x (N, K)^T on the f16x3 product form; a, b fp32."""
    ah, al = split22(a, sa)
    bh, bl = split22(b, sb)
    return (al @ bh.t() + ah @ bl.t() + ah @ bh.t()) / (sa * sb)


def pow2_scale(w, top=1024.0):
    m = w.abs().max().item()
    import math
    return 2.0 ** math.floor(math.log2(top / m))


# F(2x2, 3x3) and F(4x4, 3x3) matrices (Lavin & Gray; points 0, +-1, (+-2), inf)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]])
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]])
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1.]])
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1.]])
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1.]], dtype=torch.float64).float()
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1.]])


def wino(x, w, BT, G, AT, m, product):
    """x (B, H, W, C) fp32, w (Cout, Cin, 3, 3); product(U (rows, C), V (Cout, C)) -> (rows, Cout)."""
    B, H, W, C = x.shape
    a = m + 2
    xp = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1))
    th, tw = H // m, W // m
    tiles = torch.stack([torch.stack([xp[:, m * i:m * i + a, m * j:m * j + a, :] for j in range(tw)], 1) for i in range(th)], 1)  # B,th,tw,a,a,C
    U = torch.einsum("ai,nyxijc,bj->abnyxc", BT, tiles, BT)          # fp32 sums (a, a, B, th, tw, C)
    V = torch.einsum("ai,ocij,bj->aboc", G, w, G)                     # (a, a, Cout, Cin)
    Y = torch.empty(a, a, B * th * tw, w.shape[0])
    for i in range(a):
        for j in range(a):
            Y[i, j] = product(U[i, j].reshape(-1, C), V[i, j])
    Y = Y.reshape(a, a, B, th, tw, -1)
    out = torch.einsum("ia,abnyxo,jb->nyixjo", AT, Y, AT)             # (B, th, m, tw, m, Cout)
    return out.reshape(B, H, W, -1), U.abs().max().item()


def run(cin, cout, relu, hw=16, act_scale=1.0):
    x = torch.randn(2, hw, hw, cin) * act_scale
    if relu:
        x = x.clamp_min(0)
    w = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1)
    top = ref.abs().max().item()
    err = lambda y: (y.double() - ref).abs().max().item() / top  # noqa: E731
    # direct, f16x3: im2col rows against packed weights
    cols = torch.nn.functional.unfold(x.permute(0, 3, 1, 2), 3, padding=1).transpose(1, 2).reshape(-1, cin * 9)
    wf = w.reshape(cout, -1)
    d3 = mm3(cols, wf, 4.0, pow2_scale(wf)).reshape(ref.shape)
    d32 = (cols @ wf.t()).reshape(ref.shape)
    res = {"direct fp32": err(d32), "direct f16x3": err(d3)}
    for name, (BT, G, AT, m) in {"F(2x2)": (BT2, G2, AT2, 2), "F(4x4)": (BT4, G4, AT4, 4)}.items():
        y32, umax = wino(x, w, BT, G, AT, m, lambda U, V: U @ V.t())
        res[name + " fp32"] = err(y32)
        su = 4.0 if m == 2 else 0.25      # operand scale of U: |U| <= 4 |d| resp. 100 |d|

        def p3(U, V, su=su):
            return mm3(U, V, su, pow2_scale(V))
        y3, _ = wino(x, w, BT, G, AT, m, p3)
        res[name + " f16x3"] = err(y3)
        res[name + " max|U|/max|x|"] = umax / x.abs().max().item()
    return res


if __name__ == "__main__":
    for cin, cout, relu in ((64, 32, False), (256, 64, True), (640, 64, False), (640, 64, True)):
        r = run(cin, cout, relu)
        print(f"Cin {cin:4d} Cout {cout:3d} relu_in {int(relu)}: " + "  ".join(f"{k} {v:.2e}" for k, v in r.items()))
