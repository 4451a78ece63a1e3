"""Per-layer A/B of Winograd F(4x4, 3x3) against the direct implicit-GEMM convolution on the f16x3 engine (round 6):
the 3x3 layers of the flow decoder's heads and of the DPT head at configs[2]'s shapes (160 hypothesis-crops; 192 DPT images).

    python tools/bench_wino4.py [--reps 5]      -> one line per layer: direct ms, Winograd ms (input transform | products | output transform)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd import ops  # noqa: E402

LAYERS = [  # name, images, H = W, Cin, Cout
    ("head 640->512 @64", 160, 64, 640, 512), ("head 512->256 @64", 160, 64, 512, 256),
    ("head 640->512 @32", 160, 32, 640, 512), ("head 512->256 @32", 160, 32, 512, 256),
    ("head 640->512 @16", 160, 16, 640, 512), ("head 512->256 @16", 160, 16, 512, 256),
    ("corr1 256->192 @64", 160, 64, 256, 192), ("out0 256->128 @64", 160, 64, 256, 128),
    ("rcu 256->256 @32", 192, 32, 256, 256), ("rn2 512->256 @32", 192, 32, 512, 256), ("rn3 1024->256 @16", 192, 16, 1024, 256),
]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    print(f"{'layer':22s} {'direct ms':>10s} {'wino4 ms':>9s} {'in-tx':>7s} {'gemm':>7s} {'out-tx':>7s} {'TF/s direct':>11s} {'TF/s gemm':>9s}  max|diff|/max")
    for name, B, hw, cin, cout in LAYERS:
        x = torch.randn(B, hw, hw, cin, generator=g).cuda()
        xs = ops.split_image(x)
        del x
        wp = ops.pack_conv_weight((torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).cuda())
        bias = torch.randn(cout, generator=g).cuda()
        run = lambda: ops.conv2d(xs, wp, bias, 3, pad=1, act="relu", out_split=True, wino=True)  # noqa: E731
        ops.WINOGRAD4 = False
        d = run()
        t_d = timed(run, a.reps)
        ops.WINOGRAD4 = True
        w = run()
        t_w = timed(run, a.reps)
        diff = float((w.hl.float() - d.hl.float()).abs().max()) / float(d.hl.float().abs().max())
        # the three phases separately
        P = B * (hw // 4) ** 2
        t_in = timed(lambda: ops._winograd4_input(xs.hl.data_ptr(), cin, B, hw, hw, cin, xs.device), a.reps)
        sh = ops.winograd_shared(xs, cout=cout)
        t_shared = timed(lambda: ops.conv2d(sh, wp, bias, 3, pad=1, act="relu", out_split=True), a.reps)
        Y = torch.empty(36, P, cout, device="cuda")
        hl_t = ops.Split.empty(B * hw * hw, cout, xs.device)
        from picopose_amd import _lib
        t_out = timed(lambda: _lib.check(_lib.lib().pp_winograd4_output(Y.data_ptr(), cout, B, hw, hw, cout, bias.data_ptr(), 1, None, None, None, 0,
                                                                         hl_t.hl.data_ptr(), cout, 0, P, _lib.stream_ptr()), "out"), a.reps)
        t_g = t_shared - t_out
        fl = 2.0 * B * hw * hw * cout * 9 * cin
        if cout % 32 == 0 and hw in (16, 32, 64):   # the chained output -> input transform against the two separate kernels (next layer: cout -> cout)
            U1 = ops.Split(torch.empty(36 * P, 2 * cout, dtype=torch.float16, device="cuda"), 2)
            t_ch = timed(lambda: _lib.check(_lib.lib().pp_winograd4_chain(Y.data_ptr(), cout, B, hw, hw, cout, bias.data_ptr(), 1, 0, U1.hl.data_ptr(), P,
                                                                          _lib.stream_ptr()), "chain"), a.reps)
            t_in2 = timed(lambda: ops._winograd4_input(hl_t.hl.data_ptr(), cout, B, hw, hw, cout, xs.device), a.reps)
            name = f"{name} | out+in {t_out + t_in2:.3f} chained {t_ch:.3f}"
            del U1
        print(f"{name:22s} {t_d:10.3f} {t_w:9.3f} {t_in:7.3f} {t_g:7.3f} {t_out:7.3f} {fl / t_d / 1e9:11.1f} {fl / 4 / t_g / 1e9:9.1f}  {diff:.1e}", flush=True)
        del xs, sh, Y, hl_t, d, w
        torch.cuda.empty_cache()


if __name__ == "__main__":
    with torch.no_grad():
        main()
