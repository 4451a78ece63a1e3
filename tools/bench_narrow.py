import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from picopose_amd import ops
g = torch.Generator().manual_seed(0)
for k, n, hw in ((3, 2, 64), (1, 1, 64), (3, 2, 32)):
    x = torch.randn(160, hw, hw, 256, generator=g).cuda()
    w = ops.pack_conv_weight((torch.randn(n, 256, k, k, generator=g) / 48).cuda())
    res = torch.randn(160, hw, hw, n, generator=g).cuda()
    xs = ops.split_image(x)
    for env in ("1", "0"):
        os.environ["PP_CONV_NARROW"] = env
        for _ in range(3): ops.conv2d(xs, w, None, k, pad=k // 2, residual=res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv2d(xs, w, None, k, pad=k // 2, residual=res)
        e1.record(); torch.cuda.synchronize()
        print(f"k={k} n={n} hw={hw} narrow={env}: {e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
