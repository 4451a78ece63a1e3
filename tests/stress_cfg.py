import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from picopose_amd import ops
g = torch.Generator().manual_seed(1)
bad = 0
def run(cfg, fn):
    os.environ["PP_GEMM_FORCE_CFG"] = cfg
    return fn()
# dense shapes of a ViT-S net at small batch, conv shapes of the decoder / DPT at B = 6
dense = [(1542, 384, 1152), (1542, 384, 384), (1542, 384, 1536), (1542, 1536, 384), (4112, 384, 1152), (24576, 256, 256)]
for M, K, N in dense:
    x = torch.randn(M, K, generator=g).cuda(); w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda(); b = torch.randn(N, generator=g).cuda()
    xs = ops.Split(ops.split_activation(x, 1, M, K, 0, K))
    ref = run("0", lambda: ops.linear(xs, w, b, act="gelu"))
    for cfg in ("0", "2", "3", "7", "8", "4", "5"):
        for rep in range(int(os.environ.get('REPS', '60'))):
            out = run(cfg, lambda: ops.linear(xs, w, b, act="gelu"))
            if not torch.equal(out, ref):
                bad += 1; print("DENSE MISMATCH", M, K, N, "cfg", cfg, "rep", rep, int((out != ref).sum()), "entries, max", float((out - ref).abs().max()), flush=True); break
convs = [(6, 640, 512, 3, 1, 16), (6, 640, 512, 3, 1, 32), (6, 640, 512, 3, 1, 64), (6, 512, 256, 3, 1, 64), (6, 256, 256, 3, 1, 64), (6, 256, 192, 3, 1, 32),
         (6, 1024, 1024, 3, 2, 16), (6, 256, 256, 1, 1, 64), (6, 1024, 256, 3, 1, 8), (6, 512, 256, 3, 1, 32)]
for B, cin, cout, k, s, hw in convs:
    x = torch.randn(B, hw, hw, cin, generator=g).cuda()
    w = ops.pack_conv_weight((torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda())
    xs = ops.split_image(x)
    ref = run("0", lambda: ops.conv2d(xs, w, None, k, stride=s, pad=k // 2, act="relu"))
    for cfg in ("0", "2", "3", "7", "8", "4", "5", "6"):
        for rep in range(int(os.environ.get('REPS', '60'))):
            out = run(cfg, lambda: ops.conv2d(xs, w, None, k, stride=s, pad=k // 2, act="relu"))
            if not torch.equal(out, ref):
                bad += 1; print("CONV MISMATCH", B, cin, cout, k, s, hw, "cfg", cfg, "rep", rep, int((out != ref).sum()), "entries, max", float((out - ref).abs().max()), flush=True); break
print("mismatching (shape, cfg) pairs:", bad)
