"""Producer -> consumer under two-process contention: a 1x1 convolution (GEMM epilogue writes fp32 + operand) followed at once by the
warp kernel reading its fp32 output; the warp result is compared bit for bit with a reference.  Start two of these at once.
History: with the warp kernel's taps behind lane-masked branches 2-5 % of the iterations differed (lanes 48-63 of single
waves lost taps) whatever the producer (CFG, PRODUCER=torch), less with SYNC=1; without those branches none do.  JUNK / SIDE /
RECHECK are the variants that were used to rule out stale cache lines, stream concurrency and read glitches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from picopose_amd import ops
g = torch.Generator().manual_seed(0)
Bf, B, H, C = 2, 6, int(os.environ.get("HW", "64")), 256
x = torch.randn(Bf, H, H, C, generator=g).cuda()
w = ops.pack_conv_weight((torch.randn(C, C, 1, 1, generator=g) / 16).cuda())
bias = torch.randn(C, generator=g).cuda()
flow = (torch.randn(B, H, H, 2, generator=g) * 3).cuda()
xs = ops.split_image(x)
sync = os.environ.get("SYNC") == "1"
cfg = os.environ.get("CFG")
if cfg:
    os.environ["PP_GEMM_FORCE_CFG"] = cfg
fq_const = None


def once():
    global fq_const
    jm = os.environ.get("JUNK", "randn")
    if jm == "randn":
        junk = torch.randn(Bf, H, H, C, device="cuda")        # the allocator hands this block to fq next: stale lines of other data
        del junk
    elif jm == "const":
        junk = torch.full((Bf, H, H, C), 1000.0, device="cuda")
        del junk
    if os.environ.get("PRODUCER") == "torch":          # a torch kernel writes fq (same values every time); the GEMM only runs as load
        ops.conv2d(xs, w, bias, 1, also_split="plain")
        fq = fq_const * 1.0
    else:
        fq = ops.conv2d(xs, w, bias, 1, also_split="plain")
    if sync:
        torch.cuda.synchronize()
    Xs = ops.Split.empty(B * H * H, 640, "cuda")
    ops.warp(fq, flow, hl_into=(Xs, 256))
    if os.environ.get("RECHECK") == "1":
        torch.cuda.synchronize()
        c1 = Xs.hl[:, 512:1024].clone(); torch.cuda.synchronize()
        c2 = Xs.hl[:, 512:1024].clone(); torch.cuda.synchronize()
        ops.warp(fq, flow, hl_into=(Xs, 256)); torch.cuda.synchronize()
        c3 = Xs.hl[:, 512:1024].clone(); torch.cuda.synchronize()
        if not torch.equal(c1, c2):
            print("   memory read twice after a sync differs (read glitch)", flush=True)
        elif ref_holder and not torch.equal(c1, ref_holder[0]):
            print("   memory stably holds wrong values; a re-run of warp after the sync gives", "the reference" if torch.equal(c3, ref_holder[0]) else "wrong values again", flush=True)
        return c1, fq.clone()
    return Xs.hl[:, 512:1024].clone(), fq.clone()
ref_holder = []
fq_const = ops.conv2d(xs, w, bias, 1)
torch.cuda.synchronize()
ref, fq_ref = once(); torch.cuda.synchronize(); ref, fq_ref = once(); torch.cuda.synchronize()
ref_holder.append(ref)
bad = 0
side = torch.cuda.Stream() if os.environ.get("SIDE") == "1" else None    # SIDE=1: a second stream of the SAME process keeps the GPU busy
if side is not None:
    la = torch.randn(8192, 256, device="cuda"); lw = torch.randn(256, 256, device="cuda") / 16
    las = ops.Split(ops.split_activation(la, 1, 8192, 256, 0, 256))
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    if side is not None:
        with torch.cuda.stream(side):
            for _ in range(3):
                ops.linear(las, lw, None)
    out, fq = once()
    if not torch.equal(out, ref):
        d = out != ref
        bad += 1
        if bad <= 5:
            print("rep", r, int(d.sum()), "halfs differ; rows", d.any(1).nonzero().flatten()[:5].tolist(), "cols", int(d.any(0).nonzero()[0]), "..", int(d.any(0).nonzero()[-1]),
                  "| fq itself equal:", bool(torch.equal(fq, fq_ref)), flush=True)
            r0 = int(d.any(1).nonzero()[0])
            print("   row", r0, "got ", out[r0, 384:400].float().tolist(), flush=True)
            print("   row", r0, "want", ref[r0, 384:400].float().tolist(), flush=True)
            same_elsewhere = (out[:, 384:512] == out[r0, 384:512]).all(1).nonzero().flatten().tolist()[:5]
            match_ref_row = (ref[:, 384:512] == out[r0, 384:512]).all(1).nonzero().flatten().tolist()[:5]
            print("   rows of OUT with the same 128 halfs:", same_elsewhere, " rows of REF equal to the wrong data:", match_ref_row, flush=True)
print("differing:", bad)
