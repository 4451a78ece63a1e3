#!/bin/bash
# usage: tools/pmc_mfma_step.sh <tag> — MFMA pipe utilisation per kernel over the steps of the default bench
# (counters only, one group per run, no tracing); profiles/r01/pmc_mfma_step.txt is the same data cut to ONE step
# (the dispatches between the last two stage-1 launches).  busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs).
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d $root/gpurun_out/pmc_mfma_${tag} -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-leg > $root/gpurun_out/pmc_mfma_${tag}.log 2>&1
cd $root
python3 - "$root/gpurun_out/pmc_mfma_${tag}" <<'PY'
import csv, glob, sys, collections, re
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ids = sorted({int(r["Dispatch_Id"]) for r in rows if "s1_main" in r["Kernel_Name"]})
lo, hi = ids[1], ids[2]      # --warmup 1 --steps 1: the timed step = dispatches [s1_main #1, s1_main #2)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    if not (lo <= int(r["Dispatch_Id"]) < hi):
        continue
    n = r["Kernel_Name"]
    n = re.sub(r"\(.*", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))[:80]
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[n] += 1
print(f"ONE timed step of bench.py (default workload), dispatches {lo}..{hi - 1}")
print(f"{'kernel':82s} {'launches':>8s} {'MFMA busy':>10s} {'share of GPU-active cycles':>27s}")
tot = sum(v["GRBM_GUI_ACTIVE"] for v in agg.values())
tb = 0.0
vit = [0.0, 0.0]
for n, v in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
    tb += v["SQ_VALU_MFMA_BUSY_CYCLES"]
for n, v in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])[:18]:
    act = v["GRBM_GUI_ACTIVE"] / 8 * 1024
    busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / act if act else 0.0
    print(f"{n:82s} {cnt[n]:8d} {busy:10.3f} {v['GRBM_GUI_ACTIVE'] / tot:27.3f}")
def is_vit(n):   # the ViT path: dense (MODE 0) engine kernels, attention, LayerNorm, token assembly
    return (re.search(r"pp_gemm_u_kernel<TileCfg<[^>]*>, 0,", n) is not None) or any(k in n for k in ("attn_", "layernorm_kernel", "assemble_tokens"))
va = sum(v["GRBM_GUI_ACTIVE"] for n, v in agg.items() if is_vit(n))
vb = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for n, v in agg.items() if is_vit(n))
vg = sum(v["GRBM_GUI_ACTIVE"] for n, v in agg.items() if re.search(r"pp_gemm_u_kernel<TileCfg<[^>]*>, 0,", n))
vgb = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for n, v in agg.items() if re.search(r"pp_gemm_u_kernel<TileCfg<[^>]*>, 0,", n))
print(f"ViT path (dense engine kernels + attention + LayerNorm + token assembly): MFMA busy {vb / (va / 8 * 1024):.3f}, {va / tot:.3f} of the GPU-active cycles; its dense engine kernels alone {vgb / (vg / 8 * 1024):.3f}")
print(f"whole step: MFMA busy {tb / (tot / 8 * 1024):.3f} of the SIMD cycles while a kernel is active")
PY
