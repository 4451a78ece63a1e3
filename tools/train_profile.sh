#!/bin/bash
# usage (GPU box): bash tools/train_profile.sh [B=32] — rocprofv3 kernel stats of tools/bench_train_step.py grouped by category
B=${1:-32}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -- python3 $GRAFT_REPO_ROOT/tools/bench_train_step.py $B dinov2_vitb14 > /tmp/prof_train.log 2>&1
f=$(find /tmp/prof_train -name "*kernel_stats.csv" | head -1)
python3 - $f <<"PY"
import csv,sys,collections,re
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
cat=collections.Counter(); calls=collections.Counter()
def c(n):
    if "pp_gemm_u" in n: return "pre-split engine"
    if "gemm_f16x3_kernel" in n: return "on-the-fly f16x3 gemm"
    if "gemm_kernel" in n: return "fp32 gemm"
    if "at::native" in n or "at_cuda" in n: return "torch: "+re.sub(r".*native::","",n)[:60]
    return re.sub(r"\(.*","",n.replace("void ","").replace("(anonymous namespace)::",""))[:50]
for r in rows:
    k=c(r["Name"]); cat[k]+=float(r["TotalDurationNs"]); calls[k]+=int(r["Calls"])
print(f"kernel time {tot/1e6:.0f} ms over the 18 steps of the tool (6 per scope)")
for k,v in cat.most_common(24): print(f"{v/tot*100:6.2f}%  calls {calls[k]:6d}  {k}")
PY
