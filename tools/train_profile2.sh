#!/bin/bash
# usage (GPU box): bash tools/train_profile2.sh [B=32] — ONE full training step: wall / host-launch times without the profiler, then the kernel
# trace of one step (cut at the marker fills) grouped by category: kernel time, launch count.
B=${1:-32}
root=$GRAFT_REPO_ROOT
python3 $root/tools/bench_train_full.py $B 6 2>/dev/null | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_train2
PP_TRAIN_MARK=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_train2 -- python3 $root/tools/bench_train_full.py $B 5 > /tmp/prof_train2.log 2>&1
f=$(find /tmp/prof_train2 -name "*kernel_trace.csv" | head -1)
python3 - $f <<"PY"
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "keypoint_visibility_kernel" in r["Kernel_Name"]]   # once per step, at its start
lo, hi = marks[-2], marks[-1]
seg = rows[lo:hi]
def cat(n):
    if "pp_gemm_u" in n: return "pre-split engine"
    if "gemm_f16x3_kernel" in n: return "on-the-fly f16x3 gemm"
    if re.search(r"\bgemm_kernel<", n): return "fp32 gemm"
    if "at::native" in n or "at_cuda" in n: return "torch: " + re.sub(r".*native::", "", n)[:50]
    return re.sub(r"\(.*", "", n.replace("void ", "").replace("(anonymous namespace)::", ""))[:46]
t = collections.Counter(); c = collections.Counter()
for r in seg:
    k = cat(r["Kernel_Name"]); t[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); c[k] += 1
tot = sum(t.values()); span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
print(f"one step under the profiler: {len(seg)} launches, kernel time {tot / 1e6:.1f} ms, span {span / 1e6:.1f} ms")
for k, v in t.most_common(int(__import__("os").environ.get("PP_PROF_TOP", "26"))): print(f"{v / 1e6:8.2f} ms {v / tot * 100:5.1f}%  x{c[k]:5d}  {k}")
PY
