#!/bin/bash
# usage (GPU box): bash tools/s1_small_trace.sh <tag> — kernel trace of BASELINE configs[1] (stage 1 only, 8 crops x 42 templates, ViT-S width):
# per-kernel durations and the gaps between the launches of one matching call; and the bench line with both workgroup shapes.
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/s1small_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $root/bench.py --workload stage1_b8_n42_c384 --steps 200 --warmup 20 --no-cpu-baseline > $out/bench_trace.json 2> $out/err.txt || exit 1
cd $root
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "s1_main" in n]
i0, i1 = idx[-50], idx[-10]           # 40 steady-state calls
seg = rows[i0 - 2:i1 - 2]             # a call = qnorm, qpack, main, resolve, topk
dur = collections.defaultdict(list)
for r in seg:
    n = r["Kernel_Name"].split("(")[0][-40:]
    dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(seg, seg[1:])]
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3 / 40
print(f"per call: {span:.1f} us; kernels (us, mean over 40 calls):")
for n, v in dur.items():
    print(f"  {n:42s} {sum(v) / len(v):7.2f}  x{len(v) // 40}")
print(f"  gaps between consecutive kernels: mean {sum(gaps) / len(gaps):.2f} us, sum per call {sum(gaps) / 40:.1f} us")
PY
for w in 8 4; do PP_S1_WAVES=$w python3 bench.py --workload stage1_b8_n42_c384 --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('PP_S1_WAVES=$w', round(d['value']), 'crops/s', round(d['ms_per_step']*1e3,1), 'us/step (events', round(d['ms_per_step_median_hip_events']*1e3,1), ') kernel', round(r['kernel_ms']*1e3,1), 'us frac', round(r['frac'],3))"; done
rm -rf $out/trace
