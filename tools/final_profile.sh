#!/bin/bash
# usage (on the GPU box): bash tools/final_profile.sh <tag> — the judged profiles of a round, one after the other:
#   1. rocprofv3 --kernel-trace --stats of the default bench (3 steps) -> kernel stats csv + one-step breakdown
#   2. MFMA-busy per kernel of one step (counters only)           3. HBM traffic per kernel of one step (counters only)
#   4. the stage-1 kernel's FETCH_SIZE / WRITE_SIZE at the headline shape -> pmc_traffic_stage1.json
# Everything lands under gpurun_out/final_<tag>/ ; copy what is to be judged into profiles/<round>/.
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/final_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-leg > $out/bench_under_rocprofv3.json 2> $out/bench_under_rocprofv3.err || exit 1
cd $root
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/rocprofv3_kernel_stats_bench.csv
python3 tools/step_breakdown.py $(find $out/trace -name "*kernel_trace.csv" | head -1) > $out/step_breakdown.txt || exit 1
echo "1/4 trace done"
bash tools/pmc_mfma_step.sh $tag > $out/pmc_mfma_step.txt 2>&1 || exit 1
echo "2/4 mfma busy done"
bash tools/pmc_step.sh $tag > /dev/null 2>&1 || exit 1
cp gpurun_out/pmc_step_$tag.json $out/pmc_step.json
echo "3/4 traffic per kernel done"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/s1/$c -- python3 $root/bench.py --workload stage1_b32_n162_c768 --steps 5 --warmup 2 --no-cpu-baseline > $out/s1_$c.log 2>&1 || exit 1
done
cd $root
python3 - $out <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = []
    for f in glob.glob(f"{out}/s1/{c}/**/*counter_collection.csv", recursive=True):
        v += [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "s1_main" in r["Kernel_Name"]]
    res[c] = (sum(v) / len(v), len(v))
fetch = res["FETCH_SIZE"][0] * 1024 * 2
write = res["WRITE_SIZE"][0] * 1024
json.dump({"workload": "stage1_b32_n162_c768", "kernel": "s1_main<fast, 8 waves, fp32 bank>", "launches_averaged": res["FETCH_SIZE"][1],
           "fetch_bytes_corrected": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
           "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (tools/final_profile.sh); FETCH_SIZE x2 per the gfx950 "
                   "correction (MI355X_MICROARCH.md HBM section); KiB -> bytes; algorithmic bytes per launch 4.102e9"},
          open(f"{out}/pmc_traffic_stage1.json", "w"), indent=1)
print(open(f"{out}/pmc_traffic_stage1.json").read())
PY
echo "4/4 stage-1 traffic done"
