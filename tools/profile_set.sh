#!/bin/bash
# usage (on the GPU box): bash tools/profile_set.sh <tag> [bench.py arguments, e.g. --mode fp16 --workload full_b64_n512_vitl]
# ONE reproducible measurement set of a bench configuration: the GEMM autotuner is run once and its table is
# pinned (bench.py --tune-file), then every pass below loads that table and therefore sees IDENTICAL launches:
#   0. plain bench line (tunes, writes the table)                      -> bench.json, tune.txt
#   1. rocprofv3 --kernel-trace --stats                                 -> kernel_stats.csv, step_breakdown.txt
#   2. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA (counters only)
#   3. rocprofv3 --pmc FETCH_SIZE   4. rocprofv3 --pmc WRITE_SIZE      (separate passes, counters only)
#   -> per_kernel.json / per_kernel.txt (tools/profile_set.py): per kernel launches, ms, MFMA-busy, FETCH / WRITE bytes, and for the
#      engine kernels the algorithmic flops and bytes of bench.json's roofline.per_kernel with traffic_ratio = (FETCH + WRITE) / algorithmic
# Everything lands under gpurun_out/set_<tag>/ ; copy what is to be judged into profiles/<round>/<tag>/.
tag=$1
shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/set_$tag
mkdir -p $out
args="--steps 1 --warmup 1 --no-cpu-baseline --no-exact-leg --no-latency-leg --no-train-leg --tune-file $out/tune.txt $@"
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact-leg --no-latency-leg --no-train-leg --tune-file $out/tune.txt "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
echo "0/4 bench + tune table: $(wc -l < $out/tune.txt) shapes"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py $args > $out/bench_trace.json 2> $out/bench_trace.err || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 $root/tools/step_breakdown.py $(find $out/trace -name "*kernel_trace.csv" | head -1) 1 > $out/step_breakdown.txt || exit 1
cp $(find $out/trace -name "*kernel_trace.csv" | head -1) $out/kernel_trace.csv
echo "1/4 trace done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d $out/mfma -- python3 $root/bench.py $args > $out/bench_mfma.json 2> $out/bench_mfma.err || exit 1
echo "2/4 mfma busy done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/$c -- python3 $root/bench.py $args > $out/bench_$c.json 2> $out/bench_$c.err || exit 1
  echo "pass $c done"
done
cd $root
python3 tools/profile_set.py $out > $out/per_kernel.txt || exit 1
rm -rf $out/trace $out/mfma $out/FETCH_SIZE $out/WRITE_SIZE $out/kernel_trace.csv
cat $out/per_kernel.txt
