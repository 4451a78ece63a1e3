#!/bin/bash
# usage: tools/pmc_step.sh <tag>  — HBM traffic (FETCH_SIZE, WRITE_SIZE; separate passes, counters only) of ONE
# step of the default bench, per kernel.  Writes gpurun_out/pmc_step_<tag>.json.
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/pmc_step_${tag}/$c -- python $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-leg > $root/gpurun_out/pmc_step_${tag}_$c.log 2>&1
  echo "pass $c done"
done
cd $root
python3 tools/pmc_step.py gpurun_out/pmc_step_${tag} > gpurun_out/pmc_step_${tag}.json
cat gpurun_out/pmc_step_${tag}.json
