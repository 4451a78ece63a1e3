#!/bin/bash
# usage (GPU box): bash tools/ab_env.sh "<ENV_A>" "<ENV_B>" [rounds] [bench args...] — alternate two ENVIRONMENT settings of the same build on ONE
# box (boxes differ by up to 8 %), bench.py with a shared pinned autotuner table; prints crops/s per run.
A="$1"; B="$2"; R=${3:-3}; shift 3
root=$GRAFT_REPO_ROOT
t=$root/gpurun_out/ab_env_tune.txt
rm -f $t
cd $root
env $A python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact-leg --tune-file $t "$@" > /dev/null 2>&1
for i in $(seq $R); do
  for cfg in "$A" "$B"; do
    v=$(env $cfg python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-exact-leg --tune-file $t "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f crops/s %.2f ms' % (d['value'], d['ms_per_step']))")
    echo "[$cfg] $v"
  done
done
