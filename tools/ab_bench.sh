#!/bin/bash
# A/B of two builds of the library on the default bench, alternating on ONE box (boxes differ by several %):
#   usage: ab_bench.sh <suffix A> <suffix B> [rounds]   ("" = the default library, e.g. ab_bench.sh _base "" 2)
a=$1; b=$2; n=${3:-2}
for i in $(seq $n); do
  for s in "$a" "$b"; do
    PP_LIB_SUFFIX=$s python bench.py --no-exact-leg --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib[$s]', round(d['value'],1), 'crops/s', round(d['ms_per_step'],2), 'ms')" || exit 1
  done
done
