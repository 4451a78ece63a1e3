#!/bin/bash
# A/B of two settings on the default bench, alternating on ONE box (boxes differ by several %).  Each argument is a list
# of environment assignments (may be empty), e.g. two builds of the library or a switch of the host code:
#   ab_bench.sh "PP_LIB_SUFFIX=_base" "" 2        ab_bench.sh "PP_CORR_HL=0" "" 2
a=$1; b=$2; n=${3:-2}
for i in $(seq $n); do
  for s in "$a" "$b"; do
    env $s python bench.py --no-exact-leg --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$s]', round(d['value'],1), 'crops/s', round(d['ms_per_step'],2), 'ms')" || exit 1
  done
done
