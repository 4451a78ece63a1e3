#!/bin/bash
# A/B on one box of BASELINE configs[1] (stage 1 only, 8 x 42, C = 384): round-3 launch structure vs round 4's
cd $GRAFT_REPO_ROOT
run() { env $1 python3 bench.py --workload stage1_b8_n42_c384 --steps 2000 --warmup 100 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('[$1]', round(d['value']), 'crops/s', round(d['ms_per_step']*1e3,1), 'us/step, kernel', round(r['kernel_ms']*1e3,1), 'us, frac', round(r['frac'],3))"; }
for i in 1 2 3; do
  run "PP_S1_WAVES=8 PP_S1_QPREP=0 PP_S1_TOPK_SMALL=0"
  run "PP_S1_WAVES=4 PP_S1_QPREP=0 PP_S1_TOPK_SMALL=0"
  run "PP_S1_QPREP=1 PP_S1_TOPK_SMALL=0"
  run "PP_S1_QPREP=0 PP_S1_TOPK_SMALL=1"
  run "PP_S1_QPREP=1 PP_S1_TOPK_SMALL=1"
done
