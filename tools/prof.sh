#!/bin/bash
# usage: tools_prof.sh <tag> <cmd...>  -> prints per-kernel avg/min ns (rocprofv3 --kernel-trace --stats)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- "$@" > $out.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'at::native' in n or 'rocclr' in n: continue
    print(f"{n[:70]:70s} calls={r['Calls']:>4s} avg={float(r['AverageNs'])/1e3:9.1f}us min={float(r['MinNs'])/1e3:9.1f}us")
PY
