#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o pc -- python3 $GRAFT_REPO_ROOT/tools/study_corr_bwd_levels.py > /tmp/pc.log 2>&1 < /dev/null
echo "rocprof rc=$?"
grep -v "simple_timer\|^$" /tmp/pc.log | tail -12
f=$(find /tmp/pc -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then python3 -c "
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:6]: print(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e6, 'ms avg')
" "$f"; else echo "no stats file"; find /tmp/pc | head; fi
