#!/bin/bash
# FETCH_SIZE of the attention kernel under two builds of the library (PP_LIB_SUFFIX=_old vs the default)
cd /tmp && export TMPDIR=/tmp
for suf in _old ""; do
  rm -rf /tmp/pa$suf
  PP_LIB_SUFFIX=$suf timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pa$suf -o pa -- python3 $GRAFT_REPO_ROOT/tools/bench_attn.py > /tmp/pa$suf.log 2>&1 < /dev/null
  f=$(find /tmp/pa$suf -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$suf" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "attn_f16x3" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
v = [float(r["Counter_Value"]) for r in rows]
print(f"lib '{sys.argv[2]}': {len(v)} attention launches, FETCH_SIZE mean {sum(v) / len(v):.3e} (x 64 B x 2 correction = {sum(v) / len(v) * 128 / 1e9:.2f} GB)")
PY
  else echo "no counter file for '$suf'"; tail -3 /tmp/pa$suf.log; fi
done
