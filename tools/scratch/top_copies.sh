#!/bin/bash
# the longest torch copy / add / cat kernels of one training step (which autograd-side copies are worth removing)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt
PP_TRAIN_MARK=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python3 $GRAFT_REPO_ROOT/tools/bench_train_full.py 32 4 > /tmp/pt.log 2>&1 < /dev/null
f=$(find /tmp/pt -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] || { echo "no trace"; tail -3 /tmp/pt.log; exit 1; }
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "keypoint_visibility_kernel" in r["Kernel_Name"]]
seg = rows[marks[-2]:marks[-1]]
sel = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size", "?"))) for r in seg if "at::native" in r["Kernel_Name"]]
sel.sort(reverse=True)
print("torch kernels of one step:", len(sel), "launches,", sum(s[0] for s in sel) / 1e6, "ms")
for d, n, g in sel[:25]: print(f"{d / 1e3:8.1f} us  grid {g:>10}  {n}")
PY
