"""Per-step kernel breakdown from a rocprofv3 kernel trace of bench.py: the dispatches between two consecutive
stage-1 launches (s1_main) are exactly one step.  usage: step_breakdown.py <kernel_trace.csv> [step index, default 2]
(with --warmup 1 --steps 3 the s1_main launches 1..3 open the timed steps; later ones belong to the untimed
event / verification legs of bench.py)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "s1_main" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lo, hi = marks[k], marks[k + 1]
agg = collections.defaultdict(lambda: [0, 0])
for r in rows[lo:hi]:
    n = r["Kernel_Name"]
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n)[:80] if not n.startswith("_Z") else n[:80]
    a = agg[n]
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
busy = sum(a[1] for a in agg.values())
print(f"step span {span/1e6:.2f} ms, kernel time {busy/1e6:.2f} ms, {hi-lo} dispatches")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{n:80s} n={a[0]:5d} {a[1]/1e6:8.3f} ms {100*a[1]/span:5.1f}%")
