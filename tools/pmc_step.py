"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 1 --warmup 1` into HBM bytes per
kernel for ONE step (the dispatches between the stage-1 launches that open the timed step and the leg after it).  FETCH_SIZE is doubled (gfx950 reports
half the bytes of wide coalesced reads, MI355X_MICROARCH.md); both counters are in KiB."""
import collections
import csv
import glob
import json
import re
import sys

root = sys.argv[1]
out = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{root}/{cname}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == cname]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "s1_main" in r["Kernel_Name"]]
    lo, hi = marks[1], marks[2]   # --warmup 1 --steps 1: launch 1 opens the timed step, launch 2 the first untimed leg
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows[lo:hi]:
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        n = re.sub(r"\(.*", "", n)[:72] if not n.startswith("_Z") else n[:48]
        agg[n][0] += 1
        agg[n][1] += float(r["Counter_Value"]) * 1024 * (2 if cname == "FETCH_SIZE" else 1)
    out[cname] = {k: {"launches": v[0], "bytes": v[1]} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]}
    out[cname + "_total_bytes"] = sum(v[1] for v in agg.values())
gemm = [k for k in out["FETCH_SIZE"] if "pp_gemm_u" in k]   # the pre-split kernel, every instantiation
out["gemm_f16x3_hbm_bytes_per_step"] = sum(out["FETCH_SIZE"][k]["bytes"] for k in gemm) + sum(
    out["WRITE_SIZE"].get(k, {"bytes": 0})["bytes"] for k in gemm)
out["gemm_f16x3_launches_per_step"] = sum(out["FETCH_SIZE"][k]["launches"] for k in gemm)
out["note"] = ("rocprofv3 --pmc, one counter per pass, no tracing; one step of bench.py (default workload); FETCH_SIZE x2 "
               "(gfx950 correction), KiB -> bytes")
print(json.dumps(out, indent=1))
