"""Join the passes of tools/profile_set.sh into ONE per-kernel table of one bench step: launches and ms (kernel trace), MFMA-busy
(SQ_VALU_MFMA_BUSY_CYCLES over the SIMD cycles of GRBM_GUI_ACTIVE), HBM-side FETCH / WRITE bytes (FETCH_SIZE x 2: gfx950 counts 64 B per
128-B request of a wide read, MI355X_MICROARCH.md "HBM"; KiB -> bytes), and — for the contraction-engine kernels — the algorithmic flops
and bytes bench.py recorded for the same launches (roofline.per_kernel of bench.json, joined on `rocprof_key`).  Every pass ran with the
same pinned autotuner table, so the launch counts of the passes must agree: the script aborts if they do not.
usage: profile_set.py <set dir>  -> per_kernel.json in it, a text table on stdout."""
import collections
import csv
import glob
import json
import re
import sys

root = sys.argv[1]
TILE_CFG = {(128, 128): 0, (128, 64): 2, (256, 128): 4, (256, 256): 5}


def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n)[:96] if not n.startswith("_Z") else n[:64]


def key_of(n):
    m = re.search(r"pp_gemm_u_kernel<TileCfg<(\d+), (\d+),[^>]*>, (\d), (\d), (true|false)>", n)
    if m:
        return f"u{TILE_CFG[(int(m.group(1)), int(m.group(2)))]}:m{m.group(3)}"
    if "pp_gemm_uh_kernel" in n:
        return "u6:m1"
    m = re.search(r"pp_gemm_f_kernel<FTile<(\d+), (\d+),[^>]*>, (\d)>", n)
    if m:   # the fp32 engine (pp_gemm_f.hip): configurations 3 = 128x128, 4 = 256x128, 5 = 256x256, 6 = 128x64
        return f"f{ {(128, 128): 3, (256, 128): 4, (256, 256): 5, (128, 64): 6, (256, 192): 7}[(int(m.group(1)), int(m.group(2)))] }:m{m.group(3)}"
    m = re.search(r"gemm_f16x3_kernel<(\d), (\d), (true|false)>", n)
    if m:
        return f"gx:{m.group(1)}:{m.group(2)}"
    m = re.search(r"\bgemm_kernel<(true|false), (\d), (\d)>", n)
    if m:
        return f"gf:{m.group(2)}:{m.group(3)}"
    return None


def step_rows(rows, idkey):
    """rows of ONE step: --warmup 1 --steps 1, so the stage-1 launch #1 opens the timed step and #2 the first untimed leg."""
    rows.sort(key=idkey)
    marks = [i for i, r in enumerate(rows) if "s1_main" in r["Kernel_Name"]]
    return rows[marks[1]:marks[2]]


table = collections.defaultdict(lambda: collections.defaultdict(float))
trace = list(csv.DictReader(open(f"{root}/kernel_trace.csv")))
tr = step_rows(trace, lambda r: int(r["Start_Timestamp"]))
span = (int(tr[-1]["End_Timestamp"]) - int(tr[0]["Start_Timestamp"])) / 1e6
for r in tr:
    t = table[short(r["Kernel_Name"])]
    t["launches"] += 1
    t["ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    t["vgpr"] = max(t["vgpr"], float(r.get("VGPR_Count", 0) or 0) + float(r.get("Accum_VGPR_Count", 0) or 0))
for cdir, names in (("mfma", ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA")), ("FETCH_SIZE", ("FETCH_SIZE",)), ("WRITE_SIZE", ("WRITE_SIZE",))):
    rows = []
    for f in glob.glob(f"{root}/{cdir}/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    for cname in names:
        sel = step_rows([r for r in rows if r["Counter_Name"] == cname], lambda r: int(r["Dispatch_Id"]))
        cnt = collections.Counter()
        for r in sel:
            n = short(r["Kernel_Name"])
            table[n][cname] += float(r["Counter_Value"])
            cnt[n] += 1
        for n, c in cnt.items():
            if int(table[n]["launches"]) != c:
                sys.exit(f"pass {cdir}/{cname}: {n} ran {c} times, the trace pass saw {int(table[n]['launches'])} — the passes differ")
bench = json.loads(open(f"{root}/bench.json").read().strip().splitlines()[-1])
roof = bench.get("roofline", {})
pk = {}
for r in roof.get("per_kernel") or []:       # (conv / dense launches of one instantiation are one rocprof kernel: summed)
    e = pk.setdefault(r["rocprof_key"], {"algorithmic_flops": 0.0, "algorithmic_bytes": 0.0, "launches": 0})
    for k in e:
        e[k] += r[k]
out, used = [], collections.Counter()
for n, t in table.items():
    k = key_of(n)
    if k:
        used[k] += int(t["launches"])
for n, t in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
    act = t["GRBM_GUI_ACTIVE"] / 8 * 1024
    fetch, write = t["FETCH_SIZE"] * 1024 * 2, t["WRITE_SIZE"] * 1024
    row = {"kernel": n, "launches": int(t["launches"]), "ms": t["ms"], "share_of_step": t["ms"] / span, "registers": int(t["vgpr"]),
           "mfma_busy": t["SQ_VALU_MFMA_BUSY_CYCLES"] / act if act else None, "mfma_insts": t["SQ_INSTS_MFMA"],
           "fetch_bytes": fetch, "write_bytes": write, "hbm_side_gbs": (fetch + write) / (t["ms"] * 1e-3) / 1e9 if t["ms"] else None}
    k = key_of(n)
    # the VEC true / false instantiations of a tile share one bench key: the bench entry is attributed by launch share
    if k in pk and used[k]:
        share = t["launches"] / used[k]
        b = pk[k]
        row.update({"rocprof_key": k, "algorithmic_flops": b["algorithmic_flops"] * share, "algorithmic_bytes": b["algorithmic_bytes"] * share,
                    "useful_tflops": b["algorithmic_flops"] * share / (t["ms"] * 1e-3) / 1e12,
                    "traffic_ratio": (fetch + write) / (b["algorithmic_bytes"] * share) if b["algorithmic_bytes"] else None,
                    "bench_launches_of_key": b["launches"], "trace_launches_of_key": used[k]})
    out.append(row)
tot_act = sum(t["GRBM_GUI_ACTIVE"] for t in table.values())
tot_busy = sum(t["SQ_VALU_MFMA_BUSY_CYCLES"] for t in table.values())


def is_vit(n):   # the ViT path: dense (MODE 0) engine kernels, attention, LayerNorm, token assembly
    return (key_of(n) or "").endswith(":m0") or any(s in n for s in ("attn_", "layernorm_kernel", "assemble_tokens"))


va = sum(t["GRBM_GUI_ACTIVE"] for n, t in table.items() if is_vit(n))
vb = sum(t["SQ_VALU_MFMA_BUSY_CYCLES"] for n, t in table.items() if is_vit(n))
summary = {"config": bench["config"], "value_crops_per_s": bench["value"], "ms_per_step_bench": bench["ms_per_step"], "step_span_ms_trace": span,
           "kernel_ms_trace": sum(t["ms"] for t in table.values()), "dispatches": int(sum(t["launches"] for t in table.values())),
           "mfma_busy_whole_step": tot_busy / (tot_act / 8 * 1024), "mfma_busy_vit_path": vb / (va / 8 * 1024) if va else None,
           "vit_path_share_of_active_cycles": va / tot_act, "fetch_bytes_step": sum(r["fetch_bytes"] for r in out),
           "write_bytes_step": sum(r["write_bytes"] for r in out), "autotune": bench.get("autotune"),
           "engine_algorithmic_bytes_step": sum(r.get("algorithmic_bytes", 0.0) for r in out),
           "engine_hbm_side_bytes_step": sum(r["fetch_bytes"] + r["write_bytes"] for r in out if "rocprof_key" in r),
           "note": "one step of bench.py (dispatches between two stage-1 launches); every pass under the same pinned autotuner table; "
                   "FETCH_SIZE x2 (gfx950), KiB -> bytes; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); "
                   "FETCH / WRITE count the L2's fabric side: Infinity-Cache hits included (not pure HBM)"}
json.dump({"summary": summary, "kernels": out}, open(f"{root}/per_kernel.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
print(f"{'kernel':98s} {'n':>4s} {'ms':>8s} {'share':>6s} {'busy':>6s} {'TFLOP/s':>8s} {'fetch GB':>9s} {'write GB':>9s} {'alg GB':>8s} {'ratio':>6s}")
for r in out[:28]:
    f = lambda v, w, p: (f"{v:{w}.{p}f}" if v is not None else " " * w)   # noqa: E731
    print(f"{r['kernel']:98s} {r['launches']:4d} {r['ms']:8.3f} {r['share_of_step']:6.3f} {f(r['mfma_busy'], 6, 3)} {f(r.get('useful_tflops'), 8, 1)} "
          f"{r['fetch_bytes'] / 1e9:9.3f} {r['write_bytes'] / 1e9:9.3f} {f(r.get('algorithmic_bytes', None) and r['algorithmic_bytes'] / 1e9, 8, 3)} {f(r.get('traffic_ratio'), 6, 2)}")
