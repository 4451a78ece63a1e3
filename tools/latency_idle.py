"""GPU-idle share of ONE test image from a rocprofv3 kernel trace of tools/latency_image.py.  A chunk of detections has exactly one stage-1
launch (s1_main), so in steady state the kernels between stage-1 launch k and launch k + (chunks per image) are one image's work
(the harness's generation of the next image's synthetic inputs — a handful of torch RNG kernels — falls inside; the reference's loader
does that in worker processes).  Reports span, the union of kernel-busy time, dispatches and the largest gaps for the last traced images.
usage: latency_idle.py <kernel_trace.csv> <chunks per image>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cpi = int(sys.argv[2]) if len(sys.argv) > 2 else 2
marks = [i for i, r in enumerate(rows) if "s1_main" in r["Kernel_Name"]]
n_img = (len(marks) - 1) // cpi
for im in range(max(0, n_img - 3), n_img):
    lo, hi = marks[im * cpi], marks[(im + 1) * cpi]
    seg = rows[lo:hi]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[hi]["Start_Timestamp"])
    busy, cur_s, cur_e, gaps = 0, None, None, []
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
                gaps.append((s - cur_e, r["Kernel_Name"][:48]))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps.append((t1 - cur_e, "(next image's first stage-1 launch)"))
    span = t1 - t0
    gaps.sort(reverse=True)
    print(f"image {im}: span {span / 1e6:.3f} ms, kernel-busy {busy / 1e6:.3f} ms, idle share {1 - busy / span:.3f}, {len(seg)} dispatches "
          f"({len(seg) // cpi} per chunk); gaps > 100 us: {sum(1 for g, _ in gaps if g > 100_000)}, their sum {sum(g for g, _ in gaps if g > 100_000) / 1e6:.3f} ms; "
          f"largest (us): {[(round(g / 1e3, 1), n) for g, n in gaps[:4]]}")
