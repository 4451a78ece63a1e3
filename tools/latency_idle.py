"""GPU-idle share of ONE test image from a rocprofv3 kernel trace of tools/latency_image.py: an image is the kernels between two
consecutive GROUPS of stage-1 launches (a chunk of detections = one s1_main launch; an image = DETS / BS chunks).  Reports the span of the
last traced image (first kernel start -> last kernel end), the union of kernel-busy time inside it, dispatches, and the largest gaps.
usage: latency_idle.py <kernel_trace.csv> <chunks per image>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cpi = int(sys.argv[2]) if len(sys.argv) > 2 else 2
marks = [i for i, r in enumerate(rows) if "s1_main" in r["Kernel_Name"]]
# the images' kernels: from the first dispatch after the previous image's last one; the host builds the next image's inputs in between
# (torch RNG / fill kernels), so an image starts at the last "long gap" before its first s1_main
n_img = len(marks) // cpi
out = []
for im in range(max(0, n_img - 3), n_img):
    first_s1, last_s1 = marks[im * cpi], marks[im * cpi + cpi - 1]
    end = marks[(im + 1) * cpi] if (im + 1) * cpi < len(marks) else len(rows)
    # start: walk back from the image's first s1_main to the input-generation gap (> 200 us without a kernel)
    lo = first_s1
    while lo > 0 and int(rows[lo]["Start_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"]) < 200_000:
        lo -= 1
    # end: walk forward from the last s1_main to the next such gap
    hi = last_s1
    while hi + 1 < end and int(rows[hi + 1]["Start_Timestamp"]) - int(rows[hi]["End_Timestamp"]) < 200_000:
        hi += 1
    seg = rows[lo:hi + 1]
    t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
    busy, cur_s, cur_e, gaps = 0, None, None, []
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
                gaps.append((s - cur_e, r["Kernel_Name"][:60]))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = t1 - t0
    gaps.sort(reverse=True)
    out.append((span, busy, len(seg), gaps[:5]))
    print(f"image {im}: span {span / 1e6:.3f} ms, kernel-busy {busy / 1e6:.3f} ms, idle share {1 - busy / span:.3f}, {len(seg)} dispatches "
          f"({len(seg) // cpi} per chunk); largest gaps (us): {[(round(g / 1e3, 1), n) for g, n in gaps[:5]]}")
