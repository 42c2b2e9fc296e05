#!/usr/bin/python3
"""Per-kernel launch statistics of a rocprofv3 --kernel-trace run, ONE ROW PER LAUNCH SHAPE.

`rocprofv3 --stats` averages every launch of a kernel name: the fused-medians calls launch the same kernel twice -- a
calibration on 256 columns and the main launch on all of them -- so the average of the two says nothing about either (round 5's
C4 row: 12 calls, avg 20.3 ms, min 0.32 ms).  This groups the trace by (kernel name, grid size, workgroup size) and, inside a
group, by duration class (launches more than 8x apart are different work: the scatter kernel's calibration has the same grid
as its main launch), so that every row's average can be read against the bench line's HIP-event time.
    python3 tools/kernel_stats_by_shape.py <dir with *kernel_trace.csv> <out.csv>"""
import collections
import csv
import glob
import math
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(src + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Kind", "KERNEL_DISPATCH") != "KERNEL_DISPATCH":
            continue
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        grid = "x".join(str(int(r[k])) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        wg = "x".join(str(int(r[k])) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
        rows.append((r["Kernel_Name"], grid, wg, int(r.get("LDS_Block_Size", 0) or 0), int(r.get("VGPR_Count", 0) or 0), dur))
groups = collections.defaultdict(list)
for name, grid, wg, lds, vgpr, dur in rows:
    groups[(name, grid, wg, lds, vgpr)].append(dur)
out = []
for (name, grid, wg, lds, vgpr), durs in groups.items():
    durs.sort()
    classes, cur = [], [durs[0]]
    for d in durs[1:]:
        if d > 8 * max(cur[0], 1):          # a different piece of work under the same name and grid
            classes.append(cur)
            cur = [d]
        else:
            cur.append(d)
    classes.append(cur)
    for ci, c in enumerate(classes):
        out.append({"Name": name, "Grid_Size": grid, "Workgroup_Size": wg, "LDS_Block_Size": lds, "VGPR_Count": vgpr,
                    "Duration_Class": f"{ci + 1}/{len(classes)}", "Calls": len(c), "TotalDurationNs": sum(c),
                    "AverageNs": round(sum(c) / len(c), 1), "MinNs": c[0], "MaxNs": c[-1],
                    "StdDevNs": round(math.sqrt(sum((x - sum(c) / len(c)) ** 2 for x in c) / len(c)), 1)})
out.sort(key=lambda r: -r["TotalDurationNs"])
with open(dst, "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(out[0].keys()) if out else ["Name"])
    w.writeheader()
    w.writerows(out)
print(f"{dst}: {len(out)} rows from {len(rows)} launches")
