#!/usr/bin/env python3
"""Time-ordered listing of one training step from a rocprofv3 kernel trace: for each kernel its queue, start offset, duration,
grid and the fraction of its span during which a kernel of ANOTHER queue was running (the overlap that the backward streams buy)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_tick_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = rows[a + 1:b + 1]
t0 = int(step[0]["Start_Timestamp"])
qs = sorted({r["Queue_Id"] for r in step}, key=lambda q: -sum(1 for r in step if r["Queue_Id"] == q))
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]) for r in step]
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ov = 0
    pts = sorted((max(s, s2), min(e, e2)) for s2, e2, q2 in iv if q2 != r["Queue_Id"] and s2 < e and e2 > s)
    cur = s
    for x, y in pts:
        if y > cur:
            ov += y - max(x, cur); cur = y
    n = r["Kernel_Name"].split("(")[0]
    n = n.replace("_kernel", "")[:58]
    print(f"q{qs.index(r['Queue_Id'])} {(s-t0)/1e3:8.1f} {(e-s)/1e3:7.1f}us ov{100*ov/max(1,e-s):4.0f}%  g{int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):5d}x{r['Grid_Size_Y']:>3}x{r['Grid_Size_Z']:>3} {n}")
