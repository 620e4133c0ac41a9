#!/usr/bin/env python3
"""Summarise one training step from a rocprofv3 --kernel-trace CSV: per-kernel totals, span, idle gaps."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_tick_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
busy = 0
cur_end = t0
gaps = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_end:
        gaps += s - cur_end
    cur_end = max(cur_end, e)
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    n = r["Kernel_Name"].split("(")[0][:70]
    agg[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[n][1] += 1
print(f"step span {(t1-t0)/1e6:.3f} ms, {len(step)} dispatches, idle gaps {gaps/1e6:.3f} ms, sum of kernel time {sum(v[0] for v in agg.values())/1e6:.3f} ms")
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f"  {t/1e6:7.3f} ms  x{c:3d}  avg {t/c/1e3:7.1f} us  {n}")
if len(sys.argv) > 2:
    for i, r in enumerate(step):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if d > float(sys.argv[2]):
            print(i, f"{d:8.1f}us", r["Kernel_Name"][:75], "grid", r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
