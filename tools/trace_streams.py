#!/usr/bin/env python3
"""Per-stream accounting of one training step from a rocprofv3 kernel trace."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_tick_kernel" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
print(f"step span {(t1-t0)/1e6:.3f} ms, {len(step)} dispatches")
by = collections.defaultdict(list)
for r in step:
    by[(r["Queue_Id"], r.get("Stream_Id", ""))].append(r)
for k, rs in by.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    gaps, cur = 0, int(rs[0]["Start_Timestamp"])
    big = 0
    for r in rs:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > cur:
            gaps += s - cur
            if s - cur > 20000: big += 1
        cur = max(cur, e)
    print(f"queue/stream {k}: {len(rs)} kernels, busy {busy/1e6:.3f} ms, gaps {gaps/1e6:.3f} ms ({big} gaps > 20us), first {(int(rs[0]['Start_Timestamp'])-t0)/1e6:.3f} last {(int(rs[-1]['End_Timestamp'])-t0)/1e6:.3f}")
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        n = r["Kernel_Name"].split("(")[0][:64]
        agg[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[n][1] += 1
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"     {t/1e6:7.3f} ms x{c:3d}  {n}")
