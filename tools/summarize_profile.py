#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_round.sh into profiles/<tag>_*.csv/json (small, committed)."""
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
stats = glob.glob(os.path.join(out, "stats", "*", "*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(root, "profiles", f"{tag}_bench_bf16_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for r in rows[:40]:
        w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]])
traffic = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
for kind in ("fetch", "write"):
    fs = glob.glob(os.path.join(out, kind, "*", "*counter_collection.csv"))
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:110]
        traffic[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if kind == "fetch":
            traffic[k]["launches"] += 1
res = {}
for k, d in traffic.items():
    n = max(d["launches"], 1)
    # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM): x2
    res[k] = {"launches": d["launches"], "fetch_bytes_per_launch_corrected": 2 * d["FETCH_SIZE"] * 1024 / n,
              "write_bytes_per_launch": d["WRITE_SIZE"] * 1024 / n}
top = dict(sorted(res.items(), key=lambda kv: -(kv[1]["fetch_bytes_per_launch_corrected"] + kv[1]["write_bytes_per_launch"]) * kv[1]["launches"])[:25])
top["_meta"] = {"commit": os.environ.get("FALNET_COMMIT", "unknown"),
                "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 25 --warmup 3 --no-cpu-baseline --no-roofline",
                "units": "bytes per launch; fetch = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB"}
json.dump(top, open(os.path.join(root, "profiles", f"{tag}_bench_bf16_hbm_traffic.json"), "w"), indent=1)
for r in rows[:12]:
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
for k, v in [kv for kv in top.items() if kv[0] != '_meta'][:8]:
    print(k[:70], {a: (round(b / 1e6, 1) if 'bytes' in a else b) for a, b in v.items()}, "MB")
