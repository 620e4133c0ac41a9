#!/bin/bash
# Clock and power the chip holds under a workload: tools/power_probe.sh OUT <command...>   (rocm-smi sampled every 0.2 s while the command runs)
OUT=$1; shift
"$@" > $OUT.cmd.log 2>&1 &
pid=$!
: > $OUT
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|mclk clock level|Temperature \(Sensor (edge|junction|memory)" | tr '\n' '|' >> $OUT
  echo >> $OUT
  sleep 0.2
done
wait $pid
tail -3 $OUT.cmd.log
python3 - "$OUT" <<'PY'
import re, sys
rows = [l for l in open(sys.argv[1]) if l.strip()]
P = [float(m.group(1)) for l in rows for m in [re.search(r"Power \(W\): ([\d.]+)", l)] if m]
S = [int(m.group(1)) for l in rows for m in [re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)] if m]
if P: print(f"power W: n={len(P)} median={sorted(P)[len(P)//2]:.0f} max={max(P):.0f}")
if S: print(f"sclk MHz: n={len(S)} median={sorted(S)[len(S)//2]} min={min(S)} max={max(S)}")
print(rows[len(rows)//2][:400] if rows else "no samples")
PY
