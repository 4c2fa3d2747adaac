#!/bin/bash
# samples rocm-smi power / clocks while the bench runs (is the network phase power-capped?)
#   bash tools/power_probe.sh [extra bench.py flags, e.g. --policy-head legal]
python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-alt-policy-head "$@" > gpurun_out/power_bench.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Package Power|sclk|junction" | sed -e 's/.*: //' | tr '\n' ' '; echo
  sleep 0.5
done > gpurun_out/power_samples.txt
wait $BP
python3 - <<'PY'
import re
rows = []
for l in open('gpurun_out/power_samples.txt'):
    m = re.search(r'([\d.]+) \((\d+)Mhz\) ([\d.]+)', l)
    if m and float(m.group(3)) > 600:
        rows.append((float(m.group(1)), int(m.group(2)), float(m.group(3))))
if rows:
    print("busy samples %d: junction %.0f C, sclk avg %.0f MHz (min %d, max %d), power avg %.0f W (max %.0f)" % (
        len(rows), sum(r[0] for r in rows) / len(rows), sum(r[1] for r in rows) / len(rows), min(r[1] for r in rows),
        max(r[1] for r in rows), sum(r[2] for r in rows) / len(rows), max(r[2] for r in rows)))
PY
tail -1 gpurun_out/power_bench.log | cut -c1-160
