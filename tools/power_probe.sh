#!/bin/bash
# samples rocm-smi power / clocks while the bench runs (is the NN phase power-capped?)
python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline > gpurun_out/power_bench.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Package Power|sclk|junction" | sed -e 's/.*: //' | tr '\n' ' '; echo
  sleep 0.5
done > gpurun_out/power_samples.txt
wait $BP
sort gpurun_out/power_samples.txt | uniq -c | sort -k1nr | head -25
tail -1 gpurun_out/power_bench.log | cut -c1-160
