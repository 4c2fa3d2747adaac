#!/bin/bash
# A/B of ONE library under different environment knobs in ONE gpurun call (same box, alternating), e.g.
#   gpurun -- 'bash tools/ab_env.sh 3 FPC_TOWER_WAVES=4 FPC_TOWER_WAVES=8'
export FPC_DEV_KNOBS=1      # the engine reads its developer knobs only with this set
N=$1; shift
for i in $(seq 1 $N); do
  for KV in "$@"; do
    env $KV python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-policy-head --no-alt-dtype --no-live-traffic ${BENCH_ARGS} 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); st = d['stage_ms_per_sim_step']
print('$KV', '%.0f sims/s' % d['value'], ' '.join('%s=%.4f' % (k[:6], v) for k, v in st.items()))" || exit 1
  done
done
