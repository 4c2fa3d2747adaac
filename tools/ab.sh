#!/bin/bash
# A/B engine builds in ONE gpurun call (same box, alternating), e.g.
#   gpurun -- 'bash tools/ab.sh 3 tools/var/lib_prev.so alphazero-4-player-chess_amd/csrc/libfpc_engine.so'
N=$1; shift
for i in $(seq 1 $N); do
  for L in "$@"; do
    FPC_ENGINE_LIB=$PWD/$L python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt-policy-head --no-alt-dtype --no-dropin --no-live-traffic ${BENCH_ARGS} 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); st = d['stage_ms_per_sim_step']
print('$L', '%.0f sims/s' % d['value'], ' '.join('%s=%.4f' % (k[:6], v) for k, v in st.items()))" || exit 1
  done
done
