for i in 1 2; do
for L in tools/var/lib_b.so alphazero-4-player-chess_amd/csrc/libfpc_engine.so; do
FPC_ENGINE_LIB=$PWD/$L timeout -k 10 300 python3 bench.py --blocks 20 --hidden 256 --sims 800 --dtype fp16 --steps 2 --warmup 1 --no-cpu-baseline --no-alt-policy-head --no-alt-dtype 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); st = d['stage_ms_per_sim_step']
print('$L', '%.0f sims/s' % d['value'], ' '.join('%s=%.4f' % (k[:6], v) for k, v in st.items()))"
done
done
