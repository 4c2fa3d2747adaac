# One GPU call that produces everything profiles/ holds for a round (run from the repo root on the GPU box):
#   bash tools/profile_round.sh r03 A     and     bash tools/profile_round.sh r03 B     (two GPU calls of <= 20 min)
# 1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the summary the roofline numbers must agree with)
# 2. PMC passes over the network forward (tools/pmc_nn.sh; separate --pmc passes, no trace domains mixed in)
# 3. configs[3] (ResNet(20,256), 800 sims, fp16) and the 8x8 literal-snapshot size: bench lines + kernel stats
# 4. power / clock samples during a bench run
# 5. (round 5) the sustained-MFMA and policy-Linear memory-pattern micro-benchmarks
TAG=${1:-rXX}
PART=${2:-AB}      # A: default stats + PMC passes (both networks); B: configs[3] / 8x8 / arena / memory-side PMC / power / default run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
stats() {   # stats <name> <bench flags...>
  name=$1; shift
  rm -rf gpurun_out/prof_$name
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python3 bench.py "$@" > gpurun_out/$TAG/${name}_bench.json 2> gpurun_out/$TAG/${name}_bench.err
  cp gpurun_out/prof_$name/*/*kernel_stats.csv gpurun_out/$TAG/${name}_kernel_stats.csv 2>/dev/null || echo "no stats for $name"
  # the line was produced under the profiler: its roofline fractions from the kernel trace of this very run
  python3 tools/trace_roofline.py gpurun_out/$TAG/${name}_bench.json gpurun_out/$TAG/${name}_kernel_stats.csv
  echo "== $name"; head -8 gpurun_out/$TAG/${name}_kernel_stats.csv | cut -d, -f1-5; cut -c1-300 gpurun_out/$TAG/${name}_bench.json
}
if [[ $PART == *A* ]]; then
stats default --no-cpu-baseline --no-alt-policy-head --no-alt-dtype --no-dropin
bash tools/pmc_nn.sh > gpurun_out/$TAG/pmc_nn.log 2>&1; tail -20 gpurun_out/$TAG/pmc_nn.log | cut -c1-400
for p in a b c d e f; do cp gpurun_out/pmc_$p/*/*counter_collection.csv gpurun_out/$TAG/pmc_$p.csv 2>/dev/null; done
python3 tools/pmc_summary.py gpurun_out/$TAG r14_b10_h128_g256 > gpurun_out/$TAG/pmc_summary_c1.txt 2>&1
# the same passes over configs[3]'s network (k_towerw at hidden 256): bench.py prices a kernel only with its own counters
mkdir -p gpurun_out/$TAG/c3
PMC_TAG=c3_ FPC_NN_BLOCKS=20 FPC_NN_HIDDEN=256 bash tools/pmc_nn.sh > gpurun_out/$TAG/pmc_nn_c3.log 2>&1; tail -8 gpurun_out/$TAG/pmc_nn_c3.log | cut -c1-300
for p in a b c d e f; do cp gpurun_out/pmc_c3_$p/*/*counter_collection.csv gpurun_out/$TAG/c3/pmc_$p.csv 2>/dev/null; done
python3 tools/pmc_summary.py gpurun_out/$TAG/c3 r14_b20_h256_g256 > gpurun_out/$TAG/pmc_summary_c3.txt 2>&1
cp profiles/pmc_summary.json gpurun_out/$TAG/pmc_summary.json
fi
if [[ $PART == *B* ]]; then
stats config3 --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-policy-head --no-alt-dtype --no-dropin
stats board8 --board 8 --steps 6 --no-cpu-baseline --no-alt-policy-head --no-alt-dtype --no-dropin
# the reference's own shipped default: ResNet(15,256), 100 parallel games x 50 searches, 8x8 EIGHT_SIMPLE (alphazero.py:288-304)
stats refdefault --config ref --steps 8 --warmup 2 --no-cpu-baseline --no-alt-policy-head --no-alt-dtype --no-dropin
python3 tools/arena_bench.py > gpurun_out/$TAG/arena_1024.json 2> gpurun_out/$TAG/arena_1024.err; cut -c1-400 gpurun_out/$TAG/arena_1024.json
bash tools/pmc_fc_mem.sh > gpurun_out/$TAG/pmc_fc_mem.log 2>&1; tail -12 gpurun_out/$TAG/pmc_fc_mem.log | cut -c1-400
for p in a b c d e f; do cp gpurun_out/pmcm_$p/*/*counter_collection.csv gpurun_out/$TAG/pmcm_$p.csv 2>/dev/null; done
bash tools/power_probe.sh --no-alt-dtype > gpurun_out/$TAG/power_probe.log 2>&1; cp gpurun_out/power_samples.txt gpurun_out/$TAG/power_samples.txt; tail -3 gpurun_out/$TAG/power_probe.log | cut -c1-300
python3 bench.py > gpurun_out/$TAG/default_run.json 2> gpurun_out/$TAG/default_run.err; cut -c1-200 gpurun_out/$TAG/default_run.json
# round 5: the micro-benchmarks DESIGN.md 4.2 / 9 quote (build here if the binaries did not travel with the snapshot)
[ -x tools/micro/fc_stream ] && [ -x tools/micro/mfma_sustained ] || make -s -C tools/micro fc_stream mfma_sustained
timeout -k 10 120 tools/micro/mfma_sustained 1.0 > gpurun_out/$TAG/mfma_sustained.log 2>&1; cat gpurun_out/$TAG/mfma_sustained.log | cut -c1-200
timeout -k 10 180 tools/micro/fc_stream 300 0 > gpurun_out/$TAG/fc_stream_decomp_layout.log 2>&1; head -6 gpurun_out/$TAG/fc_stream_decomp_layout.log
timeout -k 10 180 tools/micro/fc_stream 300 4 > gpurun_out/$TAG/fc_stream_wide_tile.log 2>&1; head -12 gpurun_out/$TAG/fc_stream_wide_tile.log
# the N = 2 launcher path on the one GPU of this box (gloo): rank pinning, host_ms_per_ply, ranks_seen
python3 bench.py --gpus 2 --rehearse-one-gpu --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-alt-policy-head --no-dropin 2> gpurun_out/$TAG/bench_2rank_rehearsal.err | grep "^{" > gpurun_out/$TAG/bench_2rank_rehearsal.json; cut -c1-200 gpurun_out/$TAG/bench_2rank_rehearsal.json   # (gloo prints a connection banner on stdout)

fi
