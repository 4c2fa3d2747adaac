# One GPU call that produces everything profiles/ holds for a round (run from the repo root on the GPU box):
#   bash tools/profile_round.sh r02
# 1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the summary the roofline numbers must agree with)
# 2. PMC passes over the network forward (tools/pmc_nn.sh; separate --pmc passes, no trace domains mixed in)
# 3. configs[3] (ResNet(20,256), 800 sims, fp16) and the 8x8 literal-snapshot size: bench lines + kernel stats
# 4. power / clock samples during a bench run
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
stats() {   # stats <name> <bench flags...>
  name=$1; shift
  rm -rf gpurun_out/prof_$name
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python3 bench.py "$@" > gpurun_out/$TAG/${name}_bench.json 2> gpurun_out/$TAG/${name}_bench.err
  cp gpurun_out/prof_$name/*/*kernel_stats.csv gpurun_out/$TAG/${name}_kernel_stats.csv 2>/dev/null || echo "no stats for $name"
  echo "== $name"; head -8 gpurun_out/$TAG/${name}_kernel_stats.csv | cut -d, -f1-5; cut -c1-300 gpurun_out/$TAG/${name}_bench.json
}
stats default --no-cpu-baseline --no-alt-policy-head --no-alt-dtype
bash tools/pmc_nn.sh > gpurun_out/$TAG/pmc_nn.log 2>&1; tail -20 gpurun_out/$TAG/pmc_nn.log | cut -c1-400
for p in a b c d e f; do cp gpurun_out/pmc_$p/*/*counter_collection.csv gpurun_out/$TAG/pmc_$p.csv 2>/dev/null; done
stats config3 --blocks 20 --hidden 256 --sims 800 --dtype fp16 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-policy-head --no-alt-dtype
stats board8 --board 8 --steps 6 --no-cpu-baseline --no-alt-policy-head --no-alt-dtype
bash tools/power_probe.sh --no-alt-dtype > gpurun_out/$TAG/power_probe.log 2>&1; cp gpurun_out/power_samples.txt gpurun_out/$TAG/power_samples.txt; tail -3 gpurun_out/$TAG/power_probe.log | cut -c1-300
python3 bench.py > gpurun_out/$TAG/default_run.json 2> gpurun_out/$TAG/default_run.err; cut -c1-200 gpurun_out/$TAG/default_run.json
