#!/bin/bash
# Timing-only diagnostic: the tower kernel with parts taken out (-DTW_STRIP=<bits>, csrc/fpc_tower.h; results are wrong on
# purpose).  Build the variants first (the product build defines nothing):
#   for v in 0 1 2 4 8 16 3 6 22 23; do hipcc ... -DTW_STRIP=$v -x hip csrc/fpc_engine.cpp -o tools/tmp/lib_strip_$v.so; done
# then, in ONE gpurun call:  bash tools/tower_strip.sh 0 1 2 4 8 16 3 6 22 23
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/strip
for v in "$@"; do
  rm -rf gpurun_out/strip/p_$v
  FPC_ENGINE_LIB=$PWD/tools/tmp/lib_strip_$v.so timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/strip/p_$v -- python3 tools/nn_only.py 400 > gpurun_out/strip/p_$v.log 2>&1 || { echo "variant $v failed"; tail -3 gpurun_out/strip/p_$v.log; exit 1; }
  python3 - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
f = glob.glob('gpurun_out/strip/p_%s/*/*kernel_stats.csv' % v)[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if 'k_towerc' in n or 'k_fcw' in n:
        print('TW_STRIP=%-3s %-26s calls %s avg %.1f us' % (v, n.replace('void ', '').replace('fpc::', '')[:26], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
