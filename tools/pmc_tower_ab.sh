#!/bin/bash
export FPC_DEV_KNOBS=1      # the engine reads its developer knobs only with this set
# SQ counters of the network forward under two settings of an environment knob (separate --pmc passes, kernel trace only)
#   gpurun -- 'bash tools/pmc_tower_ab.sh FPC_TOWER_WAVES 4 8'
KNOB=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
for V in "$@"; do
  export $KNOB=$V
  run() { name=$1; shift; rm -rf gpurun_out/r03/pmct_${V}_$name; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/r03/pmct_${V}_$name -- python3 tools/nn_only.py 3 > gpurun_out/r03/pmct_${V}_$name.log 2>&1 || echo "pass $name failed"; }
  run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
  run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VMEM_RD
  run f GRBM_GUI_ACTIVE GRBM_COUNT
done
python3 - "$@" <<'PY'
import csv, glob, collections, sys
for V in sys.argv[1:]:
    tot = {}
    for name in "abf":
        fs = glob.glob('gpurun_out/r03/pmct_%s_%s/*/*counter_collection.csv' % (V, name))
        if not fs:
            print(V, name, "no output"); continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if 'k_tower' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        for c, v in agg.items():
            tot[c] = sum(v) / len(v)
    print(V, {k: round(v) for k, v in sorted(tot.items())})
    if 'SQ_WAVE_CYCLES' in tot:
        w = tot['SQ_WAVE_CYCLES']
        print(V, "wait_any %.3f wait_inst %.3f active %.3f wait_lds %.3f | mfma_busy/grbm %.3f" % (
            tot['SQ_WAIT_ANY'] / w, tot['SQ_WAIT_INST_ANY'] / w, tot['SQ_ACTIVE_INST_ANY'] / w, tot['SQ_WAIT_INST_LDS'] / w,
            tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (tot.get('GRBM_GUI_ACTIVE', 1) / 8 * 1024)))
PY
