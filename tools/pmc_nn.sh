# PMC passes over the network forward only (tools/nn_only.py); separate runs per counter group.
#   PMC_TAG=c3_ FPC_NN_BLOCKS=20 FPC_NN_HIDDEN=256 bash tools/pmc_nn.sh    # configs[3]'s network -> gpurun_out/pmc_c3_<pass>
T=${PMC_TAG:-}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; rm -rf gpurun_out/pmc_$T$name; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$T$name -- python3 tools/nn_only.py 3 > gpurun_out/pmc_$T$name.log 2>&1 || echo "pass $name failed"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VMEM_RD
run c TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum
run d FETCH_SIZE
run e WRITE_SIZE
run f GRBM_GUI_ACTIVE GRBM_COUNT
python3 - "$T" <<'PY'
import csv, glob, collections, sys
T = sys.argv[1] if len(sys.argv) > 1 else ''
for name in "abcdef":
    fs = glob.glob('gpurun_out/pmc_%s%s/*/*counter_collection.csv' % (T, name))
    if not fs:
        print(name, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'][:28]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        if not any(t in k for t in ('k_fcw<', 'k_fc16<', 'k_tower', 'k_conv3x3', 'k_fc_reduce', 'k_value')):
            continue
        print(name, k, {c: sum(v) / len(v) for c, v in d.items()})
PY
