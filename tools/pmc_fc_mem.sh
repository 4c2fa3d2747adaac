# Memory-side PMC passes over the network forward (tools/nn_only.py): what holds the policy Linear's weight stream.
# Separate runs per counter group (no trace domains mixed in); summary printed per kernel.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; rm -rf gpurun_out/pmcm_$name; timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmcm_$name -- python3 tools/nn_only.py 3 > gpurun_out/pmcm_$name.log 2>&1 || echo "pass $name failed"; }
run a TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
run b TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
run c TCC_BUSY_sum TCC_CYCLE_sum TCC_REQ_sum TCC_READ_sum
# the TA block takes two counters per pass: four in one pass made rocprofv3 abort in round 2
# (rocprofiler_create_counter_config: "Request exceeds the capabilities of the hardware", gpurun_out/pmcm_d.log)
run d TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run f TA_TA_BUSY_sum TA_BUSY_avr
run e TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum
python3 - <<'PY'
import csv, glob, collections
for name in "abcdef":
    fs = glob.glob('gpurun_out/pmcm_%s/*/*counter_collection.csv' % name)
    if not fs:
        print(name, "no output"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'][:24]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        if not any(t in k for t in ('k_fcw<', 'k_fc16<', 'k_tower<', 'k_towerc<', 'k_fc_reduce')):
            continue
        print(name, k, {c: round(sum(v) / len(v), 1) for c, v in d.items()})
PY
