cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/prof_tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tmp -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt-policy-head > gpurun_out/prof_tmp.log 2>&1; echo exit=$?; python3 - <<'PY'
import csv, collections, glob
f=glob.glob('gpurun_out/prof_tmp/*/*_kernel_trace.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    key=(r['Kernel_Name'][:60], r['Grid_Size_X'], r['Grid_Size_Y'])
    agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000.0)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:12]:
    print(k, 'n=%d avg=%.1fus min=%.1f total=%.1fms'%(len(v), sum(v)/len(v), min(v), sum(v)/1000))
PY
