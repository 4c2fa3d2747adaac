#!/usr/bin/env python3
"""Condenses the rocprofv3 --pmc passes of tools/pmc_nn.sh (gpurun_out/pmc_*) into
profiles/pmc_summary.json: per kernel, averages per launch.  HBM bytes follow the guide's gfx950
recipe: FETCH_SIZE (KB) counts wide coalesced reads at half -> doubled; WRITE_SIZE (KB) exact."""
import collections, csv, glob, json, os, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# usage: pmc_summary.py <dir holding pmc_a.csv .. pmc_f.csv> [shape key, e.g. r14_b20_h256_g256]
# (tools/profile_round.sh copies the CSVs there).  The summary is keyed by network shape so that
# bench.py never prices one kernel with another shape's counters.
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "gpurun_out", "r03")
shape = sys.argv[2] if len(sys.argv) > 2 else "r14_b10_h128_g256"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "pmc_?.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k.split("(")[0].split("::")[-1].split("<")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in agg.items():
    if not k.startswith("k_"):
        continue
    c = {n: sum(v) / len(v) for n, v in d.items()}
    e = {"counters_avg_per_launch": c}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_bytes"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        e["hbm_read_bytes_corrected"] = 2 * c["FETCH_SIZE"] * 1024
        e["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
    if "SQ_INSTS_MFMA" in c and "GRBM_GUI_ACTIVE" in c:
        e["mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "TCC_HIT_sum" in c:
        e["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
    out[k] = e
dst = os.path.join(HERE, "profiles", "pmc_summary.json")
try:
    allshapes = json.load(open(dst))
except Exception:
    allshapes = {}
allshapes[shape] = out
json.dump(allshapes, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {x: v for x, v in e.items() if x != "counters_avg_per_launch"} for k, e in out.items()}, indent=1))
