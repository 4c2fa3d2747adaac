#!/usr/bin/env python3
"""Condenses the memory-side PMC passes of tools/pmc_fc_mem.sh (pmcm_?.csv in a profiles/rNN directory) into
pmc_memory_side.json: per kernel, counter averages per launch + the derived figures DESIGN.md quotes.
    python3 tools/pmc_mem_summary.py profiles/r03"""
import collections, csv, glob, json, os, sys
src = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "pmcm_?.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if any(t in k for t in ("k_fc", "k_tower")) and "prep" not in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in agg.items():
    c = {n: sum(v) / len(v) for n, v in d.items()}
    e = dict(c)
    g = c.get
    if g("TCP_TCC_READ_REQ_sum"):
        e["avg_tcp_to_tcc_read_latency_cycles"] = g("TCP_TCC_READ_REQ_LATENCY_sum", 0) / g("TCP_TCC_READ_REQ_sum")
    if g("TCC_EA0_RDREQ_sum"):
        e["avg_hbm_read_latency_tcc_cycles"] = g("TCC_EA0_RDREQ_LEVEL_sum", 0) / g("TCC_EA0_RDREQ_sum")
    if g("TCP_GATE_EN1_sum"):
        e["tcp_pending_stall_frac"] = g("TCP_PENDING_STALL_CYCLES_sum", 0) / g("TCP_GATE_EN1_sum")
    if g("TCC_CYCLE_sum"):
        e["tcc_busy_frac"] = g("TCC_BUSY_sum", 0) / g("TCC_CYCLE_sum")
        e["avg_hbm_reads_in_flight_per_tcc_channel"] = g("TCC_EA0_RDREQ_LEVEL_sum", 0) / g("TCC_CYCLE_sum")
    if g("TA_TA_BUSY_sum"):
        e["ta_addr_stalled_by_tc_frac"] = g("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 0) / g("TA_TA_BUSY_sum")
        e["ta_data_stalled_by_tc_frac"] = g("TA_DATA_STALLED_BY_TC_CYCLES_sum", 0) / g("TA_TA_BUSY_sum")
    out[k] = e
json.dump(out, open(os.path.join(src, "pmc_memory_side.json"), "w"), indent=1)
for k, e in out.items():
    print(k, {x: round(v, 3) for x, v in e.items() if not x.isupper() and "_sum" not in x and "_avr" not in x})
