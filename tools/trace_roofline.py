#!/usr/bin/env python3
"""Rewrites a bench.py JSON line that was produced UNDER rocprofv3 (`"under_profiler": true`) so that its roofline
fractions follow from the kernel trace written by that same run (HIP-event intervals are inflated by the profiler):
    python3 tools/trace_roofline.py <bench.json> <kernel_stats.csv>
adds roofline.frac_from_kernel_trace / ms_per_launch_kernel_trace (and the same for roofline_policy_linear) and sets
roofline.frac to the trace-derived value, keeping the event-derived one as frac_from_events."""
import csv
import json
import sys


def avg_us(stats, prefix):
    """average duration (us) of the kernels whose demangled name starts with `prefix` (weighted by calls)"""
    tot = calls = 0.0
    for r in stats:
        name = r["Name"].replace("void ", "").replace("fpc::", "")
        if name.startswith(prefix + "<") or name.startswith(prefix + "("):
            tot += float(r["TotalDurationNs"]); calls += float(r["Calls"])
    return tot / calls / 1e3 if calls else None


def main():
    bench, stats_csv = sys.argv[1:3]
    line = [l for l in open(bench).read().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    stats = list(csv.DictReader(open(stats_csv)))
    kern = d["roofline"]["kernel"].split(" ")[0]
    us = avg_us(stats, kern)
    if us:
        ro = d["roofline"]
        ro["frac_from_events"] = ro["frac"]
        ro["ms_per_launch_kernel_trace"] = us / 1e3
        ro["achieved"] = ro["flops_per_launch"] / (us * 1e-6) / 1e12
        ro["frac"] = ro["frac_from_kernel_trace"] = ro["achieved"] / ro["peak"]
        ro.pop("frac_note", None)
    pl = d.get("roofline_policy_linear")
    if pl:
        parts = [avg_us(stats, k.strip()) for k in pl["kernel"].split(" (")[0].split("+")]
        if all(parts):
            us2 = sum(parts)
            pl["frac_from_events"] = pl["frac"]
            pl["ms_per_launch_kernel_trace"] = us2 / 1e3
            pl["achieved"] = pl["bytes_per_launch"] / (us2 * 1e-6) / 1e9
            pl["frac"] = pl["frac_from_kernel_trace"] = pl["achieved"] / pl["peak"]
    d["under_profiler"] = True
    open(bench, "w").write(json.dumps(d) + "\n")
    print("%s: %s %.1f us -> frac %.3f" % (bench, kern, us or 0.0, d["roofline"]["frac"]))


if __name__ == "__main__":
    main()
