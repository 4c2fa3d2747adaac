"""Host-side pieces of the measurement path that need no GPU: the K-split plan the exporter assumes for k_fcw, the
profiler detection of bench.py, and tools/trace_roofline.py on the committed round-5 profile (a bench line produced under
rocprofv3 + the kernel stats of the same run)."""
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(REPO, "alphazero-4-player-chess_amd"), REPO]


def test_fcw_split_exists_for_every_board_size_on_a_256_cu_part():
    """weights.fcw_split mirrors csrc/fpc_nn.h plan_fcw: column groups x splits <= CUs, whole 64-deep stages per block,
    at least four of them; the largest such split wins"""
    import weights
    want = {8: 8, 9: 8, 10: 8, 11: 8, 12: 6, 13: 4, 14: 4}
    for R, sk in want.items():
        assert weights.fcw_split(R) == sk, R
        A = (8 * R + 8) * R * R
        groups, stages = (A + 383) // 384, ((A + 511) // 512 * 512) // 64
        assert groups * sk <= 256 and stages % sk == 0 and stages // sk >= 4
        assert weights.default_fc_layout(R) == 2
    assert weights.fcw_split(14, cus=64) == 1 and weights.fcw_split(14, cus=32) == 0      # 62 groups need 62 CUs


def test_bench_detects_the_profiler(monkeypatch):
    import bench
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_")) or k == "LD_PRELOAD":
            monkeypatch.delenv(k, raising=False)
    assert bench.under_profiler() is False
    monkeypatch.setenv("ROCPROFILER_SDK_TOOL_LIBRARIES", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert bench.under_profiler() is True


def test_trace_roofline_recomputes_the_fractions_from_the_kernel_trace(tmp_path):
    src = os.path.join(REPO, "profiles", "r05")
    line = [l for l in open(os.path.join(src, "default_bench.json")).read().splitlines() if l.startswith("{")][-1]
    d0 = json.loads(line)
    # start from the event-derived numbers, as bench.py printed them
    d0["roofline"]["frac"] = d0["roofline"].get("frac_from_events", d0["roofline"]["frac"])
    d0["roofline_policy_linear"]["frac"] = d0["roofline_policy_linear"].get("frac_from_events", d0["roofline_policy_linear"]["frac"])
    for k in ("frac_from_kernel_trace", "frac_from_events", "ms_per_launch_kernel_trace"):
        d0["roofline"].pop(k, None); d0["roofline_policy_linear"].pop(k, None)
    bench_json = tmp_path / "b.json"
    bench_json.write_text(json.dumps(d0) + "\n")
    stats = tmp_path / "s.csv"
    shutil.copy(os.path.join(src, "default_kernel_stats.csv"), stats)
    subprocess.check_call([sys.executable, os.path.join(REPO, "tools", "trace_roofline.py"), str(bench_json), str(stats)])
    d = json.loads(bench_json.read_text())
    ro, pl = d["roofline"], d["roofline_policy_linear"]
    assert d["under_profiler"] is True
    assert abs(ro["ms_per_launch_kernel_trace"] - 0.2682) < 5e-4                         # k_towerc's trace average of that run
    assert abs(ro["frac"] - ro["flops_per_launch"] / (ro["ms_per_launch_kernel_trace"] * 1e-3) / 1e12 / ro["peak"]) < 1e-9
    assert ro["frac"] == ro["frac_from_kernel_trace"] and "frac_from_events" in ro
    assert abs(pl["ms_per_launch_kernel_trace"] - (0.2275 + 0.0188)) < 1e-3               # k_fcw + k_fc_reduce
    assert abs(pl["frac"] - pl["bytes_per_launch"] / (pl["ms_per_launch_kernel_trace"] * 1e-3) / 1e9 / pl["peak"]) < 1e-9
