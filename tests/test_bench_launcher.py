"""bench.py --gpus N without an external launcher: the parent starts N ranks itself, before any
torch / HIP call, with the torchrun environment; a failing rank takes the job down (exit != 0)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _run(n, extra_env=None, args=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--launch-dry-run", *args], env=env, capture_output=True, text=True, timeout=120)
    line = [x for x in p.stdout.splitlines() if x.startswith("{")][-1]
    return p.returncode, json.loads(line), p.stderr


def test_launcher_starts_n_ranks_with_the_torchrun_environment():
    rc, out, _ = _run(4)
    assert rc == 0 and out["launched"] == 4 and out["exit"] == 0
    ranks = out["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1, 2, 3] and [r["local_rank"] for r in ranks] == [0, 1, 2, 3]
    assert all(r["world"] == 4 for r in ranks)
    masters = {r["master"] for r in ranks}
    assert len(masters) == 1 and masters.pop().startswith("127.0.0.1:")
    assert all(r["ipc_legacy"] == "0" for r in ranks)          # dmabuf IPC only on this pool (RCCL needs it)


def test_every_rank_is_pinned_to_its_own_share_of_the_host_cores():
    """8 ranks on 16 host cores must not migrate over each other: the launcher hands every rank
    a disjoint, contiguous slice of its own affinity mask, and the rank applies it before importing torch."""
    avail = sorted(os.sched_getaffinity(0))
    n = min(4, len(avail))
    rc, out, _ = _run(n)
    assert rc == 0
    shares = [r["cpus"] for r in out["ranks"]]
    assert all(len(s) >= 1 for s in shares)
    flat = [c for s in shares for c in s]
    assert len(flat) == len(set(flat)), shares                  # disjoint
    assert sorted(flat) == avail                                # together: exactly what the job may use
    assert all(s == sorted(s) and s == avail[avail.index(s[0]):avail.index(s[0]) + len(s)] for s in shares)   # contiguous
    assert max(len(s) for s in shares) - min(len(s) for s in shares) <= 1


def test_cpu_share_with_fewer_cores_than_ranks_and_under_an_external_launcher():
    sys.path.insert(0, REPO)
    import bench
    assert [bench.cpu_share(r, 8, cpus=range(16)) for r in range(8)] == [[2 * r, 2 * r + 1] for r in range(8)]
    assert [bench.cpu_share(r, 4, cpus=[3, 5]) for r in range(4)] == [[3], [5], [3], [5]]
    # torch.distributed.run form: no FPC_BENCH_CPUS, the rank derives the same slice from LOCAL_RANK / LOCAL_WORLD_SIZE
    avail = sorted(os.sched_getaffinity(0))
    if len(avail) >= 2:
        env = {k: v for k, v in os.environ.items() if k != "FPC_BENCH_CPUS"}
        env.update(RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
        p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-dry-run"], env=env, capture_output=True, text=True, timeout=60)
        probe = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][-1])["launch_probe"]
        assert probe["cpus"] == avail[len(avail) // 2:]


def test_a_failing_rank_stops_the_job_with_its_exit_code():
    rc, out, err = _run(2, {"FPC_BENCH_PROBE_EXIT_RANK1": "7"})
    assert rc == 7 and out["exit"] == 7 and "rank 1 failed" in err


def test_parent_makes_no_gpu_call_before_spawning():
    """the launcher path must not import torch (a HIP-initialised parent must not fork/exec ranks)"""
    code = ("import sys, runpy; sys.argv = ['bench.py', '--gpus', '2', '--launch-dry-run']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert e.code == 0, e.code\n"
            "assert 'torch' not in sys.modules, 'launcher imported torch'\n" % BENCH)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=120, capture_output=True)


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launch-dry-run"], env=env, capture_output=True, text=True, timeout=60)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_workload_labels_follow_the_arguments():
    sys.path.insert(0, REPO)
    import bench
    a = bench.parse_args([])
    assert (a.sims, a.blocks, a.hidden) == (400, 10, 128)
    assert bench.workload_label(a.games, a.sims, a.blocks, a.hidden, a.board).startswith("configs[1]:")
    r = bench.parse_args(["--config", "ref"])
    assert (r.sims, r.blocks, r.hidden, r.games, r.board) == (50, 15, 256, 100, 8)
    assert bench.workload_label(r.games, r.sims, r.blocks, r.hidden, r.board).startswith("reference default (alphazero.py:288-304")
    a = bench.parse_args(["--config", "3"])
    assert (a.sims, a.blocks, a.hidden, a.games, a.board) == (800, 20, 256, 256, 14)
    assert bench.workload_label(a.games, a.sims, a.blocks, a.hidden, a.board).startswith("configs[3]:")
    a = bench.parse_args(["--blocks", "20", "--hidden", "256", "--sims", "800"])
    assert bench.workload_label(a.games, a.sims, a.blocks, a.hidden, a.board).startswith("configs[3]:")
    assert bench.workload_label(256, 400, 10, 128, 8).startswith("custom")
