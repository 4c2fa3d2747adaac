#!/usr/bin/env python3
"""bench.py -- MCTS simulations/sec of the MI355X-native self-play engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one MCTS.search call over the whole batch of concurrent games (one ply of self-play:
`sims` simulations per game through select -> encode -> ResNet -> expand, all on the GPU), followed
by the move selection + TakeAction + GetGameResult that the reference's play() loop performs
(alphazero.py:99-144), so that successive steps see realistic positions.  Workload at every N:
BASELINE.json configs[1] per GPU -- 256 concurrent games x 400 sims/move, 10-block/128-filter
ResNet, 14x14 STANDARD start, random-init weights (torch.manual_seed(0)), synthetic data.
MFMA operand type: fp16 by default -- it is the 16-bit type that meets north_star's float bar (logits
within 1e-3 of the reference's fp32 net.py: 3.1e-4, tests/test_net_gpu.py) at the same MFMA rate;
configs[1]'s bf16 does not (2.3e-3) and is measured right after the timed region and reported beside
the headline as `alt_dtype_bf16` with both measured errors in `float_parity`.
N > 1: one process per GPU -- started by this script itself (`python bench.py --gpus N`: the parent makes no
torch / HIP call, spawns the N ranks with the torchrun environment and exits non-zero if any of them fails) or by
`python -m torch.distributed.run ... bench.py --gpus N` -- games sharded 256/GPU (weak scaling), no data-path
collective inside the search; training tuples are built on the device (fpc_collect_tuples) and the all-gather
that ends an episode is one ncclAllGather issued by the engine's C++ host over RCCL/xGMI (fpc_allgather_tuples),
executed once inside the timed region (the collective only; the records are parsed afterwards, at every N alike).
A communicator that cannot be set up at --backend nccl stops the job (no silent fallback).

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel group (the MFMA
implicit-GEMM network forward) with HIP events recorded on the engine's own stream during the
timed steps (every 16th simulation step carries the events; on every step the five of them cost 2.7 %); `cpu_baseline`
times the CPU oracle (oracle/, test infrastructure) + PyTorch-CPU ResNet on a bounded sample of the
same workload on this box's host cores (N = 1 only).  `value` is measured with the full
policy head unless --policy-head legal is given (the reference's softmax -> mask -> renormalise arithmetic, op for op); after the timed
region three more steps run with the opt-in legal-only policy head and are reported separately as
`alt_policy_head_legal_only` -- never as `value`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "alphazero-4-player-chess_amd"), HERE, os.path.join(HERE, "tests")]

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0}       # dense MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def nn_macs(R, blocks, F):
    """9*R*C*(24F + 2*Nb*F^2 + F*A_ch + 24F) + A^2 + 24*R*C"""
    A_ch = 8 * R + 8
    A = A_ch * R * R
    return 9 * R * R * (24 * F + 2 * blocks * F * F + F * A_ch + 24 * F) + A * A + 24 * R * R


def tree_bytes_per_sim(R, b=35, d=4):
    """SURVEY.md 8(d) algorithmic bytes per simulation on the tree side (bf16 activations)"""
    RR = R * R
    A = (8 * R + 8) * RR
    return 24 * RR * 2 + A * 4 + 4 + (1 + b) * (RR + 16) + 26 * b + d * b * 20


class Spec:
    def __init__(self, R):
        self.R = R
        self.num_state_channels = 24
        self.num_action_channels = 8 * R + 8
        self.action_space_size = self.num_action_channels * R * R
        self.state_space_size = 24 * R * R

    def nRows(self):
        return self.R

    def nCols(self):
        return self.R


def pick_moves(res, rng, temperature):
    """alphazero.py:104-119: pi ~ child visit counts, temperature, one multinomial draw per game (seeded here;
    all games at once: the per-game Python loop cost 2 ms of host time per ply)."""
    n = np.asarray(res["n_children"]).astype(np.int64)
    G = len(n)
    width = max(int(n.max()), 1)
    live = np.arange(width)[None, :] < n[:, None]
    p = np.where(live, res["visits"][:, :width].astype(np.float64), 0.0)
    tot = p.sum(axis=1, keepdims=True)
    p = np.divide(p, tot, out=np.zeros_like(p), where=tot > 0)
    p = np.where(live, p ** (1.0 / temperature), 0.0)
    tot = p.sum(axis=1, keepdims=True)
    p = np.divide(p, tot, out=np.zeros_like(p), where=tot > 0)
    idx = (np.cumsum(p, axis=1) < rng.random(G)[:, None]).sum(axis=1)
    idx = np.minimum(idx, np.maximum(n - 1, 0))
    flats = np.asarray(res["flat"])[np.arange(G), idx].astype(np.int64)
    flats[n == 0] = -1
    return flats


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=["1", "3", "ref"], default=None,
                    help="preset: 1 = BASELINE.json configs[1] (256 games x 400 sims, ResNet(10,128), the default), "
                         "3 = configs[3] (ResNet(20,256), 800 sims/move, fp16), "
                         "ref = the reference's own shipped default (alphazero.py:288-304: ResNet(15,256), 100 parallel games x "
                         "50 searches on its compiled 8x8 EIGHT_SIMPLE board)")
    ap.add_argument("--games", type=int, default=None, help="concurrent games per GPU (default 256; 100 with --config ref)")
    ap.add_argument("--sims", type=int, default=None)
    ap.add_argument("--blocks", type=int, default=None)
    ap.add_argument("--hidden", type=int, default=None)
    ap.add_argument("--board", type=int, default=None, help="board size (default 14; 8 with --config ref)")
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16"])
    ap.add_argument("--no-alt-dtype", action="store_true", help="skip the extra measurement with the other 16-bit operand type")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timing", action="store_true", help="developer knob: no HIP events between the stages (what do they cost?)")
    ap.add_argument("--no-alt-policy-head", action="store_true", help="skip the extra legal-only-policy-head measurement")
    ap.add_argument("--no-dropin", action="store_true", help="skip the extra measurement through mcts.MCTS.search + the reference's play loop")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs (FETCH_SIZE, WRITE_SIZE over the network forward) behind the timed region; "
                         "roofline.traffic then comes from the committed profiles/pmc_summary.json")
    ap.add_argument("--policy-head", choices=["full", "legal"], default="full",
                    help="full: whole policy Linear + full softmax (reference arithmetic, the headline); legal: opt-in legal-moves-only head")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N>1 rehearsal on a single-GPU box: every rank uses device 0 (use with --backend gloo)")
    ap.add_argument("--launch-dry-run", action="store_true",
                    help="N>1 without an external launcher: start the N ranks, have each report its environment and exit (no GPU, no torch)")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="seconds the self-launcher waits for its ranks")
    args = ap.parse_args(argv)
    preset = {"1": (400, 10, 128, 256, 14), "3": (800, 20, 256, 256, 14), "ref": (50, 15, 256, 100, 8)}[args.config or "1"]
    args.sims = preset[0] if args.sims is None else args.sims
    args.blocks = preset[1] if args.blocks is None else args.blocks
    args.hidden = preset[2] if args.hidden is None else args.hidden
    args.games = preset[3] if args.games is None else args.games
    args.board = preset[4] if args.board is None else args.board
    return args


def workload_label(G, sims, blocks, hidden, R):
    """which BASELINE.json config (if any) these arguments are"""
    shape = "%d concurrent games/GPU x %d sims/move, ResNet(%d,%d), %dx%d board, start=%s" % (
        G, sims, blocks, hidden, R, R, {14: "STANDARD", 8: "EIGHT_SIMPLE"}.get(R, "%dx%d layout" % (R, R)))
    if (G, sims, blocks, hidden, R) == (256, 400, 10, 128, 14):
        return "configs[1]: " + shape
    if (sims, blocks, hidden, R) == (800, 20, 256, 14):
        return "configs[3]: " + shape
    if (G, sims, blocks, hidden, R) == (100, 50, 15, 256, 8):
        return "reference default (alphazero.py:288-304; the model and sizes the reference ships with, not a BASELINE.json config): " + shape
    return "custom (not a BASELINE.json config): " + shape


def cpu_share(local_rank, local_world, cpus=None):
    """rank r's slice of the CPUs this process may run on (contiguous, disjoint, at least one each while
    there are at least as many CPUs as ranks; with fewer CPUs than ranks the ranks take turns on them)"""
    cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    n = len(cpus)
    if n >= local_world:
        lo, hi = local_rank * n // local_world, (local_rank + 1) * n // local_world
        return cpus[lo:hi]
    return [cpus[local_rank % n]]


def pin_rank(local_rank, local_world):
    """apply this rank's CPU share: the launcher's choice (FPC_BENCH_CPUS) or, under an external launcher such as
    torch.distributed.run, the same slice computed from the inherited mask.  Returns the CPUs now in force."""
    want = os.environ.get("FPC_BENCH_CPUS")
    cpus = [int(c) for c in want.split(",") if c != ""] if want else (cpu_share(local_rank, local_world) if local_world > 1 else None)
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            pass
    return sorted(os.sched_getaffinity(0))


def launch_ranks(args, argv):
    """`bench.py --gpus N` without torchrun: this process becomes the launcher.  It has made no HIP /
    torch.cuda call (torch is not even imported yet), starts one child per GPU with the torchrun
    environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) and waits; rank 0's
    stdout (the one JSON line) passes through.  Any child that fails takes the job down: the others
    are terminated by PID and the launcher exits non-zero."""
    import signal
    import socket
    import subprocess
    n = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "FPC_BENCH_LAUNCHER_PID": str(os.getpid()),
                    # every rank gets its own share of the host cores this job may use (the per-ply host section and
                    # the launch queue of one rank must not migrate onto, or share a core with, another rank's)
                    "FPC_BENCH_CPUS": ",".join(str(c) for c in cpu_share(r, n))})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = subprocess.PIPE if args.launch_dry_run else None
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out))
    deadline = time.time() + args.launch_timeout
    rc, failed = 0, None
    live = set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc, failed = (code if code > 0 else 1), r
                break
        if rc == 0 and time.time() > deadline:      # a rank's own exit code, found in this same pass, is the verdict
            rc, failed = 124, -1
        if live and rc == 0:
            time.sleep(0.05)
    if rc != 0:
        for r in sorted(live):          # exactly the children this launcher started
            procs[r].send_signal(signal.SIGTERM)
        t_end = time.time() + 10
        for r in sorted(live):
            try:
                procs[r].wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
        print("bench.py launcher: rank %s %s; job stopped (exit %d)" %
              (failed if failed >= 0 else "?", "failed" if failed >= 0 else "timed out", rc), file=sys.stderr, flush=True)
    if args.launch_dry_run:
        probes = []
        for p_ in procs:
            txt = p_.stdout.read().decode() if p_.stdout else ""
            for line in txt.splitlines():
                if line.startswith("{"):
                    probes.append(json.loads(line)["launch_probe"])
        print(json.dumps({"launched": n, "exit": rc, "ranks": sorted(probes, key=lambda d: d["rank"])}), flush=True)
    return rc


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))         # before any torch / HIP call in this process
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    cpus_in_force = pin_rank(local, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    if args.launch_dry_run:
        print(json.dumps({"launch_probe": {"rank": rank, "local_rank": local, "world": world, "cpus": cpus_in_force,
                                           "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")),
                                           "ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}}), flush=True)
        sys.exit(int(os.environ.get("FPC_BENCH_PROBE_EXIT_RANK%d" % rank, "0")))
    import torch
    import torch.distributed as dist
    if args.rehearse_one_gpu:
        local = 0
    red_dev = "cuda" if args.backend == "nccl" else "cpu"      # where the few scalar reductions of this script live
    if world > 1:
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    import fpc_ffi
    import net
    import positions
    import tuples as tuples_mod
    import weights

    R = args.board
    INV = {8: 2, 10: 2, 13: 3, 14: 3}[R]
    G, sims = args.games, args.sims
    dt = 0 if args.dtype == "bf16" else 1
    torch.manual_seed(0)
    model = net.ResNet(Spec(R), args.blocks, args.hidden, "cpu").eval()
    eng = fpc_ffi.Engine(R, INV, max_games=G, max_sims=sims, device=local, nn_dtype=dt)
    eng.set_policy_mode(args.policy_head == "legal")
    eng.load_weights(weights.export_weights(model, dt))
    nn_kernel = (eng.L.fpc_nn_kernel(eng.h) or b"").decode()
    turn, entries = positions.start_entries(R)
    start = fpc_ffi.pods_of([fpc_ffi.board_from_dict(R, turn, entries)])[0]       # the start position as one 288-byte row
    boards = np.repeat(start[None, :], G, axis=0)                                  # [G, 288]: all concurrent games in one array
    rng = np.random.default_rng(1234 + rank)
    # episode bookkeeping: every concurrent game carries a job-wide unique id (id mod world = the rank
    # that played it); tuples are built on the device by fpc_collect_tuples, z is assigned when a game
    # ends (alphazero.py:128-137, quirk Q12) or, for games still running at the end, by the material
    # heuristic (alphazero.py:161-175)
    state = {"ids": (np.arange(G, dtype=np.int64) * world + rank).astype(np.int32), "next": G, "ply": np.zeros(G, np.int64), "step": 0,
             "host_s": 0.0, "enqueue_s": 0.0, "plies": 0}
    eng.tuples_reserve(G * (args.steps + 1))
    TURN = fpc_ffi.TURN_OFFSET

    def step(record):
        """one ply of every game.  The host section behind the search (move choice, tuples, TakeAction,
        GetGameResult, episode bookkeeping: alphazero.py:104-144) works on whole arrays: no per-game Python."""
        nonlocal boards
        t_a = time.perf_counter()
        eng.search_begin_np(boards, 3.0)
        eng.search_run(sims)
        t_b = time.perf_counter()                       # every launch of the ply is queued; the GPU is at work
        res = eng.search_results(roots_np=boards)       # blocks until the search is done; roots come back as the search left them
        t_c = time.perf_counter()
        flats = pick_moves(res, rng, 1.1)
        if record:      # (state, pi) of this ply for every game, on the device
            eng.collect_tuples(state["ids"], state["step"])
        state["step"] += 1
        ok = np.nonzero(flats >= 0)[0]
        nxt = eng.take_action_np(boards[ok], flats[ok])
        results = eng.game_result_np(nxt)               # rewrites nxt in place like the reference's GetGameResult (list order)
        state["ply"][ok] += 1
        cont = results == 0
        done = ok[~cont]
        losing_team = boards[done, TURN] & 1            # the team that just moved (Q12): the turn BEFORE the move
        done_ids = state["ids"][done].copy()
        z0 = np.where(losing_team != 0, 1.0, -1.0).astype(np.float32)
        z1 = np.where(losing_team != 1, 1.0, -1.0).astype(np.float32)
        boards[ok[cont]] = nxt[cont]
        fresh = np.concatenate([done, np.nonzero(flats < 0)[0]])
        boards[fresh] = start                           # finished game -> new episode ...
        state["ids"][done] = ((state["next"] + np.arange(len(done))) * world + rank).astype(np.int32)   # ... under a new id
        state["next"] += len(done)
        state["ply"][done] = 0
        if record and len(done):
            eng.tuples_set_z(done_ids, z0, z1)          # all games that ended on this ply in one call
        t_d = time.perf_counter()
        state["host_s"] += t_d - t_c
        state["enqueue_s"] += t_b - t_a
        state["plies"] += 1
        return int(res["sims_done"].sum())

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_pair(elapsed_s, count):
        """max-over-ranks time, sum-over-ranks count"""
        if world == 1:
            return elapsed_s, count
        tt = torch.tensor([elapsed_s], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ts = torch.tensor([float(count)], device=red_dev, dtype=torch.float64)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        return float(tt.item()), int(ts.item())

    def gather_tuples():
        """episode end: z of the games still running (heuristic), then the all-gather of this rank's
        tuples over RCCL/xGMI, issued by the engine's C++ host.  Only the collective runs
        here (inside the timed region); the records are parsed afterwards, at every N alike."""
        team = (boards[:, TURN] & 1).astype(np.int64)
        h = np.array([eng.L.fpc_board_heuristic(fpc_ffi.board_of(boards[g]), int(team[g])) for g in range(G)], np.float64) * 0.02
        eng.tuples_set_z(state["ids"], np.where(team == 0, h, -h), np.where(team == 1, h, -h))
        return tuples_mod.exchange_raw(eng)

    exchange = {"path": "single process (no exchange)", "backend": None}
    if world > 1:
        exchange = {"path": "torch.distributed all_gather of the native tuple PODs", "backend": args.backend}
        if args.backend == "nccl":
            # the engine's own RCCL communicator (C++ host); its 128-byte id travels over torch.distributed.
            # init_comm gives every rank the same verdict; with --backend nccl a failure stops the job
            ok, why = tuples_mod.init_comm(eng, device=torch.device("cuda", local))
            if not ok:
                if rank == 0:
                    print("bench.py: fpc_comm_init failed, job stopped (no silent fallback at --backend nccl): %s" % why, file=sys.stderr, flush=True)
                dist.destroy_process_group()
                sys.exit(3)
            exchange["path"] = "fpc_allgather_tuples: ncclAllGather issued by the engine's C++ host"

    for _ in range(args.warmup):
        step(False)
    state.update({"host_s": 0.0, "enqueue_s": 0.0, "plies": 0})
    eng.stats_reset()
    eng.set_timing(not args.no_stage_timing)
    sync()
    t0 = time.perf_counter()
    total = 0
    for _ in range(args.steps):
        total += step(True)
    raw = gather_tuples()
    sync()
    t1 = time.perf_counter()
    eng.set_timing(False)
    elapsed = t1 - t0
    host_ms_per_ply = 1e3 * state["host_s"] / max(state["plies"], 1)
    enqueue_ms_per_ply = 1e3 * state["enqueue_s"] / max(state["plies"], 1)
    st = eng.stats()
    elapsed, total = reduce_pair(elapsed, total)
    # outside the timed region: what arrived?  ranks_seen = distinct source ranks among the gathered game ids
    recs = tuples_mod.parse_raw(eng, raw)
    exchange["tuples_gathered"] = len(recs)
    exchange["bytes_gathered"] = len(recs) * 1280
    exchange["ranks_seen"] = len({r["game"] % world for r in recs})
    exchange["ranks_in_communicator"] = raw.get("ranks", 1)
    if exchange["ranks_seen"] != world:
        print("bench.py: tuples of %d ranks arrived, world is %d" % (exchange["ranks_seen"], world), file=sys.stderr, flush=True)
        sys.exit(4)

    # Reported beside the headline, never as `value`: the same job with the opt-in legal-only policy head
    # (fpc_set_policy_mode(FPC_POLICY_LEGAL)): the policy Linear is evaluated only at
    # the leaves' legal moves.  Same priors up to f32 rounding; not the reference's op-for-op arithmetic.
    alt = None
    if args.policy_head == "full" and not args.no_alt_policy_head:
        eng.set_policy_mode(True)
        step(False)                       # builds the row-major weight copy once
        sync()
        a0 = time.perf_counter()
        atotal = 0
        for _ in range(3):
            atotal += step(False)
        sync()
        aelapsed, atotal = reduce_pair(time.perf_counter() - a0, atotal)
        eng.set_policy_mode(False)
        alt = {"value": atotal / aelapsed, "unit": "sims/s", "steps": 3,
               "note": "NOT the headline: policy Linear evaluated only at the leaves' legal moves (softmax denominator "
                       "cancels in mask+renormalise); priors equal the full head's within 2e-5 (f32 rounding), so a PUCT near-tie can "
                       "resolve differently: tests/test_nn_gpu.py::test_legal_only_policy_head_matches_full bounds how many games "
                       "may search differently; opt-in via fpc_set_policy_mode"}

    # Reported beside the headline, never as `value`: the same job with the OTHER 16-bit MFMA operand
    # type (BASELINE configs[1] names bf16; the headline is fp16 because only fp16 meets the 1e-3 logits bar)
    alt_dtype = None
    other = "bf16" if args.dtype == "fp16" else "fp16"
    if not args.no_alt_dtype:
        eng.close()
        odt = 0 if other == "bf16" else 1
        eng = fpc_ffi.Engine(R, INV, max_games=G, max_sims=sims, device=local, nn_dtype=odt)
        eng.set_policy_mode(args.policy_head == "legal")
        eng.load_weights(weights.export_weights(model, odt))
        eng.tuples_reserve(G)
        boards = np.repeat(start[None, :], G, axis=0)
        state.update({"ids": (np.arange(G, dtype=np.int64) * world + rank).astype(np.int32), "next": G, "ply": np.zeros(G, np.int64), "step": 0})
        step(False)
        sync()
        b0 = time.perf_counter()
        btotal = 0
        for _ in range(4):
            btotal += step(False)
        sync()
        belapsed, btotal = reduce_pair(time.perf_counter() - b0, btotal)
        alt_dtype = {"dtype": other, "value": btotal / belapsed, "unit": "sims/s", "steps": 4,
                     "note": "same workload, other 16-bit MFMA operand type; see float_parity for which type meets the 1e-3 logits bar"}
    eng.close()

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    fwd = max(int(st["launches_nn"]), 1)
    A_ch = 8 * R + 8
    A = A_ch * R * R
    F, Nb = args.hidden, args.blocks
    flops_tower = 2.0 * 9 * R * R * (24 * F + 2 * Nb * F * F + F * A_ch + 24 * F) * G      # SURVEY 8(d), convs incl. both heads
    flops_fc = 2.0 * (A * A + 24 * R * R) * G                                             # policy + value Linear
    tower_ms, fc_ms = st["ms_tower"] / fwd, st["ms_fc"] / fwd
    sel_ms, exp_ms = st["ms_select"] / fwd, st["ms_expand"] / fwd
    peak = PEAK_TFLOPS[args.dtype]
    ach_tower = flops_tower / (tower_ms * 1e-3) / 1e12 if tower_ms > 0 else 0.0
    ach_fc = flops_fc / (fc_ms * 1e-3) / 1e12 if fc_ms > 0 else 0.0
    fc_gw = 384 if weights.default_fc_layout(R) == 2 else 256
    Np, Kp = (A + fc_gw - 1) // fc_gw * fc_gw, (A + 511) // 512 * 512
    # algorithmic bytes of the policy Linear inside the fused search: the 16-bit weight matrix read once + X read once.
    # (The dense f32 logits matrix -- 4 G A bytes -- is no longer written there: k_fc_reduce leaves the softmax records,
    # the expansion reads the split-K slabs at the legal moves.  FPC_DENSE_LOGITS=1 brings the write back.)
    fc_bytes = 2.0 * Np * Kp + 2.0 * G * Kp + (4.0 * G * A if os.environ.get("FPC_DENSE_LOGITS") and os.environ.get("FPC_DEV_KNOBS") == "1" else 0.0)
    pmc = {}
    try:     # HBM bytes per launch measured with rocprofv3 --pmc (tools/pmc_nn.sh), committed under profiles/
        pmc = json.load(open(os.path.join(HERE, "profiles", "pmc_summary.json")))
    except Exception:
        pass

    def pmc_traffic(kernels, shape_key):
        """counter HBM bytes per launch of exactly these kernels on exactly this network shape, or None: counted by this
        run's own rocprofv3 --pmc child passes (live_traffic) when they succeeded, else from the committed
        profiles/pmc_summary.json (keyed by shape since round 3)"""
        if live is not None:
            vals = [live.get(k) for k in kernels]
            if vals and all(v is not None for v in vals):
                return sum(vals)
        ent = pmc.get(shape_key, {})
        vals = [ent.get(k, {}).get("hbm_bytes") for k in kernels]
        return sum(vals) if vals and all(v is not None for v in vals) else None

    def traffic_source(kernels):
        if live is not None and all(live.get(k) is not None for k in kernels):
            return ("counted in THIS run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two separate child passes, kernel trace only) over "
                    "fpc_nn_forward of this shape on this box (tools/nn_only.py; FETCH_SIZE doubled per the guide's gfx950 correction); "
                    "fpc_nn_forward writes the dense logits too: k_fc_reduce carries 4 G A bytes the fused search does not")
        return "profiles/pmc_summary.json[%s] (rocprofv3 --pmc passes of an earlier run of this kernel and shape, committed; NOT counted in this run)" % shape_key

    shape_key = "r%d_b%d_h%d_g%d" % (R, Nb, F, G)
    live = None
    if world == 1 and not args.no_live_traffic and not under_profiler():
        live = live_traffic(R, Nb, F, G, dt)          # {kernel: HBM bytes per launch} counted NOW on this box, or None
    tower_desc = {"k_tower": "k_tower (residual tower megakernel, hidden 128, LDS-resident activations)",
                  "k_towerc": "k_towerc (residual tower megakernel, hidden 128, LDS-resident activations on the compact 14x14 image: 13 row tiles, LDS-DMA weight ring)",
                  "k_towerw": "k_towerw (residual tower megakernel, hidden %d, two waves per SIMD, weights L2 -> registers, LDS-resident activations)" % F,
                  "k_conv3x3": "k_conv3x3 x %d launches (per-layer implicit GEMM, activations through L2) + k_value_tail" % (2 * Nb + 3)}.get(nn_kernel, nn_kernel)
    fc_kernels = [{0: "k_fc", 1: "k_fc16", 2: "k_fcw"}[weights.default_fc_layout(R)], "k_fc_reduce"]   # the Linear (by the weight layout exported) and its split-K reduce
    tree_ms = sel_ms + exp_ms
    label = workload_label(G, sims, Nb, F, R)
    out = {
        "metric": baseline_metric() if label.startswith("configs[1]") else "MCTS simulations/sec (whole node), %d games x %d sims, %d-block/%d-filter ResNet" % (G, sims, Nb, F),
        "value": total / elapsed, "unit": "sims/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": label,
                   "games_per_gpu": G, "sims_per_move": sims, "blocks": Nb, "hidden": F, "board": R,
                   "parallelism": "games sharded, %d/GPU" % G, "ranks": world,
                   "tuple_exchange": exchange},
        # dominant kernel: the residual tower (stem + 2*Nb residual convs + both head convs), one launch per network forward
        "roofline": {"bound": "mfma", "achieved": ach_tower, "peak": peak, "unit": "TFLOP/s", "frac": ach_tower / peak,
                     "traffic": pmc_traffic([nn_kernel], shape_key),
                     "traffic_source": traffic_source([nn_kernel]),
                     "kernel": tower_desc,
                     "flops_per_launch": flops_tower, "ms_per_launch": tower_ms},
        # the policy Linear at M = 256: 255 FLOP per weight byte, below the 312 FLOP/B ridge -> HBM-bound.
        # algorithmic bytes = the 16-bit weight matrix read once + X read once (fc_bytes above)
        "roofline_policy_linear": {"bound": "hbm", "achieved": fc_bytes / (fc_ms * 1e-3) / 1e9 if fc_ms > 0 else 0.0, "peak": PEAK_HBM_GBS,
                                   "unit": "GB/s", "frac": (fc_bytes / (fc_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if fc_ms > 0 else 0.0,
                                   "traffic": pmc_traffic(fc_kernels, shape_key),
                                   "traffic_source": traffic_source(fc_kernels),
                                   "kernel": " + ".join(fc_kernels) + " (weight-streaming Linear, %.2f GB of 16-bit weights per launch)" % (2.0 * Np * Kp / 1e9),
                                   "bytes_per_launch": fc_bytes, "flops_per_launch": flops_fc, "ms_per_launch": fc_ms,
                                   "mfma_TFLOPs": ach_fc},
        "stage_ms_per_sim_step": {"select+encode": sel_ms, "tower": tower_ms, "policy_linear": fc_ms, "expand+backup": exp_ms},
        "stage_note": "HIP-event intervals on the engine's stream, sampled on every 16th simulation step and scaled: they include the "
                      "launch boundary behind each kernel and read 2-3 % above the kernel-trace durations (profiles/r03: 0.2574 ms "
                      "here vs 0.2554 ms traced for k_tower), so the four do not add up to ms_per_step / sims exactly; expand+backup is "
                      "k_expand_select = expansion + backup of step s and selection of step s+1; select+encode is what is left between "
                      "two steps plus the first step's k_select",
        # host side of one ply (one `step`): what runs between search_results returning and the next search_begin
        # (move choice, tuples, TakeAction, GetGameResult, episode bookkeeping; all array-wide), and the time the
        # launches of a ply take to queue (overlapped with the GPU).  At N ranks on one host these show contention.
        "host_ms_per_ply": host_ms_per_ply, "enqueue_ms_per_ply": enqueue_ms_per_ply,
        "host_cpus": {"count": len(cpus_in_force), "first": cpus_in_force[0], "last": cpus_in_force[-1]},
        "tree_hbm": {"bound": "hbm", "algorithmic_bytes_per_sim": tree_bytes_per_sim(R),
                     "achieved": tree_bytes_per_sim(R) * G / (tree_ms * 1e-3) / 1e9 if tree_ms > 0 else 0.0,
                     "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": (tree_bytes_per_sim(R) * G / (tree_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if tree_ms > 0 else 0.0,
                     "kernels": "k_select (first step) + k_expand_select (expansion of step s fused with the selection of step s+1; the leaf encode happens inside the tower kernel) -- latency-bound: one wavefront per game"},
    }
    if not args.no_cpu_baseline and world == 1:      # reported baselines: rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(R, INV, model, args)
        out["cpu_baseline_config0"] = cpu_baseline_config0()
    if alt is not None:
        out["alt_policy_head_legal_only"] = alt
    if alt_dtype is not None:
        out["alt_dtype_" + other] = alt_dtype
    if world == 1 and not args.no_alt_dtype:
        out["float_parity"] = float_parity(R, args.blocks, args.hidden, INV)
    if world == 1 and not args.no_dropin:
        # the same workload through the reference's own entry point (drop-in surface), beside the headline
        out["dropin"] = dropin_measure(model, R, G, sims, dt)
        out["dropin_sims_per_s"] = out["dropin"]["value"]
        out["dropin_over_value"] = out["dropin"]["value"] / out["value"]
    out["config"]["policy_head"] = args.policy_head
    if under_profiler():
        # rocprofv3 is wrapped around this process: its tracing inflates the HIP-event intervals the roofline fractions
        # above are computed from.  tools/trace_roofline.py recomputes them from the kernel trace written by the same run.
        out["under_profiler"] = True
        out["roofline"]["frac_note"] = "event time under rocprofv3 (inflated); see frac_from_kernel_trace once tools/trace_roofline.py has run"
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def dropin_measure(model, R, G, sims, dt, plies=3):
    """The same workload through the DROP-IN surface instead of the C-ABI: `mcts.MCTS(gameType, model, args).search(states)`
    followed, per game, by exactly the calls of the reference's play loop (alphazero.py:99-144): GetChildren /
    GetMoveMade().GetFlatIndex() / GetVisitCount into pi, MemoryEntry, the temperature draw, Move(flat), TakeAction,
    SetRootState(GetRootState()), GetGameResult, finished games deleted from the list -- one game at a time, in Python,
    as the reference does it.  The caller's own arithmetic (pi / pow / the draw) is done on the children's entries only:
    the same numbers as the reference's dense 23 520-wide torch ops, which belong to the caller, not to the surface
    measured here.  One untimed ply, then `plies` timed ones; sims/s counts the simulations the searches really made."""
    import alphazero_cpp as az
    az.configure(R)
    from fen_parser import parse_board_args_from_fen
    from four_player_chess_board import FourPlayerChess
    from mcts import MCTS
    import torch
    mcts = MCTS(FourPlayerChess, model, {"C": 3.0, "num_searches": sims, "nn_dtype": dt, "pool_size": 10})
    rng = np.random.default_rng(99)
    states = [FourPlayerChess(*parse_board_args_from_fen(FourPlayerChess.start_fen, R)) for _ in range(G)]
    T = 1.1
    clk = time.perf_counter
    sims_done, t0 = 0, None
    sec = {"search": 0.0, "children_reads": 0.0, "caller_arithmetic": 0.0, "take_action_game_result": 0.0, "other_surface_calls": 0.0}
    for ply in range(plies + 1):
        if ply == 1:
            torch.cuda.synchronize()
            sims_done, t0 = 0, clk()
            for k in sec:
                sec[k] = 0.0
        ta = clk()
        roots = mcts.search(states)
        sec["search"] += clk() - ta
        for i in reversed(range(len(states))):
            state = states[i]
            t1 = clk()
            flats, visits = [], []
            for child in roots[i].GetChildren():
                flats.append(child.GetMoveMade().GetFlatIndex())
                visits.append(child.GetVisitCount())
            sims_done += int(roots[i].GetVisitCount()) - 1             # root N = 1 + simulations made (mcts.py:30)
            t2 = clk()
            pi = np.asarray(visits, np.float32)
            pi /= pi.sum()
            tp = np.power(pi, np.float32(1.0 / T))
            c = np.cumsum(tp.astype(np.float64))
            action_index = flats[min(int(np.searchsorted(c, rng.random() * c[-1], side="right")), len(flats) - 1)]
            t3 = clk()
            state.AppendToMemory(az.MemoryEntry(state, (flats, pi)))
            action = az.Move(action_index)
            t4 = clk()
            next_state = state.TakeAction(action)
            t5 = clk()
            next_state.SetRootState(state.GetRootState())
            t6 = clk()
            game_state = next_state.GetGameResult()
            t7 = clk()
            if game_state != az.GameResult.IN_PROGRESS:
                del states[i]
            else:
                states[i] = next_state
            sec["children_reads"] += t2 - t1
            sec["caller_arithmetic"] += t3 - t2
            sec["other_surface_calls"] += (t4 - t3) + (t6 - t5)
            sec["take_action_game_result"] += (t5 - t4) + (t7 - t6)
        while len(states) < G:        # keep the batch at G games (the reference lets it shrink; throughput is quoted at G)
            states.append(FourPlayerChess(*parse_board_args_from_fen(FourPlayerChess.start_fen, R)))
    dt_s = clk() - t0
    eng = az.engine()
    eng.close()
    az._engine = None
    return {"value": sims_done / dt_s, "unit": "sims/s", "plies": plies, "ms_per_ply": 1e3 * dt_s / plies,
            "ms_per_ply_by_part": {k: 1e3 * v / plies for k, v in sec.items()},
            "parts": "search = MCTS.search (weights check, upload of the roots, the whole search on the GPU, root read-back); children_reads = "
                     "GetChildren + GetMoveMade().GetFlatIndex() + GetVisitCount for every root child; take_action_game_result = TakeAction + "
                     "GetGameResult per game (answered from ONE batched prefetch per search); other_surface_calls = MemoryEntry, Move(flat), "
                     "GetRootState / SetRootState; caller_arithmetic = pi, temperature, the draw (the caller's, on the children's entries)",
            "path": "mcts.MCTS.search + the reference's play loop (alphazero.py:99-144), one game at a time in Python"}


def float_parity(R, blocks, hidden, INV):
    """max|dlogit| / max|dvalue| of the engine's network against the fp32 logits recorded from the
    reference's own net.py (tests/golden/net_r*_b*_h*.npz, oracle/gen_net_golden.py), both operand
    types -- the measurement behind the choice of the headline dtype (north_star: within 1e-3)."""
    import torch
    import fpc_ffi
    import weights
    try:
        import net_cases as nc
        fx = nc.load_net_fixture(R, blocks, hidden)
    except Exception as exc:
        return {"error": "no reference-net fixture for this shape: %r" % (exc,)}
    model = nc.fixture_model(fx)
    boards = nc.fixture_boards(fx)
    n = len(boards)
    out = {"tolerance": 1e-3, "fixture": "tests/golden/net_r%d_b%d_h%d.npz (reference net.py, fp32 CPU, %d golden positions)" % (R, blocks, hidden, n)}
    for name, dt in (("fp16", 1), ("bf16", 0)):
        eng = fpc_ffi.Engine(R, INV, max_games=n, max_sims=4, nn_dtype=dt)
        eng.load_weights(weights.export_weights(model, dt))
        enc = np.concatenate([eng.encode([b]) for b in boards])
        x = torch.from_numpy(enc).cuda()
        lg = torch.empty(n, eng.A, device="cuda")
        va = torch.empty(n, device="cuda")
        torch.cuda.synchronize()
        eng.nn_forward(x.data_ptr(), n, lg.data_ptr(), va.data_ptr())
        el = float(np.abs(lg.cpu().numpy()[:, fx["idx"]] - fx["logits"]).max())
        ev = float(np.abs(va.cpu().numpy() - fx["value"]).max())
        out[name] = {"max_abs_dlogit": el, "max_abs_dvalue": ev, "meets_tolerance": bool(el < 1e-3 and ev < 1e-3)}
        eng.close()
    return out


def live_traffic(R, blocks, hidden, G, dt):
    """HBM bytes per launch of the network forward's kernels, counted now: two child runs of tools/nn_only.py under
    `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE, then WRITE_SIZE: separate passes, no other trace domain, as
    MI355X_MICROARCH.md prescribes; this process only starts them and reads their CSVs).  {kernel base name: bytes}, or
    None if rocprofv3 is missing or a pass fails (roofline.traffic then falls back to the committed summary)."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe) or R != 14 or G != 256:
        return None                                   # tools/nn_only.py drives the 14x14, 256-row forward
    tmp = tempfile.mkdtemp(prefix="fpc_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", FPC_NN_BLOCKS=str(blocks), FPC_NN_HIDDEN=str(hidden), FPC_NN_DTYPE=str(dt))
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            # its own process group: on a timeout the profiler AND the python it started are ended (exactly that group)
            pr = subprocess.Popen([exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
                                   sys.executable, os.path.join(HERE, "tools", "nn_only.py"), "3"],
                                  cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=100)
            except subprocess.TimeoutExpired:
                import signal
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.wait()
                return None
            fs = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
            if rc != 0 or not fs:
                return None
            agg = collections.defaultdict(list)
            for row in csv.DictReader(open(fs[0])):
                if row["Counter_Name"] == counter:
                    agg[row["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]].append(float(row["Counter_Value"]))
            vals[counter] = {k: sum(v) / len(v) for k, v in agg.items()}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {k: (2.0 * vals["FETCH_SIZE"][k] + vals["WRITE_SIZE"].get(k, 0.0)) * 1024.0 for k in vals["FETCH_SIZE"]}


def under_profiler():
    """True when rocprofv3 / rocprofiler-sdk is wrapped around this process (its tool library is preloaded or configured)"""
    if any(k.startswith(("ROCPROF", "ROCPROFILER_", "ROCP_")) for k in os.environ):
        return True
    return "rocprofiler" in os.environ.get("LD_PRELOAD", "")


def baseline_metric():
    try:
        return json.load(open(os.path.join(HERE, "BASELINE.json")))["metric"]
    except Exception:
        return "MCTS simulations/sec (whole node), 256 games x 400 sims, 10-block ResNet"


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_baseline(R, INV, model, args):
    """The CPU oracle's MCTS.search (oracle/fpc_oracle.cpp: scalar port of the reference algorithm) with a PyTorch-CPU fp32
    ResNet of the same shape as evaluator, on a bounded sample of the SAME BATCH SHAPE as the workload: all G concurrent
    games, a few simulations each (the policy Linear's 2.2 GB of fp32 weights are then amortised over G rows per forward,
    as on the GPU).  `small_batch` beside it is the earlier sample (16 games x 200 sims: weights amortised over 16 rows)."""
    import torch
    from oracle import orc
    import positions
    cores = host_cores()
    torch.set_num_threads(cores)
    turn, entries = positions.start_entries(R)

    def ev(enc):
        with torch.no_grad():
            lg, v = model(torch.from_numpy(np.ascontiguousarray(enc)))
        return lg.numpy(), v.squeeze(1).numpy()

    def run(Gc, sc):
        boards = [orc.board_from_dict(R, turn, [list(e) for e in entries]) for _ in range(Gc)]
        t0 = time.perf_counter()
        rc, res = orc.search(boards, R, INV, sc, 3.0, ev)
        dt = time.perf_counter() - t0
        return sum(r["sims_done"] for r in res) / dt, dt

    Gc, sc = args.games, 24           # ~12 s of CPU work on the GPU box's 16 host cores (8 sims took 4.0 s)
    v, dt = run(Gc, sc)
    out = {"value": v, "unit": "sims/s", "cores": cores, "kind": "port",
           "sample": "%d games x %d sims from the start position (the workload's batch shape), oracle tree + PyTorch-CPU fp32 ResNet(%d,%d), %.1f s"
                     % (Gc, sc, args.blocks, args.hidden, dt)}
    v2, dt2 = run(16, 100)
    out["small_batch"] = {"value": v2, "unit": "sims/s", "cores": cores,
                          "sample": "16 games x 100 sims, same evaluator, %.1f s" % dt2}
    return out


def cpu_baseline_config0():
    """BASELINE configs[0]: 1 self-play game, 100 sims/move, 4-block/64-filter ResNet on PyTorch-CPU at
    the reference's compiled board size (8x8/2, EIGHT_SIMPLE): the oracle's MCTS.search + fp32 ResNet."""
    import torch
    import net
    import positions
    from oracle import orc
    R, INV = 8, 2
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = net.ResNet(Spec(R), 4, 64, "cpu").eval()
    turn, entries = positions.start_entries(R)

    def ev(enc):
        with torch.no_grad():
            lg, v = model(torch.from_numpy(np.ascontiguousarray(enc)))
        return lg.numpy(), v.squeeze(1).numpy()

    done, t0 = 0, time.perf_counter()
    b = orc.board_from_dict(R, turn, [list(e) for e in entries])
    for _ply in range(6):                      # six plies of one game, 100 simulations each
        rc, res = orc.search([b], R, INV, 100, 3.0, ev)
        done += res[0]["sims_done"]
        if not res[0]["children"]:
            break
        best = max(res[0]["children"], key=lambda c: c[1])[0]
        b, _ = orc.take_action(res[0]["board"], R, best)
        if orc.game_result(orc.clone(b), R, INV) != 0:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "sims/s", "cores": cores, "kind": "port",
            "sample": "configs[0]: 1 game x 100 sims/move x 6 plies, 8x8 EIGHT_SIMPLE, oracle tree + PyTorch-CPU fp32 ResNet(4,64), %.1f s" % dt}


if __name__ == "__main__":
    main()
