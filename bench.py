#!/usr/bin/env python3
"""bench.py -- MCTS simulations/sec of the MI355X-native self-play engine (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one MCTS.search call over the whole batch of concurrent games (one ply of self-play:
`sims` simulations per game through select -> encode -> ResNet -> expand, all on the GPU), followed
by the move selection + TakeAction + GetGameResult that the reference's play() loop performs
(alphazero.py:99-144), so that successive steps see realistic positions.  Workload at every N:
BASELINE.json configs[1] per GPU -- 256 concurrent games x 400 sims/move, 10-block/128-filter
ResNet, 14x14 STANDARD start, bf16 MFMA operands, random-init weights (torch.manual_seed(0)),
synthetic data.  N > 1: one process per GPU (torchrun), games sharded 256/GPU (weak scaling), no
data-path collective inside the search; the (state, pi, z)-tuple all-gather over RCCL that ends an
episode is executed once inside the timed region.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel group (the MFMA
implicit-GEMM network forward) with HIP events recorded on the engine's own stream during the
timed steps (every 8th simulation step carries the events; five per step cost 2.7 %); `cpu_baseline`
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
    """alphazero.py:104-119: pi ~ child visit counts, temperature, multinomial (seeded here)."""
    G = len(res["n_children"])
    flats = np.zeros(G, np.int64)
    for g in range(G):
        n = int(res["n_children"][g])
        if n == 0:
            flats[g] = -1
            continue
        p = res["visits"][g, :n].astype(np.float64)
        p /= p.sum()
        p = p ** (1.0 / temperature)
        p /= p.sum()
        flats[g] = res["flat"][g, rng.choice(n, p=p)]
    return flats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--games", type=int, default=256, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=400)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--board", type=int, default=14)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timing", action="store_true", help="developer knob: no HIP events between the stages (what do they cost?)")
    ap.add_argument("--no-alt-policy-head", action="store_true", help="skip the extra legal-only-policy-head measurement")
    ap.add_argument("--policy-head", choices=["full", "legal"], default="full",
                    help="full: whole policy Linear + full softmax (reference arithmetic, the headline); legal: opt-in legal-moves-only head")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N>1 rehearsal on a single-GPU box: every rank uses device 0 (use with --backend gloo)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if args.rehearse_one_gpu:
        local = 0
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
    turn, entries = positions.start_entries(R)
    start = fpc_ffi.board_from_dict(R, turn, entries)
    boards = [fpc_ffi.clone_board(start) for _ in range(G)]
    rng = np.random.default_rng(1234 + rank)
    tuples = []          # compact (mailbox+turn, sparse pi) records of this rank's episode

    def step(record):
        nonlocal boards
        eng.search_begin(boards, 3.0)
        eng.search_run(sims)
        res = eng.search_results(roots=boards)
        flats = pick_moves(res, rng, 1.1)
        if record:      # (state, pi) of this ply; z is assigned at episode end (alphazero.py:112,128-137)
            for g in range(G):
                n = int(res["n_children"][g])
                tuples.append(tuples_mod.pack_record(R, bytes(boards[g])[:R * R], boards[g].turn, 0.0,
                                                     res["flat"][g, :n], res["visits"][g, :n]))
        ok = [g for g in range(G) if flats[g] >= 0]
        nxt = eng.take_action([boards[g] for g in ok], [int(flats[g]) for g in ok])
        results = eng.game_result(nxt)
        for g, nb, r in zip(ok, nxt, results):
            boards[g] = nb if r == 0 else fpc_ffi.clone_board(start)     # finished game -> new episode
        for g in range(G):
            if flats[g] < 0:
                boards[g] = fpc_ffi.clone_board(start)
        return int(res["sims_done"].sum())

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def gather_tuples():
        """episode end: all-gather of this rank's (state, pi) records over RCCL/xGMI."""
        if world == 1 or not tuples:
            return 0
        got = tuples_mod.all_gather_bytes(b"".join(tuples), device=torch.device("cuda", local))
        torch.cuda.synchronize()
        return sum(len(x) for x in got)

    for _ in range(args.warmup):
        step(False)
    eng.stats_reset()
    eng.set_timing(not args.no_stage_timing)
    sync()
    t0 = time.perf_counter()
    total = 0
    for _ in range(args.steps):
        total += step(True)
    gathered = gather_tuples()
    sync()
    t1 = time.perf_counter()
    eng.set_timing(False)
    elapsed = t1 - t0
    st = eng.stats()
    if world > 1:
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ts = torch.tensor([float(total)], device="cuda", dtype=torch.float64)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        total = int(ts.item())

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
        aelapsed = time.perf_counter() - a0
        eng.set_policy_mode(False)
        if world > 1:
            at = torch.tensor([aelapsed], device="cuda", dtype=torch.float64)
            dist.all_reduce(at, op=dist.ReduceOp.MAX)
            aelapsed = float(at.item())
            as_ = torch.tensor([float(atotal)], device="cuda", dtype=torch.float64)
            dist.all_reduce(as_, op=dist.ReduceOp.SUM)
            atotal = int(as_.item())
        alt = {"value": atotal / aelapsed, "unit": "sims/s", "steps": 3,
               "note": "NOT the headline: policy Linear evaluated only at the leaves' legal moves (softmax denominator "
                       "cancels in mask+renormalise); priors equal the full head's up to f32 rounding, visit counts identical in "
                       "tests/test_nn_gpu.py::test_legal_only_policy_head_matches_full; opt-in via fpc_set_policy_mode"}

    if rank != 0:
        if world > 1:
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
    pmc = {}
    try:     # HBM bytes per launch measured with rocprofv3 --pmc (tools/pmc_nn.sh), committed under profiles/
        pmc = json.load(open(os.path.join(HERE, "profiles", "pmc_summary.json")))
    except Exception:
        pass
    tree_ms = sel_ms + exp_ms
    out = {
        "metric": baseline_metric(),
        "value": total / elapsed, "unit": "sims/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "configs[1]: %d concurrent games/GPU x %d sims/move, ResNet(%d,%d), %dx%d board, start=%s"
                   % (G, sims, args.blocks, args.hidden, R, R, "STANDARD" if R == 14 else "default"),
                   "games_per_gpu": G, "sims_per_move": sims, "board": R, "parallelism": "games sharded, %d/GPU" % G,
                   "tuple_allgather_bytes": gathered},
        # dominant kernel: k_tower = stem + 2*Nb residual convs + both head convs, one launch per network forward
        "roofline": {"bound": "mfma", "achieved": ach_tower, "peak": peak, "unit": "TFLOP/s", "frac": ach_tower / peak,
                     "traffic": pmc.get("k_tower", {}).get("hbm_bytes"),
                     "kernel": "k_tower (residual tower megakernel, LDS-resident activations)",
                     "flops_per_launch": flops_tower, "ms_per_launch": tower_ms},
        "roofline_policy_linear": {"bound": "mfma", "achieved": ach_fc, "peak": peak, "unit": "TFLOP/s", "frac": ach_fc / peak,
                                   "traffic": pmc.get("k_fc256", {}).get("hbm_bytes"),
                                   "kernel": "k_fc256 + k_fc_reduce (weight-streaming Linear, 1.1 GB of bf16 weights per launch)",
                                   "flops_per_launch": flops_fc, "ms_per_launch": fc_ms,
                                   "weight_stream_GBps": (2.0 * ((A + 127) // 128 * 128) * ((A + 511) // 512 * 512)) / (fc_ms * 1e-3) / 1e9 if fc_ms > 0 else 0.0},
        "stage_ms_per_sim_step": {"select+encode": sel_ms, "tower": tower_ms, "policy_linear": fc_ms, "expand+backup": exp_ms},
        "tree_hbm": {"bound": "hbm", "algorithmic_bytes_per_sim": tree_bytes_per_sim(R),
                     "achieved": tree_bytes_per_sim(R) * G / (tree_ms * 1e-3) / 1e9 if tree_ms > 0 else 0.0,
                     "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": (tree_bytes_per_sim(R) * G / (tree_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if tree_ms > 0 else 0.0,
                     "kernels": "k_select + k_expand (+ the leaf encode, done inside k_tower) -- latency-bound: one wavefront per game"},
    }
    if not args.no_cpu_baseline and world == 1:      # reported baseline: rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(R, INV, model, args)
    if alt is not None:
        out["alt_policy_head_legal_only"] = alt
    out["config"]["policy_head"] = args.policy_head
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


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
    """The CPU oracle's MCTS.search (oracle/fpc_oracle.cpp: scalar port of the reference algorithm)
    with a PyTorch-CPU fp32 ResNet of the same shape as evaluator, on a bounded sample."""
    import torch
    from oracle import orc
    import positions
    cores = host_cores()
    torch.set_num_threads(cores)
    turn, entries = positions.start_entries(R)
    Gc, sc = 16, 100
    boards = [orc.board_from_dict(R, turn, [list(e) for e in entries]) for _ in range(Gc)]

    def ev(enc):
        with torch.no_grad():
            lg, v = model(torch.from_numpy(np.ascontiguousarray(enc)))
        return lg.numpy(), v.squeeze(1).numpy()

    t0 = time.perf_counter()
    rc, res = orc.search(boards, R, INV, sc, 3.0, ev)
    dt = time.perf_counter() - t0
    done = sum(r["sims_done"] for r in res)
    return {"value": done / dt, "unit": "sims/s", "cores": cores, "kind": "port",
            "sample": "%d games x %d sims from the start position, oracle tree + PyTorch-CPU fp32 ResNet(%d,%d), %.1f s"
                      % (Gc, sc, args.blocks, args.hidden, dt)}


if __name__ == "__main__":
    main()
