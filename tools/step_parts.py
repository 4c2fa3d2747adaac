#!/usr/bin/env python3
"""Developer check: where the wall time of one ply of bench.py's step() goes (search on the GPU vs the host-side
bookkeeping around it).  python3 tools/step_parts.py"""
import os, sys, time
HERE = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(HERE, "alphazero-4-player-chess_amd"), HERE, os.path.join(HERE, "tests")]
import numpy as np, torch
import fpc_ffi, net, positions, weights, bench
R, INV, G, sims = 14, 3, 256, 400
torch.manual_seed(0)
m = net.ResNet(bench.Spec(R), 10, 128, "cpu").eval()
eng = fpc_ffi.Engine(R, INV, max_games=G, max_sims=sims, nn_dtype=1)
eng.load_weights(weights.export_weights(m, 1))
turn, entries = positions.start_entries(R)
start = fpc_ffi.board_from_dict(R, turn, entries)
boards = [fpc_ffi.clone_board(start) for _ in range(G)]
rng = np.random.default_rng(1)
eng.tuples_reserve(G * 20)
T = {k: 0.0 for k in ("begin", "run", "results", "pick", "collect", "take", "result", "rest")}
def tick(): torch.cuda.synchronize(); return time.perf_counter()
for ply in range(8):
    t0 = tick(); eng.search_begin(boards, 3.0)
    t1 = tick(); eng.search_run(sims)
    t2 = tick(); res = eng.search_results(roots=boards)
    t3 = tick(); flats = bench.pick_moves(res, rng, 1.1)
    t4 = tick(); eng.collect_tuples(list(range(G)), ply)
    t5 = tick(); nxt = eng.take_action(boards, [int(f) for f in flats])
    t6 = tick(); r = eng.game_result(nxt)
    t7 = tick(); boards = [nb if rr == 0 else fpc_ffi.clone_board(start) for nb, rr in zip(nxt, r)]
    t8 = tick()
    if ply >= 2:
        for k, d in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7)): T[k] += d
tot = sum(T.values())
print({k: round(v / 6 * 1e3, 2) for k, v in T.items()}, "ms per ply; total %.1f ms; host share %.1f %%" % (tot / 6 * 1e3, 100 * (tot - T["run"]) / tot))
