#!/usr/bin/env python3
"""BASELINE.json configs[4] at its stated size: 1024 paired arena games (2048 games) between two weight sets at
temperature 0, through the fused search loop (fpc_search_run) of two engines on one GPU.  Prints one JSON line
(games/s, plies/s, sims/s + the property checks); profiles/r03/arena_1024.json is a copy of it.

    python3 tools/arena_bench.py [--pairs 1024] [--sims 100] [--max-len 40] [--blocks 10 --hidden 128]

Checks (size-independent, the oracle cannot follow at this size): every pick is the first maximum of the reported
visit counts and one of the root's legal children; every searched root reports sum N_child = #children + sims_done - 1
(quirk Q1; sims_done < sims only where a terminal leaf dropped the root, Q5); the paired games start from the same
position with the colours swapped; a second run reproduces every visit count, pick and result bit for bit."""
import argparse, json, os, sys, time
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(HERE, "alphazero-4-player-chess_amd"), HERE, os.path.join(HERE, "tests")]
import numpy as np
import torch
import arena, fpc_ffi, net, positions, weights
from bench import Spec

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=1024)
ap.add_argument("--sims", type=int, default=100)
ap.add_argument("--max-len", type=int, default=40)
ap.add_argument("--blocks", type=int, default=10)
ap.add_argument("--hidden", type=int, default=128)
ap.add_argument("--board", type=int, default=14)
ap.add_argument("--no-repeat", action="store_true")
a = ap.parse_args()
R, INV = a.board, {8: 2, 10: 2, 13: 3, 14: 3}[a.board]
G = 2 * a.pairs
models = []
for seed in (0, 1):
    torch.manual_seed(seed)
    models.append(net.ResNet(Spec(R), a.blocks, a.hidden, "cpu").eval())
engs = []
for m in models:
    e = fpc_ffi.Engine(R, INV, max_games=G, max_sims=a.sims, nn_dtype=1)
    e.load_weights(weights.export_weights(m, 1))
    engs.append(e)
turn, entries = positions.start_entries(R)
start = fpc_ffi.board_from_dict(R, turn, entries)
rng = np.random.default_rng(7)
starts = []
for k in range(a.pairs):                     # the start position advanced by 0..3 random legal plies
    b = fpc_ffi.clone_board(start)
    for _ in range(k % 4):
        lm = engs[0].legal_moves([b])[0]
        b = engs[0].take_action([b], [lm[int(rng.integers(len(lm)))][2]])[0]
    starts.append(b)

stats = {"searches": 0, "sims": 0}
def counted(eng):
    def fn(pods):
        eng.search_begin(pods, 3.0)
        eng.search_run(a.sims)
        res = eng.search_results(roots=pods)
        stats["searches"] += 1
        stats["sims"] += int(res["sims_done"].sum())
        n = res["n_children"]
        for i in range(len(pods)):           # Q1 bookkeeping of every searched root
            k = int(n[i])
            assert int(res["visits"][i, :k].sum()) == k + int(res["sims_done"][i]) - 1, (i, k)
        return res
    return fn

def run():
    stats.update(searches=0, sims=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    games = arena.play_paired(counted(engs[0]), counted(engs[1]), engs[0], [fpc_ffi.clone_board(b) for b in starts], {"max_game_length": a.max_len})
    torch.cuda.synchronize()
    return games, time.perf_counter() - t0

games, dt = run()
for g in games:
    for _t, fl, vi, pk in g.plies:
        assert pk == int(fl[int(np.argmax(vi))]) and pk in [int(x) for x in fl]
for k in range(a.pairs):
    assert games[2 * k].pair == games[2 * k + 1].pair == k and {games[2 * k].a_team, games[2 * k + 1].a_team} == {0, 1}
s = arena.summary(games)
out = {"workload": "configs[4]: %d paired games (%d games), temperature 0, ResNet(%d,%d) vs ResNet(%d,%d) (seeds 0 / 1), %d sims/move, max %d plies, %dx%d, fp16"
                   % (a.pairs, G, a.blocks, a.hidden, a.blocks, a.hidden, a.sims, a.max_len, R, R),
       "seconds": dt, "games_per_s": G / dt, "plies_per_s": s["plies"] / dt, "sims_per_s": stats["sims"] / dt,
       "searches": stats["searches"], "summary": s, "checks": "picks = first maximum of the visit counts and legal; Q1 visit bookkeeping on every searched root; pairing"}
if not a.no_repeat:
    games2, dt2 = run()
    same = all(len(g.plies) == len(h.plies) and g.winner == h.winner and all(
        p[0] == q[0] and p[3] == q[3] and np.array_equal(p[1], q[1]) and np.array_equal(p[2], q[2]) for p, q in zip(g.plies, h.plies))
        for g, h in zip(games, games2))
    assert same, "second run differs"
    out["repeat_identical"] = True
    out["seconds_repeat"] = dt2
print(json.dumps(out), flush=True)
for e in engs:
    e.close()
