#!/usr/bin/env python3
"""One-off GPU soak: many more seeds of the engine-vs-oracle cases than the committed tests run
(random mid-game roots x evaluators, other board sizes, castling, arena).  Test infrastructure
(imports tests/ and oracle/); prints one line per case and exits non-zero on the first mismatch."""
import os, sys, time
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(HERE, "tests"), os.path.join(HERE, "alphazero-4-player-chess_amd"), HERE]
import conftest  # noqa: F401  (paths)
import engine_cases as ec
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
t0 = time.time()
for seed in range(100, 100 + n):
    for R, kind, games, sims in ((8, "hash", 48, 150), (14, "hash", 32, 120), (14, "ramp", 24, 100), (8, "hashinf", 24, 100), (8, "ramp", 32, 200)):
        r = ec.case_search_random_vs_oracle("gpu", R, n_games=games, sims=sims, seed=seed, kind=kind)
        print("random", R, kind, seed, r, "%.0fs" % (time.time() - t0), flush=True)
    for R, INV in ((10, 2), (13, 3)):
        ec.case_other_sizes_vs_oracle("gpu", R, INV, n_games=12, sims=60, seed=seed)
        print("size", R, seed, "ok", flush=True)
    ec.case_castling_vs_oracle("gpu", n_games=8, plies=60, sims=40, seed=seed)
    print("castling", seed, "ok", flush=True)
    for R, rules in ((14, 15), (8, 15), (14, 5)):
        k = ec.case_fixed_rules_vs_oracle("gpu", R, n_games=6, plies=60, sims=40, seed=seed, rules=rules, noise=(seed & 1) == 0)
        print("fixed-rules", R, rules, seed, k, flush=True)
    for R in (8, 14):
        k = ec.case_arena_vs_oracle("gpu", R, n_pairs=4, sims=40, max_len=40, seed=seed)
        print("arena", R, seed, k, flush=True)
print("soak ok")
