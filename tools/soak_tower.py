#!/usr/bin/env python3
"""One-off GPU soak of the network forward: k_tower synchronises its waves by hand (a weight ring with one barrier
per tap, two waves per SIMD with the non-loading half passing the barriers early, in-place epilogues), so a rare
ordering bug would show as an output that differs from run to run.  Thousands of forwards per shape, every result
compared BIT FOR BIT with the first one of its shape, and the first one against the fp32 torch network; other
kernels (a fused search on a second engine) run in between to vary the timing.  Exits non-zero on the first
difference.      python3 tools/soak_tower.py [forwards per shape, default 1500]"""
import os, sys, time
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(HERE, "alphazero-4-player-chess_amd"), HERE, os.path.join(HERE, "tests")]
import numpy as np
import torch
import fpc_ffi, net, positions, weights
from bench import Spec
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
t0 = time.time()
for R, blocks, hidden, G in ((14, 10, 128, 256), (14, 3, 128, 37), (8, 10, 128, 256), (8, 4, 128, 300), (14, 2, 256, 64), (12, 2, 128, 64), (10, 2, 128, 64)):
    INV = {8: 2, 10: 2, 12: 3, 14: 3}[R]
    for dt in (1, 0):
        torch.manual_seed(R * 100 + blocks)
        m = net.ResNet(Spec(R), blocks, hidden, "cpu").eval()
        eng = fpc_ffi.Engine(R, INV, max_games=G, max_sims=8, nn_dtype=dt)
        eng.load_weights(weights.export_weights(m, dt))
        x = (torch.rand(G, 24, R, R, generator=torch.Generator().manual_seed(R)) < 0.1).float()
        xd = x.cuda()
        lg = torch.empty(G, eng.A, device="cuda"); va = torch.empty(G, device="cuda")
        noise = None
        if R in (8, 14):      # a second engine whose fused search runs in between (timing noise on the same GPU)
            noise = fpc_ffi.Engine(R, INV, max_games=32, max_sims=16, nn_dtype=dt)
            noise.load_weights(weights.export_weights(m, dt))
            turn, entries = positions.start_entries(R)
            roots = [fpc_ffi.board_from_dict(R, turn, entries) for _ in range(32)]
        first = None
        for i in range(N):
            eng.nn_forward(xd.data_ptr(), G, lg.data_ptr(), va.data_ptr())
            if noise is not None and i % 97 == 0:
                noise.search_begin(roots, 3.0); noise.search_run(8)
            if i % 10 == 0 or i == N - 1:
                torch.cuda.synchronize()
                cur = (lg.cpu().numpy().copy(), va.cpu().numpy().copy())
                if first is None:
                    first = cur
                    with torch.no_grad():
                        rl, rv = m(x)
                    el = float(np.abs(first[0] - rl.numpy()).max()); ev = float(np.abs(first[1] - rv.squeeze(1).numpy()).max())
                    assert el < (1e-3 if dt else 1e-2) and ev < (1e-3 if dt else 1e-2), (R, blocks, hidden, dt, el, ev)
                elif not (np.array_equal(cur[0], first[0]) and np.array_equal(cur[1], first[1])):
                    print("DIFFERENT at forward", i, (R, blocks, hidden, G, dt), np.abs(cur[0] - first[0]).max(), flush=True)
                    sys.exit(1)
        print("ok R=%d blocks=%d hidden=%d G=%d %s kernel=%s: %d forwards identical, vs fp32 %.2e / %.2e  (%.0f s)" % (
            R, blocks, hidden, G, "fp16" if dt else "bf16", eng.L.fpc_nn_kernel(eng.h).decode(), N, el, ev, time.time() - t0), flush=True)
        eng.close()
        if noise is not None:
            noise.close()
print("soak ok")
