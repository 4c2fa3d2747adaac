#!/usr/bin/env python3
"""Kernel-trace target: the UNFUSED step-wise search loop (k_select, k_expand as separate launches) on 256 games at
14x14 with the internal network -- rocprofv3 --kernel-trace --stats then prices the two halves of k_expand_select
separately.   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tree -- python3 tools/tree_split.py [sims] [board]"""
import os, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(HERE, "alphazero-4-player-chess_amd"), HERE]
import numpy as np, torch
import fpc_ffi, net, positions, weights
from bench import Spec
sims = int(sys.argv[1]) if len(sys.argv) > 1 else 200
R = int(sys.argv[2]) if len(sys.argv) > 2 else 14
INV = {8: 2, 14: 3}[R]
G = 256
torch.manual_seed(0)
m = net.ResNet(Spec(R), 2, 128, "cpu").eval()
eng = fpc_ffi.Engine(R, INV, max_games=G, max_sims=sims, nn_dtype=1)
eng.load_weights(weights.export_weights(m, 1))
turn, entries = positions.start_entries(R)
start = fpc_ffi.pods_of([fpc_ffi.board_from_dict(R, turn, entries)])[0]
boards = np.repeat(start[None, :], G, axis=0)
lg = torch.empty(G, eng.A, device="cuda"); va = torch.empty(G, device="cuda")
eng.search_begin_np(boards, 3.0)
n_live, enc = eng.search_select()
for i in range(sims):
    eng.nn_forward(enc, G, lg.data_ptr(), va.data_ptr())
    eng.search_expand(lg.data_ptr(), va.data_ptr())
    if i + 1 < sims:
        n_live, enc = eng.search_select()
res = eng.search_results()
print("done", int(res["sims_done"].sum()))
