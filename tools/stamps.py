#!/usr/bin/env python3
"""Diagnostic build only (-DFPC_EXP_STAMP -> tools/var/lib_STAMP.so): where a k_tower wave spends its
cycles (s_memtime shares; read the SHARES, not the lengths -- the stamps' fences forbid overlaps)."""
import os, sys, ctypes
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(HERE, "alphazero-4-player-chess_amd"), HERE]
import numpy as np, torch
import fpc_ffi, net, weights
fpc_ffi.LIB_PATH = os.path.join(HERE, "tools/var/lib_STAMP.so")
from bench import Spec
R, G = 14, 256
torch.manual_seed(0)
m = net.ResNet(Spec(R), 10, 128, "cpu").eval()
eng = fpc_ffi.Engine(R, 3, max_games=G, max_sims=8)
eng.load_weights(weights.export_weights(m, 0))
x = (torch.rand(G, 24, R, R) < 0.1).float().cuda()
lg = torch.empty(G, eng.A, device="cuda"); va = torch.empty(G, device="cuda")
for _ in range(5):
    eng.nn_forward(x.data_ptr(), G, lg.data_ptr(), va.data_ptr())
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
print("rc", fpc_ffi.lib().fpc_debug_stamps(buf))
a = np.array(buf[:], dtype=np.float64).reshape(64, 8)[:32, :5]
names = ["stage: frag reads + 32 mfma + ring traffic + stage barrier", "(unused)", "stage tail", "layer barrier", "layer epilogue"]
def show(a, names):
    tot = a.mean(0).sum()
    for i, n in enumerate(names):
        print("%-28s %12.0f ticks/launch (%.1f%%)" % (n, a[:, i].mean(), 100 * a[:, i].mean() / tot))
print("k_tower"); show(a, names)
