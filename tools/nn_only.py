#!/usr/bin/env python3
"""Runs only the network forward (fpc_nn_forward) a few times: a short target for rocprofv3 --pmc."""
import os, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(HERE, "alphazero-4-player-chess_amd"), HERE]
import numpy as np, torch
import fpc_ffi, net, weights
from bench import Spec
R, G, iters = 14, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 5
torch.manual_seed(0)
NB, NH = int(os.environ.get("FPC_NN_BLOCKS", "10")), int(os.environ.get("FPC_NN_HIDDEN", "128"))   # configs[3]: 20 / 256
m = net.ResNet(Spec(R), NB, NH, "cpu").eval()
DT = int(os.environ.get("FPC_NN_DTYPE", "1"))   # 1 = fp16 (the headline operand type), 0 = bf16
eng = fpc_ffi.Engine(R, 3, max_games=G, max_sims=8, nn_dtype=DT)
eng.load_weights(weights.export_weights(m, DT))
x = (torch.rand(G, 24, R, R) < 0.1).float().cuda()
lg = torch.empty(G, eng.A, device="cuda"); va = torch.empty(G, device="cuda")
import time
eng.nn_forward(x.data_ptr(), G, lg.data_ptr(), va.data_ptr())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    eng.nn_forward(x.data_ptr(), G, lg.data_ptr(), va.data_ptr())
torch.cuda.synchronize()
print("done", float(lg.abs().mean()), "ms/forward %.4f" % ((time.perf_counter() - t0) / iters * 1e3))
