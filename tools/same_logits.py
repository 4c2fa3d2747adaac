#!/usr/bin/env python3
"""Developer check: the network forward gives bit-identical logits / values under two settings of an
environment knob (e.g. FPC_TOWER_WAVES=4 vs 8: same MFMAs, same order, other wave decomposition).
    python3 tools/same_logits.py FPC_TOWER_WAVES 4 8 [board] [blocks] [hidden]"""
import os, subprocess, sys
os.environ["FPC_DEV_KNOBS"] = "1"      # the engine reads its developer knobs only with this set
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path[:0] = [os.path.join(HERE, "alphazero-4-player-chess_amd"), HERE]
    import numpy as np, torch
    import fpc_ffi, net, weights
    from bench import Spec
    R, blocks, hidden, out = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    INV = {8: 2, 10: 2, 13: 3, 14: 3}[R]
    res = {}
    for dt in (1, 0):
        torch.manual_seed(5)
        m = net.ResNet(Spec(R), blocks, hidden, "cpu").eval()
        G = 64
        eng = fpc_ffi.Engine(R, INV, max_games=G, max_sims=8, nn_dtype=dt)
        eng.load_weights(weights.export_weights(m, dt))
        x = (torch.rand(G, 24, R, R, generator=torch.Generator().manual_seed(7)) < 0.1).float().cuda()
        lg = torch.empty(G, eng.A, device="cuda"); va = torch.empty(G, device="cuda")
        eng.nn_forward(x.data_ptr(), G, lg.data_ptr(), va.data_ptr())
        torch.cuda.synchronize()
        res["lg%d" % dt] = lg.cpu().numpy(); res["va%d" % dt] = va.cpu().numpy()
        eng.close()
    np.savez(out, **res)
    sys.exit(0)
import numpy as np
knob, a, b = sys.argv[1:4]
shape = sys.argv[4:7] if len(sys.argv) >= 7 else ["14", "10", "128"]
outs = []
for v in (a, b):
    out = "/tmp/same_logits_%s_%s.npz" % (knob, os.path.basename(v))
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", *shape, out], env=dict(os.environ, **{knob: v}))
    outs.append(np.load(out))
ok = True
for k in outs[0].files:
    same = np.array_equal(outs[0][k], outs[1][k])
    print(k, "identical" if same else "DIFFER max|d| = %g" % np.abs(outs[0][k] - outs[1][k]).max(), "mean|x| = %.4f" % np.abs(outs[0][k]).mean())
    ok &= same
sys.exit(0 if ok else 1)
