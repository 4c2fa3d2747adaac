#!/usr/bin/env python3
"""Diagnostic build only (hipcc ... -DTWW_STAMPS=<conv layer index, even> -o tools/lib_stamps.so): timeline of one
conv layer of k_towerw, blocks 0 and 131, every wave -- s_memtime ticks (= shader cycles on gfx950, about 2.0 GHz under this load).
    FPC_ENGINE_LIB=$PWD/tools/lib_stamps.so FPC_NN_BLOCKS=20 FPC_NN_HIDDEN=256 python3 tools/towerw_stamps.py"""
import os, subprocess, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/tww_stamps.txt"
env = dict(os.environ, FPC_TW_STAMPS_FILE=out)
subprocess.check_call([sys.executable, os.path.join(HERE, "tools", "nn_only.py"), "3"], env=env)
rows = [[int(x) for x in l.split()] for l in open(out)]
names = ["k0"] + ["tap%d>" % t for t in range(1, 9)] + ["taps.", "barrier", "lastk", "epilog", "barrier", "load_b"]
for blk in range(2):
    rs = rows[blk * 8:(blk + 1) * 8]
    if not any(any(r) for r in rs):
        continue
    t0 = min(r[0] for r in rs if r[0])
    print("block %d: ticks since the first wave entered the layer; then the step from the previous point" % (0 if blk == 0 else 131))
    print("wave " + " ".join("%7s" % n for n in names))
    for w, r in enumerate(rs):
        print("w%d   " % w + " ".join("%7d" % (r[i] - t0) for i in range(15)))
        print("     " + " ".join("%7s" % ("" if i == 0 else "+%d" % (r[i] - r[i - 1])) for i in range(15)))
