#!/usr/bin/env python3
"""Diagnostic build only (hipcc ... -DTW_STAMPS=<tap index> -o tools/var/lib_stamps.so): timeline of two
consecutive taps of k_tower, block 0, every wave -- s_memtime (100 MHz-independent shader clock ticks)
relative to the earliest stamp.  FPC_TOWER_WAVES=4|8 picks the kernel form.
    FPC_ENGINE_LIB=$PWD/tools/var/lib_stamps.so python3 tools/tower_stamps.py"""
import os, subprocess, sys
os.environ["FPC_DEV_KNOBS"] = "1"      # the engine reads its developer knobs only with this set
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/tw_stamps.txt"
env = dict(os.environ, FPC_TW_STAMPS_FILE=out)
subprocess.check_call([sys.executable, os.path.join(HERE, "tools", "nn_only.py"), "3"], env=env)
rows = [[int(x) for x in l.split()] for l in open(out)]
t0 = min(v for r in rows for v in r if v)
names = ["k1>", "k2>", "arrive", "vmcnt", "barrier", "dma", "settap", "k3", "k0'"]
print("wave " + " ".join("%8s" % n for n in names) + "   (ticks since the first stamp; columns = time the point was REACHED)")
for tap in range(2):
    for w in range(8):
        r = rows[tap * 8 + w]
        if not any(r):
            continue
        print("t%d w%d " % (tap, w) + " ".join("%8d" % (r[i] - t0 if r[i] else -1) for i in range(9)))
