"""One rank of tests/test_comm_two_ranks_gpu.py (TEST INFRASTRUCTURE): builds tuples on the device with a few tiny
searches, then runs the engine's OWN episode-end exchange (fpc_comm_init + fpc_allgather_tuples, driven from the C++
host) against the file-based stand-in collective named by FPC_RCCL_LIB.  Usage: comm_rank_script.py <rank> <world> <dir>."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(os.path.dirname(HERE), "alphazero-4-player-chess_amd"), os.path.dirname(HERE), HERE]
rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]

import evaluators
import fpc_ffi
import positions
from fpc_testlib import run_external_search

R, sims = 8, 8
# rank 0: 3 plies x 4 games = 12 tuples in a roomy buffer; rank 1: 1 ply x 2 games = 2 tuples in a buffer of exactly 2
# -> the padded count (12) exceeds rank 1's send buffer: it has to grow it, rank 0 does not, and both must learn that
G, plies, cap = ((4, 3, 64), (2, 1, 2))[rank] if world == 2 else (3, 2, 6)
eng = fpc_ffi.Engine(R, 2, max_games=G, max_sims=sims)
turn, entries = positions.start_entries(R)
boards = [fpc_ffi.board_from_dict(R, turn, entries) for _ in range(G)]
ev = evaluators.make("hash", R)
eng.tuples_reserve(cap)
ids = [100 * rank + g for g in range(G)]
for ply in range(plies):
    res = run_external_search(eng, "gpu", boards, sims, 3.0, ev)
    eng.collect_tuples(ids, ply)
    boards = eng.take_action(boards, [int(res["flat"][g, ply % int(res["n_children"][g])]) for g in range(G)])
eng.tuples_set_z(ids, [0.25 * (rank + 1)] * G, [-0.25 * (rank + 1)] * G)
arr, n = eng.tuples_read()
open(os.path.join(d, "local_%d.bin" % rank), "wb").write(bytes(memoryview(arr).cast("B")[:n * 1280]))

idfile = os.path.join(d, "id.bin")
if rank == 0:
    uid = fpc_ffi.comm_unique_id()
    open(idfile + ".tmp", "wb").write(uid)
    os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 60, "rank 0 never published the id"
        time.sleep(0.01)
    uid = open(idfile, "rb").read()
eng.comm_init(uid, rank, world)
# FPC_TEST_COMM_FAULT = "<rank>:<point>": that rank makes one of the HIP calls of its FIRST exchange fail
# (fpc_debug_comm_fault).  Points 1-4 are routed through the status round: BOTH ranks must come back with an error from
# exchange 0 -- nobody waits in the payload collective -- and the communicator must still work for exchanges 1 and 2.
# Point 5 (the status words cannot be read back) ends with the communicator aborted on that rank and an error on the other.
fault = os.environ.get("FPC_TEST_COMM_FAULT", "")
frank, fpoint = (int(x) for x in fault.split(":")) if fault else (-1, 0)
if frank == rank:
    eng.debug_comm_fault(fpoint)
reps = 3 if fault else 2
for rep in range(reps):       # without a fault -- second exchange: every buffer is big enough now
    try:
        counts, garr, total = eng.allgather_tuples()
    except RuntimeError as ex:
        if not fault:
            raise
        open(os.path.join(d, "error_%d_%d.txt" % (rank, rep)), "w").write(str(ex))
        if fpoint == 5:
            break             # communicator gone on the faulting rank; the peer was told (ncclCommAbort)
        continue
    open(os.path.join(d, "gathered_%d_%d.bin" % (rank, rep)), "wb").write(bytes(memoryview(garr).cast("B")[:total * 1280]))
    open(os.path.join(d, "counts_%d_%d.txt" % (rank, rep)), "w").write(" ".join(str(int(c)) for c in counts))
eng.close()
print("rank %d ok: %d local tuples" % (rank, n))
