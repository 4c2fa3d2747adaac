"""The engine's own episode-end exchange (C++ host: fpc_comm_init + fpc_allgather_tuples = counts and capacities ->
buffer growth -> status round -> max-padded payload) with TWO ranks on a one-GPU box.  RCCL itself refuses two ranks on
one GPU, so the five librccl entry points are stood in for by tests/emul/libfile_collective.so (files in a shared
directory; FPC_RCCL_LIB names it, exactly the way the product names the real library): what is tested is the engine's
protocol around the collectives -- unequal tuple counts, padding, a send buffer that one rank has to grow and the other
does not (no rank may be left alone in the payload collective), the capacity word that makes the second
exchange skip the status round -- not RCCL."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

TESTS = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_exchange_their_tuples_through_the_engines_own_protocol(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(TESTS, "emul"), "file_collective"])
    env = dict(os.environ, FPC_RCCL_LIB=os.path.join(TESTS, "emul", "libfile_collective.so"), FILE_COLLECTIVE_DIR=str(tmp_path))
    env.pop("FPC_ENGINE_LIB", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(TESTS, "comm_rank_script.py"), str(r), "2", str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), outs
    local = [open(tmp_path / ("local_%d.bin" % r), "rb").read() for r in range(2)]
    assert len(local[0]) == 12 * 1280 and len(local[1]) == 2 * 1280
    for rep in range(2):
        for r in range(2):
            assert open(tmp_path / ("counts_%d_%d.txt" % (r, rep))).read().split() == ["12", "2"]
            got = open(tmp_path / ("gathered_%d_%d.bin" % (r, rep)), "rb").read()
            assert got == local[0] + local[1], (rep, r)          # rank-major, padding stripped, identical on both ranks
    # the sequence of collectives each rank went through: exchange 1 = counts, status (rank 1 grows its send buffer, both
    # grow their receive buffers), payload; exchange 2 = counts, payload -- nobody has to grow anything any more
    for r in range(2):
        seqs = sorted(int(f.split("_")[0]) for f in os.listdir(tmp_path) if f.endswith("_%d.bin" % r) and f[0].isdigit())
        assert seqs == [0, 1, 2, 3, 4], (r, seqs)
