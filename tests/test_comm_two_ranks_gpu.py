"""The engine's own episode-end exchange (C++ host: fpc_comm_init + fpc_allgather_tuples = counts and capacities ->
buffer growth -> status round -> max-padded payload) with TWO ranks on a one-GPU box.  RCCL itself refuses two ranks on
one GPU, so the five librccl entry points are stood in for by tests/emul/libfile_collective.so (files in a shared
directory; FPC_RCCL_LIB names it, exactly the way the product names the real library): what is tested is the engine's
protocol around the collectives -- unequal tuple counts, padding, a send buffer that one rank has to grow and the other
does not (no rank may be left alone in the payload collective), the status round that runs in every exchange,
local HIP failures injected on one rank (fpc_debug_comm_fault) -- not RCCL."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

TESTS = os.path.dirname(os.path.abspath(__file__))


def _run_two_ranks(tmp_path, fault=""):
    subprocess.check_call(["make", "-s", "-C", os.path.join(TESTS, "emul"), "file_collective"])
    env = dict(os.environ, FPC_RCCL_LIB=os.path.join(TESTS, "emul", "libfile_collective.so"), FILE_COLLECTIVE_DIR=str(tmp_path))
    env.pop("FPC_ENGINE_LIB", None)
    if fault:
        env["FPC_TEST_COMM_FAULT"] = fault
    procs = [subprocess.Popen([sys.executable, os.path.join(TESTS, "comm_rank_script.py"), str(r), "2", str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)          # a stranded rank shows up here (the stand-in itself gives up after 60 s)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), outs
    local = [open(tmp_path / ("local_%d.bin" % r), "rb").read() for r in range(2)]
    assert len(local[0]) == 12 * 1280 and len(local[1]) == 2 * 1280
    return local


def _collectives(tmp_path, r):
    return sorted(int(f.split("_")[0]) for f in os.listdir(tmp_path) if f.endswith("_%d.bin" % r) and f[0].isdigit())


def test_two_ranks_exchange_their_tuples_through_the_engines_own_protocol(tmp_path):
    local = _run_two_ranks(tmp_path)
    for rep in range(2):
        for r in range(2):
            assert open(tmp_path / ("counts_%d_%d.txt" % (r, rep))).read().split() == ["12", "2"]
            got = open(tmp_path / ("gathered_%d_%d.bin" % (r, rep)), "rb").read()
            assert got == local[0] + local[1], (rep, r)          # rank-major, padding stripped, identical on both ranks
    # the sequence of collectives each rank went through: every exchange = counts, status, payload (rank 1 grows its send
    # buffer and both grow their receive buffers between the first two)
    for r in range(2):
        assert _collectives(tmp_path, r) == [0, 1, 2, 3, 4, 5], r


@pytest.mark.parametrize("point", [1, 2, 3, 4])
def test_a_local_hip_failure_on_one_rank_strands_nobody_and_the_communicator_survives(tmp_path, point):
    """rank 1's counts upload (1) / counts read-back (2) / send-buffer growth (3) / status upload (4) fails
    in the first exchange.  Both ranks return an error from it within the time limit -- rank 1 its own, rank 0 FPC_ECOMM
    naming rank 1 -- after exactly two collectives (counts, status: nobody entered the payload collective), and the next
    two exchanges on the same communicator deliver every tuple to both."""
    local = _run_two_ranks(tmp_path, fault="1:%d" % point)
    e0, e1 = (open(tmp_path / ("error_%d_0.txt" % r)).read() for r in range(2))
    assert "rank 1" in e0 and ("no rank entered" in e0 or "no fresh counts record" in e0), e0
    assert "failed" in e1 and "rank 1 reported" not in e1, e1
    for rep in (1, 2):
        for r in range(2):
            assert not os.path.exists(tmp_path / ("error_%d_%d.txt" % (r, rep)))
            assert open(tmp_path / ("counts_%d_%d.txt" % (r, rep))).read().split() == ["12", "2"]
            assert open(tmp_path / ("gathered_%d_%d.bin" % (r, rep)), "rb").read() == local[0] + local[1], (rep, r)
    for r in range(2):
        assert _collectives(tmp_path, r) == list(range(2 + 3 + 3)), (r, _collectives(tmp_path, r))


def test_a_rank_that_cannot_read_the_status_words_aborts_and_its_peer_gets_an_error(tmp_path):
    """The one window agreement cannot close: rank 1 said "good" and then cannot read what the others said.  It aborts its
    communicator (ncclCommAbort); rank 0, already waiting in the payload collective, comes back with FPC_ECOMM instead of
    waiting for ever."""
    _run_two_ranks(tmp_path, fault="1:5")
    e0, e1 = (open(tmp_path / ("error_%d_0.txt" % r)).read() for r in range(2))
    assert "status read-back failed" in e1 and "aborted" in e1, e1
    assert "ncclAllGather(tuples) failed" in e0 and "aborted" in e0, e0
    assert os.path.exists(tmp_path / "aborted")
