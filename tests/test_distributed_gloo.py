"""N>1 path on CPU: world_size-2 gloo run of the game sharding + episode-end tuple all-gather
(the same code runs over RCCL on the GPUs; bench.py --gpus N)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import tuples


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _records_for(rank, world, n_games, R):
    rng = np.random.default_rng(100)       # same stream on every rank: deterministic global data set
    recs = {}
    for g in range(n_games):
        n = int(rng.integers(1, 40))
        mailbox = rng.integers(0, 256, R * R, dtype=np.uint8)
        flats = np.sort(rng.choice((8 * R + 8) * R * R, n, replace=False))
        visits = rng.integers(1, 400, n)
        recs[g] = tuples.pack_record(R, mailbox, g % 4, (-1.0) ** g, flats, visits)
    mine = tuples.shard_games(n_games, rank, world)
    return recs, mine


def _worker(rank, world, port, n_games, R, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    recs, mine = _records_for(rank, world, n_games, R)
    payload = b"".join(recs[g] for g in mine)
    got = tuples.all_gather_bytes(payload)
    ok = len(got) == world
    for r in range(world):
        exp = b"".join(recs[g] for g in tuples.shard_games(n_games, r, world))
        ok &= got[r] == exp
    allrecs = [x for r in range(world) for x in tuples.unpack_records(R, got[r])]
    ok &= len(allrecs) == n_games
    A = (8 * R + 8) * R * R
    pi = tuples.dense_pi(allrecs[0], A)
    ok &= abs(float(pi.sum()) - 1.0) < 1e-6 and int((pi > 0).sum()) == len(allrecs[0]["flat"])
    # max-over-ranks timing + whole-job throughput reduction used by bench.py
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([10.0 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    ok &= float(t) == float(world) and float(s) == 10.0 * world * (world + 1) / 2
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_allgather_world2():
    world, n_games, R = 2, 9, 14
    assert tuples.shard_games(n_games, 0, world) == [0, 2, 4, 6, 8] and tuples.shard_games(n_games, 1, world) == [1, 3, 5, 7]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_games, R, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_record_roundtrip():
    R = 8
    rec = tuples.pack_record(R, np.arange(64, dtype=np.uint8), 3, -1.0, [5, 77, 4000], [9, 8, 300])
    out = tuples.unpack_records(R, rec + rec)
    assert len(out) == 2 and out[1]["turn"] == 3 and out[1]["z"] == -1.0
    assert out[0]["flat"].tolist() == [5, 77, 4000] and out[0]["visits"].tolist() == [9, 8, 300]


# ---- the same exchange on REAL engine output -----------------------------------------------------
def _shard_episode(eng, backend, R, game_ids, plies, sims):
    """a few plies of self-play on the engine for the games `game_ids` (one batch, reference order);
    tuples are collected natively (fpc_collect_tuples) and z assigned at the end (fpc_tuples_set_z)"""
    import evaluators
    import fpc_ffi
    import positions
    import selfplay
    from fpc_testlib import run_external_search
    turn, entries = positions.start_entries(R)
    boards = [fpc_ffi.board_from_dict(R, turn, entries, _lib=eng.L) for _ in game_ids]
    ev = evaluators.make("hash", R)
    eng.tuples_reserve(len(game_ids) * plies)
    for ply in range(plies):
        res = run_external_search(eng, backend, boards, sims, 3.0, ev)
        eng.collect_tuples(game_ids, ply)
        picks = []
        for i, g in enumerate(game_ids):
            n = int(res["n_children"][i])
            picks.append(selfplay.sample_action(res["flat"][i, :n], res["visits"][i, :n], 1.1, ((g * 7 + ply * 3) % 10) / 10.0))
        boards = eng.take_action(boards, picks)
    # reward rule of the reference's max-length scoring (alphazero.py:161-175): +-heuristic by team
    z0, z1 = [], []
    for b in boards:
        h = eng.L.fpc_board_heuristic(b, b.turn & 1) * 0.02
        z0.append(h if (b.turn & 1) == 0 else -h)
        z1.append(h if (b.turn & 1) == 1 else -h)
    eng.tuples_set_z(game_ids, z0, z1)


def _engine_worker(rank, world, port, n_games, R, plies, sims, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fpc_testlib import make_engine
    mine = tuples.shard_games(n_games, rank, world)
    eng = make_engine("emul", R, 2, max_games=len(mine), max_sims=sims)
    _shard_episode(eng, "emul", R, mine, plies, sims)
    recs = tuples.exchange(eng)                      # gloo transport of the native PODs
    enc, pi, z = tuples.dense_batch(eng, recs)
    key = sorted(range(len(recs)), key=lambda i: (recs[i]["game"], recs[i]["ply"]))
    q.put((rank, [(recs[i]["game"], recs[i]["ply"], recs[i]["turn"], recs[i]["mailbox"].tobytes(), recs[i]["flat"].tolist(),
                   recs[i]["visits"].tolist(), recs[i]["z"]) for i in key],
           enc[key].numpy().tobytes(), pi[key].numpy().tobytes(), z[key].numpy().tobytes()))
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def test_world_size_2_exchange_of_real_engine_tuples():
    """games sharded g -> rank g mod N, every rank searches its shard on its own engine (the
    wavefront-emulator build of the product's kernels here), tuples are built by fpc_collect_tuples and
    all-gathered at episode end; the dense (state, pi, z) reconstructed on EVERY rank must equal what
    one process gets when it runs the same shards itself."""
    from fpc_testlib import make_engine
    R, n_games, plies, sims, world = 8, 4, 3, 12, 2
    # single-process reference: the same shards (same batch composition, quirk Q6), one engine
    ref = []
    for r in range(world):
        mine = tuples.shard_games(n_games, r, world)
        eng = make_engine("emul", R, 2, max_games=len(mine), max_sims=sims)
        _shard_episode(eng, "emul", R, mine, plies, sims)
        arr, n = eng.tuples_read()
        ref += tuples.records_of(arr, n, R)
        eng_last = eng
    assert len(ref) == n_games * plies
    enc, pi, z = tuples.dense_batch(eng_last, ref)
    key = sorted(range(len(ref)), key=lambda i: (ref[i]["game"], ref[i]["ply"]))
    want = ([(ref[i]["game"], ref[i]["ply"], ref[i]["turn"], ref[i]["mailbox"].tobytes(), ref[i]["flat"].tolist(),
              ref[i]["visits"].tolist(), ref[i]["z"]) for i in key],
            enc[key].numpy().tobytes(), pi[key].numpy().tobytes(), z[key].numpy().tobytes())
    assert abs(float(pi.sum(dim=1).min()) - 1.0) < 1e-6 and any(t[6] != 0.0 for t in want[0])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_engine_worker, args=(r, world, port, n_games, R, plies, sims, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    for rank, recs, e, p_, z_ in got:
        assert recs == want[0], rank
        assert e == want[1] and p_ == want[2] and z_ == want[3], rank


# ---- init_comm: every rank leaves with the same verdict, whatever fails where ----------
class _FakeLib:
    """the RCCL entry points of the C-ABI with a failure injected on chosen ranks"""

    def __init__(self, rank, fail_available=(), fail_id=False, fail_init=()):
        self.rank, self.fa, self.fid, self.fi = rank, fail_available, fail_id, fail_init
        self.calls = []

    def fpc_comm_available(self):
        self.calls.append("available")
        return -11 if self.rank in self.fa else 0

    def fpc_last_error(self, _h):
        return b"injected failure"

    def fpc_comm_unique_id(self, buf):
        self.calls.append("unique_id")
        if self.fid:
            return -11
        for i in range(128):
            buf[i] = bytes([i])
        return 0

    def fpc_comm_destroy(self, _h):
        self.calls.append("destroy")
        return 0


class _FakeEngine:
    def __init__(self, lib):
        self.L, self.h, self.has_comm = lib, None, False

    def comm_init(self, id128, rank, world):
        self.L.calls.append("init")
        assert bytes(id128) == bytes(range(128))
        if rank in self.L.fi:
            raise RuntimeError("injected ncclCommInitRank failure")


def _comm_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    scenarios = {"all good": {}, "library missing on rank 1": {"fail_available": (1,)}, "id fails on rank 0": {"fail_id": True},
                 "init fails on rank 1": {"fail_init": (1,)}, "init fails on rank 0": {"fail_init": (0,)}}
    for name, kw in scenarios.items():
        lib = _FakeLib(rank, **kw)
        eng = _FakeEngine(lib)
        ok, why = tuples.init_comm(eng)
        out[name] = (ok, eng.has_comm, list(lib.calls), why)
    # the real C-ABI on the wavefront-emulator build: RCCL does not exist there -> every rank says no, then
    # the exchange still works through torch.distributed
    from fpc_testlib import make_engine
    eng = make_engine("emul", 8, 2, max_games=1, max_sims=4)
    ok, why = tuples.init_comm(eng)
    out["emulator build"] = (ok, getattr(eng, "has_comm", False), [], why)
    eng.close()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_init_comm_gives_every_rank_the_same_verdict():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))     # a mismatched collective would hang here
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for name in got[0]:
        oks = [got[r][name][0] for r in range(world)]
        has = [got[r][name][1] for r in range(world)]
        assert oks == has, name
        assert len(set(oks)) == 1, (name, oks)
        assert oks[0] == (name == "all good"), name
    # nobody enters ncclCommInitRank unless every rank can load the library and rank 0 has an id
    for name in ("library missing on rank 1", "id fails on rank 0"):
        assert all("init" not in got[r][name][2] for r in range(world)), name
    # a rank whose own init succeeded gives its communicator up when another rank's failed
    assert got[0]["init fails on rank 1"][2][-1] == "destroy" and got[1]["init fails on rank 0"][2][-1] == "destroy"
    assert "rank 1" in got[1]["init fails on rank 1"][3] and got[0]["init fails on rank 1"][3]
    assert "RCCL exists only in the gfx950 build" in got[0]["emulator build"][3]
