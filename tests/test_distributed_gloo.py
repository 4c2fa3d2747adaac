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
