"""Training tuples (state, pi, z) and their episode-end exchange between the per-GPU game shards.

The reference keeps tuples as Python objects in one process (alphazero.py:53-78: Board by value,
dense pi tensor [A], reward).  Here games are sharded game g -> rank g mod N and each
rank all-gathers its compact records once per episode over RCCL (backend "nccl" on ROCm) or gloo:

    record = mailbox R*R bytes | turn u8 | n u16 | z f32 | n x (flat u16, visits u16)

Dense reference-shaped tensors (encoded state [24,R,R] via the engine, pi [A] = visits/sum) are
rebuilt on receipt (`dense_pi`).  Variable length => all-gather of byte counts, then one padded
all-gather of the payload (latency-bound: a single fused collective per episode)."""
import ctypes as C
import struct

import numpy as np
import torch
import torch.distributed as dist

import fpc_ffi


def shard_games(n_games, rank, world):
    """indices of the games owned by `rank` (game g -> rank g mod world)"""
    return list(range(rank, n_games, world))


def pack_record(R, mailbox, turn, z, flats, visits):
    n = len(flats)
    head = bytes(mailbox[:R * R]) + struct.pack("<BHf", turn, n, float(z))
    body = np.stack([np.asarray(flats, np.uint16), np.asarray(visits, np.uint16)], axis=1).tobytes() if n else b""
    return head + body


def unpack_records(R, buf):
    out, off, RR = [], 0, R * R
    buf = bytes(buf)
    while off < len(buf):
        mailbox = np.frombuffer(buf, np.uint8, RR, off)
        turn, n, z = struct.unpack_from("<BHf", buf, off + RR)
        off += RR + 7
        fv = np.frombuffer(buf, np.uint16, 2 * n, off).reshape(n, 2)
        off += 4 * n
        out.append({"mailbox": mailbox, "turn": turn, "z": z, "flat": fv[:, 0].astype(np.int64), "visits": fv[:, 1].astype(np.int64)})
    return out


def dense_pi(rec, A):
    """alphazero.py:104-110: action_probs[flat] = child visit count; /= sum"""
    pi = torch.zeros(A, dtype=torch.float32)
    pi[torch.from_numpy(rec["flat"])] = torch.from_numpy(rec["visits"]).to(torch.float32)
    return pi / pi.sum()


def all_gather_bytes(payload, device="cpu", group=None):
    """payload: bytes of this rank.  Returns the list of every rank's bytes, in rank order."""
    world = dist.get_world_size(group)
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    mine = torch.zeros(mx, dtype=torch.uint8, device=device)
    if payload:
        mine[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine, group=group)
    return [bytes(b[:s].cpu().numpy().tobytes()) for b, s in zip(bufs, sizes)]


# ---- native tuples (fpc_tuple PODs built on the device by fpc_collect_tuples) -------------------
def records_of(arr, n, R):
    """ctypes fpc_tuple array -> list of record dicts (same keys as unpack_records + game, ply)"""
    out = []
    for i in range(n):
        t = arr[i]
        k = int(t.n)
        out.append({"mailbox": np.frombuffer(bytes(t.sq), np.uint8, R * R).copy(), "turn": int(t.turn), "z": float(t.z),
                    "flat": np.asarray(t.flat[:k], np.int64), "visits": np.asarray(t.visits[:k], np.int64),
                    "game": int(t.game), "ply": int(t.ply)})
    return out


def exchange(eng, group=None):
    """Episode-end all-gather of the tuples collected on this rank's engine.  Returns the
    records of ALL ranks, rank-major.  With an RCCL communicator on the engine (init_comm; bench.py
    --gpus N) the collective is issued by the C++ host on device memory; otherwise (gloo, CPU tests)
    the same PODs travel through torch.distributed as bytes."""
    return parse_raw(eng, exchange_raw(eng, group))


def exchange_raw(eng, group=None):
    """The collective of `exchange` alone (what bench.py times): the gathered tuples stay where the
    collective left them (device memory for the RCCL path, byte blobs for the torch.distributed one;
    nothing at all in a single process).  `parse_raw` turns the result into records."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return {"kind": "local", "ranks": 1}
    if getattr(eng, "has_comm", False):
        counts, total = eng.allgather_tuples_device()
        return {"kind": "rccl", "ranks": eng.comm_world, "counts": [int(counts[i]) for i in range(eng.comm_world)], "total": total}
    arr, n = eng.tuples_read()
    payload = bytes(memoryview(arr).cast("B")[:n * C.sizeof(fpc_ffi.Tuple)]) if n else b""
    blobs = all_gather_bytes(payload, group=group)
    return {"kind": "torch", "ranks": len(blobs), "blobs": blobs}


def parse_raw(eng, raw):
    if raw["kind"] == "local":
        arr, n = eng.tuples_read()
        return records_of(arr, n, eng.R)
    if raw["kind"] == "rccl":
        return records_of(eng.gathered_read(raw["total"]), raw["total"], eng.R)
    out = []
    for blob in raw["blobs"]:
        m = len(blob) // C.sizeof(fpc_ffi.Tuple)
        if m:
            out += records_of((fpc_ffi.Tuple * m).from_buffer_copy(blob), m, eng.R)
    return out


def init_comm(eng, device=None, group=None):
    """RCCL communicator for `eng`; the 128-byte id travels over the existing torch.distributed group.

    Collective-safe: every rank runs the same sequence of torch.distributed collectives whatever fails
    where, and all ranks leave with the same answer.  Returns (True, "") when EVERY rank holds a
    communicator; otherwise no rank keeps one (fpc_comm_destroy everywhere) and the result is
    (False, reason) on every rank -- the caller then either stops (bench.py --backend nccl) or uses the
    torch.distributed path of `exchange`.
      1. every rank probes librccl (fpc_comm_available) and the verdicts are MIN-reduced, so nobody
         enters ncclCommInitRank -- which blocks until all ranks arrive -- unless everybody can;
      2. rank 0 makes the id; the broadcast carries a status byte in front of it and always runs;
      3. after fpc_comm_init a success flag is MIN-reduced."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = device if device is not None else "cpu"
    L = eng.L

    def all_ok(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return bool(int(t.item()))

    why = ""
    ok = L.fpc_comm_available() == 0
    if not ok:
        why = "rank %d: %s" % (rank, (L.fpc_last_error(None) or b"").decode())
    if not all_ok(ok):
        return False, why or "librccl is not available on another rank"
    msg = torch.zeros(129, dtype=torch.uint8)
    if rank == 0:
        try:
            msg[1:] = torch.frombuffer(bytearray(fpc_ffi.comm_unique_id(L)), dtype=torch.uint8)
            msg[0] = 1
        except RuntimeError as exc:
            why = "rank 0: %s" % (exc,)
    msg = msg.to(dev)
    dist.broadcast(msg, 0, group=group)
    msg = msg.cpu()
    if int(msg[0]) != 1:
        return False, why or "rank 0 could not create the ncclUniqueId"
    ok = True
    try:
        eng.comm_init(bytes(msg[1:].numpy().tobytes()), rank, world)
    except RuntimeError as exc:
        ok, why = False, "rank %d: %s" % (rank, exc)
    if not all_ok(ok):
        L.fpc_comm_destroy(eng.h)
        eng.has_comm = False
        return False, why or "fpc_comm_init failed on another rank"
    eng.has_comm = True
    return True, ""


def dense_batch(eng, recs):
    """(encoded state [n,24,R,R] f32, pi [n,A] f32, z [n,1] f32) as the reference's trainer stacks them
    (alphazero.py:186-197): GetEncodedState per tuple (own rotation), pi = N / sum N"""
    boards = []
    for r in recs:
        b = fpc_ffi.Board()
        for i, v in enumerate(r["mailbox"]):
            b.sq[i] = int(v)
        b.turn = r["turn"]
        for c in range(4):
            b.king[c] = fpc_ffi.NO_SQ
        boards.append(b)
    enc = np.concatenate([eng.encode([b]) for b in boards]) if boards else np.zeros((0, 24, eng.R, eng.R), np.float32)
    pi = torch.stack([dense_pi(r, eng.A) for r in recs]) if recs else torch.zeros(0, eng.A)
    z = torch.tensor([r["z"] for r in recs], dtype=torch.float32).view(-1, 1)
    return torch.from_numpy(enc), pi, z
