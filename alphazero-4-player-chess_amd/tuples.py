"""Training tuples (state, pi, z) and their episode-end exchange between the per-GPU game shards.

The reference keeps tuples as Python objects in one process (alphazero.py:53-78: Board by value,
dense pi tensor [A], reward).  Here games are sharded game g -> rank g mod N and each
rank all-gathers its compact records once per episode over RCCL (backend "nccl" on ROCm) or gloo:

    record = mailbox R*R bytes | turn u8 | n u16 | z f32 | n x (flat u16, visits u16)

Dense reference-shaped tensors (encoded state [24,R,R] via the engine, pi [A] = visits/sum) are
rebuilt on receipt (`dense_pi`).  Variable length => all-gather of byte counts, then one padded
all-gather of the payload (latency-bound: a single fused collective per episode)."""
import struct

import numpy as np
import torch
import torch.distributed as dist


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
