"""Synthetic evaluators shared by the golden generator (oracle/gen_golden.py, torch form) and the
tests (numpy form here).  All produce exactly representable f32 logits/values.

evaluator(enc[B,24,R,R] f32) -> (logits[B,A] f32, value[B] f32)
"""
import numpy as np


def make(kind, R):
    RR = R * R
    A = (8 * R + 8) * RR
    w11 = (np.arange(24 * RR) % 11).astype(np.float64).reshape(1, 24, R, R)
    widx = ((np.arange(24 * RR, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(1 << 32)).reshape(1, 24, R, R)
    i = np.arange(A, dtype=np.uint64).reshape(1, A)
    iterm = (i * np.uint64(40503) + ((i * i) % np.uint64(8191)) * np.uint64(69069))

    def ev(enc):
        B = enc.shape[0]
        if kind == "zero":
            return np.zeros((B, A), np.float32), np.zeros(B, np.float32)
        if kind == "ramp":
            logits = np.tile((-(np.arange(A) % 7).astype(np.float32) / np.float32(8)), (B, 1))
            s = (enc.astype(np.float64) * w11).sum(axis=(1, 2, 3))
            v = ((np.mod(s, 5) - 2) / 4).astype(np.float32)
            return logits, v
        h = (enc.astype(np.uint64) * widx).sum(axis=(1, 2, 3)) % np.uint64(1 << 32)
        u = ((h.reshape(B, 1) * np.uint64(2246822519) + iterm) % np.uint64(1 << 32)) >> np.uint64(16)
        if kind == "hash":
            logits = u.astype(np.float32) / np.float32(8192.0) - np.float32(4.0)
        elif kind == "hashinf":
            logits = np.where((u % np.uint64(4)) == 0, np.float32(-np.inf), np.float32(0)).astype(np.float32)
        elif kind == "hashinf1":     # sparse -inf (1 logit in 32) among hash logits: searches that run to the end with -inf in the softmax
            logits = np.where((u % np.uint64(32)) == 0, np.float32(-np.inf), u.astype(np.float32) / np.float32(8192.0) - np.float32(4.0)).astype(np.float32)
        else:
            raise ValueError(kind)
        v = ((h % np.uint64(9)).astype(np.float32) - 4) / 4
        return logits, v.astype(np.float32)

    return ev
