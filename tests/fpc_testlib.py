"""Shared helpers for the engine parity tests (CPU wavefront-emulator build and real GPU build)."""
import ctypes as C
import gzip
import json
import os
import subprocess
import sys

import numpy as np

TESTS = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(TESTS)
PKG = os.path.join(REPO, "alphazero-4-player-chess_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

import fpc_ffi  # noqa: E402

GOLD = os.path.join(TESTS, "golden")
_gold_cache = {}


def gold(R):
    if R not in _gold_cache:
        with gzip.open(os.path.join(GOLD, "ref_r%d.json.gz" % R), "rt") as f:
            _gold_cache[R] = json.load(f)
    return _gold_cache[R]


_emul = None


def emul_lib():
    """tests/emul/libfpc_emul.so: the product's tree-kernel source on the wavefront emulator."""
    global _emul
    if _emul is None:
        san = os.environ.get("FPC_SAN") == "1"       # tools/run_sanitized.sh: the ASan + UBSan build of the same sources
        subprocess.check_call(["make", "-s", "-C", os.path.join(TESTS, "emul")] + (["SAN=1"] if san else []))
        _emul = fpc_ffi.bind(C.CDLL(os.path.join(TESTS, "emul", "libfpc_emul_san.so" if san else "libfpc_emul.so")))
    return _emul


def make_engine(backend, R, INV, **kw):
    if backend == "emul":
        return fpc_ffi.Engine(R, INV, _lib=emul_lib(), **kw)
    return fpc_ffi.Engine(R, INV, **kw)


class DevPtr:
    def __init__(self, ptr, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


def run_external_search(eng, backend, roots, sims, c_puct, evaluator, fused=False):
    """Drives fpc_search_select / evaluator / fpc_search_expand exactly like mcts.py:36-38 does.
    evaluator: numpy callable(enc[G,24,R,R]) -> (logits[G,A], value[G]).
    fused: between two evaluations use fpc_search_expand_select (one launch) instead of expand + select."""
    G, R, A = len(roots), eng.R, eng.A
    eng.search_begin(roots, c_puct)
    keep = []
    n_live, enc_ptr = eng.search_select() if sims > 0 else (0, None)
    for i in range(sims):
        last = i == sims - 1
        if n_live == 0:
            if not last:
                n_live, enc_ptr = eng.search_select()
            continue
        if backend == "emul":
            enc = np.ctypeslib.as_array(C.cast(enc_ptr, C.POINTER(C.c_float)), shape=(G, 24, R, R))
            lg, v = evaluator(enc)
            lg = np.ascontiguousarray(lg, dtype=np.float32)
            v = np.ascontiguousarray(v, dtype=np.float32)
            keep = [lg, v]
            lp, vp = lg.ctypes.data, v.ctypes.data
        else:
            import torch
            enc_t = torch.as_tensor(DevPtr(enc_ptr, (G, 24, R, R)), device="cuda")
            enc = enc_t.cpu().numpy()
            lg, v = evaluator(enc)
            lg_t = torch.from_numpy(np.ascontiguousarray(lg, dtype=np.float32)).cuda()
            v_t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda()
            torch.cuda.synchronize()
            keep = [lg_t, v_t]
            lp, vp = lg_t.data_ptr(), v_t.data_ptr()
        if fused and not last:
            n_live, enc_ptr = eng.search_expand_select(lp, vp)
        else:
            eng.search_expand(lp, vp)
            if not last:
                n_live, enc_ptr = eng.search_select()
        if backend != "emul":
            import torch
            torch.cuda.synchronize()
    del keep
    return eng.search_results(roots=roots)


def expand_promos(moves):
    """device list (promotion collapsed, flag set) -> reference list with the 4 N,B,R,Q duplicates."""
    out = []
    for frm, to, flat, promo, _cap in moves:
        out.extend([[frm, to, flat]] * (4 if promo else 1))
    return out
