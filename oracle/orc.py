"""oracle/orc.py -- TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/liboracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_SQ, MAX_PL, NO_SQ = 196, 32, 255


class Board(C.Structure):
    _fields_ = [("sq", C.c_uint8 * MAX_SQ), ("pl", (C.c_uint8 * MAX_PL) * 4), ("plen", C.c_uint8 * 4),
                ("castle", C.c_uint8 * 4), ("king", C.c_uint8 * 4), ("turn", C.c_uint8), ("pad", C.c_uint8 * 3)]


class Move(C.Structure):
    _fields_ = [("frm", C.c_uint8), ("to", C.c_uint8), ("capture", C.c_uint8), ("promo", C.c_uint8),
                ("rook_from", C.c_uint8), ("rook_to", C.c_uint8), ("init_rights", C.c_uint8),
                ("new_rights", C.c_uint8)]


class SearchOut(C.Structure):
    _fields_ = [("root_visits", C.c_int), ("n_children", C.c_int), ("terminated", C.c_int), ("sims_done", C.c_int)]


class EvalCtx(C.Structure):
    _fields_ = [("R", C.c_int)]


EVAL_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float))

_lib = None


SAN = os.environ.get("FPC_SAN") == "1"      # tools/run_sanitized.sh: the ASan + UBSan build of the restatement


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"] + (["SAN=1"] if SAN else []))


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle_san.so" if SAN else "liboracle.so")
        src = os.path.join(HERE, "fpc_oracle.cpp")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        L = C.CDLL(path)
        L.orc_expf.restype = C.c_float
        L.orc_expf.argtypes = [C.c_float]
        L.orc_search.argtypes = [C.POINTER(Board), C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p,
                                 C.c_void_p, C.POINTER(SearchOut), C.c_int, C.POINTER(C.c_int),
                                 C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)]
        L.orc_set_rules.argtypes = [C.c_int]
        L.orc_set_root_noise.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.orc_policy_priors.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int,
                                        C.POINTER(C.c_float)]
        _lib = L
    return _lib


def board_from_lists(R, turn, pl):
    """pl: per colour [[sq, type], ...] in piece-list order (the fixture format)."""
    b = Board()
    lib().orc_board_init(C.byref(b), R, turn)
    for colour, col in enumerate(pl):
        for sq, typ in col:
            assert lib().orc_board_add(C.byref(b), R, colour, typ, sq) == 0
    return b


def board_from_dict(R, turn, entries, castle=None):
    """entries: [[sq, colour, type], ...] in python-dict insertion order."""
    n = len(entries)
    sqs = (C.c_uint8 * n)(*[e[0] for e in entries])
    pcs = (C.c_uint8 * n)(*[0x80 | (e[1] << 5) | (e[2] << 2) for e in entries])
    b = Board()
    cs = (C.c_uint8 * 4)(*castle) if castle else None
    lib().orc_board_from_dict(C.byref(b), R, turn, sqs, pcs, n, cs)
    return b


def set_rules(rules):
    """N4: non-strict rule set, bits as FPC_RULES_* (0 = the reference)"""
    lib().orc_set_rules(int(rules))


_noise_keep = None


def set_root_noise(gamma, eps=0.0):
    """gamma: float32 [G, stride] Gamma(alpha) draws per root child (None: off)"""
    global _noise_keep
    import numpy as np
    if gamma is None:
        _noise_keep = None
        lib().orc_set_root_noise(None, 0, C.c_float(0.0))
        return
    _noise_keep = np.ascontiguousarray(gamma, dtype=np.float32)
    lib().orc_set_root_noise(_noise_keep.ctypes.data_as(C.c_void_p), _noise_keep.shape[1], C.c_float(float(eps)))


def lists_of(b):
    out = []
    for c in range(4):
        out.append([[b.pl[c][i], (b.sq[b.pl[c][i]] >> 2) & 7] for i in range(b.plen[c])])
    return out


def clone(b):
    nb = Board()
    C.memmove(C.byref(nb), C.byref(b), C.sizeof(Board))
    return nb


def legal_moves(b, R, INV):
    buf = (Move * 300)()
    n = lib().orc_legal_moves(C.byref(b), R, INV, buf, 300)
    return [[buf[i].frm, buf[i].to, lib().orc_move_flat(R, buf[i].frm, buf[i].to)] for i in range(n)]


def game_result(b, R, INV, player=-1):
    return lib().orc_game_result(C.byref(b), R, INV, player)


def take_action(b, R, flat):
    nb = clone(b)
    rc = lib().orc_take_action_flat(C.byref(nb), R, flat)
    return nb, rc


def attack_maps(b, R, INV):
    """[6][R*R] 0/1 maps: colours 0..3 (IsAttackedByPlayer), teams 0..1 (IsAttackedByTeam); every square of the array"""
    import numpy as np
    out = np.zeros((6, R * R), dtype=np.uint8)
    lib().orc_attack_maps(C.byref(b), R, INV, out.ctypes.data_as(C.c_void_p))
    return out


def is_attacked_by_player(b, R, sq, colour):
    return bool(lib().orc_is_attacked_by_player(C.byref(b), R, sq, colour))


def encode(boards, R):
    import numpy as np
    n = len(boards)
    arr = (Board * n)(*boards)
    out = np.zeros((n, 24, R, R), dtype=np.float32)
    lib().orc_encode(arr, n, R, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def search(boards, R, INV, sims, Cpuct, evaluator, max_children=256):
    """evaluator: 'zero' | 'ramp' | python callable(enc ndarray[B,24,R,R]) -> (logits[B,A], value[B])."""
    import numpy as np
    L = lib()
    G = len(boards)
    arr = (Board * G)(*boards)
    out = (SearchOut * G)()
    cf = np.zeros((G, max_children), dtype=np.int32)
    cv = np.zeros((G, max_children), dtype=np.int32)
    cp = np.zeros((G, max_children), dtype=np.float32)
    cw = np.zeros((G, max_children), dtype=np.float64)
    ctx = EvalCtx(R)
    A = L.orc_action_size(R)
    keep = []
    if evaluator == "zero":
        fn = C.cast(L.orc_eval_zero, C.c_void_p)
    elif evaluator == "ramp":
        fn = C.cast(L.orc_eval_ramp, C.c_void_p)
    else:
        def tramp(user, enc, B, logits, value):
            e = np.ctypeslib.as_array(enc, shape=(B, 24, R, R))
            lg, v = evaluator(e)
            np.ctypeslib.as_array(logits, shape=(B, A))[:] = np.asarray(lg, dtype=np.float32).reshape(B, A)
            np.ctypeslib.as_array(value, shape=(B,))[:] = np.asarray(v, dtype=np.float32).reshape(B)
        cb = EVAL_FN(tramp)
        keep.append(cb)
        fn = C.cast(cb, C.c_void_p)
    rc = L.orc_search(arr, G, R, INV, sims, float(Cpuct), fn, C.cast(C.byref(ctx), C.c_void_p), out, max_children,
                      cf.ctypes.data_as(C.POINTER(C.c_int)), cv.ctypes.data_as(C.POINTER(C.c_int)),
                      cp.ctypes.data_as(C.POINTER(C.c_float)), cw.ctypes.data_as(C.POINTER(C.c_double)))
    res = []
    for g in range(G):
        n = out[g].n_children
        res.append({"root_n": out[g].root_visits, "terminated": out[g].terminated, "sims_done": out[g].sims_done,
                    "children": [[int(cf[g, k]), int(cv[g, k])] for k in range(n)],
                    "priors": cp[g, :n].copy(), "w": cw[g, :n].copy(), "board": clone(arr[g])})
    return rc, res
