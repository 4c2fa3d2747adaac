"""ctypes binding of the C-ABI in include/fpc_engine.h (libfpc_engine.so, built by hipcc for gfx950).

There is no CPU implementation behind this module: if the HIP library is missing or no GPU is
visible, loading / Engine() raises.  Everything above this file (alphazero_cpp.py, mcts.py,
four_player_chess_board.py) is the Python mirror of the reference's binding surface.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# FPC_ENGINE_LIB: developer knob for A/B timing of two builds of the same engine (tools/ab.sh)
LIB_PATH = os.environ.get("FPC_ENGINE_LIB") or os.path.join(HERE, "csrc", "libfpc_engine.so")

MAX_SQ, MAX_PL, NO_SQ, MAX_MOVES = 196, 16, 255, 256
RULES_STRICT, RULES_PUCT, RULES_ROTATION, RULES_PLANES, RULES_FULL_MOVES, RULES_FIXED = 0, 1, 2, 4, 8, 15


class Board(C.Structure):
    """fpc_board: the 288-byte POD position (include/fpc_engine.h)."""
    _fields_ = [("sq", C.c_uint8 * MAX_SQ), ("pl", (C.c_uint8 * MAX_PL) * 4), ("plen", C.c_uint8 * 4),
                ("king", C.c_uint8 * 4), ("castle", C.c_uint8 * 4), ("turn", C.c_uint8), ("pad", C.c_uint8 * 15)]


assert C.sizeof(Board) == 288


class Move(C.Structure):
    _fields_ = [("frm", C.c_uint8), ("to", C.c_uint8), ("capture", C.c_uint8), ("promo", C.c_uint8),
                ("flat", C.c_uint16), ("pad", C.c_uint16)]


class Config(C.Structure):
    _fields_ = [("board_size", C.c_int), ("invalid_area", C.c_int), ("max_games", C.c_int), ("max_sims", C.c_int),
                ("avg_children", C.c_int), ("device", C.c_int), ("nn_dtype", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("ms_select", C.c_double), ("ms_tower", C.c_double), ("ms_fc", C.c_double), ("ms_expand", C.c_double),
                ("launches_select", C.c_uint64), ("launches_nn", C.c_uint64), ("launches_expand", C.c_uint64),
                ("sims", C.c_uint64), ("nodes", C.c_uint64)]


MAX_TUPLE_C = 256


class Tuple(C.Structure):
    """fpc_tuple: one (state, pi, z) training record, 1280 bytes (include/fpc_engine.h)."""
    _fields_ = [("sq", C.c_uint8 * MAX_SQ), ("turn", C.c_uint8), ("pad0", C.c_uint8), ("n", C.c_uint16), ("z", C.c_float),
                ("game", C.c_int32), ("ply", C.c_int32), ("flat", C.c_uint16 * MAX_TUPLE_C), ("visits", C.c_uint16 * MAX_TUPLE_C),
                ("pad1", C.c_uint8 * 44)]


assert C.sizeof(Tuple) == 1280

P = C.POINTER
BOARD_BYTES = C.sizeof(Board)
TURN_OFFSET = Board.turn.offset          # byte of a POD row that holds the side to move


def _bp(pods):
    """Board* over a C-contiguous [n, 288] uint8 array"""
    return C.cast(pods.ctypes.data, P(Board))


def pods_of(boards):
    """list of Board -> [n, 288] uint8 array (copy)"""
    out = np.zeros((len(boards), BOARD_BYTES), np.uint8)
    for i, b in enumerate(boards):
        C.memmove(out[i].ctypes.data, C.byref(b), BOARD_BYTES)
    return out


def board_of(row):
    """one [288] uint8 row -> Board (copy)"""
    b = Board()
    C.memmove(C.byref(b), row.ctypes.data, BOARD_BYTES)
    return b


class _LazyBoards:
    """search_results()["boards"]: Board objects made when indexed (a search returns hundreds of roots; most callers
    read none of them as objects)"""

    def __init__(self, pods):
        self._pods = pods

    def __len__(self):
        return self._pods.shape[0]

    def __getitem__(self, g):
        return board_of(self._pods[g])

    def __iter__(self):
        return (board_of(self._pods[g]) for g in range(self._pods.shape[0]))


_SIGS = {
    "fpc_abi_version": (C.c_int, []),
    "fpc_create": (C.c_int, [P(Config), P(C.c_void_p)]),
    "fpc_destroy": (None, [C.c_void_p]),
    "fpc_last_error": (C.c_char_p, [C.c_void_p]),
    "fpc_num_action_channels": (C.c_int, [C.c_int]),
    "fpc_action_space_size": (C.c_int, [C.c_int]),
    "fpc_is_legal_location": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "fpc_move_flat_index": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "fpc_flat_to_move": (C.c_int, [C.c_int, C.c_int, P(C.c_int), P(C.c_int)]),
    "fpc_board_from_dict": (C.c_int, [P(Board), C.c_int, C.c_int, P(C.c_uint8), P(C.c_uint8), C.c_int, P(C.c_uint8)]),
    "fpc_board_heuristic": (C.c_int, [P(Board), C.c_int]),
    "fpc_boards_legal_moves": (C.c_int, [C.c_void_p, P(Board), C.c_int, P(Move), P(C.c_int)]),
    "fpc_boards_game_result": (C.c_int, [C.c_void_p, P(Board), C.c_int, P(C.c_int), P(C.c_int)]),
    "fpc_boards_take_action": (C.c_int, [C.c_void_p, P(Board), P(C.c_int), C.c_int, P(Board)]),
    "fpc_boards_encode": (C.c_int, [C.c_void_p, P(Board), C.c_int, C.c_void_p]),
    "fpc_boards_legal_mask": (C.c_int, [C.c_void_p, P(Board), C.c_int, C.c_void_p]),
    "fpc_search_begin": (C.c_int, [C.c_void_p, P(Board), C.c_int, C.c_double]),
    "fpc_search_select": (C.c_int, [C.c_void_p, P(C.c_int), P(C.c_void_p)]),
    "fpc_search_expand": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "fpc_search_expand_select": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_int), P(C.c_void_p)]),
    "fpc_search_run": (C.c_int, [C.c_void_p, C.c_int]),
    "fpc_search_results": (C.c_int, [C.c_void_p, P(Board), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "fpc_search_grandchildren": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, P(C.c_int), C.c_void_p, C.c_void_p]),
    "fpc_load_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "fpc_nn_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "fpc_stats_get": (C.c_int, [C.c_void_p, P(Stats)]),
    "fpc_stats_reset": (C.c_int, [C.c_void_p]),
    "fpc_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "fpc_set_policy_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "fpc_stream": (C.c_void_p, [C.c_void_p]),
    "fpc_set_rules": (C.c_int, [C.c_void_p, C.c_int]),
    "fpc_search_set_root_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float]),
    "fpc_tuples_reserve": (C.c_int, [C.c_void_p, C.c_int]),
    "fpc_tuples_reset": (C.c_int, [C.c_void_p]),
    "fpc_collect_tuples": (C.c_int, [C.c_void_p, P(C.c_int), C.c_int]),
    "fpc_tuples_set_z": (C.c_int, [C.c_void_p, P(C.c_int), P(C.c_float), P(C.c_float), C.c_int]),
    "fpc_tuples_count": (C.c_int, [C.c_void_p]),
    "fpc_tuples_read": (C.c_int, [C.c_void_p, P(Tuple), C.c_int, C.c_int]),
    "fpc_comm_unique_id": (C.c_int, [C.c_void_p]),
    "fpc_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "fpc_comm_destroy": (C.c_int, [C.c_void_p]),
    "fpc_comm_available": (C.c_int, []),
    "fpc_nn_kernel": (C.c_char_p, [C.c_void_p]),
    "fpc_allgather_tuples": (C.c_int, [C.c_void_p, P(C.c_int), P(C.c_int)]),
    "fpc_debug_comm_fault": (C.c_int, [C.c_void_p, C.c_int]),
    "fpc_boards_attack_maps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "fpc_gathered_read": (C.c_int, [C.c_void_p, P(Tuple), C.c_int, C.c_int]),
}
EXPORTS = sorted(_SIGS)


def bind(cdll):
    for name, (res, args) in _SIGS.items():
        fn = getattr(cdll, name)     # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return cdll


_lib = None


def _check_fresh():
    """The in-tree library must have been built from the sources that lie beside it (__graft_entry__.build() stores
    their content hash): a stale engine must not be tested or measured.  Variant libraries named by FPC_ENGINE_LIB
    (A/B timing) are exempt."""
    if os.environ.get("FPC_ENGINE_LIB"):
        return
    try:
        want = open(LIB_PATH + ".srchash").read().strip()
    except OSError:
        return                                  # built by hand (hipcc ... -o libfpc_engine.so): nothing recorded
    import importlib.util
    spec = importlib.util.spec_from_file_location("_fpc_graft_entry", os.path.join(os.path.dirname(HERE), "__graft_entry__.py"))
    ge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ge)
    csrc = os.path.join(HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".cpp", ".h"))]
    srcs.append(os.path.join(os.path.dirname(HERE), "include", "fpc_engine.h"))
    if ge._src_hash(srcs, ge.HIPCC_FLAGS) != want:
        raise RuntimeError("stale HIP engine library: %s was built from other sources than the ones beside it; "
                           "rebuild with `python __graft_entry__.py`" % LIB_PATH)


def lib():
    """The product library.  Raises if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("HIP engine library missing: %s (build it with `python __graft_entry__.py`); "
                               "there is no CPU fallback" % LIB_PATH)
        _check_fresh()
        # PyTorch-ROCm ships its own libamdhip64.so.7; it must be the copy already mapped when the
        # engine library is loaded, otherwise two HIP runtimes end up in one process and the second
        # one to initialise sees no GPU.  torch is plumbing here (device tensors for the evaluator
        # seam, torch.distributed), not a compute path.
        try:
            import torch  # noqa: F401
            rccl = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            if os.path.exists(rccl):
                os.environ.setdefault("FPC_RCCL_LIB", rccl)    # fpc_comm_*: the RCCL that matches this HIP runtime
        except ImportError:
            pass
        _lib = bind(C.CDLL(LIB_PATH))
    return _lib


def comm_unique_id(_lib=None):
    """rank 0: the 128-byte ncclUniqueId to hand to the other ranks"""
    L = _lib if _lib is not None else lib()
    buf = (C.c_char * 128)()
    rc = L.fpc_comm_unique_id(buf)
    if rc != 0:
        raise RuntimeError("fpc_comm_unique_id failed (%d): %s" % (rc, (L.fpc_last_error(None) or b"").decode()))
    return bytes(buf)


def clone_board(b):
    nb = Board()
    C.memmove(C.byref(nb), C.byref(b), C.sizeof(Board))
    return nb


class Engine:
    """One engine handle == one GPU.  Thin, allocation-free-on-the-hot-path wrapper over the C-ABI."""

    def __init__(self, board_size, invalid_area, max_games=256, max_sims=400, avg_children=0, device=0, nn_dtype=0,
                 _lib=None):
        self.L = _lib if _lib is not None else lib()
        self.R, self.INV = board_size, invalid_area
        self.A_ch = self.L.fpc_num_action_channels(board_size)
        self.A = self.L.fpc_action_space_size(board_size)
        self.max_games, self.max_sims = max_games, max_sims
        cfg = Config(board_size, invalid_area, max_games, max_sims, avg_children, device, nn_dtype)
        h = C.c_void_p()
        rc = self.L.fpc_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RuntimeError("fpc_create failed (%d): %s" % (rc, (self.L.fpc_last_error(None) or b"").decode()))
        self.h = h
        self.G = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.fpc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError((self.L.fpc_last_error(self.h) or b"").decode() or ("fpc error %d" % rc))

    # ---- batched position ops (boards: list of Board, mutated in place like the reference) ----
    @staticmethod
    def _arr(boards):
        return (Board * len(boards))(*boards)

    @staticmethod
    def _writeback(arr, boards):
        for i, b in enumerate(boards):
            C.memmove(C.byref(b), C.byref(arr[i]), C.sizeof(Board))

    def legal_moves(self, boards):
        n = len(boards)
        arr = self._arr(boards)
        mv = (Move * (n * MAX_MOVES))()
        cnt = (C.c_int * n)()
        self._chk(self.L.fpc_boards_legal_moves(self.h, arr, n, mv, cnt))
        self._writeback(arr, boards)
        out = []
        for i in range(n):
            out.append([(mv[i * MAX_MOVES + k].frm, mv[i * MAX_MOVES + k].to, mv[i * MAX_MOVES + k].flat,
                         mv[i * MAX_MOVES + k].promo, mv[i * MAX_MOVES + k].capture) for k in range(cnt[i])])
        return out

    def game_result(self, boards, players=None):
        n = len(boards)
        arr = self._arr(boards)
        res = (C.c_int * n)()
        pl = (C.c_int * n)(*players) if players is not None else None
        self._chk(self.L.fpc_boards_game_result(self.h, arr, n, pl, res))
        self._writeback(arr, boards)
        return list(res)

    def take_action(self, boards, flats):
        n = len(boards)
        arr = self._arr(boards)
        out = (Board * n)()
        fl = (C.c_int * n)(*flats)
        self._chk(self.L.fpc_boards_take_action(self.h, arr, fl, n, out))
        return [clone_board(out[i]) for i in range(n)]

    def encode(self, boards):
        n = len(boards)
        out = np.zeros((n, 24, self.R, self.R), dtype=np.float32)
        self._chk(self.L.fpc_boards_encode(self.h, self._arr(boards), n, out.ctypes.data))
        return out

    def attack_maps(self, boards):
        """[n][6][R*R] uint8: attacked-square maps by colour (0..3, Board::IsAttackedByPlayer) and by team (4..5)"""
        n = len(boards)
        out = np.zeros((n, 6, self.R * self.R), dtype=np.uint8)
        self._chk(self.L.fpc_boards_attack_maps(self.h, self._arr(boards), n, out.ctypes.data))
        return out

    def legal_mask(self, boards):
        n = len(boards)
        arr = self._arr(boards)
        out = np.zeros((n, self.A_ch, self.R, self.R), dtype=np.float32)
        self._chk(self.L.fpc_boards_legal_mask(self.h, arr, n, out.ctypes.data))
        self._writeback(arr, boards)
        return out

    # ---- search ----
    def search_begin(self, roots, c_puct):
        self.G = len(roots)
        self._chk(self.L.fpc_search_begin(self.h, self._arr(roots), self.G, float(c_puct)))

    def search_select(self):
        n = C.c_int()
        p = C.c_void_p()
        self._chk(self.L.fpc_search_select(self.h, C.byref(n), C.byref(p)))
        return n.value, p.value

    def search_expand(self, logits_ptr, value_ptr):
        self._chk(self.L.fpc_search_expand(self.h, logits_ptr, value_ptr))

    def search_expand_select(self, logits_ptr, value_ptr):
        """search_expand of this simulation + search_select of the next one in one launch."""
        n = C.c_int()
        p = C.c_void_p()
        self._chk(self.L.fpc_search_expand_select(self.h, logits_ptr, value_ptr, C.byref(n), C.byref(p)))
        return n.value, p.value

    def search_run(self, sims):
        self._chk(self.L.fpc_search_run(self.h, sims))

    def search_results(self, max_children=256, roots=None, roots_np=None):
        """roots: list of Board to receive the root PODs as the search left them (piece-list order);
        roots_np: the same for a [G, 288] uint8 array (no per-game Python work)"""
        G = self.G
        rv = np.zeros(G, np.int32); nc = np.zeros(G, np.int32); sd = np.zeros(G, np.int32)
        cf = np.zeros((G, max_children), np.int32); cv = np.zeros((G, max_children), np.int32)
        cp = np.zeros((G, max_children), np.float32); cw = np.zeros((G, max_children), np.float64)
        pods = np.zeros((G, BOARD_BYTES), np.uint8) if roots_np is None else roots_np
        assert pods.shape == (G, BOARD_BYTES) and pods.dtype == np.uint8 and pods.flags.c_contiguous
        self._chk(self.L.fpc_search_results(self.h, _bp(pods), rv.ctypes.data, nc.ctypes.data, sd.ctypes.data, max_children,
                                            cf.ctypes.data, cv.ctypes.data, cp.ctypes.data, cw.ctypes.data))
        if roots is not None:
            for g, b in enumerate(roots):
                C.memmove(C.byref(b), pods[g].ctypes.data, BOARD_BYTES)
        return {"root_n": rv, "n_children": nc, "sims_done": sd, "flat": cf, "visits": cv, "prior": cp, "w": cw,
                "boards": _LazyBoards(pods)}

    # ---- the same position ops on [n, 288] uint8 arrays of PODs (callers that keep a whole batch in one array:
    #      bench.py's per-ply host section, mcts.py's root-children prefetch) ----
    def search_begin_np(self, pods, c_puct):
        assert pods.ndim == 2 and pods.shape[1] == BOARD_BYTES and pods.dtype == np.uint8 and pods.flags.c_contiguous
        self.G = pods.shape[0]
        self._chk(self.L.fpc_search_begin(self.h, _bp(pods), self.G, float(c_puct)))

    def take_action_np(self, pods, flats):
        """pods [n, 288] (left untouched), flats int32 [n] -> the n successor PODs"""
        pods = np.ascontiguousarray(pods, np.uint8)
        fl = np.ascontiguousarray(flats, np.int32)
        n = pods.shape[0]
        out = np.zeros((n, BOARD_BYTES), np.uint8)
        if n:
            self._chk(self.L.fpc_boards_take_action(self.h, _bp(pods), C.cast(fl.ctypes.data, P(C.c_int)), n, _bp(out)))
        return out

    def game_result_np(self, pods, players=None):
        """GetGameResult of every POD of the C-contiguous array `pods`, which is rewritten in place (the reference's
        GetGameResult permutes the piece lists); returns int32 [n]"""
        assert pods.dtype == np.uint8 and pods.flags.c_contiguous
        n = pods.shape[0]
        res = np.zeros(n, np.int32)
        pl = None if players is None else C.cast(np.ascontiguousarray(players, np.int32).ctypes.data, P(C.c_int))
        if n:
            self._chk(self.L.fpc_boards_game_result(self.h, _bp(pods), n, pl, C.cast(res.ctypes.data, P(C.c_int))))
        return res

    def grandchildren(self, game, child_idx, max_children=256):
        n = C.c_int()
        fl = np.zeros(max_children, np.int32); vi = np.zeros(max_children, np.int32)
        self._chk(self.L.fpc_search_grandchildren(self.h, game, child_idx, max_children, C.byref(n), fl.ctypes.data,
                                                  vi.ctypes.data))
        return [[int(fl[i]), int(vi[i])] for i in range(min(n.value, max_children))]

    def load_weights(self, blob):
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        self._chk(self.L.fpc_load_weights(self.h, buf, len(blob)))

    def nn_forward(self, enc_ptr, n, logits_ptr, value_ptr):
        self._chk(self.L.fpc_nn_forward(self.h, enc_ptr, n, logits_ptr, value_ptr))

    def stats(self):
        s = Stats()
        self.L.fpc_stats_get(self.h, C.byref(s))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def stats_reset(self):
        self.L.fpc_stats_reset(self.h)

    def set_policy_mode(self, legal_only):
        """fpc_search_run's policy head: False = the whole Linear + full softmax (reference arithmetic),
        True = only the leaf's legal moves (same priors up to f32 rounding; include/fpc_engine.h)."""
        self._chk(self.L.fpc_set_policy_mode(self.h, 1 if legal_only else 0))

    def set_timing(self, on):
        self.L.fpc_set_timing(self.h, 1 if on else 0)

    # ---- N4: non-strict rule set and root Dirichlet noise (include/fpc_engine.h FPC_RULES_*) ----
    def set_rules(self, rules):
        """0 = strict reference semantics (default); RULES_FIXED = all corrections"""
        self._chk(self.L.fpc_set_rules(self.h, int(rules)))

    def set_root_noise(self, gamma, eps):
        """gamma: float32 array [n_games, MAX_MOVES] of Gamma(alpha) draws (None: off)"""
        if gamma is None:
            self._chk(self.L.fpc_search_set_root_noise(self.h, None, 0, 0.0))
            return
        g = np.ascontiguousarray(gamma, dtype=np.float32)
        assert g.ndim == 2 and g.shape[1] == MAX_MOVES
        self._chk(self.L.fpc_search_set_root_noise(self.h, g.ctypes.data, g.shape[0], float(eps)))

    # ---- training tuples (device resident) and their RCCL exchange ----
    def tuples_reserve(self, capacity):
        self._chk(self.L.fpc_tuples_reserve(self.h, int(capacity)))

    def tuples_reset(self):
        self._chk(self.L.fpc_tuples_reset(self.h))

    def collect_tuples(self, game_ids, ply):
        """one tuple per game of the search that just finished (root mailbox, side to move, sparse pi);
        game_ids: sequence or int32 array"""
        ids = None
        if game_ids is not None:
            ia = np.ascontiguousarray(game_ids, np.int32)
            ids = C.cast(ia.ctypes.data, P(C.c_int))
        self._chk(self.L.fpc_collect_tuples(self.h, ids, int(ply)))

    def tuples_set_z(self, game_ids, z_team0, z_team1):
        """one call for any number of finished games (sequences or arrays)"""
        ia = np.ascontiguousarray(game_ids, np.int32)
        z0 = np.ascontiguousarray(z_team0, np.float32)
        z1 = np.ascontiguousarray(z_team1, np.float32)
        n = ia.shape[0]
        assert z0.shape[0] == n and z1.shape[0] == n
        self._chk(self.L.fpc_tuples_set_z(self.h, C.cast(ia.ctypes.data, P(C.c_int)), C.cast(z0.ctypes.data, P(C.c_float)),
                                          C.cast(z1.ctypes.data, P(C.c_float)), n))

    def tuples_count(self):
        return self.L.fpc_tuples_count(self.h)

    def tuples_read(self, first=0, n=None):
        n = self.tuples_count() - first if n is None else n
        arr = (Tuple * max(n, 1))()
        self._chk(self.L.fpc_tuples_read(self.h, arr, first, n))
        return arr, n

    def comm_init(self, id128, rank, world):
        buf = (C.c_char * 128).from_buffer_copy(bytes(id128))
        self._chk(self.L.fpc_comm_init(self.h, buf, rank, world))
        self.comm_world = world

    def debug_comm_fault(self, point):
        """TEST HOOK: the next allgather_tuples treats one of its own HIP calls as failed (include/fpc_engine.h)"""
        self._chk(self.L.fpc_debug_comm_fault(self.h, point))

    def allgather_tuples_device(self):
        """episode end: RCCL all-gather of every rank's tuples, driven from the C++ host; the gathered
        tuples stay on the device.  Returns (per-rank counts, total)."""
        counts, total = (C.c_int * max(getattr(self, 'comm_world', 1), 1))(), C.c_int()
        self._chk(self.L.fpc_allgather_tuples(self.h, counts, C.byref(total)))
        return counts, total.value

    def gathered_read(self, total):
        arr = (Tuple * max(total, 1))()
        if total:
            self._chk(self.L.fpc_gathered_read(self.h, arr, 0, total))
        return arr

    def allgather_tuples(self):
        """(per-rank counts, ctypes array of all tuples in rank order, total)"""
        counts, total = self.allgather_tuples_device()
        return counts, self.gathered_read(total), total


def board_from_dict(R, turn, entries, castle=None, _lib=None):
    """entries: [(sq, colour, type), ...] in dict insertion order -> Board with the reference's ctor order."""
    L = _lib if _lib is not None else lib()
    n = len(entries)
    sqs = (C.c_uint8 * max(n, 1))(*[e[0] for e in entries])
    pcs = (C.c_uint8 * max(n, 1))(*[0x80 | (e[1] << 5) | (e[2] << 2) for e in entries])
    cs = (C.c_uint8 * 4)(*castle) if castle is not None else None
    b = Board()
    rc = L.fpc_board_from_dict(C.byref(b), R, turn, sqs, pcs, n, cs)
    if rc != 0:
        raise RuntimeError("fpc_board_from_dict failed (%d)" % rc)
    return b


def board_from_lists(R, turn, pl):
    """pl: per colour [[sq, type], ...] already in piece-list order (fixture format)."""
    b = Board()
    for c in range(4):
        b.king[c] = NO_SQ
    b.turn = turn
    for colour, col in enumerate(pl):
        for sq, typ in col:
            b.sq[sq] = 0x80 | (colour << 5) | (typ << 2)
            b.pl[colour][b.plen[colour]] = sq
            b.plen[colour] += 1
            if typ == 5:
                b.king[colour] = sq
    return b


def lists_of(b):
    return [[[b.pl[c][i], (b.sq[b.pl[c][i]] >> 2) & 7] for i in range(b.plen[c])] for c in range(4)]
