"""`alphazero_cpp` -- drop-in for the reference's pybind11 module of the same name
(/root/reference/src/cpp/wrapper.cpp:15-254), rebuilt over the C-ABI of the MI355X engine
(include/fpc_engine.h, libfpc_engine.so).  Put this directory on sys.path exactly like the
reference's build directory: `from alphazero_cpp import Board, Move, Node, ...` keeps working.

What is different by design: a Python `Board` is a 288-byte POD snapshot (fpc_ffi.Board) and every
game-logic call (GetLegalMoves, GetGameResult, TakeAction, encode, masks) is one wavefront on the
GPU; the MCTS tree lives in device memory, `Node` objects are read-only views of search results.
There is no CPU fallback: without the HIP library / a GPU these calls raise RuntimeError.

Board size: the reference fixes rows_/cols_/invalid_area at compile time (engine/board.h:22-24);
here it is chosen at import from FPC_BOARD_SIZE (8 -> 8x8/2, the literal snapshot; 14 -> 14x14/3,
default) or later with configure() before any Board exists.
"""
import enum
import os

import fpc_ffi as _ffi

__all__ = []


def _export(cls):
    for m in cls:
        globals()[m.name] = m
        __all__.append(m.name)
    __all__.append(cls.__name__)
    return cls


@_export
class PieceType(enum.IntEnum):          # wrapper.cpp:32-40
    PAWN = 0
    KNIGHT = 1
    BISHOP = 2
    ROOK = 3
    QUEEN = 4
    KING = 5
    NO_PIECE = 6


@_export
class PlayerColor(enum.IntEnum):        # wrapper.cpp:45-51
    UNINITIALIZED_PLAYER = -1
    RED = 0
    BLUE = 1
    YELLOW = 2
    GREEN = 3


@_export
class Team(enum.IntEnum):               # wrapper.cpp:56-59
    RED_YELLOW = 0
    BLUE_GREEN = 1


@_export
class GameResult(enum.IntEnum):         # wrapper.cpp:61-66
    IN_PROGRESS = 0
    WIN_RY = 1
    WIN_BG = 2
    STALEMATE = 3


def piece_value(t):                     # wrapper.cpp:42
    return int(t)


def color_value(c):                     # wrapper.cpp:53
    return int(c)


# ---- board geometry (compile-time constants in the reference) --------------------------------
_SIZES = {8: 2, 10: 2, 13: 3, 14: 3}
_R = int(os.environ.get("FPC_BOARD_SIZE", "14"))
_INV = _SIZES.get(_R, 3)
_engine = None
_engine_cap = (0, 0)


def configure(board_size, invalid_area=None):
    """Select the board geometry (the reference recompiles for this).  Call before creating boards."""
    global _R, _INV, _engine
    _R = int(board_size)
    _INV = int(invalid_area) if invalid_area is not None else _SIZES[_R]
    if _engine is not None:
        _engine.close()
        _engine = None
    _set_statics()


def engine(min_games=1, min_sims=1, nn_dtype=None):
    """The process-wide engine handle (one per GPU); grown on demand."""
    global _engine, _engine_cap
    want_dtype = _engine.nn_dtype if (_engine is not None and nn_dtype is None) else (nn_dtype or 0)
    if _engine is None or _engine_cap[0] < min_games or _engine_cap[1] < min_sims or _engine.nn_dtype != want_dtype:
        g = max(min_games, _engine_cap[0], 64)
        s = max(min_sims, _engine_cap[1], 64)
        if _engine is not None:
            _engine.close()
        dev = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("FPC_DEVICE") is None else int(os.environ["FPC_DEVICE"])
        _engine = _ffi.Engine(_R, _INV, max_games=g, max_sims=s, device=dev, nn_dtype=want_dtype)
        _engine.nn_dtype = want_dtype
        _engine.weights_version = None
        _engine_cap = (g, s)
    return _engine


class Player:                           # engine/board.h:57-81, wrapper.cpp:81-87
    __hash__ = None

    def __init__(self, color=PlayerColor.UNINITIALIZED_PLAYER):
        self._c = PlayerColor(int(color))

    def GetColor(self):
        return self._c

    def GetTeam(self):
        return Team.RED_YELLOW if self._c in (PlayerColor.RED, PlayerColor.YELLOW) else Team.BLUE_GREEN

    def __eq__(self, o):
        return isinstance(o, Player) and self._c == o._c

    def __ne__(self, o):
        return not self == o

    def __repr__(self):
        return "Player(%s)" % self._c.name


_COLOR_STR = {0: "Red", 1: "Blue", 2: "Yellow", 3: "Green"}
_TYPE_STR = {0: "Pawn", 1: "Knight", 2: "Bishop", 3: "Rook", 4: "Queen", 5: "King"}


class Piece:                            # engine/board.h:96-186, wrapper.cpp:89-103
    __hash__ = None

    def __init__(self, *a):
        if len(a) == 0:
            present, color, ptype = False, PlayerColor.RED, PieceType.NO_PIECE
        elif len(a) == 3:
            present, color, ptype = bool(a[0]), a[1], a[2]
        elif len(a) == 2:
            present, color, ptype = True, (a[0].GetColor() if isinstance(a[0], Player) else a[0]), a[1]
        else:
            raise TypeError("Piece(): incompatible constructor arguments")
        self._bits = ((1 if present else 0) << 7) | ((int(color) & 3) << 5) | ((int(ptype) & 7) << 2)

    @classmethod
    def _from_byte(cls, b):
        p = cls()
        if b & 0x80:
            p._bits = b & 0xFC
        return p

    def _byte(self):
        return self._bits if self._bits & 0x80 else 0

    def Present(self):
        return bool(self._bits & 0x80)

    def GetColor(self):
        return PlayerColor((self._bits >> 5) & 3)

    def GetPieceType(self):
        return PieceType((self._bits >> 2) & 7)

    def GetPlayer(self):
        return Player(self.GetColor())

    def PieceTypeToStr(self, type):
        if int(type) not in _TYPE_STR:
            raise RuntimeError("Unknown piece type")
        return _TYPE_STR[int(type)]

    def ColorToStr(self, color):
        if int(color) not in _COLOR_STR:
            raise RuntimeError("Unknown color")
        return _COLOR_STR[int(color)]

    def __eq__(self, o):
        return isinstance(o, Piece) and self._bits == o._bits

    def __ne__(self, o):
        return not self == o

    def __str__(self):
        if not self.Present():
            raise RuntimeError("Missing piece")
        return self.ColorToStr(self.GetColor()) + " " + self.PieceTypeToStr(self.GetPieceType())


class BoardLocation:                    # engine/board.h:190-224, wrapper.cpp:105-115
    def __init__(self, row=None, col=None):
        if row is None:
            self._loc = _R * _R
        else:
            row, col = int(row), int(col)
            self._loc = _R * _R if (row < 0 or row >= _R or col < 0 or col >= _R) else _R * row + col

    @classmethod
    def _from_sq(cls, sq):
        b = cls()
        b._loc = sq if sq < _R * _R else _R * _R
        return b

    def Present(self):
        return self._loc < _R * _R

    def GetRow(self):
        return self._loc // _R

    def GetCol(self):
        return self._loc % _R

    def _sq(self):
        return self._loc if self._loc < _R * _R else _ffi.NO_SQ

    def __eq__(self, o):
        return isinstance(o, BoardLocation) and self._loc == o._loc

    def __hash__(self):
        return hash(self.GetRow()) ^ hash(self.GetCol())

    def __str__(self):                  # BoardLocation::PrettyStr, engine/board.cpp:1531-1537
        return "%s%d (%d, %d)" % (chr(ord("a") + self.GetCol()), _R - self.GetRow(), self.GetRow(), self.GetCol())


class CastlingRights:                   # engine/board.h:285-328, wrapper.cpp:117-123
    __hash__ = None

    def __init__(self, *a):
        if len(a) == 0:
            self._bits = 0
        else:
            self._bits = 0x80 | ((1 if a[0] else 0) << 6) | ((1 if a[1] else 0) << 5)

    def Present(self):
        return bool(self._bits & 0x80)

    def Kingside(self):
        return bool(self._bits & 0x40)

    def Queenside(self):
        return bool(self._bits & 0x20)

    def __eq__(self, o):
        return isinstance(o, CastlingRights) and self._bits == o._bits

    def __ne__(self, o):
        return not self == o


class PlacedPiece:                      # engine/board.h:446-471, wrapper.cpp:125-131
    def __init__(self, location=None, piece=None):
        self._l = location if location is not None else BoardLocation()
        self._p = piece if piece is not None else Piece()

    def GetLocation(self):
        return self._l

    def GetPiece(self):
        return self._p

    def __str__(self):
        return str(self._p) + " at " + str(self._l)


class Move:                             # move.h:23-49, move.cpp:23-104, wrapper.cpp:135-163
    _flat = None        # Move(flat_index) keeps the index; from / to are decoded when somebody asks (move.cpp:39-61)
    _capture = None
    _promo = PieceType.NO_PIECE

    def __init__(self, *a, **kw):
        if kw:
            a = tuple(a) + tuple(kw[k] for k in ("flat_index", "action_plane", "from", "c_move", "to") if k in kw)
        if len(a) == 0:
            self._ft = (BoardLocation(), BoardLocation())
            return
        if len(a) == 1 and isinstance(a[0], Move):
            o = a[0]
            self._flat, self._capture, self._promo = o._flat, o._capture, o._promo
            if o._flat is None:
                self._ft = (o._from, o._to)
        elif len(a) == 1:                                   # Move(flat_index), move.cpp:39-61
            flat = int(a[0])
            if flat < 0 or flat >= Board.action_space_size:
                raise RuntimeError("flat index out of range")
            self._flat = flat
        elif len(a) == 2 and isinstance(a[1], BoardLocation) and not isinstance(a[0], BoardLocation):
            frm = a[1]                                      # Move(action_plane, from), move.cpp:23-37
            flat = int(a[0]) * _R * _R + frm.GetRow() * _R + frm.GetCol()
            if flat < 0 or flat >= Board.action_space_size:
                raise RuntimeError("flat index out of range")
            self._flat = flat
        else:                                               # standard / pawn-move constructors
            self._ft = (a[0], a[1])
            if len(a) > 2 and isinstance(a[2], Piece):
                self._capture = a[2]
            if len(a) == 6:
                self._promo = PieceType(int(a[5]))

    def _ends(self):
        e = self.__dict__.get("_ft")
        if e is None:
            import ctypes as C
            f, t = C.c_int(), C.c_int()
            _ffi.lib().fpc_flat_to_move(_R, int(self._flat), C.byref(f), C.byref(t))
            e = self.__dict__["_ft"] = (BoardLocation._from_sq(f.value), BoardLocation._from_sq(t.value))
        return e

    @property
    def _from(self):
        return self._ends()[0]

    @_from.setter
    def _from(self, v):
        self.__dict__["_ft"] = (v, self._ends()[1])
        self._flat = None

    @property
    def _to(self):
        return self._ends()[1]

    @_to.setter
    def _to(self, v):
        self.__dict__["_ft"] = (self._ends()[0], v)
        self._flat = None

    def From(self):
        return self._from

    def To(self):
        return self._to

    def GetIndex(self):
        flat = self.GetFlatIndex()
        return (flat // (_R * _R), self._from.GetRow(), self._from.GetCol())

    def GetFlatIndex(self):
        if self._flat is not None:
            return self._flat
        f, t = self._ends()
        flat = _ffi.lib().fpc_move_flat_index(_R, f._sq(), t._sq()) if (f.Present() and t.Present()) else -1
        if flat < 0:
            raise RuntimeError("Invalid move: No corresponding action plane index found. Did you initialize move_index_map?")
        return flat

    def __repr__(self):
        return "Move: %s -> %s" % (self._from, self._to)


class SimpleBoardState:                 # engine/board.h:491-497, wrapper.cpp:68-73 (read-write attributes)
    def __init__(self):
        self.turn = None
        self.pieces = []
        self.castlingRights = []
        self.attackedSquares = {}


class MemoryEntry:                      # board.h:133-140: Board held BY VALUE + the pi tensor
    def __init__(self, state, action):
        self.state = state._copy()
        self.action = action


class Node:
    """A search-tree node (node.h:17-79, wrapper.cpp:233-253).

    Two kinds of object share this class.  `MCTS.search` (mcts.py here) keeps the whole tree on the GPU
    and hands back READ-ONLY views of the roots: GetChildren / GetMoveMade / GetVisitCount / GetState /
    IsExpanded, which is all the training loop reads (alphazero.py:104-110).  A `Node(C, state, ...)`
    built by the caller is a HOST tree node with the reference's full method set -- ChooseLeaf,
    SelectChild, Backpropagate, BackpropagateNodes, ExpandNodes (node.cpp:19-154), so that code written
    against the reference's per-simulation loop (its own mcts.py:17-89) runs unchanged: the tree
    bookkeeping is restated here in Python (f64 PUCT with IEEE sqrt / libm log, strict `>`: the lowest
    index wins ties), every board operation underneath (GetGameResult, the child boards of an expansion,
    encode, legal moves) still goes through the engine's C-ABI.  It is the compatibility path, orders of
    magnitude slower than MCTS.search, and never used by it."""

    _batch = None       # roots returned by MCTS.search: the _SearchBatch whose arrays hold their children ...
    _g = 0              # ... and the game's row in them

    def __init__(self, C=0.0, state=None, parent=None, action_taken=None, prior=0.0, visit_count=0):
        self._C, self._state, self._parent, self._move = C, state, parent, action_taken
        self._prior, self._n, self._children = prior, visit_count, []
        self._value_sum = 0.0
        self._lazy = None

    def GetMoveMade(self):
        return self._move

    def GetState(self):
        return self._state

    def _materialise(self):
        """a search root's child views are made the first time somebody asks for them (a search over 256 games
        returns ~6 000 root children; callers that read the arrays -- child_arrays() -- never pay for objects)"""
        b = self._batch
        if b is not None and not self._children:
            self._children = b.child_views(self)
        return self._children

    def child_arrays(self):
        """(flat index, visit count) of this node's children as two int arrays, ascending flat index -- what
        alphazero.py:104-110 builds pi from -- without creating a Python object per child"""
        b = self._batch
        if b is not None:
            n = int(b.res["n_children"][self._g])
            return b.res["flat"][self._g, :n], b.res["visits"][self._g, :n]
        import numpy as np
        kids = self.GetChildren()
        return (np.array([c.GetMoveMade().GetFlatIndex() for c in kids], np.int32),
                np.array([c.GetVisitCount() for c in kids], np.int32))

    def n_children(self):
        b = self._batch
        return int(b.res["n_children"][self._g]) if b is not None else len(self.GetChildren())

    def GetChildren(self):
        if self._batch is not None:
            return list(self._materialise())
        if self._lazy is not None:
            eng, game, idx = self._lazy
            self._lazy = None
            self._children = [Node(self._C, None, self, Move(fl), 0.0, n) for fl, n in eng.grandchildren(game, idx)]
        return list(self._children)

    def GetVisitCount(self):
        return self._n

    def SetVisitCount(self, v):
        self._n = int(v)

    def IsExpanded(self):
        return len(self.GetChildren()) > 0

    # ---- host forms (node.cpp) ----------------------------------------------------------------------------
    def _host_only(self):
        if self._state is None:
            raise RuntimeError("this node is a read-only view of a tree that lives on the GPU (MCTS.search): it has no "
                               "state to descend into; build host nodes with Node(C, state, ...)")

    def SelectChild(self):              # node.cpp:49-78
        import numpy as np
        kids = self._children
        with np.errstate(all="ignore"):
            log_parent = np.log(np.sqrt(np.float64(self._n)))
            best, best_ucb = -1, -np.inf
            for i, ch in enumerate(kids):
                n = ch._n
                q = np.float64(ch._value_sum) / n if n > 0 else np.float64(0.0)
                ucb = q + np.float64(self._C) * np.sqrt(log_parent / np.float64(1 + n)) * np.float64(ch._prior)
                if ucb > best_ucb:
                    best, best_ucb = i, ucb
        if best < 0:
            raise RuntimeError("Failed to select a child.")
        return kids[best]

    def ChooseLeaf(self):               # node.cpp:19-47
        self._host_only()
        node = self
        while node._children:
            node = node.SelectChild()
        result = node._state.GetGameResult()
        if result != GameResult.IN_PROGRESS:
            node.Backpropagate(0.0 if result == GameResult.STALEMATE else -1.0)
            return None
        return node

    def Backpropagate(self, value):     # node.cpp:133-142: value_sum (double) += value (float); the sign flips per ply
        import numpy as np
        v = float(np.float32(value))
        node = self
        while node is not None:
            node._value_sum += v
            node._n += 1
            v = -v
            node = node._parent

    @staticmethod
    def BackpropagateNodes(nodes, values):          # node.cpp:144-154
        vals = values.detach().to("cpu").reshape(-1).tolist() if hasattr(values, "detach") else list(values)
        for node, v in zip(nodes, vals):
            node.Backpropagate(v)

    @staticmethod
    def ExpandNodes(nodes, policy_batch, non_zero_indices_batch, non_zero_values, pool):      # node.cpp:79-131
        """children in the order of the non-zero (plane, row, col) entries; each child's board is the parent's
        board after Move(plane, from) -- from / to only (Q9) -- and starts with visit_count 1 (node.h:28, Q1)"""
        per = [[] for _ in nodes]
        for (b, plane, row, col), prob in zip(non_zero_indices_batch, non_zero_values):
            per[int(b)].append((Move(int(plane), BoardLocation(int(row), int(col))), float(prob)))
        pods, flats = [], []
        for node, entries in zip(nodes, per):
            node._host_only()
            for mv, _p in entries:
                pods.append(node._state._b)
                flats.append(mv.GetFlatIndex())
        made = engine(min_games=1).take_action(pods, flats) if pods else []
        k = 0
        for node, entries in zip(nodes, per):
            for mv, prob in entries:
                child_state = Board._wrap(made[k], like=node._state)
                k += 1
                node._children.append(Node(node._C, child_state, node, mv, prob, 1))


class _RootChild(Node, Move):
    """Read-only view of ONE child of a search root (MCTS.search keeps the tree on the GPU): the Node read API
    the training loop uses (GetMoveMade / GetVisitCount / GetChildren / GetState / IsExpanded, alphazero.py:104-110)
    over the search's result arrays.  It is also its own `Move` -- GetMoveMade() returns the view itself, which
    answers GetFlatIndex / GetIndex / From / To -- so that reading a child costs one small object, not a Node, a Move
    and two BoardLocations (6 000 root children per ply at 256 games)."""
    __slots__ = ("_flat", "_n", "_k", "_parent")
    _state = None
    _lazy = None
    _batch = None

    def __init__(self, flat, n, k, parent):
        self._flat, self._n, self._k, self._parent = flat, n, k, parent

    # -- Node side
    def GetMoveMade(self):
        return self

    def GetChildren(self):
        kids = self.__dict__.get("_kids")
        if kids is None:
            b = self._parent._batch
            kids = [Node(self._parent._C, None, self, Move(fl), 0.0, n) for fl, n in b.eng.grandchildren(self._parent._g, self._k)]
            self.__dict__["_kids"] = kids
        return list(kids)

    @property
    def _children(self):
        return self.GetChildren()

    @property
    def _C(self):
        return self._parent._C

    @property
    def _move(self):
        return self

    @property
    def _prior(self):
        return float(self._parent._batch.res["prior"][self._parent._g, self._k])

    @property
    def _value_sum(self):
        return float(self._parent._batch.res["w"][self._parent._g, self._k])

    def _host_only(self):
        raise RuntimeError("this node is a read-only view of a tree that lives on the GPU (MCTS.search): it has no "
                           "state to descend into; build host nodes with Node(C, state, ...)")

    def SelectChild(self):
        self._host_only()

    def Backpropagate(self, value):
        self._host_only()

    # -- Move side (move.cpp:39-61: a move rebuilt from its flat index has from / to only, Q9)
    def GetFlatIndex(self):
        return self._flat

    def __repr__(self):
        return "Node(move %d, N=%d)" % (self._flat, self._n)


class _SearchBatch:
    """What one MCTS.search call left behind: the engine's result arrays for all its games (root children as
    [G, max_children] arrays) and, made on first use, the successor position of EVERY root child with its game result
    -- one batched TakeAction and one batched GetGameResult for the whole search instead of two synchronous GPU round
    trips per game (alphazero.py:119-123 calls state.TakeAction(action) and next_state.GetGameResult() for each game
    in turn: 512 round trips per ply at 256 games)."""

    def __init__(self, eng, res, C):
        self.eng, self.res, self.C = eng, res, C
        self._succ = None

    def child_views(self, root):
        g = root._g
        n = int(self.res["n_children"][g])
        flats, visits = self.res["flat"][g, :n].tolist(), self.res["visits"][g, :n].tolist()
        return [_RootChild(f, v, k, root) for k, (f, v) in enumerate(zip(flats, visits))]

    def _successors(self):
        if self._succ is None:
            import numpy as np
            n = self.res["n_children"].astype(np.int64)
            roots = self.res["boards"]._pods                         # the root PODs as the search left them
            width = self.res["flat"].shape[1]
            live = np.arange(width)[None, :] < n[:, None]
            flats = self.res["flat"][live]                           # row-major: game by game, ascending flat index
            pre = engine().take_action_np(np.repeat(roots, n, axis=0), flats)
            post = pre.copy()
            results = engine().game_result_np(post)                  # rewrites `post` in place (piece-list order)
            offs = np.concatenate([[0], np.cumsum(n)])
            self._succ = (offs, pre, post, results, roots)
        return self._succ

    def successor(self, g, flat, parent_pod):
        """(successor POD, (its bytes, POD after GetGameResult, result)) of root g's child `flat`, or None when
        `flat` is no root child or the caller's position is not the root this search was run on"""
        import ctypes as C
        import numpy as np
        offs, pre, post, results, roots = self._successors()
        n = int(self.res["n_children"][g])
        row = self.res["flat"][g, :n]
        k = int(np.searchsorted(row, flat))
        if k >= n or int(row[k]) != flat or bytes(parent_pod) != roots[g].tobytes():
            return None
        i = int(offs[g]) + k
        return _ffi.board_of(pre[i]), (pre[i].tobytes(), post[i], int(results[i]))


class Board:                            # board.h:18-131, wrapper.cpp:165-226
    num_state_channels = 24
    _gr_cache = None    # set by TakeAction from a search root: (POD bytes, POD after GetGameResult, result)

    def __init__(self, turn=None, location_to_piece=None, castling_rights=None, root_state=None):
        self._root_node, self._root_state, self._memory = None, root_state, []
        if turn is None and location_to_piece is None:
            self._b = _ffi.Board()
            return
        entries = [(loc.GetRow() * _R + loc.GetCol(), int(p.GetColor()), int(p.GetPieceType()))
                   for loc, p in location_to_piece.items()]
        castle = None
        if castling_rights:
            castle = [0, 0, 0, 0]
            for k, v in castling_rights.items():
                c = int(k.GetColor()) if isinstance(k, Player) else int(k)
                castle[c] = (1 if v.Kingside() else 0) | (2 if v.Queenside() else 0)
        self._b = _ffi.board_from_dict(_R, int(turn.GetColor()), entries, castle)

    # -- helpers
    def _copy(self):
        nb = Board.__new__(type(self))
        nb._b = _ffi.clone_board(self._b)
        nb._root_node, nb._root_state, nb._memory = self._root_node, self._root_state, list(self._memory)
        return nb

    @classmethod
    def _wrap(cls, pod, like=None):
        nb = Board.__new__(cls if like is None else type(like))
        nb._b = pod
        nb._root_node, nb._root_state, nb._memory = None, None, []
        if like is not None:
            nb._root_node, nb._root_state, nb._memory = like._root_node, like._root_state, list(like._memory)
        return nb

    # -- plain accessors
    def CalculateHeuristic(self, team):
        return _ffi.lib().fpc_board_heuristic(self._b, int(team))

    def GetTurn(self):
        return Player(PlayerColor(self._b.turn))

    def SetTurn(self, player):
        self._b.turn = int(player.GetColor())

    @staticmethod
    def GetOpponentValue(val):
        return -val

    def GetPieceAt(self, x, y):
        if not (0 <= x < _R and 0 <= y < _R):
            raise RuntimeError("Location out of bounds")
        return Piece._from_byte(self._b.sq[x * _R + y])

    def GetBoardLocation(self, x, y):
        if not (0 <= x < _R and 0 <= y < _R):
            raise RuntimeError("Location out of bounds")
        return BoardLocation(x, y)

    def GetPieces(self):
        out = []
        for c in range(4):
            out.append([PlacedPiece(BoardLocation._from_sq(self._b.pl[c][i]), Piece._from_byte(self._b.sq[self._b.pl[c][i]]))
                        for i in range(self._b.plen[c])])
        return out

    def GetRootNode(self):
        return self._root_node

    def SetRootNode(self, n):
        self._root_node = n

    def GetRootState(self):             # board.h:50-58
        return self._copy() if self._root_state is None else self._root_state

    def SetRootState(self, s):
        self._root_state = s

    def GetMemory(self):                # board.h:64-72
        return self._memory if self._root_state is None else self._root_state._memory

    def AppendToMemory(self, entry):    # board.h:74-83
        (self._memory if self._root_state is None else self._root_state._memory).append(entry)

    # -- GPU-backed game logic
    def GetGameResult(self, opt_player=None):
        c = self._gr_cache
        if c is not None:
            self._gr_cache = None
            if opt_player is None and bytes(self._b) == c[0]:       # still the position TakeAction produced
                import ctypes as C
                C.memmove(C.byref(self._b), c[1].ctypes.data, _ffi.BOARD_BYTES)   # the call permutes the piece lists
                return GameResult(c[2])
        pl = None if opt_player is None else [int(opt_player.GetColor())]
        return GameResult(engine().game_result([self._b], pl)[0])

    def IsMoveLegal(self, move):
        return False                    # the reference's implementation can never return true

    def GetLegalMoves(self):
        out = []
        for frm, to, flat, promo, cap in engine().legal_moves([self._b])[0]:
            kinds = (PieceType.KNIGHT, PieceType.BISHOP, PieceType.ROOK, PieceType.QUEEN) if promo else (PieceType.NO_PIECE,)
            for k in kinds:             # promotions are emitted 4x (engine/board.cpp:82-88)
                m = Move.__new__(Move)
                m._ft = (BoardLocation._from_sq(frm), BoardLocation._from_sq(to))
                m._capture, m._promo = Piece._from_byte(cap), k
                out.append(m)
        return out

    def TakeAction(self, move):         # board.cpp:234-239: copy, then MakeMove with (from,to) only
        flat = move.GetFlatIndex()
        rn = self._root_node
        if rn is not None and rn._batch is not None:
            hit = rn._batch.successor(rn._g, flat, self._b)      # prefetched with every other root child of that search
            if hit is not None:
                nb = Board._wrap(hit[0], like=self)
                nb._gr_cache = hit[1]
                return nb
        pod = engine().take_action([self._b], [flat])[0]
        return Board._wrap(pod, like=self)

    # ---- the attacked-square queries (wrapper.cpp:201-206; board.cpp:50-57, :120-232): one device launch per call
    def _attack_maps(self):
        return engine().attack_maps([self._b])[0]          # [6][R*R] 0/1: colours 0..3, teams 0..1

    def GetAttackedSquaresPlayers(self):                    # board.cpp:120-140: {colour: [location, ...]} row-major;
        m = self._attack_maps()                             # a colour that attacks nothing has no entry
        out = {}
        for c in range(4):
            sq = [int(x) for x in m[c].nonzero()[0]]
            if sq:
                out[PlayerColor(c)] = [BoardLocation._from_sq(q) for q in sq]
        return out

    def GetAttackedSquaresTeams(self):                      # board.cpp:212-232, the engine's IsAttackedByTeam per square
        m = self._attack_maps()
        out = {}
        for t in range(2):
            sq = [int(x) for x in m[4 + t].nonzero()[0]]
            if sq:
                out[Team(t)] = [BoardLocation._from_sq(q) for q in sq]
        return out

    def IsAttackedByPlayer(self, location, color):          # board.cpp:142-210
        if not location.Present():
            return False                                    # every probe around a missing location is missing too
        return bool(self._attack_maps()[int(color)][location._sq()])

    def GetSimpleState(self):                               # board.cpp:50-57
        st = SimpleBoardState()
        st.turn = self.GetTurn()
        st.pieces = self.GetPieces()
        # castling_rights_ is CastlingRights(false, false) unless the constructor was handed rights (engine/board.cpp:1178-1191)
        st.castlingRights = [CastlingRights(bool(self._b.castle[c] & 1), bool(self._b.castle[c] & 2)) for c in range(4)]
        st.attackedSquares = self.GetAttackedSquaresPlayers()
        return st

    @staticmethod
    def ParseActionspace(actionspaces_1d, turn):      # board.cpp:257-263
        import torch
        v = actionspaces_1d.view(-1, *Board.action_space_dims)
        return torch.rot90(v, -int(turn.GetColor()), (-2, -1))

    @staticmethod
    def ChangePerspective(tensor, rotation):          # board.cpp:252-255
        import torch
        return torch.rot90(tensor, rotation, (-2, -1))

    @staticmethod
    def IsLegalLocation(*a):
        if len(a) == 1:
            a = (a[0].GetRow(), a[0].GetCol())
        return bool(_ffi.lib().fpc_is_legal_location(_R, _INV, int(a[0]), int(a[1])))

    @staticmethod
    def nRows():
        return _R

    @staticmethod
    def nCols():
        return _R

    @staticmethod
    def invalidArea():
        return _INV

    @staticmethod
    def GetOpponent(c):                 # board.cpp:241-250
        c = c.GetColor() if isinstance(c, Player) else c
        return PlayerColor((int(c) + 1) % 4)

    @staticmethod
    def GetEncodedStates(states, device):             # board.cpp:305-356
        import torch
        if device not in ("cpu", "gpu", "cuda"):
            raise RuntimeError("Invalid device argument.")
        t = torch.from_numpy(engine().encode([s._b for s in states]))
        return t if device == "cpu" else t.cuda()

    @staticmethod
    def GetEncodedState(state, device):
        return Board.GetEncodedStates([state], device)

    @staticmethod
    def GetLegalMovesIndices(legal_moves, num_moves):  # board.cpp:424-449
        b, p, r, c = [], [], [], []
        for bi, moves in enumerate(legal_moves):
            for m in moves:
                pl, row, col = m.GetIndex()
                b.append(bi); p.append(pl); r.append(row); c.append(col)
        return b, p, r, c

    def __str__(self):                  # operator<<(Board), engine/board.cpp:1429-1467
        lines = []
        for i in range(_R):
            s = (" " if _R - i < 9 else "") + str(_R - i + 1) + ":"
            for j in range(_R):
                if Board.IsLegalLocation(i, j):
                    p = self._b.sq[i * _R + j]
                    s += " . " if not p & 0x80 else "%d%s " % ((p >> 5) & 3, "PNBRQK"[(p >> 2) & 7])
                else:
                    s += "   "
            lines.append(s)
        lines.append("   " + "".join(" %s " % chr(ord("a") + j) for j in range(_R)))
        lines.append("Turn: Player(%s)" % PlayerColor(self._b.turn).name)
        return "\n".join(lines) + "\n"


class BoardPool:                        # board.h:142-203 (vestigial in the reference: always allocates)
    def __init__(self, poolSize):
        self.poolSize = poolSize

    def acquire(self, templateBoard):
        return templateBoard._copy()

    def release(self, board):
        pass


def _set_statics():
    A_ch = 8 * _R + 8
    Board.state_space_size = 24 * _R * _R
    Board.num_action_channels = A_ch
    Board.action_space_size = A_ch * _R * _R
    Board.action_space_dims = (A_ch, _R, _R)
    Board.state_space_dims = (24, _R, _R)
    Move.num_queen_moves_per_direction = _R - 1
    Move.num_queen_moves = 8 * (_R - 1)
    Move.num_knight_moves = 8


_set_statics()
__all__ += ["piece_value", "color_value", "Player", "Piece", "BoardLocation", "CastlingRights", "PlacedPiece", "Move",
            "MemoryEntry", "Node", "Board", "BoardPool", "configure", "engine"]
