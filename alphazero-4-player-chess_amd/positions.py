"""Start positions and the chess.com 4-player FEN codec, host-side (pure Python, no engine).

Counterpart of the reference's src/py/start_fens.py:1-67 (five layouts: STANDARD 14x14, THIRTEEN,
TEN, EIGHT, EIGHT_SIMPLE) and src/py/fen_parser.py:104-170 (FEN -> (turn, {location: piece})).
Layouts are kept here as compact per-side back-rank specs and rendered to FEN4 text on demand.

Quirk kept: the reference parses the castling fields and then drops them
(fen_parser.py:137-140,170 returns only (player, location_to_piece)), so every board built from a
FEN has all castling rights false.  `parse_fen` reports the rights it read, `board_args_from_fen`
returns exactly what the reference returns.
"""

RED, BLUE, YELLOW, GREEN = 0, 1, 2, 3
PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING = 0, 1, 2, 3, 4, 5
_TYPE_OF = {"P": PAWN, "N": KNIGHT, "B": BISHOP, "R": ROOK, "Q": QUEEN, "K": KING}
_CHAR_OF = {v: k for k, v in _TYPE_OF.items()}
_COLOUR_OF = {"r": RED, "b": BLUE, "y": YELLOW, "g": GREEN}
_CCHAR_OF = {v: k for k, v in _COLOUR_OF.items()}

# (size, corner, yellow back rank left->right, blue back file top->bottom,
#  green back file top->bottom, red back rank left->right, pawns?)  -- '.' = empty square
_LAYOUTS = {
    "STANDARD": (14, 3, "RNBKQBNR", "RNBQKBNR", "RNBKQBNR", "RNBQKBNR", True),
    "THIRTEEN": (13, 3, "RNBKQBN", "RNBQKBN", "RNBKQBN", "RNBQKBN", True),
    "TEN": (10, 2, "RNKQBR", "RNQKBR", "RNKQBR", "RNQKBR", True),
    "EIGHT": (8, 2, "RKQR", "RQKR", "RKQR", "RQKR", True),
}


def _grid_from_layout(name):
    size, inv, yrank, bfile, gfile, rrank, _ = _LAYOUTS[name]
    g = [[None] * size for _ in range(size)]
    n = size - 2 * inv
    for k in range(n):
        g[0][inv + k] = (YELLOW, _TYPE_OF[yrank[k]])
        g[1][inv + k] = (YELLOW, PAWN)
        g[size - 1][inv + k] = (RED, _TYPE_OF[rrank[k]])
        g[size - 2][inv + k] = (RED, PAWN)
        g[inv + k][0] = (BLUE, _TYPE_OF[bfile[k]])
        g[inv + k][1] = (BLUE, PAWN)
        g[inv + k][size - 1] = (GREEN, _TYPE_OF[gfile[k]])
        g[inv + k][size - 2] = (GREEN, PAWN)
    return size, inv, g


def _eight_simple():
    """the reduced 16-piece 8x8 position the snapshot trains on (four_player_chess_board.py:18)"""
    size, inv = 8, 2
    g = [[None] * size for _ in range(size)]
    put = lambda r, c, col, t: g[r].__setitem__(c, (col, t))
    put(0, 2, YELLOW, ROOK); put(0, 3, YELLOW, KING); put(0, 5, YELLOW, ROOK)
    put(1, 3, YELLOW, PAWN); put(1, 4, YELLOW, PAWN); put(1, 5, YELLOW, PAWN)
    put(3, 6, GREEN, PAWN); put(3, 7, GREEN, KING)
    put(4, 0, BLUE, KING); put(4, 1, BLUE, PAWN)
    put(6, 2, RED, PAWN); put(6, 3, RED, PAWN); put(6, 4, RED, PAWN)
    put(7, 2, RED, ROOK); put(7, 4, RED, KING); put(7, 5, RED, ROOK)
    return size, inv, g


def _is_corner(size, inv, r, c):
    return (r < inv or r >= size - inv) and (c < inv or c >= size - inv)


def render_fen(name):
    """FEN4 text ('R-0,0,0,0-1,1,1,1-1,1,1,1-0,0,0,0-0-<rows>') of a named layout."""
    size, inv, g = _eight_simple() if name == "EIGHT_SIMPLE" else _grid_from_layout(name)
    rows = []
    for r in range(size):
        toks, run = [], 0
        for c in range(size):
            if _is_corner(size, inv, r, c):
                if run:
                    toks.append(str(run)); run = 0
                toks.append("x")
            elif g[r][c] is None:
                run += 1
            else:
                if run:
                    toks.append(str(run)); run = 0
                toks.append(_CCHAR_OF[g[r][c][0]] + _CHAR_OF[g[r][c][1]])
        if run:
            toks.append(str(run))
        rows.append(",".join(toks))
    return "R-0,0,0,0-1,1,1,1-1,1,1,1-0,0,0,0-0-" + "/".join(rows)


NAMES = ("STANDARD", "THIRTEEN", "TEN", "EIGHT", "EIGHT_SIMPLE")
STANDARD, THIRTEEN, TEN, EIGHT, EIGHT_SIMPLE = (render_fen(n) for n in NAMES)


def default_fen(board_size):
    """start position the reference uses for a compiled board size"""
    return {8: EIGHT_SIMPLE, 10: TEN, 13: THIRTEEN, 14: STANDARD}[board_size]


def parse_fen(fen, board_size=None):
    """-> (turn colour, [(row, col, colour, type), ...] in the reference's dict insertion order
    (rows top->bottom, columns left->right, fen_parser.py:144-168), kingside[4], queenside[4])."""
    fen = fen.replace("\n", "")
    parts = fen.split("-")
    if len(parts[0]) != 1 or parts[0] not in "RBYG":
        raise ValueError("Invalid player character in FEN string")
    turn = "RBYG".index(parts[0])

    def rights(s, what):
        f = s.split(",")
        if len(f) != 4:
            raise ValueError("Invalid %s castling availability in FEN string" % what)
        return [x == "1" for x in f]

    kingside, queenside = rights(parts[2], "kingside"), rights(parts[3], "queenside")
    pieces = []
    for r, row in enumerate(parts[-1].split("/")):
        c = 0
        for tok in row.split(","):
            if not tok:
                raise ValueError("Empty column string in piece placement")
            if tok[0] in _COLOUR_OF:
                if len(tok) != 2:
                    raise ValueError("Piece placement string for player must be of length 2")
                pieces.append((r, c, _COLOUR_OF[tok[0]], _TYPE_OF[tok[1]]))
                c += 1
            elif tok[0] == "x":
                c += 1
            else:
                try:
                    n = int(tok)
                except ValueError:
                    n = 0
                if n <= 0:
                    raise ValueError("Invalid number of empty spaces in piece placement")
                c += n
    return turn, pieces, kingside, queenside


def start_entries(board_size, fen=None):
    """[(sq, colour, type), ...] + turn, ready for fpc_board_from_dict."""
    turn, pieces, _, _ = parse_fen(fen or default_fen(board_size))
    return turn, [(r * board_size + c, col, typ) for r, c, col, typ in pieces]
