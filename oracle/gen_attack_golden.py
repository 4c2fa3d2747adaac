#!/usr/bin/env python3
"""oracle/gen_attack_golden.py -- TEST INFRASTRUCTURE ONLY.

Third golden generator (round 5): the attacked-square queries of the reference's binding surface
(/root/reference/src/cpp/wrapper.cpp:201-206: GetSimpleState, GetAttackedSquaresPlayers, GetAttackedSquaresTeams,
IsAttackedByPlayer), which feed its pygame reviewer.  Runs only in the build container, imports the real reference
(oracle/_ref/r{8,14} + /root/reference/src/py) and writes DATA ONLY:

  tests/golden/ref_attack_r{R}.json.gz   for positions of the recorded golden playouts (every 5th ply): the position
      (turn + ordered piece lists, as in ref_r{R}.json.gz), the squares GetAttackedSquaresPlayers / GetAttackedSquaresTeams
      report per colour / team (in the order the reference returns them), IsAttackedByPlayer for every (square, colour)
      of a few positions, and what GetSimpleState carries (turn, attackedSquares keys and lengths).

    python oracle/gen_attack_golden.py --size 8
    python oracle/gen_attack_golden.py --size 14
"""
import argparse
import gzip
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import gen_golden  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, required=True, choices=[8, 14])
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    gen_golden.setup_imports(args.size)
    import alphazero_cpp as az
    from four_player_chess_board import FourPlayerChess

    R = az.Board.nRows()
    assert R == args.size
    with gzip.open(os.path.join(args.out, "ref_r%d.json.gz" % R), "rt") as f:
        gold = json.load(f)

    def board_from_snapshot(snap):
        l2p = {}
        for colour, col in enumerate(snap["pl"]):
            for sq, typ in col:
                l2p[az.BoardLocation(sq // R, sq % R)] = az.Piece(az.PlayerColor(colour), az.PieceType(typ))
        return FourPlayerChess(az.Player(az.PlayerColor(snap["turn"])), l2p)

    def sqs(locs):
        return [int(l.GetRow()) * R + int(l.GetCol()) for l in locs]

    cases = []
    for gi, game in enumerate(gold["playouts"]):
        for pi in range(0, len(game), 5):
            snap = game[pi]["before"]
            b = board_from_snapshot(snap)
            players = b.GetAttackedSquaresPlayers()
            teams = b.GetAttackedSquaresTeams()
            rec = {"pos": [gi, pi], "turn": snap["turn"], "pl": snap["pl"],
                   "players": {str(int(k)): sqs(v) for k, v in players.items()},
                   "teams": {str(int(k)): sqs(v) for k, v in teams.items()}}
            if len(cases) % 8 == 0:       # the single-square query, every (square, colour)
                rec["by_player"] = [[1 if b.IsAttackedByPlayer(az.BoardLocation(sq // R, sq % R), az.PlayerColor(c)) else 0
                                     for sq in range(R * R)] for c in range(4)]
                st = b.GetSimpleState()
                rec["simple"] = {"turn": int(st.turn.GetColor()),
                                 "pieces": [[[int(pp.GetLocation().GetRow()) * R + int(pp.GetLocation().GetCol()), int(pp.GetPiece().GetPieceType())]
                                             for pp in col] for col in st.pieces],
                                 "attacked": {str(int(k)): sqs(v) for k, v in st.attackedSquares.items()}}
            cases.append(rec)
    path = os.path.join(args.out, "ref_attack_r%d.json.gz" % R)
    with gzip.open(path, "wt") as f:
        json.dump({"R": R, "source": "reference src/cpp/board.cpp:50-57, :120-232 through wrapper.cpp:201-206", "cases": cases}, f)
    print("wrote", path, os.path.getsize(path), "bytes;", len(cases), "positions,", sum("by_player" in c for c in cases), "with the per-square query")


if __name__ == "__main__":
    main()
