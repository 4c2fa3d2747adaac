"""`parse_board_args_from_fen` -- drop-in for the reference's src/py/fen_parser.py:104-170:
chess.com 4-player FEN -> (Player, {BoardLocation: Piece}), the arguments of Board(...).
The castling fields are validated and then dropped, exactly like the reference."""
import positions
from alphazero_cpp import BoardLocation, Piece, PieceType, Player, PlayerColor


def parse_board_args_from_fen(fen, board_size):
    turn, pieces, _kingside, _queenside = positions.parse_fen(fen, board_size)
    location_to_piece = {}
    for row, col, colour, ptype in pieces:
        location_to_piece[BoardLocation(row, col)] = Piece(Player(PlayerColor(colour)), PieceType(ptype))
    return Player(PlayerColor(turn)), location_to_piece
