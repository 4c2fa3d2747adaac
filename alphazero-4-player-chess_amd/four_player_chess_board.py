"""`FourPlayerChess` -- drop-in for the reference's src/py/four_player_chess_board.py:17-55."""
import torch

import positions
from alphazero_cpp import Board as BoardCpp
from alphazero_cpp import engine


class _StartFen:
    """class attribute that follows the configured board size (EIGHT_SIMPLE at 8x8, as in the
    reference's four_player_chess_board.py:18; STANDARD at 14x14)"""

    def __get__(self, obj, cls):
        return positions.default_fen(BoardCpp.nRows())


class FourPlayerChess(BoardCpp):
    start_fen = _StartFen()

    @classmethod
    def init_tensors(cls, max_moves, batch_size, device):
        """kept for API compatibility: the dense mask buffers the reference pre-allocates here are
        never materialised by the engine (movegen emits indices)."""

    @classmethod
    def get_legal_moves_mask(cls, states, device):
        """dense 0/1 mask [B, A_ch, R, R] f32 in absolute board coordinates
        (four_player_chess_board.py:36-56); GetLegalMoves' piece-list side effect included."""
        m = torch.from_numpy(engine().legal_mask([s._b for s in states]))
        return m if str(device) == "cpu" else m.to(device)
