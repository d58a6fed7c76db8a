"""Move encoding of the move-scoring network, restated from the reference without python-chess.

* :data:`MOVE_FILTER_LOOKUP` -- batch_first/__init__.py:51-95 (``generate_move_filter_table``):
  ``[from_square, to_square, promotion] -> policy plane`` in the 73-plane encoding of
  "Mastering Chess and Shogi by Self-Play ..." (arXiv:1712.01815): 56 queen-move planes
  (8 directions x 7 distances), 8 knight planes, 9 under-promotion planes.
* :data:`SQUARES_180` -- batch_first/__init__.py (``chess.SQUARES_180``): the vertical mirror
  ``sq ^ 56`` that brings a black move into the mover's frame
  (numba_negamax_zero_window.py:512-515).
* :func:`move_info_for` -- what get_move_from_and_filter_squares_and_sizes builds
  (numba_negamax_zero_window.py:485-527): ``(from_squares, move_filters, moves_per_board)``.
"""
import numpy as np

SQUARES_180 = np.arange(64, dtype=np.uint8) ^ np.uint8(56)


def generate_move_filter_table():
    diffs = {}
    for j in range(1, 8):
        diffs[(0, j)] = j - 1
        diffs[(0, -j)] = j + 6
        diffs[(j, 0)] = j + 13
        diffs[(-j, 0)] = j + 20
        diffs[(j, j)] = j + 27
        diffs[(j, -j)] = j + 34
        diffs[(-j, j)] = j + 41
        diffs[(-j, -j)] = j + 48
    diffs[(1, 2)] = 56
    diffs[(1, -2)] = 57
    diffs[(-1, 2)] = 58
    diffs[(-1, -2)] = 59
    diffs[(2, 1)] = 60
    diffs[(2, -1)] = 61
    diffs[(-2, 1)] = 62
    diffs[(-2, -1)] = 63
    table = np.zeros([64, 64, 6], dtype=np.uint8)
    for sq1 in range(64):
        for sq2 in range(64):
            file_diff = (sq2 & 7) - (sq1 & 7)
            rank_diff = (sq2 >> 3) - (sq1 >> 3)
            plane = diffs.get((file_diff, rank_diff))
            if plane is not None:
                table[sq1, sq2] = plane
                if rank_diff == 1 and file_diff in (1, 0, -1):
                    # promotion piece 2, 3, 4 = knight, bishop, rook: planes 64..72
                    table[sq1, sq2, 2:5] = 3 * (1 + file_diff) + np.arange(64, 67)
    return table


MOVE_FILTER_LOOKUP = generate_move_filter_table()


def move_info_for(turns, moves_per_board, moves):
    """``moves``: uint8[M, 3] (from, to, promotion) in board order, real squares; ``turns``: one per
    board (non-zero = white to move).  Returns the three uint8 arrays stage 2 is fed."""
    moves = np.asarray(moves, np.uint8).reshape(-1, 3)
    moves_per_board = np.asarray(moves_per_board, np.uint8)
    white = np.repeat(np.asarray(turns) != 0, moves_per_board)
    frm = np.where(white, moves[:, 0], SQUARES_180[moves[:, 0]])
    to = np.where(white, moves[:, 1], SQUARES_180[moves[:, 1]])
    return frm.astype(np.uint8), MOVE_FILTER_LOOKUP[frm, to, moves[:, 2]], moves_per_board
