"""The engine wrapper and its calibration helpers, mirroring batch_first/engine.py.

* :func:`calculate_eval_zero_shift` -- engine.py:92-151: the mean evaluation of positions whose
  expected value is 0; the search subtracts it (engine.py:260) because negamax needs draws at 0.
* :func:`generate_bin_ranges` -- engine.py:9-89: percentiles of |move score - mean| over a sample of
  positions, the edges of the open-node priority bins, and that mean (the bins' zero shift).
* :class:`BatchFirstEngine` -- engine.py:211-288: ``pick_move`` = iterative-deepening mtd(f) to
  ``search_depth`` on the native search driver with the GPU evaluators.

python-chess is not used: positions come and go as FEN strings, moves as (from, to, promotion)
triples or UCI strings.
"""
import numpy as np

from . import moves as _moves
from . import search as _search


def calculate_eval_zero_shift(struct_array, board_predictor, max_batch_size=5000):
    """Mean evaluation over ``struct_array`` (node records whose expected value is 0), evaluated in
    batches of ``max_batch_size`` like engine.py:111-135.  Returns np.float32."""
    structs = np.ascontiguousarray(struct_array)
    parts = [board_predictor.eval_structs(structs[i:i + max_batch_size]) for i in range(0, len(structs), max_batch_size)]
    return np.float32(np.mean(np.concatenate(parts))) if parts else np.float32(0)


def generate_bin_ranges(struct_array, board_predictor, percentiles=None, max_batch_size=1000):
    """(bins, zero_shift) from the move-scoring network over every legal move of the sample."""
    if percentiles is None:
        percentiles = np.arange(0, 100, .02)   # engine.py:56
    structs = np.ascontiguousarray(struct_array)
    raw = structs.view(np.uint8).reshape(len(structs), -1)
    results = []
    for lo in range(0, len(structs), max_batch_size):
        chunk = structs[lo:lo + max_batch_size]
        mv = [_search.legal_moves_of_record(raw[lo + i]) for i in range(len(chunk))]
        per = np.array([len(m) for m in mv], np.uint8)
        if per.sum() == 0:
            continue
        from .encoder import struct_array_to_ann_inputs
        nib, occ = struct_array_to_ann_inputs(chunk, chunk[:0], np.ones(len(chunk), bool), np.zeros(0, bool), len(chunk))
        finish = board_predictor.start_move_prediction(nib, occ)
        frm, flt, per = _moves.move_info_for(chunk["turn"], per, np.concatenate([m for m in mv if len(m)]))
        results.append(finish([frm, flt, per]))
    scores = np.concatenate(results)
    zero_shift = np.float32(np.mean(scores))
    return np.percentile(np.abs(scores - zero_shift), percentiles).astype(np.float32), zero_shift


def move_to_uci(move):
    frm, to, promo = (int(x) for x in move)
    s = "abcdefgh"[frm & 7] + str((frm >> 3) + 1) + "abcdefgh"[to & 7] + str((to >> 3) + 1)
    return s + ("", "", "n", "b", "r", "q")[promo] if promo else s


class BatchFirstEngine:
    """engine.py:211-288 on the native search driver."""

    def __init__(self, search_depth, board_predictor, bins=None, move_zero_shift=0.0, eval_zero_shift=0.0,
                 max_batch_size=5000, tt_bits=24, use_policy=True):
        self.search_depth = int(search_depth)
        if bins is None:
            bins = np.linspace(0, 4000, 1000)
        self._search = _search.Search.on_gpu(board_predictor, use_policy=use_policy, max_batch_size=max_batch_size,
                                             tt_bits=tt_bits, bin_edges=bins, bin_zero_shift=float(move_zero_shift),
                                             eval_zero_shift=float(eval_zero_shift))

    def start_new_game(self):
        self._search.clear_tt()          # engine.py:271-272

    def pick_move(self, fen):
        """Returns (move triple, value); the reference returns a python-chess Move (engine.py:274-288)."""
        value, move = self._search.iterative_deepening(fen, list(range(1, self.search_depth + 1)))
        return move, value

    def stats(self, reset=False):
        return self._search.stats(reset)

    def release_resources(self):
        self._search.close()
