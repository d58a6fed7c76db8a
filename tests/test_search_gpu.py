"""The native search driving the GPU evaluators (BASELINE configs[4]): same values as full negamax
over the same GPU evaluation, with and without the move-scoring network ordering the moves."""
import numpy as np
import pytest

from batch_first_amd import search as S, weights as W
from tests_fens import FENS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def predictor():
    from batch_first_amd import chestimator
    bp, _, closer = chestimator.get_inference_functions(W.make_weights(seed=1234, bn="perturbed"),
                                                        policy_weights=W.make_policy_weights(seed=4321))
    yield bp
    closer()


@pytest.mark.parametrize("use_policy", [True, False])
def test_gpu_search_equals_negamax_over_the_gpu_evaluation(predictor, use_policy):
    import ctypes
    from batch_first_amd import _lib as hip
    eval_ptr = ctypes.cast(hip.load().bf_eval_structs_host, ctypes.c_void_p).value
    s = S.Search.on_gpu(predictor, use_policy=use_policy, max_batch_size=512, tt_bits=18)
    for fen in (FENS[1], FENS[4], FENS[6]):
        for depth in (1, 2):
            ref = S.negamax(fen, depth, eval_ptr, predictor._handle())
            value, move = s.mtd_f(fen, depth)
            assert value == ref, (fen, depth, value, ref)
            assert any(np.array_equal(move, m) for m in S.legal_moves(fen))
    st = s.stats()
    assert st["boards_evaluated"] > 100
    assert (st["moves_scored"] > 0)
    s.close()


def test_gpu_search_depth_3_statistics(predictor):
    s = S.Search.on_gpu(predictor, max_batch_size=2048, tt_bits=20)   # the reference plays with 2048 (playing_chess.py:108)
    value, move = s.iterative_deepening(FENS[1], [1, 2, 3])
    st = s.stats()
    print("depth-3 search: value %.4f move %s, %d nodes expanded in %d iterations, %d boards evaluated (largest batch %d), "
          "%.3f s, %.0f nodes/s" % (value, move, st["nodes_out"], st["iterations"], st["boards_evaluated"], st["max_batch"],
                                    st["seconds"], st["nodes_out"] / st["seconds"]))
    assert np.isfinite(value) and st["max_batch"] <= 2048
    s.close()


def test_engine_wrapper_and_calibration(predictor):
    """engine.py: zero-shift and bin calibration feed the search; pick_move returns a legal move."""
    from batch_first_amd import engine, positions as P
    recs = S.random_playout_records(FENS[0], 100, seed=11)
    structs = np.frombuffer(recs.tobytes(), dtype=P.COMPACT_DTYPE)
    shift = engine.calculate_eval_zero_shift(structs, predictor, max_batch_size=32)
    assert shift.dtype == np.float32 and abs(shift - predictor.eval_structs(structs).mean()) < 1e-5
    bins, move_shift = engine.generate_bin_ranges(structs, predictor, percentiles=np.arange(0, 100, 1.0), max_batch_size=40)
    assert bins.shape == (100,) and np.all(np.diff(bins) >= 0) and np.isfinite(move_shift)
    eng = engine.BatchFirstEngine(3, predictor, bins=bins, move_zero_shift=move_shift, eval_zero_shift=shift,
                                  max_batch_size=512, tt_bits=18)
    eng.start_new_game()
    move, value = eng.pick_move(FENS[1])
    assert any(np.array_equal(move, m) for m in S.legal_moves(FENS[1])) and np.isfinite(value)
    assert len(engine.move_to_uci(move)) in (4, 5)
    # same position again from a cleared table: same move and value (the search is deterministic)
    eng.start_new_game()
    move2, value2 = eng.pick_move(FENS[1])
    assert np.array_equal(move, move2) and value == value2
    eng.release_resources()
