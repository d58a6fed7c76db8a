"""The native search driver: move generation against the reference's perft tables, make-move and
hash consistency, and the reference's own search check -- a zero-window search must land on the
right side of the full negamax value (code_testing.py:435-496) -- with the reference's fake
evaluators (material count, linspace move scores; code_testing.py:516-545, :634-640)."""
import numpy as np
import pytest

from batch_first_amd import search as S
from oracle import encoding as E

# code_testing.py:33-49 (ChessProgramming-wiki perft positions and counts)
FENS = ["rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1",
        "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1",
        "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1",
        "r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1",
        "r2q1rk1/pP1p2pp/Q4n2/bbp1p3/Np6/1B3NBn/pPPP1PPP/R3K2R b KQ - 0 1",
        "rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8 ",
        "r4rk1/1pp1qppp/p1np1n2/2b1p1B1/2B1P1b1/P1NP1N2/1PP1QPPP/R4RK1 w - - 0 10"]
PERFT = [[20, 400, 8902, 197281, 4865609], [48, 2039, 97862, 4085603, 193690690], [14, 191, 2812, 43238, 674624, 11030083],
         [6, 264, 9467, 422333, 15833292], [6, 264, 9467, 422333, 15833292], [44, 1486, 62379, 2103487, 89941194],
         [46, 2079, 89890, 3894594, 164075551]]


def test_perft_tables():
    """full_perft_tester (code_testing.py:64-83), capped at 5 M nodes as the reference does (:65)."""
    for fen, counts in zip(FENS, PERFT):
        for depth, expected in enumerate(counts, 1):
            if expected <= 5000000:
                assert S.perft(fen, depth) == expected, (fen, depth)


def test_push_keeps_incremental_hash_and_fen_semantics():
    # bfs_push recomputes the hash from scratch and fails if the incremental one differs
    for fen in FENS:
        for mv in S.legal_moves(fen):
            nxt = S.push(fen, mv)
            for mv2 in S.legal_moves(nxt)[:6]:
                S.push(nxt, mv2)
    start = FENS[0]
    after = S.push(start, [12, 28, 0])                      # 1. e4
    assert after.startswith("rnbqkbnr/pppppppp/8/8/4P3/8/PPPP1PPP/RNBQKBNR b KQkq e3 0")   # ep set after every double push
    castled = S.push("r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1", [4, 6, 0])
    assert castled.startswith("r3k2r/8/8/8/8/8/8/R4RK1 b kq - 1")
    promoted = S.push("8/P6k/8/8/8/8/8/K7 w - - 0 1", [48, 56, 2])
    assert promoted.startswith("N7/7k/8/8/8/8/8/K7 b")
    ep = S.push("8/8/8/3pP3/8/8/8/K6k w - d6 0 1", [36, 43, 0])
    assert ep.startswith("8/8/3P4/8/8/8/8/K6k b")
    with pytest.raises(RuntimeError):
        S.push(start, [12, 36, 0])


def test_records_feed_the_encoder():
    """A record made by the search library is what the evaluation path reads (bytes 0..81)."""
    from batch_first_amd import positions as P
    rec = S.record_from_fen(FENS[0])
    structs = np.frombuffer(rec.tobytes(), dtype=P.COMPACT_DTYPE)
    nib, occ = E.struct_array_to_ann_inputs(structs)
    assert occ[0] == np.uint64(0xFFFF00000000FFFF)
    assert nib.tobytes().hex() == "7542145766666666ddddddddecb98bce"   # SURVEY.md 8(c) hand vector
    recs = S.random_playout_records(FENS[0], 60, seed=3)
    assert len(recs) > 20
    s2 = np.frombuffer(recs.tobytes(), dtype=P.COMPACT_DTYPE)
    nib2, occ2 = E.struct_array_to_ann_inputs(s2)
    assert all(E.popcount(o) <= 33 for o in occ2)
    assert np.array_equal(E.decode_planes_fast(nib2, occ2).sum(axis=(1, 2, 3)), [E.popcount(o) for o in occ2])


def test_material_eval_matches_reference_fake():
    from batch_first_amd import positions as P
    recs = S.random_playout_records(FENS[1], 80, seed=5)
    structs = np.frombuffer(recs.tobytes(), dtype=P.COMPACT_DTYPE)
    ref = E.bf_piece_sum_eval(*E.struct_array_to_ann_inputs(structs))
    import ctypes
    out = np.zeros(len(recs), np.float32)
    done = ctypes.c_size_t(0)
    fn = S.BOARD_EVAL_FN(S.material_eval_pointer())
    assert fn(None, recs.ctypes.data, 88, None, len(recs), out.ctypes.data, ctypes.byref(done)) == 0
    assert done.value == len(recs)
    assert np.array_equal(out, ref.astype(np.float32))


@pytest.mark.parametrize("max_batch", [5000, 7])
def test_zero_window_search_against_negamax(max_batch):
    """zero_window_search_tester, code_testing.py:435-496."""
    rng = np.random.default_rng(1)
    for fen in FENS:
        s = S.Search(max_batch_size=max_batch, tt_bits=18)
        for depth in (1, 2, 3):
            minimax = S.negamax(fen, depth)
            for _ in range(3):
                above = np.float32(minimax + rng.random() * 1200)
                if above == minimax:
                    above = np.nextafter(np.nextafter(np.float32(minimax), np.float32(np.inf)), np.float32(np.inf))
                got = s.zero_window(fen, depth, above)
                assert minimax <= got < above, (fen, depth, minimax, above, got)
                below = np.float32(minimax - rng.random() * 1200)
                if below == minimax:
                    below = np.nextafter(np.nextafter(np.float32(minimax), np.float32(-np.inf)), np.float32(-np.inf))
                got = s.zero_window(fen, depth, below)
                assert below <= got <= minimax, (fen, depth, minimax, below, got)
        s.close()


def test_mtd_f_finds_the_negamax_value_and_a_legal_move():
    for fen in FENS[:4] + FENS[5:]:
        s = S.Search(max_batch_size=256, tt_bits=18)
        for depth in (1, 2, 3):
            value, move = s.mtd_f(fen, depth)
            assert value == S.negamax(fen, depth), (fen, depth)
            assert any(np.array_equal(move, m) for m in S.legal_moves(fen)), (fen, depth, move)
        st = s.stats()
        assert st["boards_evaluated"] > 0 and st["nodes_out"] > 0 and st["iterations"] > 0
        s.close()
    # black to move: the chosen move comes back in real squares
    fen = FENS[4]
    s = S.Search(max_batch_size=256, tt_bits=18)
    value, move = s.iterative_deepening(fen, [1, 2, 3])
    assert value == S.negamax(fen, 3)
    assert any(np.array_equal(move, m) for m in S.legal_moves(fen))
    s.close()


def test_python_callable_as_evaluator():
    """Any callable with the wire signature plugs in, as bf_piece_sum_eval does in the reference."""
    from batch_first_amd import positions as P
    import ctypes
    calls = []

    def board_eval(user, records, stride, mask, n, scores, n_scored):
        raw = np.ctypeslib.as_array(ctypes.cast(records, ctypes.POINTER(ctypes.c_uint8)), shape=(n * stride,))
        structs = np.frombuffer(raw.tobytes(), dtype=P.COMPACT_DTYPE)
        vals = E.bf_piece_sum_eval(*E.struct_array_to_ann_inputs(structs)).astype(np.float32)
        ctypes.memmove(scores, vals.ctypes.data, 4 * n)
        n_scored[0] = n
        calls.append(n)
        return 0

    s = S.Search(board_eval=board_eval, max_batch_size=64, tt_bits=16)
    value, _ = s.mtd_f(FENS[1], 2)
    assert value == S.negamax(FENS[1], 2)
    assert calls and max(calls) <= 64
    s.close()


def test_legality_sweep_all_from_to_promotion_triples():
    """move_verification_tester (code_testing.py:266-293) on DEFAULT_MOVES_TO_CHECK (:55-59): all
    64 * 63 * 5 (from, to, promotion) triples x the 7 positions.  The reference holds is_legal_move
    against python-chess, which is not installed; here two independent decisions must agree on every
    triple: membership in the generated legal moves (pinned by the perft tables above) and
    is_legal_direct, which judges the move alone (numba_board.py:1149-1151's construction)."""
    import itertools
    triples = np.array([[a, b, p] for (a, b) in itertools.permutations(range(64), 2) for p in (0, 2, 3, 4, 5)], np.uint8)
    assert triples.shape == (64 * 63 * 5, 3)
    for fen, counts in zip(FENS, PERFT):
        by_generator, direct = S.is_legal(fen, triples)
        assert np.array_equal(by_generator, direct), (fen, triples[by_generator != direct][:5])
        assert int(by_generator.sum()) == counts[0]          # every legal move is one of the triples
        legal = {tuple(m) for m in S.legal_moves(fen)}
        assert {tuple(t) for t in triples[by_generator]} == legal
    # moves the sweep cannot phrase: same square, out of range, king promotion
    g, d = S.is_legal(FENS[0], [[12, 12, 0], [12, 200, 0], [4, 12, 6], [12, 28, 5]])
    assert not g.any() and not d.any()


def test_legality_sweep_on_random_legal_positions():
    """The same agreement on positions from random playouts (en passant, promotions, checks, pins and
    castling through attacked squares turn up there), on the moves that matter: every generated move and
    every other triple from the squares of the side to move."""
    import itertools
    triples = np.array([[a, b, p] for (a, b) in itertools.permutations(range(64), 2) for p in (0, 2, 5)], np.uint8)
    fens = []
    for seed in range(1, 7):
        fen = FENS[seed % len(FENS)]
        for step in range(40):
            mv = S.legal_moves(fen)
            if len(mv) == 0:
                break
            fen = S.push(fen, mv[(seed * 7919 + step * 104729) % len(mv)])
            if step % 4 == 3:
                fens.append(fen)
    assert len(fens) >= 30
    for fen in fens:
        g, d = S.is_legal(fen, triples)
        assert np.array_equal(g, d), (fen, triples[g != d][:5])


def test_incremental_hash_over_random_lines():
    """zobrist_hash_test (code_testing.py:86-136): 1000 random lines of up to 20 plies from the start
    position (and 200 from each other test position): at every ply the hash carried along by push()
    equals the hash computed from scratch.  The VALUES are unpinned (python-chess's Polyglot array is
    not available offline, board.h); the update rules are what this pins."""
    n, bad = S.hash_walk(FENS[0], lines=1000, plies=20, seed=11)
    assert n > 19000 and bad == 0
    for i, fen in enumerate(FENS[1:]):
        n, bad = S.hash_walk(fen, lines=200, plies=20, seed=100 + i)
        assert n > 2000 and bad == 0, fen


# bf_piece_sum_eval's value table (code_testing.py:520-521), worked out by hand from the FEN text:
# mover's material minus opponent's, P N B R Q = 100 300 300 500 900 (a castling rook counts as a rook).
#   1 start 0 | 2 kiwipete 3900 - 3900 | 3 800 - 800 | 4 white 3900 (8P 2N 2B 2R Q) - black 3800 (7p ...) = +100
#   5 the mirror, black to move: +100 | 6 3700 - 3700 | 7 3900 - 3900
MATERIAL_KAT = [0, 0, 0, 100, 100, 0, 0]
EXTRA_KAT = [("8/P6k/8/8/8/8/8/K7 w - - 0 1", 100), ("8/P6k/8/8/8/8/8/K7 b - - 0 1", -100),
             ("4k3/8/8/8/8/8/8/R3K2R w KQ - 0 1", 1000), ("r3k2r/8/8/8/8/8/8/4K3 w kq - 0 1", -1000),
             ("4k3/8/8/3q4/8/8/8/1NB1K3 b - - 0 1", 300)]


def test_material_known_answers_native_and_wire_consumer():
    """The reference's fake evaluator, two ways, against hand-computed values: the native
    bfs_material_eval on node records and the oracle's restatement of bf_piece_sum_eval on the WIRE
    format (which proves nibble order, mover's frame and the code table end to end)."""
    import ctypes
    from batch_first_amd import positions as P
    lib = S.load()
    cases = list(zip(FENS, MATERIAL_KAT)) + EXTRA_KAT
    recs = np.stack([S.record_from_fen(f) for f, _ in cases])
    want = np.array([v for _, v in cases], np.float32)
    got = np.zeros(len(cases), np.float32)
    done = ctypes.c_size_t(0)
    assert lib.bfs_material_eval(None, recs.ctypes.data, 88, None, len(cases), got.ctypes.data, ctypes.byref(done)) == 0
    assert done.value == len(cases) and np.array_equal(got, want)
    structs = np.frombuffer(recs.tobytes(), dtype=P.COMPACT_DTYPE)
    nib, occ = E.struct_array_to_ann_inputs(structs)
    assert np.array_equal(E.bf_piece_sum_eval(nib, occ), want.astype(np.int64))


def test_push_record_matches_push_on_fens():
    for fen in FENS:
        rec = S.record_from_fen(fen)
        for mv in S.legal_moves(fen)[:12]:
            child = S.push_record(rec, mv)
            assert np.array_equal(child[:82], S.record_from_fen(S.push(fen, mv))[:82])
    with pytest.raises(RuntimeError):
        S.push_record(S.record_from_fen(FENS[0]), [12, 36, 0])


def test_search_does_the_same_thing_on_any_number_of_host_threads():
    """The per-node loops of a batch run on a thread pool (the reference's prange, numba_board.py:242-266);
    everything that orders nodes stays on one thread, so value, move and every counter are the same."""
    runs = []
    for threads in (1, 2, 5):
        s = S.Search(max_batch_size=64, tt_bits=16, threads=threads)   # small batches: many pool hand-overs
        assert s.threads == threads
        out = []
        for fen in FENS[:4]:
            value, move = s.iterative_deepening(fen, [1, 2, 3, 4])
            out.append((value, tuple(int(x) for x in move)))
        st = s.stats()
        s.close()
        runs.append((out, {k: st[k] for k in ("iterations", "nodes_out", "nodes_in", "boards_evaluated", "moves_scored",
                                              "max_batch", "eval_calls", "policy_calls")}))
    assert runs[0] == runs[1] == runs[2]
    assert runs[0][1]["eval_calls"] > 0 and runs[0][1]["boards_evaluated"] >= runs[0][1]["eval_calls"]


def test_wide_batches_do_the_same_thing_on_any_number_of_host_threads():
    """The same with batches large enough for the wide path (>= 320 nodes: the pool's second worker group, and from
    four threads on the tree update on the evaluation lane beside the hand-over of the move scores and the arena
    copy): value, move and every counter equal one thread's."""
    runs = []
    for threads in (1, 4, 8):
        s = S.Search(max_batch_size=2048, tt_bits=18, threads=threads)
        out = []
        for fen in FENS[:3]:
            value, move = s.iterative_deepening(fen, [1, 2, 3, 4])
            out.append((value, tuple(int(x) for x in move)))
        st = s.stats()
        s.close()
        assert st["wide_iterations"] > 10
        assert (st["overlapped_updates"] > 10) == (threads >= 4), (threads, st["overlapped_updates"])
        runs.append((out, {k: st[k] for k in ("iterations", "nodes_out", "nodes_in", "boards_evaluated", "moves_scored",
                                              "max_batch", "eval_calls", "policy_calls", "wide_iterations", "arena_peak")}))
    assert runs[0] == runs[1] == runs[2]


def test_stage_times_add_up_to_the_search_time():
    s = S.Search(max_batch_size=512, tt_bits=16, threads=2)
    s.iterative_deepening(FENS[1], [1, 2, 3, 4])
    st = s.stats()
    s.close()
    stages = sum(st[k] for k in ("expand_seconds", "settle_seconds", "policy_seconds", "value_wait_seconds",
                                 "update_seconds", "compact_seconds", "select_seconds"))
    assert 0 < stages <= st["seconds"] * 1.001
    assert stages >= 0.8 * st["seconds"]


def test_arena_limit_is_an_error_not_a_crash():
    """bfs_config.max_nodes is checked before the arena grows."""
    s = S.Search(max_batch_size=512, tt_bits=16, max_nodes=2000)
    with pytest.raises(RuntimeError, match="arena would exceed its limit of 2000 nodes"):
        s.iterative_deepening(FENS[1], [1, 2, 3, 4, 5])
    # the instance stays usable
    value, move = s.iterative_deepening(FENS[2], [1, 2])
    assert S.is_legal(FENS[2], [move])[0][0]
    s.close()
    with pytest.raises(RuntimeError, match="threads"):
        S.Search(threads=1000)
    with pytest.raises(RuntimeError, match="max_nodes .* exceeds the limit"):   # node ids are ints
        S.Search(max_nodes=2 ** 31)
