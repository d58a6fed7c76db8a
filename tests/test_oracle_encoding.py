"""Pins the encoding oracle with the reference's own test rule and fixed positions
(code_testing.py:184-263: two independent derivations of the plane per square must agree)."""
import numpy as np
import pytest

from batch_first_amd import positions as P
from oracle import encoding as E

ALL_FENS = list(P.REFERENCE_TEST_FENS) + list(P.EXTRA_FENS)


def _features_from_wire(structs):
    nib, occ = E.struct_array_to_ann_inputs(structs)
    planes = E.parse_into_ann_input_inference(nib, occ)
    # every square's plane vector sums to 0 or 1 (code_testing.py:221-225)
    sums = planes.sum(axis=1)
    assert np.all((sums == 0) | (sums == 1))
    mb = E.planes_to_mailbox(planes).reshape(-1, 8, 8)
    return np.where(mb == 15, 0, mb + 1)  # 1-based, 0 = unoccupied (code_testing.py:227-233)


def test_start_position_known_answer():
    # hand-derived vector, SURVEY.md 8(c) (5)
    s = P.structs_from_fens([P.REFERENCE_TEST_FENS[0]])
    nib, occ = E.struct_array_to_ann_inputs(s)
    assert occ[0] == np.uint64(0xFFFF00000000FFFF)
    assert bytes(nib).hex() == "7542145766666666ddddddddecb98bce"


def test_black_to_move_ep_known_answer():
    s = P.structs_from_fens([P.EXTRA_FENS[0]])  # after 1.e4
    nib, occ = E.struct_array_to_ann_inputs(s)
    real = int(s[0]["occupied"]) | (1 << 20)
    assert int(occ[0]) == int.from_bytes(real.to_bytes(8, "little"), "big")
    codes = E.decompress_squares(nib, E.popcount(occ[0]))
    squares = list(E.scan_reversed(occ[0]))
    assert codes[squares.index(44)] == 0          # e3 seen from black's side is square 44
    assert codes[squares.index(0)] == 7           # black's a8 rook: mover's castling rook on a1'
    assert codes[squares.index(63)] == 14         # white's h1 rook: opponent's castling rook


@pytest.mark.parametrize("fen", ALL_FENS)
def test_reference_fens_two_derivations_agree(fen):
    got = _features_from_wire(P.structs_from_fens([fen]))[0]
    exp = E.expected_features(P.parse_fen(fen))
    assert np.array_equal(got, exp), fen


def test_mirrored_pair_has_identical_input():
    # code_testing.py:36-37: FEN 5 is FEN 4 with colours and side to move swapped
    s = P.structs_from_fens([P.REFERENCE_TEST_FENS[3], P.REFERENCE_TEST_FENS[4]])
    a = E.struct_array_to_ann_inputs(s[:1])
    b = E.struct_array_to_ann_inputs(s[1:])
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_random_positions_two_derivations_agree():
    structs = P.random_structs(300, seed=7)
    got = _features_from_wire(structs)
    for j in range(len(structs)):
        rec = structs[j]
        piece_at = {}
        for name, ptype in (("pawns", 1), ("knights", 2), ("bishops", 3), ("rooks", 4), ("queens", 5), ("kings", 6)):
            for sq in E.scan_reversed(rec[name]):
                piece_at[sq] = (ptype, bool(int(rec["occupied_co"][1]) >> sq & 1))
        pos = {"piece_at": piece_at, "turn": bool(rec["turn"]), "castling": int(rec["castling_rights"]),
               "ep": int(rec["ep_square"]) or None}
        assert np.array_equal(got[j], E.expected_features(pos)), j


def test_uneven_splits_concatenate(golden):
    # code_testing.py:242-250 feeds the boards in uneven splits
    recs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)
    edges = [0, 1, 2, 9, 40, 41, 300, 1024]
    planes = [E.decode_planes_fast(*E.struct_array_to_ann_inputs(recs[a:b])) for a, b in zip(edges[:-1], edges[1:])]
    full = E.decode_planes_fast(golden["piece_filters"], golden["occupied_bbs"])
    assert np.array_equal(np.concatenate(planes), full)


def test_fast_decoder_equals_literal(golden):
    a = E.parse_into_ann_input_inference(golden["piece_filters"][:2000], golden["occupied_bbs"][:64])
    nib, occ = golden["piece_filters"], golden["occupied_bbs"][:64]
    b = E.decode_planes_fast(nib, occ)
    assert np.array_equal(a, b)


def test_golden_wire_matches_oracle(golden):
    recs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)
    nib, occ = E.struct_array_to_ann_inputs(recs)
    assert np.array_equal(nib, golden["piece_filters"])
    assert np.array_equal(occ, golden["occupied_bbs"])
    mb = E.planes_to_mailbox(E.decode_planes_fast(nib, occ))
    assert np.array_equal(mb, golden["mailbox"])


def test_material_evaluator_consumes_wire_format():
    # bf_piece_sum_eval (code_testing.py:523-532) is the reference's own consumer of the format
    s = P.structs_from_fens(ALL_FENS)
    vals = E.bf_piece_sum_eval(*E.struct_array_to_ann_inputs(s))
    assert vals[0] == 0
    assert vals[3] == -vals[4] or vals[3] == vals[4]  # mirrored pair, both from the mover's side
    assert vals[3] == vals[4]
    assert vals[9] == 0  # bare kings


def test_selection_masks_and_order():
    structs = P.test_set(40)
    a, b = structs[:25], structs[25:]
    ma = np.arange(25) % 3 != 0
    mb = np.arange(15) % 2 == 0
    nib, occ = E.struct_array_to_ann_inputs(a, b, ma, mb, int(ma.sum() + mb.sum()))
    nib2, occ2 = E.struct_array_to_ann_inputs(np.concatenate([a[ma], b[mb]]))
    assert np.array_equal(nib, nib2) and np.array_equal(occ, occ2)


def test_legal_playout_fixture_is_consistent():
    """tests/golden/legal_1024.npz: the stored wire format is what the oracle encoder makes of the
    stored records, every board passes the reference's plane rule, and the native host encoder agrees."""
    import os
    from batch_first_amd import encoder, positions as P
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "legal_1024.npz")) as z:
        g = {k: z[k] for k in z.files}
    structs = np.frombuffer(g["records"].tobytes(), dtype=P.COMPACT_DTYPE)
    nib, occ = E.struct_array_to_ann_inputs(structs)
    assert np.array_equal(nib, g["piece_filters"]) and np.array_equal(occ, g["occupied_bbs"])
    nib2, occ2 = encoder.struct_array_to_ann_inputs(structs, structs[:0], np.ones(len(structs), bool), np.zeros(0, bool), len(structs))
    assert np.array_equal(nib2, nib) and np.array_equal(occ2, occ)
    planes = E.decode_planes_fast(nib, occ)
    # kings: exactly one own king (plane 1) and one opposing king (plane 8) on every legal board
    assert np.all(planes[:, 1].sum(axis=(1, 2)) == 1) and np.all(planes[:, 8].sum(axis=(1, 2)) == 1)
    assert np.all(planes.sum(axis=1) <= 1)
