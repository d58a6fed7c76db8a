"""ORACLE (test infrastructure, not product): CPU restatement of the reference's board encoding.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Pinning: the reference ships no golden files for this path.  This restatement is pinned by the
reference's own test *rule* for it -- "two independent derivations of the plane index per square
must agree" (code_testing.py:184-263) -- using the reference's fixed FENs (code_testing.py:33-39),
its expected-plane rule (code_testing.py:139-181, 670-676, 766-767) and its wire-format consumer
(code_testing.py:505-532), all restated below, plus a hand-derived start-position vector
(SURVEY.md 8(c)).  python-chess, numba and TensorFlow are not installed here, so the reference
itself cannot be run: parity with the reference *implementation* is by restatement only.

Three functions, each following one piece of the reference:

* :func:`struct_array_to_ann_inputs`  -- node records -> (nibble stream, relevance masks)
  follows batch_first/numba_negamax_zero_window.py:11-105 (helpers numba_board.py:43-49,
  58-64, 93-96, 122-142; batch_first/__init__.py:297-302).
* :func:`parse_into_ann_input_inference` -- wire format -> float32 [N,15,8,8] planes
  follows batch_first/anns/ann_creation_helper.py:21-84 (and :488-510).
* :func:`expected_features` -- the independent derivation from a piece map
  follows code_testing.py:139-181 with cur_piece_to_filter_fn (:670-676), ep index 1,
  castling indices [8, 15] (:766-767).
"""
import numpy as np

PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING = range(1, 7)  # batch_first/__init__.py:178
_MASK64 = (1 << 64) - 1


def popcount(n):  # numba_board.py:58-64
    return bin(int(n)).count("1")


def flip_vertically(bb):  # batch_first/__init__.py:297-302 (a byte swap)
    bb = int(bb)
    c1 = 0x00FF00FF00FF00FF
    c2 = 0x0000FFFF0000FFFF
    bb = ((bb >> 8) & c1) | ((bb & c1) << 8)
    bb = ((bb >> 16) & c2) | ((bb & c2) << 16)
    bb = (bb >> 32) | (bb << 32)
    return bb & _MASK64


def scan_reversed(bb):  # numba_board.py:43-49 -- ascending square order despite the name
    bb = int(bb)
    for index in range(64):
        if bb & (1 << index):
            yield index
            bb ^= 1 << index


def piece_type_at(struct, square):  # numba_board.py:122-142
    mask = 1 << square
    if not int(struct["occupied"]) & mask:
        return 0
    elif int(struct["pawns"]) & mask:
        return PAWN
    elif int(struct["knights"]) & mask:
        return KNIGHT
    elif int(struct["bishops"]) & mask:
        return BISHOP
    elif int(struct["rooks"]) & mask:
        return ROOK
    elif int(struct["queens"]) & mask:
        return QUEEN
    else:
        return KING


def get_square_ary(struct, relevant_square_mask):  # numba_negamax_zero_window.py:32-50
    mirror = not struct["turn"]
    codes = []
    for square in scan_reversed(relevant_square_mask):
        if mirror:
            square ^= 0x38  # square_mirror, numba_board.py:93-96
        ptype = piece_type_at(struct, square)
        if ptype:
            bb_square = 1 << square
            code = 7 if int(struct["occupied_co"][int(struct["turn"])]) & bb_square else 14
            if int(struct["castling_rights"]) & bb_square == 0:
                code -= ptype
        else:
            code = 0
        codes.append(code)
    return codes


def compress_square_array(codes):  # numba_negamax_zero_window.py:11-19 with own_concat's pad (:53-68)
    codes = list(codes)
    if len(codes) % 2:
        codes.append(0)
    arr = np.asarray(codes, dtype=np.uint8)
    return ((arr[::2] << 4) | arr[1::2]).astype(np.uint8)


def struct_array_to_ann_inputs(child_structs, not_child_structs=None, to_score_child_mask=None,
                               to_score_not_child_mask=None, total_num_to_score=None):
    """numba_negamax_zero_window.py:71-105.  Returns (uint8 nibble stream, uint64 masks)."""
    if not_child_structs is None:
        not_child_structs = child_structs[:0]
    if to_score_child_mask is None:
        to_score_child_mask = np.ones(len(child_structs), bool)
    if to_score_not_child_mask is None:
        to_score_not_child_mask = np.ones(len(not_child_structs), bool)
    occupied_bbs = []
    all_codes = []
    for structs, mask in ((child_structs, to_score_child_mask), (not_child_structs, to_score_not_child_mask)):
        for j in range(len(structs)):
            if not mask[j]:
                continue
            struct = structs[j]
            m = int(struct["occupied"])
            if struct["ep_square"]:
                m |= 1 << int(struct["ep_square"])
            if not struct["turn"]:
                m = flip_vertically(m)
            occupied_bbs.append(m)
            all_codes.extend(get_square_ary(struct, m))
    if total_num_to_score is not None and total_num_to_score != len(occupied_bbs):
        raise ValueError("total_num_to_score does not match the masks")
    return compress_square_array(all_codes), np.asarray(occupied_bbs, dtype=np.uint64)


def parse_into_ann_input_inference(piece_filters, occupied_bbs):
    """ann_creation_helper.py:21-84: the on-device decode, as its net effect.

    The k-th set bit s (ascending) of board b's mask selects stream nibble
    ``sum(popcount(occ[:b])) + k``; even nibbles sit in the high half of their byte
    (:38-39); the one goes to ``planes[b, code, s >> 3, s & 7]`` (:29-33 after its two
    reversals, :76-79).
    """
    piece_filters = np.asarray(piece_filters, dtype=np.uint8)
    occupied_bbs = np.asarray(occupied_bbs, dtype=np.uint64)
    n = len(occupied_bbs)
    planes = np.zeros((n, 15, 8, 8), dtype=np.float32)
    nib = 0
    for b in range(n):
        for s in scan_reversed(occupied_bbs[b]):
            byte = int(piece_filters[nib >> 1])
            code = (byte >> 4) if (nib & 1) == 0 else (byte & 0x0F)
            planes[b, code, s >> 3, s & 7] += 1.0  # scatter_nd accumulates
            nib += 1
    return planes


def decode_planes_fast(piece_filters, occupied_bbs):
    """Vectorised equivalent of :func:`parse_into_ann_input_inference` for large N (tests check they agree)."""
    piece_filters = np.asarray(piece_filters, dtype=np.uint8)
    occ = np.asarray(occupied_bbs, dtype=np.uint64)
    n = len(occ)
    bits = ((occ[:, None] >> np.arange(64, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
    b_idx, s_idx = np.nonzero(bits)  # row-major => board order, ascending square
    total = len(b_idx)
    nibbles = np.empty(2 * len(piece_filters), np.uint8)
    nibbles[0::2] = piece_filters >> 4
    nibbles[1::2] = piece_filters & 0x0F
    codes = nibbles[:total]
    planes = np.zeros((n, 15, 8, 8), dtype=np.float32)
    planes[b_idx, codes, s_idx >> 3, s_idx & 7] = 1.0
    return planes


def expected_features(pos):
    """code_testing.py:139-181 for one position given as ``parse_fen``-style dict
    (piece_at {sq: (type, is_white)}, turn, castling bitboard, ep square or None).

    Returns uint8[8,8] of 1-based plane numbers (0 = no plane), mover's frame.
    """
    def piece_to_filter(ptype, white):  # cur_piece_to_filter_fn, code_testing.py:670-676
        return 8 - ptype if white else 15 - ptype

    turn = pos["turn"]
    out = np.zeros(64, dtype=np.uint8)
    for sq, (ptype, white) in pos["piece_at"].items():
        # black to move: colours are swapped before the lookup (:159-161)
        out[sq] = piece_to_filter(ptype, white if turn else not white)
    if pos["ep"] is not None:
        out[pos["ep"]] = 1
    corner_squares = [0, 7, 56, 63]  # A1, H1, A8, H8 (:154)
    castling_filter = [8, 15]
    idx = [0, 0, 1, 1] if turn else [1, 1, 0, 0]  # :171
    for corner, i in zip(corner_squares, idx):
        if pos["castling"] & (1 << corner):
            out[corner] = castling_filter[i]
    out = out.reshape(8, 8)
    if not turn:
        out = out[::-1]
    return out.copy()


def decompress_squares(to_decompress, desired_num):  # code_testing.py:505-513
    to_decompress = np.asarray(to_decompress, np.uint8)
    out = np.empty(desired_num, np.uint8)
    out[::2] = to_decompress >> 4
    if desired_num % 2:
        out[1::2] = to_decompress[:-1] & np.uint8(0x0F)
    else:
        out[1::2] = to_decompress & np.uint8(0x0F)
    return out


def bf_piece_sum_eval(compressed_piece_filters, relevant_piece_masks):  # code_testing.py:516-532
    values_for_white = np.array([0, 900, 500, 300, 300, 100, 500])
    piece_filter_values = np.r_[0, values_for_white, -values_for_white]
    counts = np.array([popcount(m) for m in relevant_piece_masks], dtype=np.int64)
    codes = decompress_squares(compressed_piece_filters, int(counts.sum()))
    vals = piece_filter_values[codes]
    edges = np.r_[0, np.cumsum(counts)]
    return np.array([vals[s:e].sum() for s, e in zip(edges[:-1], edges[1:])])


def planes_to_mailbox(planes):
    """[N,15,8,8] one-hot planes -> uint8[N,64] of codes, 15 where no plane is set."""
    planes = np.asarray(planes)
    n = planes.shape[0]
    flat = planes.reshape(n, 15, 64)
    any_set = flat.sum(axis=1) > 0
    code = flat.argmax(axis=1).astype(np.uint8)
    code[~any_set] = 15
    return code
