"""Position records for the evaluation path: the node struct layout, a FEN reader and a
seeded generator of structurally valid positions.

``NODE_DTYPE`` reproduces the *layout* of the reference's packed node record
(batch_first/classes_and_structs.py:7-29): 806 bytes, unaligned, of which only the first
82 bytes (piece bitboards, colour occupancy, turn, castling rights, ep square) are read by the
evaluation path.  ``COMPACT_DTYPE`` is the same first 82 bytes padded to an 88-byte stride;
the HIP ingest kernel takes either through its ``stride`` argument.

Conventions (batch_first/__init__.py:176-205): square = 8*rank + file, a1 = 0; bit ``sq`` of a
bitboard is that square; ``occupied_co[0]`` is black, ``[1]`` white; ``turn`` 1 = white to move;
piece types P,N,B,R,Q,K = 1..6; ``ep_square`` 0 means none.
"""
import numpy as np

MAX_MOVES_LOOKED_AT = 100  # batch_first/__init__.py:126

NODE_DTYPE = np.dtype(
    [("pawns", np.uint64),
     ("knights", np.uint64),
     ("bishops", np.uint64),
     ("rooks", np.uint64),
     ("queens", np.uint64),
     ("kings", np.uint64),
     ("occupied_co", np.uint64, (2,)),
     ("occupied", np.uint64),
     ("turn", np.int8),
     ("castling_rights", np.uint64),
     ("ep_square", np.uint8),
     ("halfmove_clock", np.uint8),
     ("hash", np.uint64),
     ("terminated", np.bool_),
     ("separator", np.float32),
     ("depth", np.uint8),
     ("best_value", np.float32),
     ("unexplored_moves", np.uint8, (MAX_MOVES_LOOKED_AT, 3)),
     ("unexplored_move_scores", np.float32, (MAX_MOVES_LOOKED_AT,)),
     ("prev_move", np.uint8, (3,)),
     ("next_move_index", np.uint8),
     ("children_left", np.uint8)])
assert NODE_DTYPE.itemsize == 806

EVAL_FIELDS = ("pawns", "knights", "bishops", "rooks", "queens", "kings",
               "occupied_co", "occupied", "turn", "castling_rights", "ep_square")

COMPACT_DTYPE = np.dtype(
    {"names": list(EVAL_FIELDS),
     "formats": [NODE_DTYPE.fields[n][0] for n in EVAL_FIELDS],
     "offsets": [NODE_DTYPE.fields[n][1] for n in EVAL_FIELDS],
     "itemsize": 88})
EVAL_BYTES = 82

# The reference's fixed test positions (code_testing.py:33-39, from the chessprogramming
# wiki perft page).  They are data: castling on both sides, a mirrored black-to-move pair,
# promotions, a side in check.
REFERENCE_TEST_FENS = (
    "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1",
    "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1",
    "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1",
    "r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1",
    "r2q1rk1/pP1p2pp/Q4n2/bbp1p3/Np6/1B3NBn/pPPP1PPP/R3K2R b KQ - 0 1",
    "rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8 ",
    "r4rk1/1pp1qppp/p1np1n2/2b1p1B1/2B1P1b1/P1NP1N2/1PP1QPPP/R4RK1 w - - 0 10",
)
EXTRA_FENS = (
    # after 1.e4: black to move with an ep square (e3)
    "rnbqkbnr/pppppppp/8/8/4P3/8/PPPP1PPP/RNBQKBNR b KQkq e3 0 1",
    # after 1.e4 c5 2.e5 d5: white to move, ep square d6
    "rnbqkbnr/pp2pppp/8/2ppP3/8/8/PPPP1PPP/RNBQKBNR w KQkq d6 0 3",
    # bare kings
    "8/8/8/4k3/8/8/3K4/8 b - - 0 1",
)

_PIECE_FIELD = {"p": "pawns", "n": "knights", "b": "bishops", "r": "rooks", "q": "queens", "k": "kings"}


def parse_fen(fen):
    """Return a plain dict: piece_at {square: (type 1..6, is_white)}, turn, castling bitboard, ep square or None."""
    parts = fen.split()
    rows = parts[0].split("/")
    if len(rows) != 8:
        raise ValueError("bad FEN board: %r" % fen)
    piece_at = {}
    for i, row in enumerate(rows):
        rank = 7 - i
        file = 0
        for ch in row:
            if ch.isdigit():
                file += int(ch)
            else:
                ptype = "pnbrqk".index(ch.lower()) + 1
                piece_at[8 * rank + file] = (ptype, ch.isupper())
                file += 1
        if file != 8:
            raise ValueError("bad FEN rank %r in %r" % (row, fen))
    turn = parts[1] == "w"
    castling = 0
    for ch in parts[2]:
        if ch == "K":
            castling |= 1 << 7
        elif ch == "Q":
            castling |= 1 << 0
        elif ch == "k":
            castling |= 1 << 63
        elif ch == "q":
            castling |= 1 << 56
    ep = None
    if parts[3] != "-":
        ep = "abcdefgh".index(parts[3][0]) + 8 * (int(parts[3][1]) - 1)
    return {"piece_at": piece_at, "turn": turn, "castling": castling, "ep": ep}


def structs_from_fens(fens, dtype=NODE_DTYPE):
    """Build node records from FEN strings (the role of create_node_info_from_fen in the reference)."""
    out = np.zeros(len(fens), dtype=dtype)
    for j, fen in enumerate(fens):
        pos = parse_fen(fen)
        rec = out[j]
        bbs = {name: 0 for name in _PIECE_FIELD.values()}
        occ = [0, 0]
        for sq, (ptype, white) in pos["piece_at"].items():
            bbs["pawns knights bishops rooks queens kings".split()[ptype - 1]] |= 1 << sq
            occ[1 if white else 0] |= 1 << sq
        for name, v in bbs.items():
            rec[name] = np.uint64(v)
        rec["occupied_co"][0] = np.uint64(occ[0])
        rec["occupied_co"][1] = np.uint64(occ[1])
        rec["occupied"] = np.uint64(occ[0] | occ[1])
        rec["turn"] = 1 if pos["turn"] else 0
        rec["castling_rights"] = np.uint64(pos["castling"])
        rec["ep_square"] = 0 if pos["ep"] is None else pos["ep"]
    return out


def to_compact(structs):
    """Copy the 82 evaluation bytes of each record into an 88-byte-stride array."""
    out = np.zeros(len(structs), dtype=COMPACT_DTYPE)
    for name in EVAL_FIELDS:
        out[name] = structs[name]
    return out


_ONE = np.uint64(1)


def _bit(sq):
    return _ONE << sq.astype(np.uint64)


def random_structs(n, seed=0, dtype=NODE_DTYPE):
    """``n`` seeded, structurally valid positions (SURVEY.md 8(d) tier-1 generator).

    Both kings present, at most 16 men a side, at most 8 pawns a side and none on the back
    ranks, castling bits only where king and rook stand on their home squares, an ep square
    only behind a just-pushed pawn with an empty origin square, side to move 50/50.
    The positions need not be reachable; the evaluation path does not care.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sq = np.arange(64, dtype=np.uint64)

    # pawns: ranks over the 48 middle squares
    mid_rank = np.argsort(np.argsort(rng.random((n, 48)), axis=1), axis=1)
    nwp = rng.integers(0, 9, n)[:, None]
    nbp = rng.integers(0, 9, n)[:, None]
    wp_mask = mid_rank < nwp
    bp_mask = (mid_rank >= nwp) & (mid_rank < nwp + nbp)
    mid_bits = (_ONE << sq[8:56])[None, :]
    wp = np.bitwise_or.reduce(np.where(wp_mask, mid_bits, np.uint64(0)), axis=1)
    bp = np.bitwise_or.reduce(np.where(bp_mask, mid_bits, np.uint64(0)), axis=1)
    pawn_occ = wp | bp

    # other men: ranks over the squares no pawn took
    pri = rng.random((n, 64))
    taken = ((pawn_occ[:, None] >> sq[None, :]) & _ONE).astype(bool)
    pri[taken] = 2.0
    order = np.argsort(np.argsort(pri, axis=1), axis=1)
    # per side: K=1, then up to 1 Q, 2 R, 2 B, 2 N
    counts = [np.ones(n, np.int64), np.ones(n, np.int64)]
    for hi in (2, 2, 3, 3, 3, 3, 3, 3):
        counts.append(rng.integers(0, hi, n))
    # layout of the count list: wK bK wQ bQ wR bR wB bB wN bN
    edges = np.cumsum(np.stack(counts, axis=1), axis=1)
    all_bits = (_ONE << sq)[None, :]

    def take(idx):
        lo = edges[:, idx - 1][:, None] if idx else 0
        hi = edges[:, idx][:, None]
        m = (order >= lo) & (order < hi)
        return np.bitwise_or.reduce(np.where(m, all_bits, np.uint64(0)), axis=1)

    wk, bk, wq, bq, wr, br, wb, bb_, wn, bn = (take(i) for i in range(10))

    # castling: for some boards move the king home and drop rooks on the corners
    rights = np.zeros(n, np.uint64)
    for white in (True, False):
        k_home = np.uint64(1 << (4 if white else 60))
        wants = [rng.random(n) < 0.2 for _ in range(2)]
        own_r = wr if white else br
        # castling rooks replace the random ones so a side never exceeds 16 men
        own_r &= np.where(wants[0] | wants[1], np.uint64(0), ~np.uint64(0))
        for want, corner in zip(wants, (7, 0) if white else (63, 56)):
            cbit = np.uint64(1 << corner)
            own_k = wk if white else bk
            opp_k = bk if white else wk
            # never evict the other king
            ok = want & ((opp_k & (k_home | cbit)) == 0)
            clear = np.where(ok, ~(k_home | cbit), ~np.uint64(0))
            for arr in (wp, bp, wq, bq, wr, br, wb, bb_, wn, bn):
                arr &= clear
            own_k[:] = np.where(ok, k_home, own_k)
            own_r |= np.where(ok, cbit, np.uint64(0))
            rights |= np.where(ok, cbit, np.uint64(0))
    # a later eviction may have removed an earlier castling rook / king: keep only rights still backed
    w_ok = (wk == np.uint64(1 << 4))
    b_ok = (bk == np.uint64(1 << 60))
    rights &= (np.where(w_ok, wr & np.uint64(0x81), np.uint64(0))
               | np.where(b_ok, br & np.uint64(0x81 << 56), np.uint64(0)))

    turn = rng.integers(0, 2, n).astype(np.int8)

    # ep squares
    ep = np.zeros(n, np.uint8)
    want = rng.random(n) < 0.15
    file = rng.integers(0, 8, n).astype(np.uint64)
    white_to_move = turn == 1
    pawn_sq = np.where(white_to_move, 32 + file, 24 + file).astype(np.uint64)
    ep_sq = np.where(white_to_move, 40 + file, 16 + file).astype(np.uint64)
    org_sq = np.where(white_to_move, 48 + file, 8 + file).astype(np.uint64)
    trio = (_ONE << pawn_sq) | (_ONE << ep_sq) | (_ONE << org_sq)
    ok = want & (((wk | bk) & trio) == 0)
    clear = np.where(ok, ~trio, ~np.uint64(0))
    for arr in (wp, bp, wq, bq, wr, br, wb, bb_, wn, bn):
        arr &= clear
    rights &= clear
    pushed = np.where(ok, _ONE << pawn_sq, np.uint64(0))
    bp |= np.where(white_to_move, pushed, np.uint64(0))
    wp |= np.where(~white_to_move, pushed, np.uint64(0))
    ep[ok] = ep_sq[ok].astype(np.uint8)

    out = np.zeros(n, dtype=dtype)
    out["pawns"] = wp | bp
    out["knights"] = wn | bn
    out["bishops"] = wb | bb_
    out["rooks"] = wr | br
    out["queens"] = wq | bq
    out["kings"] = wk | bk
    white_occ = wp | wn | wb | wr | wq | wk
    black_occ = bp | bn | bb_ | br | bq | bk
    out["occupied_co"][:, 0] = black_occ
    out["occupied_co"][:, 1] = white_occ
    out["occupied"] = white_occ | black_occ
    out["turn"] = turn
    out["castling_rights"] = rights
    out["ep_square"] = ep
    return out


def test_set(n=1024, seed=0, dtype=NODE_DTYPE):
    """The fixed fixture set: reference FENs + hand cases first, seeded random positions after."""
    fixed = structs_from_fens(list(REFERENCE_TEST_FENS) + list(EXTRA_FENS), dtype=dtype)
    if n <= len(fixed):
        return fixed[:n].copy()
    rest = random_structs(n - len(fixed), seed=seed, dtype=dtype)
    return np.concatenate([fixed, rest])
