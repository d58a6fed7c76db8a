/* ORACLE (test infrastructure, not product): serial C restatement of the reference's host
 * encoder, used as the single-thread CPU baseline and as a second witness for the NumPy
 * restatement in oracle/encoding.py.
 *
 * Follows batch_first/numba_negamax_zero_window.py:71-105 (struct_array_to_ann_inputs, a serial
 * @njit loop), :32-50 (get_square_ary), :53-68 (own_concat), :11-19 (compress_square_array);
 * numba_board.py:43-49 (scan_reversed: ascending squares), :122-142 (piece_type_at);
 * batch_first/__init__.py:297-302 (flip_vertically).  Record layout:
 * batch_first/classes_and_structs.py:7-29.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static uint64_t rd64(const unsigned char* p) { uint64_t v; memcpy(&v, p, 8); return v; }

static uint64_t flip_vertically(uint64_t bb) {
    const uint64_t c1 = 0x00FF00FF00FF00FFull, c2 = 0x0000FFFF0000FFFFull;
    bb = ((bb >> 8) & c1) | ((bb & c1) << 8);
    bb = ((bb >> 16) & c2) | ((bb & c2) << 16);
    bb = (bb >> 32) | (bb << 32);
    return bb;
}

static int piece_type_at(const unsigned char* rec, int square) {
    const uint64_t mask = 1ull << square;
    if (!(rd64(rec + 64) & mask)) return 0;
    if (rd64(rec + 0) & mask) return 1;
    if (rd64(rec + 8) & mask) return 2;
    if (rd64(rec + 16) & mask) return 3;
    if (rd64(rec + 24) & mask) return 4;
    if (rd64(rec + 32) & mask) return 5;
    return 6;
}

/* Returns the number of stream bytes written; occupied_bbs receives one mask per selected record. */
size_t oracle_struct_array_to_ann_inputs(const unsigned char* recs, size_t stride, const unsigned char* mask,
                                         size_t n, unsigned char* piece_filters, uint64_t* occupied_bbs,
                                         size_t* n_scored) {
    size_t total = 0, store = 0;
    for (size_t j = 0; j < n; ++j) {
        if (mask && !mask[j]) continue;
        const unsigned char* rec = recs + j * stride;
        const int turn = (signed char)rec[72];
        uint64_t m = rd64(rec + 64);
        if (rec[81]) m |= 1ull << rec[81];
        if (!turn) m = flip_vertically(m);
        occupied_bbs[store++] = m;
        for (int index = 0; index < 64; ++index) {
            if (!(m & (1ull << index))) continue;
            const int square = turn ? index : (index ^ 0x38);
            const int type = piece_type_at(rec, square);
            int code = 0;
            if (type) {
                const uint64_t bb = 1ull << square;
                code = (rd64(rec + 48 + 8 * (turn ? 1 : 0)) & bb) ? 7 : 14;
                if ((rd64(rec + 73) & bb) == 0) code -= type;
            }
            if (total & 1) piece_filters[total >> 1] |= (unsigned char)code;
            else piece_filters[total >> 1] = (unsigned char)(code << 4);
            ++total;
        }
    }
    if (n_scored) *n_scored = store;
    return (total + 1) / 2;
}
