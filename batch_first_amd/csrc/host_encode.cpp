// Host encoder: node records -> wire format, for callers that still want the reference's
// feed arrays.  Reference: struct_array_to_ann_inputs, get_square_ary, own_concat,
// compress_square_array (batch_first/numba_negamax_zero_window.py:11-105).
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/batchfirst.h"
#include "bf_internal.h"

namespace bf {

namespace {
thread_local char g_err[512] = "";

inline uint64_t rd64(const unsigned char* p) {
    uint64_t v;
    std::memcpy(&v, p, 8);
    return v;
}
}  // namespace

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

const char* last_error_text() { return g_err; }

uint64_t host_board_codes(const unsigned char* rec, uint8_t* codes_out, int* n_codes) {
    const uint64_t pawns = rd64(rec + 0), knights = rd64(rec + 8), bishops = rd64(rec + 16), rooks = rd64(rec + 24),
                   queens = rd64(rec + 32);
    const uint64_t occ_co[2] = {rd64(rec + 48), rd64(rec + 56)};
    const uint64_t occupied = rd64(rec + 64);
    const bool white = ((signed char)rec[72]) != 0;
    const uint64_t castling = rd64(rec + 73);
    const unsigned ep = rec[81];

    uint64_t mask = occupied;
    if (ep) mask |= 1ull << ep;
    if (!white) mask = __builtin_bswap64(mask);

    int n = 0;
    for (uint64_t m = mask; m; m &= m - 1) {
        const int s = __builtin_ctzll(m);
        const int q = white ? s : (s ^ 56);
        const uint64_t bit = 1ull << q;
        int code;
        if (!(occupied & bit)) {
            code = 0;
        } else {
            int t;
            if (pawns & bit) t = 1;
            else if (knights & bit) t = 2;
            else if (bishops & bit) t = 3;
            else if (rooks & bit) t = 4;
            else if (queens & bit) t = 5;
            else t = 6;
            code = (occ_co[white ? 1 : 0] & bit) ? 7 : 14;
            if (!(castling & bit)) code -= t;
        }
        codes_out[n++] = (uint8_t)code;
    }
    *n_codes = n;
    return mask;
}

}  // namespace bf

extern "C" const char* bf_last_error(void) { return bf::last_error_text(); }

extern "C" int bf_host_encode_wire(const void* h_structs, size_t stride, const uint8_t* h_mask, size_t n,
                                   uint8_t* h_piece_filters, uint64_t* h_occupied_bbs, size_t* n_bytes,
                                   size_t* n_scored) {
    if ((!h_structs && n) || !h_piece_filters || !h_occupied_bbs)
        return bf::fail(BF_E_ARG, "bf_host_encode_wire: null pointer");
    if (stride < BF_NODE_EVAL_BYTES) return bf::fail(BF_E_ARG, "bf_host_encode_wire: stride %zu < 82", stride);
    const unsigned char* recs = static_cast<const unsigned char*>(h_structs);
    size_t boards = 0, nibbles = 0;
    uint8_t codes[64];
    for (size_t j = 0; j < n; ++j) {
        if (h_mask && !h_mask[j]) continue;
        int nc = 0;
        h_occupied_bbs[boards++] = bf::host_board_codes(recs + j * stride, codes, &nc);
        for (int i = 0; i < nc; ++i, ++nibbles) {
            if (nibbles & 1) h_piece_filters[nibbles >> 1] |= codes[i];
            else h_piece_filters[nibbles >> 1] = (uint8_t)(codes[i] << 4);
        }
    }
    if (n_bytes) *n_bytes = (nibbles + 1) / 2;
    if (n_scored) *n_scored = boards;
    return BF_OK;
}

// Contiguous shards of a wire-format batch: boards [board_lo[p], board_lo[p+1]) go to
// part p, sizes differing by at most one (batch_first_amd/distributed.py:shard_range); nibble_lo[p] is
// the index of the shard's first code in the stream, i.e. the sum of the popcounts in front of it.  A
// shard whose nibble_lo is odd starts on the LOW nibble of byte nibble_lo / 2.  Both arrays have
// n_parts + 1 entries; the last ones are n and the total number of codes.
extern "C" int bf_host_split_wire(const uint64_t* h_occ, size_t n, size_t n_parts, size_t* board_lo, size_t* nibble_lo) {
    if (!h_occ || !board_lo || !nibble_lo || n_parts == 0) return bf::fail(BF_E_ARG, "bf_host_split_wire: bad argument");
    const size_t base = n / n_parts, extra = n % n_parts;
    size_t b = 0, nib = 0;
    for (size_t p = 0; p < n_parts; ++p) {
        board_lo[p] = b;
        nibble_lo[p] = nib;
        const size_t hi = b + base + (p < extra ? 1 : 0);
        for (; b < hi; ++b) nib += (size_t)__builtin_popcountll(h_occ[b]);
    }
    board_lo[n_parts] = b;
    nibble_lo[n_parts] = nib;
    return BF_OK;
}
