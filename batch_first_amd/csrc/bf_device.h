// Device-side helpers shared by the encode and evaluation kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Bytes 0..81 of a node record (batch_first/classes_and_structs.py:7-29).
struct BoardBits {
    uint64_t pawns, knights, bishops, rooks, queens, kings, occ_black, occ_white, occupied, castling;
    int turn;  // non-zero = white to move
    int ep;    // 0 = none
};

__device__ __forceinline__ uint64_t load_u64_any(const unsigned char* p) {
    if ((reinterpret_cast<uintptr_t>(p) & 7u) == 0) return *reinterpret_cast<const uint64_t*>(p);
    uint64_t v = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

__device__ __forceinline__ BoardBits load_board(const unsigned char* rec) {
    BoardBits b;
    b.pawns = load_u64_any(rec + 0);
    b.knights = load_u64_any(rec + 8);
    b.bishops = load_u64_any(rec + 16);
    b.rooks = load_u64_any(rec + 24);
    b.queens = load_u64_any(rec + 32);
    b.kings = load_u64_any(rec + 40);
    b.occ_black = load_u64_any(rec + 48);
    b.occ_white = load_u64_any(rec + 56);
    b.occupied = load_u64_any(rec + 64);
    b.turn = (signed char)rec[72];
    b.castling = load_u64_any(rec + 73);
    b.ep = rec[81];
    return b;
}

// Relevance mask in the mover's frame: numba_negamax_zero_window.py:90-95.
__device__ __forceinline__ uint64_t relevance_mask(const BoardBits& b) {
    uint64_t m = b.occupied;
    if (b.ep) m |= 1ull << b.ep;
    if (!b.turn) m = __builtin_bswap64(m);  // flip_vertically, batch_first/__init__.py:297-302
    return m;
}

// Code of mover-frame square s, 15 if the square is not relevant:
// get_square_ary, numba_negamax_zero_window.py:32-50 with piece_type_at, numba_board.py:122-142.
__device__ __forceinline__ int square_code(const BoardBits& b, uint64_t mask, int s) {
    if (!((mask >> s) & 1ull)) return 15;
    const int q = b.turn ? s : (s ^ 56);
    const uint64_t bit = 1ull << q;
    int t;
    if (!(b.occupied & bit)) t = 0;
    else if (b.pawns & bit) t = 1;
    else if (b.knights & bit) t = 2;
    else if (b.bishops & bit) t = 3;
    else if (b.rooks & bit) t = 4;
    else if (b.queens & bit) t = 5;
    else t = 6;
    if (t == 0) return 0;
    int code = ((b.turn ? b.occ_white : b.occ_black) & bit) ? 7 : 14;
    if (!(b.castling & bit)) code -= t;
    return code;
}

}  // namespace bf
