// k_eval_f32q: the float32 value network for SMALL calls -- one board per workgroup, eight waves (gfx950).
//
// Same network, same weight image and, bit for bit, the same scores as k_eval_f32 (eval_f32.hip; reference:
// ann_creation_helper.py:192-252, evaluation_ann.py:17-46,268).  k_eval_f32 gives a board to two waves, whose matrix
// work alone is 278 k cycles each: a call of a hundred boards (the search at depth 4, numba_negamax_zero_window.py:
// 183-202) waits 158 us while three quarters of the chip idle.  Here a board has a compute unit to itself.
//
// What makes a different split bit-identical: v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 both compute, per
// output element, a chain of fused multiply-adds in k order (tools/mfma_shape_bits.hip: 0 of 1 024 elements differ from
// fmaf on the host).  So only the ORDER of k per element is part of the result, not the tiling, and every layer below
// keeps k_eval_f32's order:
//   stage 1   the eight waves share the nine input convs, each computing its row tiles for both halves of the board
//             (the bf16 three-term form of k_eval_f32, same instruction, same sequence per accumulator; every weight
//             group crosses the L1 once) and exchange x1 through LDS; every wave then holds its half's x1 in 118
//             registers;
//   236->512  wave (half, m) takes hidden tiles 4m .. 4m+3 and leaves them, after ReLU, in LDS;
//   512->128  wave (half, m) takes output tile m over all sixteen hidden tiles in order, B operands from LDS;
//   3x3       32 output channels are ONE 32-row tile, which 32x32 tiles cannot spread over more than two waves: these
//             layers run on v_mfma_f32_16x16x4_f32, wave = (16 channels) x (16 squares), A operands picked out of the
//             same weight stream's fragments (staged chunk by chunk in LDS by the whole workgroup), chunk, tap,
//             channel ascending: k_eval_f32's order;
//   head      threads 0..127 redo k_eval_f32's per-lane partial sums from x9 in LDS, then its fixed-order reduction.
// The 1x1 pair's weights come straight from L2 through a register prefetch ring (every wave needs other groups: no
// sharing to be had; a few hundred workgroups at most).  One board's fp32 matrix work is 135 k cycles on the four
// SIMDs of a CU whatever the split, 56 us at 2.4 GHz: that, not 60 us a call, is the floor.
// LEVEL 1 / 2 (BF_PRECISION_F32_SPLIT / _SPLIT3) take the same route with the large-batch kernel's bf16-term MFMAs per
// tile; see the kernel's comment.  Measured: f32s 62-65 us (87 on k_eval_f32<., 1>), f32s3 70-75 us (108).
// Measured (LEVEL 0): 100-103 us a call for 1 .. 256 boards against 154-158 us on k_eval_f32 (wave life 190 k cycles: stage 1
// 14 k, 236->512 70 k, 512->128 37 k, 3x3 layers 67 k; profiles/r3/f32q/).
#include <hip/hip_runtime.h>

#include "bf_device.h"
#include "bf_kernels.h"
#include "bf_layout.h"

namespace bf {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int QWAVES = 8, QTHREADS = 64 * QWAVES;
constexpr int MB_W = 22, MB_BYTES = 496;          // mailbox padded by 7 squares on every side (eval_f32.hip)
constexpr int XP_W = 10, XP_PLANE = 100;          // activation planes padded by 1
constexpr int PLANE_FLOATS = 32 * XP_PLANE;       // one 32-channel chunk
constexpr int N_PLANES = 4;                       // x3's four chunks; planes 0 and 1 alternate for x4 .. x8 afterwards
constexpr int HID_FLOATS = 2 * 16 * 16 * 64;      // [half][hidden tile][register][lane]: 128 KB, the region's size
// What lives in the region, in turn: x1 [118 registers][half][lane] (stage 1 -> everybody), the hidden layer, and from
// the 3x3 layers on: four planes, two buffers of one chunk's weights (36 groups), and, over the first of those when
// the layers are done, x9 [32][64] and the head's reduction scratch.
constexpr int CHUNK_W_FLOATS = 36 * kGroupFloats; // 9 216
constexpr int WBUF_OFF = N_PLANES * PLANE_FLOATS; // 12 800
constexpr int X9_OFF = WBUF_OFF;
constexpr int RED_STRIDE = 17, RED_OFF = X9_OFF + 32 * 64;
static_assert(WBUF_OFF + 2 * CHUNK_W_FLOATS <= HID_FLOATS && RED_OFF + 128 * RED_STRIDE <= WBUF_OFF + CHUNK_W_FLOATS, "region map");
static_assert(kX1Regs * 2 * 64 <= HID_FLOATS, "x1 fits the region");

struct LdsQ {
    float region[HID_FLOATS];                     // the hidden layer; afterwards the planes, x9, the head's scratch
    unsigned char mb[MB_BYTES];
};

__device__ __forceinline__ float pick(int hl, float lo, float hi) { return hl ? hi : lo; }
__device__ __forceinline__ float relu(float x) {   // as eval_f32.hip: one integer max on the bit pattern
    return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0));
}
#ifdef BFQ_STAMPS   // diagnostic build: MODE 2, block 0 leaves eight time stamps per wave behind the last board's features
#define BFQ_STAMP(i) do { stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BFQ_STAMP(i) ((void)0)
#endif
#define BFQ_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define BFQ_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// ---- LEVEL 1 / 2: BF_PRECISION_F32_SPLIT / _SPLIT3 (eval_f32.hip: k_eval_f32<MODE, SPLIT>), the same bits ---------------
// split2 / split3 as there: bf16 terms of a pair of float32 values, packed (a low, b high).
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    const float ra = a - __builtin_bit_cast(float, hi << 16);
    const float rb = b - __builtin_bit_cast(float, hi & 0xFFFF0000u);
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, bf16x2));
}
__device__ __forceinline__ void split3(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    const float ra = a - __builtin_bit_cast(float, hi << 16);
    const float rb = b - __builtin_bit_cast(float, hi & 0xFFFF0000u);
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, bf16x2));
    const float sa = ra - __builtin_bit_cast(float, mid << 16);
    const float sb = rb - __builtin_bit_cast(float, mid & 0xFFFF0000u);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{sa, sb}, bf16x2));
}
// The split precisions' term planes (eval_f32.hip): [term][10 x 10 positions][32 channels] bf16, 64 bytes a position,
// chunk c of a position at chunk (c + padded rank) & 3.  Six sets here: x3's four chunks, two alternating for x4 .. x8.
constexpr int XS_POS_BYTES = 64, XS_PLANE_BYTES = XP_PLANE * XS_POS_BYTES, XS_SET_BYTES = 3 * XS_PLANE_BYTES;   // 19 200
constexpr int XS_SETS = 6, XS_RED_OFF = XS_SETS * XS_SET_BYTES / 4;   // float offset of the head's scratch behind them
static_assert(XS_RED_OFF + 128 * RED_STRIDE <= HID_FLOATS, "term planes and the head's scratch fit the region");

// Groups of the weight stream in a fixed consumption order, P of them in flight from L2 (an L2 hit takes several
// hundred cycles: P x the MFMA time a group feeds has to cover it).  OFF(i) -> float4 offset of the i-th group from
// `p` (compile-time after unrolling); n groups in all.
template <int P, class OFF>
struct GStream {
    const f32x4* p;
    f32x4 pre[P + 1];
    OFF off;
    int n;
    __device__ __forceinline__ GStream(const f32x4* base, OFF o, int count) : p(base), off(o), n(count) {
#pragma unroll
        for (int i = 0; i < P; ++i) pre[i] = p[off(i)];
    }
    __device__ __forceinline__ f32x4 take(int i) {
        if (i + P < n) pre[(i + P) % (P + 1)] = p[off(i + P)];
        // pin the load here: the scheduler otherwise sinks it towards its use and the look-ahead shrinks to a few groups
        __builtin_amdgcn_sched_barrier(0);
        return pre[i % (P + 1)];
    }
};

constexpr int input_group_base(int conv) {
    int g = 0;
    for (int i = 0; i < conv; ++i) g += kInputConvs[i].k * kInputConvs[i].k * kInputConvs[i].tiles * kInputSplits;
    return g;
}

// Stage 1: row tiles T0 .. T0 + NT - 1 of input conv I for BOTH halves of the board: k_eval_f32's input_conv,
// instruction for instruction per accumulator, with every weight group fetched once per workgroup (straight from L2,
// 15 groups ahead: two 32-cycle MFMAs a group) and the results, after ReLU, to the x1 exchange in LDS:
// x1buf[register][half][lane], register r = channels 2r / 2r+1 of the lane halves as in k_eval_f32.
// mb0 / mb1: the mailbox at the lane's square of half 0 / 1.
template <int I, int T0, int NT>
__device__ __forceinline__ void input_conv(const f32x4* __restrict__ wstream, int lane, const unsigned char* mb0,
                                           const unsigned char* mb1, int hl, const float* __restrict__ bias, float* x1buf_lane) {
    constexpr InputConv cv = kInputConvs[I];
    constexpr int G0 = input_group_base(I), NG = cv.k * cv.k * NT * kInputSplits;
    constexpr int ph = ((cv.k - 1) * cv.dh) / 2, pw = ((cv.k - 1) * cv.dw) / 2;
    static_assert(T0 + NT <= cv.tiles, "");
    // the i-th group this call consumes: (tap, tile T0 + tl, split term), tap-major as the stream is
    auto off = [](int i) { return (((i / kInputSplits) / NT * cv.tiles + T0 + (i / kInputSplits) % NT) * kInputSplits + i % kInputSplits) * 64; };
    GStream<15, decltype(off)> ws(wstream + (size_t)G0 * 64 + lane, off, NG);
    f32x16 acc[NT][2];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int c = 32 * (T0 + tl) + 2 * t;
            acc[tl][0][t] = acc[tl][1][t] =
                (c < cv.ch) ? pick(hl, bias[kBiasX1 + cv.chan_base + c], bias[kBiasX1 + cv.chan_base + c + 1]) : 0.0f;
        }
    auto onehot = [&](const unsigned char* mbc, int dr, int df) {
        const unsigned cc = (unsigned)((int)mbc[dr * MB_W + df] - 8 * hl);
        const unsigned one = (cc < 8u) ? (0x3F80u << ((cc & 1u) * 16u)) : 0u;
        u32x4 bu;
        bu.x = ((cc >> 1) == 0u) ? one : 0u;
        bu.y = ((cc >> 1) == 1u) ? one : 0u;
        bu.z = ((cc >> 1) == 2u) ? one : 0u;
        bu.w = ((cc >> 1) == 3u) ? one : 0u;
        return __builtin_bit_cast(bf16x8, bu);
    };
#pragma unroll
    for (int tap = 0; tap < cv.k * cv.k; ++tap) {
        const int dr = (tap / cv.k) * cv.dh - ph, df = (tap % cv.k) * cv.dw - pw;
        const bf16x8 b0 = onehot(mb0, dr, df), b1 = onehot(mb1, dr, df);
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
#pragma unroll
            for (int sp = 0; sp < kInputSplits; ++sp) {
                const bf16x8 w = __builtin_bit_cast(bf16x8, ws.take((tap * NT + tl) * kInputSplits + sp));
                acc[tl][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, b0, acc[tl][0], 0, 0, 0);
                acc[tl][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, b1, acc[tl][1], 0, 0, 0);
            }
    }
#pragma unroll
    for (int tl = 0; tl < NT; ++tl)
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (32 * (T0 + tl) + 2 * t < cv.ch) {
                float* x = x1buf_lane + (cv.chan_base / 2 + 16 * (T0 + tl) + t) * 128;
                x[0] = relu(acc[tl][0][t]);
                x[64] = relu(acc[tl][1][t]);
            }
}

// One 3x3 SAME conv over a 32-channel chunk in a zero-haloed plane, for this wave's 16 output channels x 16 squares:
// acc[t] (channel 16 mt + 4 q4 + t, square of lane & 15) += sum over taps, then channels ascending: k_eval_f32's order.
// wb: the chunk's 36 groups in LDS (the stream's 32x32x2 fragments as they are) at this lane's float4,
//     (q4 & 1) * 32 + 16 mt + r: the pair of steps (tap, s' = 2u, 2u+1) reads group 4 tap + u once (ds_read_b128: the
//     four k-step values of the lane's row) and takes components hi, 2 + hi with hi = q4 >> 1;
// xb: the plane at channel q4, one row and column back from the lane's square.  Operands BD pairs / 2 BD steps ahead.
// (With one ds_read_b32 per operand and step the eight waves ask the LDS for more than it delivers per MFMA slot.)
#ifndef BFQ_BD
#define BFQ_BD 4
#endif
constexpr int CONV_STEPS = 72, BD = BFQ_BD;
__device__ __forceinline__ void conv3x3_chunk_q(const f32x4* wb, int hi, const float* xb, f32x4& acc) {
    auto b_off = [](int i) { return 4 * (i & 7) * XP_PLANE + ((i >> 3) / 3) * XP_W + ((i >> 3) % 3); };
    f32x4 a[BD];
    float b[2 * BD];
#pragma unroll
    for (int j = 0; j < BD; ++j) {
        a[j] = wb[j * 64];
        b[2 * j] = xb[b_off(2 * j)];
        b[2 * j + 1] = xb[b_off(2 * j + 1)];
    }
#pragma unroll
    for (int j = 0; j < CONV_STEPS / 2; ++j) {   // pair j = steps 2j, 2j+1 = group j of the chunk
        const f32x4 av = a[j % BD];
        const float b0 = b[(2 * j) % (2 * BD)], b1 = b[(2 * j + 1) % (2 * BD)];
        if (j + BD < CONV_STEPS / 2) {
#ifndef BFQ_ABL_NOA   // timing ablations (wrong results): operands not re-read
            a[j % BD] = wb[(j + BD) * 64];
#endif
#ifndef BFQ_ABL_NOB
            b[(2 * j) % (2 * BD)] = xb[b_off(2 * (j + BD))];
            b[(2 * j + 1) % (2 * BD)] = xb[b_off(2 * (j + BD) + 1)];
#endif
        }
        __builtin_amdgcn_sched_barrier(0);   // the reads stay ahead of their use (2, 4 or 8 pairs ahead, pinned or not: the same time)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hi ? av.y : av.x, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hi ? av.w : av.z, b1, acc, 0, 0, 0);
    }
}

// A chunk's 36 KB of weights, fetched by the whole workgroup: 2 304 float4, at most five a thread, held in registers
// while the chunk before is computed and dropped into the other LDS buffer at its end.
struct ChunkFetch {
    f32x4 v[5];
    __device__ __forceinline__ void load(const f32x4* __restrict__ src, int tid) {
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (j < 4 || tid < CHUNK_W_FLOATS / 4 - 4 * QTHREADS) v[j] = src[tid + j * QTHREADS];
    }
    __device__ __forceinline__ void store(float* dst, int tid) const {
        f32x4* d = reinterpret_cast<f32x4*>(dst);
#pragma unroll
        for (int j = 0; j < 5; ++j)
            if (j < 4 || tid < CHUNK_W_FLOATS / 4 - 4 * QTHREADS) d[tid + j * QTHREADS] = v[j];
    }
};

// conv3x3_chunk_split of eval_f32.hip (three terms a side, its six products in its order) with the weights from L2;
// f: the chunk's first group in the wave's conv stream (54 groups a chunk, chunks contiguous).
template <class WS>
__device__ __forceinline__ void conv3x3_chunk_split_q(WS& ws, int f, const char* set, const int (&xa)[4], f32x16& acc) {
    auto load_b = [&](int idx, u32x4 (&b)[3]) {
        const int tap = idx >> 1, s = idx & 1, tr = tap / 3, tf = tap % 3;
        const char* p = set + xa[(2 * s + tr) & 3] + (tr * XP_W + tf) * XS_POS_BYTES;
#pragma unroll
        for (int term = 0; term < 3; ++term) b[term] = *reinterpret_cast<const u32x4*>(p + term * XS_PLANE_BYTES);
    };
    u32x4 b[2][3];
    load_b(0, b[0]);
#pragma unroll
    for (int idx = 0; idx < 18; ++idx) {
        if (idx + 1 < 18) load_b(idx + 1, b[(idx + 1) & 1]);
        const f32x4 wh = ws.take(f + 3 * idx), wm = ws.take(f + 3 * idx + 1), wl = ws.take(f + 3 * idx + 2);
        const u32x4 &xh = b[idx & 1][0], &xm = b[idx & 1][1], &xl = b[idx & 1][2];
        acc = BFQ_MFMA16(wl, xh, acc);
        acc = BFQ_MFMA16(wh, xl, acc);
        acc = BFQ_MFMA16(wm, xm, acc);
        acc = BFQ_MFMA16(wm, xh, acc);
        acc = BFQ_MFMA16(wh, xm, acc);
        acc = BFQ_MFMA16(wh, xh, acc);
    }
}

// The products of one k-step of the 1x1 pair in k_eval_f32<., SPLIT>'s order: T = 2 terms a side: all four;
// T = 3: the six with i + j <= 2.  w[term], x[term]: A and B fragments.
template <int T>
__device__ __forceinline__ void pair_products(const f32x4 (&w)[T], const u32x4 (&x)[T], f32x16& acc) {
    if (T == 2) {
        acc = BFQ_MFMA16(w[1], x[1], acc);
        acc = BFQ_MFMA16(w[1], x[0], acc);
        acc = BFQ_MFMA16(w[0], x[1], acc);
        acc = BFQ_MFMA16(w[0], x[0], acc);
    } else {
        acc = BFQ_MFMA16(w[T - 1], x[0], acc);
        acc = BFQ_MFMA16(w[0], x[T - 1], acc);
        acc = BFQ_MFMA16(w[1], x[1], acc);
        acc = BFQ_MFMA16(w[1], x[0], acc);
        acc = BFQ_MFMA16(w[0], x[1], acc);
        acc = BFQ_MFMA16(w[0], x[0], acc);
    }
}

// MODE 0: scores; 2: x9 features as float32 [board][32][64] into `scores` (the move-scoring network's first stage).
// LEVEL 0: float32 (k_eval_f32<., 0>); 1, 2: BF_PRECISION_F32_SPLIT / _SPLIT3 (k_eval_f32<., 1 / 2>): the 1x1 pair split
// over the waves as for float32, the hidden layer exchanged as float32 and split into its bf16 terms by its readers,
// the 3x3 layers (there a fifth of the time, on the bf16 pipe) by waves (half, 0) alone on the 32 x 32 tiles of the
// large-batch kernel, the head by the same two waves from their registers.
template <int MODE, int LEVEL>
__global__ __launch_bounds__(QTHREADS, 1) void k_eval_f32q(const uint8_t* __restrict__ mailbox, unsigned n_boards,
                                                           const f32x4* __restrict__ wstream, const float* __restrict__ bias,
                                                           const f32x4* __restrict__ headw, float* __restrict__ scores) {
    __shared__ __attribute__((aligned(16))) LdsQ lds;
    const int tid = threadIdx.x, lane = tid & 63, hl = lane >> 5, ln = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave & 1, m = wave >> 1;
    const unsigned board = blockIdx.x;   // the grid is exactly the boards
    (void)n_boards;
    const int sq = 32 * half + ln, rank = sq >> 3, file = sq & 7;

#ifdef BFQ_STAMPS
    long long stamps[8];
#endif
    BFQ_STAMP(0);
    if (tid < MB_BYTES / 4) reinterpret_cast<unsigned*>(lds.mb)[tid] = 0xFFFFFFFFu;
    __syncthreads();
    if (tid < 64) {
        const unsigned v = mailbox[(size_t)board * kMailboxBytes + (tid >> 1)];
        const unsigned c = (tid & 1) ? (v >> 4) : (v & 15u);
        lds.mb[((tid >> 3) + 7) * MB_W + (tid & 7) + 7] = (unsigned char)(c == 15u ? 0xFFu : c);
    }
    __syncthreads();

    // ---- stage 1: the eight waves share the nine input convs, each computing its row tiles for BOTH halves of the board
    // (9, 9, 9, 9, 9, 9, 13, 13 (tap, tile) pairs: every weight group crosses the L1 once per workgroup), and hand their
    // channels to each other through LDS; then every wave holds x1 of its half in 118 registers
    float x1[kX1Regs];
    {
        const unsigned char* mb0 = &lds.mb[((ln >> 3) + 7) * MB_W + (ln & 7) + 7];   // square ln of half 0
        const unsigned char* mb1 = mb0 + 4 * MB_W;                                   // square 32 + ln
        float* xe = lds.region + lane;
        switch (wave) {
            case 0: input_conv<0, 0, 1>(wstream, lane, mb0, mb1, hl, bias, xe); break;
            case 1: input_conv<0, 1, 1>(wstream, lane, mb0, mb1, hl, bias, xe); break;
            case 2: input_conv<1, 0, 1>(wstream, lane, mb0, mb1, hl, bias, xe); break;
            case 3: input_conv<2, 0, 1>(wstream, lane, mb0, mb1, hl, bias, xe); break;
            case 4: input_conv<3, 0, 1>(wstream, lane, mb0, mb1, hl, bias, xe); break;
            case 5: input_conv<4, 0, 1>(wstream, lane, mb0, mb1, hl, bias, xe); break;
            case 6:
                input_conv<5, 0, 1>(wstream, lane, mb0, mb1, hl, bias, xe);
                input_conv<7, 0, 1>(wstream, lane, mb0, mb1, hl, bias, xe);
                break;
            default:
                input_conv<6, 0, 1>(wstream, lane, mb0, mb1, hl, bias, xe);
                input_conv<8, 0, 1>(wstream, lane, mb0, mb1, hl, bias, xe);
                break;
        }
        float* xl = lds.region + half * 64 + lane;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kX1Regs; ++r) x1[r] = xl[r * 128];
        __syncthreads();   // the hidden layer overwrites the exchange
    }

    if constexpr (LEVEL > 0) {
        constexpr int T = LEVEL + 1;                                        // terms a side in the 1x1 pair
        constexpr int PER = LEVEL == 2 ? kPerC2GroupsSplit3 : kPerC2Groups;  // groups a hidden tile: 69 / 46
        constexpr int G1 = 15 * T, GP = 2 * T;                              // 236 -> 512 groups a tile; 512 -> 128 groups a (tile, c3)
        constexpr int PD = LEVEL == 2 ? 7 : 11;                             // a group feeds two 32-cycle MFMAs: deep
        // x1 as bf16 terms, packed for the B operand (k-step s takes registers 8s .. 8s+7; register 118 of lane half 0
        // is the all-ones slot that carries the hidden layer's shift)
        unsigned xs[T][60];
#pragma unroll
        for (int d = 0; d < 60; ++d) {
            const float a = 2 * d < kX1Regs ? x1[2 * d] : (2 * d == kX1Regs ? pick(hl, 1.0f, 0.0f) : 0.0f);
            const float b = 2 * d + 1 < kX1Regs ? x1[2 * d + 1] : 0.0f;
            if constexpr (T == 2) split2(a, b, xs[0][d], xs[1][d]);
            else split3(a, b, xs[0][d], xs[1][d], xs[2][d]);
        }
        BFQ_STAMP(1);
        {   // 236 -> 512: hidden tiles 4m .. 4m+3, ReLU, to LDS as float32 (their readers split them)
            auto off = [](int i) { return ((i / G1) * PER + i % G1) * 64; };
            GStream<PD, decltype(off)> ws(wstream + (size_t)(kInputGroups + 4 * m * PER) * 64 + lane, off, 4 * G1);
            float* hid = lds.region + ((half * 16 + 4 * m) * 16) * 64 + lane;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x16 acc2 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 15; ++k) {
                    f32x4 w[T];
                    u32x4 x[T];
#pragma unroll
                    for (int term = 0; term < T; ++term) {
                        w[term] = ws.take(u * G1 + T * k + term);
                        x[term] = u32x4{xs[term][4 * k], xs[term][4 * k + 1], xs[term][4 * k + 2], xs[term][4 * k + 3]};
                    }
                    pair_products<T>(w, x, acc2);
                }
#pragma unroll
                for (int t = 0; t < 16; ++t) hid[(u * 16 + t) * 64] = relu(acc2[t]);
            }
        }
        BFQ_STAMP(2);
        __syncthreads();
        BFQ_STAMP(3);
        // 512 -> 128: output tile m over the sixteen hidden tiles in order
        f32x16 acc3;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc3[t] = pick(hl, bias[kBiasX3 + 32 * m + rowmap(t, 0)], bias[kBiasX3 + 32 * m + rowmap(t, 1)]);
        {
            auto off = [](int i) { return ((i / GP) * PER + i % GP) * 64; };
            GStream<PD, decltype(off)> ws(wstream + (size_t)(kInputGroups + G1 + m * GP) * 64 + lane, off, 16 * GP);
            const float* hid = lds.region + (half * 16 * 16) * 64 + lane;
            float h[2][16];
#pragma unroll
            for (int t = 0; t < 16; ++t) h[0][t] = hid[t * 64];
#pragma unroll
            for (int c2 = 0; c2 < 16; ++c2) {
                if (c2 + 1 < 16) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) h[(c2 + 1) & 1][t] = hid[((c2 + 1) * 16 + t) * 64];
                }
                unsigned hs[T][8];
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    if constexpr (T == 2) split2(h[c2 & 1][2 * d], h[c2 & 1][2 * d + 1], hs[0][d], hs[1][d]);
                    else split3(h[c2 & 1][2 * d], h[c2 & 1][2 * d + 1], hs[0][d], hs[1][d], hs[2][d]);
                }
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    f32x4 w[T];
                    u32x4 x[T];
#pragma unroll
                    for (int term = 0; term < T; ++term) {
                        w[term] = ws.take(c2 * GP + T * v + term);
                        x[term] = u32x4{hs[term][4 * v], hs[term][4 * v + 1], hs[term][4 * v + 2], hs[term][4 * v + 3]};
                    }
                    pair_products<T>(w, x, acc3);
                }
            }
        }
        BFQ_STAMP(4);
        __syncthreads();   // nobody reads the hidden layer any more: its space becomes the term planes
        {
            f32x4* z = reinterpret_cast<f32x4*>(lds.region);
            for (int i = tid; i < XS_SETS * XS_SET_BYTES / 16; i += QTHREADS) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        char* sets = reinterpret_cast<char*>(lds.region);
        int xa[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[j] = (rank * XP_W + file) * XS_POS_BYTES + ((j + hl + rank) & 3) * 16;
        int wa[2];
#pragma unroll
        for (int cq = 0; cq < 2; ++cq) wa[cq] = ((rank + 1) * XP_W + file + 1) * XS_POS_BYTES + ((2 * hl + cq + rank + 1) & 3) * 16;
        auto store_planes = [&](char* set, const f32x16& v) {   // ReLU, three terms, 2 x 3 16-byte stores (eval_f32.hip)
#pragma unroll
            for (int cq = 0; cq < 2; ++cq) {
                u32x4 th, tm, tl;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    unsigned h_, m_, l_;
                    split3(relu(v[8 * cq + 2 * d]), relu(v[8 * cq + 2 * d + 1]), h_, m_, l_);
                    th[d] = h_; tm[d] = m_; tl[d] = l_;
                }
                *reinterpret_cast<u32x4*>(set + wa[cq]) = th;
                *reinterpret_cast<u32x4*>(set + wa[cq] + XS_PLANE_BYTES) = tm;
                *reinterpret_cast<u32x4*>(set + wa[cq] + 2 * XS_PLANE_BYTES) = tl;
            }
        };
        store_planes(sets + m * XS_SET_BYTES, acc3);   // x3: this wave's chunk m of its half board
        __syncthreads();
        BFQ_STAMP(5);
        // the six 3x3 layers: waves (half, 0) on the large-batch kernel's 32 x 32 tiles; the others keep the barriers
        f32x16 acc;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = pick(hl, bias[kBiasX4 + rowmap(t, 0)], bias[kBiasX4 + rowmap(t, 1)]);
        auto coff = [](int i) { return i * 64; };
        GStream<11, decltype(coff)> cs(wstream + (size_t)(kInputGroups + kC2Tiles * PER) * 64 + lane, coff,
                                       m == 0 ? 9 * kConvSplitChunkGroups : 0);
        if (m == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) conv3x3_chunk_split_q(cs, q * kConvSplitChunkGroups, sets + q * XS_SET_BYTES, xa, acc);
        }
        BFQ_STAMP(6);
#pragma unroll
        for (int layer = 0; layer < 5; ++layer) {
            char* set = sets + (4 + (layer & 1)) * XS_SET_BYTES;
            if (m == 0) store_planes(set, acc);
            __syncthreads();
            if (m == 0) {
                const float* bl = bias + kBiasX4 + 32 * (layer + 1);
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[t] = pick(hl, bl[rowmap(t, 0)], bl[rowmap(t, 1)]);
                conv3x3_chunk_split_q(cs, (4 + layer) * kConvSplitChunkGroups, set, xa, acc);
            }
        }
        if (MODE == 2) {
            if (m == 0) {
                float* out = scores + (size_t)board * 2048;
#pragma unroll
                for (int t = 0; t < 16; ++t) out[rowmap(t, hl) * 64 + sq] = relu(acc[t]);
            }
#ifdef BFQ_STAMPS
            BFQ_STAMP(7);
            if (board == 0 && lane == 0) {
                long long* so = reinterpret_cast<long long*>(scores + (size_t)n_boards * 2048) + wave * 8;
                for (int i = 0; i < 8; ++i) so[i] = stamps[i];
            }
#endif
            return;
        }
        // head: k_eval_f32's, by the two waves that hold x9 (threads 0 .. 127), scratch behind the planes
        float* red = lds.region + XS_RED_OFF;
        if (tid < 128) {
            float part[16];
#pragma unroll
            for (int co = 0; co < 16; ++co) part[co] = 0.0f;
            const f32x4* hw = headw + (size_t)half * 64 * 64 + lane;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float v = relu(acc[t]);
#pragma unroll
                for (int cq = 0; cq < 4; ++cq) {
                    const f32x4 w = hw[(t * 4 + cq) * 64];
                    part[4 * cq + 0] = fmaf(v, w.x, part[4 * cq + 0]);
                    part[4 * cq + 1] = fmaf(v, w.y, part[4 * cq + 1]);
                    part[4 * cq + 2] = fmaf(v, w.z, part[4 * cq + 2]);
                    part[4 * cq + 3] = fmaf(v, w.w, part[4 * cq + 3]);
                }
            }
#pragma unroll
            for (int co = 0; co < 16; ++co) red[tid * RED_STRIDE + co] = part[co];
        }
        __syncthreads();
        if (tid < 64) {
            const int co = tid & 15, seg = tid >> 4;
            float s = 0.0f;
            for (int j = 0; j < 32; ++j) s += red[(seg * 32 + j) * RED_STRIDE + co];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const float y = fmaxf(s + bias[kBiasX10 + co], 0.0f);
            float v = y * bias[kLogitW + co];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            if (tid == 0) scores[board] = v;
        }
    } else {
    BFQ_STAMP(1);
    // ---- 236 -> 512: hidden tiles 4m .. 4m+3, ReLU, to LDS as the B fragments of the next layer
    {
        auto off = [](int i) { return ((i / kGemm1Groups) * kPerC2Groups + i % kGemm1Groups) * 64; };
        GStream<5, decltype(off)> ws(wstream + (size_t)(kInputGroups + 4 * m * kPerC2Groups) * 64 + lane, off, 4 * kGemm1Groups);
        float* hid = lds.region + ((half * 16 + 4 * m) * 16) * 64 + lane;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x16 acc2 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < kGemm1Groups; ++g) {
                const f32x4 w = ws.take(u * kGemm1Groups + g);
                acc2 = BFQ_MFMA(w.x, x1[4 * g + 0], acc2);
                acc2 = BFQ_MFMA(w.y, x1[4 * g + 1], acc2);
                if (4 * g + 2 < kX1Regs) {
                    acc2 = BFQ_MFMA(w.z, x1[4 * g + 2], acc2);
                    acc2 = BFQ_MFMA(w.w, x1[4 * g + 3], acc2);
                } else {
                    acc2 = BFQ_MFMA(w.z, 1.0f, acc2);   // k-step 118: the folded batch-norm shift, B = 1
                }
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) hid[(u * 16 + t) * 64] = relu(acc2[t]);
        }
    }
    BFQ_STAMP(2);
    __syncthreads();
    BFQ_STAMP(3);

    // ---- 512 -> 128: output tile m over the sixteen hidden tiles in order
    f32x16 acc3;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc3[t] = pick(hl, bias[kBiasX3 + 32 * m + rowmap(t, 0)], bias[kBiasX3 + 32 * m + rowmap(t, 1)]);
    {
        auto off = [](int i) { return ((i >> 2) * kPerC2Groups + (i & 3)) * 64; };
        GStream<5, decltype(off)> ws(wstream + (size_t)(kInputGroups + kGemm1Groups + 4 * m) * 64 + lane, off, 64);
        const float* hid = lds.region + (half * 16 * 16) * 64 + lane;
        float hb[2][4];   // the hidden layer's B operands, one group ahead
#pragma unroll
        for (int j = 0; j < 4; ++j) hb[0][j] = hid[j * 64];
#pragma unroll
        for (int i = 0; i < 64; ++i) {   // i = 4 c2 + tq: hidden rows 4 i .. 4 i + 3 of the half's 256 registers
            if (i + 1 < 64) {
#pragma unroll
                for (int j = 0; j < 4; ++j) hb[(i + 1) & 1][j] = hid[(4 * (i + 1) + j) * 64];
            }
            const f32x4 w = ws.take(i);
            acc3 = BFQ_MFMA(w.x, hb[i & 1][0], acc3);
            acc3 = BFQ_MFMA(w.y, hb[i & 1][1], acc3);
            acc3 = BFQ_MFMA(w.z, hb[i & 1][2], acc3);
            acc3 = BFQ_MFMA(w.w, hb[i & 1][3], acc3);
        }
    }
    BFQ_STAMP(4);
    const int mt = wave >> 2, nt = wave & 3;
    const int q4 = lane >> 4, r = lane & 15;
    const f32x4* wconv = wstream + (size_t)(kInputGroups + kC2Tiles * kPerC2Groups) * 64;   // the nine chunks, contiguous
    ChunkFetch cf;
    cf.load(wconv, tid);   // chunk 0 is asked for now: it arrives while the planes are zeroed
    __syncthreads();       // nobody reads the hidden layer any more: its space becomes planes and weight buffers
    {
        f32x4* z = reinterpret_cast<f32x4*>(lds.region);
        for (int i = tid; i < N_PLANES * PLANE_FLOATS / 4; i += QTHREADS) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    cf.store(lds.region + WBUF_OFF, tid);
    __syncthreads();
    // x3: this wave's 32 channels of its half board go into plane m
    {
        float* xb = lds.region + m * PLANE_FLOATS + (rank + 1) * XP_W + file + 1;
#pragma unroll
        for (int t = 0; t < 16; ++t) xb[rowmap(t, hl) * XP_PLANE] = relu(acc3[t]);
    }
    __syncthreads();

    // ---- the six 3x3 layers on 16 x 16 tiles: wave = (16 output channels mt) x (16 squares nt).  Chunk c (four of
    // the 128 -> 32 layer, then one per 32 -> 32 layer) reads weight buffer c & 1 while the workgroup fetches chunk
    // c + 1 for the other one; one barrier a chunk.
    const int csq = 16 * nt + r, crank = csq >> 3, cfile = csq & 7;
    const int rpos = q4 * XP_PLANE + crank * XP_W + cfile;       // channel q4, one row and column back
    const int wpos = (16 * mt + 4 * q4) * XP_PLANE + (crank + 1) * XP_W + cfile + 1;
    const int apos = (q4 & 1) * 32 + 16 * mt + r;   // this lane's float4 in a group
    BFQ_STAMP(5);
    f32x4 acc;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = bias[kBiasX4 + 16 * mt + 4 * q4 + t];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        if (c + 1 < 9) cf.load(wconv + (size_t)(c + 1) * (CHUNK_W_FLOATS / 4), tid);
        const float* plane = lds.region + (c < 4 ? c : ((c - 4) & 1)) * PLANE_FLOATS;
        conv3x3_chunk_q(reinterpret_cast<const f32x4*>(lds.region + WBUF_OFF + (c & 1) * CHUNK_W_FLOATS) + apos, q4 >> 1, plane + rpos, acc);
        if (c + 1 < 9) cf.store(lds.region + WBUF_OFF + ((c + 1) & 1) * CHUNK_W_FLOATS, tid);
        if (c == 3) BFQ_STAMP(6);
        if (c >= 3 && c + 1 < 9) {   // a layer is complete: its output is the next layer's plane
            if (c == 3) __syncthreads();   // (planes 0 and 1 held x3: every wave must be done with the 128 -> 32 layer)
            float* xb = lds.region + ((c - 3) & 1) * PLANE_FLOATS;
#pragma unroll
            for (int t = 0; t < 4; ++t) xb[wpos + t * XP_PLANE] = relu(acc[t]);
            const float* bl = bias + kBiasX4 + 32 * (c - 2) + 16 * mt + 4 * q4;
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = bl[t];
        }
        __syncthreads();
    }

    if (MODE == 2) {
        float* out = scores + (size_t)board * 2048 + (16 * mt + 4 * q4) * 64 + csq;
#pragma unroll
        for (int t = 0; t < 4; ++t) out[t * 64] = relu(acc[t]);
#ifdef BFQ_STAMPS
        BFQ_STAMP(7);
        if (board == 0 && lane == 0) {
            long long* so = reinterpret_cast<long long*>(scores + (size_t)n_boards * 2048) + wave * 8;
            for (int i = 0; i < 8; ++i) so[i] = stamps[i];
        }
#endif
        return;
    }
    // ---- head: x9 to LDS, then k_eval_f32's per-lane partial sums and its reduction, by threads 0 .. 127
    {
        float* x9 = lds.region + X9_OFF + (16 * mt + 4 * q4) * 64 + csq;
#pragma unroll
        for (int t = 0; t < 4; ++t) x9[t * 64] = relu(acc[t]);
    }
    __syncthreads();
    float* red = lds.region + RED_OFF;
    if (tid < 128) {   // thread = (half, lane) of k_eval_f32's board: waves 0 and 1 are halves 0 and 1
        float part[16];
#pragma unroll
        for (int co = 0; co < 16; ++co) part[co] = 0.0f;
        const f32x4* hw = headw + (size_t)half * 64 * 64 + lane;
        const float* x9 = lds.region + X9_OFF + sq;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float v = x9[rowmap(t, hl) * 64];
#pragma unroll
            for (int cq = 0; cq < 4; ++cq) {
                const f32x4 w = hw[(t * 4 + cq) * 64];
                part[4 * cq + 0] = fmaf(v, w.x, part[4 * cq + 0]);
                part[4 * cq + 1] = fmaf(v, w.y, part[4 * cq + 1]);
                part[4 * cq + 2] = fmaf(v, w.z, part[4 * cq + 2]);
                part[4 * cq + 3] = fmaf(v, w.w, part[4 * cq + 3]);
            }
        }
#pragma unroll
        for (int co = 0; co < 16; ++co) red[tid * RED_STRIDE + co] = part[co];
    }
    __syncthreads();
    if (tid < 64) {
        const int co = tid & 15, seg = tid >> 4;
        float s = 0.0f;
        for (int j = 0; j < 32; ++j) s += red[(seg * 32 + j) * RED_STRIDE + co];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float y = fmaxf(s + bias[kBiasX10 + co], 0.0f);
        float v = y * bias[kLogitW + co];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        if (tid == 0) scores[board] = v;
    }
    }
}

}  // namespace

template <int MODE>
static hipError_t launch_q(int split, const uint8_t* d_mailbox, size_t n, const float* d_packed, float* d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const f32x4* ws = reinterpret_cast<const f32x4*>(d_packed + kOffStream);
    const float* bias = d_packed + kOffBias;
    const f32x4* head = reinterpret_cast<const f32x4*>(d_packed + kOffHead);
    if (split == 2)
        hipLaunchKernelGGL((k_eval_f32q<MODE, 2>), dim3((unsigned)n), dim3(QTHREADS), 0, stream, d_mailbox, (unsigned)n, ws, bias, head, d_out);
    else if (split == 1)
        hipLaunchKernelGGL((k_eval_f32q<MODE, 1>), dim3((unsigned)n), dim3(QTHREADS), 0, stream, d_mailbox, (unsigned)n, ws, bias, head, d_out);
    else
        hipLaunchKernelGGL((k_eval_f32q<MODE, 0>), dim3((unsigned)n), dim3(QTHREADS), 0, stream, d_mailbox, (unsigned)n, ws, bias, head, d_out);
    return hipGetLastError();
}

hipError_t launch_eval_f32q(const uint8_t* d_mailbox, size_t n, const float* d_packed, float* d_scores, hipStream_t stream, int split) {
    return launch_q<0>(split, d_mailbox, n, d_packed, d_scores, stream);
}

hipError_t launch_policy_features_f32q(const uint8_t* d_mailbox, size_t n, const float* d_packed, float* d_x9, hipStream_t stream,
                                       int split) {
    return launch_q<2>(split, d_mailbox, n, d_packed, d_x9, stream);
}

}  // namespace bf
