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
//   stage 1   the four waves of a half board share the nine input convs (the bf16 three-term form of k_eval_f32, same
//             instruction, same sequence per accumulator) and exchange x1 through LDS; every wave then holds its
//             half's x1 in 118 registers;
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
// Measured: 100-103 us a call for 1 .. 256 boards against 154-158 us on k_eval_f32 (wave life 190 k cycles: stage 1
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

// Stage 1, one input conv: k_eval_f32's input_conv, instruction for instruction, with its weights straight from L2
// (a stream of its own per conv, 15 groups ahead: one 32-cycle MFMA a group) and its result, after ReLU, to the x1
// exchange in LDS: x1buf[register][half][lane], register r = channels 2r / 2r+1 of the lane halves as in k_eval_f32.
template <int I>
__device__ __forceinline__ void input_conv(const f32x4* __restrict__ wstream, int lane, const unsigned char* mb_center, int hl,
                                           const float* __restrict__ bias, float* x1buf_lane) {
    constexpr InputConv cv = kInputConvs[I];
    constexpr int G0 = input_group_base(I), NG = cv.k * cv.k * cv.tiles * kInputSplits;
    constexpr int ph = ((cv.k - 1) * cv.dh) / 2, pw = ((cv.k - 1) * cv.dw) / 2;
    auto off = [](int i) { return i * 64; };
    GStream<15, decltype(off)> ws(wstream + (size_t)G0 * 64 + lane, off, NG);
    f32x16 acc[cv.tiles];
#pragma unroll
    for (int tile = 0; tile < cv.tiles; ++tile)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int c = 32 * tile + 2 * t;
            acc[tile][t] = (c < cv.ch) ? pick(hl, bias[kBiasX1 + cv.chan_base + c], bias[kBiasX1 + cv.chan_base + c + 1]) : 0.0f;
        }
#pragma unroll
    for (int tap = 0; tap < cv.k * cv.k; ++tap) {
        const int dr = (tap / cv.k) * cv.dh - ph, df = (tap % cv.k) * cv.dw - pw;
        const unsigned cc = (unsigned)((int)mb_center[dr * MB_W + df] - 8 * hl);
        const unsigned one = (cc < 8u) ? (0x3F80u << ((cc & 1u) * 16u)) : 0u;
        u32x4 bu;
        bu.x = ((cc >> 1) == 0u) ? one : 0u;
        bu.y = ((cc >> 1) == 1u) ? one : 0u;
        bu.z = ((cc >> 1) == 2u) ? one : 0u;
        bu.w = ((cc >> 1) == 3u) ? one : 0u;
        const bf16x8 b = __builtin_bit_cast(bf16x8, bu);
#pragma unroll
        for (int tile = 0; tile < cv.tiles; ++tile)
#pragma unroll
            for (int sp = 0; sp < kInputSplits; ++sp) {
                const f32x4 w = ws.take((tap * cv.tiles + tile) * kInputSplits + sp);
                acc[tile] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), b, acc[tile], 0, 0, 0);
            }
    }
#pragma unroll
    for (int tile = 0; tile < cv.tiles; ++tile)
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (32 * tile + 2 * t < cv.ch) x1buf_lane[(cv.chan_base / 2 + 16 * tile + t) * 128] = relu(acc[tile][t]);
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

// MODE 0: scores; 2: x9 features as float32 [board][32][64] into `scores` (the move-scoring network's first stage).
template <int MODE>
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

    // ---- stage 1: the four waves of a half board share the nine input convs (18 / 18 / 22 / 22 (tap, tile) pairs) and
    // hand their channels to each other through LDS; then every wave holds x1 of its half in 118 registers
    float x1[kX1Regs];
    {
        const unsigned char* mbc = &lds.mb[(rank + 7) * MB_W + file + 7];
        float* xl = lds.region + half * 64 + lane;
        if (m == 0) {
            input_conv<0>(wstream, lane, mbc, hl, bias, xl);
        } else if (m == 1) {
            input_conv<1>(wstream, lane, mbc, hl, bias, xl);
            input_conv<2>(wstream, lane, mbc, hl, bias, xl);
        } else if (m == 2) {
            input_conv<3>(wstream, lane, mbc, hl, bias, xl);
            input_conv<4>(wstream, lane, mbc, hl, bias, xl);
            input_conv<7>(wstream, lane, mbc, hl, bias, xl);
        } else {
            input_conv<5>(wstream, lane, mbc, hl, bias, xl);
            input_conv<6>(wstream, lane, mbc, hl, bias, xl);
            input_conv<8>(wstream, lane, mbc, hl, bias, xl);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kX1Regs; ++r) x1[r] = xl[r * 128];
        __syncthreads();   // the hidden layer overwrites the exchange
    }

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

}  // namespace

hipError_t launch_eval_f32q(const uint8_t* d_mailbox, size_t n, const float* d_packed, float* d_scores, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL((k_eval_f32q<0>), dim3((unsigned)n), dim3(QTHREADS), 0, stream, d_mailbox, (unsigned)n,
                       reinterpret_cast<const f32x4*>(d_packed + kOffStream), d_packed + kOffBias,
                       reinterpret_cast<const f32x4*>(d_packed + kOffHead), d_scores);
    return hipGetLastError();
}

hipError_t launch_policy_features_f32q(const uint8_t* d_mailbox, size_t n, const float* d_packed, float* d_x9, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL((k_eval_f32q<2>), dim3((unsigned)n), dim3(QTHREADS), 0, stream, d_mailbox, (unsigned)n,
                       reinterpret_cast<const f32x4*>(d_packed + kOffStream), d_packed + kOffBias,
                       reinterpret_cast<const f32x4*>(d_packed + kOffHead), d_x9);
    return hipGetLastError();
}

}  // namespace bf
