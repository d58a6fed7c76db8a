// The value network as one bf16 MFMA kernel (BASELINE configs[2]): 4-bit mailboxes in, one score
// per board out.  Arithmetic and weight image: bf_layout_bf16.h (bf16 weights and activations,
// float32 sums and shifts).
//
// Reference graph replaced (TensorFlow ops -> this kernel):
//   create_input_convolutions_shared_weights   ann_creation_helper.py:192-252  (stage 1)
//   build_convolutional_modules, spec at       evaluation_ann.py:268           (stages 2-4)
//   logit_layer + squeeze                      evaluation_ann.py:35-46         (stage 5)
//
// One wave = one board: its 64 squares are the columns of two v_mfma_f32_32x32x16_bf16 tiles
// (tile T = squares 32T .. 32T+31), and every A fragment (32 output channels x 16 k of weights)
// read from LDS feeds two MFMAs.  Measured on the first version (one wave per half board, one
// MFMA per fragment read): the kernel was bound by LDS traffic -- the fragment reads plus the
// LDS-DMA writes that fill the ring -- not by the matrix pipe; removing either in an ablation
// build took 0.6 ms off 2.1 ms.  A whole board per wave halves both.  The price is registers
// (x1 and the 128-channel accumulators of both tiles live in them: ~330), so one wave per SIMD.
// Layers run transposed, out^T[channel][square] = W^T[channel][k] * in^T[k][square], so that a
// finished accumulator tile, converted to bf16 in place, is the B operand of the next 1x1 layer
// (k order permuted on the host to match the accumulator's row map).
//   stage 1  B operand of a tap = the shifted square's one-hot over 16 planes: a code byte from a
//            padded mailbox in LDS, then one ds_read_b128 of that code's row of a 16-entry table.
//            The 16th plane is 1.0 everywhere and carries the folded batch-norm shift.
//   stage 2  per 32-channel hidden tile: 15 fragments (236->512), ReLU + pack, 8 fragments (512->128).
//   stage 3/4  3x3 layers through zero-haloed [10x10 positions][32 channels] bf16 planes in LDS,
//            private to the wave (LDS serves a wave's accesses in order: no barrier between the
//            writes of a layer and the shifted reads of the next); the 16-byte channel chunk c of
//            a position in padded row R sits at chunk (c + R) & 3, which makes the shifted
//            ds_read_b128 of every tap conflict-free.
//   stage 5  8x8 valid conv across the boards of the workgroup: out[co][board] on
//            v_mfma_f32_16x16x32_bf16, one k-step per square, squares split over the waves, fixed-
//            order reduction of the per-wave partial sums; ReLU; logit.
// Weights stream through a lockstep LDS ring as in the fp32 kernel (LDS-DMA into a 3-slot ring,
// one barrier per chunk), with one 1-KiB group per fragment and a 3-group register look-ahead.
// Results are bit-reproducible and independent of a board's position in the batch.
#include <hip/hip_runtime.h>

#include "bf_device.h"
#include "bf_kernels.h"
#include "bf_layout_bf16.h"

namespace bf {

namespace b16 {  // inner scope: the bf16 layout's names hide the fp32 layout's

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifndef BF16_BPB
#define BF16_BPB 4
#endif
constexpr int BPB = BF16_BPB;            // boards per workgroup = waves per workgroup
constexpr int WAVES = BPB;
constexpr int THREADS = 64 * WAVES;
#ifndef BF16_CHUNK
#define BF16_CHUNK 16
#endif
constexpr int CHUNK = BF16_CHUNK;           // groups per lockstep chunk
constexpr int SHARE = CHUNK / WAVES;        // groups each wave fetches per chunk
#ifndef BF16_SLOTS
#define BF16_SLOTS 4
#endif
constexpr int SLOTS = BF16_SLOTS;           // ring slots: one being read, one just read, SLOTS-2 landed or in flight
#ifndef BF16_LA
#define BF16_LA 7
#endif
constexpr int LA = BF16_LA;                       // groups held in registers ahead of the MFMA
#ifndef BF16_C2U
#define BF16_C2U 16
#endif
constexpr int C2_UNROLL = BF16_C2U;
static_assert(CHUNK <= kStreamChunkMax, "");
static_assert(CHUNK % WAVES == 0, "every wave fetches whole groups");
static_assert(kInputGroups % CHUNK == 0 && kInputGroups % (LA + 1) == 0, "the tile loop starts on a chunk boundary");
static_assert((kPerC2Groups * C2_UNROLL) % CHUNK == 0 && (kPerC2Groups * C2_UNROLL) % (LA + 1) == 0,
              "the loop body is whole chunks and keeps the look-ahead phase");
static_assert(kC2Tiles % C2_UNROLL == 0, "");
static_assert(LA <= CHUNK && SHARE <= CHUNK, "");

// Stage 1 finds the B operand of a tap in two steps: the shifted square's code from a byte
// mailbox padded by 7 squares on every side (22 x 22, 15 = empty or off the board), then the
// code's one-hot row from a 16-entry table shared by the workgroup (32 bytes per code: planes
// 0-7 for lane half 0, planes 8-15 for half 1; plane 15 is 1.0 in every entry).
constexpr int MB_W = 22;
constexpr int MB_WORDS = 124;               // 22*22 = 484 bytes, rounded up to whole dwords
constexpr int XP_W = 10;                    // activation plane padded by 1
constexpr int POS_BYTES = 64;               // 32 channels bf16
constexpr int PLANE_BYTES = XP_W * XP_W * POS_BYTES;  // 6400
constexpr int BOARD_BYTES = 2 * PLANE_BYTES;          // two planes; stage 1 keeps the byte mailbox in the first
static_assert(MB_WORDS * 4 <= PLANE_BYTES && 1024 <= PLANE_BYTES, "mailbox and head partial sums alias plane 0");
constexpr int TILE1 = 4 * XP_W * POS_BYTES;           // plane offset of tile 1's squares: 4 ranks up
static_assert(BPB <= 16, "the head puts boards on the 16 columns of one MFMA");

struct Lds {
    f32x4 wring[SLOTS][CHUNK][64];                       // shared A-fragment ring
    __attribute__((aligned(16))) unsigned char board[BPB][BOARD_BYTES];
    u32x4 onehot[16][2];                                 // stage 1: [code][lane half] -> 8 bf16 planes
};

// Workgroup barrier for LDS hand-offs: every LDS access of this wave has completed; global
// memory and the LDS-DMA queue are left alone (__syncthreads() would drain them: vmcnt(0)).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Chunk barrier of the weight ring: this wave's share of the chunk about to be read has landed.
// The wave's LDS-DMA queue holds, oldest first, its shares of that chunk and of the SLOTS-3 chunks
// after it; vector-memory operations retire in order, so leaving (SLOTS-3)*SHARE of them
// outstanding is enough (younger ordinary loads only make the wait stricter).  No LDS wait: the
// slot refilled after the barrier is that of the chunk BEFORE the one just finished, whose reads
// were issued more than CHUNK groups ago and have returned (LDS returns a wave's reads in order,
// and every group waits for its own fragment, issued LA groups earlier, before its MFMAs).
__device__ __forceinline__ void ring_barrier() {
    constexpr int N = (SLOTS - 3) * SHARE;
    static_assert(SLOTS >= 3 && N < 64, "vmcnt is a 6-bit counter");
#ifdef BF16_ABL_NOBAR
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

// Lockstep walk over the A-fragment stream through the LDS ring (see eval_f32.hip).
struct WStream {
    const f32x4* src;
    f32x4* ring;
    f32x4* share;
    int rd;                // float4 offset of the slot the look-ahead reads from
    int fill;              // float4 offset of the slot being refilled
    f32x4 pre[LA + 1];

    // LDS-DMA: one 1-KiB piece of this wave's share straight from L2 into a ring slot, lane-linear.
    // Issued as inline asm on purpose: the compiler models global_load_lds as a FLAT access that
    // may touch LDS, and while one is outstanding it turns every LDS wait into lgkmcnt(0), which
    // stalls each group of MFMAs on the read just issued.  ring_barrier() does the vmcnt
    // bookkeeping instead; ordinary loads interleaved by the compiler only make its own vmcnt
    // waits stricter (vector-memory operations retire in order).
    // Issuing a piece holds the wave for 60+ cycles, so the SHARE pieces of a refill are spread over
    // the first SHARE groups after the chunk barrier: each one issues while the two MFMAs of the
    // previous group execute.
    __device__ __forceinline__ void fetch_piece(int slot_off, int i) {
        typedef __attribute__((address_space(3))) f32x4* lds_ptr;
        const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)(share + slot_off);
#ifndef BF16_ABL_NODMA
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                     :: "s"(lds0 + i * 1024), "v"(src + i * 64) : "memory", "m0");
#endif
        if (i == SHARE - 1) src += CHUNK * 64;
    }
    __device__ __forceinline__ void prime(const f32x4* stream, f32x4* ring_base, int lane, int wave) {
        ring = ring_base + lane;
        share = ring_base + wave * SHARE * 64;
        src = stream + wave * SHARE * 64 + lane;
#pragma unroll
        for (int c = 0; c < SLOTS - 2; ++c)
#pragma unroll
            for (int i = 0; i < SHARE; ++i) fetch_piece(c * CHUNK * 64, i);
        rd = (SLOTS - 1) * CHUNK * 64;
    }
    // After the barrier every wave's share of chunk k is in LDS and nobody reads chunk k-2 any
    // more, so its slot takes chunk k+SLOTS-2.
    __device__ __forceinline__ void next_chunk() {
        rd = (rd == (SLOTS - 1) * CHUNK * 64) ? 0 : rd + CHUNK * 64;
        fill = rd + (SLOTS - 2) * CHUNK * 64;
        if (fill >= SLOTS * CHUNK * 64) fill -= SLOTS * CHUNK * 64;
        __builtin_amdgcn_sched_barrier(0);
        ring_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void start() {
        next_chunk();
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            if (i < SHARE) fetch_piece(fill, i);
            pre[i] = ring[rd + i * 64];
        }
    }
    // g: stream group index, a compile-time constant after unrolling; groups are taken in order.
    __device__ __forceinline__ bf16x8 take(int g) {
        const int nx = g + LA, i = nx % CHUNK;
        if (i == 0) next_chunk();
        if (i < SHARE) fetch_piece(fill, i);
#ifndef BF16_ABL_NOAREAD
        pre[nx % (LA + 1)] = ring[rd + i * 64];
#endif
        __builtin_amdgcn_sched_barrier(0);
        return __builtin_bit_cast(bf16x8, pre[g % (LA + 1)]);
    }
};

constexpr int input_group_base(int conv) {
    int g = 0;
    for (int i = 0; i < conv; ++i) g += kInputConvs[i].k * kInputConvs[i].k * kInputConvs[i].tiles;
    return g;
}
static_assert(input_group_base(kNumInputConvs) == kInputGroups, "");

__device__ __forceinline__ float pick(int hl, float lo, float hi) { return hl ? hi : lo; }

// bf16(max(a, 0)), bf16(max(b, 0)) packed low / high: rounding commutes with the clamp, and on
// the bf16 bit patterns the clamp is a signed 16-bit max with 0.
__device__ __forceinline__ unsigned pack_relu(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // one vector conversion = one v_cvt_pk_bf16_f32 (two scalar casts convert each half separately
    // and merge with v_perm; inline asm would sit behind an MFMA without the hazard wait states)
    const bf16x2 v = __builtin_convertvector(f32x2{a, b}, bf16x2);
    s16x2 s = __builtin_bit_cast(s16x2, v);
    s = __builtin_elementwise_max(s, s16x2{0, 0});
    return __builtin_bit_cast(unsigned, s);
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }

#ifdef BF_STAMPS
#define BF_STAMP(i) do { if (lane == 0) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BF_STAMP(i) ((void)0)
#endif
#define BF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x16 zero16() {
    return f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
}

// Compiler-only fence between a wave's LDS writes and its own later reads of the same bytes
// through another vector type (the hardware keeps a wave's LDS accesses in order).
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

// Stage 1, one input conv for both tiles.  reference: ann_creation_helper.py:211-236 (SAME padding:
// (k-1)*d in total, floor(/2) before).  mbc: tile 0's square of this lane in the byte mailbox
// (tile 1: 4 ranks up); tbl: the one-hot table at this lane's half.
template <int I, bool DBG>
__device__ __forceinline__ void input_conv(WStream& ws, const unsigned char* mbc, const u32x4* tbl,
                                           unsigned (&P)[2][kX1Dwords], int hl, int ln, float* dbg_board) {
    constexpr InputConv cv = kInputConvs[I];
    constexpr int G0 = input_group_base(I);
    constexpr int ph = ((cv.k - 1) * cv.dh) / 2, pw = ((cv.k - 1) * cv.dw) / 2;
    constexpr int TAPS = cv.k * cv.k;
    int code[2][TAPS];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dr = (tap / cv.k) * cv.dh - ph, df = (tap % cv.k) * cv.dw - pw;
            code[T][tap] = mbc[(4 * T + dr) * MB_W + df];
        }
    u32x4 b[2][TAPS];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) b[T][tap] = tbl[code[T][tap] * 2];
    f32x16 acc[2][cv.tiles];
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int tile = 0; tile < cv.tiles; ++tile) {
            const bf16x8 w = ws.take(G0 + tap * cv.tiles + tile);
#pragma unroll
            for (int T = 0; T < 2; ++T)
                acc[T][tile] = BF_MFMA16(w, __builtin_bit_cast(bf16x8, b[T][tap]), tap == 0 ? zero16() : acc[T][tile]);
        }
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int d = 0; d < kX1RegCount[I] / 2; ++d) {
            const int tile = (2 * d) / 16, t = (2 * d) % 16;
            const unsigned p = pack_relu(acc[T][tile][t], acc[T][tile][t + 1]);
            P[T][kX1RegBase[I] / 2 + d] = p;
            if (DBG) {
                const int c0 = 2 * (2 * d) + hl, c1 = 2 * (2 * d + 1) + hl;  // local channels of the pair
                if (c0 < cv.ch) dbg_board[(cv.chan_base + c0) * 64 + 32 * T + ln] = bf16_lo(p);
                if (c1 < cv.ch) dbg_board[(cv.chan_base + c1) * 64 + 32 * T + ln] = bf16_hi(p);
            }
        }
}

// ReLU + bf16 of one accumulator tile into this lane's position of a plane: registers 4g..4g+3
// are channels 8g + 4hl .. +3, i.e. 8 bytes at channel chunk g.
__device__ __forceinline__ void write_tile(unsigned char* wpos, int R0, int hl, const f32x16& a) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        u32x2 v;
        v.x = pack_relu(a[4 * g + 0], a[4 * g + 1]);
        v.y = pack_relu(a[4 * g + 2], a[4 * g + 3]);
        *reinterpret_cast<u32x2*>(wpos + (((g + R0) & 3) << 4) + hl * 8) = v;
    }
}

// Debug dump of a tile as the bf16 values the next layer sees.
__device__ __forceinline__ void dump_tile(float* dst_ch0, int hl, int sq, const f32x16& a) {
#pragma unroll
    for (int t = 0; t < 16; t += 2) {
        const unsigned p = pack_relu(a[t], a[t + 1]);
        dst_ch0[rowmap(t, hl) * 64 + sq] = bf16_lo(p);
        dst_ch0[rowmap(t + 1, hl) * 64 + sq] = bf16_hi(p);
    }
}

// One 3x3 SAME conv over a 32-channel chunk staged in a plane, both tiles: 9 taps x 2 k-steps.
// raddr[dr][s]: byte offset (in the plane) of tile 0's chunk 2s+hl one row and column back, dr rows down.
__device__ __forceinline__ void conv3x3_chunk(WStream& ws, int g0, const unsigned char* plane, const int (&raddr)[3][2],
                                              f32x16 (&acc)[2]) {
    auto load_b = [&](int idx, int T) -> u32x4 {
        const int tap = idx >> 1, s = idx & 1;
        return *reinterpret_cast<const u32x4*>(plane + raddr[tap / 3][s] + T * TILE1 + (tap % 3) * POS_BYTES);
    };
    u32x4 b[3][2];
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        b[0][T] = load_b(0, T);
        b[1][T] = load_b(1, T);
    }
#pragma unroll
    for (int idx = 0; idx < 18; ++idx) {
        if (idx + 2 < 18) {
            b[(idx + 2) % 3][0] = load_b(idx + 2, 0);
            b[(idx + 2) % 3][1] = load_b(idx + 2, 1);
        }
        const bf16x8 w = ws.take(g0 + idx);
        acc[0] = BF_MFMA16(w, __builtin_bit_cast(bf16x8, b[idx % 3][0]), acc[0]);
        acc[1] = BF_MFMA16(w, __builtin_bit_cast(bf16x8, b[idx % 3][1]), acc[1]);
    }
}

// MODE 0: scores; 1: scores + every intermediate activation (debug); 2: features -- x9 as float32
// [board][32][64] into `scores`, no head (the move-scoring network, see eval_f32.hip).
template <int MODE>
__global__ __launch_bounds__(THREADS, 1) void k_eval_bf16(const uint8_t* __restrict__ mailbox, unsigned n_boards,
                                                           const f32x4* __restrict__ wstream,
                                                           const float* __restrict__ bias,
                                                           const f32x4* __restrict__ headw,
                                                           float* __restrict__ scores, float* __restrict__ dbg) {
    constexpr bool DBG = MODE == 1;
    __shared__ __attribute__((aligned(16))) Lds lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63, hl = lane >> 5, ln = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rank = ln >> 3, file = ln & 7;                 // tile 0; tile 1 is 4 ranks up
    const unsigned board_raw = blockIdx.x * BPB + wave;
    const unsigned board = board_raw < n_boards ? board_raw : n_boards - 1;  // padding boards repeat the last one

#ifdef BF_STAMPS
    long long stamps[16];
#endif
    BF_STAMP(0);
    WStream ws;
    ws.prime(wstream, &lds.wring[0][0][0], lane, wave);

    // ---- byte mailbox (15 everywhere, then the board's codes) and the one-hot table ----------
    unsigned char* region = lds.board[wave];
    for (int i = lane; i < MB_WORDS; i += 64) reinterpret_cast<unsigned*>(region)[i] = 0x0F0F0F0Fu;
    if (tid < 32) {
        const unsigned c = tid >> 1, hh = tid & 1, j = c & 7u;
        const unsigned one = (c != 15u && (c >> 3) == hh) ? (0x3F80u << ((j & 1u) * 16u)) : 0u;
        u32x4 e = u32x4{0u, 0u, 0u, hh ? 0x3F800000u : 0u};  // plane 15 = 1.0 in every entry
        e.x |= ((j >> 1) == 0u) ? one : 0u;
        e.y |= ((j >> 1) == 1u) ? one : 0u;
        e.z |= ((j >> 1) == 2u) ? one : 0u;
        e.w |= ((j >> 1) == 3u) ? one : 0u;
        lds.onehot[c][hh] = e;
    }
    lds_order();
    {
        const unsigned v = mailbox[(size_t)board * kMailboxBytes + (lane >> 1)];
        const unsigned c = (lane & 1) ? (v >> 4) : (v & 15u);
        region[((lane >> 3) + 7) * MB_W + (lane & 7) + 7] = (unsigned char)c;
    }
    ws.start();  // its barrier also orders the table writes before stage 1 reads them

    float* dbg_board = DBG ? dbg + (size_t)board * kDebugFloatsPerBoard : nullptr;
    BF_STAMP(1);

    // ---- stage 1: input convs -> x1 as 60 packed dwords per tile -----------------------------
    unsigned P[2][kX1Dwords];
    {
        const unsigned char* mbc = region + (rank + 7) * MB_W + file + 7;
        const u32x4* tbl = &lds.onehot[0][hl];
        input_conv<0, DBG>(ws, mbc, tbl, P, hl, ln, dbg_board);
        input_conv<1, DBG>(ws, mbc, tbl, P, hl, ln, dbg_board);
        input_conv<2, DBG>(ws, mbc, tbl, P, hl, ln, dbg_board);
        input_conv<3, DBG>(ws, mbc, tbl, P, hl, ln, dbg_board);
        input_conv<4, DBG>(ws, mbc, tbl, P, hl, ln, dbg_board);
        input_conv<5, DBG>(ws, mbc, tbl, P, hl, ln, dbg_board);
        input_conv<6, DBG>(ws, mbc, tbl, P, hl, ln, dbg_board);
        input_conv<7, DBG>(ws, mbc, tbl, P, hl, ln, dbg_board);
        input_conv<8, DBG>(ws, mbc, tbl, P, hl, ln, dbg_board);
    }
    // The byte mailbox is dead (only this wave reads it): its space becomes the two zero-haloed planes.
    lds_order();
    for (int i = lane; i < 2 * PLANE_BYTES / 16; i += 64) reinterpret_cast<u32x4*>(region)[i] = u32x4{0u, 0u, 0u, 0u};

    BF_STAMP(2);
    // ---- stage 2: 1x1 236->512 (+ReLU) chained into 1x1 512->128 ---------------------------
    f32x16 acc3[2][4];
#pragma unroll
    for (int c3 = 0; c3 < 4; ++c3)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            acc3[0][c3][t] = pick(hl, bias[kBiasX3 + 32 * c3 + rowmap(t, 0)], bias[kBiasX3 + 32 * c3 + rowmap(t, 1)]);
            acc3[1][c3][t] = acc3[0][c3][t];
        }

#pragma unroll 1
    for (int it = 0; it < kC2Tiles / C2_UNROLL; ++it) {
#pragma unroll
        for (int u = 0; u < C2_UNROLL; ++u) {
            const int g0 = kGemmBase + u * kPerC2Groups;
            // The two tiles run half a phase apart: tile 0 uses each fragment as it arrives and
            // keeps it in registers for tile 1, so that one tile's ReLU + pack (which has to wait
            // for its last MFMA) rides in the MFMA gaps of the other tile instead of idling the pipe.
            f32x16 acc2[2];
            bf16x8 f[kGemm1Groups];
            if (it * C2_UNROLL + u == 2) BF_STAMP(8);
#pragma unroll
            for (int s = 0; s < kGemm1Groups; ++s) {  // phase A: 236->512, tile 0
                f[s] = ws.take(g0 + s);
                const u32x4 b = u32x4{P[0][4 * s], P[0][4 * s + 1], P[0][4 * s + 2], P[0][4 * s + 3]};
                acc2[0] = BF_MFMA16(f[s], __builtin_bit_cast(bf16x8, b), s == 0 ? zero16() : acc2[0]);
            }
            if (it * C2_UNROLL + u == 2) BF_STAMP(9);
            unsigned H[2][8];
#pragma unroll
            for (int s = 0; s < kGemm1Groups; ++s) {  // phase B: 236->512, tile 1 | pack tile 0
                const u32x4 b = u32x4{P[1][4 * s], P[1][4 * s + 1], P[1][4 * s + 2], P[1][4 * s + 3]};
                acc2[1] = BF_MFMA16(f[s], __builtin_bit_cast(bf16x8, b), s == 0 ? zero16() : acc2[1]);
                if (s >= 3 && s < 11) H[0][s - 3] = pack_relu(acc2[0][2 * (s - 3)], acc2[0][2 * (s - 3) + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (it * C2_UNROLL + u == 2) BF_STAMP(10);
            bf16x8 h[kGemm2Groups];
#pragma unroll
            for (int i = 0; i < kGemm2Groups; ++i) {  // phase C: 512->128, tile 0 | pack tile 1
                const int s = i >> 2, c3 = i & 3;
                h[i] = ws.take(g0 + kGemm1Groups + i);
                const u32x4 b = u32x4{H[0][4 * s], H[0][4 * s + 1], H[0][4 * s + 2], H[0][4 * s + 3]};
                acc3[0][c3] = BF_MFMA16(h[i], __builtin_bit_cast(bf16x8, b), acc3[0][c3]);
                if (i >= 1) H[1][i - 1] = pack_relu(acc2[1][2 * (i - 1)], acc2[1][2 * (i - 1) + 1]);
                if (i == 7) H[1][7] = pack_relu(acc2[1][14], acc2[1][15]);
            }
            (void)ws.take(g0 + kPerC2Groups - 1);  // padding group
            if (it * C2_UNROLL + u == 2) BF_STAMP(11);
#pragma unroll
            for (int i = 0; i < kGemm2Groups; ++i) {  // phase D: 512->128, tile 1
                const int s = i >> 2, c3 = i & 3;
                const u32x4 b = u32x4{H[1][4 * s], H[1][4 * s + 1], H[1][4 * s + 2], H[1][4 * s + 3]};
                acc3[1][c3] = BF_MFMA16(h[i], __builtin_bit_cast(bf16x8, b), acc3[1][c3]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (it * C2_UNROLL + u == 2) BF_STAMP(12);
            if (it * C2_UNROLL + u == 3) BF_STAMP(13);
        }
    }
    if (DBG) {
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int c3 = 0; c3 < 4; ++c3) dump_tile(dbg_board + 64 * (236 + 32 * c3), hl, 32 * T + ln, acc3[T][c3]);
    }

    BF_STAMP(3);
    // ---- stages 3 and 4: the six 3x3 layers through the wave's own planes ------------------
    const int R0 = rank + 1;                                // padded row of tile 0's square (tile 1: +4, same & 3)
    const int woff = (R0 * XP_W + file + 1) * POS_BYTES;    // tile 0's square; tile 1 at + TILE1
    int raddr[3][2];
#pragma unroll
    for (int dr = 0; dr < 3; ++dr)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            raddr[dr][s] = ((rank + dr) * XP_W + file) * POS_BYTES + (((2 * s + hl + rank + dr) & 3) << 4);

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        acc[0][t] = pick(hl, bias[kBiasX4 + rowmap(t, 0)], bias[kBiasX4 + rowmap(t, 1)]);
        acc[1][t] = acc[0][t];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned char* plane = region + (q & 1) * PLANE_BYTES;
        lds_order();
        write_tile(plane + woff, R0, hl, acc3[0][q]);
        write_tile(plane + woff + TILE1, R0, hl, acc3[1][q]);
        lds_order();
        conv3x3_chunk(ws, kConvBase + q * 18, plane, raddr, acc);
    }
    BF_STAMP(4);
#pragma unroll
    for (int layer = 0; layer < 5; ++layer) {
        unsigned char* plane = region + (layer & 1) * PLANE_BYTES;
        lds_order();
        write_tile(plane + woff, R0, hl, acc[0]);
        write_tile(plane + woff + TILE1, R0, hl, acc[1]);
        lds_order();
        if (DBG) {
            dump_tile(dbg_board + 64 * (236 + 128 + 32 * layer), hl, ln, acc[0]);
            dump_tile(dbg_board + 64 * (236 + 128 + 32 * layer), hl, 32 + ln, acc[1]);
        }
        const float* bl = bias + kBiasX4 + 32 * (layer + 1);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            acc[0][t] = pick(hl, bl[rowmap(t, 0)], bl[rowmap(t, 1)]);
            acc[1][t] = acc[0][t];
        }
        conv3x3_chunk(ws, kConvBase + kConv4Groups + layer * 18, plane, raddr, acc);
    }

    BF_STAMP(5);
    // ---- stage 5: 8x8 valid conv (2048 -> 16) across the workgroup's boards, ReLU, logit ------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's look-ahead DMA must not outlive the workgroup
    if (MODE == 2) {
        if (board_raw < n_boards) {
            float* out = scores + (size_t)board * 2048;
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int t = 0; t < 16; ++t) out[rowmap(t, hl) * 64 + 32 * T + ln] = fmaxf(acc[T][t], 0.0f);
        }
        return;
    }
    lds_order();
    write_tile(region + PLANE_BYTES + woff, R0, hl, acc[0]);   // x9 -> plane 1
    write_tile(region + PLANE_BYTES + woff + TILE1, R0, hl, acc[1]);
    if (DBG) {
        dump_tile(dbg_board + 64 * (236 + 128 + 32 * 5), hl, ln, acc[0]);
        dump_tile(dbg_board + 64 * (236 + 128 + 32 * 5), hl, 32 + ln, acc[1]);
    }
    constexpr int SQW = 64 / WAVES;  // squares (k-steps) per wave
    const int col = lane & 15, kc = lane >> 4;
    u32x4 ha[SQW];
#pragma unroll
    for (int i = 0; i < SQW; ++i)
        ha[i] = __builtin_bit_cast(u32x4, headw[(size_t)(wave * SQW + i) * 64 + lane]);
    __syncthreads();
    {
        const unsigned char* xb = lds.board[col < BPB ? col : 0] + PLANE_BYTES;
        f32x4 hacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < SQW; ++i) {
            const int sqk = wave * SQW + i;
            const int Rk = (sqk >> 3) + 1;
            const u32x4 b = *reinterpret_cast<const u32x4*>(xb + (Rk * XP_W + (sqk & 7) + 1) * POS_BYTES + (((kc + Rk) & 3) << 4));
            hacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ha[i]), __builtin_bit_cast(bf16x8, b),
                                                           hacc, 0, 0, 0);
        }
        // partial sums go to plane 0 of this wave's board (nobody else reads it)
        reinterpret_cast<f32x4*>(region)[lane] = hacc;
    }
    __syncthreads();
    if (tid < 16 * BPB) {
        const int b = tid >> 4, co = tid & 15;
        // D[row = co][col = b] sits in lane b + 16*(co>>2), register co&3
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w)
            s += reinterpret_cast<const float*>(lds.board[w])[(b + 16 * (co >> 2)) * 4 + (co & 3)];
        const float y = fmaxf(s + bias[kBiasX10 + co], 0.0f);
        const unsigned out_raw = blockIdx.x * BPB + b;
        if (DBG && out_raw < n_boards) dbg[(size_t)out_raw * kDebugFloatsPerBoard + 64 * (236 + 128 + 32 * 6) + co] = y;
        float v = y * bias[kLogitW + co];  // logit_layer: no bias, no batch-norm (evaluation_ann.py:35-44)
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        if (co == 0 && out_raw < n_boards) scores[out_raw] = v;
    }
#ifdef BF_STAMPS
    BF_STAMP(6);
    if (!DBG && dbg && lane == 0) {
        long long* out = reinterpret_cast<long long*>(dbg) + ((size_t)blockIdx.x * WAVES + wave) * 16;
        for (int i = 0; i < 7; ++i) out[i] = stamps[i];
        for (int i = 8; i < 14; ++i) out[i] = stamps[i];
    }
#endif
}

}  // namespace

hipError_t launch(const uint8_t* d_mailbox, size_t n, const unsigned char* d_image, float* d_scores, float* d_debug,
                  hipStream_t stream);

}  // namespace b16

hipError_t launch_eval_bf16(const uint8_t* d_mailbox, size_t n, const unsigned char* d_image, float* d_scores,
                            float* d_debug, hipStream_t stream) {
    return b16::launch(d_mailbox, n, d_image, d_scores, d_debug, stream);
}

hipError_t launch_policy_features_bf16(const uint8_t* d_mailbox, size_t n, const unsigned char* d_image, float* d_x9,
                                       hipStream_t stream) {
    using namespace b16;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_eval_bf16<2>, dim3((unsigned)((n + BPB - 1) / BPB)), dim3(THREADS), 0, stream, d_mailbox,
                       (unsigned)n, reinterpret_cast<const f32x4*>(d_image + b16::kOffStream),
                       reinterpret_cast<const float*>(d_image + b16::kOffBias),
                       reinterpret_cast<const f32x4*>(d_image + b16::kOffHead), d_x9, nullptr);
    return hipGetLastError();
}

hipError_t b16::launch(const uint8_t* d_mailbox, size_t n, const unsigned char* d_image, float* d_scores,
                            float* d_debug, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const f32x4* ws = reinterpret_cast<const f32x4*>(d_image + kOffStream);
    const float* bias = reinterpret_cast<const float*>(d_image + kOffBias);
    const f32x4* head = reinterpret_cast<const f32x4*>(d_image + kOffHead);
    const unsigned blocks = (unsigned)((n + BPB - 1) / BPB);
#ifdef BF_STAMPS
    if (false)
#else
    if (d_debug)
#endif
        hipLaunchKernelGGL(k_eval_bf16<1>, dim3(blocks), dim3(THREADS), 0, stream, d_mailbox, (unsigned)n, ws, bias,
                           head, d_scores, d_debug);
    else
        hipLaunchKernelGGL(k_eval_bf16<0>, dim3(blocks), dim3(THREADS), 0, stream, d_mailbox, (unsigned)n, ws, bias,
                           head, d_scores, d_debug);
    return hipGetLastError();
}

}  // namespace bf
