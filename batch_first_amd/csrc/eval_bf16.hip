// The value network as one bf16 MFMA kernel (BASELINE configs[2]): 4-bit mailboxes in, one score
// per board out.  Arithmetic and weight image: bf_layout_bf16.h (bf16 weights and activations,
// float32 sums and shifts).
//
// Reference graph replaced (TensorFlow ops -> this kernel):
//   create_input_convolutions_shared_weights   ann_creation_helper.py:192-252  (stage 1)
//   build_convolutional_modules, spec at       evaluation_ann.py:268           (stages 2-4)
//   logit_layer + squeeze                      evaluation_ann.py:35-46         (stage 5)
//
// Same shape as the fp32 kernel (eval_f32.hip): one board = two waves, wave `half` owns squares
// 32*half .. 32*half+31 as the 32 columns of every v_mfma_f32_32x32x16_bf16 tile, layers run
// transposed so that a finished accumulator tile, converted to bf16 in place, is the B operand of
// the next 1x1 layer (k order permuted on the host to match the accumulator's row map).
//   stage 1  B operand of a tap = the shifted square's one-hot over 16 planes: a code byte from a
//            padded mailbox in LDS, then one ds_read_b128 of that code's row of a 16-entry table.
//            The 16th plane is 1.0 everywhere and carries the folded batch-norm shift.
//   stage 2  per 32-channel hidden tile: 15 MFMAs (236->512), ReLU + pack, 8 MFMAs (512->128).
//   stage 3/4  3x3 layers through zero-haloed [10x10 positions][32 channels] bf16 planes in LDS;
//            the 16-byte channel chunk c of a position in padded row R sits at chunk (c + R) & 3,
//            which makes the shifted ds_read_b128 of every tap conflict-free.
//   stage 5  8x8 valid conv across the boards of the workgroup: out[co][board] on
//            v_mfma_f32_16x16x32_bf16, one k-step per square, squares split over the waves, fixed-
//            order reduction of the per-wave partial sums; ReLU; logit.
// Weights stream through the same lockstep LDS ring as in the fp32 kernel (global_load_lds into a
// 3-slot ring, one barrier per chunk), here with one 1-KiB group per MFMA and a 3-group register
// look-ahead.  Results are bit-reproducible and independent of a board's position in the batch.
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
#define BF16_BPB 2
#endif
constexpr int BPB = BF16_BPB;            // boards per workgroup
constexpr int WAVES = 2 * BPB;
constexpr int THREADS = 64 * WAVES;
#ifndef BF16_CHUNK
#define BF16_CHUNK ((BF16_BPB >= 4) ? 16 : 8)
#endif
constexpr int CHUNK = BF16_CHUNK;           // groups per lockstep chunk
constexpr int SHARE = CHUNK / WAVES;        // groups each wave fetches per chunk
#ifndef BF16_SLOTS
#define BF16_SLOTS 3
#endif
constexpr int SLOTS = BF16_SLOTS;           // ring slots: one being read, SLOTS-1 landed or in flight
#ifndef BF16_LA
#define BF16_LA 3
#endif
constexpr int LA = BF16_LA;                       // groups held in registers ahead of the MFMA
constexpr int C2_UNROLL = 2;
static_assert(CHUNK <= kStreamChunkMax, "");
static_assert(CHUNK % WAVES == 0, "every wave fetches whole groups");
static_assert(kInputGroups % CHUNK == 0 && kInputGroups % (LA + 1) == 0, "the tile loop starts on a chunk boundary");
static_assert((kPerC2Groups * C2_UNROLL) % CHUNK == 0 && (kPerC2Groups * C2_UNROLL) % (LA + 1) == 0,
              "the loop body is whole chunks and keeps the look-ahead phase");
static_assert(kC2Tiles % C2_UNROLL == 0, "");
static_assert(LA <= CHUNK, "");

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
static_assert(MB_WORDS * 4 <= PLANE_BYTES && 2 * 1024 <= PLANE_BYTES, "mailbox and head partial sums alias plane 0");
#ifndef BF16_WPS
#define BF16_WPS 2
#endif
constexpr int WPS = BF16_WPS;               // waves per SIMD the register budget is set for
static_assert(BPB <= 16, "the head puts boards on the 16 columns of one MFMA");

struct Lds {
    f32x4 wring[SLOTS][CHUNK][64];                       // shared A-fragment ring
    __attribute__((aligned(16))) unsigned char board[BPB][BOARD_BYTES];
    u32x4 onehot[16][2];                                 // stage 1: [code][lane half] -> 8 bf16 planes
};

// Workgroup barrier for LDS hand-offs: every LDS access of this wave has completed; global
// memory and the LDS-DMA queue are left alone (__syncthreads() would drain them: vmcnt(0)).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Chunk barrier of the weight ring: additionally this wave's share of the chunk about to be read
// has landed.  The wave's LDS-DMA queue holds, oldest first, its shares of that chunk and of the
// SLOTS-2 chunks after it; vector-memory operations retire in order, so leaving (SLOTS-2)*SHARE of
// them outstanding is enough (younger ordinary loads only make the wait stricter).
__device__ __forceinline__ void ring_barrier() {
    constexpr int N = (SLOTS - 2) * SHARE;
    static_assert(N < 64, "vmcnt is a 6-bit counter");
#ifdef BF16_ABL_NOBAR
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

// Lockstep walk over the A-fragment stream through the LDS ring (see eval_f32.hip).
struct WStream {
    const f32x4* src;
    f32x4* ring;
    f32x4* share;
    int rd;                // float4 offset of the slot the look-ahead reads from
    f32x4 pre[LA + 1];

    // LDS-DMA: SHARE x 1 KiB straight from L2 into ring slot `slot_off`, lane-linear.  Issued as
    // inline asm on purpose: the compiler models global_load_lds as a FLAT access that may touch
    // LDS, and while one is outstanding it turns every LDS wait into lgkmcnt(0), which stalls each
    // group of MFMAs on the read just issued.  ring_barrier() does the vmcnt bookkeeping instead;
    // ordinary loads interleaved by the compiler only make its own vmcnt waits stricter
    // (vector-memory operations retire in order).
    __device__ __forceinline__ void fetch_into(int slot_off) {
        typedef __attribute__((address_space(3))) f32x4* lds_ptr;
        const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)(share + slot_off);
#ifndef BF16_ABL_NODMA
#pragma unroll
        for (int i = 0; i < SHARE; ++i)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                         :: "s"(lds0 + i * 1024), "v"(src + i * 64) : "memory", "m0");
#endif
        src += CHUNK * 64;
    }
    __device__ __forceinline__ void prime(const f32x4* stream, f32x4* ring_base, int lane, int wave) {
        ring = ring_base + lane;
        share = ring_base + wave * SHARE * 64;
        src = stream + wave * SHARE * 64 + lane;
#pragma unroll
        for (int i = 0; i < SLOTS - 1; ++i) fetch_into(i * CHUNK * 64);
        rd = (SLOTS - 1) * CHUNK * 64;
    }
    // After the barrier every wave's share of chunk k is in LDS and nobody reads chunk k-1 any
    // more (its last group went to registers LA groups ago), so its slot takes chunk k+SLOTS-1.
    __device__ __forceinline__ void next_chunk() {
        const int free_slot = rd;
        rd = (rd == (SLOTS - 1) * CHUNK * 64) ? 0 : rd + CHUNK * 64;
        __builtin_amdgcn_sched_barrier(0);
        ring_barrier();
        __builtin_amdgcn_sched_barrier(0);
        fetch_into(free_slot);
    }
    __device__ __forceinline__ void start() {
        next_chunk();
#pragma unroll
        for (int i = 0; i < LA; ++i) pre[i] = ring[rd + i * 64];
    }
    // g: stream group index, a compile-time constant after unrolling; groups are taken in order.
    __device__ __forceinline__ bf16x8 take(int g) {
        const int nx = g + LA;
        if (nx % CHUNK == 0) next_chunk();
#ifndef BF16_ABL_NOAREAD
        pre[nx % (LA + 1)] = ring[rd + (nx % CHUNK) * 64];
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

// Stage 1, one input conv.  reference: ann_creation_helper.py:211-236 (SAME padding:
// (k-1)*d in total, floor(/2) before).  mbc: this lane's square in the byte mailbox; tbl: the one-hot table at this lane's half.
template <int I, bool DBG>
__device__ __forceinline__ void input_conv(WStream& ws, const unsigned char* mbc, const u32x4* tbl,
                                           unsigned (&P)[kX1Dwords], int hl, int sq, float* dbg_board) {
    constexpr InputConv cv = kInputConvs[I];
    constexpr int G0 = input_group_base(I);
    constexpr int ph = ((cv.k - 1) * cv.dh) / 2, pw = ((cv.k - 1) * cv.dw) / 2;
    int code[cv.k * cv.k];
#pragma unroll
    for (int tap = 0; tap < cv.k * cv.k; ++tap) {
        const int dr = (tap / cv.k) * cv.dh - ph, df = (tap % cv.k) * cv.dw - pw;
        code[tap] = mbc[dr * MB_W + df];
    }
    u32x4 b[cv.k * cv.k];
#pragma unroll
    for (int tap = 0; tap < cv.k * cv.k; ++tap) b[tap] = tbl[code[tap] * 2];
    f32x16 acc[cv.tiles];
#pragma unroll
    for (int tap = 0; tap < cv.k * cv.k; ++tap)
#pragma unroll
        for (int tile = 0; tile < cv.tiles; ++tile) {
            const bf16x8 w = ws.take(G0 + tap * cv.tiles + tile);
            acc[tile] = BF_MFMA16(w, __builtin_bit_cast(bf16x8, b[tap]), tap == 0 ? zero16() : acc[tile]);
        }
#pragma unroll
    for (int d = 0; d < kX1RegCount[I] / 2; ++d) {
        const int tile = (2 * d) / 16, t = (2 * d) % 16;
        const unsigned p = pack_relu(acc[tile][t], acc[tile][t + 1]);
        P[kX1RegBase[I] / 2 + d] = p;
        if (DBG) {
            const int c0 = 2 * (2 * d) + hl, c1 = 2 * (2 * d + 1) + hl;  // local channels of the pair
            if (c0 < cv.ch) dbg_board[(cv.chan_base + c0) * 64 + sq] = bf16_lo(p);
            if (c1 < cv.ch) dbg_board[(cv.chan_base + c1) * 64 + sq] = bf16_hi(p);
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

// One 3x3 SAME conv over a 32-channel chunk staged in a plane: 9 taps x 2 k-steps.
// raddr[dr][s]: byte offset (in the plane) of this lane's chunk 2s+hl one row and column back, dr rows down.
__device__ __forceinline__ void conv3x3_chunk(WStream& ws, int g0, const unsigned char* plane, const int (&raddr)[3][2],
                                              f32x16& acc) {
    auto load_b = [&](int idx) -> u32x4 {
        const int tap = idx >> 1, s = idx & 1;
        return *reinterpret_cast<const u32x4*>(plane + raddr[tap / 3][s] + (tap % 3) * POS_BYTES);
    };
    u32x4 b[3];
    b[0] = load_b(0);
    b[1] = load_b(1);
#pragma unroll
    for (int idx = 0; idx < 18; ++idx) {
        if (idx + 2 < 18) b[(idx + 2) % 3] = load_b(idx + 2);
        const bf16x8 w = ws.take(g0 + idx);
        acc = BF_MFMA16(w, __builtin_bit_cast(bf16x8, b[idx % 3]), acc);
    }
}

template <bool DBG>
__global__ __launch_bounds__(THREADS, WPS) void k_eval_bf16(const uint8_t* __restrict__ mailbox, unsigned n_boards,
                                                           const f32x4* __restrict__ wstream,
                                                           const float* __restrict__ bias,
                                                           const f32x4* __restrict__ headw,
                                                           float* __restrict__ scores, float* __restrict__ dbg) {
    __shared__ __attribute__((aligned(16))) Lds lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63, hl = lane >> 5, ln = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bib = wave >> 1, half = wave & 1;
    const int btid = tid & 127;
    const int sq = 32 * half + ln, rank = sq >> 3, file = sq & 7;
    const unsigned board_raw = blockIdx.x * BPB + bib;
    const unsigned board = board_raw < n_boards ? board_raw : n_boards - 1;  // padding boards repeat the last one

#ifdef BF_STAMPS
    long long stamps[8];
#endif
    BF_STAMP(0);
    WStream ws;
    ws.prime(wstream, &lds.wring[0][0][0], lane, wave);

    // ---- byte mailbox (15 everywhere, then the board's codes) and the one-hot table ----------
    unsigned char* region = lds.board[bib];
    u32x4* zero_area = reinterpret_cast<u32x4*>(region);
    if (btid < MB_WORDS) reinterpret_cast<unsigned*>(region)[btid] = 0x0F0F0F0Fu;
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
    __syncthreads();
    if (btid < 64) {
        const unsigned v = mailbox[(size_t)board * kMailboxBytes + (btid >> 1)];
        const unsigned c = (btid & 1) ? (v >> 4) : (v & 15u);
        region[((btid >> 3) + 7) * MB_W + (btid & 7) + 7] = (unsigned char)c;
    }
    ws.start();  // its barrier also orders the table writes before stage 1 reads them

    float* dbg_board = DBG ? dbg + (size_t)board * kDebugFloatsPerBoard : nullptr;
    BF_STAMP(1);

    // ---- stage 1: input convs -> x1 as 60 packed dwords -------------------------------------
    unsigned P[kX1Dwords];
    {
        const unsigned char* mbc = region + (rank + 7) * MB_W + file + 7;
        const u32x4* tbl = &lds.onehot[0][hl];
        input_conv<0, DBG>(ws, mbc, tbl, P, hl, sq, dbg_board);
        input_conv<1, DBG>(ws, mbc, tbl, P, hl, sq, dbg_board);
        input_conv<2, DBG>(ws, mbc, tbl, P, hl, sq, dbg_board);
        input_conv<3, DBG>(ws, mbc, tbl, P, hl, sq, dbg_board);
        input_conv<4, DBG>(ws, mbc, tbl, P, hl, sq, dbg_board);
        input_conv<5, DBG>(ws, mbc, tbl, P, hl, sq, dbg_board);
        input_conv<6, DBG>(ws, mbc, tbl, P, hl, sq, dbg_board);
        input_conv<7, DBG>(ws, mbc, tbl, P, hl, sq, dbg_board);
        input_conv<8, DBG>(ws, mbc, tbl, P, hl, sq, dbg_board);
    }

    BF_STAMP(2);
    // ---- stage 2: 1x1 236->512 (+ReLU) chained into 1x1 512->128 ---------------------------
    f32x16 acc3[4];
#pragma unroll
    for (int c3 = 0; c3 < 4; ++c3)
#pragma unroll
        for (int t = 0; t < 16; ++t)
            acc3[c3][t] = pick(hl, bias[kBiasX3 + 32 * c3 + rowmap(t, 0)], bias[kBiasX3 + 32 * c3 + rowmap(t, 1)]);

#pragma unroll 1
    for (int it = 0; it < kC2Tiles / C2_UNROLL; ++it) {
        if (it == 1) {
            // Every wave is past stage 1 (at least one chunk barrier lies behind us): the byte
            // mailbox is dead, its space becomes the two zero-haloed activation planes.
            for (int i = btid; i < 2 * PLANE_BYTES / 16; i += 128) zero_area[i] = u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < C2_UNROLL; ++u) {
            const int g0 = kGemmBase + u * kPerC2Groups;
            f32x16 acc2;
#pragma unroll
            for (int s = 0; s < kGemm1Groups; ++s) {
                const bf16x8 w = ws.take(g0 + s);
                const u32x4 b = u32x4{P[4 * s], P[4 * s + 1], P[4 * s + 2], P[4 * s + 3]};
                acc2 = BF_MFMA16(w, __builtin_bit_cast(bf16x8, b), s == 0 ? zero16() : acc2);
            }
            unsigned H[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) H[i] = pack_relu(acc2[2 * i], acc2[2 * i + 1]);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int c3 = 0; c3 < 4; ++c3) {
                    const bf16x8 w = ws.take(g0 + kGemm1Groups + s * 4 + c3);
                    const u32x4 b = u32x4{H[4 * s], H[4 * s + 1], H[4 * s + 2], H[4 * s + 3]};
                    acc3[c3] = BF_MFMA16(w, __builtin_bit_cast(bf16x8, b), acc3[c3]);
                }
            (void)ws.take(g0 + kPerC2Groups - 1);  // padding group
        }
    }
    if (DBG) {
#pragma unroll
        for (int c3 = 0; c3 < 4; ++c3)
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                const unsigned p = pack_relu(acc3[c3][t], acc3[c3][t + 1]);
                dbg_board[64 * 236 + (32 * c3 + rowmap(t, hl)) * 64 + sq] = bf16_lo(p);
                dbg_board[64 * 236 + (32 * c3 + rowmap(t + 1, hl)) * 64 + sq] = bf16_hi(p);
            }
    }

    BF_STAMP(3);
    // ---- stages 3 and 4: the six 3x3 layers through LDS -----------------------------------
    const int R0 = rank + 1;
    const int woff = (R0 * XP_W + file + 1) * POS_BYTES;   // this lane's square
    int raddr[3][2];
#pragma unroll
    for (int dr = 0; dr < 3; ++dr)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            raddr[dr][s] = ((rank + dr) * XP_W + file) * POS_BYTES + (((2 * s + hl + rank + dr) & 3) << 4);

    f32x16 acc;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = pick(hl, bias[kBiasX4 + rowmap(t, 0)], bias[kBiasX4 + rowmap(t, 1)]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned char* plane = region + (q & 1) * PLANE_BYTES;
        write_tile(plane + woff, R0, hl, acc3[q]);
        lds_barrier();
        conv3x3_chunk(ws, kConvBase + q * 18, plane, raddr, acc);
    }
    BF_STAMP(4);
#pragma unroll
    for (int layer = 0; layer < 5; ++layer) {
        unsigned char* plane = region + (layer & 1) * PLANE_BYTES;
        write_tile(plane + woff, R0, hl, acc);
        if (DBG) {
#pragma unroll
            for (int t = 0; t < 16; t += 2) {
                const unsigned p = pack_relu(acc[t], acc[t + 1]);
                dbg_board[64 * (236 + 128 + 32 * layer) + rowmap(t, hl) * 64 + sq] = bf16_lo(p);
                dbg_board[64 * (236 + 128 + 32 * layer) + rowmap(t + 1, hl) * 64 + sq] = bf16_hi(p);
            }
        }
        lds_barrier();
        const float* bl = bias + kBiasX4 + 32 * (layer + 1);
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = pick(hl, bl[rowmap(t, 0)], bl[rowmap(t, 1)]);
        conv3x3_chunk(ws, kConvBase + kConv4Groups + layer * 18, plane, raddr, acc);
    }

    BF_STAMP(5);
    // ---- stage 5: 8x8 valid conv (2048 -> 16) across the workgroup's boards, ReLU, logit ------
    // x9 goes to plane 1 (layer 4 read plane 0; plane 1 was last read one barrier earlier).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's look-ahead DMA must not outlive the workgroup
    write_tile(region + PLANE_BYTES + woff, R0, hl, acc);
    if (DBG) {
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
            const unsigned p = pack_relu(acc[t], acc[t + 1]);
            dbg_board[64 * (236 + 128 + 32 * 5) + rowmap(t, hl) * 64 + sq] = bf16_lo(p);
            dbg_board[64 * (236 + 128 + 32 * 5) + rowmap(t + 1, hl) * 64 + sq] = bf16_hi(p);
        }
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
        // partial sums go to plane 0 of this wave's board (last read before the barrier above)
        reinterpret_cast<f32x4*>(region + half * 1024)[lane] = hacc;
    }
    __syncthreads();
    if (tid < 16 * BPB) {
        const int b = tid >> 4, co = tid & 15;
        // D[row = co][col = b] sits in lane b + 16*(co>>2), register co&3
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w)
            s += reinterpret_cast<const float*>(lds.board[w >> 1] + (w & 1) * 1024)[(b + 16 * (co >> 2)) * 4 + (co & 3)];
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
        long long* out = reinterpret_cast<long long*>(dbg) + ((size_t)blockIdx.x * WAVES + wave) * 8;
        for (int i = 0; i < 7; ++i) out[i] = stamps[i];
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
        hipLaunchKernelGGL(k_eval_bf16<true>, dim3(blocks), dim3(THREADS), 0, stream, d_mailbox, (unsigned)n, ws, bias,
                           head, d_scores, d_debug);
    else
        hipLaunchKernelGGL(k_eval_bf16<false>, dim3(blocks), dim3(THREADS), 0, stream, d_mailbox, (unsigned)n, ws, bias,
                           head, d_scores, d_debug);
    return hipGetLastError();
}

}  // namespace bf
