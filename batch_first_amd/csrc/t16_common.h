// Pieces shared by the two 16x16x32 bf16 kernels (eval_t16.hip: one wave = one board; eval_t16s.hip: one
// workgroup = one board, for small calls): vector types, the lockstep walk over the A-fragment stream through
// the LDS ring, ReLU + bf16 packing.
#pragma once
#include <hip/hip_runtime.h>

#include "bf_device.h"
#include "bf_layout_t16.h"

namespace bf {
namespace t16 {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef T16_WAVES
#define T16_WAVES 4
#endif
constexpr int BPB = 4;                   // boards per workgroup
constexpr int WAVES = T16_WAVES;         // eval_t16.hip: one wave a board; eval_t16h.hip: two (eight waves a workgroup)
constexpr int THREADS = 64 * WAVES;
#ifndef T16_CHUNK
#define T16_CHUNK 16
#endif
#ifndef T16_SLOTS
#define T16_SLOTS 4
#endif
constexpr int CHUNK = T16_CHUNK;         // groups per lockstep chunk of the weight ring
constexpr int SHARE = CHUNK / WAVES;     // groups each wave fetches per chunk
constexpr int SLOTS = T16_SLOTS;         // ring slots: one being read, one just read, SLOTS - 2 landed or in flight
#ifndef T16_LA
#define T16_LA 3
#endif
constexpr int LA = T16_LA;               // groups held in registers ahead of the MFMAs
constexpr int AHEAD = SLOTS - 2;         // chunks the refill runs ahead of the chunk being read
constexpr int WALK_CHUNKS = kStreamWalkGroups / CHUNK;
static_assert(kStreamWalkGroups % CHUNK == 0 && WALK_CHUNKS % SLOTS == 0 && kStreamWalkGroups % (LA + 1) == 0,
              "every board starts in the same phase of the ring and of the register look-ahead");
static_assert(SHARE >= 1 && SHARE <= 4 && CHUNK % WAVES == 0 && LA <= CHUNK, "fetch_piece: up to four pieces with immediate offsets");

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Chunk barrier of the weight ring: this wave's share of the chunk about to be read has landed.  The wave's
// LDS-DMA queue holds, oldest first, its shares of that chunk and of the AHEAD - 1 chunks after it; vector-memory
// operations retire in order, so leaving (AHEAD - 1) * SHARE of them outstanding is enough (younger ordinary
// loads only make the wait stricter).  No LDS wait: the slot refilled after the barrier is that of the chunk
// BEFORE the one just finished, whose reads were issued more than CHUNK groups ago and have returned (LDS
// returns a wave's reads in order, and every group waits for its own fragment, issued LA groups earlier).
// EXTRA = 1 for the AHEAD barriers after the next board's mailbox prefetch (one more LDS-DMA, issued right
// after a chunk's pieces): it is younger than the pieces those barriers need and may stay in flight.
template <int EXTRA>
__device__ __forceinline__ void ring_barrier() {
    constexpr int N = (AHEAD - 1) * SHARE + EXTRA;
#ifdef T16_ABL_NOBAR   // timing ablation: waves do not wait for each other (wrong scores)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

// The next board's mailbox is prefetched after the take of this group (the take that issues the last piece
// of its chunk), early in stage 2.
constexpr int kPrefetchGroup = ((kPairBase + 24 + LA + CHUNK - 1) / CHUNK) * CHUNK + SHARE - 1 - LA;
static_assert((kPrefetchGroup + LA) % CHUNK == SHARE - 1 && kPrefetchGroup >= kPairBase &&
              kPrefetchGroup + 3 * CHUNK < kStreamGroups, "");

// Lockstep walk over the A-fragment stream through the LDS ring (DESIGN.md 3.2 gives the measurements behind
// each choice).  LDS-DMA: one 1-KiB piece of this wave's share straight from L2 into a ring slot, lane-linear,
// ONE instruction per piece: global address = scalar base + lane * 16 + immediate, LDS address = M0 + immediate
// + lane * 16 (the immediate applies to both sides), so M0 and the scalar base are set once per chunk and the
// SHARE pieces differ in their immediates only (with one wave per SIMD every scalar ALU instruction in an MFMA
// gap costs 8 cycles of issue, the piece itself next to nothing: tools/issue_rate.hip, tools/dma_issue_cost.hip).
// Inline asm on purpose: the compiler models global_load_lds as a FLAT access that may touch LDS, and while one
// is outstanding it turns every LDS wait into lgkmcnt(0), which stalls each group of MFMAs on the read just
// issued; ring_barrier() does the vmcnt bookkeeping instead.
// The scalar base of a chunk's pieces comes out of two scalar adds inside an asm statement, where the chunk's
// refill starts: left to itself the compiler keeps all 40 chunk bases of a board in SGPRs and spills them to
// VGPR lanes, and round-1 builds of that form faulted on the GPU (memory aperture violation,
// profiles/r2/aperture_violation_isa.txt): a spilled base came back by v_readlane_b32 directly in front of the
// asm global_load_lds_dwordx4 that reads it; a vector-memory instruction that reads an SGPR written by a VALU
// instruction needs five wait states, which hipcc pads for its own loads but not for an asm block.
// INVARIANT (tools/isa_hazards.py checks every build's listing; a build that breaks it yields no object file):
// no VALU write of an SGPR within five wait states in front of the asm LDS-DMA that reads it as its base, and
// no compiler write of M0.
struct WStream {
    const unsigned char* sbase;
    const unsigned char* sbase0;
    unsigned voff;
    unsigned lds_share;
    f32x4* ring;
    int rd;
    int fill;
    f32x4 pre[LA + 1];

    __device__ __forceinline__ void set_m0(int slot_off) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_share + (unsigned)slot_off * 16u) : "memory", "m0");
    }
    __device__ __forceinline__ void set_base(int chunk) {
        const unsigned long long b0 = (unsigned long long)(uintptr_t)sbase0;
        unsigned lo, hi;
        asm volatile("s_add_u32 %0, %2, %4\n\ts_addc_u32 %1, %3, 0"
                     : "=&s"(lo), "=s"(hi)
                     : "s"((unsigned)b0), "s"((unsigned)(b0 >> 32)), "s"((unsigned)(chunk * CHUNK * 1024))
                     : "scc");
        sbase = reinterpret_cast<const unsigned char*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
    }
    __device__ __forceinline__ void fetch_piece(int i) {
#ifdef T16_ABL_NODMA   // timing ablation: no weight DMA (stale ring contents, wrong scores)
        return;
#endif
        if (i == 0) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase) : "memory");
        if (i == 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(voff), "s"(sbase) : "memory");
        if (i == 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(voff), "s"(sbase) : "memory");
        if (i == 3) asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(voff), "s"(sbase) : "memory");
    }
    __device__ __forceinline__ void prefetch_mailbox(const unsigned char* board_bytes, unsigned lds_dst, unsigned voff) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2"
                     ::"s"(lds_dst), "v"(voff), "s"(board_bytes) : "memory", "m0");
    }
    __device__ __forceinline__ void prime(const f32x4* stream, f32x4* ring_base, int lane, int wave) {
        typedef __attribute__((address_space(3))) f32x4* lds_ptr;
        ring = ring_base + lane;
        lds_share = (unsigned)(uintptr_t)(lds_ptr)(ring_base + wave * SHARE * 64);
        sbase0 = sbase = reinterpret_cast<const unsigned char*>(stream + wave * SHARE * 64);
        voff = (unsigned)lane * 16u;
#pragma unroll
        for (int c = 0; c < AHEAD; ++c) {
            set_m0(c * CHUNK * 64);
            set_base(c);
#pragma unroll
            for (int i = 0; i < SHARE; ++i) fetch_piece(i);
        }
        rd = (SLOTS - 1) * CHUNK * 64;
    }
    template <int EXTRA>
    __device__ __forceinline__ void next_chunk() {
        rd = (rd == (SLOTS - 1) * CHUNK * 64) ? 0 : rd + CHUNK * 64;
        __builtin_amdgcn_sched_barrier(0);
        ring_barrier<EXTRA>();
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void aim_refill(int chunk) {
        fill = rd + (AHEAD + 1) * CHUNK * 64;
        while (fill >= SLOTS * CHUNK * 64) fill -= SLOTS * CHUNK * 64;
        set_m0(fill);
        set_base(chunk % WALK_CHUNKS);
    }
    __device__ __forceinline__ void refill_step(int i, int c) {
        if (i < SHARE) fetch_piece(i);
        if (i == SHARE) aim_refill(c + 1 + AHEAD);
    }
    __device__ __forceinline__ void start() {
        next_chunk<0>();
        fill = rd + AHEAD * CHUNK * 64;
        if (fill >= SLOTS * CHUNK * 64) fill -= SLOTS * CHUNK * 64;
        set_m0(fill);
        set_base(AHEAD);
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            refill_step(i, 0);
            pre[i] = ring[rd + i * 64];
        }
    }
    __device__ __forceinline__ void board_start() { rd = 0; }
    // g: stream group index, a compile-time constant after unrolling; groups are taken in order.
    __device__ __forceinline__ bf16x8 take(int g) {
        const int nx = g + LA, i = nx % CHUNK;
        const bool extra = nx > kPrefetchGroup + LA && nx <= kPrefetchGroup + LA + AHEAD * CHUNK;
        if (i == 0) { if (extra) next_chunk<1>(); else next_chunk<0>(); }
        refill_step(i, nx / CHUNK);
        pre[nx % (LA + 1)] = ring[rd + i * 64];
        __builtin_amdgcn_sched_barrier(0);
        return __builtin_bit_cast(bf16x8, pre[g % (LA + 1)]);
    }
};

// bf16(max(a, 0)), bf16(max(b, 0)) packed low / high (one v_cvt_pk_bf16_f32 + one signed 16-bit max)
__device__ __forceinline__ unsigned pack_relu(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = __builtin_convertvector(f32x2{a, b}, bf16x2);
    s16x2 s = __builtin_bit_cast(s16x2, v);
    s = __builtin_elementwise_max(s, s16x2{0, 0});
    return __builtin_bit_cast(unsigned, s);
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }

// ---- pieces shared by the kernels whose column tile is a run-time value (eval_t16s.hip, eval_t16h.hip) ----------
// One-hot image of the board, shared by the waves of a board: grid[rank -7 .. 14][file -7 .. 16][2 halves] of
// 16-byte entries.  Every tap of every conv (up to 7 ranks / files away) of every square lands inside it, so a
// unit's offset is a compile-time immediate on top of the lane's square whatever column tile the wave has.
// Off-board entries are zeroed once per workgroup and never written; the 64 squares are overwritten per board.
constexpr int SG_ROW0 = 7, SG_ROWS = 22, SG_COLS = 24, SG_COL0 = 7, SG_ENTRY = 32;
constexpr int SG_ROW_BYTES = SG_COLS * SG_ENTRY;        // 768: three bank windows, the two ranks of a tile never meet in a bank
constexpr int SG_BYTES = SG_ROWS * SG_ROW_BYTES;        // 16 896
constexpr int XP_W = 10, POS_BYTES = 64, ROW_BYTES = XP_W * POS_BYTES, PLANE_BYTES = XP_W * ROW_BYTES;

// ---- stage 1: units = (conv, k-step), every one computed ------------------------------------------------
constexpr int kUnits = 39;
struct S1Split {
    int n;
    short conv[kUnits + 1], g0[kUnits + 1];
    int imm[kUnits + 1], da[kUnits + 1], db[kUnits + 1];
    bool new_base[kUnits + 1], first_of_conv[kUnits + 1];
};
constexpr S1Split make_s1_split() {
    S1Split T{};
    int n = 0, g = 0;
    for (int i = 0; i < kNumInputConvs; ++i)
        for (int s = 0; s < steps_of(i); ++s) {
            const TapPair tp = step_taps(i, s);
            const int tb = tp.b >= 0 ? tp.b : tp.a;
            const int off_a = ((tap_dr(i, tp.a) + SG_ROW0) * SG_COLS + tap_df(i, tp.a) + SG_COL0) * SG_ENTRY;
            const int off_b = ((tap_dr(i, tb) + SG_ROW0) * SG_COLS + tap_df(i, tb) + SG_COL0) * SG_ENTRY;
            T.conv[n] = (short)i;
            T.g0[n] = (short)g;
            T.imm[n] = off_a < off_b ? off_a : off_b;
            T.da[n] = off_a - T.imm[n];
            T.db[n] = off_b - T.imm[n];
            T.new_base[n] = n == 0 || T.da[n] != T.da[n - 1] || T.db[n] != T.db[n - 1];
            T.first_of_conv[n] = s == 0;
            g += kRowTiles[i];
            ++n;
        }
    T.n = n;
    return T;
}
constexpr S1Split kS1s = make_s1_split();
constexpr bool s1_split_in_grid() {
    for (int u = 0; u < kS1s.n; ++u) {
        // the furthest entry a lane reads: rank 7 + (2 * 3 + 1), file 7 of the unit's larger offset
        const int hi = kS1s.imm[u] + (kS1s.da[u] > kS1s.db[u] ? kS1s.da[u] : kS1s.db[u]);
        if (kS1s.imm[u] < 0 || hi + (7 * SG_COLS + 7) * SG_ENTRY + 2 * 16 > SG_BYTES || hi >= 65536) return false;
    }
    return true;
}
static_assert(kS1s.n == kUnits && s1_split_in_grid(), "every tap of every square lands inside the image");

#ifdef BF_STAMPS
#define BF_STAMP(i) do { stamps[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define BF_STAMP(i) ((void)0)
#endif
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

}  // namespace
}  // namespace t16
}  // namespace bf
