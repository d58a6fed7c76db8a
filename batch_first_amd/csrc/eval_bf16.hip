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
// (x1 and the 128-channel accumulators of both tiles live in them: ~400), so one wave per SIMD.
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
// Weights stream through a lockstep LDS ring as in the fp32 kernel (LDS-DMA into a 4-slot ring,
// one barrier per 16-group chunk), with one 1-KiB group per fragment and a 7-group register
// look-ahead.  Workgroups are persistent (one per compute unit): the stream is walked as a ring,
// board after board, the next board's mailbox is prefetched by LDS-DMA, the head kernels stay in
// registers.  With one wave per SIMD the kernel is bound by instructions issued per MFMA gap
// (tools/issue_rate.hip: scalar ALU, v_accvgpr_read and dependent VALU instructions cost 8 cycles
// of issue each), hence: MFMA results in VGPRs (build flag -amdgpu-mfma-vgpr-form), one
// instruction per LDS-DMA piece, every stage a flat list of MFMA slots with the LDS reads, packs
// and LDS writes of its neighbours placed in the gaps.  DESIGN.md section 3.2 has the measurements.
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
// Chunks the refill runs ahead of the chunk being read: with three or more slots the slot refilled at a
// chunk barrier is that of the chunk before the one just finished (SLOTS-2 ahead); with two slots it is
// the one just finished (one ahead), and the barrier also waits for this wave's own LDS reads.
constexpr int AHEAD = SLOTS == 2 ? 1 : SLOTS - 2;
static_assert(CHUNK <= kStreamChunkMax, "");
static_assert(SHARE < CHUNK && CHUNK % WAVES == 0, "every wave fetches whole groups");
static_assert(LA <= CHUNK && SHARE <= CHUNK && SLOTS >= 2, "");
constexpr int WALK_CHUNKS = kStreamWalkGroups / CHUNK;   // chunks per board (the stream is walked as a ring)
static_assert(kStreamWalkGroups % CHUNK == 0 && WALK_CHUNKS % SLOTS == 0 && kStreamWalkGroups % (LA + 1) == 0,
              "every board starts in the same phase of the ring and of the register look-ahead");

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
    __attribute__((aligned(16))) float bias3[6][2][16];  // shifts of x4..x9 in accumulator order: [layer][lane half][register]
    __attribute__((aligned(16))) float bias2[4][2][16];  // shifts of x3 the same way: [32-channel tile][lane half][register]
    f32x4 hpart[WAVES][64];                              // stage 5: per-wave partial sums of the 8x8 layer
    __attribute__((aligned(16))) unsigned char next_mailbox[WAVES][256];  // the wave's next board, prefetched by LDS-DMA
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
// EXTRA = 1 for the AHEAD barriers after the next board's mailbox prefetch (one more LDS-DMA, issued
// right after a chunk's pieces): it is younger than the pieces those barriers need and may stay in
// flight; the third barrier after it retires it (in order) with the pieces it waits for.
template <int EXTRA>
__device__ __forceinline__ void ring_barrier() {
    constexpr int N = (AHEAD - 1) * SHARE + EXTRA;
    static_assert(N < 64, "vmcnt is a 6-bit counter");
#ifdef BF16_ABL_NOBAR
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#else
    if (SLOTS == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

// The next board's mailbox is prefetched after the take of this group (stage 2, second hidden tile):
// the take that issues the last piece of its chunk.
constexpr int kPrefetchGroup = ((kGemmBase + kPerC2Groups + LA + CHUNK - 1) / CHUNK) * CHUNK + SHARE - 1 - LA;
static_assert((kPrefetchGroup + LA) % CHUNK == SHARE - 1 && kPrefetchGroup >= kGemmBase &&
              kPrefetchGroup + 3 * CHUNK < kStreamGroups, "");

// Lockstep walk over the A-fragment stream through the LDS ring (see eval_f32.hip).
struct WStream {
    const unsigned char* sbase;  // wave-uniform: this wave's share of the next chunk to fetch (global memory)
    const unsigned char* sbase0; // ... of chunk 0: the walk wraps after WALK_CHUNKS chunks
    unsigned voff;               // lane * 16
    unsigned lds_share;          // LDS byte address of this wave's share of ring slot 0
    f32x4* ring;
    int rd;                // float4 offset of the slot the look-ahead reads from
    int fill;              // float4 offset of the slot being refilled
    f32x4 pre[LA + 1];

    // LDS-DMA: one 1-KiB piece of this wave's share straight from L2 into a ring slot, lane-linear.
    // One instruction per piece: global address = scalar base + lane * 16 + immediate, LDS address
    // = M0 + immediate + lane * 16 (the instruction's immediate offset applies to both sides), so
    // M0 and the scalar base are set once per chunk (set_m0, and the advance after the last piece)
    // and the SHARE pieces differ in their immediates only.  Measured (tools/issue_rate.hip,
    // tools/dma_issue_cost.hip): with one wave per SIMD every scalar ALU instruction in an MFMA gap
    // costs 8 cycles of issue and the piece itself next to nothing, so the earlier form (64-bit
    // address arithmetic + s_mov m0 + s_nop per piece) cost more than the load.
    // Inline asm on purpose: the compiler models global_load_lds as a FLAT access that may touch
    // LDS, and while one is outstanding it turns every LDS wait into lgkmcnt(0), which stalls each
    // group of MFMAs on the read just issued.  ring_barrier() does the vmcnt bookkeeping instead;
    // ordinary loads interleaved by the compiler only make its own vmcnt waits stricter
    // (vector-memory operations retire in order).  M0 is not written by anything else in this
    // kernel (no dynamic register indexing, no other LDS-DMA, no GWS).
    __device__ __forceinline__ void set_m0(int slot_off) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_share + (unsigned)slot_off * 16u) : "memory", "m0");
    }
    // The scalar base of a chunk's pieces is computed where the chunk's refill starts, by two scalar adds
    // the compiler cannot hoist.  Left to itself it keeps all 40 chunk bases of a board in SGPRs for the
    // whole kernel and spills SGPRs to VGPR lanes.  Round-1 builds of that form faulted on the GPU (memory
    // aperture violation).  Cause (profiles/r2/aperture_violation_isa.txt): a spilled base came back by
    //     v_readlane_b32 s2, v255, 13 / v_readlane_b32 s3, v255, 14 / global_load_lds_dwordx4 v252, s[2:3]
    // with no wait state between.  A vector-memory instruction that reads an SGPR written by a VALU
    // instruction needs five; hipcc pads that pair for its own loads, but an asm block is opaque to its
    // hazard recogniser, so the DMA issued with the pair's previous contents as its 64-bit base.
    // INVARIANT: no VALU write of an SGPR within five wait states in front of an asm vector-memory
    // instruction that reads it, and no compiler write of M0.  This form meets it by construction where it
    // can (the base comes out of scalar adds, which need no wait states, and is consumed within the same
    // chunk; sgpr_spill_count is 0), and the Makefile checks the listing of every build for it
    // (tools/isa_hazards.py): a build that breaks it yields no object file.
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
        static_assert(SHARE <= 8, "two runs of four pieces: immediate offsets of a global instruction reach 4095");
        if (i == 4) {   // second half of an 8-piece share: both sides 4 KiB on
            asm volatile("s_add_u32 m0, m0, 0x1000\n\ts_nop 0" ::: "memory", "m0", "scc");
            sbase += 4096;
        }
#ifndef BF16_ABL_NODMA
        if ((i & 3) == 0) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase) : "memory");
        if ((i & 3) == 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(voff), "s"(sbase) : "memory");
        if ((i & 3) == 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(voff), "s"(sbase) : "memory");
        if ((i & 3) == 3) asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(voff), "s"(sbase) : "memory");
#endif
    }
    // The wave's next board (32 mailbox bytes) into lds.next_mailbox by LDS-DMA: lanes 0-7 carry it,
    // the others repeat it.  Call right after the last piece of a chunk (M0 is set again by the
    // next take: aim_refill) and use ring_barrier<1> at the next two barriers.
    __device__ __forceinline__ void prefetch_mailbox(const unsigned char* board_bytes, unsigned lds_dst, int lane) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2"
                     ::"s"(lds_dst), "v"((unsigned)(lane & 7) * 4u), "s"(board_bytes) : "memory", "m0");
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
    // After the barrier every wave's share of chunk k is in LDS and nobody reads chunk k-2 (k-1 with two
    // slots) any more, so its slot takes chunk k+AHEAD.  Nothing but the barrier happens here: the pipe
    // holds one MFMA (32 cycles) of work while the wave stands at it, so M0 and the scalar base of the
    // refill were set in an MFMA gap of the chunk before (aim_refill).
    template <int EXTRA>
    __device__ __forceinline__ void next_chunk() {
        rd = (rd == (SLOTS - 1) * CHUNK * 64) ? 0 : rd + CHUNK * 64;
        __builtin_amdgcn_sched_barrier(0);
        ring_barrier<EXTRA>();
        __builtin_amdgcn_sched_barrier(0);
    }
    // M0 and scalar base for the refill that follows the NEXT chunk barrier: stream chunk `chunk` into
    // the slot AHEAD + 1 after the one being read.  Called once the pieces of the current refill (and
    // the mailbox prefetch, which borrows M0 right after them) have been issued.
    __device__ __forceinline__ void aim_refill(int chunk) {
        fill = rd + (AHEAD + 1) * CHUNK * 64;
        while (fill >= SLOTS * CHUNK * 64) fill -= SLOTS * CHUNK * 64;
        set_m0(fill);
        set_base(chunk % WALK_CHUNKS);
    }
    // What the walk does at position i of the chunk being looked ahead into (chunk index c of the walk)
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
    // Every board starts with the ring in this state (WALK_CHUNKS is a multiple of SLOTS): telling the
    // compiler so keeps the slot offsets compile-time constants inside the board loop.
    __device__ __forceinline__ void board_start() { rd = 0; }
    // g: stream group index, a compile-time constant after unrolling; groups are taken in order.
    __device__ __forceinline__ bf16x8 take(int g) {
        const int nx = g + LA, i = nx % CHUNK;
        // the mailbox prefetch follows the take of group kPrefetchGroup: its two next barriers allow for it
        const bool extra = nx > kPrefetchGroup + LA && nx <= kPrefetchGroup + LA + AHEAD * CHUNK;
        if (i == 0) { if (extra) next_chunk<1>(); else next_chunk<0>(); }
        refill_step(i, nx / CHUNK);

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

#ifndef BF_STAMP_P0
#define BF_STAMP_P0 (15 + 23)   // stage 2, flat position of stamp 8: start of the second step
#endif
#ifndef BF_STAMP_LEN
#define BF_STAMP_LEN 23         // groups between stamps 8 and 9
#endif
#ifdef BF_STAMPS
// wave-uniform, low 32 bits, no branch: a conditional store would split the board loop's basic block
// (and the compiler sinks work into the new blocks), 64-bit per-lane stamps made the kernel spill
#define BF_STAMP(i) do { stamps[i] = (unsigned)__builtin_amdgcn_s_memtime(); } while (0)
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

// Stage 1 as one flat list of (conv, tap) steps.  reference: ann_creation_helper.py:211-236 (SAME
// padding: (k-1)*d in total, floor(/2) before).
struct S1Table {
    int conv[96], tap[96], off[96][2], group0[96], first_step[kNumInputConvs + 1];
    int steps;
};
constexpr S1Table make_s1_table() {
    S1Table t{};
    int k = 0, g = 0;
    for (int i = 0; i < kNumInputConvs; ++i) {
        const InputConv cv = kInputConvs[i];
        const int ph = ((cv.k - 1) * cv.dh) / 2, pw = ((cv.k - 1) * cv.dw) / 2;
        t.first_step[i] = k;
        for (int tap = 0; tap < cv.k * cv.k; ++tap, ++k) {
            const int dr = (tap / cv.k) * cv.dh - ph, df = (tap % cv.k) * cv.dw - pw;
            t.conv[k] = i;
            t.tap[k] = tap;
            t.off[k][0] = dr * MB_W + df;            // tile 0's square of this lane
            t.off[k][1] = (4 + dr) * MB_W + df;      // tile 1: 4 ranks up
            t.group0[k] = g;
            g += cv.tiles;
        }
    }
    t.first_step[kNumInputConvs] = k;
    t.steps = k;
    for (int j = k; j < 96; ++j) {  // look-ahead past the end re-reads the last step (harmless)
        t.conv[j] = kNumInputConvs - 1;
        t.tap[j] = 0;
        t.off[j][0] = t.off[k - 1][0];
        t.off[j][1] = t.off[k - 1][1];
        t.group0[j] = g;
    }
    return t;
}
constexpr S1Table kS1 = make_s1_table();
static_assert(kS1.steps == 71 && kS1.group0[70] + 1 == kInputGroups, "");

// ReLU + bf16 of the finished accumulators of input conv I, piece by piece (one piece = one packed
// dword of one tile), so that the pieces can ride in the MFMA gaps of the next conv.
template <bool DBG>
__device__ __forceinline__ void s1_pack_piece(int I, int piece, const f32x16 (&acc)[2][2], unsigned (&P)[2][kX1Dwords], int hl,
                                              int ln, float* dbg_board) {
    const int half = kX1RegCount[I] / 2;
    const int T = piece / half, d = piece % half;
    const int tile = (2 * d) / 16, t = (2 * d) % 16;
    unsigned p = pack_relu(acc[T][tile][t], acc[T][tile][t + 1]);
    // pinned here: with nothing but later uses, the compiler sinks all 120 packs past stage 1's last MFMA
    // into the block where stage 2 starts, and the matrix pipe waits for them
    asm volatile("" : "+v"(p));
    P[T][kX1RegBase[I] / 2 + d] = p;
    if (DBG) {
        const int c0 = 2 * (2 * d) + hl, c1 = 2 * (2 * d + 1) + hl;  // local channels of the pair
        if (c0 < kInputConvs[I].ch) dbg_board[(kInputConvs[I].chan_base + c0) * 64 + 32 * T + ln] = bf16_lo(p);
        if (c1 < kInputConvs[I].ch) dbg_board[(kInputConvs[I].chan_base + c1) * 64 + 32 * T + ln] = bf16_hi(p);
    }
}

// The B operand of a step is found in two dependent LDS reads (code byte of the shifted square,
// then that code's one-hot row), so both run ahead of the MFMAs: codes CQ + BQ steps, rows BQ
// steps.  The ReLU + pack of conv I rides in the gaps of conv I+1 (its last MFMA is then at least
// two steps old), at most 4 pieces per group.  mbc: tile 0's square of this lane in the byte
// mailbox; tbl: the one-hot table at this lane's half.
template <bool DBG>
__device__ __forceinline__ void stage1(WStream& ws, const unsigned char* mbc, const u32x4* tbl, unsigned (&P)[2][kX1Dwords],
                                       int hl, int ln, float* dbg_board) {
#ifndef BF16_S1_Q
#define BF16_S1_Q 2
#endif
    constexpr int CQ = BF16_S1_Q, BQ = BF16_S1_Q, NC = CQ + BQ;
    int code[NC][2];
    u32x4 b[BQ][2];
    f32x16 acc[2][2][2];  // [conv parity][board tile][channel tile]
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int T = 0; T < 2; ++T) code[k][T] = mbc[kS1.off[k][T]];
#pragma unroll
    for (int k = 0; k < BQ; ++k)
#pragma unroll
        for (int T = 0; T < 2; ++T) b[k][T] = tbl[code[k][T] * 2];
#pragma unroll
    for (int k = 0; k < kS1.steps; ++k) {
        const int I = kS1.conv[k], tap = kS1.tap[k], par = I & 1;
        const int tiles = kInputConvs[I].tiles;
        u32x4 bk[2] = {b[k % BQ][0], b[k % BQ][1]};
#pragma unroll
        for (int tile = 0; tile < 2; ++tile) {
            if (tile >= tiles) continue;
            const bf16x8 w = ws.take(kS1.group0[k] + tile);
#pragma unroll
            for (int T = 0; T < 2; ++T)
                acc[par][T][tile] = BF_MFMA16(w, __builtin_bit_cast(bf16x8, bk[T]), tap == 0 ? zero16() : acc[par][T][tile]);
            if (tile == 0) {
                // rows for step k+BQ (their codes were read CQ steps ago), codes for step k+NC
#pragma unroll
                for (int T = 0; T < 2; ++T) b[k % BQ][T] = tbl[code[(k + BQ) % NC][T] * 2];
#pragma unroll
                for (int T = 0; T < 2; ++T) code[k % NC][T] = mbc[kS1.off[k + NC][T]];
            }
            if (I > 0) {
                const int j = tap * tiles + tile;  // group of this conv
                const int total = kX1RegCount[I - 1];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int piece = 4 * (j - 1) + q;
                    if (j >= 1 && piece < total) s1_pack_piece<DBG>(I - 1, piece, acc[par ^ 1], P, hl, ln, dbg_board);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int piece = 0; piece < kX1RegCount[kNumInputConvs - 1]; ++piece)
        s1_pack_piece<DBG>(kNumInputConvs - 1, piece, acc[(kNumInputConvs - 1) & 1], P, hl, ln, dbg_board);
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

// MODE 0: scores; 1: scores + every intermediate activation (debug); 2: features -- x9 as float32
// [board][32][64] into `scores`, no head (the move-scoring network, see eval_f32.hip).
template <int MODE>
__global__ __launch_bounds__(THREADS, 1) void k_eval_bf16(const uint8_t* __restrict__ mailbox, unsigned n_boards,
                                                           const f32x4* __restrict__ wstream,
                                                           const float* __restrict__ bias,
                                                           const f32x4* __restrict__ headw,
                                                           float* __restrict__ scores, float* __restrict__ dbg) {
    // Persistent workgroups: one per compute unit, each walks the board quads q = blockIdx.x,
    // blockIdx.x + gridDim.x, ...  The weight ring never drains between boards (the stream is walked
    // as a ring), the next board's mailbox arrives by LDS-DMA during stage 2, and the head kernels of
    // the wave's squares stay in registers for the lifetime of the workgroup.
    constexpr bool DBG = MODE == 1;
    __shared__ __attribute__((aligned(16))) Lds lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63, hl = lane >> 5, ln = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rank = ln >> 3, file = ln & 7;                 // tile 0; tile 1 is 4 ranks up
    const unsigned n_quads = (n_boards + BPB - 1) / BPB;
    typedef __attribute__((address_space(3))) unsigned char* lds_bytes;

#ifdef BF_STAMPS
    unsigned stamps[16] = {};
#endif
    WStream ws;
    ws.prime(wstream, &lds.wring[0][0][0], lane, wave);

    // ---- once per workgroup: the one-hot table, the 3x3 layers' shifts, the head kernels ----------
    unsigned char* region = lds.board[wave];
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
    for (int i = tid; i < 6 * 32; i += THREADS)
        lds.bias3[i >> 5][(i >> 4) & 1][i & 15] = bias[kBiasX4 + 32 * (i >> 5) + rowmap(i & 15, (i >> 4) & 1)];
    for (int i = tid; i < 4 * 32; i += THREADS)
        lds.bias2[i >> 5][(i >> 4) & 1][i & 15] = bias[kBiasX3 + 32 * (i >> 5) + rowmap(i & 15, (i >> 4) & 1)];
    constexpr int SQW = 64 / WAVES;  // squares (k-steps of the head) per wave
    u32x4 ha[SQW];
    if (MODE != 2) {
        const f32x4* hwp = headw + (size_t)wave * SQW * 64 + lane;
#pragma unroll
        for (int i = 0; i < SQW; ++i) ha[i] = __builtin_bit_cast(u32x4, hwp[i * 64]);
    }
    // shift and logit weight of this thread's channel of the 8x8 layer (the reducing threads, tid < 16 * BPB):
    // loaded once, because a compiler-tracked global load inside the board loop would be waited for with
    // vmcnt(0), i.e. behind every LDS-DMA piece in flight
    const float head_shift = MODE != 2 ? bias[kBiasX10 + (tid & 15)] : 0.0f;
    const float head_logit = MODE != 2 ? bias[kLogitW + (tid & 15)] : 0.0f;
    // the wave's two zero-haloed activation planes, zeroed once (see the end of stage 1)
    for (int i = lane; i < 2 * PLANE_BYTES / 16; i += 64) reinterpret_cast<u32x4*>(region)[i] = u32x4{0u, 0u, 0u, 0u};
    {   // the first board's mailbox directly (every later one is prefetched)
        const unsigned b0 = blockIdx.x * BPB + wave;
        const unsigned bc = b0 < n_boards ? b0 : n_boards - 1;
        if (lane < 8)
            reinterpret_cast<unsigned*>(lds.next_mailbox[wave])[lane] =
                reinterpret_cast<const unsigned*>(mailbox + (size_t)bc * kMailboxBytes)[lane];
    }
    lds_order();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the table writes have reached LDS ...
    ws.start();  // ... before its barrier lets stage 1 of any wave read them

#pragma unroll 1
    for (unsigned quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
    const unsigned board_raw = quad * BPB + wave;
    const unsigned board = board_raw < n_boards ? board_raw : n_boards - 1;  // padding boards repeat the last one
    unsigned next_raw = (quad + gridDim.x) * BPB + wave;                      // prefetched in stage 2
    const unsigned next_board = __builtin_amdgcn_readfirstlane(next_raw < n_boards ? next_raw : n_boards - 1);
    ws.board_start();
    BF_STAMP(0);
#ifdef BF_STAMPS
    stamps[7] = (unsigned)__builtin_amdgcn_s_memrealtime();  // constant 100 MHz clock, common to the chip
#endif

    // ---- byte mailbox: 15 everywhere, then the board's codes ---------------------------------
    // (two unconditional stores, 512 bytes: the four words past the mailbox are halo bytes zeroed again after stage 1)
    static_assert(MB_WORDS <= 128, "");
    reinterpret_cast<unsigned*>(region)[lane] = 0x0F0F0F0Fu;
    reinterpret_cast<unsigned*>(region)[lane + 64] = 0x0F0F0F0Fu;
    lds_order();
    {
#ifndef BF16_NO_PREFETCH
        const unsigned v = lds.next_mailbox[wave][lane >> 1];
#else
        const unsigned v = mailbox[(size_t)board * kMailboxBytes + (lane >> 1)];
#endif
        const unsigned c = (lane & 1) ? (v >> 4) : (v & 15u);
        region[((lane >> 3) + 7) * MB_W + (lane & 7) + 7] = (unsigned char)c;
    }
    lds_order();

    float* dbg_board = DBG ? dbg + (size_t)board * kDebugFloatsPerBoard : nullptr;
    BF_STAMP(1);

    // ---- stage 1: input convs -> x1 as 60 packed dwords per tile -----------------------------
    unsigned P[2][kX1Dwords];
    {
        const unsigned char* mbc = region + (rank + 7) * MB_W + file + 7;
        const u32x4* tbl = &lds.onehot[0][hl];
        stage1<DBG>(ws, mbc, tbl, P, hl, ln, dbg_board);
    }
    // The byte mailbox is dead (only this wave reads it): its bytes are halo squares 0-7 of plane 0 again.
    // Everything else in the two planes is either halo, zero since the workgroup started (the layers write
    // interior squares only), or interior, which every layer overwrites before it is read.
    lds_order();
    // One unconditional 1-KiB write (no branch: a new basic block here is where the compiler would sink
    // stage 1's packs to): squares 0-15 of plane 0, i.e. the mailbox bytes in the first halo row and a few
    // squares after it that hold nothing live at this point.
    static_assert(MB_WORDS * 4 <= 1024 && 1024 <= PLANE_BYTES, "");
    reinterpret_cast<u32x4*>(region)[lane] = u32x4{0u, 0u, 0u, 0u};

    BF_STAMP(2);
    // ---- stage 2: 1x1 236->512 (+ReLU) chained into 1x1 512->128 ---------------------------
    // The 128-channel accumulators start from the shifts of x3, read from LDS (lds.bias2) four registers
    // at a time in the gaps of the first tile's 236->512 groups: scalar loads here would sit between stage 1's
    // last MFMA and stage 2's first with the matrix pipe idle, and turn LDS waits into lgkmcnt(0).
    f32x16 acc3[2][4];
    auto load_shift = [&](int r) {   // r = 0..31: board tile, 32-channel tile, quarter
        const int T = r >> 4, c3 = (r >> 2) & 3, q = r & 3;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&lds.bias2[c3][hl][4 * q]);
        acc3[T][c3][4 * q + 0] = v.x;
        acc3[T][c3][4 * q + 1] = v.y;
        acc3[T][c3][4 * q + 2] = v.z;
        acc3[T][c3][4 * q + 3] = v.w;
    };

    // Software pipeline across the hidden layer's 32-channel tiles, in the stream order kGemmOrder: while
    // tile u+1's 15 groups of the 236->512 layer multiply (both board tiles share each fragment as it
    // arrives: one LDS read per two MFMAs), tile u's accumulators are packed (ReLU + bf16, two
    // instructions per dword, at most three dwords per group) and its 8 groups of the 512->128 layer
    // are interleaved.  No fragment is held in registers, no phase reads a fragment per MFMA, and the
    // only pack the matrix pipe waits for is that of the last tile.
    {
        f32x16 acc2[2][2];   // [buffer = hidden tile & 1][board tile]
        unsigned H[2][8];    // packed hidden tile: [board tile][dword], B operands of its 512->128 groups
        auto pack_piece = [&](int buf, int k) {   // k = 0..15: dwords 0-3 of tile 0, of tile 1, then dwords 4-7
            const int T = (k >> 2) & 1, j = (k & 3) + 4 * (k >> 3);
#ifdef BF16_ABL_NOPACK   // timing ablation: the hidden tile is never packed (wrong scores)
            asm volatile("" : "=v"(H[T][j]) : "v"(acc2[buf][T][2 * j]), "v"(acc2[buf][T][2 * j + 1]));
#else
            H[T][j] = pack_relu(acc2[buf][T][2 * j], acc2[buf][T][2 * j + 1]);
#endif
        };
#pragma unroll
        for (int p = 0; p < kC2Tiles * kPerC2Groups; ++p) {
            const int g = kGemmBase + p;
            const int kind = kGemmOrder.kind[p], tile = kGemmOrder.tile[p], idx = kGemmOrder.idx[p];
            if (p == BF_STAMP_P0) BF_STAMP(8);
            if (p == BF_STAMP_P0 + BF_STAMP_LEN) BF_STAMP(9);
            const bf16x8 w = ws.take(g);
#ifndef BF16_NO_PREFETCH
            if (g == kPrefetchGroup)
                ws.prefetch_mailbox(mailbox + (size_t)next_board * kMailboxBytes,
                                    (unsigned)(uintptr_t)(lds_bytes)lds.next_mailbox[wave], lane);
#endif
            if (p < 11) {   // the first 512->128 group is at p = 19
                load_shift(3 * p);
                load_shift(3 * p + 1);
                if (3 * p + 2 < 32) load_shift(3 * p + 2);
            }
            if (kind == 0) {
#pragma unroll
                for (int T = 0; T < 2; ++T) {
                    const u32x4 b = u32x4{P[T][4 * idx], P[T][4 * idx + 1], P[T][4 * idx + 2], P[T][4 * idx + 3]};
                    acc2[tile & 1][T] = BF_MFMA16(w, __builtin_bit_cast(bf16x8, b), idx == 0 ? zero16() : acc2[tile & 1][T]);
                }
            } else if (kind == 1) {
                const int sk = idx >> 2, c3 = idx & 3;
#pragma unroll
                for (int T = 0; T < 2; ++T) {
                    const u32x4 b = u32x4{H[T][4 * sk], H[T][4 * sk + 1], H[T][4 * sk + 2], H[T][4 * sk + 3]};
                    acc3[T][c3] = BF_MFMA16(w, __builtin_bit_cast(bf16x8, b), acc3[T][c3]);
                }
            }
            // packs of the tile whose 236->512 groups ended with the previous step
            if (p >= 15 && p < 15 + 15 * 23) {
                const int u = (p - 15) / 23, j = (p - 15) % 23;
                // (two dwords in each of groups 0-3 instead measured slower: a pack in group 0 waits out the
                // MFMA-result hazard of the previous step's last groups)
                const int first = j == 0 ? 0 : j == 1 ? 0 : j == 2 ? 3 : j == 3 ? 6 : 8 + (j - 4);
                const int count = j == 0 ? 0 : j <= 2 ? 3 : j == 3 ? 2 : j <= 11 ? 1 : 0;
#pragma unroll
                for (int k = 0; k < count; ++k) pack_piece(u & 1, first + k);
            }
            if (p == 15 + 15 * 23 - 1) {   // the last tile: its first half before its own 512->128 groups start
#pragma unroll
                for (int k = 0; k < 8; ++k) pack_piece(15 & 1, k);
            }
            if (p >= 15 + 15 * 23 && p < 15 + 15 * 23 + 4) {
                pack_piece(15 & 1, 8 + 2 * (p - 15 - 15 * 23));
                pack_piece(15 & 1, 9 + 2 * (p - 15 - 15 * 23));
            }
            __builtin_amdgcn_sched_barrier(0);
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
    // One flat list of 9 x 36 MFMA slots: chunk c = 0..3 are the four 32-channel input chunks of
    // the 128->32 layer, c = 4..8 the five 32->32 layers; chunk c reads plane c & 1.  A layer
    // needs the whole previous layer, so with one wave per SIMD a layer-by-layer walk leaves the
    // matrix pipe idle for every ReLU + pack -> LDS write -> shifted read chain.  Instead the two
    // tiles of chunks 3..8 run skewed: tile 0 first (its 18 fragments stay in registers for tile
    // 1), so that tile 0's result is packed and written while tile 1 multiplies, and tile 1's while
    // the next layer's tile 0 runs its taps of the two upper rows (dr = -1, 0), which read tile 0's
    // ranks only.  Chunks 0..2 have all their input from the start: both tiles share each fragment
    // as it arrives and the next chunk's input is written in their gaps.  Shifts come from LDS
    // (lds.bias3) as the C operand of a layer's first MFMAs; B operands are read D slots ahead.
    const int R0 = rank + 1;                                // padded row of tile 0's square (tile 1: +4, same & 3)
    const int woff = (R0 * XP_W + file + 1) * POS_BYTES;    // tile 0's square; tile 1 at + TILE1
    int raddr[3][2];
#pragma unroll
    for (int dr = 0; dr < 3; ++dr)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            raddr[dr][s] = ((rank + dr) * XP_W + file) * POS_BYTES + (((2 * s + hl + rank + dr) & 3) << 4);

    // ReLU + bf16 of registers 4g..4g+3 of a tile (channels 8g + 4hl .. +3) into this lane's
    // position of plane `pl`: 8 bytes at channel chunk g, chunk position rotated by the padded row.
    auto write_piece = [&](int pl, int T, int g, const f32x16& a) {
        u32x2 v;
        v.x = pack_relu(a[4 * g + 0], a[4 * g + 1]);
        v.y = pack_relu(a[4 * g + 2], a[4 * g + 3]);
        lds_order();
        *reinterpret_cast<u32x2*>(region + pl * PLANE_BYTES + woff + T * TILE1 + (((g + R0) & 3) << 4) + hl * 8) = v;
        lds_order();
    };
    // slot n -> chunk, tile, (tap, k-step) index
    auto slot_c = [](int n) { return n / 36; };
    auto slot_T = [](int n) { return (n / 36) < 3 ? (n % 36) & 1 : (n % 36) / 18; };
    auto slot_idx = [](int n) { return (n / 36) < 3 ? (n % 36) >> 1 : (n % 36) % 18; };
    auto load_b = [&](int n) -> u32x4 {
        const int c = slot_c(n), T = slot_T(n), idx = slot_idx(n);
        const int tap = idx >> 1, s = idx & 1;
        return *reinterpret_cast<const u32x4*>(region + (c & 1) * PLANE_BYTES + raddr[tap / 3][s] + T * TILE1 + (tap % 3) * POS_BYTES);
    };
#ifndef BF16_CONV_D
#define BF16_CONV_D 4
#endif
    constexpr int D = BF16_CONV_D, NSLOTS = 9 * 36;   // B operands read D slots ahead
    static_assert(D == 4, "a read further ahead passes the write_piece of its input (D = 8: wrong scores, and no faster)");
    f32x16 A[2][2];   // [set][tile]: chunks 0..3 accumulate in set 0, layer c >= 4 in set (c & 1) ^ 1
    f32x16 bs;        // shift of the layer about to start
    u32x4 bq[D];
    bf16x8 f[18];
    bf16x8 wj{};
    auto load_bias = [&](int l, int i) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(&lds.bias3[l][hl][4 * i]);
        bs[4 * i] = v.x; bs[4 * i + 1] = v.y; bs[4 * i + 2] = v.z; bs[4 * i + 3] = v.w;
    };
#pragma unroll
    for (int p = 0; p < 8; ++p) write_piece(0, p >> 2, p & 3, acc3[p >> 2][0]);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_bias(0, i);
#pragma unroll
    for (int n = 0; n < D; ++n) bq[n] = load_b(n);
#pragma unroll
    for (int n = 0; n < NSLOTS; ++n) {
        const int c = slot_c(n), r = n % 36, T = slot_T(n), idx = slot_idx(n);
        if (n == 4 * 36) BF_STAMP(4);
        const bool joint = c < 3;
        const int o = c <= 3 ? 0 : ((c & 1) ^ 1);
        bf16x8 w;
        if (T == 0) {
            w = ws.take(kConvBase + c * 18 + idx);
            if (joint) wj = w; else f[idx] = w;
        } else {
            w = joint ? wj : f[idx];
        }
        const bool fresh = idx == 0 && (c == 0 || c >= 4);
        A[o][T] = BF_MFMA16(w, __builtin_bit_cast(bf16x8, bq[n % D]), fresh ? bs : A[o][T]);
        if (n + D < NSLOTS) bq[n % D] = load_b(n + D);
        if (joint) {
            // input of chunk c+1 (plane (c+1) & 1; chunk c-1's reads of it were issued long ago)
            if (r >= 2 && r <= 16 && (r & 1) == 0) write_piece((c + 1) & 1, (r - 2) >> 3, ((r - 2) >> 1) & 3, acc3[(r - 2) >> 3][c + 1]);
        } else {
            if (c < 8 && r >= 20 && r < 24) write_piece((c + 1) & 1, 0, r - 20, A[o][0]);   // tile 0 of x(c+1)
            if (c < 8 && r >= 26 && r < 30) load_bias(c + 1 - 3, r - 26);
            if (c >= 4 && r >= 2 && r < 6) write_piece(c & 1, 1, r - 2, A[o ^ 1][1]);      // tile 1 of this layer's input
            if (DBG && c < 8 && r == 20) dump_tile(dbg_board + 64 * (236 + 128 + 32 * (c - 3)), hl, ln, A[o][0]);
            if (DBG && c >= 4 && r == 2) dump_tile(dbg_board + 64 * (236 + 128 + 32 * (c - 4)), hl, 32 + ln, A[o ^ 1][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x16 (&acc)[2] = A[1];  // x9 before its ReLU

    BF_STAMP(5);
    // The walk's padding groups: they bring the ring and the look-ahead back to their start phase,
    // with the next board's first fragments already on their way into registers.
#pragma unroll
    for (int g = kStreamGroups; g < kStreamWalkGroups; ++g) (void)ws.take(g);

    // ---- stage 5: 8x8 valid conv (2048 -> 16) across the workgroup's boards, ReLU, logit ------
    if (MODE == 2) {
        if (board_raw < n_boards) {
            float* out = scores + (size_t)board * 2048;
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int t = 0; t < 16; ++t) out[rowmap(t, hl) * 64 + 32 * T + ln] = fmaxf(acc[T][t], 0.0f);
        }
    } else {
    lds_order();
#pragma unroll
    for (int p = 0; p < 8; ++p) write_piece(1, p >> 2, p & 3, acc[p >> 2]);   // x9 -> plane 1
    if (DBG) {
        dump_tile(dbg_board + 64 * (236 + 128 + 32 * 5), hl, ln, acc[0]);
        dump_tile(dbg_board + 64 * (236 + 128 + 32 * 5), hl, 32 + ln, acc[1]);
    }
    const int col = lane & 15, kc = lane >> 4;
    lds_barrier();   // every wave's x9 is in its plane 1 (LDS only: the ring's DMA stays in flight)
    {
        // out[board][co] = sum over (square, ci) of x9[board][square][ci] * W[square][ci][co]: the two
        // operands of v_mfma_f32_16x16x32_bf16 have the same lane layout, so the activations go first
        // (rows = boards) and the kernels second, where they can stay in the registers they were loaded
        // into; all of the wave's activation reads are issued before the first MFMA.
        const unsigned char* xb = lds.board[col < BPB ? col : 0] + PLANE_BYTES;
        u32x4 hb[SQW];
#pragma unroll
        for (int i = 0; i < SQW; ++i) {
            const int sqk = wave * SQW + i;
            const int Rk = (sqk >> 3) + 1;
            hb[i] = *reinterpret_cast<const u32x4*>(xb + (Rk * XP_W + (sqk & 7) + 1) * POS_BYTES + (((kc + Rk) & 3) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
        // Inline asm so that the kernels are read from the AGPRs they live in (the builtin copies each to
        // VGPRs first: 4 x 8 cycles of issue per 16-cycle MFMA).  One accumulation chain on one register
        // (no wait states needed between its links); the s_nop after the last link covers the
        // MFMA-result-to-LDS-store hazard, which the compiler does not see through inline asm.
        f32x4 hacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < SQW; ++i)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(hacc) : "v"(hb[i]), "a"(ha[i]));
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(hacc));
        lds.hpart[wave][lane] = hacc;
    }
    // After this barrier nobody reads a plane of this board any more (the next board may overwrite
    // them), and hpart is complete; it is rewritten one board later, behind the next barrier of
    // this kind, which the reducing wave reaches only after its reads.
    lds_barrier();
    if (tid < 16 * BPB) {
        const int b = tid >> 4, co = tid & 15;
        // D[row = b][col = co] sits in lane co + 16*(b>>2), register b&3
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w)
            s += reinterpret_cast<const float*>(lds.hpart[w])[(co + 16 * (b >> 2)) * 4 + (b & 3)];
        const float y = fmaxf(s + head_shift, 0.0f);
        const unsigned out_raw = quad * BPB + b;
        if (DBG && out_raw < n_boards) dbg[(size_t)out_raw * kDebugFloatsPerBoard + 64 * (236 + 128 + 32 * 6) + co] = y;
        float v = y * head_logit;  // logit_layer: no bias, no batch-norm (evaluation_ann.py:35-44)
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        if (co == 0 && out_raw < n_boards) scores[out_raw] = v;
    }
    }
    BF_STAMP(6);
#ifdef BF_STAMPS
    stamps[14] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
    }  // board loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's look-ahead DMA must not outlive the workgroup
#ifdef BF_STAMPS
    if (!DBG && dbg && lane == 0) {   // stamps of the workgroup's last board
        long long* out = reinterpret_cast<long long*>(dbg) + ((size_t)blockIdx.x * WAVES + wave) * 16;
        for (int i = 0; i < 15; ++i) out[i] = (long long)stamps[i];   // low 32 bits: differences are taken modulo 2^32
    }
#endif
}

}  // namespace

hipError_t launch(const uint8_t* d_mailbox, size_t n, const unsigned char* d_image, float* d_scores, float* d_debug,
                  hipStream_t stream);

// One persistent workgroup per compute unit of the current device (fewer for small batches).
hipError_t grid_blocks(size_t n, unsigned* blocks) {
    static int cus[64] = {0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    int count = (dev >= 0 && dev < 64) ? cus[dev] : 0;
    if (count == 0) {
        e = hipDeviceGetAttribute(&count, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        if (count <= 0) count = 256;
        if (dev >= 0 && dev < 64) cus[dev] = count;
    }
    const size_t quads = (n + BPB - 1) / BPB;
    *blocks = (unsigned)(quads < (size_t)count ? quads : (size_t)count);
    return hipSuccess;
}

}  // namespace b16

hipError_t launch_eval_bf16(const uint8_t* d_mailbox, size_t n, const unsigned char* d_image, float* d_scores,
                            float* d_debug, hipStream_t stream) {
    return b16::launch(d_mailbox, n, d_image, d_scores, d_debug, stream);
}

hipError_t launch_policy_features_bf16(const uint8_t* d_mailbox, size_t n, const unsigned char* d_image, float* d_x9,
                                       hipStream_t stream) {
    using namespace b16;
    if (n == 0) return hipSuccess;
    unsigned blocks = 0;
    hipError_t e = grid_blocks(n, &blocks);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_eval_bf16<2>, dim3(blocks), dim3(THREADS), 0, stream, d_mailbox,
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
    unsigned blocks = 0;
    hipError_t e = grid_blocks(n, &blocks);
    if (e != hipSuccess) return e;
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
