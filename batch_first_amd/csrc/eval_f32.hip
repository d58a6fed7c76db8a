// The value network as one fp32 MFMA kernel: 4-bit mailboxes in, one score per board out.
//
// Reference graph replaced (TensorFlow ops -> this kernel):
//   create_input_convolutions_shared_weights   ann_creation_helper.py:192-252  (stage 1)
//   build_convolutional_modules, spec at       evaluation_ann.py:268           (stages 2-4)
//   logit_layer + squeeze                      evaluation_ann.py:35-46         (stage 5)
//
// Shape of the computation.  One board = two waves; wave `half` owns squares 32*half ..
// 32*half+31 as the 32 columns of every MFMA tile.  All layers run transposed,
//     out^T[channel][square] += W^T[channel][k] * in^T[k][square]      (v_mfma_f32_32x32x2_f32)
// so that an accumulator tile (rows = channels in registers, columns = squares on lanes) is
// directly the B operand of the next 1x1 layer: nothing but the 3x3 layers' halo exchange ever
// leaves the register file, and no intermediate activation touches HBM.
//   stage 1  nine dilated input convs as implicit GEMM on v_mfma_f32_32x32x16_bf16: B = one-hot
//            of the shifted square's code (exact in bf16) built in registers from a padded byte
//            mailbox in LDS, A = the fp32 weight split into three bf16 terms (exact sum), so
//            fp32 accuracy at 3 x 32 cycles per (tile, tap) instead of 8 x 64.  236 channels
//            end up in 118 registers (channel 2r on lanes 0-31, 2r+1 on lanes 32-63).
//   stage 2  1x1 236->512 and 512->128 chained per 32-channel tile of the hidden layer: the
//            512-wide activation never exists; 128 outputs accumulate in 4 tiles.
//   stage 3  3x3 128->32: each 32-channel tile goes through a zero-haloed 10x10 LDS plane,
//            the B operand is a shifted ds_read.
//   stage 4  five 3x3 32->32 layers the same way.
//   stage 5  8x8 valid conv (2048->16 dense) + ReLU + logit on the VALU, fixed-order reduction.
//
// Weights: a linear stream of A fragments in consumption order (bf_layout.h, pack.cpp).  A
// workgroup is BPB boards (2*BPB waves) walking the stream in lockstep, CHUNK groups at a
// time: every wave fetches 1/(2*BPB) of the chunk two chunks ahead into registers, drops it into
// a 3-slot LDS ring one chunk ahead, and all waves read their A fragments from LDS
// (ds_read_b128, lane-linear, conflict-free).  One barrier per chunk.  L2 traffic per board is
// 1/(2*BPB) of what private streaming costs (measured: private streaming of 2.1 MB per board
// held the kernel 16 % under its matrix-pipe time).  Two workgroups share a CU, so a wave
// waiting at a barrier leaves its SIMD's matrix pipe to a wave of the other workgroup.
//
// Batch-norm is folded into the weights; its shift initialises the accumulators, so every
// epilogue is a bare max(x, 0).  Results are bit-reproducible run to run and independent of a
// board's position in the batch (fixed summation order, no atomics).
#include <hip/hip_runtime.h>

#include "bf_device.h"
#include "bf_kernels.h"
#include "bf_layout.h"

namespace bf {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifdef BF_STAMPS
#define BF_STAMP(i) do { if (lane == 0) stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BF_STAMP(i) ((void)0)
#endif
#define BF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

#ifndef BF_BPB
#define BF_BPB 2
#endif
constexpr int BPB = BF_BPB;            // boards per workgroup
constexpr int WAVES = 2 * BPB;
constexpr int THREADS = 64 * WAVES;
constexpr int CHUNK = kStreamChunk;    // groups per lockstep chunk
// (a wave fetches CHUNK / WAVES groups of every chunk: WStreamT::SHARE)
constexpr int SLOTS = 3;
// BF_PRECISION_F32_SPLIT consumes a group every two 32-cycle MFMAs (the float32 form: every four 64-cycle ones), so a
// chunk lasts 512 cycles instead of 2048 and a fetch has two barriers = 0.4 us to land.  Measured with 3, 4 and 5
// slots (profiles/r3/f32s/ab_slots.txt; 5 needs kStreamPadGroups = 48): 7.81, 7.80, 7.82 ms, no difference -- with
// two workgroups per CU the other workgroup's wave covers the wait -- so the ring keeps three.
#ifndef BF_F32_SPLIT_SLOTS
#define BF_F32_SPLIT_SLOTS 3
#endif
constexpr int SPLIT_SLOTS = BF_F32_SPLIT_SLOTS;
// SPLIT level 2 (BF_PRECISION_F32_SPLIT3) keeps the third term of x1 in LDS (60 KB a workgroup), so its ring has
// chunks of 4 groups: 12 KB, two workgroups a CU as for the other two forms.
constexpr int chunk_of(int split) { return split == 2 && BPB == 2 ? 4 : CHUNK; }   // (BF_BPB = 4 experiments: one workgroup a CU, room for the full ring)
constexpr int slots_of(int split) { return split ? SPLIT_SLOTS : SLOTS; }
#ifndef BF_F32_LA
#define BF_F32_LA 3
#endif
// Groups held in registers ahead of the MFMAs.  One is enough where a group feeds four 64-cycle MFMAs; in
// stage 1 a group feeds ONE 32-cycle MFMA, and with a single group in flight every one of its 240 steps sat
// out an LDS read (stamps, one wave per SIMD: 23.8 k cycles for 7.7 k cycles of MFMA work).
constexpr int LA = BF_F32_LA;
static_assert(LA >= 1 && LA <= 3 && CHUNK % (LA + 1) == 0 && chunk_of(2) % (LA + 1) == 0, "the look-ahead ring must keep its phase over a chunk");
constexpr int C2_UNROLL = 4;           // hidden tiles per loop body: 4 * 46 groups = 23 chunks
static_assert(CHUNK % WAVES == 0 && chunk_of(2) % WAVES == 0, "every wave fetches whole groups");
static_assert(kInputGroups % CHUNK == 0, "the tile loop must start on a chunk boundary");
static_assert((kPerC2Groups * C2_UNROLL) % CHUNK == 0, "loop body must be whole chunks");
static_assert(BPB != 2 || (kPerC2GroupsSplit3 * C2_UNROLL) % chunk_of(2) == 0, "loop body must be whole chunks");
static_assert(kC2Tiles % C2_UNROLL == 0, "");
static_assert(kStreamPadGroups >= SPLIT_SLOTS * CHUNK + LA && kStreamPadGroups >= 3 * CHUNK + LA, "prefetch overrun must stay inside the image");

constexpr int MB_W = 22;     // mailbox padded by 7 squares (largest dilation) on every side
constexpr int MB_BYTES = 496;
constexpr int XP_W = 10;     // activation plane padded by 1 for the 3x3 layers
constexpr int XP_PLANE = 100;
constexpr int XBUF_FLOATS = 32 * XP_PLANE;
constexpr int RED_STRIDE = 17;
static_assert(128 * RED_STRIDE <= XBUF_FLOATS, "the head's reduction scratch aliases one activation buffer");

// SPLIT: the 3x3 layers' activations as THREE bf16 terms (24 significant bits), one plane per term,
// [10x10 positions][32 channels] bf16 = 64 bytes a position: 19.2 KB a board, ONE buffer (two would leave room for one
// workgroup per CU only), so a chunk's writes wait for the readers of the chunk before it at a second barrier.
constexpr int XS_POS_BYTES = 64;
constexpr int XS_PLANE_BYTES = XP_PLANE * XS_POS_BYTES;   // 6 400
constexpr int XS_FLOATS = 3 * XS_PLANE_BYTES / 4;         // 4 800
static_assert(128 * RED_STRIDE <= XS_FLOATS, "the head's reduction scratch aliases the activation planes");

// SPLIT level 2: the third bf16 term of x1, per wave [15 k-steps][64 lanes] x 16 bytes (the B fragment of
// v_mfma_f32_32x32x16_bf16, lane-linear: wave-private, conflict-free), in the space the term planes take afterwards.
constexpr int X1LO_WAVE_FLOATS = 15 * 64 * 4;             // 15 KB
static_assert(2 * X1LO_WAVE_FLOATS >= XS_FLOATS, "the term planes alias the x1 region");

template <int SPLIT>
struct LdsT {
    f32x4 wring[slots_of(SPLIT)][chunk_of(SPLIT)][64];   // shared A-fragment ring
    // per board: two alternating zero-haloed activation buffers (float32), or the three term planes (level 2: in the
    // space that holds the third term of x1 during the 1x1 pair)
    float xbuf[BPB][SPLIT ? 1 : 2][SPLIT == 2 ? 2 * X1LO_WAVE_FLOATS : SPLIT ? XS_FLOATS : XBUF_FLOATS];
    unsigned char mb[BPB][MB_BYTES];     // per board: padded code mailbox
#ifdef BF_ABL_LDSPAD
    char pad[20000];
#endif
};

// Lockstep walk over the A-fragment stream through the LDS ring of S slots.
template <int S, int C>
struct WStreamT {
    static constexpr int CHUNK = C, SHARE = C / WAVES;   // (shadow the file-scope float32 values)
    const f32x4* src;      // this lane's element of this wave's share of the chunk to fetch next
    f32x4* ring;           // &lds.wring[0][0][lane]
    unsigned lds_share;    // LDS byte address of &lds.wring[0][wave * SHARE][0]: where this wave's share of slot 0 lands
    int rd;                // float4 offset of the slot being read
    f32x4 pre[LA + 1];     // group being consumed and the LA groups read from LDS ahead of it

    // LDS-DMA: this wave's SHARE groups of the next chunk go straight from L2 into ring slot
    // `slot_off` (lane-linear 1 KiB per instruction, exactly the layout the readers expect); no
    // staging registers, no ds_write.  One instruction per group: global address = this lane's pointer +
    // immediate, LDS address = M0 + immediate + lane * 16.
    // Inline asm on purpose, as in t16_common.h: the compiler models the builtin (global_load_lds) as an
    // access that may touch LDS, and while one is outstanding it turns every LDS wait into lgkmcnt(0): every
    // second group of MFMAs then sat out the LDS read issued right in front of it (1302 such waits in the
    // listing, 5 % of the 1x1 pair's time).  next_chunk() does the vmcnt bookkeeping instead; loads the
    // compiler interleaves only make its own vmcnt waits stricter (vector-memory operations retire in order).
    // M0 is written by nothing else in this kernel; tools/isa_hazards.py checks every build's listing.
    __device__ __forceinline__ void fetch_into(int slot_off) {
        static_assert(SHARE <= 4, "immediate offsets of a global instruction reach 4095");
#ifdef BF_ABL_NODMA   // timing ablation (wrong scores): nothing is fetched
        return;
#endif
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_share + (unsigned)slot_off * 16u) : "memory", "m0");
#pragma unroll
        for (int i = 0; i < SHARE; ++i) {
            if (i == 0) asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(src) : "memory");
            if (i == 1) asm volatile("global_load_lds_dwordx4 %0, off offset:1024" ::"v"(src) : "memory");
            if (i == 2) asm volatile("global_load_lds_dwordx4 %0, off offset:2048" ::"v"(src) : "memory");
            if (i == 3) asm volatile("global_load_lds_dwordx4 %0, off offset:3072" ::"v"(src) : "memory");
        }
#ifndef BF_ABL_NOSTREAM
        src += CHUNK * 64;
#endif
    }
    __device__ __forceinline__ void prime(const f32x4* stream, f32x4* ring_base, int lane, int wave) {
        typedef __attribute__((address_space(3))) f32x4* lds_ptr;
        ring = ring_base + lane;
        lds_share = (unsigned)(uintptr_t)(lds_ptr)(ring_base + wave * SHARE * 64);
        src = stream + wave * SHARE * 64 + lane;
#pragma unroll
        for (int k = 0; k < S - 1; ++k) fetch_into(k * CHUNK * 64);   // chunks 0 .. S-2
        rd = (S - 1) * CHUNK * 64;  // start()'s next_chunk() moves to slot 0
    }
    // Chunk boundary (a fixed place in the code, the same for all waves of the block): after the
    // barrier every wave's share of chunk k is in LDS and nobody reads chunk k-1 any more, so its
    // slot takes chunk k+S-1.
    __device__ __forceinline__ void next_chunk() {
        const int free_slot = rd;       // chunk k-1
        rd = (rd == (S - 1) * CHUNK * 64) ? 0 : rd + CHUNK * 64;
        // keep the barrier behind the previous group's MFMAs: they cover the latency of the LDS read
        // just issued, which the barrier's lgkmcnt(0) would otherwise sit out with the pipe idle
        __builtin_amdgcn_sched_barrier(0);
#ifdef BF_F32_FULLWAIT
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
        // Only chunk k has to have landed here: this wave's share of it was fetched S-1 barriers ago, the shares
        // of chunks k+1 .. k+S-2 (the (S-2) SHARE youngest loads) may still be in flight.  __syncthreads() would
        // wait for vmcnt(0), i.e. give the fetch one chunk (2048 cycles) to land instead of two.
#ifdef BF_ABL_NOBAR   // timing ablation (races): waves do not wait for each other
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * SHARE) : "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((S - 2) * SHARE) : "memory");
#endif
#endif
        __builtin_amdgcn_sched_barrier(0);
        fetch_into(free_slot);
    }
    // Read groups 0 .. LA-1 once everything staged before them is visible.
    __device__ __forceinline__ void start() {
        next_chunk();
#pragma unroll
        for (int i = 0; i < LA; ++i) pre[i] = ring[rd + i * 64];
    }
    // g: stream group index, a compile-time constant after unrolling; groups are taken in order.
    // Returns group g (already in registers) and starts the LDS read of group g+LA, so the read's
    // latency hides under the MFMAs of groups g .. g+LA-1.  When g+LA opens a chunk, the barrier happens here: this
    // wave holds all of the old chunk in registers by now.
    __device__ __forceinline__ f32x4 take(int g) {
        const int nx = g + LA;
        if (nx % CHUNK == 0) next_chunk();
        pre[nx % (LA + 1)] = ring[rd + (nx % CHUNK) * 64];
        // pin the read ahead of group g's MFMAs (the scheduler otherwise sinks it to its use and
        // the wave sits out the LDS latency with the matrix pipe idle)
        __builtin_amdgcn_sched_barrier(0);
        return pre[g % (LA + 1)];
    }
};

constexpr int input_group_base(int conv) {
    int g = 0;
    for (int i = 0; i < conv; ++i) g += kInputConvs[i].k * kInputConvs[i].k * kInputConvs[i].tiles * kInputSplits;
    return g;
}

__device__ __forceinline__ float pick(int hl, float lo, float hi) { return hl ? hi : lo; }

// max(x, 0) as one integer max on the bit pattern: negative floats (and -0) are negative
// integers, positive floats keep their order.  fmaxf costs a canonicalising v_max on top.
__device__ __forceinline__ float relu(float x) {
    return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0));
}

// Stage 1, one input conv.  reference: ann_creation_helper.py:211-236 (SAME padding:
// (k-1)*d in total, floor(/2) before).
template <int I, class WStream>
__device__ __forceinline__ void input_conv(WStream& ws, const unsigned char* mb_center, int hl,
                                           const float* __restrict__ bias, float (&x1)[kX1Regs]) {
    constexpr InputConv cv = kInputConvs[I];
    constexpr int G0 = input_group_base(I);
    constexpr int ph = ((cv.k - 1) * cv.dh) / 2, pw = ((cv.k - 1) * cv.dw) / 2;
    f32x16 acc[cv.tiles];
#pragma unroll
    for (int tile = 0; tile < cv.tiles; ++tile)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int c = 32 * tile + 2 * t;  // channel of lanes 0-31; lanes 32-63 hold c+1
            acc[tile][t] = (c < cv.ch) ? pick(hl, bias[kBiasX1 + cv.chan_base + c], bias[kBiasX1 + cv.chan_base + c + 1])
                                       : 0.0f;
        }
    // all of this conv's shifted codes first, so their LDS latency is paid once per conv
    int codes[cv.k * cv.k];
#pragma unroll
    for (int tap = 0; tap < cv.k * cv.k; ++tap) {
        const int dr = (tap / cv.k) * cv.dh - ph, df = (tap % cv.k) * cv.dw - pw;
        codes[tap] = mb_center[dr * MB_W + df];
    }
#pragma unroll
    for (int tap = 0; tap < cv.k * cv.k; ++tap) {
        // B fragment: one-hot over planes 8*hl .. 8*hl+7 of the shifted square's code, as 8 bf16
        const unsigned cc = (unsigned)(codes[tap] - 8 * hl);
        const unsigned one = (cc < 8u) ? (0x3F80u << ((cc & 1u) * 16u)) : 0u;  // bf16 1.0 in its half
        u32x4 bu;
        bu.x = ((cc >> 1) == 0u) ? one : 0u;
        bu.y = ((cc >> 1) == 1u) ? one : 0u;
        bu.z = ((cc >> 1) == 2u) ? one : 0u;
        bu.w = ((cc >> 1) == 3u) ? one : 0u;
        const bf16x8 b = __builtin_bit_cast(bf16x8, bu);
#pragma unroll
        for (int tile = 0; tile < cv.tiles; ++tile) {
            const int g = G0 + (tap * cv.tiles + tile) * kInputSplits;
#pragma unroll
            for (int sp = 0; sp < kInputSplits; ++sp) {
                const f32x4 w = ws.take(g + sp);
                acc[tile] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), b, acc[tile], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int tile = 0; tile < cv.tiles; ++tile)
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (32 * tile + 2 * t < cv.ch) x1[cv.chan_base / 2 + 16 * tile + t] = relu(acc[tile][t]);
}

// One 3x3 SAME conv over a 32-channel chunk staged in a zero-haloed LDS plane:
// acc[co][sq] += sum_tap sum_ci W[tap][ci][co] * x[ci][sq + tap offset].
// g0 must be a compile-time constant at every (inlined) call site.
template <class WStream>
__device__ __forceinline__ void conv3x3_chunk(WStream& ws, int g0, const float* xb_lane, f32x16& acc) {
    // B operands are read from LDS one group (4 k-steps) ahead of the MFMAs that use them.
    auto load_b = [&](int idx, float (&b)[4]) {
        const int tap = idx >> 2, sg = idx & 3;
        const int off = (tap / 3) * XP_W + (tap % 3);  // xb_lane already points one row and column back
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = xb_lane[(8 * sg + 2 * j) * XP_PLANE + off];
    };
    float b[2][4];
    load_b(0, b[0]);
#pragma unroll
    for (int idx = 0; idx < 36; ++idx) {
        if (idx + 1 < 36) load_b(idx + 1, b[(idx + 1) & 1]);
        const f32x4 w = ws.take(g0 + idx);
        acc = BF_MFMA(w.x, b[idx & 1][0], acc);
        acc = BF_MFMA(w.y, b[idx & 1][1], acc);
        acc = BF_MFMA(w.z, b[idx & 1][2], acc);
        acc = BF_MFMA(w.w, b[idx & 1][3], acc);
    }
}

// Two bf16 terms of a pair of float32 values, packed (a low, b high): hi = bf16(x) to nearest-even,
// mid = bf16(x - hi); hi + mid carries 16 significant bits of x.
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
    const float ra = a - __builtin_bit_cast(float, hi << 16);
    const float rb = b - __builtin_bit_cast(float, hi & 0xFFFF0000u);
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ra, rb}, bf16x2));
}
// Three terms: hi + mid + lo carries 24 significant bits (all of a float32 unless it is subnormal).
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
#define BF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// SPLIT form of conv3x3_chunk: kernels and activations as three bf16 terms each, the six products whose weight is
// at least 2^-16 of the leading one (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi), small ones first, on
// v_mfma_f32_32x32x16_bf16: 108 32-cycle MFMAs a chunk instead of 144 64-cycle ones, 54 stream groups (pack.cpp).
// Plane layout: position p (10 x 10, zero halo) is 64 bytes = four 16-byte chunks; chunk c = 2 s + h holds the eight
// channels of k-step s, lane half h, i.e. accumulator registers 8 h .. 8 h + 7 of the writing lane half s, and sits at
// chunk (c + padded rank) & 3 of its position: the four lane groups of a ds_read_b128 (MI355X_MICROARCH.md, LDS) hold
// one run of four files from each of four consecutive ranks, so every shifted read is conflict-free.
// xa[j]: this lane's byte address of chunk (j + h + rank) & 3 of the position one row and one column back.
template <class WStream>
__device__ __forceinline__ void conv3x3_chunk_split(WStream& ws, int g0, const char* const (&xa)[4], f32x16& acc) {
    auto load_b = [&](int idx, u32x4 (&b)[3]) {
        const int tap = idx >> 1, s = idx & 1, tr = tap / 3, tf = tap % 3;
        const char* p = xa[(2 * s + tr) & 3] + (tr * XP_W + tf) * XS_POS_BYTES;
#pragma unroll
        for (int term = 0; term < 3; ++term) b[term] = *reinterpret_cast<const u32x4*>(p + term * XS_PLANE_BYTES);
    };
    u32x4 b[2][3];
    load_b(0, b[0]);
#pragma unroll
    for (int idx = 0; idx < 18; ++idx) {
        if (idx + 1 < 18) load_b(idx + 1, b[(idx + 1) & 1]);
        const f32x4 wh = ws.take(g0 + 3 * idx), wm = ws.take(g0 + 3 * idx + 1), wl = ws.take(g0 + 3 * idx + 2);
        const u32x4 &xh = b[idx & 1][0], &xm = b[idx & 1][1], &xl = b[idx & 1][2];
#ifndef BF_ABL_5PROD   // timing ablation (wrong scores)
        acc = BF_MFMA16(wl, xh, acc);
#endif
#ifndef BF_ABL_4PROD
        acc = BF_MFMA16(wh, xl, acc);
#endif
        acc = BF_MFMA16(wm, xm, acc);
        acc = BF_MFMA16(wm, xh, acc);
        acc = BF_MFMA16(wh, xm, acc);
        acc = BF_MFMA16(wh, xh, acc);
    }
}

// MODE 0: scores; 1: scores + every intermediate activation (debug); 2: features -- x9, the
// activations after the six 3x3 layers, as float32 [board][32][64] into `scores`, no head (the
// move-scoring network, whose 1x1 logit layer is applied per requested move by k_policy_scores).
// SPLIT (BF_PRECISION_F32_SPLIT, include/batchfirst.h): the 1x1 pair, 68 % of the network's FLOPs, runs on
// v_mfma_f32_32x32x16_bf16 with both operands as TWO bf16 terms (16 significant bits) and all four products,
// float32 sums: 92 32-cycle MFMAs a hidden tile instead of 183 64-cycle ones (same stream shape, 30 + 16 groups a
// hidden tile, other contents: pack.cpp); and the six 3x3 layers with both operands as THREE bf16 terms
// (conv3x3_chunk_split).  Stage 1 and the head are this kernel unchanged.
template <int MODE, int SPLIT>
__global__ __launch_bounds__(THREADS, 4 / BPB) void k_eval_f32(const uint8_t* __restrict__ mailbox, unsigned n_boards,
                                                          const f32x4* __restrict__ wstream,
                                                          const float* __restrict__ bias,
                                                          const f32x4* __restrict__ headw,
                                                          float* __restrict__ scores, float* __restrict__ dbg) {
    constexpr bool DBG = MODE == 1;
    __shared__ __attribute__((aligned(16))) LdsT<SPLIT> lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63, hl = lane >> 5, ln = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bib = wave >> 1, half = wave & 1;           // board in block, half of the board
    const int btid = tid & 127;                           // thread index within the board
    const int sq = 32 * half + ln, rank = sq >> 3, file = sq & 7;
    // a trailing odd board is paired with a copy of itself; the copy's score is not stored
    const unsigned board_raw = blockIdx.x * BPB + bib;
    const bool live = board_raw < n_boards;
    const unsigned board = live ? board_raw : n_boards - 1;

#ifdef BF_STAMPS
    long long stamps[12];
#endif
    BF_STAMP(0);
    WStreamT<slots_of(SPLIT), chunk_of(SPLIT)> ws;
    ws.prime(wstream, &lds.wring[0][0][0], lane, wave);

    // ---- LDS set-up: empty mailbox (0xFF everywhere), zero activation planes ------------
    if (btid < MB_BYTES / 4) reinterpret_cast<unsigned*>(lds.mb[bib])[btid] = 0xFFFFFFFFu;
    if (SPLIT != 2) {   // (level 2 zeroes its planes after the 1x1 pair, when the x1 region they alias is free)
        f32x4* z = reinterpret_cast<f32x4*>(&lds.xbuf[bib][0][0]);
        for (int i = btid; i < (int)(sizeof(lds.xbuf[0]) / 16); i += 128) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (btid < 64) {
        const unsigned v = mailbox[(size_t)board * kMailboxBytes + (btid >> 1)];
        const unsigned c = (btid & 1) ? (v >> 4) : (v & 15u);
        lds.mb[bib][((btid >> 3) + 7) * MB_W + (btid & 7) + 7] = (unsigned char)(c == 15u ? 0xFFu : c);
    }
    ws.start();  // its barrier also orders the mailbox writes before stage 1 reads them

    BF_STAMP(1);
    float* dbg_board = DBG ? dbg + (size_t)board * kDebugFloatsPerBoard : nullptr;

    // ---- stage 1: input convs -> x1 (118 registers) -------------------------------------
    float x1[kX1Regs];
    {
        const unsigned char* mbc = &lds.mb[bib][(rank + 7) * MB_W + file + 7];
        input_conv<0>(ws, mbc, hl, bias, x1);
        input_conv<1>(ws, mbc, hl, bias, x1);
        input_conv<2>(ws, mbc, hl, bias, x1);
        input_conv<3>(ws, mbc, hl, bias, x1);
        input_conv<4>(ws, mbc, hl, bias, x1);
        input_conv<5>(ws, mbc, hl, bias, x1);
        input_conv<6>(ws, mbc, hl, bias, x1);
        input_conv<7>(ws, mbc, hl, bias, x1);
        input_conv<8>(ws, mbc, hl, bias, x1);
    }
    if (DBG) {
#pragma unroll
        for (int r = 0; r < kX1Regs; ++r) dbg_board[(2 * r + hl) * 64 + sq] = x1[r];
    }

    BF_STAMP(2);
    // ---- stage 2: 1x1 236->512 (+ReLU) chained into 1x1 512->128 -------------------------
    f32x16 acc3[4];
#pragma unroll
    for (int c3 = 0; c3 < 4; ++c3)
#pragma unroll
        for (int t = 0; t < 16; ++t)
            acc3[c3][t] = pick(hl, bias[kBiasX3 + 32 * c3 + rowmap(t, 0)], bias[kBiasX3 + 32 * c3 + rowmap(t, 1)]);

    if (SPLIT == 2) {
        // BF_PRECISION_F32_SPLIT3: three terms a side, the six products of conv3x3_chunk_split.  x1's hi and mid terms
        // stay in registers as in level 1; the third term goes to this wave's own LDS region as ready-made B
        // fragments, one ds_read_b128 a k-step and hidden tile (120 + 60 registers would leave one wave per SIMD).
        unsigned xs[2][60];
        u32x4* x1lo = reinterpret_cast<u32x4*>(&lds.xbuf[bib][0][half * X1LO_WAVE_FLOATS]) + lane;
#pragma unroll
        for (int s = 0; s < 15; ++s) {
            u32x4 lo4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = 4 * s + j;
                const float a = 2 * d < kX1Regs ? x1[2 * d] : (2 * d == kX1Regs ? pick(hl, 1.0f, 0.0f) : 0.0f);
                const float b = 2 * d + 1 < kX1Regs ? x1[2 * d + 1] : 0.0f;
                unsigned l_;
                split3(a, b, xs[0][d], xs[1][d], l_);
                lo4[j] = l_;
            }
            x1lo[s * 64] = lo4;
        }
        u32x4 xl[2];
        xl[0] = x1lo[0];
#pragma unroll 1
        for (int it = 0; it < kC2Tiles / C2_UNROLL; ++it) {
#pragma unroll
            for (int u = 0; u < C2_UNROLL; ++u) {
                const int g0 = kInputGroups + u * kPerC2GroupsSplit3;
                f32x16 acc2 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 15; ++s) {
                    xl[(s + 1) & 1] = x1lo[((s + 1) % 15) * 64];   // (the last one is the next tile's first)
                    const f32x4 wh = ws.take(g0 + 3 * s), wm = ws.take(g0 + 3 * s + 1), wl = ws.take(g0 + 3 * s + 2);
                    const u32x4 xh = u32x4{xs[0][4 * s], xs[0][4 * s + 1], xs[0][4 * s + 2], xs[0][4 * s + 3]};
                    const u32x4 xm = u32x4{xs[1][4 * s], xs[1][4 * s + 1], xs[1][4 * s + 2], xs[1][4 * s + 3]};
                    acc2 = BF_MFMA16(wl, xh, acc2);
                    acc2 = BF_MFMA16(wh, xl[s & 1], acc2);
                    acc2 = BF_MFMA16(wm, xm, acc2);
                    acc2 = BF_MFMA16(wm, xh, acc2);
                    acc2 = BF_MFMA16(wh, xm, acc2);
                    acc2 = BF_MFMA16(wh, xh, acc2);
                }
                // 15 is odd: after a tile the fragment for s = 0 sits in xl[1]; keep the parity the loop expects
                xl[0] = xl[1];
                unsigned hs[3][8];
#pragma unroll
                for (int d = 0; d < 8; ++d) split3(relu(acc2[2 * d]), relu(acc2[2 * d + 1]), hs[0][d], hs[1][d], hs[2][d]);
#pragma unroll
                for (int c3 = 0; c3 < 4; ++c3)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        const int g = g0 + 45 + (c3 * 2 + v) * 3;
                        const f32x4 wh = ws.take(g), wm = ws.take(g + 1), wl = ws.take(g + 2);
                        const u32x4 hh = u32x4{hs[0][4 * v], hs[0][4 * v + 1], hs[0][4 * v + 2], hs[0][4 * v + 3]};
                        const u32x4 hm = u32x4{hs[1][4 * v], hs[1][4 * v + 1], hs[1][4 * v + 2], hs[1][4 * v + 3]};
                        const u32x4 hl_ = u32x4{hs[2][4 * v], hs[2][4 * v + 1], hs[2][4 * v + 2], hs[2][4 * v + 3]};
                        acc3[c3] = BF_MFMA16(wl, hh, acc3[c3]);
                        acc3[c3] = BF_MFMA16(wh, hl_, acc3[c3]);
                        acc3[c3] = BF_MFMA16(wm, hm, acc3[c3]);
                        acc3[c3] = BF_MFMA16(wm, hh, acc3[c3]);
                        acc3[c3] = BF_MFMA16(wh, hm, acc3[c3]);
                        acc3[c3] = BF_MFMA16(wh, hh, acc3[c3]);
                    }
            }
        }
        // the term planes of the 3x3 layers take the x1 region's place: both halves of the board are done with it
        __syncthreads();
        {
            f32x4* z = reinterpret_cast<f32x4*>(&lds.xbuf[bib][0][0]);
            for (int i = btid; i < XS_FLOATS / 4; i += 128) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
    } else if (SPLIT) {
        // x1 as two bf16 terms, packed for the B operand of v_mfma_f32_32x32x16_bf16: k-step s takes registers
        // 8s .. 8s+7 (element j of lane half h = channel 16s + 2j + h); register 118 of half 0 is the all-ones slot
        // that carries the hidden layer's shift, the rest of the padding is zero.
        unsigned xs[2][60];
#pragma unroll
        for (int d = 0; d < 60; ++d) {
            const float a = 2 * d < kX1Regs ? x1[2 * d] : (2 * d == kX1Regs ? pick(hl, 1.0f, 0.0f) : 0.0f);
            const float b = 2 * d + 1 < kX1Regs ? x1[2 * d + 1] : 0.0f;
            split2(a, b, xs[0][d], xs[1][d]);
        }
#pragma unroll 1
        for (int it = 0; it < kC2Tiles / C2_UNROLL; ++it) {
#pragma unroll
            for (int u = 0; u < C2_UNROLL; ++u) {
                const int g0 = kInputGroups + u * kPerC2Groups;
                f32x16 acc2 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 15; ++s) {
                    const f32x4 wh = ws.take(g0 + 2 * s), wm = ws.take(g0 + 2 * s + 1);
                    const u32x4 xh = u32x4{xs[0][4 * s], xs[0][4 * s + 1], xs[0][4 * s + 2], xs[0][4 * s + 3]};
                    const u32x4 xm = u32x4{xs[1][4 * s], xs[1][4 * s + 1], xs[1][4 * s + 2], xs[1][4 * s + 3]};
                    acc2 = BF_MFMA16(wm, xm, acc2);
                    acc2 = BF_MFMA16(wm, xh, acc2);
                    acc2 = BF_MFMA16(wh, xm, acc2);
                    acc2 = BF_MFMA16(wh, xh, acc2);
                }
                // ReLU, then the hidden tile as two bf16 terms: k-step v takes registers 8v .. 8v+7
                unsigned hs[2][8];
#pragma unroll
                for (int d = 0; d < 8; ++d) split2(relu(acc2[2 * d]), relu(acc2[2 * d + 1]), hs[0][d], hs[1][d]);
#pragma unroll
                for (int c3 = 0; c3 < 4; ++c3)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        const f32x4 wh = ws.take(g0 + kGemm1Groups + c3 * 4 + 2 * v), wm = ws.take(g0 + kGemm1Groups + c3 * 4 + 2 * v + 1);
                        const u32x4 hh = u32x4{hs[0][4 * v], hs[0][4 * v + 1], hs[0][4 * v + 2], hs[0][4 * v + 3]};
                        const u32x4 hm = u32x4{hs[1][4 * v], hs[1][4 * v + 1], hs[1][4 * v + 2], hs[1][4 * v + 3]};
                        acc3[c3] = BF_MFMA16(wm, hm, acc3[c3]);
                        acc3[c3] = BF_MFMA16(wm, hh, acc3[c3]);
                        acc3[c3] = BF_MFMA16(wh, hm, acc3[c3]);
                        acc3[c3] = BF_MFMA16(wh, hh, acc3[c3]);
                    }
            }
        }
    } else {
#pragma unroll 1
    for (int it = 0; it < kC2Tiles / C2_UNROLL; ++it) {
#pragma unroll
        for (int u = 0; u < C2_UNROLL; ++u) {
            const int c2 = it * C2_UNROLL + u;
            const int g0 = kInputGroups + u * kPerC2Groups;
            (void)c2;
            if (it == 1 && u == 0) BF_STAMP(8);
            if (it == 1 && u == 1) BF_STAMP(10);
            f32x16 acc2 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < kGemm1Groups; ++g) {
                const f32x4 w = ws.take(g0 + g);
                acc2 = BF_MFMA(w.x, x1[4 * g + 0], acc2);
                acc2 = BF_MFMA(w.y, x1[4 * g + 1], acc2);
                if (4 * g + 2 < kX1Regs) {
                    acc2 = BF_MFMA(w.z, x1[4 * g + 2], acc2);
                    acc2 = BF_MFMA(w.w, x1[4 * g + 3], acc2);
                } else {
                    // k-step 118 carries the folded batch-norm shift: A = shift per row, B = 1
                    acc2 = BF_MFMA(w.z, 1.0f, acc2);
                }
            }
            if (it == 1 && u == 0) BF_STAMP(9);
#pragma unroll
            for (int c3 = 0; c3 < 4; ++c3)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    const f32x4 w = ws.take(g0 + kGemm1Groups + c3 * 4 + tq);
                    if (c3 == 0) {  // ReLU of the four rows this group needs, just ahead of their MFMAs
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc2[4 * tq + j] = relu(acc2[4 * tq + j]);
                    }
                    acc3[c3] = BF_MFMA(w.x, acc2[4 * tq + 0], acc3[c3]);
                    acc3[c3] = BF_MFMA(w.y, acc2[4 * tq + 1], acc3[c3]);
                    acc3[c3] = BF_MFMA(w.z, acc2[4 * tq + 2], acc3[c3]);
                    acc3[c3] = BF_MFMA(w.w, acc2[4 * tq + 3], acc3[c3]);
                }
        }
    }
    }
    if (DBG) {
#pragma unroll
        for (int c3 = 0; c3 < 4; ++c3)
#pragma unroll
            for (int t = 0; t < 16; ++t)
                dbg_board[64 * 236 + (32 * c3 + rowmap(t, hl)) * 64 + sq] = relu(acc3[c3][t]);
    }

    BF_STAMP(3);
    // ---- stages 3 and 4: the six 3x3 layers through LDS -----------------------------------
    constexpr int GC = kInputGroups;  // static chunk phase after the loop (the loop preserves it)
    f32x16 acc;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = pick(hl, bias[kBiasX4 + rowmap(t, 0)], bias[kBiasX4 + rowmap(t, 1)]);
    if (SPLIT) {
        char* planes = reinterpret_cast<char*>(&lds.xbuf[bib][0][0]);
        const char* xa[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[j] = planes + (rank * XP_W + file) * XS_POS_BYTES + ((j + hl + rank) & 3) * 16;
        // this lane's square: registers 8 cq .. 8 cq + 7 are chunk 2 hl + cq, padded rank = rank + 1
        char* wa[2];
#pragma unroll
        for (int cq = 0; cq < 2; ++cq)
            wa[cq] = planes + ((rank + 1) * XP_W + file + 1) * XS_POS_BYTES + ((2 * hl + cq + rank + 1) & 3) * 16;
        auto store_planes = [&](const f32x16& v) {   // ReLU, three terms, 2 x 3 16-byte stores
#pragma unroll
            for (int cq = 0; cq < 2; ++cq) {
                u32x4 th, tm, tl;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    unsigned h_, m_, l_;
                    split3(relu(v[8 * cq + 2 * d]), relu(v[8 * cq + 2 * d + 1]), h_, m_, l_);
                    th[d] = h_; tm[d] = m_; tl[d] = l_;
                }
                *reinterpret_cast<u32x4*>(wa[cq]) = th;
                *reinterpret_cast<u32x4*>(wa[cq] + XS_PLANE_BYTES) = tm;
                *reinterpret_cast<u32x4*>(wa[cq] + 2 * XS_PLANE_BYTES) = tl;
            }
        };
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#ifndef BF_ABL_NOBAR2   // timing ablation (races)
            if (q > 0) __syncthreads();   // the other half's reads of the chunk before this one are done
#endif
            store_planes(acc3[q]);
            __syncthreads();
            conv3x3_chunk_split(ws, GC + q * kConvSplitChunkGroups, xa, acc);
        }
        BF_STAMP(4);
#pragma unroll
        for (int layer = 0; layer < 5; ++layer) {
            if (DBG) {
#pragma unroll
                for (int t = 0; t < 16; ++t) dbg_board[64 * (236 + 128 + 32 * layer) + rowmap(t, hl) * 64 + sq] = relu(acc[t]);
            }
#ifndef BF_ABL_NOBAR2
            __syncthreads();
#endif
            store_planes(acc);
            __syncthreads();
            const float* bl = bias + kBiasX4 + 32 * (layer + 1);
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = pick(hl, bl[rowmap(t, 0)], bl[rowmap(t, 1)]);
            conv3x3_chunk_split(ws, GC + (4 + layer) * kConvSplitChunkGroups, xa, acc);
        }
    } else {
    const int wpos = (rank + 1) * XP_W + file + 1;           // where this lane's square lives
    const int rpos = hl * XP_PLANE + rank * XP_W + file;     // one row and column back, plane hl
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float* xb = lds.xbuf[bib][q & 1];
#pragma unroll
        for (int t = 0; t < 16; ++t) xb[rowmap(t, hl) * XP_PLANE + wpos] = relu(acc3[q][t]);
        __syncthreads();
        conv3x3_chunk(ws, GC + q * 36, xb + rpos, acc);
    }
    BF_STAMP(4);
#pragma unroll
    for (int layer = 0; layer < 5; ++layer) {
        float* xb = lds.xbuf[bib][layer & 1];  // continues the 0,1,0,1 alternation of the four chunks above
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float v = relu(acc[t]);
            xb[rowmap(t, hl) * XP_PLANE + wpos] = v;
            if (DBG) dbg_board[64 * (236 + 128 + 32 * layer) + rowmap(t, hl) * 64 + sq] = v;
        }
        __syncthreads();
        const float* bl = bias + kBiasX4 + 32 * (layer + 1);
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = pick(hl, bl[rowmap(t, 0)], bl[rowmap(t, 1)]);
        conv3x3_chunk(ws, GC + 144 + layer * 36, xb + rpos, acc);
    }
    }

    BF_STAMP(5);
    // The stream's prefetch runs two chunks past the last group used; the compiler does not know of the asm
    // LDS-DMA, so nothing else waits for it: no piece may still be on its way into this workgroup's LDS when
    // the workgroup ends and the CU hands the allocation to the next one.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE == 2) {
        if (live) {
            float* out = scores + (size_t)board * 2048;
#pragma unroll
            for (int t = 0; t < 16; ++t) out[rowmap(t, hl) * 64 + sq] = relu(acc[t]);
        }
        return;
    }
    // ---- stage 5: 8x8 valid conv (2048 -> 16), ReLU, logit -----------------------------
    float part[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) part[co] = 0.0f;
    const f32x4* hw = headw + (size_t)half * 64 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float v = relu(acc[t]);
        if (DBG) dbg_board[64 * (236 + 128 + 32 * 5) + rowmap(t, hl) * 64 + sq] = v;
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
#ifdef BF_ABL_NOHEAD
            const f32x4 w = f32x4{0.5f, 0.25f, 0.125f, 1.0f};
#else
            const f32x4 w = hw[(t * 4 + cq) * 64];
#endif
            part[4 * cq + 0] = fmaf(v, w.x, part[4 * cq + 0]);
            part[4 * cq + 1] = fmaf(v, w.y, part[4 * cq + 1]);
            part[4 * cq + 2] = fmaf(v, w.z, part[4 * cq + 2]);
            part[4 * cq + 3] = fmaf(v, w.w, part[4 * cq + 3]);
        }
    }
    // The last 3x3 layer read buffer 0 (layer 4); buffer 1 was last read one barrier earlier,
    // so its space can hold the reduction scratch (this block is done with the planes).
    // (SPLIT has one buffer: a barrier first, so that the other half's reads of the last layer are done.)
    if (SPLIT) __syncthreads();
    float* red = lds.xbuf[bib][SPLIT ? 0 : 1];
#pragma unroll
    for (int co = 0; co < 16; ++co) red[btid * RED_STRIDE + co] = part[co];
    __syncthreads();
    if (btid < 64) {
        const int co = btid & 15, seg = btid >> 4;
        float s = 0.0f;
        for (int j = 0; j < 32; ++j) s += red[(seg * 32 + j) * RED_STRIDE + co];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float y = fmaxf(s + bias[kBiasX10 + co], 0.0f);
        if (DBG && btid < 16) dbg_board[64 * (236 + 128 + 32 * 6) + co] = y;
        float v = y * bias[kLogitW + co];  // logit_layer: no bias, no batch-norm (evaluation_ann.py:35-44)
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        if (btid == 0 && live) scores[board] = v;
    }
#ifdef BF_STAMPS
    BF_STAMP(6);
    if (!DBG && dbg && lane == 0) {
        long long* out = reinterpret_cast<long long*>(dbg) + ((size_t)blockIdx.x * WAVES + wave) * 16;
        for (int i = 0; i < 7; ++i) out[i] = stamps[i];
        out[7] = stamps[9] - stamps[8];   // GEMM1 of one tile: 119 MFMAs
        out[8] = stamps[10] - stamps[9];  // ReLU + GEMM2 of that tile: 64 MFMAs
    }
#endif
}

// One thread per requested move: logit[filter][from] = sum_c W[filter][c] * x9[board][c][from]
// (logit_layer + reshape + gather_nd, move_evaluation_ann.py:35-73), fixed summation order.
__global__ __launch_bounds__(256) void k_policy_scores(const float* __restrict__ x9, const float* __restrict__ logit_w,
                                                       const unsigned* __restrict__ board_index,
                                                       const uint8_t* __restrict__ from_sq,
                                                       const uint8_t* __restrict__ move_filter, unsigned n_moves,
                                                       unsigned n_boards, int* __restrict__ err,
                                                       float* __restrict__ out) {
    const unsigned m = blockIdx.x * 256 + threadIdx.x;
    if (m >= n_moves) return;
    unsigned b = board_index[m], sq = from_sq[m], f = move_filter[m];
    if (b >= n_boards || sq >= 64u || f >= (unsigned)kPolicyFilters) {  // reads stay in bounds; the caller is told
        *err = 4;
        b = min(b, n_boards - 1);
        sq &= 63u;
        f = min(f, (unsigned)kPolicyFilters - 1);
    }
    const float* x = x9 + (size_t)b * 2048 + sq;
    const float* w = logit_w + (size_t)f * 32;
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 32; ++c) s = fmaf(w[c], x[c * 64], s);
    out[m] = s;
}

}  // namespace

// split: 0 = float32, 1 = BF_PRECISION_F32_SPLIT, 2 = BF_PRECISION_F32_SPLIT3
template <int MODE>
static hipError_t launch_mode(int split, unsigned blocks, hipStream_t stream, const uint8_t* d_mailbox, unsigned n, const f32x4* ws,
                              const float* bias, const f32x4* head, float* out, float* dbg) {
#ifdef BF_ONLY_SPLIT_SCORES   // experiment builds (tools/build_variant.sh): one instantiation, k_eval_f32<0, BF_ONLY_SPLIT_VALUE>
#ifndef BF_ONLY_SPLIT_VALUE
#define BF_ONLY_SPLIT_VALUE 1
#endif
    if (MODE != 0 || split != BF_ONLY_SPLIT_VALUE) return hipErrorNotSupported;
    hipLaunchKernelGGL((k_eval_f32<0, BF_ONLY_SPLIT_VALUE>), dim3(blocks), dim3(THREADS), 0, stream, d_mailbox, n, ws, bias, head, out, dbg);
#else
    if (split == 2)
        hipLaunchKernelGGL((k_eval_f32<MODE, 2>), dim3(blocks), dim3(THREADS), 0, stream, d_mailbox, n, ws, bias, head, out, dbg);
    else if (split == 1)
        hipLaunchKernelGGL((k_eval_f32<MODE, 1>), dim3(blocks), dim3(THREADS), 0, stream, d_mailbox, n, ws, bias, head, out, dbg);
    else
        hipLaunchKernelGGL((k_eval_f32<MODE, 0>), dim3(blocks), dim3(THREADS), 0, stream, d_mailbox, n, ws, bias, head, out, dbg);
#endif
    return hipGetLastError();
}

hipError_t launch_policy_features_f32(const uint8_t* d_mailbox, size_t n, const float* d_packed, float* d_x9,
                                      hipStream_t stream, int split) {
    if (n == 0) return hipSuccess;
    return launch_mode<2>(split, (unsigned)((n + BPB - 1) / BPB), stream, d_mailbox, (unsigned)n,
                          reinterpret_cast<const f32x4*>(d_packed + kOffStream), d_packed + kOffBias,
                          reinterpret_cast<const f32x4*>(d_packed + kOffHead), d_x9, nullptr);
}

hipError_t launch_policy_scores(const float* d_x9, const float* d_logit_w, const unsigned* d_board_index,
                                const uint8_t* d_from, const uint8_t* d_filter, size_t n_moves, size_t n_boards,
                                int* d_err, float* d_out, hipStream_t stream) {
    if (n_moves == 0 || n_boards == 0) return hipSuccess;
    hipLaunchKernelGGL(k_policy_scores, dim3((unsigned)((n_moves + 255) / 256)), dim3(256), 0, stream, d_x9,
                       d_logit_w, d_board_index, d_from, d_filter, (unsigned)n_moves, (unsigned)n_boards,
                       d_err, d_out);
    return hipGetLastError();
}

hipError_t launch_eval_f32(const uint8_t* d_mailbox, size_t n, const float* d_packed, float* d_scores,
                           float* d_debug, hipStream_t stream, int split) {
    if (n == 0) return hipSuccess;
    const f32x4* ws = reinterpret_cast<const f32x4*>(d_packed + kOffStream);
    const float* bias = d_packed + kOffBias;
    const f32x4* head = reinterpret_cast<const f32x4*>(d_packed + kOffHead);
    const unsigned blocks = (unsigned)((n + BPB - 1) / BPB);
#ifdef BF_STAMPS   // the debug pointer carries the stamps of the scores kernel
    return launch_mode<0>(split, blocks, stream, d_mailbox, (unsigned)n, ws, bias, head, d_scores, d_debug);
#else
    if (d_debug) return launch_mode<1>(split, blocks, stream, d_mailbox, (unsigned)n, ws, bias, head, d_scores, d_debug);
    return launch_mode<0>(split, blocks, stream, d_mailbox, (unsigned)n, ws, bias, head, d_scores, d_debug);
#endif
}

}  // namespace bf
