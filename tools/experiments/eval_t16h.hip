// EXPERIMENT, NOT PART OF THE LIBRARY (round 3; results: profiles/r3/t16h/).  Built in round 3 as
// batch_first_amd/csrc/eval_t16h.hip behind BF_T16_KERNEL=h, bit-identical to k_eval_t16 on 65 536 boards and 6 %
// SLOWER (1.849 against 1.740 ms interleaved): two waves per SIMD in lockstep on one weight ring do not raise the
// matrix pipe's share (stage stamps equal to k_eval_t16's within 5 %), and the ring's chunk barriers cost 8 % with
// eight waves.  Kept as the record of what was measured; to rebuild it, copy it next to eval_t16.hip, add it to the
// Makefile's object lists and declare launch_t16_half in bf_kernels.h.
//
// The 16x16x32 bf16 network with TWO WAVES PER BOARD: eight waves a workgroup (two per SIMD), four boards a
// workgroup, every wave two of its board's four 16-square column tiles (wave w: board w & 3, tiles 2 (w >> 2)
// and 2 (w >> 2) + 1, i.e. ranks 0-3 or 4-7).
//
// Why: eval_t16.hip gives a wave a whole board, which costs ~490 registers, so one wave per SIMD,
// and an in-order wave has nothing to put into its own stalls: every LDS-DMA piece it issues, every chunk barrier,
// every pack -> LDS write -> shifted read chain of the 3x3 layers and all of stage 1's operand bookkeeping show up
// as idle matrix-pipe cycles (matrix pipe busy 0.66 of the kernel, profiles/r2).  Half a board fits 256 registers:
// two waves share a SIMD, and one's MFMAs run in the other's gaps.  The price: a weight fragment read from LDS
// feeds two MFMAs instead of four (128 B/clk of the LDS array's 256 in the 1x1 pair), the 3x3 layers exchange tiles
// between the two waves of a board through shared planes with a workgroup barrier per 32-channel chunk, and stage 1
// computes every k-step for every tile (the tile index is a run-time value, as in eval_t16s.hip).
//
// Weight image, fragment order, accumulation order and every rounding are those of eval_t16.hip and eval_t16s.hip
// (bf_layout_t16.h): scores and move-scoring features are bit-identical to theirs, which tests/test_gpu_bf16.py
// checks for every entry point; a context may therefore pick its kernel by call size.
//
// Reference graph replaced: as eval_t16.hip (ann_creation_helper.py:192-252, evaluation_ann.py:268, :35-46).
#include <hip/hip_runtime.h>

#include "bf_device.h"
#include "bf_kernels.h"
#include "bf_layout_t16.h"

#define T16_WAVES 8
#ifndef T16_CHUNK
#define T16_CHUNK 8      // 8 groups a chunk, one piece per wave and chunk: the ring is 32 KiB
#endif
#include "t16_common.h"

namespace bf {

namespace t16 {

namespace {

constexpr int TPW = 2;   // column tiles per wave

struct LdsH {
    f32x4 wring[SLOTS][CHUNK][64];                                       // shared A-fragment ring
    __attribute__((aligned(16))) unsigned char image[BPB][SG_BYTES];     // one-hot image of every board (t16_common.h)
    __attribute__((aligned(16))) unsigned char plane[BPB][2][PLANE_BYTES];   // zero-haloed [10 x 10][32 channels] bf16
    u32x4 onehot[16][2];
    __attribute__((aligned(16))) float bias3[6 * 32];
    __attribute__((aligned(16))) float bias2[128];
    f32x4 hpart[BPB][64];                                                // stage 5: partial sums of square groups 0..3
    __attribute__((aligned(16))) unsigned char next_mailbox[WAVES][256]; // every wave prefetches its own copy
};
static_assert(sizeof(LdsH) <= 160 * 1024, "one workgroup per compute unit");

// MODE 0: scores; 2: features (x9 as float32 [board][32][64] into `out`); + 4: node records instead of mailboxes.
template <int MODE_>
__global__ __launch_bounds__(THREADS) void k_eval_t16h(const uint8_t* __restrict__ src, unsigned n_boards,
                                                        const f32x4* __restrict__ wstream, const float* __restrict__ bias,
                                                        const f32x4* __restrict__ headw, float* __restrict__ out,
                                                        unsigned src_stride, float* __restrict__ dbg) {
    constexpr int MODE = MODE_ & 3;
    constexpr bool RECS = (MODE_ & 4) != 0;
    constexpr int SRC_DWORDS = RECS ? 22 : 8;
    __shared__ __attribute__((aligned(16))) LdsH lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63, c16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bslot = wave & 3, half = wave >> 2;            // this wave's board of the quad, and which half of it
    const int rk = c16 >> 3, file = c16 & 7;
    const unsigned n_quads = (n_boards + BPB - 1) / BPB;
    typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
    typedef const __attribute__((address_space(3))) u32x4* lds_entry;

#ifdef BF_STAMPS
    unsigned stamps[16] = {};
#endif
    WStream ws;
    ws.prime(wstream, &lds.wring[0][0][0], lane, wave);

    // ---- once per workgroup ---------------------------------------------------------------------------
    if (tid < 32) {
        const unsigned c = tid >> 1, hh = tid & 1, j = c & 7u;
        const unsigned one = (c != 15u && (c >> 3) == hh) ? (0x3F80u << ((j & 1u) * 16u)) : 0u;
        u32x4 e = u32x4{0u, 0u, 0u, 0u};
        e.x |= ((j >> 1) == 0u) ? one : 0u;
        e.y |= ((j >> 1) == 1u) ? one : 0u;
        e.z |= ((j >> 1) == 2u) ? one : 0u;
        e.w |= ((j >> 1) == 3u) ? one : 0u;
        lds.onehot[c][hh] = e;
    }
    const float shift3 = tid < 6 * 32 ? bias[kBiasX4 + tid] : 0.0f;
    const float shift2 = tid < 128 ? bias[kBiasX3 + tid] : 0.0f;
    {
        const unsigned b0 = blockIdx.x * BPB + bslot;
        const unsigned bc = b0 < n_boards ? b0 : n_boards - 1;
        if (lane < SRC_DWORDS)
            reinterpret_cast<unsigned*>(lds.next_mailbox[wave])[lane] =
                reinterpret_cast<const unsigned*>(src + (size_t)bc * src_stride)[lane];
    }
    const float head_shift = MODE != 2 ? bias[kBiasX10 + (tid & 15)] : 0.0f;
    const float head_logit = MODE != 2 ? bias[kLogitW + (tid & 15)] : 0.0f;
    for (int i = tid; i < BPB * SG_BYTES / 16; i += THREADS) reinterpret_cast<u32x4*>(&lds.image[0][0])[i] = u32x4{0u, 0u, 0u, 0u};
    for (int i = tid; i < BPB * 2 * PLANE_BYTES / 16; i += THREADS) reinterpret_cast<u32x4*>(&lds.plane[0][0][0])[i] = u32x4{0u, 0u, 0u, 0u};
    if (tid < 6 * 32) lds.bias3[tid] = shift3;
    if (tid < 128) lds.bias2[tid] = shift2;
    lds_order();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ws.start();   // (its first chunk barrier also orders the zeroing above before any wave's first use)

    // lane constants: this lane's square of the wave's FIRST tile in its board's image and planes; the second tile is
    // two ranks up (immediate offsets)
    const int my_rank = 4 * half + rk;
    unsigned char* const image = lds.image[bslot];
    unsigned char* const plane0 = lds.plane[bslot][0];
    const unsigned grid0 = (unsigned)(uintptr_t)(lds_bytes)image + (my_rank * SG_COLS + file) * SG_ENTRY + (q & 1) * 16;
    const int waddr = ((my_rank + 1) * XP_W + file + 1) * POS_BYTES + (((q + file + 1) & 3) << 4);
    int raddr[3];   // shifted reads df = -1, 0, +1, one padded row above the square
#pragma unroll
    for (int d = 0; d < 3; ++d) raddr[d] = (my_rank * XP_W + file + d) * POS_BYTES + (((q + file + d) & 3) << 4);
    // the image entry this lane writes per board: square 32 * half + (lane & 31), plane half lane >> 5
    const int wsq = 32 * half + (lane & 31), whh = lane >> 5;
    unsigned char* const my_entry = image + (((wsq >> 3) + SG_ROW0) * SG_COLS + (wsq & 7) + SG_COL0) * SG_ENTRY + whh * 16;
    const bool q2 = q >= 2;
    const unsigned ones = (q & 1) ? 0x3F800000u : 0u;

#pragma unroll 1
    for (unsigned quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
    const unsigned board_raw = quad * BPB + bslot;
    const unsigned board = board_raw < n_boards ? board_raw : n_boards - 1;   // padding boards repeat the last one
    const unsigned next_raw = (quad + gridDim.x) * BPB + bslot;
    const unsigned next_board = __builtin_amdgcn_readfirstlane(next_raw < n_boards ? next_raw : n_boards - 1);
    ws.board_start();
    BF_STAMP(0);

    // ---- this wave's thirty-two squares of its board's one-hot image -------------------------------------
    {
        unsigned c;
        if (RECS) {
            const BoardBits bits = load_board(lds.next_mailbox[wave]);
            c = (unsigned)square_code(bits, relevance_mask(bits), wsq);
        } else {
            const unsigned v = lds.next_mailbox[wave][wsq >> 1];
            c = (wsq & 1) ? (v >> 4) : (v & 15u);
        }
        *reinterpret_cast<u32x4*>(my_entry) = lds.onehot[c][whh];
    }
    lds_barrier();
    BF_STAMP(1);

    // ---- stage 1 ------------------------------------------------------------------------------------
    unsigned X[TPW][kX1Dwords];
    {
        constexpr int BQ = 3;                      // units the B operands are read ahead of their MFMAs
        u32x4 b[BQ][TPW];
        f32x4 acc[2][4][TPW];                      // [conv parity][row tile][column tile]
        unsigned base = grid0;
        auto read_b = [&](int u, int tt) -> u32x4 {
            const int uu = u < kUnits ? u : kUnits - 1;
            if (tt == 0 && u < kUnits && kS1s.new_base[uu]) {
                base = grid0 + (unsigned)(q2 ? kS1s.db[uu] : kS1s.da[uu]);
                asm volatile("" : "+v"(base));
            }
            return *(lds_entry)(uintptr_t)(base + (unsigned)(kS1s.imm[uu] + tt * 2 * SG_ROW_BYTES));
        };
        auto pack_conv = [&](int i) {
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                for (int d = kX1RegBase[i] / 2; d < (kX1RegBase[i] + kX1RegCount[i]) / 2; ++d) {
                    const X1Slot a = x1_slot(2 * d), bb = x1_slot(2 * d + 1);
                    unsigned p = pack_relu(acc[a.conv & 1][a.tile][tt][a.r], acc[bb.conv & 1][bb.tile][tt][bb.r]);
                    asm volatile("" : "+v"(p));
                    X[tt][d] = p;
                }
        };
#pragma unroll
        for (int u = 0; u < BQ; ++u)
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) b[u][tt] = read_b(u, tt);
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) b[0][tt].w |= ones;
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            const int I = kS1s.conv[u], par = I & 1;
            bf16x8 fr[4];
#pragma unroll
            for (int tile = 0; tile < 4; ++tile)
                if (tile < kRowTiles[I]) fr[tile] = ws.take(kS1s.g0[u] + tile);
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) {
                const bf16x8 bu = __builtin_bit_cast(bf16x8, b[u % BQ][tt]);
#pragma unroll
                for (int tile = 0; tile < 4; ++tile)
                    if (tile < kRowTiles[I])
                        acc[par][tile][tt] = MFMA16(fr[tile], bu, kS1s.first_of_conv[u] ? zero4() : acc[par][tile][tt]);
            }
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) b[u % BQ][tt] = read_b(u + BQ, tt);
            if (u + 1 < kUnits) {
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) b[(u + 1) % BQ][tt].w |= ones;
            }
            // the previous conv's channels, one unit after its last (its accumulators are the other parity's)
            if (kS1s.first_of_conv[u] && I > 0) pack_conv(I - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        pack_conv(kNumInputConvs - 1);
    }

    BF_STAMP(2);
    // ---- stage 2: 1x1 236->512 (+ReLU) chained into 1x1 512->128, this wave's two column tiles ------------
    f32x4 acc3[kOutTiles][TPW];
    {
        f32x4 acc2[2][2][TPW];
        unsigned H[TPW][4];
#pragma unroll
        for (int p = 0; p < kPairGroups; ++p) {
            const int g = kPairBase + p;
            const int kind = kPairOrder.kind[p], a = kPairOrder.a[p], bi = kPairOrder.b[p];
            const bf16x8 w = ws.take(g);
            if (g == kPrefetchGroup)
                ws.prefetch_mailbox(src + (size_t)next_board * src_stride, (unsigned)(uintptr_t)(lds_bytes)lds.next_mailbox[wave],
                                    (unsigned)(lane < SRC_DWORDS ? lane : SRC_DWORDS - 1) * 4u);
            if (kind == 0) {
                const int par = (a >> 1) & 1;
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) {
                    const u32x4 x = u32x4{X[tt][4 * bi], X[tt][4 * bi + 1], X[tt][4 * bi + 2], X[tt][4 * bi + 3]};
                    acc2[par][a & 1][tt] = MFMA16(w, __builtin_bit_cast(bf16x8, x), bi == 0 ? zero4() : acc2[par][a & 1][tt]);
                }
            } else {
                // the first group of an output tile starts from the tile's shift (float32-exact), read here
                f32x4 seed = zero4();
                if (a == 0) seed = *reinterpret_cast<const f32x4*>(&lds.bias2[16 * bi + 4 * q]);
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) {
                    const u32x4 h = u32x4{H[tt][0], H[tt][1], H[tt][2], H[tt][3]};
                    acc3[bi][tt] = MFMA16(w, __builtin_bit_cast(bf16x8, h), a == 0 ? seed : acc3[bi][tt]);
                }
            }
            // the hidden pair h is packed under the 236->512 groups of pair h + 1: tile tt in groups 2 + tt
            auto pack_pair = [&](int par, int tt) {
#pragma unroll
                for (int d = 0; d < 4; ++d) H[tt][d] = pack_relu(acc2[par][d >> 1][tt][2 * (d & 1)], acc2[par][d >> 1][tt][2 * (d & 1) + 1]);
            };
            if (p >= 16 && p < 16 + 24 * 15) {
                const int j = (p - 16) % 24;
                if (j >= 2 && j < 2 + TPW) pack_pair(((p - 16) / 24) & 1, j - 2);
            }
            if (p == 16 + 24 * 15 - 1) {
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt) pack_pair(1, tt);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    BF_STAMP(3);
    // ---- stages 3 and 4: the six 3x3 layers; the two waves of a board exchange tiles through the board's planes ----
    // Chunk c = 0..3: the four 32-channel input chunks of the 128->32 layer, c = 4..8: the five 32->32 layers.  Chunk c
    // reads plane c & 1; its input is written during chunk c - 1 (x3 chunks: from the accumulators of stage 2) or after
    // it (a layer's output), and a workgroup barrier stands between every chunk: the partner's rows are in place, and
    // the plane written next was last read a chunk ago.
    auto write_tile = [&](unsigned char* pl, int tt, const f32x4& a0, const f32x4& a1) {
        u32x4 v;
        v.x = pack_relu(a0[0], a0[1]);
        v.y = pack_relu(a0[2], a0[3]);
        v.z = pack_relu(a1[0], a1[1]);
        v.w = pack_relu(a1[2], a1[3]);
        lds_order();
        *reinterpret_cast<u32x4*>(pl + waddr + 2 * tt * ROW_BYTES) = v;
        lds_order();
    };
    f32x4 A[2][TPW];   // [output row tile][column tile]
    f32x4 bs[2];
    u32x4 bq[TPW][9];
    auto load_taps = [&](const unsigned char* pl) {
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                bq[tt][tap] = *reinterpret_cast<const u32x4*>(pl + raddr[tap % 3] + (2 * tt + tap / 3) * ROW_BYTES);
    };
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) write_tile(plane0, tt, acc3[0][tt], acc3[1][tt]);
    bs[0] = *reinterpret_cast<const f32x4*>(&lds.bias3[4 * q]);
    bs[1] = *reinterpret_cast<const f32x4*>(&lds.bias3[16 + 4 * q]);
    lds_barrier();
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        load_taps(plane0 + (c & 1) * PLANE_BYTES);
        const bool fresh_layer = c == 0 || c >= 4;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const bf16x8 f0 = ws.take(kConvBase + c * 18 + 2 * tap);
            const bf16x8 f1 = ws.take(kConvBase + c * 18 + 2 * tap + 1);
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) {
                const bf16x8 bn = __builtin_bit_cast(bf16x8, bq[tt][tap]);
                A[0][tt] = MFMA16(f0, bn, (fresh_layer && tap == 0) ? bs[0] : A[0][tt]);
                A[1][tt] = MFMA16(f1, bn, (fresh_layer && tap == 0) ? bs[1] : A[1][tt]);
            }
            // the next input chunk of the 128->32 layer into the other plane, under this chunk's MFMAs
            if (c < 3 && tap >= 1 && tap < 1 + TPW)
                write_tile(plane0 + ((c + 1) & 1) * PLANE_BYTES, tap - 1, acc3[2 * (c + 1)][tap - 1], acc3[2 * (c + 1) + 1][tap - 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c >= 3 && c < 8) {
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) write_tile(plane0 + ((c + 1) & 1) * PLANE_BYTES, tt, A[0][tt], A[1][tt]);
            bs[0] = *reinterpret_cast<const f32x4*>(&lds.bias3[32 * (c + 1 - 3) + 4 * q]);
            bs[1] = *reinterpret_cast<const f32x4*>(&lds.bias3[32 * (c + 1 - 3) + 16 + 4 * q]);
        }
        if (c < 8) lds_barrier();
        if (c == 3) BF_STAMP(4);
    }
    BF_STAMP(5);
    // the walk's padding groups bring the ring and the look-ahead back to their start phase
#pragma unroll
    for (int g = kStreamGroups; g < kStreamWalkGroups; ++g) (void)ws.take(g);

    if (MODE == 2) {
        if (board_raw < n_boards) {
            float* o = out + (size_t)board * 2048;
#pragma unroll
            for (int tile = 0; tile < 2; ++tile)
#pragma unroll
                for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        o[(16 * tile + 4 * r + q) * 64 + 16 * (2 * half + tt) + c16] = fmaxf(A[tile][tt][r], 0.0f);
        }
    } else {
        // ---- stage 5: 8x8 valid conv across the workgroup's boards, exactly as eval_t16.hip sums it: waves 0..3 take the
        // square groups 16 j .. 16 j + 15 (one MFMA a square, rows = boards), then the four partial sums in order ----
        // their kernels come from global memory (L2), asked for before the planes are complete
        u32x4 hk[16];
        {
            // Every wave loads (waves 4..7 the same lines as their SIMD partners 0..3, which the vector cache has just
            // fetched; a load under `if (wave < 4)` made the register allocator spill the kernels across the barrier).
            // Wave-uniform base + one lane offset made here, so that sixteen 64-bit addresses are not kept across the
            // whole board loop.
            const unsigned char* hbase = reinterpret_cast<const unsigned char*>(headw + (size_t)(wave & 3) * 16 * 64);
            unsigned hoff = (unsigned)lane * 16u;
            asm volatile("" : "+v"(hoff));
#pragma unroll
            for (int i = 0; i < 16; ++i) hk[i] = *reinterpret_cast<const u32x4*>(hbase + i * 1024 + hoff);
        }
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) write_tile(plane0 + PLANE_BYTES, tt, A[0][tt], A[1][tt]);   // x9 -> plane 1 (chunk 7 read it last)
        lds_barrier();
        if (wave < 4) {
            const unsigned char* xb = lds.plane[c16 < BPB ? c16 : 0][1];
            f32x4 hacc = zero4();
#pragma unroll
            for (int h8 = 0; h8 < 2; ++h8) {   // two batches of eight squares (sixteen operands at once would spill)
                u32x4 hb[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int sqk = wave * 16 + 8 * h8 + i;     // wave is uniform: the address is lane constant + scalar
                    const int Rk = (sqk >> 3) + 1, Fk = (sqk & 7) + 1;
                    hb[i] = *reinterpret_cast<const u32x4*>(xb + (Rk * XP_W + Fk) * POS_BYTES + (((q + Fk) & 3) << 4));
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    hacc = MFMA16(__builtin_bit_cast(bf16x8, hb[i]), __builtin_bit_cast(bf16x8, hk[8 * h8 + i]), hacc);
                __builtin_amdgcn_sched_barrier(0);
            }
            lds.hpart[wave][lane] = hacc;
        }
        lds_barrier();
        if (tid < 16 * BPB) {
            const int b = tid >> 4, co = tid & 15;
            // D[row = b][col = co] sits in lane co + 16 * (b >> 2), register b & 3
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < BPB; ++w)
                s += reinterpret_cast<const float*>(lds.hpart[w])[(co + 16 * (b >> 2)) * 4 + (b & 3)];
            const float y = fmaxf(s + head_shift, 0.0f);
            const unsigned out_raw = quad * BPB + b;
            float v = y * head_logit;   // logit_layer: no bias, no batch-norm (evaluation_ann.py:35-44)
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            if (co == 0 && out_raw < n_boards) out[out_raw] = v;
        }
    }
    BF_STAMP(6);
    // the next board's first barrier (after its image) also orders this board's last reads of the planes and of
    // hpart in front of the next board's writes
    }  // board loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's look-ahead DMA must not outlive the workgroup
#ifdef BF_STAMPS
    if (dbg && lane == 0) {
        long long* o = reinterpret_cast<long long*>(dbg) + ((size_t)blockIdx.x * WAVES + wave) * 16;
        for (int i = 0; i < 16; ++i) o[i] = (long long)stamps[i];
    }
#endif
}

hipError_t grid_blocks_h(size_t n, unsigned* blocks) {
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

template <int MODE_>
hipError_t launch_h(const uint8_t* d_src, size_t stride, size_t n, const unsigned char* d_image, float* d_out, float* d_dbg,
                    hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (n > 0xFFFFFFFFull || stride > 0xFFFFFFFFull) return hipErrorInvalidValue;
    unsigned blocks = 0;
    hipError_t e = grid_blocks_h(n, &blocks);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_eval_t16h<MODE_>, dim3(blocks), dim3(THREADS), 0, stream, d_src, (unsigned)n,
                       reinterpret_cast<const f32x4*>(d_image + kOffStream), reinterpret_cast<const float*>(d_image + kOffBias),
                       reinterpret_cast<const f32x4*>(d_image + kOffHead), d_out, (unsigned)stride, d_dbg);
    return hipGetLastError();
}

}  // namespace

}  // namespace t16

// `records`: d_src holds node records (t16_takes_records must hold), otherwise 4-bit mailboxes (stride = kMailboxBytes).
// features: x9 as float32 [n][32][64] instead of scores.  d_stamps: diagnostics builds only (BF_STAMPS), else null.
hipError_t launch_t16_half(const uint8_t* d_src, size_t stride, bool records, bool features, size_t n, const unsigned char* d_image,
                           float* d_out, float* d_stamps, hipStream_t stream) {
    if (features)
        return records ? t16::launch_h<6>(d_src, stride, n, d_image, d_out, d_stamps, stream)
                       : t16::launch_h<2>(d_src, stride, n, d_image, d_out, d_stamps, stream);
    return records ? t16::launch_h<4>(d_src, stride, n, d_image, d_out, d_stamps, stream)
                   : t16::launch_h<0>(d_src, stride, n, d_image, d_out, d_stamps, stream);
}

}  // namespace bf
