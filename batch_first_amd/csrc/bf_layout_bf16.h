// Fragment-ordered weight image of the bf16 evaluation kernel (eval_bf16.hip, pack_bf16.cpp).
//
// Arithmetic of the bf16 variant (BASELINE configs[2]; restated on the CPU by
// oracle/network.py:forward_bf16_model): batch-norm is folded as in the fp32 path
// (W' = W * inv in float32), W' is rounded to bf16 (round-to-nearest-even), every activation
// x1..x9 is rounded to bf16 after its ReLU, all sums are float32 (v_mfma_f32_32x32x16_bf16 and
// v_mfma_f32_16x16x32_bf16), the folded batch-norm shifts stay float32-exact, x10 and the logit
// are float32.
//
// Same orientation as the fp32 kernel: out^T[channel][square] = W^T[channel][k] * in^T[k][square],
// squares on lanes, one wave = 32 squares of one board.  One "group" = the A fragment of one
// MFMA = 64 lanes x 8 bf16 = 1 KiB; lane (row, h) holds k = 8h .. 8h+7 of its channel row.
//
//   stage 1   per (conv, tap, tile): K = 16 "planes": the 15 one-hot planes + a 16th that is 1.0
//             on every square, padding included.  Its weights carry the folded shift, split into
//             three bf16 terms placed in taps 0, 1, 2 (hi + mid + lo == shift exactly), so the
//             accumulators start from the literal 0.
//   x1        the 236 channels live in 120 registers per lane (channel pairs on the two lane
//             halves) = 60 packed dwords = the B operands of 15 k-steps.  Convs 4 and 5 (18
//             channels) are padded to 20 rows; their 4 padding rows are "ones" channels
//             (all-zero taps, shift 1.0): k-slots whose B element is exactly 1.0, through which
//             the 1x1 236->512 layer receives its shift, again as three exact bf16 terms.
//   1x1 pair  per 32-channel tile of the hidden layer: 15 groups (236->512), 8 groups
//             (512->128: k-step s = 0,1 x output tile 0..3), 1 padding group; stream order kGemmOrder.
//   3x3       per (32-channel input chunk, tap, k-step s): conv4 = 4 x 9 x 2, conv5..9 = 5 x 9 x 2.
//   head      8x8 valid conv as out[co][board] = sum over (sq, ci): one v_mfma_f32_16x16x32_bf16
//             per square, A[row = co][k = ci] (separate image, not part of the stream).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "bf_layout.h"

namespace bf {
namespace b16 {

constexpr int kGroupBytes = 1024;

// x1 register layout: first register of every input conv, registers per conv (even, so that
// pairs never straddle a conv)
constexpr int kX1RegBase[kNumInputConvs] = {0, 32, 48, 58, 68, 78, 88, 96, 108};
constexpr int kX1RegCount[kNumInputConvs] = {32, 16, 10, 10, 10, 10, 8, 12, 12};
constexpr int kX1Regs = 120;
constexpr int kX1Dwords = 60;
// the two all-ones registers (one per lane half each): conv 4 and conv 5, local register 9
constexpr int kOnesRegA = 68 + 9, kOnesRegB = 78 + 9;

constexpr int kInputGroups = 80;        // (tile, tap) pairs of stage 1
constexpr int kGemm1Groups = 15;
constexpr int kGemm2Groups = 8;
constexpr int kPerC2Groups = 24;        // 15 + 8 + 1 padding group: two tiles = whole chunks
constexpr int kC2Tiles = 16;
// Order of the 16 x 24 groups of the 1x1 pair in the stream = the order the kernel takes them in.  The
// pair is software-pipelined across the hidden layer's 32-channel tiles: while tile u+1's 15 groups of
// the 236->512 layer multiply (both board tiles share each fragment as it arrives), tile u's result is
// packed and its 8 groups of the 512->128 layer are interleaved, so no phase of the pair reads a
// fragment per MFMA or waits for a pack.  kind 0: 236->512 (idx = k-step 0..14); kind 1: 512->128
// (idx = 4 * k-step + output tile, 0..7); kind 2: padding group (all at the end of the section).
struct GemmOrder {
    signed char kind[16 * 24], tile[16 * 24], idx[16 * 24];
};
constexpr int kGemmStepSeq[23] = {0, 1, 2, 3, 100, 4, 101, 5, 102, 6, 103, 7, 8, 104, 9, 105, 10, 106, 11, 107, 12, 13, 14};
constexpr GemmOrder make_gemm_order() {
    GemmOrder o{};
    int p = 0;
    for (int s = 0; s < 15; ++s, ++p) { o.kind[p] = 0; o.tile[p] = 0; o.idx[p] = (signed char)s; }
    for (int u = 0; u < 15; ++u)
        for (int j = 0; j < 23; ++j, ++p) {
            const int e = kGemmStepSeq[j];
            if (e < 100) { o.kind[p] = 0; o.tile[p] = (signed char)(u + 1); o.idx[p] = (signed char)e; }
            else { o.kind[p] = 1; o.tile[p] = (signed char)u; o.idx[p] = (signed char)(e - 100); }
        }
    for (int i = 0; i < 8; ++i, ++p) { o.kind[p] = 1; o.tile[p] = 15; o.idx[p] = (signed char)i; }
    for (; p < 16 * 24; ++p) { o.kind[p] = 2; o.tile[p] = 0; o.idx[p] = 0; }
    return o;
}
constexpr GemmOrder kGemmOrder = make_gemm_order();
static_assert(kC2Tiles == 16 && kPerC2Groups == 24 && kGemm1Groups == 15 && kGemm2Groups == 8, "kGemmOrder is written for these");

constexpr int kConv4Groups = 4 * 9 * 2;   // 72
constexpr int kConv59Groups = 5 * 9 * 2;  // 90
constexpr int kGemmBase = kInputGroups;
constexpr int kConvBase = kGemmBase + kC2Tiles * kPerC2Groups;          // 464
constexpr int kStreamGroups = kConvBase + kConv4Groups + kConv59Groups;  // 626
constexpr int kStreamChunkMax = 32;
// The kernel walks the stream as a ring, board after board: padded to 640 groups = 40 chunks of 16,
// a multiple of the ring's slot count and of the register look-ahead, so that every board starts in
// the same ring phase; the prefetch wraps to group 0 instead of running past the end.
constexpr int kStreamPadGroups = 14;
constexpr int kStreamWalkGroups = kStreamGroups + kStreamPadGroups;  // 640
static_assert(kStreamWalkGroups % kStreamChunkMax == 0, "");
constexpr size_t kStreamBytes = (size_t)kStreamWalkGroups * kGroupBytes;

// float32 shifts that initialise accumulators: x3 (128), x4..x9 (6 x 32), x10 (16), logit weights (16)
constexpr int kBiasX3 = 0, kBiasX4 = 128, kBiasX10 = 128 + 192, kLogitW = kBiasX10 + 16;
constexpr int kBiasFloats = kLogitW + 16;  // 352
constexpr size_t kBiasBytes = kBiasFloats * sizeof(float);

// head image: [sq][lane][8] bf16: lane l holds W8[sq][ci = 8(l>>4) + j][co = l & 15] * inv10[co]
constexpr size_t kHeadBytes = 64 * 64 * 16;  // 64 KiB

constexpr size_t kOffStream = 0;
constexpr size_t kOffBias = kStreamBytes;
constexpr size_t kOffHead = kOffBias + ((kBiasBytes + 255) & ~(size_t)255);
constexpr size_t kImageBytes = kOffHead + kHeadBytes;

// Which x1 register a (conv, local channel c) lands in, and on which lane half.
constexpr int x1_reg_of(int conv, int c) { return kX1RegBase[conv] + (c >> 1); }

}  // namespace b16
}  // namespace bf
