// Fragment-ordered weight image of the 16x16x32 bf16 evaluation kernel (eval_t16.hip, pack_t16.cpp).
//
// Arithmetic of the bf16 variant (BASELINE configs[2]; restated on the CPU by
// oracle/network.py:forward_bf16_model): batch-norm is folded as in the fp32 path (W' = W * inv in
// float32), W' is rounded to bf16 (round-to-nearest-even), every activation x1..x9 is rounded to bf16
// after its ReLU, all sums are float32, the folded batch-norm shifts stay float32-exact, x10 and the
// logit are float32.  Tiled on v_mfma_f32_16x16x32_bf16: on this part the chip holds a higher clock on
// that shape than on 32x32x16 at equal cycles per FLOP, and a board's 64 squares become
// FOUR column tiles, which gives every layer four independent accumulation chains to interleave.
//
// Lane maps of v_mfma_f32_16x16x32_bf16 (lane l: c = l & 15, q = l >> 4):
//   A[row = c][k = 8q + j], B[k = 8q + j][col = c], j = 0..7 (one 16-byte fragment per lane);
//   D[row = 4q + r][col = c], r = 0..3.
// Orientation as before: out^T[channel][square] = W^T[channel][k] * in^T[k][square]; column tile t holds
// squares 16t + c, i.e. ranks 2t and 2t + 1.  One "group" = one A fragment = 64 lanes x 8 bf16 = 1 KiB.
//
// Row <-> channel inside a 16-row tile: row 4q + r is local channel 4r + q, so that register r of a tile
// holds four consecutive channels (one per q) and a partly filled tile keeps its channels in its first
// registers.  A finished tile pair feeds the next layer without lane movement: the B fragment of a
// k-step is registers r = 0..3 of one row tile (j = 0..3) and of a second one (j = 4..7), so k-slot
// (q, j) is channel 4(j & 3) + q of tile (j >> 2); the packer permutes the next layer's k accordingly.
//
//   stage 1   k-step = TWO taps x 16 planes (15 one-hot planes + a 16th that is 1.0 on every square,
//             padding included): lanes q = 0,1 look up tap A's square, q = 2,3 tap B's; q & 1 selects
//             planes 0-7 / 8-15.  A 3x3 conv takes five k-steps (taps paired so that a step whose taps
//             all fall off the board for a column tile is skipped for that tile), a 2x2 conv two.  The
//             folded shift rides on the ones-plane of three taps that no column tile skips, as three
//             bf16 terms whose sum is the float32 shift.
//   x1        per column tile 64 live registers = 256 channel slots = the 8 k-steps of the 236->512 layer:
//             the 236 channels, three "ones" channels in the padding rows of convs 4 and 5 (they carry
//             that layer's shift the same way), zero slots for the rest; 32 packed dwords.
//   1x1 pair  per pair of 16-row hidden tiles: 16 groups (236->512: 2 tiles x 8 k-steps) and 8 groups
//             (512->128: that pair's k-step x 8 output tiles); stream order kPairOrder.
//   3x3       per (32-channel input chunk, tap, output row tile): conv4 = 4 x 9 x 2, conv5..9 = 5 x 9 x 2.
//             Activation planes keep a position's 32 channels in the order k = 8q + 4*tile + r (one
//             16-byte store per lane writes its registers of both row tiles), so k-slot (q, j) is channel
//             16(j >> 2) + 4(j & 3) + q, the same map as between the two 1x1 layers.
//   head      8x8 valid conv as out[board][co]: one MFMA per square, k in the plane order above.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "bf_layout.h"

namespace bf {
namespace t16 {

constexpr int kGroupBytes = 1024;

// ---- stage 1 ---------------------------------------------------------------------------------------
constexpr int kRowTiles[kNumInputConvs] = {4, 2, 2, 2, 2, 2, 1, 2, 2};     // 16-row tiles per conv
// live registers per column tile and conv: channels / 4 rounded up to an EVEN count (a packed dword never
// straddles two convs), convs 4 and 5 padded 18 -> 20 channels first
constexpr int kX1RegCount[kNumInputConvs] = {16, 8, 6, 6, 6, 6, 4, 6, 6};
constexpr int kX1RegBase[kNumInputConvs] = {0, 16, 24, 30, 36, 42, 48, 52, 58};
constexpr int kX1Regs = 64, kX1Dwords = 32;
// the all-ones channel slots: conv 4 local channels 18 and 19, conv 5 local channel 18
constexpr int kOnesConvA = 4, kOnesConvB = 5;

struct TapPair { signed char a, b; };   // tap indices (row-major in the k x k kernel); b = -1: none
// 3x3: row -d: taps 0 1 2, row 0: 3 4 5, row +d: 6 7 8.  Steps 0 and 1 hold one row each, step 2 the two
// corner taps of the right column, steps 3 and 4 the middle row (never skipped: they carry the shift).
constexpr TapPair kSteps3[5] = {{0, 1}, {6, 7}, {2, 8}, {3, 4}, {5, -1}};
constexpr TapPair kSteps2[2] = {{0, 1}, {2, 3}};
constexpr int steps_of(int conv) { return kInputConvs[conv].k == 3 ? 5 : 2; }
constexpr TapPair step_taps(int conv, int s) { return kInputConvs[conv].k == 3 ? kSteps3[s] : kSteps2[s]; }
// tap whose ones-plane weight carries term `term` (0, 1, 2) of the folded shift
constexpr int shift_tap(int conv, int term) { return kInputConvs[conv].k == 3 ? 3 + term : term; }
constexpr int tap_dr(int conv, int tap) {
    return (tap / kInputConvs[conv].k) * kInputConvs[conv].dh - ((kInputConvs[conv].k - 1) * kInputConvs[conv].dh) / 2;
}
constexpr int tap_df(int conv, int tap) {
    return (tap % kInputConvs[conv].k) * kInputConvs[conv].dw - ((kInputConvs[conv].k - 1) * kInputConvs[conv].dw) / 2;
}
// does the tap reach the board from some square of column tile t (ranks 2t, 2t + 1)?
constexpr bool tap_reaches(int conv, int tap, int t) {
    const int dr = tap_dr(conv, tap);
    return (2 * t + dr >= 0 && 2 * t + dr <= 7) || (2 * t + 1 + dr >= 0 && 2 * t + 1 + dr <= 7);
}
// a step is computed for column tile t unless all its taps miss the board there; the 2x2 (4,2) conv is
// never skipped (no tap of it reaches every tile, and the shift needs three taps that every tile computes)
constexpr bool step_valid(int conv, int s, int t) {
    if (kInputConvs[conv].k == 2) return true;
    const TapPair p = step_taps(conv, s);
    return tap_reaches(conv, p.a, t) || (p.b >= 0 && tap_reaches(conv, p.b, t));
}
constexpr int input_groups() {
    int g = 0;
    for (int i = 0; i < kNumInputConvs; ++i) g += steps_of(i) * kRowTiles[i];
    return g;
}
constexpr int kInputGroups = input_groups();   // 83
static_assert(kInputGroups == 83, "");
constexpr int input_group_base(int conv) {
    int g = 0;
    for (int i = 0; i < conv; ++i) g += steps_of(i) * kRowTiles[i];
    return g;
}

// x1 register n (0..59) of a column tile -> conv, row tile, register
struct X1Slot { int conv, tile, r; };
constexpr X1Slot x1_slot(int n) {
    int i = 0;
    while (i + 1 < kNumInputConvs && n >= kX1RegBase[i + 1]) ++i;
    const int local = n - kX1RegBase[i];
    return X1Slot{i, local / 4, local % 4};
}
// local channel of (row tile, register r, lane quarter q)
constexpr int tile_channel(int tile, int r, int q) { return 16 * tile + 4 * r + q; }

// ---- the 1x1 pair ------------------------------------------------------------------------------------
constexpr int kL1Steps = 8;            // k-steps of the 236->512 layer (kX1Regs / 8)
constexpr int kHiddenPairs = 16;       // pairs of 16-row hidden tiles = k-steps of the 512->128 layer
constexpr int kOutTiles = 8;           // 16-row tiles of x3
constexpr int kPairGroups = kHiddenPairs * (2 * kL1Steps + kOutTiles);   // 384
// Stream order of the pair's groups = the order the kernel takes them in.  kind 0: 236->512, a = hidden
// row tile (0..31), b = k-step; kind 1: 512->128, a = hidden pair (its k-step), b = output tile.
// Prologue: pair 0's sixteen kind-0 groups.  Then per pair h: eight kind-0 groups of pair h+1 (the packs of
// pair h ride in their gaps), then (kind 0, kind 1) alternating eight times.  Epilogue: pair 15's kind 1.
struct PairOrder {
    signed char kind[kPairGroups], a[kPairGroups], b[kPairGroups];
};
constexpr PairOrder make_pair_order() {
    PairOrder o{};
    int p = 0;
    for (int tile = 0; tile < 2; ++tile)
        for (int s = 0; s < kL1Steps; ++s, ++p) { o.kind[p] = 0; o.a[p] = (signed char)tile; o.b[p] = (signed char)s; }
    for (int h = 0; h + 1 < kHiddenPairs; ++h) {
        for (int s = 0; s < kL1Steps; ++s, ++p) { o.kind[p] = 0; o.a[p] = (signed char)(2 * h + 2); o.b[p] = (signed char)s; }
        for (int s = 0; s < kL1Steps; ++s) {
            o.kind[p] = 0; o.a[p] = (signed char)(2 * h + 3); o.b[p] = (signed char)s; ++p;
            o.kind[p] = 1; o.a[p] = (signed char)h; o.b[p] = (signed char)s; ++p;
        }
    }
    for (int s = 0; s < kOutTiles; ++s, ++p) { o.kind[p] = 1; o.a[p] = (signed char)(kHiddenPairs - 1); o.b[p] = (signed char)s; }
    return o;
}
constexpr PairOrder kPairOrder = make_pair_order();

// ---- 3x3 layers and the stream -----------------------------------------------------------------------
constexpr int kConv4Groups = 4 * 9 * 2;   // 72
constexpr int kConv59Groups = 5 * 9 * 2;  // 90
constexpr int kPairBase = kInputGroups;
constexpr int kConvBase = kPairBase + kPairGroups;                          // 467
constexpr int kStreamGroups = kConvBase + kConv4Groups + kConv59Groups;     // 629
constexpr int kStreamWalkGroups = 640;    // walked as a ring, board after board: padded to 40 chunks of 16, a multiple of the ring's slot count and of the register look-ahead, so that every board starts in the same ring phase
constexpr int kStreamPadGroups = kStreamWalkGroups - kStreamGroups;         // 11
static_assert(kStreamPadGroups >= 0, "");
constexpr size_t kStreamBytes = (size_t)kStreamWalkGroups * kGroupBytes;

// channel of k-slot (q, j) inside a 32-channel activation chunk (plane order)
constexpr int plane_channel(int q, int j) { return 16 * (j >> 2) + 4 * (j & 3) + q; }

// float32 shifts: x3 (128), x4..x9 (6 x 32), x10 (16), logit weights (16), each 16-row tile stored as
// [q][r] = channel 4r + q so that a lane reads its four accumulator seeds with one 16-byte load
constexpr int kBiasX3 = 0, kBiasX4 = 128, kBiasX10 = 128 + 192, kLogitW = kBiasX10 + 16;
constexpr int kBiasFloats = kLogitW + 16;  // 352
constexpr size_t kBiasBytes = kBiasFloats * sizeof(float);

// head image: [sq][lane][8] bf16: lane l holds W8[sq][ci = plane_channel(l >> 4, j)][co = l & 15] * inv10[co]
constexpr size_t kHeadBytes = 64 * 64 * 16;  // 64 KiB

constexpr size_t kOffStream = 0;
constexpr size_t kOffBias = kStreamBytes;
constexpr size_t kOffHead = kOffBias + ((kBiasBytes + 255) & ~(size_t)255);
constexpr size_t kImageBytes = kOffHead + kHeadBytes;

}  // namespace t16
}  // namespace bf
