// Shared host/device description of the value network and of the fragment-ordered weight image.
//
// The evaluation kernel keeps "channels in rows, squares on lanes": every layer is computed as
// out^T[channel][square] = W^T[channel][k] * in^T[k][square] with v_mfma_f32_32x32x2_f32, so a
// finished 32x32 accumulator tile (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), col = lane&31) is
// already the B operand of the next layer: register t of a tile is k-step t, lanes 0-31 carry
// the k of row rowmap(t,0), lanes 32-63 that of rowmap(t,1).  The weights are therefore stored
// once, on the host, as a linear stream of A fragments in exactly the order one wave consumes
// them ("k-step" = 64 floats, one per lane; "group" = 4 consecutive k-steps interleaved so a
// lane reads its 4 floats with one 16-byte load).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace bf {

constexpr int kPlanes = 15;
constexpr int kNumInputConvs = 9;
constexpr int kX1Channels = 236;
constexpr int kX1Regs = 118;  // channel pair (2r, 2r+1) lives in register r: lanes 0-31 / 32-63
constexpr int kC2 = 512, kC3 = 128, kC4 = 32, kC10 = 16;
constexpr float kBnEps = 1e-3f;

struct InputConv { int k, dh, dw, ch, chan_base, tiles; };
// reference: ann_creation_helper.py:211-236, evaluation_ann.py:22
constexpr InputConv kInputConvs[kNumInputConvs] = {
    {3, 1, 1, 64, 0, 2},   {3, 2, 2, 32, 64, 1},  {3, 3, 3, 20, 96, 1},
    {3, 4, 4, 20, 116, 1}, {3, 5, 5, 18, 136, 1}, {3, 6, 6, 18, 154, 1},
    {3, 7, 7, 16, 172, 1}, {2, 2, 4, 24, 188, 1}, {2, 4, 2, 24, 212, 1}};

constexpr int rowmap(int t, int hl) { return (t & 3) + 8 * (t >> 2) + 4 * hl; }

// ---- canonical blob offsets (floats) -------------------------------------------------
constexpr size_t kInputKernelFloats(int i) {
    return (size_t)kInputConvs[i].k * kInputConvs[i].k * kPlanes * kInputConvs[i].ch;
}
struct TowerLayer { int k, cin, cout; };
constexpr TowerLayer kTower[9] = {{1, 236, 512}, {1, 512, 128}, {3, 128, 32}, {3, 32, 32}, {3, 32, 32},
                                  {3, 32, 32},   {3, 32, 32},   {3, 32, 32},  {8, 32, 16}};
constexpr size_t kWeightFloats = 333608;
// move-scoring network: no 8x8 layer (and its batch-norm), logit kernel 32 x 73
constexpr int kPolicyFilters = 73;
constexpr size_t kPolicyWeightFloats = kWeightFloats - (8 * 8 * 32 * 16 + 3 * 16 + 16) + 32 * kPolicyFilters;

// ---- packed image layout (floats) -----------------------------------------------------
constexpr int kGroupFloats = 256;  // 64 lanes x 4 k-steps
// groups per section of the A-fragment stream
constexpr int kInputTapTiles = 7 * 9 + 9 /*conv0 second tile*/ + 2 * 4;  // 80 (tile,tap) pairs
// Stage 1 runs on v_mfma_f32_32x32x16_bf16: the one-hot operand is exact in bf16, and the fp32
// weight is split into three bf16 terms (hi + mid + lo == w exactly), so one (tile, tap) costs
// three 32-cycle MFMAs instead of eight 64-cycle fp32 ones at unchanged fp32 accuracy.  One
// group = one MFMA's A fragment (8 bf16 per lane: planes 8h..8h+7 of the lane's channel row).
constexpr int kInputSplits = 3;
constexpr int kInputGroups = kInputTapTiles * kInputSplits;               // 240
constexpr int kGemm1Groups = 30;   // 118 k-steps padded to 120
constexpr int kGemm2Groups = 16;   // 4 out tiles x 16 k-steps
constexpr int kC2Tiles = 16;
constexpr int kPerC2Groups = kGemm1Groups + kGemm2Groups;  // 46
constexpr int kConvTapGroups = 4;                          // 16 k-steps per tap and 32-ch chunk
constexpr int kConv4Groups = 4 * 9 * kConvTapGroups;       // 144
constexpr int kConv59Groups = 5 * 9 * kConvTapGroups;      // 180
constexpr int kStreamGroups = kInputGroups + kC2Tiles * kPerC2Groups + kConv4Groups + kConv59Groups;  // 1300
// BF_PRECISION_F32_SPLIT: the same input and 1x1 sections (other contents, pack.cpp); a 3x3 chunk is 9 taps x 2 k-steps
// of 16 channels x 3 bf16 terms of the kernels, one group (one v_mfma_f32_32x32x16_bf16 A fragment) each
constexpr int kConvSplitChunkGroups = 9 * 2 * 3;           // 54
constexpr int kStreamGroupsSplit = kInputGroups + kC2Tiles * kPerC2Groups + 9 * kConvSplitChunkGroups;  // 1462 (4 + 5 chunks)
// BF_PRECISION_F32_SPLIT3: the pair with three terms a side too: 15 + 8 k-steps of 16, three kernel terms each
constexpr int kPerC2GroupsSplit3 = (15 + 8) * 3;           // 69
constexpr int kStreamGroupsSplit3 = kInputGroups + kC2Tiles * kPerC2GroupsSplit3 + 9 * kConvSplitChunkGroups;  // 1830
constexpr int kStreamChunk = 8;       // groups a workgroup walks between two barriers (eval_f32.hip)
constexpr int kStreamPadGroups = 32;  // prefetch runs up to three chunks past the end
// one image size for all three precisions (the shorter streams leave the tail of the region unused)
constexpr size_t kStreamFloats = (size_t)(kStreamGroupsSplit3 + kStreamPadGroups) * kGroupFloats;
static_assert(kStreamGroupsSplit3 >= kStreamGroupsSplit && kStreamGroupsSplit >= kStreamGroups, "");

// bias table: folded batch-norm shift per channel, layers back to back
constexpr int kBiasX1 = 0, kBiasX2 = 236, kBiasX3 = 748, kBiasX4 = 876;  // x4..x9: 32 each
constexpr int kBiasX10 = 876 + 6 * 32;                                     // 1068
constexpr int kLogitW = kBiasX10 + 16;                                     // 1084
constexpr int kBiasFloats = kLogitW + 16;                                  // 1100 -> pad
constexpr size_t kBiasFloatsPadded = 1104;

// head image: [half][t*4+cq][lane][4] = W8[sq][ci=rowmap(t,hl)][co=4cq+j] * scale10[co]
constexpr size_t kHeadFloats = 2 * 64 * 64 * 4;

constexpr size_t kOffStream = 0;
constexpr size_t kOffBias = kStreamFloats;
constexpr size_t kOffHead = kOffBias + kBiasFloatsPadded;
constexpr size_t kPackedFloats = kOffHead + kHeadFloats;

constexpr int kMailboxBytes = 32;

}  // namespace bf
