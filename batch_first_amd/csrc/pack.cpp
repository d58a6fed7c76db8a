// Host-side weight preparation: batch-norm folding and MFMA fragment ordering.
//
// Input: the canonical blob (include/batchfirst.h).  Output: the packed image described in
// bf_layout.h -- an A-fragment stream in consumption order, a per-channel bias table and the
// head image.  Batch-norm (ann_creation_helper.py:240-246, :312-317; scale=False, eps 1e-3)
// is folded: W' = W * inv, b' = beta - mean * inv with inv = 1/sqrt(var + eps); the bias goes
// into the accumulator's initial value inside the kernel.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/batchfirst.h"
#include "bf_layout.h"
#include "bf_internal.h"

namespace bf {

namespace {

struct Canonical {
    const float* in_w[kNumInputConvs];
    const float *bn_mean[10], *bn_var[10], *bn_beta[10];
    const float* tower_w[9];
    const float* logit_w;
};

// `policy`: the move-scoring network's blob (batch_first/anns/move_evaluation_ann.py:18-45, :280, :291):
// same input convs and tower without the 8x8 layer, logit kernel [1][1][32][73].
Canonical slice(const float* blob, bool policy = false) {
    Canonical c;
    size_t pos = 0;
    for (int i = 0; i < kNumInputConvs; ++i) {
        c.in_w[i] = blob + pos;
        pos += kInputKernelFloats(i);
    }
    auto bn = [&](int idx, int ch) {
        c.bn_mean[idx] = blob + pos;
        c.bn_var[idx] = blob + pos + ch;
        c.bn_beta[idx] = blob + pos + 2 * ch;
        pos += 3 * (size_t)ch;
    };
    bn(0, kX1Channels);
    const int layers = policy ? 8 : 9;
    for (int j = 0; j < 9; ++j) {
        if (j >= layers) {
            c.tower_w[j] = nullptr;
            c.bn_mean[j + 1] = c.bn_var[j + 1] = c.bn_beta[j + 1] = nullptr;
            continue;
        }
        c.tower_w[j] = blob + pos;
        pos += (size_t)kTower[j].k * kTower[j].k * kTower[j].cin * kTower[j].cout;
        bn(j + 1, kTower[j].cout);
    }
    c.logit_w = blob + pos;
    pos += policy ? kPolicyFilters * 32 : 16;
    return c;
}

// channel (within a 32-row tile) that sits in accumulator row `row` when channel c is kept at
// register c>>1 of lane half c&1 (so that x1 register r holds channels 2r / 2r+1)
int pair_channel_of_row(int row) {
    int hl = (row >> 2) & 1;
    int t = (row & 3) + 4 * (row >> 3);
    return 2 * t + hl;
}


uint16_t bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);  // finite inputs only (weights)
    return (uint16_t)(u >> 16);
}
float bf16_to_float(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// w == hi + mid + lo exactly: each step removes at least 8 significant bits of the remainder
void split_bf16x3(float w, uint16_t out[3]) {
    float r = w;
    for (int i = 0; i < 3; ++i) {
        out[i] = bf16_rne(r);
        r -= bf16_to_float(out[i]);
    }
}

}  // namespace

void pack_weights(const float* blob, float* packed, bool policy, int split) {
    const Canonical c = slice(blob, policy);
    std::memset(packed, 0, kPackedFloats * sizeof(float));

    // folded scale per batch-norm layer
    std::vector<std::vector<float>> inv(10);
    const int bn_ch[10] = {236, 512, 128, 32, 32, 32, 32, 32, 32, 16};
    float* bias = packed + kOffBias;
    const int bias_off[10] = {kBiasX1, kBiasX2, kBiasX3, kBiasX4, kBiasX4 + 32, kBiasX4 + 64,
                              kBiasX4 + 96, kBiasX4 + 128, kBiasX4 + 160, kBiasX10};
    for (int l = 0; l < (policy ? 9 : 10); ++l) {
        inv[l].resize(bn_ch[l]);
        for (int ch = 0; ch < bn_ch[l]; ++ch) {
            float s = 1.0f / std::sqrt(c.bn_var[l][ch] + kBnEps);
            inv[l][ch] = s;
            bias[bias_off[l] + ch] = c.bn_beta[l][ch] - c.bn_mean[l][ch] * s;
        }
    }
    if (!policy)
        for (int co = 0; co < 16; ++co) bias[kLogitW + co] = c.logit_w[co];

    // ---- A-fragment stream -----------------------------------------------------------
    float* stream = packed + kOffStream;
    size_t kstep = 0;  // running k-step index; element (kstep, lane) lives at
                       // stream[(kstep/4)*256 + lane*4 + kstep%4]
    auto put = [&](int lane, float v) { stream[(kstep >> 2) * kGroupFloats + lane * 4 + (kstep & 3)] = v; };

    // 1. input convs: for conv, for tap, for tile, three bf16 split terms; one group each.
    //    Lane (row, h) holds planes 8h..8h+7 of its channel row as 8 bf16 = 4 dwords.
    for (int i = 0; i < kNumInputConvs; ++i) {
        const InputConv& cv = kInputConvs[i];
        for (int tap = 0; tap < cv.k * cv.k; ++tap) {
            for (int tile = 0; tile < cv.tiles; ++tile) {
                for (int split = 0; split < kInputSplits; ++split) {
                    uint16_t* grp = reinterpret_cast<uint16_t*>(stream + (kstep >> 2) * kGroupFloats);
                    for (int lane = 0; lane < 64; ++lane) {
                        int h = lane >> 5, row = lane & 31;
                        int ch = tile * 32 + pair_channel_of_row(row);
                        for (int j = 0; j < 8; ++j) {
                            int plane = 8 * h + j;
                            float v = 0.0f;
                            if (ch < cv.ch && plane < kPlanes)
                                v = c.in_w[i][((size_t)tap * kPlanes + plane) * cv.ch + ch] * inv[0][cv.chan_base + ch];
                            uint16_t parts[3];
                            split_bf16x3(v, parts);
                            grp[lane * 8 + j] = parts[split];
                        }
                    }
                    kstep += 4;  // one whole group
                }
            }
        }
    }
    // 2. the 1x1 pair, per 32-channel tile of the 512-wide hidden layer
    // split (BF_PRECISION_F32_SPLIT): the same 30 + 16 groups a tile, every group one A fragment of
    // v_mfma_f32_32x32x16_bf16 (lane (row, h): 8 bf16, k = (h, j)), k-steps of 16 with the kernels as two bf16
    // terms (hi, then mid = bf16(w - hi)) in consecutive groups.  236 -> 512: k-step s, element (h, j) = input channel
    // 16 s + 2 j + h (x1 register 8 s + j of lane half h); channel slot 236 holds the folded shift (its B element is
    // 1.0).  512 -> 128: k-step v, element (h, j) = hidden row rowmap(8 v + j, h) (accumulator register 8 v + j).
    // split level 2 (BF_PRECISION_F32_SPLIT3): the same with three terms (hi, mid, lo), 45 + 24 groups a tile.
    if (split) {
        const int terms = split == 2 ? 3 : 2;
        auto put_terms = [&](auto value_of) {   // value_of(lane, j) -> float; writes the hi group, then mid (then lo)
            for (int term = 0; term < terms; ++term) {
                uint16_t* grp = reinterpret_cast<uint16_t*>(stream + (kstep >> 2) * kGroupFloats);
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        uint16_t parts[3];
                        split_bf16x3(value_of(lane, j), parts);   // (two terms: the first two of the three)
                        grp[lane * 8 + j] = parts[term];
                    }
                kstep += 4;
            }
        };
        for (int c2 = 0; c2 < kC2Tiles; ++c2) {
            for (int s = 0; s < 15; ++s)
                put_terms([&](int lane, int j) {
                    const int h = lane >> 5, co = c2 * 32 + (lane & 31), ci = 16 * s + 2 * j + h;
                    if (ci < kX1Channels) return c.tower_w[0][(size_t)ci * kC2 + co] * inv[1][co];
                    return ci == kX1Channels ? bias[kBiasX2 + co] : 0.0f;
                });
            for (int c3 = 0; c3 < 4; ++c3)
                for (int v = 0; v < 2; ++v)
                    put_terms([&](int lane, int j) {
                        const int h = lane >> 5, co = c3 * 32 + (lane & 31), ci = c2 * 32 + rowmap(8 * v + j, h);
                        return c.tower_w[1][(size_t)ci * kC3 + co] * inv[2][co];
                    });
        }
    } else
    for (int c2 = 0; c2 < kC2Tiles; ++c2) {
        for (int r = 0; r < 120; ++r) {  // 236 -> 512, k-step r = input channels 2r, 2r+1
            for (int lane = 0; lane < 64; ++lane) {
                int hl = lane >> 5, co = c2 * 32 + (lane & 31);
                int ci = 2 * r + hl;
                float v = 0.0f;
                if (r < kX1Regs) v = c.tower_w[0][(size_t)ci * kC2 + co] * inv[1][co];
                else if (r == kX1Regs && hl == 0) v = bias[kBiasX2 + co];  // shift rides on a B operand of 1
                put(lane, v);
            }
            ++kstep;
        }
        for (int c3 = 0; c3 < 4; ++c3) {  // 512 -> 128, k-step t = hidden rows rowmap(t, hl)
            for (int t = 0; t < 16; ++t) {
                for (int lane = 0; lane < 64; ++lane) {
                    int hl = lane >> 5, co = c3 * 32 + (lane & 31);
                    int ci = c2 * 32 + rowmap(t, hl);
                    put(lane, c.tower_w[1][(size_t)ci * kC3 + co] * inv[2][co]);
                }
                ++kstep;
            }
        }
    }
    // 3. 3x3 128 -> 32 in four 32-channel chunks, then five 3x3 32 -> 32
    auto conv3 = [&](const float* w, int cin, int chunk, const std::vector<float>& scale) {
        for (int tap = 0; tap < 9; ++tap)
            for (int s = 0; s < 16; ++s) {
                for (int lane = 0; lane < 64; ++lane) {
                    int hl = lane >> 5, co = lane & 31;
                    int ci = chunk * 32 + 2 * s + hl;
                    put(lane, w[((size_t)tap * cin + ci) * 32 + co] * scale[co]);
                }
                ++kstep;
            }
    };
    // split: per tap two k-steps of 16 input channels, the kernels as THREE bf16 terms (hi, mid, lo: hi + mid + lo == w
    // exactly) in consecutive groups.  k-step s, element (h, j) = input channel rowmap(8 h + j, s) of the chunk: the
    // channel that accumulator register 8 h + j of lane half s held in the layer before (eval_f32.hip, plane layout).
    auto conv3_split = [&](const float* w, int cin, int chunk, const std::vector<float>& scale) {
        for (int tap = 0; tap < 9; ++tap)
            for (int s = 0; s < 2; ++s)
                for (int term = 0; term < 3; ++term) {
                    uint16_t* grp = reinterpret_cast<uint16_t*>(stream + (kstep >> 2) * kGroupFloats);
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int h = lane >> 5, co = lane & 31, ci = chunk * 32 + rowmap(8 * h + j, s);
                            uint16_t parts[3];
                            split_bf16x3(w[((size_t)tap * cin + ci) * 32 + co] * scale[co], parts);
                            grp[lane * 8 + j] = parts[term];
                        }
                    kstep += 4;
                }
    };
    if (split) {
        for (int q = 0; q < 4; ++q) conv3_split(c.tower_w[2], 128, q, inv[3]);
        for (int l = 0; l < 5; ++l) conv3_split(c.tower_w[3 + l], 32, 0, inv[4 + l]);
    } else {
        for (int q = 0; q < 4; ++q) conv3(c.tower_w[2], 128, q, inv[3]);
        for (int l = 0; l < 5; ++l) conv3(c.tower_w[3 + l], 32, 0, inv[4 + l]);
    }

    // ---- head: 8x8 valid conv as a 2048 -> 16 dense layer --------------------------
    float* head = packed + kOffHead;
    if (policy) {
        // no 8x8 layer: the head region carries the logit kernel transposed to [filter][channel]
        // (logit_layer: 1x1, 32 -> 73, no bias, move_evaluation_ann.py:35-44)
        for (int f = 0; f < kPolicyFilters; ++f)
            for (int ci = 0; ci < 32; ++ci) head[f * 32 + ci] = c.logit_w[ci * kPolicyFilters + f];
        return;
    }
    for (int half = 0; half < 2; ++half)
        for (int t = 0; t < 16; ++t)
            for (int cq = 0; cq < 4; ++cq)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        int hl = lane >> 5, sq = half * 32 + (lane & 31);
                        int ci = rowmap(t, hl), co = 4 * cq + j;
                        head[(((size_t)half * 64 + t * 4 + cq) * 64 + lane) * 4 + j] =
                            c.tower_w[8][((size_t)sq * 32 + ci) * 16 + co] * inv[9][co];
                    }
}

}  // namespace bf

extern "C" size_t bf_packed_floats(void) { return bf::kPackedFloats; }

extern "C" int bf_pack_policy_weights(const float* h_weights, size_t n_floats, float* h_packed, size_t n_packed) {
    if (!h_weights || !h_packed) return bf::fail(BF_E_ARG, "bf_pack_policy_weights: null pointer");
    if (n_floats != bf::kPolicyWeightFloats)
        return bf::fail(BF_E_ARG, "bf_pack_policy_weights: expected %zu weight floats, got %zu", bf::kPolicyWeightFloats, n_floats);
    if (n_packed != bf::kPackedFloats)
        return bf::fail(BF_E_ARG, "bf_pack_policy_weights: expected room for %zu packed floats, got %zu", bf::kPackedFloats, n_packed);
    bf::pack_weights(h_weights, h_packed, true, false);
    return BF_OK;
}

extern "C" int bf_pack_weights_ex(const float* h_weights, size_t n_floats, float* h_packed, size_t n_packed, int precision) {
    if (!h_weights || !h_packed) return bf::fail(BF_E_ARG, "bf_pack_weights: null pointer");
    if (n_floats != bf::kWeightFloats)
        return bf::fail(BF_E_ARG, "bf_pack_weights: expected %zu weight floats, got %zu", bf::kWeightFloats, n_floats);
    if (n_packed != bf::kPackedFloats)
        return bf::fail(BF_E_ARG, "bf_pack_weights: expected room for %zu packed floats, got %zu", bf::kPackedFloats, n_packed);
    if (precision != BF_PRECISION_F32 && precision != BF_PRECISION_F32_SPLIT && precision != BF_PRECISION_F32_SPLIT3)
        return bf::fail(BF_E_ARG, "bf_pack_weights_ex: precision %d has no image of this kind", precision);
    bf::pack_weights(h_weights, h_packed, false,
                     precision == BF_PRECISION_F32_SPLIT ? 1 : precision == BF_PRECISION_F32_SPLIT3 ? 2 : 0);
    return BF_OK;
}

extern "C" int bf_pack_weights(const float* h_weights, size_t n_floats, float* h_packed, size_t n_packed) {
    return bf_pack_weights_ex(h_weights, n_floats, h_packed, n_packed, BF_PRECISION_F32);
}
