// Host-side weight preparation for the 16x16x32 bf16 evaluation kernel: batch-norm folding, rounding
// to bf16 and MFMA fragment ordering (layout and arithmetic: bf_layout_t16.h).
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/batchfirst.h"
#include "bf_internal.h"
#include "bf_layout_t16.h"

namespace bf {

namespace t16 {

namespace {

struct Canonical {
    const float* in_w[kNumInputConvs];
    const float *bn_mean[10], *bn_var[10], *bn_beta[10];
    const float* tower_w[9];
    const float* logit_w;
};

Canonical slice(const float* blob, bool policy) {
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
    for (int j = 0; j < 9; ++j) {
        if (policy && j == 8) {  // the move-scoring network has no 8x8 layer (move_evaluation_ann.py:291)
            c.tower_w[j] = nullptr;
            c.bn_mean[j + 1] = c.bn_var[j + 1] = c.bn_beta[j + 1] = nullptr;
            continue;
        }
        c.tower_w[j] = blob + pos;
        pos += (size_t)kTower[j].k * kTower[j].k * kTower[j].cin * kTower[j].cout;
        bn(j + 1, kTower[j].cout);
    }
    c.logit_w = blob + pos;
    return c;
}

uint16_t to_bf16(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);  // round to nearest even; finite inputs only
    return (uint16_t)(u >> 16);
}
float from_bf16(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// v == out[0] + out[1] + out[2] exactly
void split3(float v, uint16_t out[3]) {
    float r = v;
    for (int i = 0; i < 3; ++i) {
        out[i] = to_bf16(r);
        r -= from_bf16(out[i]);
    }
}

}  // namespace

void pack(const float* blob, unsigned char* image, bool policy) {
    const Canonical c = slice(blob, policy);
    std::memset(image, 0, kImageBytes);

    const int bn_ch[10] = {236, 512, 128, 32, 32, 32, 32, 32, 32, 16};
    std::vector<std::vector<float>> inv(10), shift(10);
    for (int l = 0; l < (policy ? 9 : 10); ++l) {
        inv[l].resize(bn_ch[l]);
        shift[l].resize(bn_ch[l]);
        for (int ch = 0; ch < bn_ch[l]; ++ch) {
            float s = 1.0f / std::sqrt(c.bn_var[l][ch] + kBnEps);
            inv[l][ch] = s;
            shift[l][ch] = c.bn_beta[l][ch] - c.bn_mean[l][ch] * s;
        }
    }
    // shifts, every 16-row tile as [q][r] = channel 4r + q
    float* bias = reinterpret_cast<float*>(image + kOffBias);
    auto put_tiles = [&](int base, const std::vector<float>& v, int channels) {
        for (int tile = 0; tile < channels / 16; ++tile)
            for (int q = 0; q < 4; ++q)
                for (int r = 0; r < 4; ++r) bias[base + 16 * tile + 4 * q + r] = v[16 * tile + 4 * r + q];
    };
    put_tiles(kBiasX3, shift[2], 128);
    for (int l = 0; l < 6; ++l) put_tiles(kBiasX4 + 32 * l, shift[3 + l], 32);
    if (!policy) {
        for (int ch = 0; ch < 16; ++ch) bias[kBiasX10 + ch] = shift[9][ch];
        for (int co = 0; co < 16; ++co) bias[kLogitW + co] = c.logit_w[co];
    }

    uint16_t* stream = reinterpret_cast<uint16_t*>(image + kOffStream);
    size_t group = 0;
    // lane l = 16q + row holds A[row][k = 8q + j]
    auto slot = [&](int row, int q, int j) -> uint16_t& { return stream[group * 512 + (16 * q + row) * 8 + j]; };

    // 1. input convs: conv, step (two taps), row tile
    for (int i = 0; i < kNumInputConvs; ++i) {
        const InputConv& cv = kInputConvs[i];
        for (int s = 0; s < steps_of(i); ++s) {
            const TapPair tp = step_taps(i, s);
            for (int tile = 0; tile < kRowTiles[i]; ++tile) {
                for (int row = 0; row < 16; ++row) {
                    const int ch = tile_channel(tile, row & 3, row >> 2);   // row 4q' + r -> local channel 16 tile + 4r + q'
                    const bool ones_row = (i == kOnesConvA && (ch == 18 || ch == 19)) || (i == kOnesConvB && ch == 18);
                    for (int q = 0; q < 4; ++q) {
                        const int tap = (q < 2) ? tp.a : tp.b;
                        for (int j = 0; j < 8; ++j) {
                            const int plane = 8 * (q & 1) + j;
                            uint16_t v = 0;
                            if (tap >= 0) {
                                if (ch < cv.ch) {
                                    if (plane < kPlanes) {
                                        v = to_bf16(c.in_w[i][((size_t)tap * kPlanes + plane) * cv.ch + ch] * inv[0][cv.chan_base + ch]);
                                    } else {
                                        uint16_t parts[3];
                                        split3(shift[0][cv.chan_base + ch], parts);
                                        for (int term = 0; term < 3; ++term)
                                            if (tap == shift_tap(i, term)) v = parts[term];
                                    }
                                } else if (ones_row && plane == 15 && tap == shift_tap(i, 0)) {
                                    v = to_bf16(1.0f);
                                }
                            }
                            slot(row, q, j) = v;
                        }
                    }
                }
                ++group;
            }
        }
    }
    // 2. the 1x1 pair in the order the kernel takes the groups (kPairOrder)
    auto write_l1 = [&](int tile, int s) {   // hidden row tile `tile`, k-step s of x1
        for (int row = 0; row < 16; ++row) {
            const int co = 16 * tile + tile_channel(0, row & 3, row >> 2);
            uint16_t parts[3];
            split3(shift[1][co], parts);
            for (int q = 0; q < 4; ++q)
                for (int j = 0; j < 8; ++j) {
                    const int n = 8 * s + j;
                    uint16_t v = 0;
                    if (n < kX1Regs) {
                        const X1Slot sl = x1_slot(n);
                        const int local = tile_channel(sl.tile, sl.r, q);
                        if (local < kInputConvs[sl.conv].ch) {
                            const int ci = kInputConvs[sl.conv].chan_base + local;
                            v = to_bf16(c.tower_w[0][(size_t)ci * kC2 + co] * inv[1][co]);
                        } else if (sl.conv == kOnesConvA && local == 18) v = parts[0];
                        else if (sl.conv == kOnesConvA && local == 19) v = parts[1];
                        else if (sl.conv == kOnesConvB && local == 18) v = parts[2];
                    }
                    slot(row, q, j) = v;
                }
        }
    };
    auto write_l2 = [&](int h, int o) {   // k-step h = hidden tiles 2h, 2h + 1; output row tile o
        for (int row = 0; row < 16; ++row) {
            const int co = 16 * o + tile_channel(0, row & 3, row >> 2);
            for (int q = 0; q < 4; ++q)
                for (int j = 0; j < 8; ++j) {
                    const int ci = 16 * (2 * h + (j >> 2)) + 4 * (j & 3) + q;   // register j & 3 of hidden tile 2h + (j >> 2)
                    slot(row, q, j) = to_bf16(c.tower_w[1][(size_t)ci * kC3 + co] * inv[2][co]);
                }
        }
    };
    for (int p = 0; p < kPairGroups; ++p) {
        if (kPairOrder.kind[p] == 0) write_l1(kPairOrder.a[p], kPairOrder.b[p]);
        else write_l2(kPairOrder.a[p], kPairOrder.b[p]);
        ++group;
    }
    // 3. 3x3 layers: per (chunk, tap, output row tile); k in plane order
    auto conv3 = [&](const float* w, int cin, int chunk, const std::vector<float>& scale) {
        for (int tap = 0; tap < 9; ++tap)
            for (int tile = 0; tile < 2; ++tile) {
                for (int row = 0; row < 16; ++row) {
                    const int co = 16 * tile + tile_channel(0, row & 3, row >> 2);
                    for (int q = 0; q < 4; ++q)
                        for (int j = 0; j < 8; ++j) {
                            const int ci = chunk * 32 + plane_channel(q, j);
                            slot(row, q, j) = to_bf16(w[((size_t)tap * cin + ci) * 32 + co] * scale[co]);
                        }
                }
                ++group;
            }
    };
    for (int k = 0; k < 4; ++k) conv3(c.tower_w[2], 128, k, inv[3]);
    for (int l = 0; l < 5; ++l) conv3(c.tower_w[3 + l], 32, 0, inv[4 + l]);

    if (policy) {
        // no 8x8 layer: the head region carries the logit kernel, float32 [filter][channel]
        float* lw = reinterpret_cast<float*>(image + kOffHead);
        for (int f = 0; f < kPolicyFilters; ++f)
            for (int ci = 0; ci < 32; ++ci) lw[f * 32 + ci] = c.logit_w[ci * kPolicyFilters + f];
        return;
    }
    // ---- head: one fragment per square, lane l: co = l & 15, k-slot (q = l >> 4, j) in plane order
    uint16_t* head = reinterpret_cast<uint16_t*>(image + kOffHead);
    for (int sq = 0; sq < 64; ++sq)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int co = lane & 15, ci = plane_channel(lane >> 4, j);
                head[((size_t)sq * 64 + lane) * 8 + j] = to_bf16(c.tower_w[8][((size_t)sq * 32 + ci) * 16 + co] * inv[9][co]);
            }
}

}  // namespace t16

void pack_weights_t16(const float* blob, unsigned char* image, bool policy) { t16::pack(blob, image, policy); }

}  // namespace bf

extern "C" size_t bf_packed_t16_bytes(void) { return bf::t16::kImageBytes; }

extern "C" int bf_pack_weights_t16(const float* h_weights, size_t n_floats, void* h_image, size_t n_bytes) {
    if (!h_weights || !h_image) return bf::fail(BF_E_ARG, "bf_pack_weights_t16: null pointer");
    if (n_floats != bf::kWeightFloats)
        return bf::fail(BF_E_ARG, "bf_pack_weights_t16: expected %zu weight floats, got %zu", bf::kWeightFloats, n_floats);
    if (n_bytes != bf::t16::kImageBytes)
        return bf::fail(BF_E_ARG, "bf_pack_weights_t16: expected room for %zu bytes, got %zu", bf::t16::kImageBytes, n_bytes);
    bf::pack_weights_t16(h_weights, static_cast<unsigned char*>(h_image), false);
    return BF_OK;
}
