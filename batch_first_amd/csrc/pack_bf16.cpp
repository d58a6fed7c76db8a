// Host-side weight preparation for the bf16 evaluation kernel: batch-norm folding, rounding to
// bf16 and MFMA fragment ordering (layout and arithmetic: bf_layout_bf16.h).
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/batchfirst.h"
#include "bf_internal.h"
#include "bf_layout_bf16.h"

namespace bf {

namespace b16 {  // inner scope: the bf16 layout's names hide the fp32 layout's

namespace {

struct Canonical16 {
    const float* in_w[kNumInputConvs];
    const float *bn_mean[10], *bn_var[10], *bn_beta[10];
    const float* tower_w[9];
    const float* logit_w;
};

Canonical16 slice16(const float* blob, bool policy) {
    Canonical16 c;
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
    const Canonical16 c = slice16(blob, policy);
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
    float* bias = reinterpret_cast<float*>(image + kOffBias);
    for (int ch = 0; ch < 128; ++ch) bias[kBiasX3 + ch] = shift[2][ch];
    for (int l = 0; l < 6; ++l)
        for (int ch = 0; ch < 32; ++ch) bias[kBiasX4 + 32 * l + ch] = shift[3 + l][ch];
    if (!policy) {
        for (int ch = 0; ch < 16; ++ch) bias[kBiasX10 + ch] = shift[9][ch];
        for (int co = 0; co < 16; ++co) bias[kLogitW + co] = c.logit_w[co];
    }

    uint16_t* stream = reinterpret_cast<uint16_t*>(image + kOffStream);
    size_t group = 0;
    auto slot = [&](int lane, int j) -> uint16_t& { return stream[group * 512 + lane * 8 + j]; };

    // 1. input convs: conv, tap, tile.  Row rho of a tile holds local channel 32*tile + 2t + hr with
    //    t = (rho&3) + 4*(rho>>3), hr = (rho>>2)&1, so that accumulator register t of lane half hr
    //    is that channel (bf_layout_bf16.h).
    for (int i = 0; i < kNumInputConvs; ++i) {
        const InputConv& cv = kInputConvs[i];
        for (int tap = 0; tap < cv.k * cv.k; ++tap)
            for (int tile = 0; tile < cv.tiles; ++tile) {
                for (int lane = 0; lane < 64; ++lane) {
                    const int rho = lane & 31, h = lane >> 5;
                    const int t = (rho & 3) + 4 * (rho >> 3), hr = (rho >> 2) & 1;
                    const int ch = 32 * tile + 2 * t + hr;
                    const bool ones_row = (i == 4 || i == 5) && (ch == 18 || ch == 19);
                    for (int j = 0; j < 8; ++j) {
                        const int plane = 8 * h + j;
                        uint16_t v = 0;
                        if (ch < cv.ch) {
                            if (plane < kPlanes) {
                                v = to_bf16(c.in_w[i][((size_t)tap * kPlanes + plane) * cv.ch + ch] * inv[0][cv.chan_base + ch]);
                            } else if (tap < 3) {
                                uint16_t parts[3];
                                split3(shift[0][cv.chan_base + ch], parts);
                                v = parts[tap];
                            }
                        } else if (ones_row && plane == 15 && tap == 0) {
                            v = to_bf16(1.0f);
                        }
                        slot(lane, j) = v;
                    }
                }
                ++group;
            }
    }
    // 2. the 1x1 pair per 32-channel hidden tile
    auto x1_channel = [&](int reg, int h) -> int {  // global x1 channel of register `reg`, lane half h; -1 = none
        for (int i = 0; i < kNumInputConvs; ++i)
            if (reg >= kX1RegBase[i] && reg < kX1RegBase[i] + kX1RegCount[i]) {
                int cl = 2 * (reg - kX1RegBase[i]) + h;
                return cl < kInputConvs[i].ch ? kInputConvs[i].chan_base + cl : -1;
            }
        return -1;
    };
    auto write_g1 = [&](int c2, int s) {
        for (int lane = 0; lane < 64; ++lane) {
            const int h = lane >> 5, co = c2 * 32 + (lane & 31);
            uint16_t parts[3];
            split3(shift[1][co], parts);
            for (int j = 0; j < 8; ++j) {
                const int reg = 8 * s + j;
                const int ci = x1_channel(reg, h);
                uint16_t v = 0;
                if (ci >= 0) v = to_bf16(c.tower_w[0][(size_t)ci * kC2 + co] * inv[1][co]);
                else if (reg == kOnesRegA) v = parts[h];            // hi on half 0, mid on half 1
                else if (reg == kOnesRegB && h == 0) v = parts[2];  // lo
                slot(lane, j) = v;
            }
        }
    };
    auto write_g2 = [&](int c2, int i) {
        const int s = i >> 2, c3 = i & 3;
        for (int lane = 0; lane < 64; ++lane) {
            const int h = lane >> 5, co = c3 * 32 + (lane & 31);
            for (int j = 0; j < 8; ++j) {
                const int ci = c2 * 32 + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
                slot(lane, j) = to_bf16(c.tower_w[1][(size_t)ci * kC3 + co] * inv[2][co]);
            }
        }
    };
    for (int p = 0; p < kC2Tiles * kPerC2Groups; ++p) {   // in the order the kernel takes them (kGemmOrder)
        if (kGemmOrder.kind[p] == 0) write_g1(kGemmOrder.tile[p], kGemmOrder.idx[p]);
        else if (kGemmOrder.kind[p] == 1) write_g2(kGemmOrder.tile[p], kGemmOrder.idx[p]);
        ++group;   // kind 2: padding group, zeros
    }
    // 3. 3x3 layers: natural channel order, k = 16s + 8h + j within a 32-channel chunk
    auto conv3 = [&](const float* w, int cin, int chunk, const std::vector<float>& scale) {
        for (int tap = 0; tap < 9; ++tap)
            for (int s = 0; s < 2; ++s) {
                for (int lane = 0; lane < 64; ++lane) {
                    const int h = lane >> 5, co = lane & 31;
                    for (int j = 0; j < 8; ++j) {
                        const int ci = chunk * 32 + 16 * s + 8 * h + j;
                        slot(lane, j) = to_bf16(w[((size_t)tap * cin + ci) * 32 + co] * scale[co]);
                    }
                }
                ++group;
            }
    };
    for (int q = 0; q < 4; ++q) conv3(c.tower_w[2], 128, q, inv[3]);
    for (int l = 0; l < 5; ++l) conv3(c.tower_w[3 + l], 32, 0, inv[4 + l]);

    if (policy) {
        // no 8x8 layer: the head region carries the logit kernel, float32 [filter][channel]
        float* lw = reinterpret_cast<float*>(image + kOffHead);
        for (int f = 0; f < kPolicyFilters; ++f)
            for (int ci = 0; ci < 32; ++ci) lw[f * 32 + ci] = c.logit_w[ci * kPolicyFilters + f];
        return;
    }
    // ---- head: one 16x16x32 A fragment per square ------------------------------------
    uint16_t* head = reinterpret_cast<uint16_t*>(image + kOffHead);
    for (int sq = 0; sq < 64; ++sq)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int co = lane & 15, ci = 8 * (lane >> 4) + j;
                head[((size_t)sq * 64 + lane) * 8 + j] = to_bf16(c.tower_w[8][((size_t)sq * 32 + ci) * 16 + co] * inv[9][co]);
            }
}

}  // namespace b16

void pack_weights_bf16(const float* blob, unsigned char* image, bool policy) { b16::pack(blob, image, policy); }

}  // namespace bf

extern "C" size_t bf_packed_bf16_bytes(void) { return bf::b16::kImageBytes; }

extern "C" int bf_pack_weights_bf16(const float* h_weights, size_t n_floats, void* h_image, size_t n_bytes) {
    if (!h_weights || !h_image) return bf::fail(BF_E_ARG, "bf_pack_weights_bf16: null pointer");
    if (n_floats != bf::kWeightFloats)
        return bf::fail(BF_E_ARG, "bf_pack_weights_bf16: expected %zu weight floats, got %zu", bf::kWeightFloats, n_floats);
    if (n_bytes != bf::b16::kImageBytes)
        return bf::fail(BF_E_ARG, "bf_pack_weights_bf16: expected room for %zu bytes, got %zu", bf::b16::kImageBytes, n_bytes);
    bf::pack_weights_bf16(h_weights, static_cast<unsigned char*>(h_image), false);
    return BF_OK;
}
