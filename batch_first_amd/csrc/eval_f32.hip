// The value network as one fp32 MFMA kernel: 4-bit mailboxes in, one score per board out.
//
// Reference graph replaced (TensorFlow ops -> this kernel):
//   create_input_convolutions_shared_weights   ann_creation_helper.py:192-252  (stage 1)
//   build_convolutional_modules, spec at       evaluation_ann.py:268           (stages 2-4)
//   logit_layer + squeeze                      evaluation_ann.py:35-46         (stage 5)
//
// Shape of the computation.  One workgroup = one board = two waves; wave `half` owns squares
// 32*half .. 32*half+31 as the 32 columns of every MFMA tile.  All layers run transposed,
//     out^T[channel][square] += W^T[channel][k] * in^T[k][square]      (v_mfma_f32_32x32x2_f32)
// so that an accumulator tile (rows = channels in registers, columns = squares on lanes) is
// directly the B operand of the next 1x1 layer: nothing but the 3x3 layers' halo exchange ever
// leaves the register file, and no intermediate activation touches HBM.
//   stage 1  nine dilated input convs as implicit GEMM: B = one-hot of the shifted square's
//            code built in registers from a padded byte mailbox in LDS; 236 channels end up
//            in 118 registers (channel 2r on lanes 0-31, 2r+1 on lanes 32-63).
//   stage 2  1x1 236->512 and 512->128 chained per 32-channel tile of the hidden layer: the
//            512-wide activation never exists; 128 outputs accumulate in 4 tiles.
//   stage 3  3x3 128->32: each 32-channel tile goes through a zero-haloed 10x10 LDS plane,
//            the B operand is a shifted ds_read.
//   stage 4  five 3x3 32->32 layers the same way.
//   stage 5  8x8 valid conv (2048->16 dense) + ReLU + logit on the VALU, fixed-order reduction.
// Weights arrive as a linear stream of A fragments in consumption order (bf_layout.h, pack.cpp),
// read with one 16-byte load per lane per 4 k-steps through a software prefetch ring; both
// waves of a board read the same stream (second reader hits L1/L2).  Batch-norm is folded into
// the weights; its shift initialises the accumulators, so every epilogue is a bare max(x, 0).
//
// Cost model: 4848 MFMAs (64 cycles each on one SIMD) per wave and board half; everything else
// is hidden under them at two waves per SIMD.  Results are bit-reproducible run to run (fixed
// summation order, no atomics).
#include <hip/hip_runtime.h>

#include "bf_device.h"
#include "bf_kernels.h"
#include "bf_layout.h"

namespace bf {

namespace {

#ifdef BF_ABL_NOBARRIER
#define BF_CONV_SYNC() ((void)0)
#else
#define BF_CONV_SYNC() __syncthreads()
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define BF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

#ifndef BF_RING
#define BF_RING 4
#endif
constexpr int RING = BF_RING;              // prefetch depth in groups (1 KiB per wave each)
constexpr int C2_UNROLL = RING / 2;   // hidden tiles per loop body; 46 * C2_UNROLL % RING == 0
static_assert((kPerC2Groups * C2_UNROLL) % RING == 0, "loop body must preserve the ring phase");
static_assert(kInputGroups % RING == 0, "ring phase at loop entry");
static_assert(kC2Tiles % C2_UNROLL == 0, "");
static_assert(kStreamPadGroups >= RING, "prefetch overrun must stay inside the image");

constexpr int MB_W = 22;     // mailbox padded by 7 squares (largest dilation) on every side
constexpr int MB_BYTES = 496;
constexpr int XP_W = 10;     // activation plane padded by 1 for the 3x3 layers
constexpr int XP_PLANE = 100;
constexpr int XBUF_FLOATS = 32 * XP_PLANE;
constexpr int RED_STRIDE = 17;

struct Lds {
    float xbuf[2][XBUF_FLOATS];
    float red[128 * RED_STRIDE];
    unsigned char mb[MB_BYTES];
};

// Software prefetch ring over the A-fragment stream.
struct WStream {
    const f32x4* p;  // this lane's slot in the next group to load
    f32x4 ring[RING];
    __device__ __forceinline__ void prime(const f32x4* base, int lane) {
        p = base + lane;
#pragma unroll
        for (int i = 0; i < RING; ++i) {
            ring[i] = *p;
            p += 64;
        }
    }
    // g must be a compile-time constant after unrolling; groups are taken in order.
    __device__ __forceinline__ f32x4 take(int g) {
        const f32x4 v = ring[g % RING];
        ring[g % RING] = *p;
#ifndef BF_ABL_NOSTREAM
        p += 64;
#endif
        return v;
    }
};

constexpr int input_group_base(int conv) {
    int g = 0;
    for (int i = 0; i < conv; ++i) g += kInputConvs[i].k * kInputConvs[i].k * kInputConvs[i].tiles * kInputSplits;
    return g;
}

__device__ __forceinline__ float pick(int hl, float lo, float hi) { return hl ? hi : lo; }

// Stage 1, one input conv.  reference: ann_creation_helper.py:211-236 (SAME padding:
// (k-1)*d in total, floor(/2) before).
template <int I, bool DBG>
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
#pragma unroll
    for (int tap = 0; tap < cv.k * cv.k; ++tap) {
        const int dr = (tap / cv.k) * cv.dh - ph, df = (tap % cv.k) * cv.dw - pw;
        // B fragment: one-hot over planes 8*hl .. 8*hl+7 of the shifted square's code, as 8 bf16
        const unsigned cc = (unsigned)((int)mb_center[dr * MB_W + df] - 8 * hl);
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
            if (32 * tile + 2 * t < cv.ch) x1[cv.chan_base / 2 + 16 * tile + t] = fmaxf(acc[tile][t], 0.0f);
}

// One 3x3 SAME conv over a 32-channel chunk staged in a zero-haloed LDS plane:
// acc[co][sq] += sum_tap sum_ci W[tap][ci][co] * x[ci][sq + tap offset].
// g0 must be a compile-time constant at every (inlined) call site.
__device__ __forceinline__ void conv3x3_chunk(WStream& ws, int g0, const float* xb_lane, f32x16& acc) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int off = (tap / 3) * XP_W + (tap % 3);  // xb_lane already points one row and column back
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            const f32x4 w = ws.take(g0 + tap * 4 + sg);
            acc = BF_MFMA(w.x, xb_lane[(8 * sg + 0) * XP_PLANE + off], acc);
            acc = BF_MFMA(w.y, xb_lane[(8 * sg + 2) * XP_PLANE + off], acc);
            acc = BF_MFMA(w.z, xb_lane[(8 * sg + 4) * XP_PLANE + off], acc);
            acc = BF_MFMA(w.w, xb_lane[(8 * sg + 6) * XP_PLANE + off], acc);
        }
    }
}

template <bool DBG>
__global__ __launch_bounds__(128, 2) void k_eval_f32(const uint8_t* __restrict__ mailbox, unsigned n_boards,
                                                      const f32x4* __restrict__ wstream,
                                                      const float* __restrict__ bias,
                                                      const f32x4* __restrict__ headw,
                                                      float* __restrict__ scores, float* __restrict__ dbg) {
    __shared__ __attribute__((aligned(16))) Lds lds;
    const unsigned board = blockIdx.x;
    if (board >= n_boards) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, hl = lane >> 5, ln = lane & 31;
    const int half = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sq = 32 * half + ln, rank = sq >> 3, file = sq & 7;

    WStream ws;
    ws.prime(wstream, lane);

    // ---- LDS set-up: empty mailbox (0xFF everywhere), zero activation planes ------------
    if (tid < MB_BYTES / 4) reinterpret_cast<unsigned*>(lds.mb)[tid] = 0xFFFFFFFFu;
    {
        f32x4* z = reinterpret_cast<f32x4*>(&lds.xbuf[0][0]);
        for (int i = tid; i < 2 * XBUF_FLOATS / 4; i += 128) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (tid < 64) {
        const unsigned v = mailbox[(size_t)board * kMailboxBytes + (tid >> 1)];
        const unsigned c = (tid & 1) ? (v >> 4) : (v & 15u);
        lds.mb[((tid >> 3) + 7) * MB_W + (tid & 7) + 7] = (unsigned char)(c == 15u ? 0xFFu : c);
    }
    __syncthreads();

    float* dbg_board = DBG ? dbg + (size_t)board * kDebugFloatsPerBoard : nullptr;

    // ---- stage 1: input convs -> x1 (118 registers) -------------------------------------
    float x1[kX1Regs];
    {
        const unsigned char* mbc = &lds.mb[(rank + 7) * MB_W + file + 7];
        input_conv<0, DBG>(ws, mbc, hl, bias, x1);
        input_conv<1, DBG>(ws, mbc, hl, bias, x1);
        input_conv<2, DBG>(ws, mbc, hl, bias, x1);
        input_conv<3, DBG>(ws, mbc, hl, bias, x1);
        input_conv<4, DBG>(ws, mbc, hl, bias, x1);
        input_conv<5, DBG>(ws, mbc, hl, bias, x1);
        input_conv<6, DBG>(ws, mbc, hl, bias, x1);
        input_conv<7, DBG>(ws, mbc, hl, bias, x1);
        input_conv<8, DBG>(ws, mbc, hl, bias, x1);
    }
    if (DBG) {
#pragma unroll
        for (int r = 0; r < kX1Regs; ++r) dbg_board[(2 * r + hl) * 64 + sq] = x1[r];
    }

    // ---- stage 2: 1x1 236->512 (+ReLU) chained into 1x1 512->128 -------------------------
    f32x16 acc3[4];
#pragma unroll
    for (int c3 = 0; c3 < 4; ++c3)
#pragma unroll
        for (int t = 0; t < 16; ++t)
            acc3[c3][t] = pick(hl, bias[kBiasX3 + 32 * c3 + rowmap(t, 0)], bias[kBiasX3 + 32 * c3 + rowmap(t, 1)]);

#pragma unroll 1
    for (int it = 0; it < kC2Tiles / C2_UNROLL; ++it) {
#pragma unroll
        for (int u = 0; u < C2_UNROLL; ++u) {
            const int c2 = it * C2_UNROLL + u;
            const int g0 = kInputGroups + u * kPerC2Groups;
            const float* b2 = bias + kBiasX2 + 32 * c2;
            f32x16 acc2;
#pragma unroll
            for (int t = 0; t < 16; ++t) acc2[t] = pick(hl, b2[rowmap(t, 0)], b2[rowmap(t, 1)]);
#pragma unroll
            for (int g = 0; g < kGemm1Groups; ++g) {
                const f32x4 w = ws.take(g0 + g);
                acc2 = BF_MFMA(w.x, x1[4 * g + 0], acc2);
                acc2 = BF_MFMA(w.y, x1[4 * g + 1], acc2);
                if (4 * g + 2 < kX1Regs) {
                    acc2 = BF_MFMA(w.z, x1[4 * g + 2], acc2);
                    acc2 = BF_MFMA(w.w, x1[4 * g + 3], acc2);
                }
            }
#pragma unroll
            for (int t = 0; t < 16; ++t) acc2[t] = fmaxf(acc2[t], 0.0f);
#pragma unroll
            for (int c3 = 0; c3 < 4; ++c3)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    const f32x4 w = ws.take(g0 + kGemm1Groups + c3 * 4 + tq);
                    acc3[c3] = BF_MFMA(w.x, acc2[4 * tq + 0], acc3[c3]);
                    acc3[c3] = BF_MFMA(w.y, acc2[4 * tq + 1], acc3[c3]);
                    acc3[c3] = BF_MFMA(w.z, acc2[4 * tq + 2], acc3[c3]);
                    acc3[c3] = BF_MFMA(w.w, acc2[4 * tq + 3], acc3[c3]);
                }
        }
    }
    if (DBG) {
#pragma unroll
        for (int c3 = 0; c3 < 4; ++c3)
#pragma unroll
            for (int t = 0; t < 16; ++t)
                dbg_board[64 * 236 + (32 * c3 + rowmap(t, hl)) * 64 + sq] = fmaxf(acc3[c3][t], 0.0f);
    }

    // ---- stages 3 and 4: the six 3x3 layers through LDS -----------------------------------
    constexpr int GC = kInputGroups;  // static ring phase after the loop (the loop preserves it)
    const int wpos = (rank + 1) * XP_W + file + 1;           // where this lane's square lives
    const int rpos = hl * XP_PLANE + rank * XP_W + file;     // one row and column back, plane hl
    f32x16 acc;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = pick(hl, bias[kBiasX4 + rowmap(t, 0)], bias[kBiasX4 + rowmap(t, 1)]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float* xb = lds.xbuf[q & 1];
#pragma unroll
        for (int t = 0; t < 16; ++t) xb[rowmap(t, hl) * XP_PLANE + wpos] = fmaxf(acc3[q][t], 0.0f);
        BF_CONV_SYNC();
        conv3x3_chunk(ws, GC + q * 36, xb + rpos, acc);
    }
#pragma unroll
    for (int layer = 0; layer < 5; ++layer) {
        float* xb = lds.xbuf[layer & 1];  // continues the 0,1,0,1 alternation of the four chunks above
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float v = fmaxf(acc[t], 0.0f);
            xb[rowmap(t, hl) * XP_PLANE + wpos] = v;
            if (DBG) dbg_board[64 * (236 + 128 + 32 * layer) + rowmap(t, hl) * 64 + sq] = v;
        }
        BF_CONV_SYNC();
        const float* bl = bias + kBiasX4 + 32 * (layer + 1);
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = pick(hl, bl[rowmap(t, 0)], bl[rowmap(t, 1)]);
        conv3x3_chunk(ws, GC + 144 + layer * 36, xb + rpos, acc);
    }

    // ---- stage 5: 8x8 valid conv (2048 -> 16), ReLU, logit -----------------------------
    float part[16];
#pragma unroll
    for (int co = 0; co < 16; ++co) part[co] = 0.0f;
    const f32x4* hw = headw + (size_t)half * 64 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float v = fmaxf(acc[t], 0.0f);
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
#pragma unroll
    for (int co = 0; co < 16; ++co) lds.red[tid * RED_STRIDE + co] = part[co];
    __syncthreads();
    if (tid < 64) {
        const int co = tid & 15, seg = tid >> 4;
        float s = 0.0f;
        for (int j = 0; j < 32; ++j) s += lds.red[(seg * 32 + j) * RED_STRIDE + co];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float y = fmaxf(s + bias[kBiasX10 + co], 0.0f);
        if (DBG && tid < 16) dbg_board[64 * (236 + 128 + 32 * 6) + co] = y;
        float v = y * bias[kLogitW + co];  // logit_layer: no bias, no batch-norm (evaluation_ann.py:35-44)
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        if (tid == 0) scores[board] = v;
    }
}

}  // namespace

hipError_t launch_eval_f32(const uint8_t* d_mailbox, size_t n, const float* d_packed, float* d_scores,
                           float* d_debug, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const f32x4* ws = reinterpret_cast<const f32x4*>(d_packed + kOffStream);
    const float* bias = d_packed + kOffBias;
    const f32x4* head = reinterpret_cast<const f32x4*>(d_packed + kOffHead);
    if (d_debug)
        hipLaunchKernelGGL(k_eval_f32<true>, dim3((unsigned)n), dim3(128), 0, stream, d_mailbox, (unsigned)n, ws,
                           bias, head, d_scores, d_debug);
    else
        hipLaunchKernelGGL(k_eval_f32<false>, dim3((unsigned)n), dim3(128), 0, stream, d_mailbox, (unsigned)n, ws,
                           bias, head, d_scores, d_debug);
    return hipGetLastError();
}

}  // namespace bf
