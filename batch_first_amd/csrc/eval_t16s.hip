// The 16x16x32 bf16 network for SMALL CALLS: one board per workgroup, one 16-square column tile per wave.
//
// eval_t16.hip gives every wave a whole board (four column tiles), which is the densest form when there are
// thousands of boards; a call of a hundred boards then waits for ONE wave's walk through all 629 weight
// fragments x 4 tiles (25-28 us) while three quarters of the chip idle.  Here the four waves of a workgroup
// take the four column tiles of the same board: the 1x1 layers need nothing from the other tiles, the input
// convs read one shared one-hot image, and the 3x3 layers exchange their tiles through one shared pair of
// activation planes with a workgroup barrier per 32-channel chunk.  Weight image, fragment order, accumulation
// order and every rounding are those of eval_t16.hip (bf_layout_t16.h): the scores are bit-identical, which
// tests/test_gpu_bf16.py checks for every entry point.  A k-step of the input convs that eval_t16.hip skips
// for a column tile (all its taps off the board) is computed here (the tile index is a run-time value): it
// multiplies zeros, and the ones-plane weights of such taps are zero (bf_layout_t16.h), so the sums are equal.
//
// Reference graph replaced: as eval_t16.hip (ann_creation_helper.py:192-252, evaluation_ann.py:268, :35-46).
#include <hip/hip_runtime.h>

#include "bf_device.h"
#include "bf_kernels.h"
#include "bf_layout_t16.h"
#include "bf_split.h"
#include "t16_common.h"

namespace bf {

namespace t16 {

namespace {

struct LdsS {
    f32x4 wring[SLOTS][CHUNK][64];                       // shared A-fragment ring, 64 KiB
    __attribute__((aligned(16))) unsigned char image[SG_BYTES];
    __attribute__((aligned(16))) unsigned char plane[2][PLANE_BYTES];   // zero-haloed [10 x 10 positions][32 channels] bf16: x4..x9
    __attribute__((aligned(16))) unsigned char x3plane[4][PLANE_BYTES]; // the four 32-channel chunks of x3, same form
    u32x4 onehot[16][2];
    __attribute__((aligned(16))) float bias3[6 * 32];
    __attribute__((aligned(16))) float bias2[128];
    f32x4 hpart[WAVES][64];
    __attribute__((aligned(16))) unsigned char next_mailbox[WAVES][256];   // every wave prefetches its own copy
    unsigned wire_window[WAVES][16];                     // wire input: the 40 stream bytes that hold the board's codes
};

// Wire-format input (MODE_ + 8; the host entry points' staging, api.hip: wire_to_mailbox): the board's relevance
// mask, the number of codes in front of it, and the nibble stream.  The decode is k_to_mailbox<WireSrc>'s
// (encode.hip; ann_creation_helper.py:55-79), done by the waves that evaluate the board: one launch a call.
struct WireArgs {
    const uint64_t* occ;
    const unsigned* offsets;     // codes of the boards before this one
    const uint8_t* nib;          // 4-byte aligned, at least 3 readable bytes behind n_bytes
    unsigned n_bytes;
    unsigned start_nibble;
    int* err;                    // sticky malformed-input flag: 1 = stream too short, 2 = code 15
};

// MODE 0: scores; 2: features (x9 as float32 [board][32][64] into `out`); + 4: node records instead of mailboxes;
// + 8: wire format (`wire`; src and src_stride unused).
template <int MODE_>
__global__ __launch_bounds__(THREADS, 1) void k_eval_t16s(const uint8_t* __restrict__ src, unsigned n_boards,
                                                           const f32x4* __restrict__ wstream,
                                                           const float* __restrict__ bias,
                                                           const f32x4* __restrict__ headw, float* __restrict__ out,
                                                           unsigned src_stride, WireArgs wire) {
    constexpr int MODE = MODE_ & 3;
    constexpr bool RECS = (MODE_ & 4) != 0;
    constexpr bool WIRE = (MODE_ & 8) != 0;
    static_assert(!(RECS && WIRE), "one input format");
    constexpr int SRC_DWORDS = WIRE ? 3 : RECS ? 22 : 8;   // wire: mask low, mask high, offset
    __shared__ __attribute__((aligned(16))) LdsS lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63, c16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = this wave's column tile: ranks 2 * wave, 2 * wave + 1
    const int rk = c16 >> 3, file = c16 & 7;
    typedef __attribute__((address_space(3))) unsigned char* lds_bytes;
    typedef const __attribute__((address_space(3))) u32x4* lds_entry;

#ifdef BF_STAMPS   // diagnostics build (tools/stamps_t16s.py): s_memtime at the stage boundaries of workgroup 0's last board
    unsigned stamps[16] = {};
    stamps[8] = (unsigned)__builtin_amdgcn_s_memtime();
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
    // every global load first (one latency, not one per table), the LDS zeroing while they are in flight
    static_assert(6 * 32 <= THREADS && 128 <= THREADS, "one shift per thread");
    const float shift3 = tid < 6 * 32 ? bias[kBiasX4 + tid] : 0.0f;
    const float shift2 = tid < 128 ? bias[kBiasX3 + tid] : 0.0f;
    // the first board's mailbox directly (every later one is prefetched)
    // wire input: what a lane adds to the mask array's base to reach its dword of board b's (mask, offset): 8 b + 4 lane
    // for the mask's halves, the distance to the offsets + 4 b for every other lane
    const unsigned wire_lane_off = WIRE ? (lane < 2 ? 4u * (unsigned)lane
                                                    : (unsigned)(reinterpret_cast<const unsigned char*>(wire.offsets) -
                                                                 reinterpret_cast<const unsigned char*>(wire.occ)))
                                        : 0u;
    const unsigned wire_lane_mul = lane < 2 ? 8u : 4u;
    const unsigned first_dword =
        WIRE ? *reinterpret_cast<const unsigned*>(reinterpret_cast<const unsigned char*>(wire.occ) + wire_lane_off +
                                                  wire_lane_mul * blockIdx.x)
             : reinterpret_cast<const unsigned*>(src + (size_t)blockIdx.x * src_stride)[lane < SRC_DWORDS ? lane : SRC_DWORDS - 1];
    u32x4 hk[16];   // the 8x8 kernels of this wave's sixteen squares
    if (MODE != 2) {
        const f32x4* hwp = headw + (size_t)wave * 16 * 64 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) hk[i] = __builtin_bit_cast(u32x4, hwp[i * 64]);
    }
    const float head_shift = MODE != 2 ? bias[kBiasX10 + (tid & 15)] : 0.0f;
    const float head_logit = MODE != 2 ? bias[kLogitW + (tid & 15)] : 0.0f;
    for (int i = tid; i < SG_BYTES / 16; i += THREADS) reinterpret_cast<u32x4*>(lds.image)[i] = u32x4{0u, 0u, 0u, 0u};
    for (int i = tid; i < 6 * PLANE_BYTES / 16; i += THREADS) reinterpret_cast<u32x4*>(lds.plane[0])[i] = u32x4{0u, 0u, 0u, 0u};
    if (tid < 6 * 32) lds.bias3[tid] = shift3;
    if (tid < 128) lds.bias2[tid] = shift2;
    if (lane < SRC_DWORDS) reinterpret_cast<unsigned*>(lds.next_mailbox[wave])[lane] = first_dword;
    lds_order();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ws.start();   // (its first chunk barrier also orders the zeroing above before any wave's first use)

    // lane constants: this lane's square in the image and in the planes (eval_t16.hip), at this wave's tile
    const int my_rank = 2 * wave + rk;
    const unsigned image0 = (unsigned)(uintptr_t)(lds_bytes)lds.image;
    const unsigned grid0 = image0 + (my_rank * SG_COLS + file) * SG_ENTRY + (q & 1) * 16;
    const int waddr = ((my_rank + 1) * XP_W + file + 1) * POS_BYTES + (((q + file + 1) & 3) << 4);
    int raddr[3];   // shifted reads df = -1, 0, +1, one padded row above the square
#pragma unroll
    for (int d = 0; d < 3; ++d) raddr[d] = (my_rank * XP_W + file + d) * POS_BYTES + (((q + file + d) & 3) << 4);
    // the square whose image entry this lane writes per board: lanes 0..31 = (square 16 * wave + (lane & 15), half lane >> 4)
    const int wsq = 16 * wave + c16;
    unsigned char* my_entry = lds.image + (((wsq >> 3) + SG_ROW0) * SG_COLS + (wsq & 7) + SG_COL0) * SG_ENTRY + (q & 1) * 16;
    const bool q2 = q >= 2;
    const unsigned ones = (q & 1) ? 0x3F800000u : 0u;

#pragma unroll 1
    for (unsigned board = blockIdx.x; board < n_boards; board += gridDim.x) {
    const unsigned next_raw = board + gridDim.x;
    const unsigned next_board = __builtin_amdgcn_readfirstlane(next_raw < n_boards ? next_raw : n_boards - 1);
    ws.board_start();
    BF_STAMP(0);

    // ---- this wave's sixteen squares of the one-hot image ----------------------------------------------
    if (WIRE) {   // the (at most 33) stream bytes of this board's codes, 4-byte aligned window of 40
        const unsigned* nm = reinterpret_cast<const unsigned*>(lds.next_mailbox[wave]);
        const unsigned first_nibble = wire.start_nibble + nm[2];
        const unsigned wb = (first_nibble >> 1) & ~3u;
        const unsigned at = wb + 4u * (unsigned)(lane < 10 ? lane : 9);
        unsigned wv = 0u;
        if (at < wire.n_bytes) wv = *reinterpret_cast<const unsigned*>(wire.nib + at);
        if (lane < 10) lds.wire_window[wave][lane] = wv;
        lds_order();
    }
    if (lane < 32) {
        unsigned c;
        if (WIRE) {
            const unsigned* nm = reinterpret_cast<const unsigned*>(lds.next_mailbox[wave]);
            const unsigned long long m = ((unsigned long long)nm[1] << 32) | nm[0];
            const unsigned first_nibble = wire.start_nibble + nm[2];
            const unsigned wb = (first_nibble >> 1) & ~3u;
            c = 15u;
            if ((m >> wsq) & 1ull) {
                const unsigned nibble = first_nibble + (unsigned)__popcll(m & ((1ull << wsq) - 1ull));
                const unsigned byte = nibble >> 1;
                if (byte >= wire.n_bytes) {   // never read past the stream; the length check reports it
                    atomicOr(wire.err, 1);
                } else {
                    // (byte - wb <= 3 + 32 < 40: a mask has at most 64 bits)
                    const unsigned v = reinterpret_cast<const unsigned char*>(lds.wire_window[wave])[byte - wb];
                    c = (nibble & 1u) ? (v & 15u) : (v >> 4);   // even nibble = high half
                    if (c == 15u) atomicOr(wire.err, 2);
                }
            }
        } else if (RECS) {
            const BoardBits bits = load_board(lds.next_mailbox[wave]);
            c = (unsigned)square_code(bits, relevance_mask(bits), wsq);
        } else {
            const unsigned v = lds.next_mailbox[wave][wsq >> 1];
            c = (wsq & 1) ? (v >> 4) : (v & 15u);
        }
        *reinterpret_cast<u32x4*>(my_entry) = lds.onehot[c][q & 1];
    }
    lds_barrier();
    BF_STAMP(1);

    // ---- stage 1 ------------------------------------------------------------------------------------
    unsigned X[kX1Dwords];
    {
        constexpr int BQ = 4;
        u32x4 b[BQ];
        f32x4 acc[2][4];   // [conv parity][row tile]
        unsigned base = grid0;
        auto read_b = [&](int u) -> u32x4 {
            const int uu = u < kUnits ? u : kUnits - 1;
            if (u < kUnits && kS1s.new_base[uu]) {
                base = grid0 + (unsigned)(q2 ? kS1s.db[uu] : kS1s.da[uu]);
                asm volatile("" : "+v"(base));
            }
            return *(lds_entry)(uintptr_t)(base + (unsigned)kS1s.imm[uu]);
        };
        auto pack_conv = [&](int i) {
#pragma unroll
            for (int d = kX1RegBase[i] / 2; d < (kX1RegBase[i] + kX1RegCount[i]) / 2; ++d) {
                const X1Slot a = x1_slot(2 * d), bb = x1_slot(2 * d + 1);
                unsigned p = pack_relu(acc[a.conv & 1][a.tile][a.r], acc[bb.conv & 1][bb.tile][bb.r]);
                asm volatile("" : "+v"(p));
                X[d] = p;
            }
        };
#pragma unroll
        for (int u = 0; u < BQ; ++u) b[u] = read_b(u);
        b[0].w |= ones;
#pragma unroll
        for (int u = 0; u < kUnits; ++u) {
            const int I = kS1s.conv[u], par = I & 1;
            bf16x8 fr[4];
#pragma unroll
            for (int tile = 0; tile < 4; ++tile)
                if (tile < kRowTiles[I]) fr[tile] = ws.take(kS1s.g0[u] + tile);
            const bf16x8 bu = __builtin_bit_cast(bf16x8, b[u % BQ]);
#pragma unroll
            for (int tile = 0; tile < 4; ++tile)
                if (tile < kRowTiles[I])
                    acc[par][tile] = MFMA16(fr[tile], bu, kS1s.first_of_conv[u] ? zero4() : acc[par][tile]);
            b[u % BQ] = read_b(u + BQ);
            if (u + 1 < kUnits) b[(u + 1) % BQ].w |= ones;
            // the previous conv's channels, one unit after its last (its accumulators are the other parity's)
            if (kS1s.first_of_conv[u] && I > 0) pack_conv(I - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        pack_conv(kNumInputConvs - 1);
    }

    BF_STAMP(2);
    // ---- stage 2: 1x1 236->512 (+ReLU) chained into 1x1 512->128, this wave's column tile -----------------
    f32x4 acc3[kOutTiles];
    {
        f32x4 acc2[2][2];
        unsigned H[4];
        f32x4 seed[kOutTiles];
#pragma unroll
        for (int p = 0; p < kPairGroups; ++p) {
            const int g = kPairBase + p;
            const int kind = kPairOrder.kind[p], a = kPairOrder.a[p], bi = kPairOrder.b[p];
            const bf16x8 w = ws.take(g);
            if (g == kPrefetchGroup)
                if (WIRE)
                    ws.prefetch_mailbox(reinterpret_cast<const unsigned char*>(wire.occ),
                                        (unsigned)(uintptr_t)(lds_bytes)lds.next_mailbox[wave], wire_lane_off + wire_lane_mul * next_board);
                else
                    ws.prefetch_mailbox(src + (size_t)next_board * src_stride, (unsigned)(uintptr_t)(lds_bytes)lds.next_mailbox[wave],
                                        (unsigned)(lane < SRC_DWORDS ? lane : SRC_DWORDS - 1) * 4u);
            if (p >= 23 && p < 23 + 2 * kOutTiles && ((p - 23) & 1) == 0)
                seed[(p - 23) >> 1] = *reinterpret_cast<const f32x4*>(&lds.bias2[16 * ((p - 23) >> 1) + 4 * q]);
            if (kind == 0) {
                const int par = (a >> 1) & 1;
                const u32x4 x = u32x4{X[4 * bi], X[4 * bi + 1], X[4 * bi + 2], X[4 * bi + 3]};
                acc2[par][a & 1] = MFMA16(w, __builtin_bit_cast(bf16x8, x), bi == 0 ? zero4() : acc2[par][a & 1]);
            } else {
                const u32x4 h = u32x4{H[0], H[1], H[2], H[3]};
                acc3[bi] = MFMA16(w, __builtin_bit_cast(bf16x8, h), a == 0 ? seed[bi] : acc3[bi]);
            }
            // the hidden pair h is packed under the 236->512 groups of pair h + 1, two groups after its last MFMA
            auto pack_pair = [&](int par) {
#pragma unroll
                for (int d = 0; d < 4; ++d) H[d] = pack_relu(acc2[par][d >> 1][2 * (d & 1)], acc2[par][d >> 1][2 * (d & 1) + 1]);
            };
            if (p >= 16 && p < 16 + 24 * 15 && (p - 16) % 24 == 2) pack_pair(((p - 16) / 24) & 1);
            if (p == 16 + 24 * 15 - 1) pack_pair(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    BF_STAMP(3);
    // ---- stages 3 and 4: the six 3x3 layers, tiles exchanged through the shared planes ---------------------
    auto write_tile = [&](unsigned char* pl, const f32x4& a0, const f32x4& a1) {
        u32x4 v;
        v.x = pack_relu(a0[0], a0[1]);
        v.y = pack_relu(a0[2], a0[3]);
        v.z = pack_relu(a1[0], a1[1]);
        v.w = pack_relu(a1[2], a1[3]);
        lds_order();
        *reinterpret_cast<u32x4*>(pl + waddr) = v;
        lds_order();
    };
    auto load_taps = [&](u32x4 (&dst)[9], const unsigned char* pl) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            dst[tap] = *reinterpret_cast<const u32x4*>(pl + raddr[tap % 3] + (tap / 3) * ROW_BYTES);
    };
    f32x4 A[2];   // the two output row tiles of this wave's column tile
    f32x4 bs[2];
    u32x4 bq[2][9];
    // Chunks 0..3 are the four 32-channel input chunks of the 128->32 layer: all of x3 goes into its own four planes
    // first (last read four barriers ago), so that one barrier covers them and their 72 groups run without a stop.
#pragma unroll
    for (int c = 0; c < 4; ++c) write_tile(lds.x3plane[c], acc3[2 * c], acc3[2 * c + 1]);
    bs[0] = *reinterpret_cast<const f32x4*>(&lds.bias3[4 * q]);
    bs[1] = *reinterpret_cast<const f32x4*>(&lds.bias3[16 + 4 * q]);
    lds_barrier();
    load_taps(bq[0], lds.x3plane[0]);
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        if (c < 3) load_taps(bq[(c + 1) & 1], lds.x3plane[c + 1]);
        const bool fresh_layer = c == 0 || c >= 4;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const bf16x8 f0 = ws.take(kConvBase + c * 18 + 2 * tap);
            const bf16x8 f1 = ws.take(kConvBase + c * 18 + 2 * tap + 1);
            const bf16x8 bn = __builtin_bit_cast(bf16x8, bq[c & 1][tap]);
            A[0] = MFMA16(f0, bn, (fresh_layer && tap == 0) ? bs[0] : A[0]);
            A[1] = MFMA16(f1, bn, (fresh_layer && tap == 0) ? bs[1] : A[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c >= 3 && c < 8) {
            // x(c + 1) into plane (c + 1) & 1, whose readers (chunk c - 1) finished before the last barrier; the next
            // chunk reads it after this one
            write_tile(lds.plane[(c + 1) & 1], A[0], A[1]);
            bs[0] = *reinterpret_cast<const f32x4*>(&lds.bias3[32 * (c + 1 - 3) + 4 * q]);
            bs[1] = *reinterpret_cast<const f32x4*>(&lds.bias3[32 * (c + 1 - 3) + 16 + 4 * q]);
            lds_barrier();
            load_taps(bq[(c + 1) & 1], lds.plane[(c + 1) & 1]);
        }
        if (c == 3) BF_STAMP(4);
    }
    BF_STAMP(5);
    // the walk's padding groups bring the ring and the look-ahead back to their start phase
#pragma unroll
    for (int g = kStreamGroups; g < kStreamWalkGroups; ++g) (void)ws.take(g);
    BF_STAMP(6);

    if (MODE == 2) {
        float* o = out + (size_t)board * 2048;
#pragma unroll
        for (int tile = 0; tile < 2; ++tile)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(16 * tile + 4 * r + q) * 64 + 16 * wave + c16] = fmaxf(A[tile][r], 0.0f);
    } else {
        // ---- stage 5: this wave's sixteen squares of the 8x8 valid conv (every row of the MFMA is this board) ----
        write_tile(lds.plane[1], A[0], A[1]);   // x9 -> plane 1 (last read in chunk 7); the wave reads back only its own tile
        u32x4 hb[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int Fk = (i & 7) + 1;
            hb[i] = *reinterpret_cast<const u32x4*>(lds.plane[1] + ((2 * wave + (i >> 3) + 1) * XP_W + Fk) * POS_BYTES +
                                                    (((q + Fk) & 3) << 4));
        }
        f32x4 hacc = zero4();
#pragma unroll
        for (int i = 0; i < 16; ++i)
            hacc = MFMA16(__builtin_bit_cast(bf16x8, hb[i]), __builtin_bit_cast(bf16x8, hk[i]), hacc);
        lds.hpart[wave][lane] = hacc;
        lds_barrier();
        if (tid < 16) {
            float s = 0.0f;   // D[row 0][col = co] sits in lane co, register 0
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s += reinterpret_cast<const float*>(lds.hpart[w])[tid * 4];
            const float y = fmaxf(s + head_shift, 0.0f);
            float v = y * head_logit;
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            if (tid == 0) out[board] = v;
        }
    }
    BF_STAMP(7);
    // the next board's first barrier (after its image) also orders this board's last reads of the planes and of
    // hpart in front of the next board's writes
    }  // board loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's look-ahead DMA must not outlive the workgroup
#ifdef BF_STAMPS
    if (MODE == 0 && blockIdx.x == 0 && lane == 0) {
        stamps[9] = (unsigned)__builtin_amdgcn_s_memtime();
        unsigned* o = reinterpret_cast<unsigned*>(out) + 1024 + wave * 16;   // the caller's score buffer has room behind 1024 scores
        for (int i = 0; i < 16; ++i) o[i] = stamps[i];
    }
#endif
}

hipError_t split_blocks(size_t n, unsigned* blocks) {
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
    *blocks = (unsigned)(n < (size_t)count ? n : (size_t)count);
    return hipSuccess;
}

template <int MODE_>
hipError_t launch_split(const uint8_t* d_src, size_t stride, size_t n, const unsigned char* d_image, float* d_out, hipStream_t stream,
                        const WireArgs& wire = WireArgs{}) {
    unsigned blocks = 0;
    hipError_t e = split_blocks(n, &blocks);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_eval_t16s<MODE_>, dim3(blocks), dim3(THREADS), 0, stream, d_src, (unsigned)n,
                       reinterpret_cast<const f32x4*>(d_image + kOffStream), reinterpret_cast<const float*>(d_image + kOffBias),
                       reinterpret_cast<const f32x4*>(d_image + kOffHead), d_out, (unsigned)stride, wire);
    return hipGetLastError();
}

}  // namespace

}  // namespace t16

hipError_t launch_eval_t16_split(const uint8_t* d_src, size_t stride, bool records, size_t n, const unsigned char* d_image,
                                 float* d_scores, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (n > 0xFFFFFFFFull || stride > 0xFFFFFFFFull) return hipErrorInvalidValue;
    return records ? t16::launch_split<4>(d_src, stride, n, d_image, d_scores, stream)
                   : t16::launch_split<0>(d_src, stride, n, d_image, d_scores, stream);
}

hipError_t launch_policy_features_t16_split(const uint8_t* d_src, size_t stride, bool records, size_t n,
                                            const unsigned char* d_image, float* d_x9, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (n > 0xFFFFFFFFull || stride > 0xFFFFFFFFull) return hipErrorInvalidValue;
    return records ? t16::launch_split<6>(d_src, stride, n, d_image, d_x9, stream)
                   : t16::launch_split<2>(d_src, stride, n, d_image, d_x9, stream);
}

// Wire format in (see WireArgs above): d_nib 4-byte aligned with at least 3 readable bytes behind n_bytes, d_offsets
// within 4 GiB behind d_occ (both hold for the host entry points' staging buffer).  features: x9 instead of scores.
hipError_t launch_t16_split_wire(const uint8_t* d_nib, size_t n_bytes, const uint64_t* d_occ, const unsigned* d_offsets, size_t n,
                                 unsigned start_nibble, int* d_err, const unsigned char* d_image, float* d_out, bool features,
                                 hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const ptrdiff_t gap = reinterpret_cast<const unsigned char*>(d_offsets) - reinterpret_cast<const unsigned char*>(d_occ);
    if (n > 0x0FFFFFFFull || n_bytes > 0xFFFFFFF0ull || (reinterpret_cast<uintptr_t>(d_nib) & 3u) || gap < 0 ||
        (unsigned long long)gap + 4ull * n > 0xFFFFFFFFull || !d_err)
        return hipErrorInvalidValue;
    const t16::WireArgs wire{d_occ, d_offsets, d_nib, (unsigned)n_bytes, start_nibble, d_err};
    return features ? t16::launch_split<10>(nullptr, 0, n, d_image, d_out, stream, wire)
                    : t16::launch_split<8>(nullptr, 0, n, d_image, d_out, stream, wire);
}

}  // namespace bf
