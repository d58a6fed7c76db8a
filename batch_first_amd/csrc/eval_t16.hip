// The value network as one bf16 kernel on v_mfma_f32_16x16x32_bf16 (BASELINE configs[2]): 4-bit mailboxes
// in, one score per board out.  Arithmetic and weight image: bf_layout_t16.h.  Successor of round 1's kernel on
// 32x32x16 tiles (removed in round 3), same structure around the arithmetic: persistent workgroups (one per compute unit),
// one wave = one board, the weight stream walked as a ring through LDS by LDS-DMA, every stage a flat,
// fully unrolled list of MFMA slots with the LDS reads, packs and LDS writes of its neighbours in the gaps.
//
// Reference graph replaced (TensorFlow ops -> this kernel):
//   create_input_convolutions_shared_weights   ann_creation_helper.py:192-252  (stage 1)
//   build_convolutional_modules, spec at       evaluation_ann.py:268           (stages 2-4)
//   logit_layer + squeeze                      evaluation_ann.py:35-46         (stage 5)
//
// Why this shape: the 32x32x16 kernel is limited by the clock the chip holds under its
// load; in a bare loop the 16x16x32 instruction delivers 9-15 % more FLOP/s at that limit.  A board's 64
// squares are four 16-column tiles (ranks 2t, 2t+1), every 1-KiB weight fragment read from LDS feeds four
// MFMAs, and every layer has four independent accumulation chains, which is what hides the
// pack -> LDS write -> shifted read chain of the 3x3 layers: within a layer the column tiles run one
// after the other (the layer's 18 fragments stay in registers after the first), and the tile a column
// tile needs from the previous layer was finished at least two tiles earlier.
//   stage 1  B operand of a k-step = two taps x 16 planes: lanes q = 0,1 look up tap A's shifted square,
//            q = 2,3 tap B's, with ONE ds_read_b128 at an immediate offset from a one-hot image of the board
//            (grid[12 ranks][24 files][2 halves] of 16-byte entries, written once per board; the all-ones
//            16th plane comes from a register); a step whose taps all miss the board for a column tile is
//            skipped there.
//   stage 2  per pair of 16-row hidden tiles: 16 fragments (236->512), ReLU + pack, 8 fragments (512->128),
//            software-pipelined across pairs (stream order kPairOrder).
//   stage 3/4  3x3 layers through zero-haloed [10x10 positions][32 channels] bf16 planes private to the
//            wave; a position's 16-byte channel chunk q sits at chunk (q + padded file) & 3, which makes
//            the shifted ds_read_b128 of every tap conflict-free and the ds_write_b128 two-way at worst.
//   stage 5  8x8 valid conv across the workgroup's boards (one MFMA per square), ReLU, logit.
// Results are bit-reproducible and independent of a board's position in the batch.
#include <hip/hip_runtime.h>

#include "bf_device.h"
#include "bf_kernels.h"
#include "bf_layout_t16.h"
#include "t16_common.h"

namespace bf {

namespace t16 {

namespace {


// Stage 1 reads its B operands from a one-hot image of the board in LDS: grid[row][col][half] of 16-byte
// entries (8 bf16 planes each), rows -2 .. 9 and columns -7 .. 16 around the board.  Columns: the taps reach 7
// files sideways; 24 a row makes a rank 768 bytes = three bank windows, so the two ranks of a column tile
// never meet in a bank.  Rows: a unit is only computed when one of its two ranks lands on the board, so the
// other is at most one rank off it (3x3 convs), two for the never-skipped 2x2 convs; a tap that is further
// off is pointed at ranks -2, -1 instead (all zero, as every off-board entry is).  The all-ones 16th plane is
// OR-ed in from a register, so the image is zero wherever no piece stands and can alias the activation planes.
constexpr int GR_ROWS = 12, GR_COLS = 24, GR_ROW0 = 2, GR_COL0 = 7;
constexpr int GR_ENTRY = 32;             // two halves of 16 bytes
constexpr int GR_ROW_BYTES = GR_COLS * GR_ENTRY;       // 768
constexpr int GR_BYTES = GR_ROWS * GR_ROW_BYTES;       // 9216
// (activation planes: XP_W, POS_BYTES, ROW_BYTES, PLANE_BYTES in t16_common.h)
constexpr int BOARD_BYTES = 2 * PLANE_BYTES;
static_assert(GR_BYTES <= BOARD_BYTES && GR_BYTES % 1024 == 0, "the one-hot image aliases the activation planes");

struct Lds {
    f32x4 wring[SLOTS][CHUNK][64];                       // shared A-fragment ring, 64 KiB
    __attribute__((aligned(16))) unsigned char board[BPB][BOARD_BYTES];
    u32x4 onehot[16][2];                                 // stage 1: [code][plane half] -> 8 bf16 planes
    __attribute__((aligned(16))) float bias3[6 * 32];    // shifts of x4..x9: [layer][row tile][q][r]
    __attribute__((aligned(16))) float bias2[128];       // shifts of x3: [row tile][q][r]
    f32x4 hpart[WAVES][64];                              // stage 5: per-wave partial sums
    u32x4 headk[WAVES][8][64];                           // stage 5: the 8x8 kernels of each wave's squares 8..15 (the other eight live in registers)
    __attribute__((aligned(16))) unsigned char next_mailbox[WAVES][256];
};





// ---- stage 1 as a flat list of units = (conv, step, column tile) that are computed -------------------
constexpr int kMaxUnits = 176, kMaxJobs = 8;
struct S1Table {
    int n;
    short conv[kMaxUnits], step[kMaxUnits], t[kMaxUnits], g0[kMaxUnits];
    int imm[kMaxUnits];                              // grid byte offset common to both taps of the unit (the smaller one)
    int da[kMaxUnits], db[kMaxUnits];                // what tap A's / tap B's lanes add to it (one of them is 0)
    bool new_base[kMaxUnits];                        // (da, db) differs from the previous unit's
    bool first_of_step[kMaxUnits], first_of_tile[kMaxUnits];
    short njobs[kMaxUnits];
    short job[kMaxUnits][kMaxJobs];                  // pack jobs run after the unit's MFMAs: t * 32 + dword
    short ntail;
    short tail[64];                                  // jobs left after the last unit
};
constexpr S1Table make_s1_table() {
    S1Table T{};
    int n = 0, g = 0;
    int last_unit_of_conv[kNumInputConvs] = {};
    int first_unit_of_conv[kNumInputConvs + 2] = {};
    for (int i = 0; i < kNumInputConvs; ++i) {
        first_unit_of_conv[i] = n;
        bool seen[4] = {false, false, false, false};
        for (int s = 0; s < steps_of(i); ++s) {
            const TapPair tp = step_taps(i, s);
            bool first = true;
            for (int t = 0; t < 4; ++t) {
                if (!step_valid(i, s, t)) continue;
                T.conv[n] = (short)i; T.step[n] = (short)s; T.t[n] = (short)t; T.g0[n] = (short)g;
                // ranks of the two taps for this column tile; a tap whose ranks leave the grid reads ranks -2, -1
                auto grid_dr = [&](int tap) {
                    const int dr = tap_dr(i, tap);
                    const bool in_grid = 2 * t + dr >= -GR_ROW0 && 2 * t + 1 + dr <= GR_ROWS - GR_ROW0 - 1;
                    return in_grid ? dr : -2 * t - 2;
                };
                const int tb = tp.b >= 0 ? tp.b : tp.a;
                const int off_a = ((grid_dr(tp.a) + 2 * t + GR_ROW0) * GR_COLS + tap_df(i, tp.a) + GR_COL0) * GR_ENTRY;
                const int off_b = ((grid_dr(tb) + 2 * t + GR_ROW0) * GR_COLS + tap_df(i, tb) + GR_COL0) * GR_ENTRY;
                T.imm[n] = off_a < off_b ? off_a : off_b;
                T.da[n] = off_a - T.imm[n];
                T.db[n] = off_b - T.imm[n];
                T.new_base[n] = n == 0 || T.da[n] != T.da[n - 1] || T.db[n] != T.db[n - 1];
                T.first_of_step[n] = first;
                T.first_of_tile[n] = !seen[t];
                first = false;
                seen[t] = true;
                ++n;
            }
            g += kRowTiles[i];
        }
        last_unit_of_conv[i] = n - 1;
    }
    first_unit_of_conv[kNumInputConvs] = first_unit_of_conv[kNumInputConvs + 1] = n;
    T.n = n;
    // pack jobs in order (conv, column tile, dword); a conv's jobs may start one unit after its last unit and
    // must be done before conv + 2 starts (the accumulators alternate between two buffers)
    int jc = 0, jt = 0, jd = 0;   // next job: conv jc, tile jt, dword jd (local to the conv)
    auto job_code = [&](int c, int t, int d) { return (short)(t * 32 + kX1RegBase[c] / 2 + d); };
    auto advance = [&]() {
        if (++jd == kX1RegCount[jc] / 2) { jd = 0; if (++jt == 4) { jt = 0; ++jc; } }
    };
    for (int u = 0; u < n; ++u) {
        int k = 0;
        const bool flush = (u + 1 == n) ? false : (T.conv[u + 1] != T.conv[u]);   // last unit of its conv
        while (jc < kNumInputConvs && last_unit_of_conv[jc] < u) {
            const bool must = flush && jc + 1 <= T.conv[u];   // conv jc's buffer is overwritten by the next unit
            if (k >= 2 && !must) break;
            if (k >= kMaxJobs) break;
            T.job[u][k++] = job_code(jc, jt, jd);
            advance();
        }
        T.njobs[u] = (short)k;
    }
    int nt = 0;
    while (jc < kNumInputConvs) { T.tail[nt++] = job_code(jc, jt, jd); advance(); }
    T.ntail = (short)nt;
    return T;
}
constexpr S1Table kS1 = make_s1_table();
constexpr bool s1_jobs_meet_deadlines() {
    // every job of conv c is placed in a unit of conv <= c + 1, or in the tail when c + 2 does not exist
    for (int u = 0; u < kS1.n; ++u)
        for (int k = 0; k < kS1.njobs[u]; ++k) {
            const int d = kS1.job[u][k] % 32;
            const int c = x1_slot(2 * d).conv;
            if (kS1.conv[u] > c + 1 || kS1.conv[u] <= c) return false;
        }
    for (int k = 0; k < kS1.ntail; ++k)
        if (x1_slot(2 * (kS1.tail[k] % 32)).conv + 2 < kNumInputConvs) return false;
    return true;
}
static_assert(kS1.n <= kMaxUnits && kS1.ntail <= 64 && s1_jobs_meet_deadlines(), "");

// ReLU + bf16 of x1 dword d of column tile t from the finished accumulators of its conv
__device__ __forceinline__ unsigned s1_pack(int t, int d, const f32x4 (&acc)[2][4][4]) {
    const X1Slot a = x1_slot(2 * d), b = x1_slot(2 * d + 1);
    unsigned p = pack_relu(acc[a.conv & 1][a.tile][t][a.r], acc[b.conv & 1][b.tile][t][b.r]);
    // pinned where it is scheduled (the compiler sinks unpinned packs into the next basic block), and pinned INTO THE
    // ACCUMULATOR FILE: x1 is only ever a B operand, which an MFMA reads from an AGPR directly; left to the
    // allocator it stays in the 256 architectural VGPRs, where stage 2's 192 accumulator registers then no
    // longer fit, and the 512->128 accumulators get copied in and out of AGPRs around their MFMAs (730 copy
    // instructions and 700 cycles of hazard nops per board in the first build)
#ifndef T16_X_IN_VGPR
    asm volatile("" : "+a"(p));
#else
    asm volatile("" : "+v"(p));
#endif
    return p;
}

template <bool DBG>
__device__ __forceinline__ void s1_dump(int t, int d, unsigned p, int q, int c16, float* dbg_board) {
    if (!DBG) return;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const X1Slot s = x1_slot(2 * d + e);
        const int local = tile_channel(s.tile, s.r, 0) + q;
        if (local < kInputConvs[s.conv].ch)
            dbg_board[(kInputConvs[s.conv].chan_base + local) * 64 + 16 * t + c16] = e ? bf16_hi(p) : bf16_lo(p);
    }
}

// grid0: this lane's entry of column tile 0's square at tap offset (0, 0), rank -GR_ROW0, file -GR_COL0 (so that
// every unit's offset is a non-negative immediate), at this lane's plane half; q2: lane quarter >= 2 (looks up
// the step's second tap); ones: the all-ones 16th plane of this lane's half (bf16 1.0 in element 7 of half 1).
template <bool DBG>
__device__ __forceinline__ void stage1(WStream& ws, unsigned grid0, bool q2, unsigned ones, unsigned (&X)[4][kX1Dwords],
                                       int q, int c16, float* dbg_board, unsigned* stamps) {
    typedef const __attribute__((address_space(3))) u32x4* lds_entry;
#ifndef T16_S1_Q
#define T16_S1_Q 4
#endif
    constexpr int BQ = T16_S1_Q;   // B operands are read BQ units ahead of their MFMAs
    constexpr int N = kS1.n;
    u32x4 b[BQ];
    f32x4 acc[2][4][4];   // [conv parity][row tile][column tile]
    bf16x8 fr[4];
    unsigned base = grid0;   // LDS byte address: grid0 + what this lane's tap adds for the units being read
    auto read_b = [&](int u) -> u32x4 {
        const int uu = u < N ? u : N - 1;   // look-ahead past the end re-reads the last unit (harmless)
        if (u < N && kS1.new_base[uu]) {
            base = grid0 + (unsigned)(q2 ? kS1.db[uu] : kS1.da[uu]);
            asm volatile("" : "+v"(base));   // computed here, once per change, not hoisted for all units
        }
        return *(lds_entry)(uintptr_t)(base + (unsigned)kS1.imm[uu]);
    };
#pragma unroll
    for (int u = 0; u < BQ; ++u) b[u] = read_b(u);
    b[0].w |= ones;
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const int I = kS1.conv[u], t = kS1.t[u], par = I & 1;
#ifdef BF_STAMPS   // first unit of convs 1, 2, 4, 6, 7: slots 8..12; the tail packs: slot 13
        if (u > 0 && kS1.conv[u - 1] != I && (I == 1 || I == 2 || I == 4 || I == 6 || I == 7))
            stamps[I == 1 ? 8 : I == 2 ? 9 : I == 4 ? 10 : I == 6 ? 11 : 12] = (unsigned)__builtin_amdgcn_s_memtime();
#endif
        if (kS1.first_of_step[u]) {
#pragma unroll
            for (int tile = 0; tile < 4; ++tile)
                if (tile < kRowTiles[I]) fr[tile] = ws.take(kS1.g0[u] + tile);
        }
        const bf16x8 bu = __builtin_bit_cast(bf16x8, b[u % BQ]);
#pragma unroll
        for (int tile = 0; tile < 4; ++tile)
            if (tile < kRowTiles[I])
                acc[par][tile][t] = MFMA16(fr[tile], bu, kS1.first_of_tile[u] ? zero4() : acc[par][tile][t]);
        b[u % BQ] = read_b(u + BQ);
        // the all-ones plane (1.0 on every square, padding included) of the NEXT unit's operand, which was read
        // three units ago: here the OR issues under this unit's MFMAs, in front of its own MFMA it costs the
        // wait states of a VALU result read by an MFMA
        if (u + 1 < N) b[(u + 1) % BQ].w |= ones;
#pragma unroll
        for (int k = 0; k < kMaxJobs; ++k)
            if (k < kS1.njobs[u]) {
                const int jt = kS1.job[u][k] / 32, jd = kS1.job[u][k] % 32;
                X[jt][jd] = s1_pack(jt, jd, acc);
                s1_dump<DBG>(jt, jd, X[jt][jd], q, c16, dbg_board);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef BF_STAMPS
    stamps[13] = (unsigned)__builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int k = 0; k < 64; ++k)
        if (k < kS1.ntail) {
            const int jt = kS1.tail[k] / 32, jd = kS1.tail[k] % 32;
            X[jt][jd] = s1_pack(jt, jd, acc);
            s1_dump<DBG>(jt, jd, X[jt][jd], q, c16, dbg_board);
        }
}

// Debug dump of a 16-row tile (channels 16 * tile_index + 4r + q of column tile t) as the bf16 values the
// next layer sees.
__device__ __forceinline__ void dump_tile(float* dst_ch0, int q, int sq, const f32x4& a) {
    const unsigned p0 = pack_relu(a[0], a[1]), p1 = pack_relu(a[2], a[3]);
    dst_ch0[(0 + q) * 64 + sq] = bf16_lo(p0);
    dst_ch0[(4 + q) * 64 + sq] = bf16_hi(p0);
    dst_ch0[(8 + q) * 64 + sq] = bf16_lo(p1);
    dst_ch0[(12 + q) * 64 + sq] = bf16_hi(p1);
}

// MODE 0: scores; 1: scores + every intermediate activation (debug); 2: features -- x9 as float32
// [board][32][64] into `scores`, no head (the move-scoring network).
// MODE_ + 4: the boards come as node records (bytes 0..81 of each, `src_stride` apart, 4-byte aligned) instead of
// 4-bit mailboxes: the record -> code step of k_to_mailbox<StructSrc> done by the wave that evaluates the board,
// one kernel instead of two for a call (a launch and a dependent-launch gap are a tenth of a small call).
template <int MODE_>
__global__ __launch_bounds__(THREADS, 1) void k_eval_t16(const uint8_t* __restrict__ mailbox, unsigned n_boards,
                                                          const f32x4* __restrict__ wstream,
                                                          const float* __restrict__ bias,
                                                          const f32x4* __restrict__ headw,
                                                          float* __restrict__ scores, float* __restrict__ dbg,
                                                          unsigned src_stride) {
    constexpr int MODE = MODE_ & 3;
    constexpr bool RECS = (MODE_ & 4) != 0;
    constexpr int SRC_DWORDS = RECS ? 22 : 8;   // 88 bytes of a record / 32 bytes of a mailbox
    constexpr bool DBG = MODE == 1;
    __shared__ __attribute__((aligned(16))) Lds lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63, c16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rk = c16 >> 3, file = c16 & 7;                 // column tile 0's square; tile t is 2t ranks up
    const unsigned n_quads = (n_boards + BPB - 1) / BPB;
    typedef __attribute__((address_space(3))) unsigned char* lds_bytes;

#ifdef BF_STAMPS
    unsigned stamps[16] = {};
#endif
    WStream ws;
    ws.prime(wstream, &lds.wring[0][0][0], lane, wave);

    // ---- once per workgroup: the one-hot table, the shifts, the head kernels ---------------------------
    unsigned char* region = lds.board[wave];
    if (tid < 32) {
        const unsigned c = tid >> 1, hh = tid & 1, j = c & 7u;
        const unsigned one = (c != 15u && (c >> 3) == hh) ? (0x3F80u << ((j & 1u) * 16u)) : 0u;
        u32x4 e = u32x4{0u, 0u, 0u, 0u};   // (plane 15, the all-ones plane, is OR-ed in from a register: stage1)
        e.x |= ((j >> 1) == 0u) ? one : 0u;
        e.y |= ((j >> 1) == 1u) ? one : 0u;
        e.z |= ((j >> 1) == 2u) ? one : 0u;
        e.w |= ((j >> 1) == 3u) ? one : 0u;
        lds.onehot[c][hh] = e;
    }
    for (int i = tid; i < 6 * 32; i += THREADS) lds.bias3[i] = bias[kBiasX4 + i];
    for (int i = tid; i < 128; i += THREADS) lds.bias2[i] = bias[kBiasX3 + i];
    constexpr int SQW = 64 / WAVES;  // squares (k-steps of the head) per wave
    constexpr int SQR = SQW / 2;     // of which this many have their kernels in registers, the rest in LDS
    u32x4 ha[SQR];
    if (MODE != 2) {
        const f32x4* hwp = headw + (size_t)wave * SQW * 64 + lane;
#pragma unroll
        for (int i = 0; i < SQR; ++i) ha[i] = DBG ? u32x4{0u, 0u, 0u, 0u} : __builtin_bit_cast(u32x4, hwp[i * 64]);
#pragma unroll
        for (int i = SQR; i < SQW; ++i) lds.headk[wave][i - SQR][lane] = __builtin_bit_cast(u32x4, hwp[i * 64]);
    }
    const float head_shift = MODE != 2 ? bias[kBiasX10 + (tid & 15)] : 0.0f;
    const float head_logit = MODE != 2 ? bias[kLogitW + (tid & 15)] : 0.0f;
    for (int i = lane; i < 2 * PLANE_BYTES / 16; i += 64) reinterpret_cast<u32x4*>(region)[i] = u32x4{0u, 0u, 0u, 0u};
    {   // the first board's mailbox directly (every later one is prefetched)
        const unsigned b0 = blockIdx.x * BPB + wave;
        const unsigned bc = b0 < n_boards ? b0 : n_boards - 1;
        if (lane < SRC_DWORDS)
            reinterpret_cast<unsigned*>(lds.next_mailbox[wave])[lane] =
                reinterpret_cast<const unsigned*>(mailbox + (size_t)bc * src_stride)[lane];
    }
    lds_order();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ws.start();

    // lane constants of the activation planes: padded file F = file + 1, chunk of quarter q at (q + F) & 3
    const int waddr = ((rk + 1) * XP_W + file + 1) * POS_BYTES + (((q + file + 1) & 3) << 4);   // tile 0's square; tile t at + 2t rows
    int raddr[3];   // shifted reads, df = -1, 0, +1, one padded row ABOVE the square (dr = -1): row offsets stay >= 0
#pragma unroll
    for (int d = 0; d < 3; ++d) raddr[d] = (rk * XP_W + file + d) * POS_BYTES + (((q + file + d) & 3) << 4);

#pragma unroll 1
    for (unsigned quad = blockIdx.x; quad < n_quads; quad += gridDim.x) {
    const unsigned board_raw = quad * BPB + wave;
    const unsigned board = board_raw < n_boards ? board_raw : n_boards - 1;  // padding boards repeat the last one
    unsigned next_raw = (quad + gridDim.x) * BPB + wave;
    const unsigned next_board = __builtin_amdgcn_readfirstlane(next_raw < n_boards ? next_raw : n_boards - 1);
    ws.board_start();
    BF_STAMP(0);
#ifdef BF_STAMPS
    stamps[7] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif

    // ---- one-hot image of the board: zero (the planes hold the previous board's activations), then every
    // square's two entries; a wave's LDS operations are carried out in the order it issues them
#pragma unroll
    for (int i = 0; i < GR_BYTES / 1024; ++i) reinterpret_cast<u32x4*>(region)[lane + 64 * i] = u32x4{0u, 0u, 0u, 0u};
    lds_order();
    unsigned char* my_entry = region + (((lane >> 3) + GR_ROW0) * GR_COLS + (lane & 7) + GR_COL0) * GR_ENTRY;
    {
        unsigned c;
        if (RECS) {
            const BoardBits bits = load_board(lds.next_mailbox[wave]);
            c = (unsigned)square_code(bits, relevance_mask(bits), lane);
        } else {
            const unsigned v = lds.next_mailbox[wave][lane >> 1];
            c = (lane & 1) ? (v >> 4) : (v & 15u);
        }
        reinterpret_cast<u32x4*>(my_entry)[0] = lds.onehot[c][0];
        reinterpret_cast<u32x4*>(my_entry)[1] = lds.onehot[c][1];
    }
    lds_order();

    float* dbg_board = DBG ? dbg + (size_t)board * kDebugFloatsPerBoard : nullptr;
    BF_STAMP(1);

    // ---- stage 1: input convs -> x1 as 32 packed dwords per column tile ---------------------------
    unsigned X[4][kX1Dwords];
    {
        const unsigned grid0 = (unsigned)(uintptr_t)(lds_bytes)region + (rk * GR_COLS + file) * GR_ENTRY + (q & 1) * 16;
#ifdef BF_STAMPS
        unsigned* sp = stamps;
#else
        unsigned* sp = nullptr;
#endif
        stage1<DBG>(ws, grid0, q >= 2, (q & 1) ? 0x3F800000u : 0u, X, q, c16, dbg_board, sp);
    }
    // the image is dead: with this lane's square cleared the planes' halo is all zero again
    lds_order();
    reinterpret_cast<u32x4*>(my_entry)[0] = u32x4{0u, 0u, 0u, 0u};
    reinterpret_cast<u32x4*>(my_entry)[1] = u32x4{0u, 0u, 0u, 0u};

    BF_STAMP(2);
    // ---- stage 2: 1x1 236->512 (+ReLU) chained into 1x1 512->128 -----------------------------------
    f32x4 acc3[kOutTiles][4];
    {
        f32x4 acc2[2][2][4];   // [pair parity][tile of the pair][column tile]
        unsigned H[4][4];      // packed hidden pair: [column tile][dword]
        f32x4 seed[kOutTiles];
        auto pack_piece = [&](int par, int k) {   // k = 0..15: column tile k >> 2, dword k & 3
            const int t = k >> 2, d = k & 3;
            H[t][d] = pack_relu(acc2[par][d >> 1][t][2 * (d & 1)], acc2[par][d >> 1][t][2 * (d & 1) + 1]);
        };
#pragma unroll
        for (int p = 0; p < kPairGroups; ++p) {
            const int g = kPairBase + p;
            const int kind = kPairOrder.kind[p], a = kPairOrder.a[p], b = kPairOrder.b[p];
            const bf16x8 w = ws.take(g);
            if (g == kPrefetchGroup)
                ws.prefetch_mailbox(mailbox + (size_t)next_board * src_stride,
                                    (unsigned)(uintptr_t)(lds_bytes)lds.next_mailbox[wave],
                                    (unsigned)(lane < SRC_DWORDS ? lane : SRC_DWORDS - 1) * 4u);
            // the shift of output tile b, two groups ahead of the first 512->128 group that starts from it
            if (p >= 23 && p < 23 + 2 * kOutTiles && ((p - 23) & 1) == 0)
                seed[(p - 23) >> 1] = *reinterpret_cast<const f32x4*>(&lds.bias2[16 * ((p - 23) >> 1) + 4 * q]);
            if (kind == 0) {
                const int par = (a >> 1) & 1;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const u32x4 x = u32x4{X[t][4 * b], X[t][4 * b + 1], X[t][4 * b + 2], X[t][4 * b + 3]};
                    acc2[par][a & 1][t] = MFMA16(w, __builtin_bit_cast(bf16x8, x), b == 0 ? zero4() : acc2[par][a & 1][t]);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const u32x4 h = u32x4{H[t][0], H[t][1], H[t][2], H[t][3]};
                    acc3[b][t] = MFMA16(w, __builtin_bit_cast(bf16x8, h), a == 0 ? seed[b] : acc3[b][t]);
                }
            }
            // packs of pair h ride in the first eight groups of step h (the 236->512 groups of pair h + 1)
            if (p >= 16 && p < 16 + 24 * 15) {
                const int h = (p - 16) / 24, j = (p - 16) % 24;
                if (j < 8) { pack_piece(h & 1, 2 * j); pack_piece(h & 1, 2 * j + 1); }
            }
            if (p == 16 + 24 * 15 - 1) {   // the last pair: packed in front of its own 512->128 groups
#pragma unroll
                for (int k = 0; k < 16; ++k) pack_piece(1, k);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (DBG) {
#pragma unroll
        for (int o = 0; o < kOutTiles; ++o)
#pragma unroll
            for (int t = 0; t < 4; ++t) dump_tile(dbg_board + 64 * (236 + 16 * o), q, 16 * t + c16, acc3[o][t]);
    }

    BF_STAMP(3);
    // ---- stages 3 and 4: the six 3x3 layers through the wave's own planes ------------------------
    // Flat list of 9 chunks x 36 slots: chunk c = 0..3 are the four 32-channel input chunks of the 128->32
    // layer, c = 4..8 the five 32->32 layers; chunk c reads plane c & 1.  Slot = (column tile, tap): one
    // shifted B read feeds the MFMAs of both output row tiles; column tile 0 takes the chunk's 18 fragments
    // from the ring, tiles 1-3 reuse them from registers.  A layer's column tile t is written to the next
    // plane a few slots after it finishes, i.e. while tile t + 1 multiplies; the next layer's tile t needs
    // tiles t - 1 .. t + 1, the last of which was finished two tiles earlier.
    auto write_tile = [&](int pl, int t, const f32x4& a0, const f32x4& a1) {
        u32x4 v;
        v.x = pack_relu(a0[0], a0[1]);
        v.y = pack_relu(a0[2], a0[3]);
        v.z = pack_relu(a1[0], a1[1]);
        v.w = pack_relu(a1[2], a1[3]);
        lds_order();
        *reinterpret_cast<u32x4*>(region + pl * PLANE_BYTES + waddr + 2 * t * ROW_BYTES) = v;
        lds_order();
    };
    auto load_b = [&](int n) -> u32x4 {
        const int c = n / 36, t = (n % 36) / 9, tap = n % 9;
        return *reinterpret_cast<const u32x4*>(region + (c & 1) * PLANE_BYTES + raddr[tap % 3] + (2 * t + tap / 3) * ROW_BYTES);
    };
#ifndef T16_CONV_D
#define T16_CONV_D 6
#endif
    constexpr int D = T16_CONV_D, NSLOTS = 9 * 36;
    static_assert(D <= 8, "a B read further ahead could pass the write of its input");
    f32x4 A[2][2][4];   // [set][row tile][column tile]: chunks 0..3 accumulate in set 0, chunk c >= 4 in set c & 1 ^ 1 ... see `set`
    f32x4 bs[2];        // shifts of the layer about to start, per row tile
    u32x4 bq[D];
    bf16x8 f[18];
#pragma unroll
    for (int t = 0; t < 4; ++t) write_tile(0, t, acc3[0][t], acc3[1][t]);
    bs[0] = *reinterpret_cast<const f32x4*>(&lds.bias3[4 * q]);
    bs[1] = *reinterpret_cast<const f32x4*>(&lds.bias3[16 + 4 * q]);
#pragma unroll
    for (int n = 0; n < D; ++n) bq[n] = load_b(n);
#pragma unroll
    for (int n = 0; n < NSLOTS; ++n) {
        const int c = n / 36, r = n % 36, t = r / 9, tap = r % 9;
        if (n == 4 * 36) BF_STAMP(4);
        const int set = c <= 3 ? 0 : ((c & 1) ^ 1);   // c = 4 -> 1, 5 -> 0, ...
        if (t == 0) {
            f[2 * tap] = ws.take(kConvBase + c * 18 + 2 * tap);
            f[2 * tap + 1] = ws.take(kConvBase + c * 18 + 2 * tap + 1);
        }
        const bool fresh = tap == 0 && (c == 0 || c >= 4);
        const bf16x8 bn = __builtin_bit_cast(bf16x8, bq[n % D]);
        A[set][0][t] = MFMA16(f[2 * tap], bn, fresh ? bs[0] : A[set][0][t]);
        A[set][1][t] = MFMA16(f[2 * tap + 1], bn, fresh ? bs[1] : A[set][1][t]);
        if (n + D < NSLOTS) bq[n % D] = load_b(n + D);
        if (c < 3) {
            // input of chunk c + 1 into the other plane (chunk c - 1's reads of it were issued long ago)
            if (r >= 1 && r <= 4) write_tile((c + 1) & 1, r - 1, acc3[2 * (c + 1)][r - 1], acc3[2 * (c + 1) + 1][r - 1]);
        } else if (c < 8) {
            // column tile tw of x(c+1) two slots after its last MFMA: tiles 0..2 inside this chunk ...
            if (r == 11 || r == 20 || r == 29) write_tile((c + 1) & 1, r / 9 - 1, A[set][0][r / 9 - 1], A[set][1][r / 9 - 1]);
            if (r == 30) {   // shifts of the next layer
                bs[0] = *reinterpret_cast<const f32x4*>(&lds.bias3[32 * (c + 1 - 3) + 4 * q]);
                bs[1] = *reinterpret_cast<const f32x4*>(&lds.bias3[32 * (c + 1 - 3) + 16 + 4 * q]);
            }
        }
        if (c >= 4 && r == 2) {   // ... tile 3 of the previous chunk at the start of this one
            const int ps = c == 4 ? 0 : (((c - 1) & 1) ^ 1);
            write_tile(c & 1, 3, A[ps][0][3], A[ps][1][3]);
        }
        if (DBG && c >= 3 && c < 8 && r == 35) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                dump_tile(dbg_board + 64 * (236 + 128 + 32 * (c - 3)), q, 16 * tt + c16, A[set][0][tt]);
                dump_tile(dbg_board + 64 * (236 + 128 + 32 * (c - 3) + 16), q, 16 * tt + c16, A[set][1][tt]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 (&x9)[2][4] = A[1];  // chunk 8: set 1; x9 before its ReLU

    BF_STAMP(5);
    // the walk's padding groups bring the ring and the look-ahead back to their start phase
#pragma unroll
    for (int g = kStreamGroups; g < kStreamWalkGroups; ++g) (void)ws.take(g);

    // ---- stage 5: 8x8 valid conv (2048 -> 16) across the workgroup's boards, ReLU, logit ----------
    if (MODE == 2) {
        if (board_raw < n_boards) {
            float* out = scores + (size_t)board * 2048;
#pragma unroll
            for (int tile = 0; tile < 2; ++tile)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        out[(16 * tile + 4 * r + q) * 64 + 16 * t + c16] = fmaxf(x9[tile][t][r], 0.0f);
        }
    } else {
    lds_order();
#pragma unroll
    for (int t = 0; t < 4; ++t) write_tile(1, t, x9[0][t], x9[1][t]);   // x9 -> plane 1
    if (DBG) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            dump_tile(dbg_board + 64 * (236 + 128 + 32 * 5), q, 16 * tt + c16, x9[0][tt]);
            dump_tile(dbg_board + 64 * (236 + 128 + 32 * 5 + 16), q, 16 * tt + c16, x9[1][tt]);
        }
    }
    lds_barrier();   // every wave's x9 is in its plane 1 (LDS only: the ring's DMA stays in flight)
    {
        // out[board][co]: activations first (rows = boards, lane l reads board l & 15, chunk l >> 4), kernels
        // second, read from the AGPRs they were loaded into (inline asm: the builtin copies them to VGPRs first)
        const unsigned char* xb = lds.board[c16 < BPB ? c16 : 0] + PLANE_BYTES;
        u32x4 hb[SQW], hk[SQW - SQR];
#pragma unroll
        for (int i = 0; i < SQW; ++i) {
            const int sqk = wave * SQW + i;
            const int Rk = (sqk >> 3) + 1, Fk = (sqk & 7) + 1;
            hb[i] = *reinterpret_cast<const u32x4*>(xb + (Rk * XP_W + Fk) * POS_BYTES + (((q + Fk) & 3) << 4));
        }
#pragma unroll
        for (int i = 0; i < SQW - SQR; ++i) hk[i] = lds.headk[wave][i][lane];
        __builtin_amdgcn_sched_barrier(0);
        f32x4 hacc = f32x4{0.f, 0.f, 0.f, 0.f};
        // register half: inline asm so that the kernels are read from the AGPRs they live in (s_nop: an AGPR the
        // compiler has just moved is not yet visible to an MFMA that hipcc does not know is one)
        if (DBG) {
            // the debug build spills registers: it reads the kernels again instead of trusting asm operands
            const f32x4* hwp = headw + (size_t)wave * SQW * 64 + lane;
#pragma unroll
            for (int i = 0; i < SQR; ++i)
                hacc = MFMA16(__builtin_bit_cast(bf16x8, hb[i]), __builtin_bit_cast(bf16x8, hwp[i * 64]), hacc);
        } else {
#pragma unroll
            for (int i = 0; i < SQR; ++i)
                asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(hacc) : "v"(hb[i]), "a"(ha[i]));
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(hacc));
        }
#pragma unroll
        for (int i = SQR; i < SQW; ++i)
            hacc = MFMA16(__builtin_bit_cast(bf16x8, hb[i]), __builtin_bit_cast(bf16x8, hk[i - SQR]), hacc);
        lds.hpart[wave][lane] = hacc;
    }
    lds_barrier();
    if (tid < 16 * BPB) {
        const int b = tid >> 4, co = tid & 15;
        // D[row = b][col = co] sits in lane co + 16*(b>>2), register b&3
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w)
            s += reinterpret_cast<const float*>(lds.hpart[w])[(co + 16 * (b >> 2)) * 4 + (b & 3)];
        const float y = fmaxf(s + head_shift, 0.0f);
        const unsigned out_raw = quad * BPB + b;
        if (DBG && out_raw < n_boards) dbg[(size_t)out_raw * kDebugFloatsPerBoard + 64 * (236 + 128 + 32 * 6) + co] = y;
        float v = y * head_logit;  // logit_layer: no bias, no batch-norm (evaluation_ann.py:35-44)
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        if (co == 0 && out_raw < n_boards) scores[out_raw] = v;
    }
    }
    BF_STAMP(6);
#ifdef BF_STAMPS
    stamps[14] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
    }  // board loop
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's look-ahead DMA must not outlive the workgroup
#ifdef BF_STAMPS
    if (!DBG && dbg && lane == 0) {
        long long* out = reinterpret_cast<long long*>(dbg) + ((size_t)blockIdx.x * WAVES + wave) * 16;
        for (int i = 0; i < 15; ++i) out[i] = (long long)stamps[i];
    }
#endif
}

hipError_t grid_blocks(size_t n, unsigned* blocks) {
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

}  // namespace

}  // namespace t16

hipError_t launch_eval_t16(const uint8_t* d_mailbox, size_t n, const unsigned char* d_image, float* d_scores,
                           float* d_debug, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const f32x4* ws = reinterpret_cast<const f32x4*>(d_image + t16::kOffStream);
    const float* bias = reinterpret_cast<const float*>(d_image + t16::kOffBias);
    const f32x4* head = reinterpret_cast<const f32x4*>(d_image + t16::kOffHead);
    unsigned blocks = 0;
    hipError_t e = t16::grid_blocks(n, &blocks);
    if (e != hipSuccess) return e;
#if defined(BF_STAMPS) || defined(T16_NO_DEBUG_KERNEL)
    if (false)
#else
    if (d_debug)
#endif
        hipLaunchKernelGGL(t16::k_eval_t16<1>, dim3(blocks), dim3(t16::THREADS), 0, stream, d_mailbox, (unsigned)n, ws, bias, head,
                           d_scores, d_debug, (unsigned)kMailboxBytes);
    else
        hipLaunchKernelGGL(t16::k_eval_t16<0>, dim3(blocks), dim3(t16::THREADS), 0, stream, d_mailbox, (unsigned)n, ws, bias, head,
                           d_scores, d_debug, (unsigned)kMailboxBytes);
    return hipGetLastError();
}

bool t16_takes_records(const void* d_structs, size_t stride) {
    return (reinterpret_cast<uintptr_t>(d_structs) & 3u) == 0 && (stride & 3u) == 0 && stride >= 88 && stride <= 0xFFFFFFFFull;
}

// Scores straight from node records (t16_takes_records(d_structs, stride) must hold).
hipError_t launch_eval_t16_structs(const void* d_structs, size_t stride, size_t n, const unsigned char* d_image, float* d_scores,
                                   hipStream_t stream) {
    if (n == 0) return hipSuccess;
    unsigned blocks = 0;
    hipError_t e = t16::grid_blocks(n, &blocks);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(t16::k_eval_t16<4>, dim3(blocks), dim3(t16::THREADS), 0, stream, static_cast<const uint8_t*>(d_structs),
                       (unsigned)n, reinterpret_cast<const f32x4*>(d_image + t16::kOffStream),
                       reinterpret_cast<const float*>(d_image + t16::kOffBias),
                       reinterpret_cast<const f32x4*>(d_image + t16::kOffHead), d_scores, nullptr, (unsigned)stride);
    return hipGetLastError();
}

hipError_t launch_policy_features_t16_structs(const void* d_structs, size_t stride, size_t n, const unsigned char* d_image,
                                              float* d_x9, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    unsigned blocks = 0;
    hipError_t e = t16::grid_blocks(n, &blocks);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(t16::k_eval_t16<6>, dim3(blocks), dim3(t16::THREADS), 0, stream, static_cast<const uint8_t*>(d_structs),
                       (unsigned)n, reinterpret_cast<const f32x4*>(d_image + t16::kOffStream),
                       reinterpret_cast<const float*>(d_image + t16::kOffBias),
                       reinterpret_cast<const f32x4*>(d_image + t16::kOffHead), d_x9, nullptr, (unsigned)stride);
    return hipGetLastError();
}

hipError_t launch_policy_features_t16(const uint8_t* d_mailbox, size_t n, const unsigned char* d_image, float* d_x9,
                                      hipStream_t stream) {
    if (n == 0) return hipSuccess;
    unsigned blocks = 0;
    hipError_t e = t16::grid_blocks(n, &blocks);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(t16::k_eval_t16<2>, dim3(blocks), dim3(t16::THREADS), 0, stream, d_mailbox, (unsigned)n,
                       reinterpret_cast<const f32x4*>(d_image + t16::kOffStream),
                       reinterpret_cast<const float*>(d_image + t16::kOffBias),
                       reinterpret_cast<const f32x4*>(d_image + t16::kOffHead), d_x9, nullptr, (unsigned)kMailboxBytes);
    return hipGetLastError();
}

}  // namespace bf
