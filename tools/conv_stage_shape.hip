// Microbenchmark for the round-3 question of DESIGN.md 3.2.1: the 3x3 stage of the bf16 network (per 32-channel chunk
// and column tile: nine shifted B reads from an LDS plane, eighteen A fragments from an LDS ring, eighteen MFMAs, then
// ReLU + pack + one 16-byte plane write per tile) in two shapes on a full chip,
//   A  one wave = one board: four column tiles per wave, four waves per workgroup (one per SIMD), no barrier between
//      layers (a wave's own LDS operations are ordered), B reads six slots ahead          -- the shape of k_eval_t16;
//   B  two waves = one board: two column tiles per wave, eight waves per workgroup (two per SIMD), a workgroup barrier
//      per chunk, B reads at the start of the chunk                                        -- the candidate.
// Both do the same MFMAs per board (9 chunks x 4 tiles x 18) and read every ring fragment once per wave.  Prints the
// time per board quad and the matrix pipe's share of it.  Numbers are wrong on purpose (no real weights).
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/conv_stage_shape.hip -o build/conv_stage_shape
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

constexpr int XP_W = 10, POS = 64, ROW = XP_W * POS, PLANE = XP_W * 672;   // zero-haloed 10 x 10 x 32 bf16 (room for shape C's pitch)
constexpr int ROWC = 672;
constexpr int RING = 64;                                                  // fragments resident in LDS (64 KiB)

__device__ __forceinline__ unsigned pack_relu(float a, float b) {
    const bf16x2 v = __builtin_convertvector(f32x2{a, b}, bf16x2);
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return u & ~((u >> 15) & 0x00010001u) * 0xFFFFu;   // a stand-in of the same length as the signed max
}

// SHARED (shape C, round 3): shape A with a column tile = ranks (t, t + 4) instead of (2t, 2t + 1).  The B operand of tap
// (dr, df) of tile t is then the pair of plane rows (t + dr, t + 4 + dr) = "virtual tile" v = t + dr, the same fragment
// for (t, dr) = (v, 0), (v - 1, +1), (v + 1, -1): 6 virtual tiles x 3 file shifts = 18 B reads a chunk instead of 36,
// kept in registers while three tiles use them.  The plane's row pitch is 672 bytes so that rows four apart fall into
// different bank halves (4 x 672 = 10 x 256 + 128).
template <int TPW, bool SHARED = false>   // tiles per wave: 4 (shape A) or 2 (shape B)
__global__ __launch_bounds__(64 * 16 / TPW, 1) void k(float* out, int quads) {
    constexpr int WAVES = 16 / TPW;   // four boards a workgroup
    constexpr int WPB = 4 / TPW;      // waves per board
    __shared__ __attribute__((aligned(16))) f32x4 ring[RING][64];
    __shared__ __attribute__((aligned(16))) unsigned char planes[4][2][PLANE];
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int board = wave / WPB, part = wave % WPB;   // part: which TPW tiles of the board
    for (int i = tid; i < RING * 64; i += 64 * WAVES) (&ring[0][0])[i] = f32x4{1e-3f * (i & 7), 2e-3f, -1e-3f, 5e-4f};
    for (int i = tid; i < 4 * 2 * PLANE / 16; i += 64 * WAVES) reinterpret_cast<u32x4*>(&planes[0][0][0])[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    const int rk = c16 >> 3, file = c16 & 7;
    unsigned char* mine = &planes[board][0][0];
    const int waddr = ((rk + 1) * XP_W + file + 1) * POS + (((q + file + 1) & 3) << 4);
    int raddr[3];
    for (int d = 0; d < 3; ++d) raddr[d] = (rk * XP_W + file + d) * POS + (((q + file + d) & 3) << 4);
    f32x4 acc[TPW][2];
    for (int t = 0; t < TPW; ++t) acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* rp = &ring[0][lane];
    for (int quad = 0; quad < quads; ++quad) {
#pragma unroll 1
        for (int c = 0; c < 9; ++c) {
            const unsigned char* src = mine + (c & 1) * PLANE;
            unsigned char* dst = mine + ((c + 1) & 1) * PLANE;
            if (TPW == 4 && SHARED) {
                const int waddrC = ((4 * rk + 1) * ROWC) + (file + 1) * POS + (((q + file + 1) & 3) << 4);   // tile 0's square; tile t one row up each
                int raddrC[3];
                for (int d = 0; d < 3; ++d) raddrC[d] = (4 * rk) * ROWC + (file + d) * POS + (((q + file + d) & 3) << 4);   // virtual tile -1
                u32x4 bv[4][3];   // ring of four virtual tiles
                bf16x8 f[18];
                auto load_v = [&](int v) {   // v = -1 .. 4
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                        bv[(v + 1) & 3][d] = *reinterpret_cast<const u32x4*>(src + raddrC[d] + (v + 1) * ROWC);
                };
                load_v(-1); load_v(0); load_v(1);
#pragma unroll
                for (int n = 0; n < 36; ++n) {
                    const int t = n / 9, tap = n % 9;
                    if (t == 0) {
                        f[2 * tap] = __builtin_bit_cast(bf16x8, rp[((c * 18 + 2 * tap) % RING) * 64]);
                        f[2 * tap + 1] = __builtin_bit_cast(bf16x8, rp[((c * 18 + 2 * tap + 1) % RING) * 64]);
                    }
                    if (tap == 0 && t < 3) load_v(t + 2);   // for the next tile, a whole tile ahead
                    const int v = t + tap / 3 - 1;
                    const bf16x8 b = __builtin_bit_cast(bf16x8, bv[(v + 1) & 3][tap % 3]);
                    acc[t][0] = MFMA16(f[2 * tap], b, acc[t][0]);
                    acc[t][1] = MFMA16(f[2 * tap + 1], b, acc[t][1]);
                    if (tap == 2 && t > 0) {
                        u32x4 v4;
                        v4.x = pack_relu(acc[t - 1][0][0], acc[t - 1][0][1]);
                        v4.y = pack_relu(acc[t - 1][0][2], acc[t - 1][0][3]);
                        v4.z = pack_relu(acc[t - 1][1][0], acc[t - 1][1][1]);
                        v4.w = pack_relu(acc[t - 1][1][2], acc[t - 1][1][3]);
                        *reinterpret_cast<u32x4*>(dst + waddrC + (t - 1) * ROWC) = v4;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                u32x4 v4;
                v4.x = pack_relu(acc[3][0][0], acc[3][0][1]);
                v4.y = pack_relu(acc[3][0][2], acc[3][0][3]);
                v4.z = pack_relu(acc[3][1][0], acc[3][1][1]);
                v4.w = pack_relu(acc[3][1][2], acc[3][1][3]);
                *reinterpret_cast<u32x4*>(dst + waddrC + 3 * ROWC) = v4;
            } else if (TPW == 4) {
                // shape A: flat list of 36 slots (tile, tap), B six slots ahead, fragments of the chunk kept after tile 0
                constexpr int D = 6;
                u32x4 bq[D];
                bf16x8 f[18];
                auto load_b = [&](int n) -> u32x4 {
                    const int t = n / 9, tap = n % 9;
                    return *reinterpret_cast<const u32x4*>(src + raddr[tap % 3] + (2 * t + tap / 3) * ROW);
                };
#pragma unroll
                for (int n = 0; n < D; ++n) bq[n] = load_b(n);
#pragma unroll
                for (int n = 0; n < 36; ++n) {
                    const int t = n / 9, tap = n % 9;
                    if (t == 0) {
                        f[2 * tap] = __builtin_bit_cast(bf16x8, rp[((c * 18 + 2 * tap) % RING) * 64]);
                        f[2 * tap + 1] = __builtin_bit_cast(bf16x8, rp[((c * 18 + 2 * tap + 1) % RING) * 64]);
                    }
                    const bf16x8 b = __builtin_bit_cast(bf16x8, bq[n % D]);
                    acc[t % TPW][0] = MFMA16(f[2 * tap], b, acc[t % TPW][0]);
                    acc[t % TPW][1] = MFMA16(f[2 * tap + 1], b, acc[t % TPW][1]);
                    if (n + D < 36) bq[n % D] = load_b(n + D);
                    if (tap == 2 && t > 0) {   // the previous tile's output, two slots after its last MFMA
                        u32x4 v;
                        v.x = pack_relu(acc[(t - 1) % TPW][0][0], acc[(t - 1) % TPW][0][1]);
                        v.y = pack_relu(acc[(t - 1) % TPW][0][2], acc[(t - 1) % TPW][0][3]);
                        v.z = pack_relu(acc[(t - 1) % TPW][1][0], acc[(t - 1) % TPW][1][1]);
                        v.w = pack_relu(acc[(t - 1) % TPW][1][2], acc[(t - 1) % TPW][1][3]);
                        *reinterpret_cast<u32x4*>(dst + waddr + 2 * (t - 1) * ROW) = v;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                u32x4 v;
                v.x = pack_relu(acc[3 % TPW][0][0], acc[3 % TPW][0][1]);
                v.y = pack_relu(acc[3 % TPW][0][2], acc[3 % TPW][0][3]);
                v.z = pack_relu(acc[3 % TPW][1][0], acc[3 % TPW][1][1]);
                v.w = pack_relu(acc[3 % TPW][1][2], acc[3 % TPW][1][3]);
                *reinterpret_cast<u32x4*>(dst + waddr + 6 * ROW) = v;
            } else {
                // shape B: this wave's two tiles; all B operands at the start of the chunk, a barrier at its end
                u32x4 bq[2][9];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap)
                        bq[t][tap] = *reinterpret_cast<const u32x4*>(src + raddr[tap % 3] + (2 * (2 * part + t) + tap / 3) * ROW);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const bf16x8 f0 = __builtin_bit_cast(bf16x8, rp[((c * 18 + 2 * tap) % RING) * 64]);
                    const bf16x8 f1 = __builtin_bit_cast(bf16x8, rp[((c * 18 + 2 * tap + 1) % RING) * 64]);
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bf16x8 b = __builtin_bit_cast(bf16x8, bq[t][tap]);
                        acc[t % TPW][0] = MFMA16(f0, b, acc[t % TPW][0]);
                        acc[t % TPW][1] = MFMA16(f1, b, acc[t % TPW][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    u32x4 v;
                    v.x = pack_relu(acc[t % TPW][0][0], acc[t % TPW][0][1]);
                    v.y = pack_relu(acc[t % TPW][0][2], acc[t % TPW][0][3]);
                    v.z = pack_relu(acc[t % TPW][1][0], acc[t % TPW][1][1]);
                    v.w = pack_relu(acc[t % TPW][1][2], acc[t % TPW][1][3]);
                    *reinterpret_cast<u32x4*>(dst + waddr + 2 * (2 * part + t) * ROW) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
    }
    float s = 0.f;
    for (int t = 0; t < TPW; ++t) s += acc[t][0][0] + acc[t][1][3];
    out[blockIdx.x * 64 * WAVES + tid] = s;
}

template <int TPW, bool SHARED = false>
void run(const char* name) {
    const int blocks = 256, quads = 400;
    float* out;
    hipMalloc(&out, blocks * 1024 * sizeof(float));
    hipLaunchKernelGGL((k<TPW, SHARED>), dim3(blocks), dim3(64 * 16 / TPW), 0, 0, out, quads);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<TPW, SHARED>), dim3(blocks), dim3(64 * 16 / TPW), 0, 0, out, quads);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns_quad = ms * 1e6 / quads;
    const double mfma_ns = 9 * 72 * 8.0;   // per SIMD: 648 MFMAs of 16 cycles = 8.0 ns at 2.0 GHz (mfma_bf16_chain)
    printf("%-44s %8.1f ns per board quad   (648 MFMAs per SIMD = %.0f ns at 8 ns each: %.2f)\n", name, ns_quad, mfma_ns, mfma_ns / ns_quad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("error: %s\n", hipGetErrorString(e));
    hipFree(out);
}

int main() {
    run<4>("A: one wave a board, four tiles a wave");
    run<2>("B: two waves a board, two tiles a wave");
    run<4, true>("C: as A, 18 shared B reads a chunk");
    run<4>("A again");
    run<2>("B again");
    run<4, true>("C again");
    return 0;
}
