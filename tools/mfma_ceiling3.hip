// Microbenchmark 3: the GEMM tile loop of k_eval_f32 in isolation, features added one at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// MODE bit0: LDS-fed A; bit1: restage (global load + ds_write per 32 MFMA); bit2: barrier per chunk;
// bit3: GEMM2 phase (relu + 4 accumulators) after every 120 MFMAs
template <int MODE>
__global__ __launch_bounds__(256) void chain(float* out, const f32x4* src, int iters) {
    __shared__ f32x4 ring[3 * 8 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 3 * 8 * 64; i += 256) ring[i] = f32x4{1.f, .5f, .25f, .125f};
    __syncthreads();
    float x1[120];
    for (int i = 0; i < 120; ++i) x1[i] = 0.001f * (lane + i);
    f32x16 acc3[4];
    for (int c = 0; c < 4; ++c) for (int t = 0; t < 16; ++t) acc3[c][t] = t;
    f32x4 pre[2];
    pre[0] = ring[lane];
    f32x4 stage[2];
    const f32x4* p = src + wave * 128 + lane;
    stage[0] = p[0]; stage[1] = p[64];
    int rd = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // one tile: 30 groups GEMM1 + (16 groups GEMM2) = 46 groups; chunk phase drifts, so use g%8 on a running counter
#pragma unroll
        for (int tile = 0; tile < 4; ++tile) {
            f32x16 acc2 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int g = 0; g < 46; ++g) {
                const int gg = tile * 46 + g, nx = gg + 1;
                if ((MODE & 2) && nx % 8 == 4) {
                    const int wr = (rd == 2 * 512) ? 0 : rd + 512;
                    if (!(MODE & 32)) { ring[wr + wave * 128 + lane] = stage[0];
                    ring[wr + wave * 128 + 64 + lane] = stage[1]; }
                    if (!(MODE & 16)) { stage[0] = p[0]; stage[1] = p[64];
                    p += 512; }
                }
                if (nx % 8 == 0) {
                    rd = (rd == 2 * 512) ? 0 : rd + 512;
                    if (MODE & 4) { __builtin_amdgcn_sched_barrier(0); __syncthreads(); __builtin_amdgcn_sched_barrier(0); }
                }
                if (MODE & 1) pre[nx & 1] = ring[rd + (nx % 8) * 64 + lane];
                const f32x4 w = (MODE & 1) ? pre[gg & 1] : f32x4{1.f, .5f, .25f, .125f};
                if (g < 30 || !(MODE & 8)) {
                    const int r = (4 * g) % 116;
                    acc2 = MF(w.x, x1[r + 0], acc2);
                    acc2 = MF(w.y, x1[r + 1], acc2);
                    acc2 = MF(w.z, x1[r + 2], acc2);
                    acc2 = MF(w.w, x1[r + 3], acc2);
                } else {
                    if (g == 30) for (int t = 0; t < 16; ++t) acc2[t] = fmaxf(acc2[t], 0.f);
                    const int c3 = (g - 30) / 4, tq = (g - 30) % 4;
                    acc3[c3] = MF(w.x, acc2[4 * tq + 0], acc3[c3]);
                    acc3[c3] = MF(w.y, acc2[4 * tq + 1], acc3[c3]);
                    acc3[c3] = MF(w.z, acc2[4 * tq + 2], acc3[c3]);
                    acc3[c3] = MF(w.w, acc2[4 * tq + 3], acc3[c3]);
                }
            }
            if (!(MODE & 8)) for (int t = 0; t < 16; ++t) acc3[0][t] += acc2[t];
        }
        if (it % 4 == 3) p = src + wave * 128 + lane;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < 4; ++c) for (int t = 0; t < 16; ++t) s += acc3[c][t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;
}
template <int MODE>
void run(const char* name, const f32x4* src) {
    float* d; (void)hipMalloc(&d, (1 << 23) + 64);
    int iters = 200;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(chain<MODE>, dim3(256), dim3(256), 0, 0, d, src, iters);
        (void)hipDeviceSynchronize();
    }
    long long cyc; (void)hipMemcpy(&cyc, ((long long*)d) + (1 << 20), 8, hipMemcpyDeviceToHost);
    printf("%-60s cycles/MFMA %.2f\n", name, (double)cyc / (iters * 4 * 46 * 4.0));
    (void)hipFree(d);
}
int main() {
    f32x4* src; (void)hipMalloc(&src, 4 << 20); (void)hipMemset(src, 0, 4 << 20);
    run<3 + 16>("A from LDS + ds_write only", src);
    run<3 + 32>("A from LDS + global loads only", src);
    run<1>("+ A from LDS", src);
    run<3>("+ restage (global load, ds_write)", src);
    run<7>("+ barrier per chunk", src);
    run<15>("+ relu and GEMM2 with 4 accumulators", src);
    run<8>("relu and GEMM2 only (A constant)", src);
    return 0;
}
