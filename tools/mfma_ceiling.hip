// Microbenchmark: what rate does a chain of dependent v_mfma_f32_32x32x2_f32 reach with 1 or 2
// waves per SIMD?  (ceiling for k_eval_f32, which is made of exactly such chains)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void chain(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int t = 0; t < 16; ++t) acc[i][t] = threadIdx.x + t;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int t = 0; t < 16; ++t) s += acc[i][t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;
}
template <int NACC>
void run(const char* name, int blocks_per_cu, int lds_pad) {
    float* d; hipMalloc(&d, (1 << 23) + 64);
    int iters = 2000;
    int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(chain<NACC>, dim3(blocks), dim3(256), lds_pad, 0, d, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, ((long long*)d) + (1 << 20), 8, hipMemcpyDeviceToHost);
    double mfma_per_wave = (double)iters * 16 * NACC;
    double waves_per_simd = blocks_per_cu;  // 4 waves per block, 4 SIMDs
    printf("%-28s %8.3f ms  memtime cycles/MFMA/SIMD %.2f  (wave0: %lld cyc for %.0f MFMA; clock %.2f GHz by cyc/ms)\n", name, ms,
           (double)cyc / (mfma_per_wave * waves_per_simd), cyc, mfma_per_wave, cyc / (ms * 1e6));
    hipFree(d);
}
int main() {
    run<1>("1 wave/SIMD, 1 acc", 1, 0);
    run<1>("2 waves/SIMD, 1 acc", 2, 0);
    run<4>("1 wave/SIMD, 4 acc", 1, 0);
    run<4>("2 waves/SIMD, 4 acc", 2, 0);
    return 0;
}
