// Microbenchmark 2: dependent fp32 MFMA chain fed the way k_eval_f32's GEMM loop feeds it:
// A from an LDS read one group ahead (ds_read_b128 per 4 MFMAs), B from distinct registers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void chain(float* out, int iters) {
    __shared__ f32x4 ring[8 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8 * 64; i += 256) ring[i] = f32x4{1.f, .5f, .25f, .125f};
    __syncthreads();
    float b[32];
    for (int i = 0; i < 32; ++i) b[i] = 0.001f * (lane + i);
    f32x16 acc;
    for (int t = 0; t < 16; ++t) acc[t] = t;
    f32x4 pre[2];
    pre[0] = ring[lane];
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MODE >= 1) pre[(g + 1) & 1] = ring[((g + 1) & 7) * 64 + lane];
            f32x4 w = (MODE >= 1) ? pre[g & 1] : f32x4{1.f, .5f, .25f, .125f};
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, b[4 * g + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, b[4 * g + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, b[4 * g + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, b[4 * g + 3], acc, 0, 0, 0);
        }
        if (MODE >= 2) __syncthreads();
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int t = 0; t < 16; ++t) s += acc[t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;
}
template <int MODE>
void run(const char* name, int blocks_per_cu) {
    float* d; (void)hipMalloc(&d, (1 << 23) + 64);
    int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(chain<MODE>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, d, iters);
        (void)hipDeviceSynchronize();
    }
    long long cyc; (void)hipMemcpy(&cyc, ((long long*)d) + (1 << 20), 8, hipMemcpyDeviceToHost);
    printf("%-44s cycles/MFMA/SIMD %.2f\n", name, (double)cyc / (iters * 32.0 * blocks_per_cu));
    (void)hipFree(d);
}
int main() {
    run<0>("regs only, 1 wave/SIMD", 1);
    run<1>("+ ds_read_b128 per 4 MFMA, 1 wave/SIMD", 1);
    run<2>("+ barrier per 32 MFMA, 1 wave/SIMD", 1);
    run<1>("+ ds_read_b128 per 4 MFMA, 2 waves/SIMD", 2);
    run<2>("+ barrier per 32 MFMA, 2 waves/SIMD", 2);
    return 0;
}
