// Microbenchmark: bf16 MFMA shape against the clock the chip holds.  k_eval_bf16 is power-limited
// (DESIGN.md 3.2), so the question for the next kernel is which instruction delivers more FLOP/s at the
// clock it lets the chip run at: v_mfma_f32_32x32x16_bf16 (one 1-KiB A fragment from LDS feeds 2 MFMAs
// of 32 cycles) or v_mfma_f32_16x16x32_bf16 (one 1-KiB A fragment feeds 4 MFMAs of 16 cycles), both with
// random operands, one wave per SIMD, every CU busy, for a few milliseconds.
//   hipcc -O3 -Wno-inline-asm --offload-arch=gfx950 tools/mfma_shape_power.hip -o build/mfma_shape_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// SHAPE 0: 32x32x16, two column tiles per fragment; 1: 16x16x32, four column tiles per fragment
template <int SHAPE>
__global__ __launch_bounds__(256, 1) void probe(float* out, const f32x4* src, int iters) {
    __shared__ f32x4 ring[16 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) ring[i] = src[i];   // random bf16 pairs around 1.0
    __syncthreads();
    bf16x8 b[4];
    for (int t = 0; t < 4; ++t) b[t] = __builtin_bit_cast(bf16x8, src[1024 + 64 * t + lane]);
    f32x16 acc32[2];
    f32x4 acc16[4];
    for (int t = 0; t < 16; ++t) { acc32[0][t] = 0.f; acc32[1][t] = 0.f; }
    for (int t = 0; t < 4; ++t) acc16[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int LA = 4;
    f32x4 pre[LA + 1];
    const f32x4* rp = ring + lane;
    for (int i = 0; i < LA; ++i) pre[i] = rp[i * 64];
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 60; ++g) {
            pre[(g + LA) % (LA + 1)] = rp[((g + LA) & 15) * 64];
            const bf16x8 a = __builtin_bit_cast(bf16x8, pre[g % (LA + 1)]);
            if (SHAPE == 0) {
                acc32[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[0], acc32[0], 0, 0, 0);
                acc32[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[1], acc32[1], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc16[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[t], acc16[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int t = 0; t < 16; ++t) s += acc32[0][t] + acc32[1][t];
    for (int t = 0; t < 4; ++t) s += acc16[t].x + acc16[t].y + acc16[t].z + acc16[t].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        long long* o = reinterpret_cast<long long*>(out + (1 << 20)) + blockIdx.x * 2;
        o[0] = t1 - t0;
        o[1] = r1 - r0;
    }
}
template <int SHAPE>
void run(const char* name, float* d, const f32x4* src, int blocks) {
    const int iters = 2500;   // 150 000 fragments per wave: ~5 ms
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    long long cyc = 0, rt = 0;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<SHAPE>), dim3(blocks), dim3(256), 0, 0, d, src, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
        long long c[2];
        (void)hipMemcpy(c, d + (1 << 20), 16, hipMemcpyDeviceToHost);
        cyc = c[0];
        rt = c[1];
    }
    const double flop = (double)blocks * 4 * iters * 60.0 * 2 * 32768.0;   // per fragment: 2 x 32x32x16 or 4 x 16x16x32 = 65 536 FLOP
    printf("%-28s %7.3f ms  %7.1f TFLOP/s  cycles per fragment %.2f  clock %.3f GHz\n", name, best, flop / (best * 1e-3) / 1e12,
           (double)cyc / (iters * 60.0), (double)cyc / (double)rt * 0.1);
}
int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const size_t nwords = (1024 + 256) * 4;
    unsigned* h = (unsigned*)malloc(nwords * 4);
    srand(7);
    for (size_t i = 0; i < nwords; ++i) {
        const unsigned lo = 0x3F00u + (rand() & 0xFF) + ((rand() & 1) << 15), hi = 0x3F00u + (rand() & 0xFF) + ((rand() & 1) << 15);
        h[i] = lo | (hi << 16);   // bf16 values of magnitude 0.5 .. 1, random sign
    }
    f32x4* src;
    (void)hipMalloc(&src, nwords * 4);
    (void)hipMemcpy(src, h, nwords * 4, hipMemcpyHostToDevice);
    float* d;
    (void)hipMalloc(&d, (1 << 22) + 8192);
    run<0>("v_mfma_f32_32x32x16_bf16", d, src, cus);
    run<1>("v_mfma_f32_16x16x32_bf16", d, src, cus);
    run<0>("v_mfma_f32_32x32x16_bf16", d, src, cus);
    run<1>("v_mfma_f32_16x16x32_bf16", d, src, cus);
    return 0;
}
