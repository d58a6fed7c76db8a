// Cycles per v_mfma_f32_16x16x32_bf16 in one wave per SIMD: one dependent accumulator chain against two
// and four independent ones, with and without a ds_read_b128 per MFMA three ahead (the shape of
// k_eval_t16s: one column tile per wave = one chain per layer; k_eval_t16 runs four).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_bf16_chain.hip -o build/mfma_bf16_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, bool LDS>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cycles, int iters) {
    __shared__ f32x4 ring[16 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) ring[i] = f32x4{1.f, 0.5f, 0.25f, 0.125f};
    __syncthreads();
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 bsrc = f32x4{(float)lane * 1e-3f, 1.f, 2.f, 3.f};
    const bf16x8 b = __builtin_bit_cast(bf16x8, bsrc);
    f32x4 pre[4];
    for (int i = 0; i < 4; ++i) pre[i] = ring[i * 64 + lane];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const bf16x8 w = __builtin_bit_cast(bf16x8, pre[g & 3]);
            if (LDS) pre[(g + 3) & 3] = ring[((g + 3) & 15) * 64 + lane];
            acc[g % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b, acc[g % CHAINS], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int t = 0; t < 4; ++t) s += acc[c][t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int CHAINS, bool LDS>
void run(const char* name, int blocks) {
    float* out; long long* cyc;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(long long));
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<CHAINS, LDS>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[4];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    // s_memtime counts at 100 MHz on this part: report against the wall clock of the kernel instead
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<CHAINS, LDS>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %7.2f memtime ticks per MFMA   %7.2f ns per MFMA\n", name, (double)h[0] / (iters * 16.0), ms * 1e6 / (iters * 16.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    const int blocks = 256;
    run<1, false>("1 chain", blocks);
    run<2, false>("2 chains", blocks);
    run<4, false>("4 chains", blocks);
    run<1, true>("1 chain  + ds_read_b128 per MFMA", blocks);
    run<2, true>("2 chains + ds_read_b128 per MFMA", blocks);
    run<4, true>("4 chains + ds_read_b128 per MFMA", blocks);
    return 0;
}
