// Cycles per v_mfma_f32_16x16x16_bf16 (the K = 16 form) against v_mfma_f32_16x16x32_bf16, one wave per SIMD, four
// independent accumulators: does a half k-step cost half an instruction?  (The 236->512 layer of k_eval_t16 pads
// 239 used k-slots to 8 steps of 32; 7 steps + one K = 16 step would cover 240.)
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_k16.hip -o build/mfma_k16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// MIX: of every 8 MFMAs, this many are the K = 16 form (0: all K = 32, 8: all K = 16, 1: the kernel's ratio)
template <int MIX>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cycles, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 bsrc = f32x4{(float)lane * 1e-3f, 1.f, 2.f, 3.f};
    const bf16x8 b = __builtin_bit_cast(bf16x8, bsrc);
    const bf16x8 w = __builtin_bit_cast(bf16x8, f32x4{1.f, 0.5f, (float)lane, 0.125f});
    const s16x4 b4 = s16x4{1, 2, (short)lane, 4}, w4 = s16x4{(short)(lane * 3), 5, 6, 7};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if ((g & 7) < MIX)
                acc[g & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(w4, b4, acc[g & 3], 0, 0, 0);
            else
                acc[g & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, b, acc[g & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < 4; ++c)
        for (int t = 0; t < 4; ++t) s += acc[c][t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MIX>
void run(const char* name) {
    const int blocks = 256, iters = 4000;
    float* out; long long* cyc;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(long long));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MIX>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[4];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MIX>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %7.2f cycles per MFMA   %7.2f ns per MFMA\n", name, (double)h[0] / (iters * 16.0), ms * 1e6 / (iters * 16.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("all 16x16x32");
    run<8>("all 16x16x16");
    run<1>("1 of 8 is 16x16x16");
    return 0;
}
