// Cycles per v_mfma_f32_32x32x2_f32 in one wave per SIMD: one dependent accumulator chain against two
// and four independent ones, with and without a ds_read_b128 + counted wait per four MFMAs (the shape of
// k_eval_f32's 1x1 pair).    hipcc -O3 --offload-arch=gfx950 tools/mfma_f32_chain.hip -o build/mfma_f32_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, bool LDS>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cycles, int iters) {
    __shared__ f32x4 ring[8 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8 * 64; i += 256) ring[i] = f32x4{1.f, 0.5f, 0.25f, 0.125f};
    __syncthreads();
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int t = 0; t < 16; ++t) acc[c][t] = 0.f;
    float b = (float)lane * 1e-3f;
    f32x4 w = ring[lane];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f32x4 wn = w;
            if (LDS) wn = ring[((g + 1) & 7) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = (4 * g + j) % CHAINS;
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j], b, acc[c], 0, 0, 0);
            }
            w = wn;
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int t = 0; t < 16; ++t) s += acc[c][t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int CHAINS, bool LDS>
void run(const char* name, int blocks) {
    float* out; long long* cyc;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&cyc, blocks * sizeof(long long));
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<CHAINS, LDS>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[4];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("%-44s %6.2f cycles per MFMA\n", name, (double)h[0] / (iters * 32.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    const int blocks = 256;   // one workgroup of four waves per CU
    run<1, false>("1 chain", blocks);
    run<2, false>("2 chains", blocks);
    run<4, false>("4 chains", blocks);
    run<1, true>("1 chain  + ds_read_b128 per 4 MFMAs", blocks);
    run<2, true>("2 chains + ds_read_b128 per 4 MFMAs", blocks);
    run<1, false>("1 chain, 2 workgroups per CU", 512);
    return 0;
}
