// Do v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 round alike?  (an fp32 small-call
// kernel must be bit-identical to k_eval_f32, which pins the accumulation; a 16x16 tiling would let the 32-channel
// 3x3 layers be split over more waves.)  One wave computes a 32x32 tile of A(32xK) B(Kx32) + C three ways and the
// host compares bits: (a) K/2 steps of 32x32x2, (b) K/4 steps of 16x16x4 on the four 16x16 sub-tiles, (c) on the host,
// fmaf in k order.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_shape_bits.hip -o build/tools/mfma_shape_bits && build/tools/mfma_shape_bits
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 64;

__global__ void k_shapes(const float* A, const float* B, const float* C, float* D32, float* D16) {
    const int l = threadIdx.x;
    {   // (a) 32x32x2
        f32x16 acc;
        for (int t = 0; t < 16; ++t) acc[t] = C[((t & 3) + 8 * (t >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
        for (int s = 0; s < K / 2; ++s) {
            const int k = 2 * s + (l >> 5);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * K + k], B[k * 32 + (l & 31)], acc, 0, 0, 0);
        }
        for (int t = 0; t < 16; ++t) D32[((t & 3) + 8 * (t >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[t];
    }
    for (int ti = 0; ti < 2; ++ti)
        for (int tj = 0; tj < 2; ++tj) {   // (b) 16x16x4 on sub-tile (ti, tj)
            f32x4 acc;
            for (int t = 0; t < 4; ++t) acc[t] = C[(16 * ti + 4 * (l >> 4) + t) * 32 + 16 * tj + (l & 15)];
            for (int s = 0; s < K / 4; ++s) {
                const int k = 4 * s + (l >> 4);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(16 * ti + (l & 15)) * K + k], B[k * 32 + 16 * tj + (l & 15)], acc, 0, 0, 0);
            }
            for (int t = 0; t < 4; ++t) D16[(16 * ti + 4 * (l >> 4) + t) * 32 + 16 * tj + (l & 15)] = acc[t];
        }
}

int main() {
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> A(32 * K), B(K * 32), C(32 * 32), D32(1024), D16(1024), H(1024), H2(1024);
    for (auto& v : A) v = nd(rng);
    for (auto& v : B) v = nd(rng);
    for (auto& v : C) v = nd(rng);
    float *dA, *dB, *dC, *d32, *d16;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&d32, 4096); hipMalloc(&d16, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_shapes, dim3(1), dim3(64), 0, 0, dA, dB, dC, d32, d16);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(D32.data(), d32, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(D16.data(), d16, 4096, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float acc = C[i * 32 + j];
            for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B[k * 32 + j], acc);
            H[i * 32 + j] = acc;
            float a2 = C[i * 32 + j];   // pairs summed exactly first, then one rounding per 32x32x2 step
            for (int k = 0; k < K; k += 2) a2 = (float)((double)a2 + (double)A[i * K + k] * B[k * 32 + j] + (double)A[i * K + k + 1] * B[(k + 1) * 32 + j]);
            H2[i * 32 + j] = a2;
        }
    int d_shapes = 0, d_fma32 = 0, d_fma16 = 0, d_pair32 = 0;
    double worst = 0;
    for (int i = 0; i < 1024; ++i) {
        d_shapes += std::memcmp(&D32[i], &D16[i], 4) != 0;
        d_fma32 += std::memcmp(&D32[i], &H[i], 4) != 0;
        d_fma16 += std::memcmp(&D16[i], &H[i], 4) != 0;
        d_pair32 += std::memcmp(&D32[i], &H2[i], 4) != 0;
        worst = std::fmax(worst, std::fabs((double)D32[i] - D16[i]));
    }
    printf("K = %d, 1024 elements: 32x32x2 vs 16x16x4 differ in %d (worst |diff| %.3g); 32x32x2 vs host fmaf chain %d; 16x16x4 vs host fmaf chain %d; 32x32x2 vs host 'pair then round' %d\n",
           K, d_shapes, worst, d_fma32, d_fma16, d_pair32);
    return 0;
}
