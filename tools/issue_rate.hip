// Microbenchmark: how many other instructions one wave per SIMD can issue per v_mfma_f32_32x32x16_bf16
// before the matrix pipe starts to idle, by kind of filler instruction.
//   hipcc -O3 -Wno-inline-asm --offload-arch=gfx950 tools/issue_rate.hip -o build/issue_rate && build/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// KIND: 0 v_add_u32 on K independent registers; 1 v_accvgpr_read; 2 s_add_u32; 3 s_waitcnt lgkmcnt(15) (never waits);
//       4 ds_read_b128 (4 in flight, waited for 4 later); 5 dependent chain v_cvt_pk_bf16_f32 -> v_pk_max_i16;
//       6 ds_read_u8; 7 s_nop 0
template <int KIND, int K>
__global__ __launch_bounds__(256, 1) void probe(float* out, int iters) {
    __shared__ f32x4 ring[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += 256) ring[i] = f32x4{1.f, .5f, .25f, .125f};
    __syncthreads();
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
    f32x16 acc[2];
    for (int t = 0; t < 16; ++t) { acc[0][t] = t; acc[1][t] = -t; }
    unsigned x[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    unsigned sreg = 0;
    f32x4 q[4] = {ring[lane], ring[lane + 64], ring[lane + 128], ring[lane + 192]};
    unsigned char c8[4] = {0, 0, 0, 0};
    const unsigned char* bytes = reinterpret_cast<const unsigned char*>(ring) + lane;
    float p0 = lane, p1 = -lane;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 64; ++g) {
            acc[g & 1] = MF(a, b, acc[g & 1]);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) asm volatile("v_add_u32 %0, %0, 1" : "+v"(x[k & 7]));
                if (KIND == 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x[k & 7]) : "a"(acc[(g + 1) & 1][k & 15]));
                if (KIND == 2) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg) :: "scc");
                if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
                if (KIND == 4) q[(g * K + k) & 3] = ring[lane + 64 * ((g * K + k) & 15)];
                if (KIND == 5) {
                    if (k & 1) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(x[0]));
                    else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[0]) : "v"(p0), "v"(p1));
                }
                if (KIND == 6) c8[(g * K + k) & 3] = bytes[(g * K + k) & 1023];
                if (KIND == 7) asm volatile("s_nop 0");
            }
            if (KIND == 4) asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int t = 0; t < 16; ++t) s += acc[0][t] + acc[1][t];
    for (int k = 0; k < 8; ++k) s += x[k];
    for (int k = 0; k < 4; ++k) s += q[k].x + c8[k];
    s += sreg;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;
}
template <int KIND, int K>
double run1(float* d) {
    const int iters = 50;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<KIND, K>), dim3(256), dim3(256), 0, 0, d, iters);
        (void)hipDeviceSynchronize();
    }
    long long cyc;
    (void)hipMemcpy(&cyc, ((long long*)d) + (1 << 20), 8, hipMemcpyDeviceToHost);
    return (double)cyc / (iters * 64.0);
}
template <int KIND>
void sweep(const char* name, float* d) {
    printf("%-44s", name);
    printf(" K=0 %.1f", run1<KIND, 0>(d));
    printf("  2 %.1f", run1<KIND, 2>(d));
    printf("  4 %.1f", run1<KIND, 4>(d));
    printf("  6 %.1f", run1<KIND, 6>(d));
    printf("  8 %.1f", run1<KIND, 8>(d));
    printf("  12 %.1f", run1<KIND, 12>(d));
    printf("  16 %.1f\n", run1<KIND, 16>(d));
}
int main() {
    float* d;
    (void)hipMalloc(&d, (1 << 23) + 64);
    printf("ticks per MFMA with K filler instructions in each gap (one wave per SIMD)\n");
    sweep<0>("v_add_u32 (independent)", d);
    sweep<1>("v_accvgpr_read_b32", d);
    sweep<2>("s_add_u32", d);
    sweep<3>("s_waitcnt lgkmcnt(15)", d);
    sweep<7>("s_nop 0", d);
    sweep<4>("ds_read_b128", d);
    sweep<6>("ds_read_u8", d);
    sweep<5>("v_cvt_pk_bf16_f32 -> v_pk_max_i16 chain", d);
    return 0;
}
