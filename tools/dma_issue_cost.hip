// Microbenchmark: what one 1-KiB LDS-DMA piece costs the wave that issues it between bf16 MFMAs
// (one wave per SIMD, 4 waves per workgroup, as k_eval_bf16 runs), for the ways of addressing it.
//   hipcc -O3 --offload-arch=gfx950 tools/dma_issue_cost.hip -o build/dma_issue_cost && build/dma_issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// V: 0 none; 1 global_load_lds, 64-bit VGPR address; 2 buffer_load offen lds (32-bit VGPR offset);
//    3 buffer_load lds, no VGPR (descriptor adds the lane id, stride 16); 4 global_load_lds, SGPR base + VGPR offset
//    5 as 1 but four global_load_lds_dword (256 B each) instead of one dwordx4
// EVERY: one piece per EVERY groups of 2 MFMAs.  BURST: pieces issued back to back at that point.
template <int V, int EVERY, int BURST, int LDSREAD = 0>
__global__ __launch_bounds__(256, 1) void probe(float* out, const f32x4* src, int iters) {
    __shared__ f32x4 ring[4 * 16 * 64];  // 64 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4 * 16 * 64; i += 256) ring[i] = f32x4{1.f, .5f, .25f, .125f};
    __syncthreads();
    typedef __attribute__((address_space(3))) f32x4* lds_ptr;
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr)(ring + wave * 16 * 64);
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
    f32x16 acc[2];
    for (int t = 0; t < 16; ++t) { acc[0][t] = t; acc[1][t] = -t; }
    const f32x4* gp = src + wave * 4096 + lane;
    const unsigned voff = (wave * 4096 + lane) * 16;
    i32x4 rsrc;
    rsrc.x = (int)(uintptr_t)src;
    rsrc.y = (int)(((uintptr_t)src) >> 32) | (V == 3 ? (16 << 16) : 0);
    rsrc.z = V == 3 ? (1 << 20) : (16 << 20);  // records: elements of 16 B (strided) or bytes
    rsrc.w = V == 3 ? (1 << 23) : 0x00020000;
    f32x4 pre[4];
    const f32x4* rp = ring + ((wave + 1) & 3) * 16 * 64 + lane;  // another wave's part of the ring, as the kernel's A fragments
    for (int i = 0; i < 3; ++i) pre[i] = rp[i * 64];
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 64; ++g) {
            if (LDSREAD) {
                pre[(g + 3) & 3] = rp[((g + 3) & 15) * 64];
                a = __builtin_bit_cast(bf16x8, pre[g & 3]);
                if (LDSREAD > 1) b = __builtin_bit_cast(bf16x8, rp[((g + 5) & 15) * 64 + 1024 * (LDSREAD - 2)]);
            }
            acc[0] = MF(a, b, acc[0]);
            acc[1] = MF(a, b, acc[1]);
            if (V != 0 && g % EVERY == EVERY - 1) {
#pragma unroll
                for (int q = 0; q < BURST; ++q) {
                    const int piece = ((g / EVERY) * BURST + q) % 16;
                    const unsigned m0v = lds0 + piece * 1024;
                    if (V == 1)
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(gp + piece * 64) : "memory", "m0");
                    if (V == 2)
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff + piece * 1024), "s"(rsrc) : "memory", "m0");
                    if (V == 3)
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 off, %1, %2 lds" ::"s"(m0v), "s"(rsrc), "s"((wave * 4096 + piece * 64) * 16) : "memory", "m0");
                    if (V == 4)
                        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff + piece * 1024), "s"(src) : "memory", "m0");
                    if (V == 5) {
#pragma unroll
                        for (int d = 0; d < 4; ++d)
                            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(m0v + d * 256), "v"((const float*)(gp + piece * 64) + d * 64) : "memory", "m0");
                    }
                }
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int t = 0; t < 16; ++t) s += acc[0][t] + acc[1][t];
    s += ring[(lane * 7 + wave * 16 * 64) % (4 * 16 * 64)].x;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;
}
template <int V, int EVERY, int BURST, int LDSREAD = 0>
void run(const char* name, const f32x4* src, double base) {
    float* d;
    (void)hipMalloc(&d, (1 << 23) + 64);
    const int iters = 100;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<V, EVERY, BURST, LDSREAD>), dim3(256), dim3(256), 0, 0, d, src, iters);
        (void)hipDeviceSynchronize();
    }
    long long cyc;
    (void)hipMemcpy(&cyc, ((long long*)d) + (1 << 20), 8, hipMemcpyDeviceToHost);
    const double per_group = (double)cyc / (iters * 64.0);
    const double pieces_per_group = V ? (double)BURST / EVERY : 0.0;
    printf("%-64s ticks/group %.2f", name, per_group);
    if (V) printf("   extra per piece vs no-DMA bare loop %.1f", (per_group - base) / pieces_per_group);
    printf("\n");
    (void)hipFree(d);
}
int main() {
    f32x4* src;
    (void)hipMalloc(&src, 16 << 20);
    (void)hipMemset(src, 0, 16 << 20);
    float* d;
    (void)hipMalloc(&d, (1 << 23) + 64);
    hipLaunchKernelGGL((probe<0, 4, 1>), dim3(256), dim3(256), 0, 0, d, src, 100);
    hipLaunchKernelGGL((probe<0, 4, 1>), dim3(256), dim3(256), 0, 0, d, src, 100);
    (void)hipDeviceSynchronize();
    long long cyc;
    (void)hipMemcpy(&cyc, ((long long*)d) + (1 << 20), 8, hipMemcpyDeviceToHost);
    const double base = (double)cyc / (100 * 64.0);
    printf("%-64s ticks/group %.2f  (2 MFMAs)\n", "no DMA", base);
    run<1, 4, 1>("global_load_lds_dwordx4 v[64-bit], 1 per 4 groups", src, base);
    run<2, 4, 1>("buffer_load_dwordx4 offen lds, 1 per 4 groups", src, base);
    run<3, 4, 1>("buffer_load_dwordx4 lds, lane id from descriptor, 1 per 4", src, base);
    run<4, 4, 1>("global_load_lds_dwordx4 v_off, s[base], 1 per 4 groups", src, base);
    run<5, 4, 1>("4 x global_load_lds_dword, 1 per 4 groups", src, base);
    run<1, 16, 4>("global_load_lds_dwordx4 v[64-bit], 4 back to back per 16", src, base);
    run<3, 16, 4>("buffer_load lds no-VGPR, 4 back to back per 16", src, base);
    run<1, 1, 1>("global_load_lds_dwordx4 v[64-bit], 1 per group", src, base);
    run<3, 1, 1>("buffer_load lds no-VGPR, 1 per group", src, base);
    printf("-- with one ds_read_b128 of an A fragment per group (3 groups ahead)\n");
    run<0, 4, 1, 1>("no DMA, A from LDS", src, base);
    run<1, 4, 1, 1>("global_load_lds_dwordx4 v[64-bit], 1 per 4 groups, A from LDS", src, base);
    run<2, 4, 1, 1>("buffer_load_dwordx4 offen lds, 1 per 4 groups, A from LDS", src, base);
    run<3, 4, 1, 1>("buffer_load lds no-VGPR, 1 per 4 groups, A from LDS", src, base);
    run<4, 4, 1, 1>("global_load_lds_dwordx4 v_off, s[base], 1 per 4, A from LDS", src, base);
    printf("-- with A and B from LDS\n");
    run<0, 4, 1, 2>("no DMA, A and B from LDS", src, base);
    run<1, 4, 1, 2>("global_load_lds_dwordx4 v[64-bit], 1 per 4 groups, A+B from LDS", src, base);
    run<3, 4, 1, 2>("buffer_load lds no-VGPR, 1 per 4 groups, A+B from LDS", src, base);
    return 0;
}
