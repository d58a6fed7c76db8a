// Microbenchmark: does it matter where a v_mfma_f32_32x32x16_bf16 chain keeps its accumulator
// (VGPRs or AGPRs) while its A fragments stream in from LDS (one ds_read_b128 per MFMA, a few ahead)?
//   hipcc -O3 -Wno-inline-asm --offload-arch=gfx950 tools/mfma_regclass.hip -o build/mfma_regclass
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ACC: 0 accumulator in AGPRs, 1 in VGPRs.  BSRC: 0 B operand in VGPRs, 1 in AGPRs.
// READS: ds_read_b128 per MFMA (0, 1 or 2; the second one is not consumed).  CHAINS: 1 or 2 accumulators in turn.
template <int ACC, int BSRC, int READS, int CHAINS, int SAME = 0>
__global__ __launch_bounds__(256, 1) void probe(float* out, int iters) {
    __shared__ f32x4 ring[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) ring[i] = f32x4{1.f, .5f, .25f, .125f};
    __syncthreads();
    f32x16 acc0, acc1;
    for (int t = 0; t < 16; ++t) { acc0[t] = t; acc1[t] = -t; }
    f32x4 b = f32x4{0.5f, 0.25f, 1.f, 2.f};
    constexpr int LA = 4;
    f32x4 pre[LA + 1], junk[LA + 1];
    // SAME: all four waves of the workgroup read the same fragments (as k_eval_bf16's waves do), else each its own quarter
    const f32x4* rp = ring + (SAME ? 0 : ((wave + 1) & 3) * 1024) + lane;
    for (int i = 0; i < LA; ++i) pre[i] = rp[i * 64];
    for (int i = 0; i <= LA; ++i) junk[i] = b;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 60; ++g) {
            if (READS >= 1) pre[(g + LA) % (LA + 1)] = rp[((g + LA) & 15) * 64];
            if (READS >= 2) junk[(g + LA) % (LA + 1)] = rp[((g + LA) & 15) * 64 + 1024];
            const f32x4 a = READS ? pre[g % (LA + 1)] : b;
            if (READS >= 2) asm volatile("" ::"v"(junk[g % (LA + 1)]));
            f32x16& acc = (CHAINS == 2 && (g & 1)) ? acc1 : acc0;
            if (ACC == 0 && BSRC == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
            if (ACC == 1 && BSRC == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            if (ACC == 0 && BSRC == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(b));
            if (ACC == 1 && BSRC == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int t = 0; t < 16; ++t) s += acc0[t] + acc1[t];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;
}
template <int ACC, int BSRC, int READS, int CHAINS, int SAME = 0>
void run(const char* name, float* d) {
    const int iters = 100;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<ACC, BSRC, READS, CHAINS, SAME>), dim3(256), dim3(256), 0, 0, d, iters);
        (void)hipDeviceSynchronize();
    }
    long long cyc;
    (void)hipMemcpy(&cyc, ((long long*)d) + (1 << 20), 8, hipMemcpyDeviceToHost);
    printf("%-72s ticks/MFMA %.2f\n", name, (double)cyc / (iters * 60.0));
}
int main() {
    float* d;
    (void)hipMalloc(&d, (1 << 23) + 64);
    run<0, 0, 0, 1>("acc AGPR, B VGPR, no LDS reads, 1 chain", d);
    run<1, 0, 0, 1>("acc VGPR, B VGPR, no LDS reads, 1 chain", d);
    run<0, 0, 1, 1>("acc AGPR, B VGPR, A from LDS, 1 chain", d);
    run<1, 0, 1, 1>("acc VGPR, B VGPR, A from LDS, 1 chain", d);
    run<0, 1, 1, 1>("acc AGPR, B AGPR, A from LDS, 1 chain", d);
    run<1, 1, 1, 1>("acc VGPR, B AGPR, A from LDS, 1 chain", d);
    run<0, 0, 1, 2>("acc AGPR, B VGPR, A from LDS, 2 chains", d);
    run<1, 0, 1, 2>("acc VGPR, B VGPR, A from LDS, 2 chains", d);
    run<1, 1, 1, 2>("acc VGPR, B AGPR, A from LDS, 2 chains", d);
    run<0, 0, 2, 1>("acc AGPR, B VGPR, A from LDS + 1 more ds_read_b128, 1 chain", d);
    run<1, 0, 2, 1>("acc VGPR, B VGPR, A from LDS + 1 more ds_read_b128, 1 chain", d);
    run<1, 1, 2, 1>("acc VGPR, B AGPR, A from LDS + 1 more ds_read_b128, 1 chain", d);
    run<1, 1, 1, 1, 1>("acc VGPR, B AGPR, A from LDS, 1 chain, all waves read the same fragments", d);
    run<1, 1, 2, 1, 1>("acc VGPR, B AGPR, A + 1 more ds_read_b128, all waves the same fragments", d);
    run<0, 0, 1, 1, 1>("acc AGPR, B VGPR, A from LDS, 1 chain, all waves read the same fragments", d);
    return 0;
}
