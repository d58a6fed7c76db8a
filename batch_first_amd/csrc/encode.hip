// Encoding kernels: node records or wire format -> 4-bit mailboxes / float planes.
//
// Reference path replaced: struct_array_to_ann_inputs (numba_negamax_zero_window.py:71-105)
// and parse_into_ann_input_inference (ann_creation_helper.py:21-84).  All integer work; the
// planes kernel is a pure HBM writer (3840 B out per board, 16-byte stores).
#include <hip/hip_runtime.h>

#include "bf_device.h"
#include "bf_encode.h"
#include "bf_kernels.h"

namespace bf {

namespace {

constexpr int kScanBoardsPerBlock = 1024;  // 256 threads x 4 boards

// Per-board nibble offsets, level 1: offsets local to each 1024-board block + block totals.
__global__ __launch_bounds__(256) void k_popcount_scan(const uint64_t* __restrict__ occ, unsigned n,
                                                        unsigned* __restrict__ local_off,
                                                        unsigned* __restrict__ block_total) {
    __shared__ unsigned wave_sum[4];
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned b0 = blockIdx.x * kScanBoardsPerBlock + tid * 4;
    unsigned pc[4];
    unsigned mine = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        pc[i] = (b0 + i < n) ? (unsigned)__popcll(occ[b0 + i]) : 0u;
        mine += pc[i];
    }
    unsigned incl = mine;  // inclusive scan over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned v = __shfl_up(incl, d);
        if (lane >= (unsigned)d) incl += v;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    unsigned base = 0;
    for (unsigned w = 0; w < wave; ++w) base += wave_sum[w];
    unsigned run = base + incl - mine;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (b0 + i < n) local_off[b0 + i] = run;
        run += pc[i];
    }
    if (tid == 255) block_total[blockIdx.x] = run;
}

// Level 2: one block turns block totals into exclusive block bases and validates the stream
// length (own_concat pads an odd count with one nibble: numba_negamax_zero_window.py:58,65-66).
__global__ __launch_bounds__(1024) void k_scan_block_totals(unsigned* __restrict__ block_total, unsigned n_blocks,
                                                             unsigned long long n_bytes, unsigned start_nibble,
                                                             int* __restrict__ err_flag) {
    __shared__ unsigned wave_sum[16];
    __shared__ unsigned carry_s;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (unsigned start = 0; start < n_blocks; start += 1024) {
        const unsigned i = start + tid;
        const unsigned mine = (i < n_blocks) ? block_total[i] : 0u;
        unsigned incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            unsigned v = __shfl_up(incl, d);
            if (lane >= (unsigned)d) incl += v;
        }
        if (lane == 63) wave_sum[wave] = incl;
        __syncthreads();
        unsigned base = carry_s;
        for (unsigned w = 0; w < wave; ++w) base += wave_sum[w];
        if (i < n_blocks) block_total[i] = base + incl - mine;
        __syncthreads();
        if (tid == 1023) carry_s = base + incl;
        __syncthreads();
    }
    if (tid == 0) {
        // a shard of a longer stream may begin on the low nibble of its first byte (start_nibble = 1)
        const unsigned long long total = (unsigned long long)carry_s + start_nibble;
        if ((total + 1) / 2 != n_bytes) atomicOr(err_flag, 1);
    }
}

// ---- per-square code sources -----------------------------------------------------------

struct WireSrc {
    const uint8_t* nib;
    unsigned long long n_bytes;
    const uint64_t* occ;
    const unsigned* local_off;
    const unsigned* block_base;
    int* err_flag;
    unsigned start_nibble;   // 0, or 1 when the boards are a shard that begins mid-byte
    // ann_creation_helper.py:55-79: the k-th set bit of the mask takes stream nibble off+k
    __device__ __forceinline__ int code(unsigned b, int s) const {
        const uint64_t m = occ[b];
        if (!((m >> s) & 1ull)) return 15;
        const unsigned long long nibble =
            (unsigned long long)start_nibble + local_off[b] + block_base[b / kScanBoardsPerBlock] +
            __popcll(m & ((1ull << s) - 1ull));
        const unsigned long long byte = nibble >> 1;
        if (byte >= n_bytes) {  // never read past the stream; the length check reports it
            atomicOr(err_flag, 1);
            return 15;
        }
        const unsigned v = nib[byte];
        const int c = (nibble & 1ull) ? (int)(v & 15u) : (int)(v >> 4);  // even nibble = high half, :38-39
        if (c == 15) atomicOr(err_flag, 2);
        return c;
    }
};

struct StructSrc {
    const unsigned char* recs;
    unsigned long long stride;
    __device__ __forceinline__ int code(unsigned b, int s) const {
        const BoardBits bits = load_board(recs + (unsigned long long)b * stride);
        return square_code(bits, relevance_mask(bits), s);
    }
};

struct MailboxSrc {
    const uint8_t* mailbox;
    __device__ __forceinline__ int code(unsigned b, int s) const {
        const unsigned v = mailbox[(unsigned long long)b * 32 + (s >> 1)];
        return (s & 1) ? (int)(v >> 4) : (int)(v & 15u);
    }
};

// 4 boards per 256-thread block, one thread per square.
template <class Src>
__global__ __launch_bounds__(256) void k_to_mailbox(Src src, unsigned n, uint8_t* __restrict__ mailbox) {
    const unsigned tid = threadIdx.x;
    const unsigned b = blockIdx.x * 4 + (tid >> 6);
    const int s = tid & 63;
    int c = 15;
    if (b < n) c = src.code(b, s);
    const int hi = __shfl_down(c, 1);
    if (b < n && !(s & 1)) mailbox[(unsigned long long)b * 32 + (s >> 1)] = (uint8_t)(c | (hi << 4));
}

// planes[b][code][rank][file] = 1.0f (ann_creation_helper.py:76-79), written as float4.
template <class Src>
__global__ __launch_bounds__(256) void k_to_planes(Src src, unsigned n, float* __restrict__ planes) {
    __shared__ __attribute__((aligned(16))) unsigned char codes[4][64];
    const unsigned tid = threadIdx.x;
    const unsigned b0 = blockIdx.x * 4;
    {
        const unsigned b = b0 + (tid >> 6);
        codes[tid >> 6][tid & 63] = (b < n) ? (unsigned char)src.code(b, tid & 63) : (unsigned char)15;
    }
    __syncthreads();
    f32x4* out = reinterpret_cast<f32x4*>(planes);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned idx4 = tid + 256u * i;  // 960 float4 per block
        if (idx4 >= 960u) break;
        const unsigned j = idx4 / 240u, rem = idx4 % 240u;
        if (b0 + j >= n) continue;
        const unsigned plane = rem >> 4, sq4 = (rem & 15u) * 4u;
        const unsigned c4 = *reinterpret_cast<const unsigned*>(&codes[j][sq4]);
        f32x4 v;
        v.x = ((c4 & 0xFFu) == plane) ? 1.0f : 0.0f;
        v.y = (((c4 >> 8) & 0xFFu) == plane) ? 1.0f : 0.0f;
        v.z = (((c4 >> 16) & 0xFFu) == plane) ? 1.0f : 0.0f;
        v.w = ((c4 >> 24) == plane) ? 1.0f : 0.0f;
        __builtin_nontemporal_store(v, &out[(unsigned long long)(b0 + j) * 240u + rem]);
    }
}

inline unsigned blocks_for(size_t n, unsigned per_block) { return (unsigned)((n + per_block - 1) / per_block); }

hipError_t scan_offsets(const uint64_t* d_occ, size_t n, size_t n_bytes, unsigned start_nibble, const Workspace& ws,
                        hipStream_t stream) {
    const unsigned nb = blocks_for(n, kScanBoardsPerBlock);
    hipLaunchKernelGGL(k_popcount_scan, dim3(nb), dim3(256), 0, stream, d_occ, (unsigned)n, ws.local_off, ws.block_total);
    hipLaunchKernelGGL(k_scan_block_totals, dim3(1), dim3(1024), 0, stream, ws.block_total, nb,
                       (unsigned long long)n_bytes, start_nibble, ws.err_flag);
    return hipGetLastError();
}

}  // namespace

size_t scan_blocks_for(size_t n) { return (n + kScanBoardsPerBlock - 1) / kScanBoardsPerBlock; }

hipError_t launch_mailbox_from_wire(const uint8_t* d_nib, size_t n_bytes, const uint64_t* d_occ, size_t n,
                                    const Workspace& ws, uint8_t* d_mailbox, hipStream_t stream, unsigned start_nibble) {
    if (n == 0) return hipSuccess;
    hipError_t e = scan_offsets(d_occ, n, n_bytes, start_nibble, ws, stream);
    if (e != hipSuccess) return e;
    WireSrc src{d_nib, (unsigned long long)n_bytes, d_occ, ws.local_off, ws.block_total, ws.err_flag, start_nibble};
    hipLaunchKernelGGL(k_to_mailbox<WireSrc>, dim3(blocks_for(n, 4)), dim3(256), 0, stream, src, (unsigned)n, d_mailbox);
    return hipGetLastError();
}

// The same with every board's nibble offset already known (the host entry points add up the popcounts anyway to
// check the stream's length): no scan kernels.  d_offsets[b] = codes of the boards before b; d_zero_bases = zeros,
// one per kScanBoardsPerBlock boards (the per-block bases of the device scan, which the offsets here include).
hipError_t launch_mailbox_from_wire_offsets(const uint8_t* d_nib, size_t n_bytes, const uint64_t* d_occ, size_t n,
                                            const unsigned* d_offsets, const unsigned* d_zero_bases, int* d_err,
                                            uint8_t* d_mailbox, hipStream_t stream, unsigned start_nibble) {
    if (n == 0) return hipSuccess;
    WireSrc src{d_nib, (unsigned long long)n_bytes, d_occ, d_offsets, d_zero_bases, d_err, start_nibble};
    hipLaunchKernelGGL(k_to_mailbox<WireSrc>, dim3(blocks_for(n, 4)), dim3(256), 0, stream, src, (unsigned)n, d_mailbox);
    return hipGetLastError();
}

hipError_t launch_mailbox_from_structs(const void* d_structs, size_t stride, size_t n, uint8_t* d_mailbox,
                                       hipStream_t stream) {
    if (n == 0) return hipSuccess;
    StructSrc src{static_cast<const unsigned char*>(d_structs), (unsigned long long)stride};
    hipLaunchKernelGGL(k_to_mailbox<StructSrc>, dim3(blocks_for(n, 4)), dim3(256), 0, stream, src, (unsigned)n, d_mailbox);
    return hipGetLastError();
}

hipError_t launch_planes_from_wire(const uint8_t* d_nib, size_t n_bytes, const uint64_t* d_occ, size_t n,
                                   const Workspace& ws, float* d_planes, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipError_t e = scan_offsets(d_occ, n, n_bytes, 0u, ws, stream);
    if (e != hipSuccess) return e;
    WireSrc src{d_nib, (unsigned long long)n_bytes, d_occ, ws.local_off, ws.block_total, ws.err_flag, 0u};
    hipLaunchKernelGGL(k_to_planes<WireSrc>, dim3(blocks_for(n, 4)), dim3(256), 0, stream, src, (unsigned)n, d_planes);
    return hipGetLastError();
}

hipError_t launch_planes_from_structs(const void* d_structs, size_t stride, size_t n, float* d_planes,
                                      hipStream_t stream) {
    if (n == 0) return hipSuccess;
    StructSrc src{static_cast<const unsigned char*>(d_structs), (unsigned long long)stride};
    hipLaunchKernelGGL(k_to_planes<StructSrc>, dim3(blocks_for(n, 4)), dim3(256), 0, stream, src, (unsigned)n, d_planes);
    return hipGetLastError();
}

hipError_t launch_planes_from_mailbox(const uint8_t* d_mailbox, size_t n, float* d_planes, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    MailboxSrc src{d_mailbox};
    hipLaunchKernelGGL(k_to_planes<MailboxSrc>, dim3(blocks_for(n, 4)), dim3(256), 0, stream, src, (unsigned)n, d_planes);
    return hipGetLastError();
}

}  // namespace bf
