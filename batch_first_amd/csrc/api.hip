// C-ABI entry points (include/batchfirst.h): context, device and host evaluation calls.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <functional>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/batchfirst.h"
#include "bf_internal.h"
#include "bf_encode.h"
#include "bf_kernels.h"
#include "bf_layout.h"
#include "bf_layout_t16.h"
#include "bf_split.h"

#define BF_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return bf::fail(BF_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// Scratch, staging and stream of one evaluation pipeline.  The value network and the move-scoring
// network each own one, so that the search's two worker threads (numba_negamax_zero_window.py:178,
// :199) can have one call of each in flight at the same time.
struct Lane {
    int* d_err = nullptr;            // sticky malformed-input flag
    int* h_flag = nullptr;           // pinned landing place of its copy (a pageable one would cost a staged copy per call)
    // scratch grown on demand; guarded by `mu` (calls on one lane are serialised)
    std::mutex mu;
    size_t cap_boards = 0;
    unsigned* d_local_off = nullptr;
    unsigned* d_block_total = nullptr;
    uint8_t* d_mailbox = nullptr;
    // Kernels of different calls share this scratch, and calls may come on different streams (the
    // caller's for the device entry points, `stream` below for the host ones): the last kernel that
    // touches it records `scratch_done` on its stream, and the next call makes its stream wait for
    // that event before it enqueues anything.  Both happen under `mu`, so the order is total.
    hipEvent_t scratch_done = nullptr;
    bool scratch_used = false;
    // host entry points: own stream, pinned staging, device staging
    hipStream_t stream = nullptr;
    size_t cap_in_bytes = 0;
    unsigned char* h_in = nullptr;   // pinned
    unsigned char* d_in = nullptr;
    size_t cap_scores = 0;
    float* h_scores = nullptr;       // pinned
    float* d_scores = nullptr;
    // Small calls skip the copy engine: the kernels read the pinned staging and write the pinned scores over
    // the bus themselves (a copy costs 5-8 us of queueing each way, more than moving a few KB): the device's
    // addresses of h_in / h_scores.  The search's batches at depth 4 are a hundred boards (profiles/r2/search).
    unsigned char* h_in_dev = nullptr;
    float* h_scores_dev = nullptr;
};

struct bf_ctx : Lane {               // the base is the value network's lane
    int device = 0;
    uint8_t* d_static_mailbox = nullptr;  // this device's copy of `scratch` (bf_eval_structs_noalloc)
    int precision = BF_PRECISION_F32;
    int cu_count = 0;                     // compute units of the device (k_eval_f32q's limit: one board a unit)
    // BF_T16_TWO_KERNELS=1 at creation: records always go through k_to_mailbox first (A/B and the equality test);
    // otherwise the 16x16x32 kernel decodes 4-byte-aligned records itself (one launch a call)
    bool two_kernels = false;
    float* d_packed = nullptr;            // fp32 image (bf_layout.h) unless precision is BF_PRECISION_BF16
    unsigned char* d_image16 = nullptr;   // bf16 image (bf_layout_t16.h) when precision == BF_PRECISION_BF16
    // move-scoring network (bf_ctx_set_policy)
    Lane policy;
    float* d_policy_packed = nullptr;          // fp32 image, or ...
    unsigned char* d_policy_image16 = nullptr; // ... bf16 image, by the context's precision
    size_t cap_x9_boards = 0;
    float* d_x9 = nullptr;                // [boards][32][64] features of the batch in flight
    size_t policy_boards = 0;             // boards of the started, not yet finished, prediction
    bool policy_no_flags = false;         // the started prediction came from records (no device-side format checks)
    size_t policy_in_used = 0;            // bytes of the policy lane's staging that the started prediction's records occupy
    bool policy_started = false;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int ensure_boards(Lane* c, size_t n) {
    if (n <= c->cap_boards) return BF_OK;
    size_t cap = c->cap_boards ? c->cap_boards : 4096;
    while (cap < n) cap *= 2;
    if (c->d_local_off) (void)hipFree(c->d_local_off);
    if (c->d_block_total) (void)hipFree(c->d_block_total);
    if (c->d_mailbox) (void)hipFree(c->d_mailbox);
    c->d_local_off = nullptr; c->d_block_total = nullptr; c->d_mailbox = nullptr; c->cap_boards = 0;
    BF_HIP(hipMalloc(&c->d_local_off, cap * sizeof(unsigned)));
    BF_HIP(hipMalloc(&c->d_block_total, bf::scan_blocks_for(cap) * sizeof(unsigned) + 16));
    BF_HIP(hipMalloc(&c->d_mailbox, cap * bf::kMailboxBytes));
    c->cap_boards = cap;
    return BF_OK;
}

int ensure_staging(Lane* c, size_t in_bytes, size_t n_scores) {
    if (in_bytes > c->cap_in_bytes) {
        size_t cap = c->cap_in_bytes ? c->cap_in_bytes : (1u << 20);
        while (cap < in_bytes) cap *= 2;
        if (c->h_in) (void)hipHostFree(c->h_in);
        if (c->d_in) (void)hipFree(c->d_in);
        c->h_in = nullptr; c->d_in = nullptr; c->cap_in_bytes = 0;
        BF_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_in), cap, hipHostMallocDefault));
        BF_HIP(hipMalloc(&c->d_in, cap));
        c->h_in_dev = nullptr;
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_in_dev), c->h_in, 0) != hipSuccess) {
            (void)hipGetLastError();
            c->h_in_dev = nullptr;   // not mapped: every call goes through the copies
        }
        c->cap_in_bytes = cap;
    }
    if (n_scores > c->cap_scores) {
        size_t cap = c->cap_scores ? c->cap_scores : 4096;
        while (cap < n_scores) cap *= 2;
        if (c->h_scores) (void)hipHostFree(c->h_scores);
        if (c->d_scores) (void)hipFree(c->d_scores);
        c->h_scores = nullptr; c->d_scores = nullptr; c->cap_scores = 0;
        BF_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_scores), cap * sizeof(float), hipHostMallocDefault));
        BF_HIP(hipMalloc(&c->d_scores, cap * sizeof(float)));
        c->h_scores_dev = nullptr;
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&c->h_scores_dev), c->h_scores, 0) != hipSuccess) {
            (void)hipGetLastError();
            c->h_scores_dev = nullptr;
        }
        c->cap_scores = cap;
    }
    return BF_OK;
}

// bf16 calls of at most this many boards run the one-board-per-workgroup kernel (eval_t16s.hip): bit-identical
// scores, a third of the time while the boards are fewer than the workgroups the chip holds twice over
// (bf_set_split_boards; BF_SPLIT_BOARDS in the environment sets the start value, 0 turns it off).
std::atomic<size_t> g_split_boards{[] {
    const char* e = std::getenv("BF_SPLIT_BOARDS");
    return e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)BF_SPLIT_BOARDS_DEFAULT;
}()};
// 0 = float32, 1 = BF_PRECISION_F32_SPLIT, 2 = BF_PRECISION_F32_SPLIT3: what pack_weights and the float32 kernel's launchers take
int split_level(int precision) { return precision == BF_PRECISION_F32_SPLIT ? 1 : precision == BF_PRECISION_F32_SPLIT3 ? 2 : 0; }
bool use_split(const bf_ctx* c, size_t n) {
    return c->precision == BF_PRECISION_BF16 && n <= g_split_boards.load(std::memory_order_relaxed);
}

// A float32 (or split-precision) context runs calls of up to this many boards on k_eval_f32q (one board per workgroup, eight waves; the same
// bits as k_eval_f32: eval_f32q.hip).  Its 132 KB of LDS allow one workgroup per compute unit, so beyond one board
// per unit it runs in rounds and k_eval_f32 is as fast: the limit is the device's unit count, at most
// BF_F32Q_BOARDS (environment, default 256; 0 turns the kernel off).
size_t f32q_boards(const bf_ctx* c) {
    static const size_t cap = [] {
        const char* e = std::getenv("BF_F32Q_BOARDS");
        return e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)256;
    }();
    return std::min(cap, (size_t)c->cu_count);
}
bool use_f32q(const bf_ctx* c, size_t n) { return c->precision != BF_PRECISION_BF16 && n <= f32q_boards(c); }

// The value network in the context's precision.
hipError_t launch_eval(bf_ctx* c, const uint8_t* d_mailbox, size_t n, float* d_scores, float* d_debug, hipStream_t s) {
    if (!d_debug && use_split(c, n))
        return bf::launch_eval_t16_split(d_mailbox, bf::kMailboxBytes, false, n, c->d_image16, d_scores, s);
    if (!d_debug && use_f32q(c, n)) return bf::launch_eval_f32q(d_mailbox, n, c->d_packed, d_scores, s, split_level(c->precision));
    if (c->precision == BF_PRECISION_BF16)
        return bf::launch_eval_t16(d_mailbox, n, c->d_image16, d_scores, d_debug, s);
    return bf::launch_eval_f32(d_mailbox, n, c->d_packed, d_scores, d_debug, s, split_level(c->precision));
}

bf::Workspace workspace(Lane* c) { return bf::Workspace{c->d_local_off, c->d_block_total, c->d_err}; }

// Call with the lane's mutex held, before enqueuing work that reads or writes the lane's scratch on
// stream `s` / after the last such kernel has been enqueued (see Lane::scratch_done).
int scratch_acquire(Lane* c, hipStream_t s) {
    if (c->scratch_used) BF_HIP(hipStreamWaitEvent(s, c->scratch_done, 0));
    return BF_OK;
}
int scratch_release(Lane* c, hipStream_t s) {
    BF_HIP(hipEventRecord(c->scratch_done, s));
    c->scratch_used = true;
    return BF_OK;
}

// calls of at most this many boards / moves use the direct path (Lane::h_in_dev); BF_DIRECT_BOARDS=0 turns it off
size_t direct_boards() {
    static const size_t v = [] {
        const char* e = std::getenv("BF_DIRECT_BOARDS");
        return e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)2048;
    }();
    return v;
}
constexpr size_t kDirectMovesPerBoard = 64;
constexpr size_t kHostOffsetBoards = 4096;   // wire-format host calls up to this size skip the device scan (see bf_eval_wire_host)

// Wire format into a lane's pinned staging with every board's nibble offset beside it (the caller has checked that
// the stream's length matches the popcounts): occupied u64[n] | offsets u32[n] | zero bases u32[blocks] | nibble bytes.
struct StagedWire {
    size_t off_offsets = 0, off_bases = 0, off_pf = 0, bytes = 0;
};
StagedWire wire_layout(size_t n, size_t n_bytes) {
    StagedWire w;
    w.off_offsets = n * sizeof(uint64_t);
    w.off_bases = w.off_offsets + n * sizeof(uint32_t);
    w.off_pf = (w.off_bases + bf::scan_blocks_for(n) * sizeof(uint32_t) + 15) & ~(size_t)15;
    w.bytes = w.off_pf + n_bytes;
    return w;
}
void stage_wire(unsigned char* h_in, const StagedWire& w, const uint8_t* h_pf, size_t n_bytes, const uint64_t* h_occ, size_t n,
                unsigned start_nibble) {
    std::memcpy(h_in, h_occ, n * sizeof(uint64_t));
    uint32_t* off = reinterpret_cast<uint32_t*>(h_in + w.off_offsets);
    uint32_t run = 0;
    for (size_t i = 0; i < n; ++i) {
        off[i] = run;
        run += (uint32_t)__builtin_popcountll(h_occ[i]);
    }
    (void)start_nibble;   // the kernel adds it
    std::memset(h_in + w.off_bases, 0, w.off_pf - w.off_bases);
    if (n_bytes) std::memcpy(h_in + w.off_pf, h_pf, n_bytes);
}

// Wire format -> mailboxes of lane L on its own stream, staged through its pinned buffer (caller holds L->mu, has
// checked the stream's length against the popcounts and called ensure_boards + scratch_acquire).
// Small calls: the boards' nibble offsets are added up here, on the host, so that the two scan kernels are not
// needed, and (may_direct) the kernel reads the pinned staging itself: launches and copies are what a small call
// costs.  Large calls keep the device scan, which beats one host thread from a few thousand boards on.
// *direct: the staging was not copied to the device; *staged: bytes of staging in use.
// fuse (nullable): the caller runs a kernel that decodes the wire format itself (eval_t16s.hip): for the small-call
// branch no mailbox kernel is launched, *fuse receives the staging layout and *fuse_in the device's view of it.
int wire_to_mailbox(Lane* L, const uint8_t* h_pf, size_t n_bytes, const uint64_t* h_occ, size_t n, unsigned start_nibble,
                    size_t n_scores, bool may_direct, bool* direct, size_t* staged, StagedWire* fuse = nullptr,
                    const unsigned char** fuse_in = nullptr) {
    if (fuse_in) *fuse_in = nullptr;
    const bool host_offsets = n <= kHostOffsetBoards;
    *direct = host_offsets && may_direct && n <= direct_boards() && L->h_in_dev != nullptr;
    if (host_offsets) {
        const StagedWire w = wire_layout(n, n_bytes);
        if (int r = ensure_staging(L, w.bytes + 16, n_scores)) return r;
        *direct = *direct && L->h_in_dev != nullptr;   // (the buffer may just have been regrown)
        stage_wire(L->h_in, w, h_pf, n_bytes, h_occ, n, start_nibble);
        const unsigned char* in = *direct ? L->h_in_dev : L->d_in;
        if (!*direct) BF_HIP(hipMemcpyAsync(L->d_in, L->h_in, w.bytes, hipMemcpyHostToDevice, L->stream));
        if (fuse && fuse_in) {
            *fuse = w;
            *fuse_in = in;
            *staged = w.bytes;
            return BF_OK;
        }
        BF_HIP(bf::launch_mailbox_from_wire_offsets(in + w.off_pf, n_bytes, reinterpret_cast<const uint64_t*>(in), n,
                                                    reinterpret_cast<const unsigned*>(in + w.off_offsets),
                                                    reinterpret_cast<const unsigned*>(in + w.off_bases), L->d_err,
                                                    L->d_mailbox, L->stream, start_nibble));
        *staged = w.bytes;
    } else {
        const size_t occ_bytes = n * sizeof(uint64_t);
        const size_t pf_off = (occ_bytes + 15) & ~(size_t)15;
        if (int r = ensure_staging(L, pf_off + n_bytes + 16, n_scores)) return r;
        std::memcpy(L->h_in, h_occ, occ_bytes);
        if (n_bytes) std::memcpy(L->h_in + pf_off, h_pf, n_bytes);
        BF_HIP(hipMemcpyAsync(L->d_in, L->h_in, pf_off + n_bytes, hipMemcpyHostToDevice, L->stream));
        BF_HIP(bf::launch_mailbox_from_wire(L->d_in + pf_off, n_bytes, reinterpret_cast<const uint64_t*>(L->d_in), n,
                                            workspace(L), L->d_mailbox, L->stream, start_nibble));
        *staged = pf_off + n_bytes;
    }
    return BF_OK;
}

// Records -> scores (value network) on stream `s`.  `L` = the lane whose mailbox scratch the two-kernel form uses.
int eval_from_structs(bf_ctx* c, const void* d_structs, size_t stride, size_t n, float* d_scores, hipStream_t s) {
    if (c->precision == BF_PRECISION_BF16 && !c->two_kernels && bf::t16_takes_records(d_structs, stride)) {
        if (use_split(c, n))
            BF_HIP(bf::launch_eval_t16_split(static_cast<const uint8_t*>(d_structs), stride, true, n, c->d_image16, d_scores, s));
        else
            BF_HIP(bf::launch_eval_t16_structs(d_structs, stride, n, c->d_image16, d_scores, s));
        return BF_OK;   // no scratch involved
    }
    if (int r = ensure_boards(c, n)) return r;
    if (int r = scratch_acquire(c, s)) return r;
    BF_HIP(bf::launch_mailbox_from_structs(d_structs, stride, n, c->d_mailbox, s));
    BF_HIP(launch_eval(c, c->d_mailbox, n, d_scores, nullptr, s));
    return scratch_release(c, s);
}

// Call with the lane's mutex held: the landing slot h_flag (allocated with the lane, in bf_ctx_create_ex) is the lane's.
int read_err_flag(Lane* c, hipStream_t s) {
    *c->h_flag = 0;
    BF_HIP(hipMemcpyAsync(c->h_flag, c->d_err, sizeof(int), hipMemcpyDeviceToHost, s));
    BF_HIP(hipStreamSynchronize(s));
    const int flag = *c->h_flag;
    if (flag) {
        BF_HIP(hipMemsetAsync(c->d_err, 0, sizeof(int), s));
        BF_HIP(hipStreamSynchronize(s));
        if (flag & 4) return bf::fail(BF_E_FORMAT, "move input: board index, from-square or move filter out of range");
        if (flag & 2) return bf::fail(BF_E_FORMAT, "wire input holds code 15, which is not a piece code");
        return bf::fail(BF_E_FORMAT, "wire input: piece_filters length does not match the popcounts of occupied_bbs");
    }
    return BF_OK;
}

constexpr size_t kMaxBoards = (size_t)1 << 26;
constexpr size_t kPolicyMoveRoom = 64;   // moves a board that bf_policy_start_structs_host reserves staging for

// ---- scratch of the allocation-free entry point (bf_eval_structs_noalloc) -----------------------
// The mailboxes of one call, uint8 [BF_NOALLOC_MAX_BOARDS][32] = 8 MiB, in a file-scope device array:
// every GPU the library runs on holds its own copy, at its own address, until the process ends.  In
// host code the array's NAME is only a handle for hipGetSymbolAddress (passing it to a kernel would
// compile and fault); the per-device address is looked up once, with that device current, when the
// first context on the device is created, and only that pointer is used afterwards.
static __device__ __align__(256) unsigned char scratch[BF_NOALLOC_MAX_BOARDS * 32];
constexpr int kMaxDevices = 64;
unsigned char* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mu;

// `device` must be the current device.
int static_scratch_address(int device, uint8_t** out) {
    if (device < 0 || device >= kMaxDevices) return bf::fail(BF_E_ARG, "device index %d is beyond the %d this library keeps scratch for", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return bf::fail(BF_E_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", device, hipGetErrorString(e));
        g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    }
    *out = g_scratch_ptr[device];
    return BF_OK;
}

int check_common(const bf_ctx* c, size_t n, const char* who) {
    if (!c) return bf::fail(BF_E_ARG, "%s: null context", who);
    if (n > kMaxBoards) return bf::fail(BF_E_ARG, "%s: %zu boards exceeds the limit of %zu per call", who, n, kMaxBoards);
    return BF_OK;
}

}  // namespace

extern "C" const char* bf_version(void) { return "batchfirst-hip 0.1 gfx950"; }

extern "C" int bf_ctx_create(bf_ctx** out, int device, const float* h_weights, size_t n_floats) {
    return bf_ctx_create_ex(out, device, h_weights, n_floats, BF_PRECISION_F32);
}

extern "C" size_t bf_set_split_boards(size_t n) { return g_split_boards.exchange(n); }

extern "C" int bf_ctx_precision(const bf_ctx* c) { return c ? c->precision : BF_E_ARG; }

extern "C" int bf_ctx_create_ex(bf_ctx** out, int device, const float* h_weights, size_t n_floats, int precision) {
    if (!out || !h_weights) return bf::fail(BF_E_ARG, "bf_ctx_create: null pointer");
    *out = nullptr;
    if (precision != BF_PRECISION_F32 && precision != BF_PRECISION_BF16 && precision != BF_PRECISION_F32_SPLIT &&
        precision != BF_PRECISION_F32_SPLIT3)
        return bf::fail(BF_E_ARG, "bf_ctx_create: unknown precision %d", precision);
    if (n_floats != bf::kWeightFloats)
        return bf::fail(BF_E_ARG, "bf_ctx_create: expected %zu weight floats, got %zu", bf::kWeightFloats, n_floats);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return bf::fail(BF_E_NOGPU, "bf_ctx_create: no HIP device");
    if (device < 0 || device >= count) return bf::fail(BF_E_ARG, "bf_ctx_create: device %d of %d", device, count);
    hipDeviceProp_t prop;
    BF_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return bf::fail(BF_E_NOGPU, "bf_ctx_create: device %d is %s, this library is built for gfx950 only", device,
                        prop.gcnArchName);
    DeviceGuard guard(device);
    if (!guard.ok) return bf::fail(BF_E_HIP, "bf_ctx_create: hipSetDevice(%d) failed", device);

    bf_ctx* c = new (std::nothrow) bf_ctx;
    if (!c) return bf::fail(BF_E_ARG, "bf_ctx_create: out of host memory");
    c->device = device;
    c->precision = precision;
    if (hipDeviceGetAttribute(&c->cu_count, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) c->cu_count = 0;
    auto bail = [&](int code) {
        bf_ctx_destroy(c);
        return code;
    };
    if (precision == BF_PRECISION_BF16) {
        const char* two = std::getenv("BF_T16_TWO_KERNELS");
        c->two_kernels = two && std::strcmp(two, "1") == 0;

        std::vector<unsigned char> image(bf::t16::kImageBytes);
        bf::pack_weights_t16(h_weights, image.data());
        if (hipMalloc(&c->d_image16, image.size()) != hipSuccess)
            return bail(bf::fail(BF_E_HIP, "bf_ctx_create: hipMalloc of the weight image failed"));
        if (hipMemcpy(c->d_image16, image.data(), image.size(), hipMemcpyHostToDevice) != hipSuccess)
            return bail(bf::fail(BF_E_HIP, "bf_ctx_create: weight upload failed"));
    } else {
        std::vector<float> packed(bf::kPackedFloats);
        bf::pack_weights(h_weights, packed.data(), false, split_level(precision));
        if (hipMalloc(&c->d_packed, bf::kPackedFloats * sizeof(float)) != hipSuccess)
            return bail(bf::fail(BF_E_HIP, "bf_ctx_create: hipMalloc of the weight image failed"));
        if (hipMemcpy(c->d_packed, packed.data(), bf::kPackedFloats * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return bail(bf::fail(BF_E_HIP, "bf_ctx_create: weight upload failed"));
    }
    for (Lane* lane : {static_cast<Lane*>(c), &c->policy}) {
        if (hipMalloc(&lane->d_err, 256) != hipSuccess || hipMemset(lane->d_err, 0, 256) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&lane->h_flag), 64, hipHostMallocDefault) != hipSuccess)
            return bail(bf::fail(BF_E_HIP, "bf_ctx_create: flag allocation failed"));
        if (hipStreamCreateWithFlags(&lane->stream, hipStreamNonBlocking) != hipSuccess)
            return bail(bf::fail(BF_E_HIP, "bf_ctx_create: stream creation failed"));
        if (hipEventCreateWithFlags(&lane->scratch_done, hipEventDisableTiming) != hipSuccess)
            return bail(bf::fail(BF_E_HIP, "bf_ctx_create: event creation failed"));
    }
    if (int r = static_scratch_address(device, &c->d_static_mailbox)) return bail(r);
    *out = c;
    return BF_OK;
}

extern "C" void bf_ctx_destroy(bf_ctx* c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    for (Lane* lane : {static_cast<Lane*>(c), &c->policy}) {
        if (lane->stream) {
            (void)hipStreamSynchronize(lane->stream);
            (void)hipStreamDestroy(lane->stream);
        }
        if (lane->scratch_done) (void)hipEventDestroy(lane->scratch_done);
        if (lane->d_err) (void)hipFree(lane->d_err);
        if (lane->h_flag) (void)hipHostFree(lane->h_flag);
        if (lane->d_local_off) (void)hipFree(lane->d_local_off);
        if (lane->d_block_total) (void)hipFree(lane->d_block_total);
        if (lane->d_mailbox) (void)hipFree(lane->d_mailbox);
        if (lane->h_in) (void)hipHostFree(lane->h_in);
        if (lane->d_in) (void)hipFree(lane->d_in);
        if (lane->h_scores) (void)hipHostFree(lane->h_scores);
        if (lane->d_scores) (void)hipFree(lane->d_scores);
    }
    if (c->d_packed) (void)hipFree(c->d_packed);
    if (c->d_image16) (void)hipFree(c->d_image16);
    if (c->d_policy_packed) (void)hipFree(c->d_policy_packed);
    if (c->d_policy_image16) (void)hipFree(c->d_policy_image16);
    if (c->d_x9) (void)hipFree(c->d_x9);
    delete c;
}

// ---- device-pointer entry points -------------------------------------------------------

extern "C" int bf_encode_mailbox_wire(bf_ctx* c, const uint8_t* d_pf, size_t n_bytes, const uint64_t* d_occ, size_t n,
                                      uint8_t* d_mailbox, void* stream) {
    if (int r = check_common(c, n, "bf_encode_mailbox_wire")) return r;
    if (n == 0) return BF_OK;
    if (!d_occ || !d_mailbox || (!d_pf && n_bytes)) return bf::fail(BF_E_ARG, "bf_encode_mailbox_wire: null pointer");
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lock(c->mu);
    if (int r = ensure_boards(c, n)) return r;
    if (int r = scratch_acquire(c, (hipStream_t)stream)) return r;
    BF_HIP(bf::launch_mailbox_from_wire(d_pf, n_bytes, d_occ, n, workspace(c), d_mailbox, (hipStream_t)stream));
    return scratch_release(c, (hipStream_t)stream);
}

extern "C" int bf_encode_mailbox_structs(bf_ctx* c, const void* d_structs, size_t stride, size_t n,
                                         uint8_t* d_mailbox, void* stream) {
    if (int r = check_common(c, n, "bf_encode_mailbox_structs")) return r;
    if (n == 0) return BF_OK;
    if (!d_structs || !d_mailbox) return bf::fail(BF_E_ARG, "bf_encode_mailbox_structs: null pointer");
    if (stride < BF_NODE_EVAL_BYTES) return bf::fail(BF_E_ARG, "bf_encode_mailbox_structs: stride %zu < 82", stride);
    DeviceGuard guard(c->device);
    BF_HIP(bf::launch_mailbox_from_structs(d_structs, stride, n, d_mailbox, (hipStream_t)stream));
    return BF_OK;
}

extern "C" int bf_encode_planes_wire(bf_ctx* c, const uint8_t* d_pf, size_t n_bytes, const uint64_t* d_occ, size_t n,
                                     float* d_planes, void* stream) {
    if (int r = check_common(c, n, "bf_encode_planes_wire")) return r;
    if (n == 0) return BF_OK;
    if (!d_occ || !d_planes || (!d_pf && n_bytes)) return bf::fail(BF_E_ARG, "bf_encode_planes_wire: null pointer");
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lock(c->mu);
    if (int r = ensure_boards(c, n)) return r;
    if (int r = scratch_acquire(c, (hipStream_t)stream)) return r;
    BF_HIP(bf::launch_planes_from_wire(d_pf, n_bytes, d_occ, n, workspace(c), d_planes, (hipStream_t)stream));
    return scratch_release(c, (hipStream_t)stream);
}

extern "C" int bf_encode_planes_structs(bf_ctx* c, const void* d_structs, size_t stride, size_t n, float* d_planes,
                                        void* stream) {
    if (int r = check_common(c, n, "bf_encode_planes_structs")) return r;
    if (n == 0) return BF_OK;
    if (!d_structs || !d_planes) return bf::fail(BF_E_ARG, "bf_encode_planes_structs: null pointer");
    if (stride < BF_NODE_EVAL_BYTES) return bf::fail(BF_E_ARG, "bf_encode_planes_structs: stride %zu < 82", stride);
    DeviceGuard guard(c->device);
    BF_HIP(bf::launch_planes_from_structs(d_structs, stride, n, d_planes, (hipStream_t)stream));
    return BF_OK;
}

extern "C" int bf_eval_mailbox(bf_ctx* c, const uint8_t* d_mailbox, size_t n, float* d_scores, void* stream) {
    if (int r = check_common(c, n, "bf_eval_mailbox")) return r;
    if (n == 0) return BF_OK;
    if (!d_mailbox || !d_scores) return bf::fail(BF_E_ARG, "bf_eval_mailbox: null pointer");
    DeviceGuard guard(c->device);
    BF_HIP(launch_eval(c, d_mailbox, n, d_scores, nullptr, (hipStream_t)stream));
    return BF_OK;
}

extern "C" int bf_eval_wire(bf_ctx* c, const uint8_t* d_pf, size_t n_bytes, const uint64_t* d_occ, size_t n,
                            float* d_scores, void* stream) {
    if (int r = check_common(c, n, "bf_eval_wire")) return r;
    if (n == 0) return BF_OK;
    if (!d_occ || !d_scores || (!d_pf && n_bytes)) return bf::fail(BF_E_ARG, "bf_eval_wire: null pointer");
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lock(c->mu);
    if (int r = ensure_boards(c, n)) return r;
    if (int r = scratch_acquire(c, (hipStream_t)stream)) return r;
    BF_HIP(bf::launch_mailbox_from_wire(d_pf, n_bytes, d_occ, n, workspace(c), c->d_mailbox, (hipStream_t)stream));
    BF_HIP(launch_eval(c, c->d_mailbox, n, d_scores, nullptr, (hipStream_t)stream));
    return scratch_release(c, (hipStream_t)stream);
}

extern "C" int bf_eval_structs(bf_ctx* c, const void* d_structs, size_t stride, size_t n, float* d_scores,
                               void* stream) {
    if (int r = check_common(c, n, "bf_eval_structs")) return r;
    if (n == 0) return BF_OK;
    if (!d_structs || !d_scores) return bf::fail(BF_E_ARG, "bf_eval_structs: null pointer");
    if (stride < BF_NODE_EVAL_BYTES) return bf::fail(BF_E_ARG, "bf_eval_structs: stride %zu < 82", stride);
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lock(c->mu);
    return eval_from_structs(c, d_structs, stride, n, d_scores, (hipStream_t)stream);
}

// Variant of bf_eval_structs for callers that cannot allocate: the mailboxes between the two kernels
// live in this device's copy of `scratch` (see above), nothing is allocated, grown or freed, no lock is
// taken and nothing synchronises.  Every mailbox the evaluation reads ([0, n)) is written by the
// encode kernel of the same call first, so the array needs no clearing between calls; calls that
// overlap on one GPU (two streams, two contexts, two threads) share it and race.
extern "C" int bf_eval_structs_noalloc(bf_ctx* c, const void* d_structs, size_t stride, size_t n, float* d_scores,
                                       void* stream) {
    if (!c) return bf::fail(BF_E_ARG, "bf_eval_structs_noalloc: null context");
    if (n > BF_NOALLOC_MAX_BOARDS)
        return bf::fail(BF_E_ARG, "bf_eval_structs_noalloc: %zu boards exceeds the static scratch of %u boards per call "
                        "(split the batch, or use bf_eval_structs)", n, (unsigned)BF_NOALLOC_MAX_BOARDS);
    if (n == 0) return BF_OK;
    if (!d_structs || !d_scores) return bf::fail(BF_E_ARG, "bf_eval_structs_noalloc: null pointer");
    if (stride < BF_NODE_EVAL_BYTES) return bf::fail(BF_E_ARG, "bf_eval_structs_noalloc: stride %zu < 82", stride);
    DeviceGuard guard(c->device);
    uint8_t* const mailbox = c->d_static_mailbox;   // the looked-up device address, never the array's name
    BF_HIP(bf::launch_mailbox_from_structs(d_structs, stride, n, mailbox, (hipStream_t)stream));
    BF_HIP(launch_eval(c, mailbox, n, d_scores, nullptr, (hipStream_t)stream));
    return BF_OK;
}

extern "C" int bf_eval_mailbox_debug(bf_ctx* c, const uint8_t* d_mailbox, size_t n, float* d_scores, float* d_debug,
                                     void* stream) {
    if (int r = check_common(c, n, "bf_eval_mailbox_debug")) return r;
    if (n == 0) return BF_OK;
    if (!d_mailbox || !d_scores || !d_debug) return bf::fail(BF_E_ARG, "bf_eval_mailbox_debug: null pointer");
    DeviceGuard guard(c->device);
    BF_HIP(launch_eval(c, d_mailbox, n, d_scores, d_debug, (hipStream_t)stream));
    return BF_OK;
}

extern "C" size_t bf_debug_floats_per_board(void) { return bf::kDebugFloatsPerBoard; }

extern "C" int bf_ctx_check(bf_ctx* c, void* stream) {
    if (!c) return bf::fail(BF_E_ARG, "bf_ctx_check: null context");
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lock(c->mu);   // the flag's landing slot belongs to the lane
    return read_err_flag(c, (hipStream_t)stream);
}

// ---- host-pointer entry points -----------------------------------------------------------

extern "C" int bf_eval_wire_host(bf_ctx* c, const uint8_t* h_pf, size_t n_bytes, const uint64_t* h_occ, size_t n,
                                 float* h_scores) {
    if (int r = check_common(c, n, "bf_eval_wire_host")) return r;
    if (n == 0) return BF_OK;
    if (!h_occ || !h_scores || (!h_pf && n_bytes)) return bf::fail(BF_E_ARG, "bf_eval_wire_host: null pointer");
    // validate before touching the GPU: ann_creation_helper.py:41 allows at most 33 codes a board
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) total += (size_t)__builtin_popcountll(h_occ[i]);
    if ((total + 1) / 2 != n_bytes)
        return bf::fail(BF_E_FORMAT, "bf_eval_wire_host: %zu piece_filters bytes, but the masks select %zu codes (%zu bytes)",
                        n_bytes, total, (total + 1) / 2);
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lock(c->mu);
    if (int r = ensure_boards(c, n)) return r;
    if (int r = scratch_acquire(c, c->stream)) return r;
    bool direct = false;
    size_t staged = 0;
    StagedWire fw;
    const unsigned char* fin = nullptr;
    const bool want_fused = use_split(c, n) && n <= kHostOffsetBoards;
    if (int r = wire_to_mailbox(c, h_pf, n_bytes, h_occ, n, 0, n, c->h_scores_dev != nullptr, &direct, &staged,
                                want_fused ? &fw : nullptr, want_fused ? &fin : nullptr))
        return r;
    direct = direct && c->h_scores_dev != nullptr;
    if (fin)   // small bf16 call: the evaluating workgroups decode the wire format themselves, one launch
        BF_HIP(bf::launch_t16_split_wire(fin + fw.off_pf, n_bytes, reinterpret_cast<const uint64_t*>(fin),
                                         reinterpret_cast<const unsigned*>(fin + fw.off_offsets), n, 0u, c->d_err, c->d_image16,
                                         direct ? c->h_scores_dev : c->d_scores, false, c->stream));
    else
        BF_HIP(launch_eval(c, c->d_mailbox, n, direct ? c->h_scores_dev : c->d_scores, nullptr, c->stream));
    if (int r = scratch_release(c, c->stream)) return r;
    if (!direct) BF_HIP(hipMemcpyAsync(c->h_scores, c->d_scores, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (int r = read_err_flag(c, c->stream)) return r;
    std::memcpy(h_scores, c->h_scores, n * sizeof(float));
    return BF_OK;
}

extern "C" int bf_eval_structs_host(bf_ctx* c, const void* h_structs, size_t stride, const uint8_t* h_mask, size_t n,
                                    float* h_scores, size_t* n_scored) {
    if (n_scored) *n_scored = 0;
    if (int r = check_common(c, n, "bf_eval_structs_host")) return r;
    if (n == 0) return BF_OK;
    if (!h_structs || !h_scores) return bf::fail(BF_E_ARG, "bf_eval_structs_host: null pointer");
    if (stride < BF_NODE_EVAL_BYTES) return bf::fail(BF_E_ARG, "bf_eval_structs_host: stride %zu < 82", stride);
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lock(c->mu);
    // gather the 82 evaluation bytes of every selected record into an 88-byte-stride staging copy
    constexpr size_t kCompact = 88;
    size_t m = 0;
    if (h_mask) {
        for (size_t j = 0; j < n; ++j) m += h_mask[j] ? 1 : 0;
    } else {
        m = n;
    }
    if (m == 0) return BF_OK;
    if (int r = ensure_staging(c, m * kCompact, m)) return r;
    const unsigned char* recs = static_cast<const unsigned char*>(h_structs);
    size_t k = 0;
    for (size_t j = 0; j < n; ++j) {
        if (h_mask && !h_mask[j]) continue;
        std::memcpy(c->h_in + k * kCompact, recs + j * stride, BF_NODE_EVAL_BYTES);
        ++k;
    }
    const bool direct = m <= direct_boards() && c->h_in_dev && c->h_scores_dev;
    if (!direct) BF_HIP(hipMemcpyAsync(c->d_in, c->h_in, m * kCompact, hipMemcpyHostToDevice, c->stream));
    if (int r = eval_from_structs(c, direct ? c->h_in_dev : c->d_in, kCompact, m, direct ? c->h_scores_dev : c->d_scores, c->stream))
        return r;
    if (!direct) BF_HIP(hipMemcpyAsync(c->h_scores, c->d_scores, m * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    BF_HIP(hipStreamSynchronize(c->stream));
    std::memcpy(h_scores, c->h_scores, m * sizeof(float));
    if (n_scored) *n_scored = m;
    return BF_OK;
}


// ---- several GPUs behind one call (north_star: independent boards on each GPU on separate HIP streams,
// scores gathered back to the search host; SURVEY.md 8(e): one process driving the devices) ---------

struct bf_multi {
    std::vector<bf_ctx*> ctx;      // one context per device, borrowed (the caller destroys them)
    size_t min_shard = 1024;       // a device is used only if every device in use gets at least this many boards
    std::mutex mu;                 // one multi-device value call at a time (the contexts may still be used directly)
    std::vector<size_t> board_lo, nibble_lo;   // scratch of the call in flight
    // the move-scoring network over the same devices: the prediction in flight (bf_multi_policy_*)
    std::mutex policy_mu;
    std::vector<size_t> policy_lo; // first board of every shard of the started prediction (parts + 1 entries)
    bool policy_started = false;
};

extern "C" int bf_multi_create(bf_multi** out, bf_ctx* const* ctxs, size_t n_ctx, size_t min_shard) {
    if (!out || !ctxs || n_ctx == 0) return bf::fail(BF_E_ARG, "bf_multi_create: no contexts");
    *out = nullptr;
    for (size_t i = 0; i < n_ctx; ++i) {
        if (!ctxs[i]) return bf::fail(BF_E_ARG, "bf_multi_create: context %zu is null", i);
        if (ctxs[i]->precision != ctxs[0]->precision)
            return bf::fail(BF_E_ARG, "bf_multi_create: contexts differ in precision (scores would depend on the shard)");
        for (size_t j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return bf::fail(BF_E_ARG, "bf_multi_create: context %zu given twice", i);
    }
    bf_multi* m = new (std::nothrow) bf_multi;
    if (!m) return bf::fail(BF_E_ARG, "bf_multi_create: out of host memory");
    m->ctx.assign(ctxs, ctxs + n_ctx);
    m->min_shard = min_shard ? min_shard : 1;
    *out = m;
    return BF_OK;
}

extern "C" void bf_multi_destroy(bf_multi* m) { delete m; }

extern "C" size_t bf_multi_devices(const bf_multi* m) { return m ? m->ctx.size() : 0; }

extern "C" size_t bf_multi_set_min_shard(bf_multi* m, size_t min_shard) {
    if (!m) return 0;
    std::lock_guard<std::mutex> lock(m->mu);
    std::lock_guard<std::mutex> lock2(m->policy_mu);
    const size_t prev = m->min_shard;
    if (min_shard) m->min_shard = min_shard;
    return prev;
}

namespace {

size_t parts_for(const bf_multi* m, size_t n) {
    size_t parts = n / m->min_shard;
    if (parts < 1) parts = 1;
    return parts < m->ctx.size() ? parts : m->ctx.size();
}

// The mutexes of the first `parts` contexts' lanes (value lane, or the move-scoring lane), taken in ADDRESS order --
// two handles built over the same contexts in different orders cannot deadlock -- and released when this goes out
// of scope, an exception unwinding through the call included.
struct ShardLocks {
    std::vector<std::unique_lock<std::mutex>> held;
    ShardLocks(const bf_multi* m, size_t parts, bool policy_lane) {
        std::vector<bf_ctx*> order(m->ctx.begin(), m->ctx.begin() + parts);
        std::sort(order.begin(), order.end(), std::less<bf_ctx*>());
        held.reserve(parts);
        for (bf_ctx* c : order) held.emplace_back(policy_lane ? c->policy.mu : c->mu);
    }
};

// Enqueue one shard's host-to-device copy, kernels and device-to-host copy on its context's stream.
// The context's mutex is held by the caller until finish_shard().
int start_wire_shard(bf_ctx* c, const uint8_t* h_pf, size_t n_bytes, unsigned start_nibble, const uint64_t* h_occ, size_t n) {
    DeviceGuard guard(c->device);
    if (!guard.ok) return bf::fail(BF_E_HIP, "hipSetDevice(%d) failed", c->device);
    if (int r = ensure_boards(c, n)) return r;
    if (int r = scratch_acquire(c, c->stream)) return r;
    bool direct = false;
    size_t staged = 0;
    if (int r = wire_to_mailbox(c, h_pf, n_bytes, h_occ, n, start_nibble, n, false, &direct, &staged)) return r;
    BF_HIP(launch_eval(c, c->d_mailbox, n, c->d_scores, nullptr, c->stream));
    if (int r = scratch_release(c, c->stream)) return r;
    BF_HIP(hipMemcpyAsync(c->h_scores, c->d_scores, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    return BF_OK;
}

int start_struct_shard(bf_ctx* c, const unsigned char* compact, size_t m) {
    constexpr size_t kCompact = 88;
    DeviceGuard guard(c->device);
    if (!guard.ok) return bf::fail(BF_E_HIP, "hipSetDevice(%d) failed", c->device);
    if (int r = ensure_staging(c, m * kCompact, m)) return r;
    std::memcpy(c->h_in, compact, m * kCompact);
    BF_HIP(hipMemcpyAsync(c->d_in, c->h_in, m * kCompact, hipMemcpyHostToDevice, c->stream));
    if (int r = eval_from_structs(c, c->d_in, kCompact, m, c->d_scores, c->stream)) return r;
    BF_HIP(hipMemcpyAsync(c->h_scores, c->d_scores, m * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    return BF_OK;
}

int finish_shard(bf_ctx* c, size_t n, float* h_scores, bool wire) {
    DeviceGuard guard(c->device);
    if (wire) {
        if (int r = read_err_flag(c, c->stream)) return r;
    } else {
        BF_HIP(hipStreamSynchronize(c->stream));
    }
    std::memcpy(h_scores, c->h_scores, n * sizeof(float));
    return BF_OK;
}

}  // namespace

extern "C" int bf_multi_eval_wire_host(bf_multi* m, const uint8_t* h_pf, size_t n_bytes, const uint64_t* h_occ, size_t n,
                                       float* h_scores) {
    if (!m) return bf::fail(BF_E_ARG, "bf_multi_eval_wire_host: null handle");
    if (int r = check_common(m->ctx[0], n, "bf_multi_eval_wire_host")) return r;
    if (n == 0) return BF_OK;
    if (!h_occ || !h_scores || (!h_pf && n_bytes)) return bf::fail(BF_E_ARG, "bf_multi_eval_wire_host: null pointer");
    std::lock_guard<std::mutex> lock(m->mu);
    const size_t parts = parts_for(m, n);
    m->board_lo.assign(parts + 1, 0);
    m->nibble_lo.assign(parts + 1, 0);
    if (int r = bf_host_split_wire(h_occ, n, parts, m->board_lo.data(), m->nibble_lo.data())) return r;
    const size_t total = m->nibble_lo[parts];
    if ((total + 1) / 2 != n_bytes)
        return bf::fail(BF_E_FORMAT, "bf_multi_eval_wire_host: %zu piece_filters bytes, but the masks select %zu codes (%zu bytes)",
                        n_bytes, total, (total + 1) / 2);
    // all shards are enqueued before any is waited for: the devices run side by side
    ShardLocks locks(m, parts, false);
    int rc = BF_OK;
    size_t started = 0;
    for (; started < parts && rc == BF_OK; ++started) {
        bf_ctx* c = m->ctx[started];
        const size_t s_lo = m->nibble_lo[started], s_hi = m->nibble_lo[started + 1];
        const size_t byte_lo = s_lo >> 1, byte_hi = (s_hi + 1) >> 1;
        rc = start_wire_shard(c, h_pf + byte_lo, byte_hi - byte_lo, (unsigned)(s_lo & 1), h_occ + m->board_lo[started],
                              m->board_lo[started + 1] - m->board_lo[started]);
    }
    for (size_t p = 0; p < started; ++p) {
        bf_ctx* c = m->ctx[p];
        // a shard that failed to start may have enqueued part of its work: drain it either way
        const int r = finish_shard(c, rc == BF_OK ? m->board_lo[p + 1] - m->board_lo[p] : 0, h_scores + m->board_lo[p], true);
        if (rc == BF_OK) rc = r;
    }
    return rc;
}

extern "C" int bf_multi_eval_structs_host(bf_multi* m, const void* h_structs, size_t stride, const uint8_t* h_mask, size_t n,
                                          float* h_scores, size_t* n_scored) {
    if (n_scored) *n_scored = 0;
    if (!m) return bf::fail(BF_E_ARG, "bf_multi_eval_structs_host: null handle");
    if (int r = check_common(m->ctx[0], n, "bf_multi_eval_structs_host")) return r;
    if (n == 0) return BF_OK;
    if (!h_structs || !h_scores) return bf::fail(BF_E_ARG, "bf_multi_eval_structs_host: null pointer");
    if (stride < BF_NODE_EVAL_BYTES) return bf::fail(BF_E_ARG, "bf_multi_eval_structs_host: stride %zu < 82", stride);
    std::lock_guard<std::mutex> lock(m->mu);
    // gather the selected records once (88-byte stride), then hand contiguous runs to the devices
    constexpr size_t kCompact = 88;
    size_t count = 0;
    if (h_mask) {
        for (size_t j = 0; j < n; ++j) count += h_mask[j] ? 1 : 0;
    } else {
        count = n;
    }
    if (count == 0) return BF_OK;
    std::vector<unsigned char> compact(count * kCompact);
    const unsigned char* recs = static_cast<const unsigned char*>(h_structs);
    size_t k = 0;
    for (size_t j = 0; j < n; ++j) {
        if (h_mask && !h_mask[j]) continue;
        std::memcpy(compact.data() + k * kCompact, recs + j * stride, BF_NODE_EVAL_BYTES);
        ++k;
    }
    const size_t parts = parts_for(m, count);
    ShardLocks locks(m, parts, false);
    int rc = BF_OK;
    size_t started = 0;
    auto lo_of = [&](size_t p) { return p * (count / parts) + (p < count % parts ? p : count % parts); };
    for (; started < parts && rc == BF_OK; ++started) {
        bf_ctx* c = m->ctx[started];
        rc = start_struct_shard(c, compact.data() + lo_of(started) * kCompact, lo_of(started + 1) - lo_of(started));
    }
    for (size_t p = 0; p < started; ++p) {
        bf_ctx* c = m->ctx[p];
        const int r = finish_shard(c, rc == BF_OK ? lo_of(p + 1) - lo_of(p) : 0, h_scores + lo_of(p), false);
        if (rc == BF_OK) rc = r;
    }
    if (rc == BF_OK && n_scored) *n_scored = count;
    return rc;
}

// ---- the move-scoring network ---------------------------------------------------------------

namespace {

int check_policy(const bf_ctx* c, size_t n, const char* who) {
    if (int r = check_common(c, n, who)) return r;
    if (!c->d_policy_packed && !c->d_policy_image16)
        return bf::fail(BF_E_ARG, "%s: no move-scoring network loaded (bf_ctx_set_policy)", who);
    return BF_OK;
}

hipError_t launch_policy_features(bf_ctx* c, const uint8_t* d_mailbox, size_t n, float* d_x9, hipStream_t s) {
    if (use_split(c, n))
        return bf::launch_policy_features_t16_split(d_mailbox, bf::kMailboxBytes, false, n, c->d_policy_image16, d_x9, s);
    if (c->precision == BF_PRECISION_BF16)
        return bf::launch_policy_features_t16(d_mailbox, n, c->d_policy_image16, d_x9, s);
    if (use_f32q(c, n)) return bf::launch_policy_features_f32q(d_mailbox, n, c->d_policy_packed, d_x9, s, split_level(c->precision));
    return bf::launch_policy_features_f32(d_mailbox, n, c->d_policy_packed, d_x9, s, split_level(c->precision));
}

// Records -> x9 (move-scoring network's tower) on stream `s`, as eval_from_structs.
int features_from_structs(bf_ctx* c, const void* d_structs, size_t stride, size_t n, float* d_x9, hipStream_t s) {
    Lane* L = &c->policy;
    if (c->precision == BF_PRECISION_BF16 && !c->two_kernels && bf::t16_takes_records(d_structs, stride)) {
        if (use_split(c, n))
            BF_HIP(bf::launch_policy_features_t16_split(static_cast<const uint8_t*>(d_structs), stride, true, n,
                                                        c->d_policy_image16, d_x9, s));
        else
            BF_HIP(bf::launch_policy_features_t16_structs(d_structs, stride, n, c->d_policy_image16, d_x9, s));
        return BF_OK;
    }
    if (int r = ensure_boards(L, n)) return r;
    if (int r = scratch_acquire(L, s)) return r;
    BF_HIP(bf::launch_mailbox_from_structs(d_structs, stride, n, L->d_mailbox, s));
    BF_HIP(launch_policy_features(c, L->d_mailbox, n, d_x9, s));
    return scratch_release(L, s);
}

const float* policy_logit_w(const bf_ctx* c) {
    if (c->precision == BF_PRECISION_BF16) return reinterpret_cast<const float*>(c->d_policy_image16 + bf::t16::kOffHead);
    return c->d_policy_packed + bf::kOffHead;
}

int ensure_x9(bf_ctx* c, size_t n) {
    if (n <= c->cap_x9_boards) return BF_OK;
    size_t cap = c->cap_x9_boards ? c->cap_x9_boards : 1024;
    while (cap < n) cap *= 2;
    if (c->d_x9) (void)hipFree(c->d_x9);
    c->d_x9 = nullptr; c->cap_x9_boards = 0;
    BF_HIP(hipMalloc(&c->d_x9, cap * BF_POLICY_FEATURES_PER_BOARD * sizeof(float)));
    c->cap_x9_boards = cap;
    return BF_OK;
}

}  // namespace

extern "C" int bf_ctx_set_policy(bf_ctx* c, const float* h_weights, size_t n_floats) {
    if (!c || !h_weights) return bf::fail(BF_E_ARG, "bf_ctx_set_policy: null pointer");
    if (n_floats != bf::kPolicyWeightFloats)
        return bf::fail(BF_E_ARG, "bf_ctx_set_policy: expected %zu weight floats, got %zu", bf::kPolicyWeightFloats, n_floats);
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lock(c->policy.mu);
    BF_HIP(hipStreamSynchronize(c->policy.stream));
    if (c->precision == BF_PRECISION_BF16) {
        std::vector<unsigned char> image(bf::t16::kImageBytes);
        bf::pack_weights_t16(h_weights, image.data(), true);
        if (!c->d_policy_image16) BF_HIP(hipMalloc(&c->d_policy_image16, image.size()));
        BF_HIP(hipMemcpy(c->d_policy_image16, image.data(), image.size(), hipMemcpyHostToDevice));
    } else {
        std::vector<float> packed(bf::kPackedFloats);
        bf::pack_weights(h_weights, packed.data(), true, split_level(c->precision));
        if (!c->d_policy_packed) BF_HIP(hipMalloc(&c->d_policy_packed, bf::kPackedFloats * sizeof(float)));
        BF_HIP(hipMemcpy(c->d_policy_packed, packed.data(), bf::kPackedFloats * sizeof(float), hipMemcpyHostToDevice));
    }
    c->policy_started = false;
    return BF_OK;
}

extern "C" int bf_policy_features_wire(bf_ctx* c, const uint8_t* d_pf, size_t n_bytes, const uint64_t* d_occ, size_t n,
                                       float* d_x9, void* stream) {
    if (int r = check_policy(c, n, "bf_policy_features_wire")) return r;
    if (n == 0) return BF_OK;
    if (!d_occ || !d_x9 || (!d_pf && n_bytes)) return bf::fail(BF_E_ARG, "bf_policy_features_wire: null pointer");
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lock(c->policy.mu);
    if (int r = ensure_boards(&c->policy, n)) return r;
    if (int r = scratch_acquire(&c->policy, (hipStream_t)stream)) return r;
    BF_HIP(bf::launch_mailbox_from_wire(d_pf, n_bytes, d_occ, n, workspace(&c->policy), c->policy.d_mailbox, (hipStream_t)stream));
    BF_HIP(launch_policy_features(c, c->policy.d_mailbox, n, d_x9, (hipStream_t)stream));
    return scratch_release(&c->policy, (hipStream_t)stream);
}

extern "C" int bf_policy_features_structs(bf_ctx* c, const void* d_structs, size_t stride, size_t n, float* d_x9,
                                          void* stream) {
    if (int r = check_policy(c, n, "bf_policy_features_structs")) return r;
    if (n == 0) return BF_OK;
    if (!d_structs || !d_x9) return bf::fail(BF_E_ARG, "bf_policy_features_structs: null pointer");
    if (stride < BF_NODE_EVAL_BYTES) return bf::fail(BF_E_ARG, "bf_policy_features_structs: stride %zu < 82", stride);
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lock(c->policy.mu);
    return features_from_structs(c, d_structs, stride, n, d_x9, (hipStream_t)stream);
}

extern "C" int bf_policy_scores(bf_ctx* c, const float* d_x9, size_t n_boards, const uint32_t* d_board_index,
                                const uint8_t* d_from, const uint8_t* d_filter, size_t n_moves, float* d_scores,
                                void* stream) {
    if (int r = check_policy(c, n_boards, "bf_policy_scores")) return r;
    if (n_moves == 0) return BF_OK;
    if (n_boards == 0) return bf::fail(BF_E_ARG, "bf_policy_scores: moves given without boards");
    if (n_moves > ((size_t)1 << 31)) return bf::fail(BF_E_ARG, "bf_policy_scores: %zu moves exceeds the limit of 2^31", n_moves);
    if (!d_x9 || !d_board_index || !d_from || !d_filter || !d_scores) return bf::fail(BF_E_ARG, "bf_policy_scores: null pointer");
    DeviceGuard guard(c->device);
    BF_HIP(bf::launch_policy_scores(d_x9, policy_logit_w(c), d_board_index, d_from, d_filter, n_moves, n_boards,
                                    c->policy.d_err, d_scores, (hipStream_t)stream));
    return BF_OK;
}

extern "C" int bf_policy_check(bf_ctx* c, void* stream) {
    if (!c) return bf::fail(BF_E_ARG, "bf_policy_check: null context");
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lock(c->policy.mu);
    return read_err_flag(&c->policy, (hipStream_t)stream);
}

namespace {

// ---- the host side of a prediction, with the move-scoring lane's mutex held by the caller (one context's
// entry points below, and the multi-device ones, which hold every shard's lane) ------------------------------

// Stage 1 from a wire-format shard whose length the caller has checked; start_nibble = 1: the first code is the
// low nibble of h_pf[0].
int policy_start_wire_locked(bf_ctx* c, const uint8_t* h_pf, size_t n_bytes, const uint64_t* h_occ, size_t n, unsigned start_nibble) {
    Lane* L = &c->policy;
    c->policy_started = false;
    if (n) {
        // buffers are reused (and may be regrown): whatever an abandoned prediction left running is drained
        BF_HIP(hipStreamSynchronize(L->stream));
        if (int r = ensure_boards(L, n)) return r;
        if (int r = ensure_x9(c, n)) return r;
        if (int r = scratch_acquire(L, L->stream)) return r;
        bool direct = false;
        size_t staged = 0;
        StagedWire fw;
        const unsigned char* fin = nullptr;
        const bool want_fused = use_split(c, n) && n <= kHostOffsetBoards;
        if (int r = wire_to_mailbox(L, h_pf, n_bytes, h_occ, n, start_nibble, 1, true, &direct, &staged, want_fused ? &fw : nullptr,
                                    want_fused ? &fin : nullptr))
            return r;
        if (fin)
            BF_HIP(bf::launch_t16_split_wire(fin + fw.off_pf, n_bytes, reinterpret_cast<const uint64_t*>(fin),
                                             reinterpret_cast<const unsigned*>(fin + fw.off_offsets), n, start_nibble, L->d_err,
                                             c->d_policy_image16, c->d_x9, true, L->stream));
        else
            BF_HIP(launch_policy_features(c, L->d_mailbox, n, c->d_x9, L->stream));
        if (int r = scratch_release(L, L->stream)) return r;
        c->policy_in_used = (staged + 255) & ~(size_t)255;
        c->policy_no_flags = false;  // the wire decoder reports malformed input through the flag
    }
    c->policy_boards = n;
    c->policy_started = true;
    return BF_OK;   // the GPU works on while the caller generates the moves to score
}

// Stage 1 from node records: the m records that h_mask (nullable = all) selects among n.
int policy_start_structs_locked(bf_ctx* c, const void* h_structs, size_t stride, const uint8_t* h_mask, size_t n, size_t m) {
    Lane* L = &c->policy;
    c->policy_started = false;
    constexpr size_t kCompact = 88;
    if (m) {
        BF_HIP(hipStreamSynchronize(L->stream));
        if (int r = ensure_x9(c, m)) return r;
        // room behind the records for stage 2's move lists (kPolicyMoveRoom moves a board), so that
        // the finish call can stage them while stage 1 still reads the records
        const size_t rec_bytes = (m * kCompact + 255) & ~(size_t)255;
        if (int r = ensure_staging(L, rec_bytes + m * kPolicyMoveRoom * 6 + 64, m * kPolicyMoveRoom)) return r;
        const unsigned char* recs = static_cast<const unsigned char*>(h_structs);
        size_t k = 0;
        for (size_t j = 0; j < n; ++j) {
            if (h_mask && !h_mask[j]) continue;
            std::memcpy(L->h_in + k * kCompact, recs + j * stride, BF_NODE_EVAL_BYTES);
            ++k;
        }
        const bool direct = m <= direct_boards() && L->h_in_dev;
        if (!direct) BF_HIP(hipMemcpyAsync(L->d_in, L->h_in, m * kCompact, hipMemcpyHostToDevice, L->stream));
        if (int r = features_from_structs(c, direct ? L->h_in_dev : L->d_in, kCompact, m, c->d_x9, L->stream)) return r;
        c->policy_in_used = rec_bytes;
        c->policy_no_flags = true;   // nothing in this prediction's first stage can raise the device's error flag
    }
    c->policy_boards = m;
    c->policy_started = true;
    return BF_OK;
}

// Stage 2 in two halves, so that several devices' second stages run side by side: enqueue (argument checks,
// staging of the move lists, the gather kernel, the copy back), then wait (synchronise, flag, scores out).
struct PolicyPending {
    size_t n_moves = 0;
    bool sync_only = false;   // nothing can have raised the flag: synchronise without reading it
};

int policy_finish_enqueue_locked(bf_ctx* c, const uint8_t* h_from, const uint8_t* h_filter, const uint8_t* h_moves_per_board,
                                 size_t n_boards, size_t n_moves, PolicyPending* pending, const char* who) {
    Lane* L = &c->policy;
    *pending = PolicyPending{};
    if (!c->policy_started) return bf::fail(BF_E_ARG, "%s: no prediction was started", who);
    c->policy_started = false;
    if (n_boards != c->policy_boards)
        return bf::fail(BF_E_ARG, "%s: %zu boards, but the prediction was started with %zu", who, n_boards, c->policy_boards);
    if (n_moves && (!h_from || !h_filter || !h_moves_per_board))
        return bf::fail(BF_E_ARG, "%s: null pointer", who);
    size_t total = 0;
    for (size_t b = 0; b < n_boards; ++b) total += h_moves_per_board[b];
    if (total != n_moves)
        return bf::fail(BF_E_FORMAT, "%s: moves_per_board sums to %zu, but %zu moves were given", who, total, n_moves);
    for (size_t m = 0; m < n_moves; ++m)
        if (h_from[m] >= 64 || h_filter[m] >= BF_POLICY_FILTERS)
            return bf::fail(BF_E_FORMAT, "%s: move %zu has from-square %u, filter %u", who, m, h_from[m], h_filter[m]);
    pending->n_moves = n_moves;
    if (n_moves == 0) return BF_OK;
    // staging: board index u32[n_moves] | from u8[n_moves] | filter u8[n_moves]; np.repeat of the
    // board index as numpy_style_repeat_1d_creator does on the device (ann_creation_helper.py:488-510)
    const size_t from_off = n_moves * sizeof(uint32_t), filter_off = from_off + ((n_moves + 15) & ~(size_t)15);
    const size_t bytes = filter_off + n_moves;
    // The move lists go behind the records that stage 1 may still be reading; only when the room reserved
    // there (or the scores buffer) is too small is stage 1 drained and the staging regrown.
    size_t base = c->policy_in_used;
    if (base + bytes + 16 > L->cap_in_bytes || n_moves > L->cap_scores) {
        BF_HIP(hipStreamSynchronize(L->stream));
        if (int r = ensure_staging(L, bytes + 16, n_moves)) return r;
        base = 0;
    }
    uint32_t* idx = reinterpret_cast<uint32_t*>(L->h_in + base);
    size_t m = 0;
    for (size_t b = 0; b < n_boards; ++b)
        for (unsigned k = 0; k < h_moves_per_board[b]; ++k) idx[m++] = (uint32_t)b;
    std::memcpy(L->h_in + base + from_off, h_from, n_moves);
    std::memcpy(L->h_in + base + filter_off, h_filter, n_moves);
    // from-squares, filters and the per-board counts were checked above and the board index is made here, so
    // the kernel's range checks cannot fire: the direct path reads the flag only after a wire-format first stage
    const bool direct = n_moves <= direct_boards() * kDirectMovesPerBoard && L->h_in_dev && L->h_scores_dev;
    const unsigned char* src = direct ? L->h_in_dev + base : L->d_in + base;
    if (!direct) BF_HIP(hipMemcpyAsync(L->d_in + base, L->h_in + base, bytes, hipMemcpyHostToDevice, L->stream));
    BF_HIP(bf::launch_policy_scores(c->d_x9, policy_logit_w(c), reinterpret_cast<const unsigned*>(src), src + from_off,
                                    src + filter_off, n_moves, n_boards, L->d_err, direct ? L->h_scores_dev : L->d_scores,
                                    L->stream));
    if (!direct) BF_HIP(hipMemcpyAsync(L->h_scores, L->d_scores, n_moves * sizeof(float), hipMemcpyDeviceToHost, L->stream));
    pending->sync_only = direct && c->policy_no_flags;
    return BF_OK;
}

int policy_finish_wait_locked(bf_ctx* c, const PolicyPending& pending, float* h_scores) {
    Lane* L = &c->policy;
    if (pending.n_moves == 0) {
        BF_HIP(hipStreamSynchronize(L->stream));
        return read_err_flag(L, L->stream);
    }
    if (pending.sync_only) {
        BF_HIP(hipStreamSynchronize(L->stream));
    } else if (int r = read_err_flag(L, L->stream)) {
        return r;
    }
    std::memcpy(h_scores, L->h_scores, pending.n_moves * sizeof(float));
    return BF_OK;
}

size_t count_selected(const uint8_t* h_mask, size_t n) {
    if (!h_mask) return n;
    size_t m = 0;
    for (size_t j = 0; j < n; ++j) m += h_mask[j] ? 1 : 0;
    return m;
}

}  // namespace

extern "C" int bf_policy_start_wire_host(bf_ctx* c, const uint8_t* h_pf, size_t n_bytes, const uint64_t* h_occ, size_t n) {
    if (int r = check_policy(c, n, "bf_policy_start_wire_host")) return r;
    if (n && (!h_occ || (!h_pf && n_bytes))) return bf::fail(BF_E_ARG, "bf_policy_start_wire_host: null pointer");
    size_t total = 0;
    for (size_t i = 0; i < n; ++i) total += (size_t)__builtin_popcountll(h_occ[i]);
    if ((total + 1) / 2 != n_bytes)
        return bf::fail(BF_E_FORMAT, "bf_policy_start_wire_host: %zu piece_filters bytes, but the masks select %zu codes (%zu bytes)",
                        n_bytes, total, (total + 1) / 2);
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lock(c->policy.mu);
    return policy_start_wire_locked(c, h_pf, n_bytes, h_occ, n, 0);
}

extern "C" int bf_policy_start_structs_host(bf_ctx* c, const void* h_structs, size_t stride, const uint8_t* h_mask, size_t n) {
    if (int r = check_policy(c, n, "bf_policy_start_structs_host")) return r;
    if (n && !h_structs) return bf::fail(BF_E_ARG, "bf_policy_start_structs_host: null pointer");
    if (stride < BF_NODE_EVAL_BYTES) return bf::fail(BF_E_ARG, "bf_policy_start_structs_host: stride %zu < 82", stride);
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lock(c->policy.mu);
    return policy_start_structs_locked(c, h_structs, stride, h_mask, n, count_selected(h_mask, n));
}

extern "C" int bf_policy_finish_host(bf_ctx* c, const uint8_t* h_from, const uint8_t* h_filter,
                                     const uint8_t* h_moves_per_board, size_t n_boards, size_t n_moves, float* h_scores) {
    if (int r = check_policy(c, n_boards, "bf_policy_finish_host")) return r;
    if (n_moves && !h_scores) return bf::fail(BF_E_ARG, "bf_policy_finish_host: null pointer");
    DeviceGuard guard(c->device);
    std::lock_guard<std::mutex> lock(c->policy.mu);
    PolicyPending pending;
    if (int r = policy_finish_enqueue_locked(c, h_from, h_filter, h_moves_per_board, n_boards, n_moves, &pending, "bf_policy_finish_host"))
        return r;
    return policy_finish_wait_locked(c, pending, h_scores);
}

// ---- the move-scoring network over the devices of a bf_multi ------------------------------------------------
// Boards are cut into contiguous shards exactly as the value calls cut them; every device runs stage 1 on its
// shard; the finish call hands every device the moves of its boards (moves arrive board by board, so a shard's
// moves are one contiguous run), enqueues every second stage and only then waits.

namespace {

int multi_policy_check(const bf_multi* m, size_t n, const char* who) {
    if (!m) return bf::fail(BF_E_ARG, "%s: null handle", who);
    for (bf_ctx* c : m->ctx)
        if (int r = check_policy(c, n, who)) return r;
    return BF_OK;
}

}  // namespace

extern "C" int bf_multi_policy_start_wire_host(bf_multi* m, const uint8_t* h_pf, size_t n_bytes, const uint64_t* h_occ, size_t n) {
    if (int r = multi_policy_check(m, n, "bf_multi_policy_start_wire_host")) return r;
    if (n && (!h_occ || (!h_pf && n_bytes))) return bf::fail(BF_E_ARG, "bf_multi_policy_start_wire_host: null pointer");
    std::lock_guard<std::mutex> lock(m->policy_mu);
    m->policy_started = false;
    const size_t parts = parts_for(m, n);
    std::vector<size_t> nibble_lo(parts + 1, 0);
    m->policy_lo.assign(parts + 1, 0);
    if (n) {
        if (int r = bf_host_split_wire(h_occ, n, parts, m->policy_lo.data(), nibble_lo.data())) return r;
    }
    const size_t total = nibble_lo[parts];
    if ((total + 1) / 2 != n_bytes)
        return bf::fail(BF_E_FORMAT, "bf_multi_policy_start_wire_host: %zu piece_filters bytes, but the masks select %zu codes (%zu bytes)",
                        n_bytes, total, (total + 1) / 2);
    ShardLocks locks(m, parts, true);
    for (size_t p = 0; p < parts; ++p) {
        bf_ctx* c = m->ctx[p];
        DeviceGuard guard(c->device);
        if (!guard.ok) return bf::fail(BF_E_HIP, "hipSetDevice(%d) failed", c->device);
        const size_t s_lo = nibble_lo[p], s_hi = nibble_lo[p + 1];
        const size_t byte_lo = s_lo >> 1, byte_hi = (s_hi + 1) >> 1;
        if (int r = policy_start_wire_locked(c, h_pf + byte_lo, byte_hi - byte_lo, h_occ + m->policy_lo[p],
                                             m->policy_lo[p + 1] - m->policy_lo[p], (unsigned)(s_lo & 1)))
            return r;
    }
    m->policy_started = true;
    return BF_OK;
}

extern "C" int bf_multi_policy_start_structs_host(bf_multi* m, const void* h_structs, size_t stride, const uint8_t* h_mask, size_t n) {
    if (int r = multi_policy_check(m, n, "bf_multi_policy_start_structs_host")) return r;
    if (n && !h_structs) return bf::fail(BF_E_ARG, "bf_multi_policy_start_structs_host: null pointer");
    if (stride < BF_NODE_EVAL_BYTES) return bf::fail(BF_E_ARG, "bf_multi_policy_start_structs_host: stride %zu < 82", stride);
    std::lock_guard<std::mutex> lock(m->policy_mu);
    m->policy_started = false;
    const size_t count = count_selected(h_mask, n);
    const size_t parts = parts_for(m, count);
    m->policy_lo.assign(parts + 1, 0);
    for (size_t p = 0; p <= parts; ++p) m->policy_lo[p] = p * (count / parts) + (p < count % parts ? p : count % parts);
    // first record (selected or not) of every shard: shard p takes the records from there up to shard p + 1's first
    std::vector<size_t> rec_lo(parts + 1, n);
    {
        size_t k = 0, p = 0;
        for (size_t j = 0; j < n && p < parts; ++j) {
            if (h_mask && !h_mask[j]) continue;
            while (p < parts && k == m->policy_lo[p]) rec_lo[p++] = j;
            ++k;
        }
        rec_lo[0] = 0;
    }
    ShardLocks locks(m, parts, true);
    const unsigned char* recs = static_cast<const unsigned char*>(h_structs);
    for (size_t p = 0; p < parts; ++p) {
        bf_ctx* c = m->ctx[p];
        DeviceGuard guard(c->device);
        if (!guard.ok) return bf::fail(BF_E_HIP, "hipSetDevice(%d) failed", c->device);
        const size_t j0 = rec_lo[p], j1 = rec_lo[p + 1];
        if (int r = policy_start_structs_locked(c, recs + j0 * stride, stride, h_mask ? h_mask + j0 : nullptr, j1 - j0,
                                                m->policy_lo[p + 1] - m->policy_lo[p]))
            return r;
    }
    m->policy_started = true;
    return BF_OK;
}

extern "C" int bf_multi_policy_finish_host(bf_multi* m, const uint8_t* h_from, const uint8_t* h_filter,
                                           const uint8_t* h_moves_per_board, size_t n_boards, size_t n_moves, float* h_scores) {
    if (int r = multi_policy_check(m, n_boards, "bf_multi_policy_finish_host")) return r;
    if (n_moves && (!h_from || !h_filter || !h_moves_per_board || !h_scores))
        return bf::fail(BF_E_ARG, "bf_multi_policy_finish_host: null pointer");
    std::lock_guard<std::mutex> lock(m->policy_mu);
    if (!m->policy_started) return bf::fail(BF_E_ARG, "bf_multi_policy_finish_host: no prediction was started");
    m->policy_started = false;
    const size_t parts = m->policy_lo.size() - 1;
    if (n_boards != m->policy_lo[parts])
        return bf::fail(BF_E_ARG, "bf_multi_policy_finish_host: %zu boards, but the prediction was started with %zu", n_boards,
                        m->policy_lo[parts]);
    size_t total = 0;
    std::vector<size_t> move_lo(parts + 1, 0);
    for (size_t p = 0, b = 0; p < parts; ++p) {
        move_lo[p] = total;
        for (; b < m->policy_lo[p + 1]; ++b) total += h_moves_per_board[b];
    }
    move_lo[parts] = total;
    if (total != n_moves)
        return bf::fail(BF_E_FORMAT, "bf_multi_policy_finish_host: moves_per_board sums to %zu, but %zu moves were given", total, n_moves);
    ShardLocks locks(m, parts, true);
    std::vector<PolicyPending> pending(parts);
    int rc = BF_OK;
    size_t started = 0;
    for (; started < parts && rc == BF_OK; ++started) {
        bf_ctx* c = m->ctx[started];
        DeviceGuard guard(c->device);
        const size_t b0 = m->policy_lo[started], m0 = move_lo[started];
        rc = policy_finish_enqueue_locked(c, h_from ? h_from + m0 : nullptr, h_filter ? h_filter + m0 : nullptr,
                                          h_moves_per_board ? h_moves_per_board + b0 : nullptr, m->policy_lo[started + 1] - b0,
                                          move_lo[started + 1] - m0, &pending[started], "bf_multi_policy_finish_host");
    }
    for (size_t p = 0; p < started; ++p) {
        bf_ctx* c = m->ctx[p];
        DeviceGuard guard(c->device);
        if (rc == BF_OK) {
            rc = policy_finish_wait_locked(c, pending[p], h_scores ? h_scores + move_lo[p] : nullptr);
        } else {
            (void)hipStreamSynchronize(c->policy.stream);   // drain what a failed call left enqueued
        }
    }
    // shards that never reached their second stage keep an unfinished prediction: abandon it
    for (size_t p = 0; p < parts; ++p) m->ctx[p]->policy_started = false;
    return rc;
}
