// Launch wrappers implemented in the .hip translation units.  All of them only enqueue.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace bf {

// Scratch the wire decode needs: per-board nibble offsets (two-level exclusive scan of the
// popcounts) and the sticky malformed-input flag.
struct Workspace {
    unsigned* local_off;    // [n]
    unsigned* block_total;  // [scan_blocks_for(n)]
    int* err_flag;          // [1]
};

size_t scan_blocks_for(size_t n);

// start_nibble = 1: the boards are a shard of a longer stream and their first code is the LOW nibble of d_nib[0]
hipError_t launch_mailbox_from_wire(const uint8_t* d_nib, size_t n_bytes, const uint64_t* d_occ, size_t n,
                                    const Workspace& ws, uint8_t* d_mailbox, hipStream_t stream,
                                    unsigned start_nibble = 0);
hipError_t launch_mailbox_from_structs(const void* d_structs, size_t stride, size_t n, uint8_t* d_mailbox,
                                       hipStream_t stream);
hipError_t launch_planes_from_wire(const uint8_t* d_nib, size_t n_bytes, const uint64_t* d_occ, size_t n,
                                   const Workspace& ws, float* d_planes, hipStream_t stream);
hipError_t launch_planes_from_structs(const void* d_structs, size_t stride, size_t n, float* d_planes,
                                      hipStream_t stream);
hipError_t launch_planes_from_mailbox(const uint8_t* d_mailbox, size_t n, float* d_planes, hipStream_t stream);

// The value network: mailboxes -> scores.  `d_packed` is the image bf::pack_weights builds.
// `d_debug` (nullable) receives every intermediate activation, kDebugFloatsPerBoard per board.
constexpr size_t kDebugFloatsPerBoard = 64 * (236 + 128 + 6 * 32) + 16;
// split: the level the image was packed for and the kernel runs at: 0 float32, 1 BF_PRECISION_F32_SPLIT (1x1 pair as
// two-term, 3x3 layers as three-term bf16 products), 2 BF_PRECISION_F32_SPLIT3 (three terms everywhere).
hipError_t launch_eval_f32(const uint8_t* d_mailbox, size_t n, const float* d_packed, float* d_scores,
                           float* d_debug, hipStream_t stream, int split = 0);

// The move-scoring network (packed with pack_weights(..., policy = true)): mailboxes -> x9 features
// float32 [n][32][64]; then one logit per requested move.
hipError_t launch_policy_features_f32(const uint8_t* d_mailbox, size_t n, const float* d_packed, float* d_x9,
                                      hipStream_t stream, int split = 0);
// The same two for small calls: one board per workgroup, eight waves, scores and features bit-identical to the two
// above at the same split level (eval_f32q.hip).
hipError_t launch_eval_f32q(const uint8_t* d_mailbox, size_t n, const float* d_packed, float* d_scores, hipStream_t stream,
                            int split = 0);
hipError_t launch_policy_features_f32q(const uint8_t* d_mailbox, size_t n, const float* d_packed, float* d_x9, hipStream_t stream,
                                       int split = 0);
// d_logit_w: the logit kernel as float32 [73][32] (both images keep it at their head offset)
hipError_t launch_policy_scores(const float* d_x9, const float* d_logit_w, const unsigned* d_board_index,
                                const uint8_t* d_from, const uint8_t* d_filter, size_t n_moves, size_t n_boards,
                                int* d_err, float* d_out, hipStream_t stream);

// Same network, bf16 operands with float32 sums, on v_mfma_f32_16x16x32_bf16 (eval_t16.hip); `d_image` is the
// image bf::pack_weights_t16 builds.
hipError_t launch_eval_t16(const uint8_t* d_mailbox, size_t n, const unsigned char* d_image, float* d_scores,
                           float* d_debug, hipStream_t stream);
hipError_t launch_policy_features_t16(const uint8_t* d_mailbox, size_t n, const unsigned char* d_image, float* d_x9,
                                      hipStream_t stream);
// The same two from node records (88 useful bytes at a 4-byte-aligned address and stride), without a mailbox:
// the evaluating wave decodes its board's record itself.
bool t16_takes_records(const void* d_structs, size_t stride);
hipError_t launch_eval_t16_structs(const void* d_structs, size_t stride, size_t n, const unsigned char* d_image, float* d_scores,
                                   hipStream_t stream);
hipError_t launch_policy_features_t16_structs(const void* d_structs, size_t stride, size_t n, const unsigned char* d_image,
                                              float* d_x9, hipStream_t stream);

}  // namespace bf
