/* batchfirst.h -- C ABI of the MI355X-native Batch-First board-evaluation path.
 *
 * One shared library (libbatchfirst_hip.so, built from batch_first_amd/csrc for gfx950).
 * Plain pointers and sizes only; no torch, numpy or TensorFlow types.  Every entry point
 * names the reference interface it stands in for (paths relative to the reference tree).
 *
 * Conventions
 *   - return value 0 = ok, negative = error (BF_E_*); bf_last_error() gives the message of
 *     the calling thread's last failure.
 *   - "d_" arguments are device pointers on the context's GPU, "h_" arguments host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  Device
 *     entry points only enqueue work on it; they never synchronise.  Calls on one context may use
 *     different streams and threads: entry points that go through the context's scratch (wire
 *     decode, mailboxes) order themselves behind the previous such call with an event, so a call
 *     on stream B never overwrites scratch that a call on stream A is still reading (the callers'
 *     own input and output buffers remain the callers' business).
 *   - Square numbering, bitboards, colours as in batch_first/__init__.py:176-205
 *     (a1 = bit 0, h8 = bit 63, occupied_co[0] = black, [1] = white, turn 1 = white).
 *
 * Wire format (what the reference's evaluation callables are fed,
 * batch_first/numba_negamax_zero_window.py:71-105 and :11-19):
 *   occupied_bbs  uint64[n]   per board: occupied | ep-square bit, byte-swapped when black
 *                             is to move (mover's frame);
 *   piece_filters uint8[ceil(S/2)], S = sum of popcounts: one 4-bit code per set bit, boards
 *                             back to back, ascending square order, even-index code in the
 *                             HIGH nibble, a trailing 0 nibble when S is odd.
 *   codes: 0 ep square, 1..6 mover's K Q R B N P, 7 mover's castling rook,
 *          8..13 opponent's K Q R B N P, 14 opponent's castling rook.
 *
 * Node records (batch_first/classes_and_structs.py:7-29): the path reads bytes 0..81 of each
 * record: pawns@0 knights@8 bishops@16 rooks@24 queens@32 kings@40 occupied_co[2]@48
 * occupied@64 (all uint64) turn@72 (int8) castling_rights@73 (uint64, unaligned)
 * ep_square@81 (uint8, 0 = none).  `stride` is the record size in bytes (806 for the
 * reference's dtype, >= 82 for any compact copy); no alignment is required.
 *
 * Planes: float32 [n][15][8][8], planes[b][code][rank][file] = 1.0f for every relevant square
 * in the mover's frame, else 0.0f (batch_first/anns/ann_creation_helper.py:76-79).
 *
 * Weights: one float32 blob, see BF_WEIGHT_FLOATS and batch_first_amd/weights.py:
 *   9 input-conv kernels HWIO [k][k][15][C_i], C = 64,32,20,20,18,18,16,24,24,
 *     k = 3 (dilation 1..7) then 2 (dilation 2x4, 4x2)       (ann_creation_helper.py:211-236)
 *   input batch-norm mean[236], var[236], beta[236]           (ann_creation_helper.py:240-246)
 *   9 tower layers, each kernel HWIO then mean, var, beta:
 *     1x1 236->512, 1x1 512->128, 3x3 128->32, 5x 3x3 32->32, 8x8(valid) 32->16
 *                                                             (evaluation_ann.py:268)
 *   logit kernel [1][1][16][1]                                (evaluation_ann.py:35-44)
 *   batch-norm is inference mode, no gamma, epsilon 1e-3; no conv has a bias.
 */
#ifndef BATCHFIRST_H
#define BATCHFIRST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BF_WEIGHT_FLOATS 333608u /* 330356 kernel floats + 3 * 1084 batch-norm floats */
#define BF_NODE_EVAL_BYTES 82u
#define BF_PLANES_PER_BOARD 960u /* 15 * 64 floats */

enum {
    BF_OK = 0,
    BF_E_ARG = -1,      /* bad argument (null pointer, size mismatch) */
    BF_E_HIP = -2,      /* a HIP runtime call failed */
    BF_E_FORMAT = -3,   /* malformed wire input (stream length vs popcounts, code 15) */
    BF_E_NOGPU = -4     /* no usable gfx950 device */
};

/* Arithmetic of the value network inside a context (bf_ctx_create_ex).
 *   BF_PRECISION_F32   float32 operands and sums (v_mfma_f32_32x32x2_f32): the parity path, scores
 *                      within 1e-4 relative of the float32 CPU evaluation of the reference's graph.
 *   BF_PRECISION_BF16  batch-norm-folded weights and every activation x1..x9 rounded to bf16
 *                      (round-to-nearest-even), float32 sums, shifts, x10 and logit
 *                      (v_mfma_f32_16x16x32_bf16).  No reference counterpart: the reference runs
 *                      TensorRT/TensorFlow in FP32 only (ann_creation_helper.py:182-187).
 *   BF_PRECISION_F32_SPLIT  the F32 kernel's structure with the layers that take 97 % of its matrix time on
 *                      v_mfma_f32_32x32x16_bf16, as sums of bf16 x bf16 products with float32 accumulation: the 1x1
 *                      pair (236 -> 512 -> 128) with kernels and activations as TWO bf16 terms each (hi + mid = 16
 *                      significant bits, four products), the six 3x3 layers with THREE terms each (24 bits, the six
 *                      products that matter); input convolutions, head and every shift as in F32.  Scores within
 *                      1e-4 relative of the float32 evaluation (measured worst 0.62 of that bound on the golden sets,
 *                      F32 itself 0.08), not bit-equal to F32's; 2.1-2.2 times F32's boards per second.
 *   BF_PRECISION_F32_SPLIT3  the same with THREE terms in the 1x1 pair as well (the third term of the 236-channel
 *                      activation lives in LDS): every matrix product carries 24 significant bits, scores as close
 *                      to the float32 evaluation as F32's own (measured worst 0.11 of the bound, F32 0.08); 1.7 times F32's boards per
 *                      second (F32_SPLIT: 2.2).  Neither has a reference counterpart (as BF16). */
enum { BF_PRECISION_F32 = 0, BF_PRECISION_BF16 = 1, BF_PRECISION_F32_SPLIT = 2, BF_PRECISION_F32_SPLIT3 = 3 };

typedef struct bf_ctx bf_ctx;

/* Thread-local text of the last error returned to this thread ("" if none). */
const char* bf_last_error(void);

/* Library identification: "batchfirst-hip <version> gfx950". */
const char* bf_version(void);

/* Create an evaluator on GPU `device`: folds batch-norm into the kernels, arranges them in
 * MFMA fragment order and uploads them.  Stands in for loading the frozen graph into a
 * tf.Session: batch_first/chestimator.py:51-52, :10-26. */
int bf_ctx_create(bf_ctx** out, int device, const float* h_weights, size_t n_floats);

/* bf_ctx_create with a choice of arithmetic (BF_PRECISION_*); bf_ctx_create is the F32 case.
 * Every evaluation entry point below then runs in that precision; encoding is unaffected. */
int bf_ctx_create_ex(bf_ctx** out, int device, const float* h_weights, size_t n_floats, int precision);

/* The precision a context was created with. */
int bf_ctx_precision(const bf_ctx* ctx);

/* BF16 contexts run calls of at most this many boards on a second kernel with the same arithmetic (one
 * board per workgroup instead of one per wave: a third of the latency while the boards are few; the
 * scores are bit-identical).  Process-wide; returns the previous value; 0 turns the second kernel off.
 * The environment variable BF_SPLIT_BOARDS sets the start value.  No counterpart in the reference (a
 * tuning knob of this library: the search's batches at shallow depths are a hundred boards). */
#define BF_SPLIT_BOARDS_DEFAULT 512u
size_t bf_set_split_boards(size_t n_boards);

/* Free device memory and streams.  Stands in for closer_fn / sess.close():
 * batch_first/chestimator.py:72. */
void bf_ctx_destroy(bf_ctx* ctx);

/* ---- evaluation, device pointers (asynchronous on `stream`) ------------------------- */

/* scores[b] = value_network(planes(board b)) for wire-format input.  Stands in for the graph
 * run behind board_predictor: input_parser + value_network/Squeeze,
 * batch_first/chestimator.py:38,55-62; ann_creation_helper.py:21-84; evaluation_ann.py:17-46.
 * Malformed input sets a sticky flag read by bf_ctx_check(); out-of-range reads are clamped. */
int bf_eval_wire(bf_ctx* ctx, const uint8_t* d_piece_filters, size_t n_bytes,
                 const uint64_t* d_occupied_bbs, size_t n, float* d_scores, void* stream);

/* Same scores straight from node records: replaces the host encoder
 * struct_array_to_ann_inputs (batch_first/numba_negamax_zero_window.py:71-105) plus the
 * host-to-device copy plus the graph run.  One score per record, in order (select records on
 * the host with bf_eval_structs_host's mask, or gather them before the call). */
int bf_eval_structs(bf_ctx* ctx, const void* d_structs, size_t stride, size_t n,
                    float* d_scores, void* stream);

/* The same call for callers that cannot allocate device memory: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.
 * bf_eval_structs keeps the 4-bit mailboxes between its two kernels in scratch that the context
 * allocates and grows on demand; here they live in a file-scope device array of the library
 * (8 MiB on every GPU the library runs on, held until the process ends), so the call allocates,
 * grows and frees nothing, takes no lock and synchronises nothing.  Same scores, bit for bit.
 * Limit: n <= BF_NOALLOC_MAX_BOARDS per call (BF_E_ARG beyond it).  All calls on one GPU share the
 * array: calls that overlap in time on one GPU (two streams, two contexts, two threads) race;
 * calls in stream order do not.  (Creating the context does allocate: do that outside the
 * restricted region.) */
#define BF_NOALLOC_MAX_BOARDS 262144u
int bf_eval_structs_noalloc(bf_ctx* ctx, const void* d_structs, size_t stride, size_t n,
                            float* d_scores, void* stream);

/* ---- encoding alone, device pointers (for the bit-exact check and the HBM figure) ---- */

/* Wire format -> planes.  Stands in for parse_into_ann_input_inference,
 * batch_first/anns/ann_creation_helper.py:21-84. */
int bf_encode_planes_wire(bf_ctx* ctx, const uint8_t* d_piece_filters, size_t n_bytes,
                          const uint64_t* d_occupied_bbs, size_t n, float* d_planes, void* stream);

/* Node records -> planes: struct_array_to_ann_inputs followed by
 * parse_into_ann_input_inference in one pass (numba_negamax_zero_window.py:71-105,
 * ann_creation_helper.py:21-84). */
int bf_encode_planes_structs(bf_ctx* ctx, const void* d_structs, size_t stride, size_t n,
                             float* d_planes, void* stream);

/* Node records / wire format -> 4-bit mailboxes, uint8 [n][32]: square 2j in the low nibble of
 * byte j, square 2j+1 in the high nibble, 15 = square not relevant.  This is the compact
 * device-side form of the planes that the evaluation kernel consumes.  No reference
 * counterpart (the reference materialises the float planes). */
int bf_encode_mailbox_wire(bf_ctx* ctx, const uint8_t* d_piece_filters, size_t n_bytes,
                           const uint64_t* d_occupied_bbs, size_t n, uint8_t* d_mailbox, void* stream);
int bf_encode_mailbox_structs(bf_ctx* ctx, const void* d_structs, size_t stride, size_t n,
                              uint8_t* d_mailbox, void* stream);

/* scores from mailboxes (the network alone). */
int bf_eval_mailbox(bf_ctx* ctx, const uint8_t* d_mailbox, size_t n, float* d_scores, void* stream);

/* Synchronise `stream` and report the sticky malformed-input flag (then clear it). */
int bf_ctx_check(bf_ctx* ctx, void* stream);

/* ---- evaluation, host pointers (synchronous; what the Python callable uses) --------- */

/* board_predictor(piece_filters, occupied_bbs) -> float32[n]
 * (batch_first/chestimator.py:38; caller batch_first/numba_negamax_zero_window.py:191-197).
 * Validates the stream length against the popcounts, copies in, evaluates, copies out.
 * Re-entrant: calls on one context are serialised internally. */
int bf_eval_wire_host(bf_ctx* ctx, const uint8_t* h_piece_filters, size_t n_bytes,
                      const uint64_t* h_occupied_bbs, size_t n, float* h_scores);

/* start_board_evaluations' work without the host encoder
 * (batch_first/numba_negamax_zero_window.py:183-202): gathers the records selected by h_mask
 * (nullable = all), evaluates them, writes one score per selected record.
 * *n_scored (nullable) receives the count. */
int bf_eval_structs_host(bf_ctx* ctx, const void* h_structs, size_t stride, const uint8_t* h_mask,
                         size_t n, float* h_scores, size_t* n_scored);

/* ---- several GPUs behind one call ------------------------------------------------------------
 *
 * No counterpart in the reference (every placement there is '/GPU:0', chestimator.py:33).  north_star:
 * "independent boards on each GPU on separate HIP streams ... gather scores back to the search host".
 * A bf_multi borrows one context per GPU (same weights and precision in each, which the caller
 * guarantees by creating them from one blob) and splits every batch into contiguous shards, one per
 * device: all shards' copies and kernels are enqueued on the contexts' own streams from the calling
 * thread before any is waited for, so the devices run side by side, and the scores land in one host
 * array in board order.  The wire format is split at sum-of-popcount boundaries; a shard that begins
 * on an odd code starts on the low nibble of its first byte, which the decode kernel is told
 * (no re-packing).  Batches below min_shard boards per device use fewer devices (a search's small
 * batches stay on one GPU).  Signatures match bf_eval_wire_host / bf_eval_structs_host with the
 * handle in place of the context, so the search driver takes either. */
typedef struct bf_multi bf_multi;
int bf_multi_create(bf_multi** out, bf_ctx* const* ctxs, size_t n_ctx, size_t min_shard);
void bf_multi_destroy(bf_multi* m);   /* the contexts are the caller's to destroy */
size_t bf_multi_devices(const bf_multi* m);
int bf_multi_eval_wire_host(bf_multi* m, const uint8_t* h_piece_filters, size_t n_bytes,
                            const uint64_t* h_occupied_bbs, size_t n, float* h_scores);
int bf_multi_eval_structs_host(bf_multi* m, const void* h_structs, size_t stride, const uint8_t* h_mask,
                               size_t n, float* h_scores, size_t* n_scored);
/* Change min_shard (0 = leave it); returns the previous value.  batch_first_amd/chestimator.py measures the
 * crossover on the box it runs on (the batch size from which two devices finish a call sooner than one). */
size_t bf_multi_set_min_shard(bf_multi* m, size_t min_shard);

/* ---- the move-scoring ("policy") network ------------------------------------------------
 *
 * batch_first/anns/move_evaluation_ann.py:10-73: the same nine input convolutions and tower as
 * the value network minus its 8x8 layer ([[512,1],[128,1]] + 6*[[32,3]], :291), a bias-free 1x1
 * logit conv 32 -> 73 (:35-44, NUM_LOGIT_FILTERS :280), reshaped to [boards][73][64] and gathered
 * at (board, move_filter, from_square) (:46-73).  The reference runs it in two stages through
 * session.partial_run (batch_first/chestimator.py:40-46): stage 1 is fed the boards, the search
 * generates moves meanwhile, stage 2 is fed (from_square, move_filter, moves_per_board) and
 * returns one score per move.  Here stage 1 leaves the 32 x 64 activations in front of the logit
 * layer on the GPU and stage 2 applies the logit kernel only where a move asks for it.
 *
 * Weights: float32 blob of BF_POLICY_WEIGHT_FLOATS, laid out as the value network's with the
 * 8x8 layer and its batch-norm left out and the logit kernel HWIO [1][1][32][73].  float32 only.
 * Squares and filters arrive in the mover's frame (SQUARES_180 for black,
 * numba_negamax_zero_window.py:512-515; MOVE_FILTER_LOOKUP, batch_first/__init__.py:51-95). */
#define BF_POLICY_WEIGHT_FLOATS 303112u
#define BF_POLICY_FILTERS 73u
#define BF_POLICY_FEATURES_PER_BOARD 2048u /* 32 channels x 64 squares */

/* Attach (or replace) the move-scoring network of a context.  Stands in for the policy_network/
 * half of the frozen graph: chestimator.py:23-26, ann_creation_helper.py:130-146. */
int bf_ctx_set_policy(bf_ctx* ctx, const float* h_weights, size_t n_floats);

/* Stage 1 on device pointers: features float32 [n][32][64] of the boards (asynchronous). */
int bf_policy_features_wire(bf_ctx* ctx, const uint8_t* d_piece_filters, size_t n_bytes,
                            const uint64_t* d_occupied_bbs, size_t n, float* d_features, void* stream);
int bf_policy_features_structs(bf_ctx* ctx, const void* d_structs, size_t stride, size_t n,
                               float* d_features, void* stream);

/* Stage 2 on device pointers: scores[m] = logits[board_index[m]][move_filter[m]][from_square[m]]
 * ("policy_network/requested_move_scores", move_evaluation_ann.py:73).  board_index is the
 * np.repeat of arange(n_boards) by moves_per_board (move_evaluation_ann.py:59-61).  Out-of-range
 * values are clamped and reported by bf_policy_check(). */
int bf_policy_scores(bf_ctx* ctx, const float* d_features, size_t n_boards, const uint32_t* d_board_index,
                     const uint8_t* d_from_square, const uint8_t* d_move_filter, size_t n_moves,
                     float* d_scores, void* stream);
int bf_policy_check(bf_ctx* ctx, void* stream);

/* start_move_prediction(piece_filters, occupied_bbs) (chestimator.py:40-42): copies the boards in
 * and starts stage 1 on the context's policy stream, then returns; at most one prediction per
 * context is in flight (a new start abandons an unfinished one).  May overlap a value-network
 * call from another thread (numba_negamax_zero_window.py:178,199). */
int bf_policy_start_wire_host(bf_ctx* ctx, const uint8_t* h_piece_filters, size_t n_bytes,
                              const uint64_t* h_occupied_bbs, size_t n);

/* The same from node records (start_move_scoring without the host encoder,
 * numba_negamax_zero_window.py:164-180): the records selected by h_mask (nullable = all). */
int bf_policy_start_structs_host(bf_ctx* ctx, const void* h_structs, size_t stride, const uint8_t* h_mask, size_t n);

/* The closure start_move_prediction returns (chestimator.py:44-46): feed move_info =
 * (from_square, move_filter, moves_per_board), all uint8 (move_evaluation_ann.py:244-246), get one
 * float32 score per move, boards' moves back to back.  n_boards must equal the started n. */
int bf_policy_finish_host(bf_ctx* ctx, const uint8_t* h_from_square, const uint8_t* h_move_filter,
                          const uint8_t* h_moves_per_board, size_t n_boards, size_t n_moves, float* h_scores);

/* The same three calls over the devices of a bf_multi handle, every context of which must have been given a
 * move-scoring network by bf_ctx_set_policy: the
 * boards are cut into contiguous shards as the value calls cut them, every device runs stage 1 on its shard,
 * and the finish call hands each device the moves of its boards (moves arrive board by board, so a shard's
 * moves are one run), enqueues every second stage and only then waits.  One prediction per handle is in
 * flight.  Signatures match the single-context calls with the handle in place of the context, so the search
 * driver takes either.  No counterpart in the reference (one '/GPU:0' session, chestimator.py:33). */
int bf_multi_policy_start_wire_host(bf_multi* m, const uint8_t* h_piece_filters, size_t n_bytes,
                                    const uint64_t* h_occupied_bbs, size_t n);
int bf_multi_policy_start_structs_host(bf_multi* m, const void* h_structs, size_t stride, const uint8_t* h_mask, size_t n);
int bf_multi_policy_finish_host(bf_multi* m, const uint8_t* h_from_square, const uint8_t* h_move_filter,
                                const uint8_t* h_moves_per_board, size_t n_boards, size_t n_moves, float* h_scores);

/* ---- host-only helpers (no GPU needed) ----------------------------------------------- */

/* The host encoder itself, for callers that still want the wire format:
 * struct_array_to_ann_inputs (batch_first/numba_negamax_zero_window.py:71-105).
 * h_piece_filters must hold ceil(33*n_selected/2)+1 bytes.  Returns the number of stream
 * bytes through *n_bytes and the number of boards through *n_scored. */
int bf_host_encode_wire(const void* h_structs, size_t stride, const uint8_t* h_mask, size_t n,
                        uint8_t* h_piece_filters, uint64_t* h_occupied_bbs,
                        size_t* n_bytes, size_t* n_scored);

/* How bf_multi_eval_wire_host cuts a wire-format batch into n_parts contiguous shards (host only):
 * board_lo[p] and nibble_lo[p] (n_parts + 1 entries each) are shard p's first board and the index of
 * its first code in the stream; an odd nibble_lo means the shard starts on a low nibble. */
int bf_host_split_wire(const uint64_t* h_occupied_bbs, size_t n, size_t n_parts, size_t* board_lo, size_t* nibble_lo);

/* Size in floats of the fragment-ordered weight image the kernels read, and the image itself
 * (bf_ctx_create uploads exactly this).  Exposed so the arrangement can be checked on a CPU. */
size_t bf_packed_floats(void);
int bf_pack_weights(const float* h_weights, size_t n_floats, float* h_packed, size_t n_packed);
/* The image a context of the given precision uploads: BF_PRECISION_F32 (= bf_pack_weights), BF_PRECISION_F32_SPLIT or
 * BF_PRECISION_F32_SPLIT3 (the same size, the 1x1 pair and the 3x3 layers as bf16 terms; tests/kernel_sim.py walks
 * all three on a CPU).  BF_PRECISION_BF16 has an image of its own (below) and is refused here. */
int bf_pack_weights_ex(const float* h_weights, size_t n_floats, float* h_packed, size_t n_packed, int precision);

int bf_pack_policy_weights(const float* h_weights, size_t n_floats, float* h_packed, size_t n_packed);

/* The same for the image of the bf16 kernels (bytes; batch_first_amd/csrc/bf_layout_t16.h), which is
 * what a BF_PRECISION_BF16 context uploads; checked on a CPU by tests/kernel_sim_t16.py. */
size_t bf_packed_t16_bytes(void);
int bf_pack_weights_t16(const float* h_weights, size_t n_floats, void* h_image, size_t n_bytes);

#ifdef __cplusplus
}
#endif
#endif /* BATCHFIRST_H */
