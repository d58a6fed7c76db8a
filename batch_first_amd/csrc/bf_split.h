// Launchers of the small-call form of the 16x16x32 bf16 network (eval_t16s.hip): one board per workgroup,
// one 16-square column tile per wave.  Same weight image, same arithmetic in the same order as eval_t16.hip
// (bit-identical results); a board's walk through the network takes a quarter of the instructions per wave,
// which is what a call of a few hundred boards waits for (the search's batches at depth 4,
// profiles/r2/search).  Reference interface replaced: the same as eval_t16.hip's.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace bf {

// `records`: d_src holds node records (bytes 0..81 of each, `stride` apart, t16_takes_records must hold),
// otherwise 4-bit mailboxes (stride = kMailboxBytes).
hipError_t launch_eval_t16_split(const uint8_t* d_src, size_t stride, bool records, size_t n, const unsigned char* d_image,
                                 float* d_scores, hipStream_t stream);
hipError_t launch_policy_features_t16_split(const uint8_t* d_src, size_t stride, bool records, size_t n,
                                            const unsigned char* d_image, float* d_x9, hipStream_t stream);

// The same from the wire format (nibble stream + relevance masks + every board's code offset, as the host entry
// points stage it): the decode of k_to_mailbox<WireSrc> inside the evaluating workgroup, one launch a call.
// d_nib 4-byte aligned with 3 readable bytes behind n_bytes; d_offsets less than 4 GiB behind d_occ; d_err the lane's
// sticky malformed-input flag (stream too short: 1, code 15: 2).  features: x9 instead of scores.
hipError_t launch_t16_split_wire(const uint8_t* d_nib, size_t n_bytes, const uint64_t* d_occ, const unsigned* d_offsets, size_t n,
                                 unsigned start_nibble, int* d_err, const unsigned char* d_image, float* d_out, bool features,
                                 hipStream_t stream);

}  // namespace bf
