// Launchers of encode.hip that only api.hip uses (kept out of bf_kernels.h, which every evaluation kernel includes).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace bf {

// launch_mailbox_from_wire with every board's nibble offset already known (the host entry points add up the
// popcounts anyway to check the stream's length): no scan kernels.  d_offsets[b] = codes of the boards before b;
// d_zero_bases = zeros, one per kScanBoardsPerBlock boards (the per-block bases of the device scan, which the
// offsets here already include).
hipError_t launch_mailbox_from_wire_offsets(const uint8_t* d_nib, size_t n_bytes, const uint64_t* d_occ, size_t n,
                                            const unsigned* d_offsets, const unsigned* d_zero_bases, int* d_err,
                                            uint8_t* d_mailbox, hipStream_t stream, unsigned start_nibble = 0);

}  // namespace bf
