// Internal declarations shared by the C-ABI translation units.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace bf {

// Record an error message for the calling thread and return `code`.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

const char* last_error_text();

// canonical blob -> packed image (bf_layout.h); both sized by the caller
void pack_weights(const float* blob, float* packed, bool policy = false, int split = 0);   // split: 1 BF_PRECISION_F32_SPLIT, 2 BF_PRECISION_F32_SPLIT3


// canonical blob -> image of the 16x16x32 bf16 kernel (bf_layout_t16.h), t16::kImageBytes bytes
void pack_weights_t16(const float* blob, unsigned char* image, bool policy = false);

// Host restatement of one record's relevance mask and codes, shared by the host encoder.
// codes_out receives popcount(mask) codes in ascending square order (mover's frame).
uint64_t host_board_codes(const unsigned char* rec, uint8_t* codes_out, int* n_codes);

}  // namespace bf
