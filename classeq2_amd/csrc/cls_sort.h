#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace cls {
constexpr int ORDER_BIN_BITS = 12;                 // top bits of the locality key the reads are binned by
constexpr uint32_t ORDER_BINS = 1u << ORDER_BIN_BITS;
size_t order_temp_bytes(uint32_t n);
// idx_out = the reads 0 .. n-1 ordered by the top ORDER_BIN_BITS of their `key_bits`-bit keys (all-ones key: last);
// `tmp` holds order_temp_bytes(n); asynchronous on `stream`.
hipError_t order_reads(void* tmp, const uint64_t* keys, uint32_t* idx_out, uint32_t n, int key_bits, hipStream_t stream);
}  // namespace cls
