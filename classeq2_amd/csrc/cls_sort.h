#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace cls {
size_t sort_temp_bytes(uint32_t n, int end_bit);
hipError_t sort_pairs(void* tmp, size_t tmp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in,
                      uint32_t* vals_out, uint32_t n, int begin_bit, int end_bit, hipStream_t stream);
}  // namespace cls
