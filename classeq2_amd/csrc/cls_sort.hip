// Key/value radix sort used by the locality ordering of the fast path (rocPRIM via hipCUB).
#include "cls_sort.h"

#include <hipcub/hipcub.hpp>

namespace cls {

size_t sort_temp_bytes(uint32_t n, int end_bit) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                             (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, 0, end_bit);
    return bytes;
}

hipError_t sort_pairs(void* tmp, size_t tmp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in,
                      uint32_t* vals_out, uint32_t n, int begin_bit, int end_bit, hipStream_t stream) {
    return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, begin_bit, end_bit, stream);
}

}  // namespace cls
