// Launch interface of cls_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "cls_device.h"

namespace cls {

// Workgroups to launch for `n_reads` on a device with `n_cu` compute units.
uint32_t place_grid_blocks(uint32_t n_reads, uint32_t n_cu, const DbDev& db, bool stats);
// u32 words of per-wave child-counter workspace the launch needs (0 for trees
// whose nodes have at most two non-LEAF children).
uint32_t place_ws_words(const DbDev& db, uint32_t grid_blocks);
// Asynchronous on `stream`; all pointers are device pointers.
hipError_t launch_place(const DbDev& db, const PlaceParams& prm, const uint8_t* d_bases, const uint64_t* d_offsets,
                        uint32_t n_reads, cls_placement* d_out, cls_query_stats* d_stats, uint32_t* d_ws,
                        uint32_t grid_blocks, hipStream_t stream);

}  // namespace cls
