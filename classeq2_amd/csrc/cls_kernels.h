// Launch interface of cls_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "cls_device.h"

namespace cls {

// LDS-tiled long-read kernel (cls_tile.hip): binary FMT_SPLIT index with a direct table.
struct TilePlan {
    uint32_t threads, lookups, bases, cap_kmers, grid, set_words, cap_entries;  // the configuration that gives a read a whole CU's LDS
    size_t smem;
    uint32_t half_grid, half_set_words, half_cap_entries;          // two workgroups per CU (half_grid == 0: not for reads this long)
    size_t half_smem;
    uint64_t scratch_words;                                        // global scratch of the resident workgroups
};

// Grid sizes + scratch layout of one placement batch.
struct PlacePlan {
    uint32_t grid[2];          // workgroups per wave-per-read class
    uint32_t grid_blk;         // workgroups of the workgroup-per-read class
    uint32_t grid_key;         // workgroups of the locality-key kernel
    bool ordered;              // class-0 reads are processed in locality order (fast path, large batches)
    uint64_t keys_off_words;   // sort keys / indices
    uint64_t sort_off_words;   // radix-sort scratch
    size_t sort_bytes;
    uint64_t child_off_words;  // offset of the child-counter area inside the workspace
    uint32_t grid_long;        // workgroups of the long-read class (0: none in this launch)
    uint32_t long_cap;         // k-mers per read its slices hold (0: reads beyond MAX_READ_KMERS are refused)
    uint32_t long_arity;       // child counters per slice (padded arity)
    uint64_t long_set;         // entries of the distinct-hit set (power of two)
    uint64_t long_stride_words;
    uint64_t long_off_words;
    uint32_t grid_tile;        // workgroups of the LDS-tiled long-read class (0: none in this launch)
    uint32_t tile_threads;     // 512 (two workgroups per CU) or 1024
    uint32_t tile_lookups;     // table lookups per read its LDS holds
    uint32_t tile_bases;       // bases per read its LDS holds
    uint32_t tile_cap_kmers;   // the same as a k-mer count (classification bound)
    size_t tile_smem;          // dynamic LDS of that kernel
    TilePlan tile;             // the whole plan of that class (cls_tile.hip)
    uint64_t tile_off_words;   // its scratch + the list of reads handed from the two-per-CU launch to the one-per-CU one
    uint64_t ws_bytes;         // device scratch the launch needs
};
// LDS-tiled long-read kernel (cls_tile.hip): launch interface
bool tile_usable(const DbDev& db);
TilePlan tile_plan(const DbDev& db, uint32_t want_kmers, uint32_t n_long, uint32_t n_cu);
std::string tile_kernel_name(const DbDev& db, bool stats, uint32_t threads);
// reads of `list` (device, *list_len of them) -> records; reads its code set cannot hold are appended to `spill_list`
void tile_launch(const DbDev& db, const PlaceParams& prm, const TilePlan& p, bool stats, const uint8_t* d_bases, const uint64_t* d_offsets,
                 const uint32_t* list, const uint32_t* list_len, cls_placement* d_out, cls_query_stats* d_stats, uint32_t* spill_list,
                 uint32_t* spill_len, uint32_t* scratch, uint32_t* big_list, uint32_t* big_len, hipStream_t stream);
// `long_cap`: k-mer capacity wanted for reads beyond MAX_READ_KMERS (0 = refuse them), `n_long`: how many
// such reads the batch may hold (bounds the number of workspace slices).
PlacePlan plan_place(const DbDev& db, uint32_t n_reads, uint32_t n_cu, bool stats, uint32_t long_cap, uint32_t n_long);
// Template instance of the class-0 placement kernel launch_place() picks for `db` (as rocprofv3 names it).
std::string dominant_kernel_name(const DbDev& db, bool stats, const PlacePlan* plan = nullptr);
// Asynchronous on `stream`; all pointers are device pointers; `d_ws` holds plan.ws_bytes.
// `ev_start`/`ev_stop` (may be null) are recorded around the dominant placement kernel (dominant_kernel_name).
hipError_t launch_place(const DbDev& db, const PlaceParams& prm, const PlacePlan& plan, const uint8_t* d_bases,
                        const uint64_t* d_offsets, uint32_t n_reads, cls_placement* d_out, cls_query_stats* d_stats,
                        uint32_t* d_ws, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop);

}  // namespace cls
