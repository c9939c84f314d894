// Launch interface of cls_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "cls_device.h"

namespace cls {

// LDS-tiled long-read kernel (cls_tile.hip): binary FMT_SPLIT index with a direct table.
// One launch of the LDS-tiled kernel (cls_tile.hip): `threads` x `grid`, a read of up to `cap_kmers` k-mers (`lookups` table
// lookups, `bases` bases) per workgroup.
struct TileCfg {
    uint32_t threads, grid, lookups, bases, cap_kmers, set_words, cap_entries;
    size_t smem;
    uint64_t scratch_off;   // the resident workgroups' slots of the global scratch (u32 words into the tile scratch)
};
// The reads of the LDS-tiled kernel are binned by k-mer count into up to three SHARED launches (8, 4, 2 workgroups per CU: 128,
// 256, 512 threads) and the WHOLE one (a read has the CU's LDS to itself: 1024 threads), which also takes the reads whose
// entries overflow a shared launch's LDS.
constexpr int TILE_MAX_SUB = 3;
struct TilePlan {
    TileCfg sub[TILE_MAX_SUB];
    uint32_t n_sub;
    TileCfg whole;
    uint64_t scratch_words;
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
    uint32_t max_kmers;        // the longest read the launch is provisioned for: the classes beyond it are not launched
    bool tiled;                // the launch has the LDS-tiled classes (cls_tile.hip)
    bool time_tile;            // ... and its kernel is the one that is timed (a handle provisioned for reads beyond MAX_READ_KMERS, or CLS_TIME_CLASS=2)
    TilePlan tile;
    uint32_t tile_from;        // reads with more k-mers than this (and at most tile.whole.cap_kmers) are the LDS-tiled kernel's
    uint32_t tile_name_threads; // the configuration that takes the longest read the launch is provisioned for (cls_db_kernel_name)
    uint64_t tile_off_words;   // its scratch
    uint64_t ws_bytes;         // device scratch the launch needs
};
// LDS-tiled long-read kernel (cls_tile.hip): launch interface
bool tile_usable(const DbDev& db);
TilePlan tile_plan(const DbDev& db, uint32_t from_kmers, uint32_t max_kmers, uint32_t n_reads, uint32_t n_cu);
std::string tile_kernel_name(const DbDev& db, bool stats, uint32_t threads);
// reads of the shared launches' lists and of `big_list` (device) -> records; reads the kernel cannot hold (its code set, its
// entries) are appended to `spill_list`
void tile_launch(const DbDev& db, const PlaceParams& prm, const TilePlan& p, bool stats, const uint8_t* d_bases, const uint64_t* d_offsets,
                 const uint32_t* const* sub_lists, const uint32_t* const* sub_lens, uint32_t* big_list, uint32_t* big_len,
                 cls_placement* d_out, cls_query_stats* d_stats, uint32_t* spill_list, uint32_t* spill_len, uint32_t* scratch, bool ordered, hipStream_t stream);
// `n_long` != 0: `long_cap` is the k-mer count of the longest read to provision for (classes beyond it are not launched, a
// longer read may be refused) and `n_long` how many reads beyond the wave-per-read kernels the batch may hold (bounds the grids
// and the workspace slices); `n_long` == 0: reads of up to MAX_READ_KMERS k-mers.
PlacePlan plan_place(const DbDev& db, uint32_t n_reads, uint32_t n_cu, bool stats, uint32_t long_cap, uint32_t n_long);
// Template instance of the class-0 placement kernel launch_place() picks for `db` (as rocprofv3 names it).
std::string dominant_kernel_name(const DbDev& db, bool stats, const PlacePlan* plan = nullptr);
// Asynchronous on `stream`; all pointers are device pointers; `d_ws` holds plan.ws_bytes.
// `ev_start`/`ev_stop` (may be null) are recorded around the dominant placement kernel (dominant_kernel_name).
hipError_t launch_place(const DbDev& db, const PlaceParams& prm, const PlacePlan& plan, const uint8_t* d_bases,
                        const uint64_t* d_offsets, uint32_t n_reads, cls_placement* d_out, cls_query_stats* d_stats,
                        uint32_t* d_ws, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop);

}  // namespace cls
