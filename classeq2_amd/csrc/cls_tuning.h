// Experiment knobs of the library, in one place.  None changes a result.  The product never reads the environment
// on its own: a host sets a knob through cls_set_tuning() (include/cls_place.h), and tools/ + bench.py may call
// cls_tuning_from_env() once to take them from CLS_* variables (A/B runs on the GPU box).
#pragma once

namespace cls {

struct Tuning {
    int no_fast = 0;              // CLS_NO_FAST: generic kernels only
    int no_order = 0;             // CLS_NO_ORDER: no locality order
    int force_list = 0;           // CLS_FORCE_LIST: sorted-list postings even when every node set is closed
    int no_mask_halves = 0;       // CLS_NO_MASK_HALVES: no second copy of the split records with narrow parts as bit masks (set before cls_db_create)
    int no_fat_direct = 0;        // CLS_NO_FAT_DIRECT: no denormalised 16-byte direct table for k <= 12 (set before cls_db_create)
    int no_tile = 0;              // CLS_NO_TILE: long reads through the workspace kernel only
    int tile_pass_codes = 0;      // CLS_TILE_PASS_CODES: lookups per pass of the LDS-tiled kernel's code set (0: all of a read's in one pass)
    int tile_set_words = 0;       // CLS_TILE_SET_WORDS: words of that set (0: two per lookup; a small set with one pass overflows on a
                                  // long read, which then takes the spill path: tests)
    int tile_min_kmers = 0;       // CLS_TILE_MIN_KMERS: reads with more k-mers than this take the LDS-tiled kernel rather than the workgroup-per-read one (0: only reads that one cannot hold)
    int tile_deal = 64;           // CLS_TILE_DEAL: consecutive reads of the locality order an XCD takes at a time (LDS-tiled classes)
    int no_tile_order = 0;        // CLS_NO_TILE_ORDER: the LDS-tiled classes' reads in batch order, not in locality order
    int tile_one_per_cu = 0;      // CLS_TILE_ONE_PER_CU: the LDS-tiled kernel as ONE 1024-thread workgroup a CU even where two 512-thread ones fit
    int time_class = 0;           // CLS_TIME_CLASS: 2 = cls_db_kernel_time / cls_db_kernel_name follow the workgroup-per-read kernel (reads of 513..4096 + k - 1 bases)
    int blocks_per_cu = 0;        // CLS_BLOCKS_PER_CU: grid of the wave-per-read kernels (0: what is resident)
    int key_blocks_per_cu = 0;    // CLS_KEY_BLOCKS_PER_CU
    int long_blocks_per_cu = 2;   // CLS_LONG_BLOCKS_PER_CU: workspace long-read kernel
    int order_mode = 0;           // CLS_ORDER_MODE: 0 = {leaf neighbourhood, MinHash}, 1 = {MinHash of all k-mers, leaf neighbourhood}
    int order_windows = 64;       // CLS_ORDER_WINDOWS: windows of a read that make its locality key
    int order_both_strands = 0;   // CLS_ORDER_BOTH_STRANDS
    int order_block_shift = 2;    // CLS_ORDER_BLOCK_SHIFT
    int order_sample_shift = 32;  // CLS_ORDER_SAMPLE_SHIFT
    int profile_stop = 0;         // CLS_PROFILE_STOP (needs a -DCLS_PROFILE_HOOKS build)
    int timing = 0;               // CLS_TIMING: phase times of cls_place_sequences on stderr
};

Tuning& tuning();

}  // namespace cls
