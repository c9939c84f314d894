// Device-side layout of the placement index (what lives in HBM) and the
// host<->kernel contract.  See DESIGN.md "Data layout in HBM".
#pragma once
#include <stdint.h>

#include "cls_place.h"

namespace cls {

// ---- tree -------------------------------------------------------------------
// One 32-byte row per clade, in the engine's own BFS order in which every
// node's non-LEAF children come first and are consecutive rows.  `pre` is the
// node's index in the DFS pre-order that visits children in that same order,
// `size` its subtree size, so subtree(v) == pre interval [pre, pre+size) and
// the non-LEAF children of v tile [pre+1, ...) back to back.
struct DNode {
    uint32_t pre;
    uint32_t size;
    uint32_t first_child;   // row of the first (non-LEAF first) child
    uint32_t n_nonleaf;     // children whose kind != LEAF (clade.rs:166-172)
    uint64_t id;            // Clade.id
    uint32_t split;         // pre + size of the FIRST child row = where the second child starts (0 if childless)
    uint32_t flags;         // bit0: has_children (children: Some(..)); bits 8..31: min(n_children, 2^24-1)
};
static_assert(sizeof(DNode) == 32, "DNode");

// ---- k-mer table --------------------------------------------------------------
// Open addressing, linear probing, power-of-two capacity, load <= 0.5.
// key  = murmur3_x64_128(kmer,0).0 (kmers_map.rs:157-159)
// loc  = postings word offset (40 bits) << 24 | minimizer-bucket index (24 bits)
struct Slot {
    uint64_t hash;
    uint64_t loc;
};
static_assert(sizeof(Slot) == 16, "Slot");
constexpr uint64_t SLOT_EMPTY = ~0ULL;
// The same table with the k-mer's initial descent state inside the slot (FMT_SPLIT without a direct table, i.e.
// k > 15): a hit needs no header read.  Same capacity, same probe sequence as `Slot`.
//   off      header record of the k-mer (0 = empty slot)
//   x        root split
//   vlo_lg   first tip | bit_length(n_tips) << 27   (0xFFFFFFFF: no tip below the root)
//   vhi_root last tip | has_root << 31
struct FSlot {
    uint64_t hash;
    uint32_t off;
    uint32_t x;
    uint32_t vlo_lg;
    uint32_t vhi_root;
    uint32_t bucket;
    uint32_t pad_;
};
static_assert(sizeof(FSlot) == 32, "FSlot");
constexpr int LOC_BUCKET_BITS = 24;
constexpr uint64_t LOC_BUCKET_MASK = (1ULL << LOC_BUCKET_BITS) - 1;

// ---- postings -----------------------------------------------------------------
// Per k-mer, at word offset `off` of the u32 postings array:
//   w0 = n_elems | POST_HAS_ROOT | POST_CLOSED
//   w1 = number of LEAF-kind ids in the original node set (statistics only)
//   n_elems ascending pre-order indices, the root (pre 0) stripped:
//     POST_CLOSED : the node set was closed under `parent` -> only its TIPS
//                   (members with no member below them) are stored;
//                   v in set  <=>  some tip in [pre(v), pre(v)+size(v))
//     otherwise   : every member is stored; v in set <=> pre(v) stored
constexpr uint32_t POST_HAS_ROOT = 1u << 31;
constexpr uint32_t POST_CLOSED = 1u << 30;
constexpr uint32_t POST_LEN_MASK = (1u << 30) - 1;
constexpr uint32_t POST_HEADER_WORDS = 2;

// ---- postings, "split-tree" form (FMT_SPLIT) ----------------------------------
// Used when EVERY node set is closed under `parent` (every `cls build-db` output).  16-byte records; a k-mer with n tips owns 2 header
// records + (n-1) split nodes:
//   header 0 : {n | POST_HAS_ROOT | POST_CLOSED, root split (record index, 0 = none), first tip, last tip}
//   header 1 : {n_leaf_ids (statistics), hash lo, hash hi, minimizer-bucket index}
//   split i  : {tip[i-1], L, tip[i], R}            (1 <= i < n): one 8-byte half per direction
// tip[] = ascending pre-order indices.  Split i is the node of the Cartesian tree
// over depth(LCA(tip[i-1], tip[i])): for the tips inside one clade's interval the
// shallowest such LCA is where the clade's two children part them, so L / R
// (absolute record indices, 0 = none) are the splits of the left / right part.
// Descending one level costs ONE 16-byte read per k-mer that has tips on both
// sides, and none otherwise.  Split nodes are stored in DFS pre-order, heavier
// child first, so the successive reads of one k-mer tend to share a 64-byte line.
// Record 0/1 of the array are a dummy header pair {0, 0, 0xFFFFFFFF, 0}: "no k-mer"
// reads land there and decode to an inactive state without a branch.
constexpr uint32_t SPLIT_HEADER_RECS = 2;
constexpr uint32_t SPLIT_FIRST_REC = 2;

// ---- direct k-mer table (k <= DIRECT_MAX_K, FMT_SPLIT) ---------------------------------------
// For small k every possible k-mer is enumerated once at cls_db_create(): its 2-bit
// code (A=0 C=1 T=2 G=3 = (ascii >> 1) & 3, first base in the low bits) indexes a table
// of 16-byte entries that hold the k-mer's whole initial descent state, so the query
// side needs neither MurmurHash, nor a probe loop, nor the header record:
//   {record offset of the header (0 = not in the index), root split,
//    first tip | bit_length(n_tips) << 27   (0xFFFFFFFF: absent / no tip below the root),
//    last tip  | has_root << 31}
// The bit length feeds the locality ordering of cls_kernels.hip (order_key_kernel).
// Only built when every index entry sits in the minimizer bucket of its own prefix (true
// for every `cls build-db` output), which makes the bucket filter of kmers_map.rs:295-297
// a no-op, and when pre-order indices fit DIRECT_TIP_BITS.
constexpr uint32_t DIRECT_MAX_K = 15;
constexpr uint32_t DIRECT_TIP_BITS = 27;
constexpr uint32_t DIRECT_TIP_MASK = (1u << DIRECT_TIP_BITS) - 1;
struct TipRec {
    uint32_t tip_prev;  // last tip of the left part
    uint32_t l;         // split of the left part
    uint32_t tip;       // first tip of the right part
    uint32_t r;         // split of the right part
};
static_assert(sizeof(TipRec) == 16, "TipRec");

enum : uint32_t { FMT_LIST = 0, FMT_SPLIT = 1 };

struct DbDev {
    const DNode* nodes;
    const Slot* table;
    const uint32_t* postings;   // FMT_LIST: u32 words; FMT_SPLIT: TipRec records (16-byte units)
    const uint64_t* bucket_key;
    const uint32_t* direct;     // 4^k entries of 4 words (above) or nullptr
    const FSlot* ftable;        // state-carrying hash table (above) or nullptr
    uint64_t table_mask;
    uint32_t n_nodes;
    uint32_t n_buckets;
    uint32_t k;          // kSize
    uint32_t m_eff;      // min(mSize, kSize): chars().take(m), kmers_map.rs:11
    uint32_t max_nonleaf_arity;
    uint32_t format;     // FMT_*; Slot.loc offsets are in words (LIST) or records (SPLIT)
    uint32_t addr32;     // postings and direct table are both below 4 GiB: 32-bit byte offsets suffice
    uint32_t binary_tree; // every clade has exactly zero or two children
    uint32_t canonical;   // direct table: every k-mer and its reverse complement hold the same entry state (index built from both strands)
    uint32_t hdr_bits;    // FMT_SPLIT: bits that hold any k-mer's header record offset (the sort key of the locality order)
};

// Resolved Option<> arguments (place_sequence.rs:64-75)
struct PlaceParams {
    int32_t max_iterations;
    uint32_t remove_intersection;
    double min_match_coverage;
};

// Per-read k-mer capacity of the widest kernel (reads beyond it are reported
// as CLS_ERR_READ_TOO_LONG).
constexpr uint32_t MAX_READ_KMERS = 64 * 8 * 16;

}  // namespace cls
