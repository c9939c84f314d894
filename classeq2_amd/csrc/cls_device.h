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
// FMT_LIST:  Slot{hash, loc}, loc = postings word offset (40 bits) << 24 | minimizer-bucket index (24 bits)
// FMT_SPLIT: TSlot{hash, set, bucket}: the k-mer's tip set (index into `sets`, never 0 for a k-mer of the index:
//            set == 0 marks an empty slot) and its minimizer bucket
struct Slot {
    uint64_t hash;
    uint64_t loc;
};
static_assert(sizeof(Slot) == 16, "Slot");
constexpr uint64_t SLOT_EMPTY = ~0ULL;
struct TSlot {
    uint64_t hash;
    uint32_t set;
    uint32_t bucket;  // minimizer-bucket index (24 bits) | specificity tier of the set << 30
};
// A direct-table entry and TSlot.bucket carry the set's specificity tier in their top two bits: the bit length of
// its tip count is <= 3 (0), <= 6 (1), <= 9 (2) or larger / no tips at all (3).  The locality keys prefer the
// k-mers that are specific to a small clade (order_key_kernel).
constexpr int TIER_SHIFT = 30;
// The minimizer-bucket filter (kmers_map.rs:295-297) asks whether the bucket of a table hit is keyed by the hash of the
// query k-mer's first m characters.  Bucket keys are distinct, so that is "bucket index == the index of the bucket keyed by
// MurmurHash3(first m characters)" -- a function of 2m bits, tabulated at load time (m = 4, the reference's default: 256
// entries): no second hash per k-mer and no read of the bucket's key in the kernels.  MZ_NO_BUCKET: no bucket has that key.
constexpr uint32_t MZ_TABLE_MAX_M = 8;
constexpr uint32_t MZ_NO_BUCKET = 0xFFFFFFFFu;
constexpr uint32_t SET_ID_MASK = (1u << TIER_SHIFT) - 1;
static_assert(sizeof(TSlot) == 16, "TSlot");
constexpr int LOC_BUCKET_BITS = 24;
constexpr uint64_t LOC_BUCKET_MASK = (1ULL << LOC_BUCKET_BITS) - 1;

// ---- postings (FMT_LIST) --------------------------------------------------------
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

// ---- tip sets + split trees (FMT_SPLIT) -----------------------------------------------
// Used when EVERY node set is closed under `parent` (every `cls build-db` output): a node set is then the union
// of the root->tip paths of its TIPS (members with no member below them), and `v in set` <=> some tip lies in
// [pre(v), pre(v)+size(v)).  k-mers with the same (tips, has-root) share ONE set: neighbouring k-mers of a
// conserved region, a k-mer and its reverse complement (C3: 459 k sets for 3.2 M k-mers).
//   sets[s]    16 bytes {x, vlo_lg, vhi_root, n_leaf}; s = 0 is the "no such k-mer" set {0, MAX, 0, 0}
//     x        record index of the set's root split (0: fewer than two tips)
//     vlo_lg   first tip | bit_length(n_tips) << 27       (0xFFFFFFFF: no tip below the root)
//     vhi_root last tip | has_root << 31
//     n_leaf   LEAF-kind ids of the node set (statistics: cls_query_stats.leaf_postings)
//   Sets are numbered in ascending (first tip, last tip): reads processed in locality order (below) touch
//   neighbouring set records.  The bit length feeds the locality keys (order_key_kernel).
//   splits[r]  16 bytes {tip[i-1], L, tip[i], R}: node i (1 <= i < n) of the Cartesian tree over
//     depth(LCA(tip[i-1], tip[i])) of a set with n ascending tips.  For the tips inside one clade's interval the
//     shallowest such LCA is where the clade's two children part them, so L / R (absolute record indices, 0 = none)
//     are the splits of the left / right part: descending one level costs ONE 8-byte read (the half for the side
//     taken) per set that has tips on both sides, and none otherwise.  Split nodes are stored per set, in DFS
//     pre-order, heavier child first, so the successive reads of one set tend to share a 64-byte line.  Record 0
//     is a dummy that decodes to "inactive" {0, 0, MAX, 0}.
// The k-mer level is a plain map k-mer -> set: `direct` (k <= DIRECT_MAX_K: 4 bytes per possible k-mer, indexed by
// its 2-bit code A=0 C=1 T=2 G=3 = (ascii >> 1) & 3, first base in the low bits) and/or the TSlot hash table.
// The direct table is only built when every index entry sits in the minimizer bucket of its own prefix (true for
// every `cls build-db` output), which makes the bucket filter of kmers_map.rs:295-297 a no-op.
struct SetRec {
    uint32_t x;
    uint32_t vlo_lg;
    uint32_t vhi_root;
    uint32_t n_leaf;
};
static_assert(sizeof(SetRec) == 16, "SetRec");
constexpr uint32_t DIRECT_MAX_K = 15;
// For small k the direct table is also kept DENORMALISED: 16-byte entries {root split x, first tip | lg << 27,
// last tip | has_root << 31, set id | tier << 30} = the set record with the set id in place of n_leaf, so that the
// wave-per-read kernels get a k-mer's whole initial descent state from ONE read instead of two dependent ones
// (table entry -> set record).  4^12 entries = 268 MB; the 4-byte table stays (locality keys, long reads).
constexpr uint32_t FAT_DIRECT_MAX_K = 12;
constexpr uint32_t DIRECT_TIP_BITS = 27;
constexpr uint32_t DIRECT_TIP_MASK = (1u << DIRECT_TIP_BITS) - 1;
struct TipRec {
    uint32_t tip_prev;  // last tip of the left part
    uint32_t l;         // split of the left part
    uint32_t tip;       // first tip of the right part
    uint32_t r;         // split of the right part
};
static_assert(sizeof(TipRec) == 16, "TipRec");
// MASK halves (DbDev.postings2, the wave-per-read kernels): a part (left or right of a split) whose
// tips span at most 32 pre-order rows is not walked through further split records: the half that leads into it is
//   {tip_prev | MASK_HALF, bits}   (left part; bit i = row (first tip of the part) + i is a tip; bit 0 is always set)
//   {tip      | MASK_HALF, bits}   (right part; bit i = row tip + i is a tip)
// and every later narrowing of that group is bit arithmetic on `bits`.  The last four or five levels of a descent run
// inside clades of at most 16 leaves: their split records are the ones hardly any other read asks for (every one a
// line from HBM), and with masks they are not read at all.
// (Measured and rejected: a third form for wider parts of at most three tips, {tip | flag, middle tip} -- on C3 the reads
// asked for stayed at 748 a read against 746 and the kernel went from 5.35 to 5.6 ms: parts that wide have more tips.)
constexpr uint32_t MASK_HALF = 0x80000000u;
constexpr uint32_t MASK_HALF_SPAN = 32;
constexpr uint32_t FAT_X_IS_BITS = 0x40000000u;  // direct16 entry, word 2: word 0 holds the set's tips as bits relative to its first tip

enum : uint32_t { FMT_LIST = 0, FMT_SPLIT = 1 };

struct DbDev {
    const DNode* nodes;
    const uint32_t* kids;       // per node row 4 words {start of the 3rd child, of the 4th, of the 5th, 0} in pre-order space (the end
                                // of the clade where it has fewer children): with DNode.pre / .split the boundaries of up to four
                                // non-LEAF children, so that a small polytomy is scored from two scalar loads
    const Slot* table;          // FMT_LIST: Slot; FMT_SPLIT: TSlot (same size, same probe sequence)
    const uint32_t* postings;   // FMT_LIST: u32 words; FMT_SPLIT: TipRec split records (16-byte units)
    const uint32_t* postings2;  // FMT_SPLIT: the same records, same numbering, with MASK halves (above), or nullptr (knob no_mask_halves)
    const uint64_t* bucket_key;
    const uint32_t* mz_bucket;  // 4^m_eff entries, or nullptr (m_eff > MZ_TABLE_MAX_M): the minimizer-bucket filter without hashing, below
    const uint32_t* direct;     // FMT_SPLIT, k <= DIRECT_MAX_K: 4^k set ids, or nullptr
    const uint32_t* direct16;   // k <= FAT_DIRECT_MAX_K: the same table with the set record inside the entry (below), or nullptr
    const SetRec* sets;         // FMT_SPLIT: tip sets
    const SetRec* sets2;        // the same with x = the set's tips as bits and FAT_X_IS_BITS in word 2 where it spans at most 32 rows
                                // (read by the wave-per-read kernels without the fat direct table), or nullptr
    uint64_t table_mask;
    uint32_t n_nodes;
    uint32_t n_buckets;
    uint32_t k;          // kSize
    uint32_t m_eff;      // min(mSize, kSize): chars().take(m), kmers_map.rs:11
    uint32_t max_nonleaf_arity;
    uint32_t format;     // FMT_*
    uint32_t addr32;     // splits, sets and direct table are each below 4 GiB: 32-bit byte offsets suffice
    uint32_t binary_tree; // every clade has exactly zero or two children
    uint32_t canonical;   // direct table: every k-mer and its reverse complement map to the same set (index built from both strands)
    uint32_t set_bits;    // FMT_SPLIT: bits that hold any set id (the low part of the locality sort key)
    uint32_t n_sets;      // FMT_SPLIT: entries of `sets` (incl. the dummy)
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
