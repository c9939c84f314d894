// Host-side encoded database (see cls_db.cpp) before upload.
#pragma once
#include <string>
#include <vector>

#include "cls_device.h"

namespace cls {

constexpr uint64_t MAX_K = 1024;

struct EncodedDb {
    std::vector<DNode> nodes;
    std::vector<Slot> table;
    std::vector<uint32_t> postings;   // FMT_LIST words, or FMT_SPLIT records (4 words each)
    uint32_t format = FMT_LIST;
    bool strictly_binary = false;
    bool canonical = false;           // direct table symmetric under reverse complement
    std::vector<uint64_t> bucket_key;
    std::vector<FSlot> ftable;        // FMT_SPLIT without a direct table: hash table slots that carry the descent state
    std::vector<uint32_t> direct;     // 4^k x {record offset, meta} (FMT_SPLIT, k <= DIRECT_MAX_K) or empty
    uint32_t k = 0, m = 0, m_eff = 0;
    uint32_t max_depth = 0, max_nonleaf_arity = 0;
    uint64_t n_kmers = 0, n_closed = 0;
    uint64_t n_sets = 0;              // FMT_SPLIT: distinct tip lists (k-mers with the same one share a split tree)
    bool root_has_children = false;
};

// Validates `d` and fills `E`; returns CLS_OK or a CLS_E_* code with `err` set.
int encode_db(const cls_db_desc* d, EncodedDb& E, std::string& err);

}  // namespace cls
