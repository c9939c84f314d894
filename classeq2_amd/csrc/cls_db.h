// Host-side encoded database (see cls_db.cpp) before upload.
#pragma once
#include <stdlib.h>
#include <sys/mman.h>

#include <new>
#include <string>
#include <vector>

#include "cls_device.h"

namespace cls {

constexpr uint64_t MAX_K = 1024;

// Allocator of the encoder's big tables: 2 MiB-aligned and advised to transparent huge pages before the first touch.
// The builds probe gigabyte hash tables at random (4^k probes for the direct table): on 4 KiB pages every probe is a
// TLB miss as well.
template <class T>
struct HugeAlloc {
    using value_type = T;
    HugeAlloc() = default;
    template <class U> HugeAlloc(const HugeAlloc<U>&) {}
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < (8u << 20)) { void* p = malloc(bytes ? bytes : 1); if (!p) throw std::bad_alloc(); return static_cast<T*>(p); }
        const size_t huge = 2u << 20, padded = (bytes + huge - 1) & ~(huge - 1);
        void* p = aligned_alloc(huge, padded);
        if (!p) throw std::bad_alloc();
        (void)madvise(p, padded, MADV_HUGEPAGE);
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t) { free(p); }
    template <class U> bool operator==(const HugeAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const HugeAlloc<U>&) const { return false; }
};
template <class T> using HugeVec = std::vector<T, HugeAlloc<T>>;

struct EncodedDb {
    std::vector<DNode> nodes;
    std::vector<uint32_t> kids;       // 4 words per node: where its 3rd / 4th / 5th child starts (cls_device.h)
    HugeVec<Slot> table;              // FMT_LIST: Slot; FMT_SPLIT: TSlot (same size)
    HugeVec<uint32_t> postings2;      // FMT_SPLIT: the split records again, narrow parts as bit masks (cls_device.h)
    HugeVec<uint32_t> postings;       // FMT_LIST words, or FMT_SPLIT split records (4 words each)
    uint32_t format = FMT_LIST;
    bool strictly_binary = false;
    bool canonical = false;           // direct table symmetric under reverse complement
    std::vector<uint64_t> bucket_key;
    std::vector<uint32_t> mz_bucket;  // m_eff <= MZ_TABLE_MAX_M: per 2-bit code of a k-mer's first m characters the bucket keyed by their hash (cls_device.h)
    HugeVec<SetRec> sets;             // FMT_SPLIT: tip sets (entry 0 = "no such k-mer")
    HugeVec<SetRec> sets2;            // the same for the wave-per-read kernels: a set that spans at most 32 rows carries its bits (MASK halves)
    HugeVec<uint32_t> direct;         // 4^k set ids (FMT_SPLIT, k <= DIRECT_MAX_K) or empty
    HugeVec<uint32_t> direct16;       // 4^k x 4 words: set record + set id (k <= FAT_DIRECT_MAX_K) or empty
    uint32_t k = 0, m = 0, m_eff = 0;
    uint32_t max_depth = 0, max_nonleaf_arity = 0;
    uint64_t n_kmers = 0, n_closed = 0;
    uint64_t n_sets = 0;              // FMT_SPLIT: distinct tip lists (k-mers with the same one share a split tree)
    bool root_has_children = false;
};

// Validates `d` and fills `E`; returns CLS_OK or a CLS_E_* code with `err` set.
int encode_db(const cls_db_desc* d, EncodedDb& E, std::string& err);

}  // namespace cls
