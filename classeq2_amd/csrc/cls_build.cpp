// Host-side index builder: the reference's `map_kmers_to_tree`
// (core/src/use_cases/build_database/mod.rs:26-181) on an already parsed tree.
//   every MSA record -> forward + reverse-complement k-mers (kmers_map.rs:375-398)
//   -> (minimizer bucket = murmur3 of the first m chars, k-mer hash) -> union of the root->leaf id paths
//      of the leaves it is filed under (clade.rs:127-156, kmers_map.rs:125-149).
// CLS_BUILD_REFERENCE_HEADER_SHIFT reproduces the reference's record/header skew
// (build_database/mod.rs:93-116: a record's k-mers are sent with the NEXT record's header, the first
// header receives none, the last record is never indexed) -- what every database written by the
// reference actually contains, and what the golden fixture pins (tests/test_builder.py).
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "cls_host.h"
#include "cls_host_internal.h"
#include "cls_murmur.h"

namespace {
struct Rec {
    uint64_t bkey, hash;
    uint32_t leaf;  // index into the leaf table
};
}  // namespace

extern "C" int cls_tree_build_kmers_map(cls_tree* t, const char* msa_text, size_t msa_len, uint64_t k_size, uint64_t m_size,
                                        uint32_t flags) {
    if (!t || (!msa_text && msa_len) || k_size == 0) return cls_host_fail(CLS_E_INVALID_ARG, "cls_tree_build_kmers_map: invalid argument");
    try {
        // leaves in get_leaves_with_paths order (DFS), with their root->leaf id paths
        struct Leaf { std::string name; std::vector<uint64_t> path; };
        std::vector<Leaf> leaves;
        std::map<std::string, uint32_t> by_name;  // first match wins, like `tree_leaves.iter().find(..)` (mod.rs:139-141)
        cls_tree_visit_leaves(t, [&](const char* name, const std::vector<uint64_t>& path) {
            by_name.emplace(name ? name : "", (uint32_t)leaves.size());
            leaves.push_back({name ? name : "", path});
        });
        cls_fasta fa;
        int rc = cls_fasta_parse(msa_text, msa_len, &fa);
        if (rc != CLS_OK) return cls_host_fail(rc, "cls_tree_build_kmers_map: cannot parse the MSA");
        const bool shift = flags & CLS_BUILD_REFERENCE_HEADER_SHIFT, fwd_only = flags & CLS_BUILD_FORWARD_ONLY;
        const uint32_t K = (uint32_t)k_size, M = (uint32_t)std::min<uint64_t>(m_size, k_size);
        std::vector<Rec> recs;
        std::string err;
        for (uint32_t i = 0; i < fa.n && err.empty(); ++i) {
            // which header are record i's k-mers filed under?
            uint32_t hi = i;
            if (shift) { if (i + 1 >= fa.n) break; hi = i + 1; }  // the last record is never indexed
            const std::string header(fa.headers + fa.header_off[hi], fa.headers + fa.header_off[hi + 1]);
            auto it = by_name.find(header);
            if (it == by_name.end()) { err = "The sequence header does not match any tree leaf: " + header; break; }
            const char* s = fa.bases + fa.base_off[i];
            const uint64_t L = fa.base_off[i + 1] - fa.base_off[i];
            if (L < K) continue;  // build_kmer_from_string: shorter than k -> []
            for (int strand = 0; strand < (fwd_only ? 1 : 2); ++strand) {
                for (uint64_t p = 0; p + K <= L; ++p) {
                    auto get = [&](uint32_t j) -> uint8_t {
                        if (!strand) return (uint8_t)s[p + j];
                        const char c = s[L - 1 - p - j];
                        return (uint8_t)(c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A');
                    };
                    recs.push_back({m_size == 0 ? 0ull : cls::murmur3_h1(get, M), cls::murmur3_h1(get, K), it->second});
                }
            }
        }
        if (shift && fa.n) {  // the first header still has to name a leaf (it receives an empty k-mer list)
            const std::string h0(fa.headers + fa.header_off[0], fa.headers + fa.header_off[1]);
            if (err.empty() && !by_name.count(h0)) err = "The sequence header does not match any tree leaf: " + h0;
        }
        cls_fasta_free(&fa);
        if (!err.empty()) return cls_host_fail(CLS_E_BAD_DB, err);
        std::sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) {
            if (a.bkey != b.bkey) return a.bkey < b.bkey;
            if (a.hash != b.hash) return a.hash < b.hash;
            return a.leaf < b.leaf;
        });
        std::vector<uint64_t> bucket_key, bucket_kmer_off{0}, kmer_hash, kmer_node_off{0}, node_ids;
        std::vector<uint64_t> set;
        for (size_t i = 0; i < recs.size();) {
            size_t j = i;
            set.clear();
            uint32_t prev = UINT32_MAX;
            for (; j < recs.size() && recs[j].bkey == recs[i].bkey && recs[j].hash == recs[i].hash; ++j) {
                if (recs[j].leaf == prev) continue;
                prev = recs[j].leaf;
                set.insert(set.end(), leaves[prev].path.begin(), leaves[prev].path.end());
            }
            std::sort(set.begin(), set.end());
            set.erase(std::unique(set.begin(), set.end()), set.end());
            if (bucket_key.empty() || bucket_key.back() != recs[i].bkey) {
                if (!bucket_key.empty()) bucket_kmer_off.push_back(kmer_hash.size());
                bucket_key.push_back(recs[i].bkey);
            }
            kmer_hash.push_back(recs[i].hash);
            node_ids.insert(node_ids.end(), set.begin(), set.end());
            kmer_node_off.push_back(node_ids.size());
            i = j;
        }
        if (!bucket_key.empty()) bucket_kmer_off.push_back(kmer_hash.size());
        cls_tree_set_kmers_map(t, k_size, m_size, std::move(bucket_key), std::move(bucket_kmer_off), std::move(kmer_hash),
                               std::move(kmer_node_off), std::move(node_ids));
        return CLS_OK;
    } catch (const std::exception& e) {
        return cls_host_fail(CLS_E_INTERNAL, std::string("cls_tree_build_kmers_map: ") + e.what());
    } catch (...) {
        return cls_host_fail(CLS_E_INTERNAL, "cls_tree_build_kmers_map: unknown exception");
    }
}
