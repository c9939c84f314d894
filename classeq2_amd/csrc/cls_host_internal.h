// Internal glue between cls_host.cpp and cls_build.cpp (not part of the C-ABI).
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "cls_host.h"

int cls_host_fail(int code, const std::string& msg);
// Leaves in DFS order with (name or nullptr, root->leaf id path) -- Clade::get_leaves_with_paths (clade.rs:127-156).
void cls_tree_visit_leaves(const cls_tree* t, const std::function<void(const char*, const std::vector<uint64_t>&)>& fn);
void cls_tree_set_kmers_map(cls_tree* t, uint64_t k, uint64_t m, std::vector<uint64_t>&& bucket_key,
                            std::vector<uint64_t>&& bucket_kmer_off, std::vector<uint64_t>&& kmer_hash,
                            std::vector<uint64_t>&& kmer_node_off, std::vector<uint64_t>&& node_ids);
