// Internal glue between cls_host.cpp, cls_build.cpp and cls_treeio.cpp (not part of the C-ABI).
#pragma once
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "cls_host.h"
#include "cls_json.h"

namespace cls_host {

// ---- data model (clade.rs:18-38, annotation.rs:5-34, tree.rs:9-52) ---------------------------------------
struct Clade {
    uint64_t id = 0;
    bool has_parent = false;
    uint64_t parent = 0;
    int kind = CLS_KIND_NODE;
    bool has_name = false, has_support = false, has_length = false, has_children = false;
    std::string name;
    double support = 0, length = 0;
    std::vector<Clade> children;
};

struct Tag { std::string name; bool is_int = false; uint64_t ival = 0; std::string sval; };
struct Annotation { uint32_t clade = 0; bool has_meta = false; std::vector<Tag> meta; };

std::string fmt_f64(double v);                      // Rust's ryu formatting (serde_json / serde_yaml floats)
std::string json_f64(double v);
void json_str(std::string& o, const std::string& s);
void yaml_str(std::string& o, const std::string& s, size_t indent);
void yaml_clade(std::string& o, const Clade& c, size_t ind, bool first_inline);
std::string read_file(const char* path);
std::string with_extension(const std::string& path, const char* ext);
Clade clade_from_json(const cls::JVal& j);

}  // namespace cls_host

struct cls_tree {
    // Tree header (tree.rs:9-52); empty for `--only-tree` exports
    bool has_header = false;
    std::string uuid, name, in_memory_size;
    bool has_in_memory_size = false;
    double min_branch_support = 0;
    cls_host::Clade root;
    bool has_annotations = false;
    std::vector<cls_host::Annotation> annotations;
    // flattened (BFS rows; children consecutive, in Clade.children order)
    std::vector<cls_node> rows;
    std::vector<const cls_host::Clade*> row_clade;
    std::map<uint64_t, uint32_t> first_row_of_id;  // get_node_by_id: first match in DFS order (clade.rs:95-109)
    // k-mer map (optional)
    bool has_kmers = false;
    uint64_t k_size = 0, m_size = 0;
    std::vector<uint64_t> bucket_key, bucket_kmer_off, kmer_hash, kmer_node_off, node_ids;
};

namespace cls_host {
void flatten(cls_tree* t);
// Fill `t` from a parsed database / tree document (JSON, or YAML parsed into the same DOM).
void tree_from_doc(const cls::JVal& doc, cls_tree* t);
}  // namespace cls_host

int cls_host_fail(int code, const std::string& msg);
// Leaves in DFS order with (name or nullptr, root->leaf id path) -- Clade::get_leaves_with_paths (clade.rs:127-156).
void cls_tree_visit_leaves(const cls_tree* t, const std::function<void(const char*, const std::vector<uint64_t>&)>& fn);
void cls_tree_set_kmers_map(cls_tree* t, uint64_t k, uint64_t m, std::vector<uint64_t>&& bucket_key,
                            std::vector<uint64_t>&& bucket_kmer_off, std::vector<uint64_t>&& kmer_hash,
                            std::vector<uint64_t>&& kmer_node_off, std::vector<uint64_t>&& node_ids);
