/*
 * cls_host.h -- host-side mirror of the reference's batch driver (part of libclsplace.so).
 *
 * The reference's `place_sequences` (core/src/use_cases/place_sequences/mod.rs:43-270) reads a
 * multi-FASTA, places every record and appends one YAML document or JSON line per record to
 * `<out>.yaml|.jsonl`, per-query errors to `<out>.error`.  These entry points do the same above the
 * GPU path, for callers that do not keep the Rust front-end (the C++ `cls-place` tool in csrc/ is one).
 * Tree / index / annotations come from the reference's own file formats:
 *   - database: the JSON export of `cls convert database -f json` (ports/cli/src/cmds/convert.rs:161-205);
 *   - annotations: the YAML list of `Annotation` (core/src/domain/dtos/annotation.rs:25-34).
 */
#ifndef CLS_HOST_H
#define CLS_HOST_H

#include "cls_place.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cls_tree cls_tree; /* Tree: root clade (+ names, supports, lengths), annotations, k-mer map */

enum { CLS_FORMAT_YAML = 0, CLS_FORMAT_JSONL = 1 }; /* OutputFormat, core/src/domain/dtos/output_format.rs:3-11 */

/* Parse a database (or `--only-tree`) JSON export.  Replaces load_database
 * (ports/lib/src/functions/load_database.rs:9-53) for the JSON form. */
int cls_tree_load_json(const char* path, cls_tree** out);
void cls_tree_free(cls_tree* t);
/* Tree::init_from_file (core/src/domain/dtos/tree.rs:164-230): parse a rooted Newick tree (.nwk/.newick/.tree),
 * node ids = pre-order of the text (what the phylotree arena yields), internal labels = support values, then
 * sanitize (tree.rs:252-291: a clade whose support is below `min_branch_support` hands its children to its
 * parent) and fix_parent_ids.  Tree.name = the file name, Tree.id = UUID v3 (DNS namespace) of it. */
int cls_tree_init_from_file(const char* tree_path, double min_branch_support, cls_tree** out);
/* The same from text already in memory (`tree_name` NULL: "UnnamedTree"). */
int cls_tree_from_newick(const char* newick_text, const char* tree_name, double min_branch_support, cls_tree** out);
/* load_database (ports/lib/src/functions/load_database.rs:9-53): a `.cls` file (zstd-compressed YAML, needs the
 * system's libzstd.so.1 at run time), plain YAML, or the JSON export; a whole database or an `--only-tree` file;
 * also the k-mer-keyed map of databases written before the minimizer buckets existed. */
int cls_tree_load(const char* path, cls_tree** out);
/* `cls convert database -f {zstd|yaml|json} [--only-tree]` (ports/cli/src/cmds/convert.rs:161-205) and the file
 * `cls build-db` writes (ports/cli/src/cmds/build_db.rs:70-76): serde_yaml / serde_json::to_writer_pretty of the
 * Tree (or of its root clade).  cls_tree_save forces the reference's extensions (.cls / .cls.yaml / .cls.json). */
#define CLS_DB_FORMAT_ZSTD 0
#define CLS_DB_FORMAT_YAML 1
#define CLS_DB_FORMAT_JSON 2
int cls_tree_serialize(const cls_tree* t, int format, int only_tree, char** out, size_t* out_len); /* cls_host_free(*out) */
int cls_tree_save(const cls_tree* t, const char* path, int format, int only_tree);
/* `-a/--annotations-file-path` of `cls place` (ports/cli/src/cmds/place_sequences.rs:137-144). */
int cls_tree_set_annotations_yaml(cls_tree* t, const char* path);
/* `cls build-db` on an already parsed tree: map_kmers_to_tree (core/src/use_cases/build_database/mod.rs:26-181).
 * `msa_text` is the multi-FASTA whose headers name the tree's leaves.  Replaces the tree's k-mer map. */
#define CLS_BUILD_REFERENCE_HEADER_SHIFT 1u /* file record i's k-mers under header i+1, never index the last record
                                             * (what the reference does, build_database/mod.rs:93-116) */
#define CLS_BUILD_FORWARD_ONLY 2u           /* forward k-mers only (builds older than the reverse-complement change) */
int cls_tree_build_kmers_map(cls_tree* t, const char* msa_text, size_t msa_len, uint64_t k_size, uint64_t m_size,
                             uint32_t flags);
/* Borrowed flat view for cls_db_create(); valid while `t` lives. */
int cls_tree_desc(const cls_tree* t, cls_db_desc* d);

/* PlacementResponse serialisation (mod.rs:170-248, placement_response.rs:30-94): one YAML document
 * ("---\n" + mapping) or one JSON line per non-error record, in input order; the messages of error
 * records concatenated as the reference appends them to `<out>.error`.  Buffers are malloc'ed; release
 * them with cls_host_free().  `err_text` may be NULL. */
int cls_serialize_results(const cls_tree* t, const char* headers, const uint64_t* header_off, uint32_t n,
                          const cls_placement* recs, int format, char** out_text, size_t* out_len,
                          char** err_text, size_t* err_len);
void cls_host_free(void* p);

/* The whole use-case: FASTA (`query_path`, "-" = stdin) -> placements on the GPU -> result + error files.
 * `out_file` gets its extension replaced like PathBuf::set_extension (mod.rs:76-81); `overwrite` is
 * `--force-overwrite`.  Returns the number of records read and the UCPLACE0001->0002 wall time. */
int cls_place_sequences(cls_db* db, const cls_tree* t, const char* query_path, const char* out_file,
                        const cls_params* params, int overwrite, int format, uint32_t* n_placed, double* seconds);

const char* cls_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
