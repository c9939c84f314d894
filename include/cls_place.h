/*
 * cls_place.h -- C-ABI of the MI355X placement engine (libclsplace.so).
 *
 * Drop-in boundary for ONE path of LepistaBioinformatics/classeq2: the
 * `core::use_cases::place_sequences` use-case.  The reference has no FFI of
 * its own; the seam a maintainer binds is the pure function
 *
 *     place_sequence(header, sequence, &Tree, Option<i32>, Option<f64>,
 *                    Option<bool>) -> Result<PlacementStatus, MappedErrors>
 *     (core/src/use_cases/place_sequences/place_sequence.rs:42-50)
 *
 * called once per query by the batch driver
 * (core/src/use_cases/place_sequences/mod.rs:123-159).  The entry points
 * below replace that call for a whole batch; everything around it (FASTA
 * reader, annotations, YAML/JSONL writer) can stay in the Rust caller, see
 * INTEGRATION.md for the Rust `extern "C"` shim.
 *
 * Plain pointers and sizes only; no C++ or torch types; nothing unwinds
 * across this boundary.  All functions return 0 on success, a negative
 * CLS_E_* code otherwise; cls_last_error() gives the thread-local message.
 */
#ifndef CLS_PLACE_H
#define CLS_PLACE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLS_ABI_VERSION 2u  /* 1 is still accepted: cls_db_desc without the trailing node_set_kind */

/* ---- error codes ------------------------------------------------------- */
#define CLS_OK 0
#define CLS_E_INVALID_ARG (-1)   /* null pointer, bad sizes, bad abi_version     */
#define CLS_E_BAD_TREE (-2)      /* node table is not a tree rooted at row 0     */
#define CLS_E_BAD_DB (-3)        /* k-mer map inconsistent / unsupported shape   */
#define CLS_E_NO_DEVICE (-4)     /* no HIP device / HIP extension unusable       */
#define CLS_E_HIP (-5)           /* a HIP runtime call failed                    */
#define CLS_E_NOMEM (-6)
#define CLS_E_INTERNAL (-7)

/* ---- tree: Clade (core/src/domain/dtos/clade.rs:18-38) ------------------ */
/* NodeType, clade.rs:5-16.  LEAF-ness is decided by `kind` only
 * (clade.rs:166-172), never by the absence of children. */
enum { CLS_KIND_ROOT = 0, CLS_KIND_NODE = 1, CLS_KIND_LEAF = 2 };

#define CLS_NO_PARENT UINT64_MAX

/* One row per clade.  Row 0 is `tree.root`; the children of a row occupy
 * `n_children` consecutive rows starting at `first_child`, in the order of
 * `Clade.children`.  `id` is the clade's (arbitrary, non-dense) u64 id. */
typedef struct cls_node {
    uint64_t id;
    uint64_t parent;       /* Clade.parent or CLS_NO_PARENT (informational)      */
    uint32_t first_child;  /* row of the first child (ignored if n_children==0)  */
    uint32_t n_children;
    uint8_t kind;          /* CLS_KIND_*                                         */
    uint8_t has_children;  /* 0: `children: None`; 1: `Some(vec)` (maybe empty)  */
    uint8_t pad_[6];
} cls_node;                /* 32 bytes */

/* ---- k-mer index: KmersMap (core/src/domain/dtos/kmers_map.rs:77-87) ---- */
/* Borrowed flat view of
 *   KmersMap{ k_size, m_size, map: HashMap<MinimizerKey, MinimizerValue> }
 *   MinimizerValue(HashMap<u64 /+kmer hash+/, HashSet<u64 /+node ids+/>>)
 * as two nested CSR levels.  Bucket b holds k-mers
 * [bucket_kmer_off[b], bucket_kmer_off[b+1]); k-mer j holds node ids
 * node_ids[kmer_node_off[j] .. kmer_node_off[j+1]) in any order. */
typedef struct cls_db_desc {
    uint32_t abi_version;  /* CLS_ABI_VERSION */
    uint32_t n_nodes;
    const cls_node* nodes;
    uint64_t k_size;       /* kSize */
    uint64_t m_size;       /* mSize */
    uint64_t n_buckets;
    const uint64_t* bucket_key;       /* [n_buckets]   MinimizerKey.0            */
    const uint64_t* bucket_kmer_off;  /* [n_buckets+1]                           */
    uint64_t n_kmers;
    const uint64_t* kmer_hash;        /* [n_kmers]     murmur3_x64_128(kmer,0).0 */
    const uint64_t* kmer_node_off;    /* [n_kmers+1]                             */
    const uint64_t* node_ids;         /* [kmer_node_off[n_kmers]] clade ids      */
    /* abi_version >= 2 */
    uint32_t node_set_kind;           /* CLS_SETS_*                              */
    uint32_t pad_;
} cls_db_desc;

/* What `node_ids` lists per k-mer:
 * CLS_SETS_EXPLICIT  every member of the node set, as the reference's index file holds it
 *                    (MinimizerValue: HashSet<u64> of clade ids, kmers_map.rs:16-17);
 * CLS_SETS_LEAVES    only the LEAF-kind members.  `map_kmers_to_tree` builds every node set as a union of
 *                    root->leaf paths (build_database/mod.rs:160-169; `get_path_to_root`, clade.rs:127-156), so the
 *                    leaves determine it: node set := union over the listed leaves of {leaf, its ancestors, root}.
 *                    The explicit sets of a deep tree are depth times larger (50 k leaves at depth 900: terabytes);
 *                    this form is what a caller derives from them (filter by kind) or builds directly.  Every id
 *                    must be a childless LEAF-kind clade of the tree (else CLS_E_BAD_DB). */
#define CLS_SETS_EXPLICIT 0u
#define CLS_SETS_LEAVES 1u

/* ---- per-call parameters: the three Option<> arguments ------------------ */
#define CLS_HAS_MAX_ITERATIONS 1u      /* Some(max_iterations); else 1000        */
#define CLS_HAS_MIN_MATCH_COVERAGE 2u  /* Some(min_match_coverage); else 0.7     */
#define CLS_HAS_REMOVE_INTERSECTION 4u /* Some(remove_intersection); else false  */

typedef struct cls_params {
    uint32_t flags;             /* CLS_HAS_* bits                                */
    int32_t max_iterations;     /* place_sequence.rs:65                          */
    double min_match_coverage;  /* clamped to [0,1], place_sequence.rs:67-75     */
    uint8_t remove_intersection;/* place_sequence.rs:64                          */
    uint8_t pad_[7];
} cls_params;                   /* NULL == all None */

/* ---- result: PlacementStatus / MappedErrors as a fixed record ----------- */
/* One per query, in input order.  The caller rebuilds
 * Result<PlacementStatus, MappedErrors> from it:
 *
 * status                         reference outcome (place_sequence.rs)
 * CLS_UNCLASSIFIABLE_NO_MATCH    Ok(Unclassifiable("Query sequence {header:?} may not be related to the phylogeny")) :130-139
 * CLS_UNCLASSIFIABLE_NO_ROOT     Ok(Unclassifiable("Query sequence has no overlapping kmers with the reference tree")) :156-164
 * CLS_UNCLASSIFIABLE_COVERAGE    Ok(Unclassifiable("Insufficient kmers coverage: {one}")) :247-254   (one = coverage)
 * CLS_UNCLASSIFIABLE_LEVEL1      Ok(Unclassifiable("Tree introspection not possible. ...")) :446-453
 * CLS_IDENTITY_FOUND             Ok(IdentityFound(AdherenceTest{clade=clade_id, one, rest})) update_introspection_node.rs:45-85
 * CLS_MAX_RESOLUTION             Ok(MaxResolutionReached(clade_id, "LCA Accepted")) :461-464
 * CLS_INCONCLUSIVE               Ok(Inconclusive(.., "Multiple proposals")) :584-598 (set-theoretically
 *                                unreachable; one = number of tied proposals, clade_id = parent)
 * CLS_ERR_TOO_FEW_KMERS          Err("The sequence does not contain enough kmers.", UCPLACE0005) :98-102
 * CLS_ERR_MAX_ITER               Err("The maximum number of iterations has been reached.", UCPLACE0010) :295-301
 * CLS_ERR_ROOT_NO_CHILDREN       Err("The root node does not have children. This is unexpected.") :199-206
 * CLS_ERR_INVALID_BASE           the reference panics (kmers_map.rs:440); reported per read instead
 * CLS_ERR_READ_TOO_LONG          device-buffer entry only: the read is longer than the handle provisions for
 *                                (cls_db_set_max_read_len; cls_db_info.max_read_kmers); the host-buffer entries size
 *                                themselves by the batch and place reads of up to 2^25 bases
 */
enum {
    CLS_UNCLASSIFIABLE_NO_MATCH = 0,
    CLS_UNCLASSIFIABLE_NO_ROOT = 1,
    CLS_UNCLASSIFIABLE_COVERAGE = 2,
    CLS_UNCLASSIFIABLE_LEVEL1 = 3,
    CLS_IDENTITY_FOUND = 4,
    CLS_MAX_RESOLUTION = 5,
    CLS_INCONCLUSIVE = 6,
    CLS_ERR_TOO_FEW_KMERS = 7,
    CLS_ERR_MAX_ITER = 8,
    CLS_ERR_ROOT_NO_CHILDREN = 9,
    CLS_ERR_INVALID_BASE = 10,
    CLS_ERR_READ_TOO_LONG = 11
};

typedef struct cls_placement {
    uint8_t status;     /* CLS_* above                                           */
    uint8_t pad_[3];
    int32_t one;        /* AdherenceTest.one  (adherence_test.rs:12)             */
    int32_t rest;       /* AdherenceTest.rest (adherence_test.rs:15)             */
    uint32_t levels;    /* introspection levels entered (`iteration`, :280)      */
    uint64_t clade_id;  /* Clade.id of the placement                             */
} cls_placement;        /* 24 bytes */

/* Optional per-query counters (the tracing span fields
 * place_sequence.rs:30-41); used by parity tests and by the roofline
 * accounting of bench.py. */
typedef struct cls_query_stats {
    uint32_t n_query_kmers;    /* query.kmers.count       = 2(L-k+1)             */
    uint32_t n_matched;        /* query.kmers.treeMatches = |M|                  */
    uint32_t n_with_root;      /* subject.kmers.queryMatches = |M_root|          */
    uint32_t index_bytes;      /* engine-side accounting, not a reference quantity: bytes of index data (table
                                * entries, node records, split records) the kernels asked for to place this read;
                                * bench.py's roofline numerator; 0 where a kernel does not count it */
    uint64_t leaf_postings;    /* sum over M of |{LEAF-kind ids in nodes(h)}|    */
} cls_query_stats;             /* 24 bytes */

typedef struct cls_db cls_db;  /* opaque, immutable after create (Send + Sync)   */

typedef struct cls_db_info {
    uint32_t n_nodes;
    uint32_t max_depth;        /* levels below the root                          */
    uint32_t max_nonleaf_arity;
    uint32_t k_size;
    uint32_t m_size;
    uint32_t n_buckets;
    uint64_t n_kmers;
    uint64_t n_closed_kmers;   /* node-set closed under `parent` (tip-compressed)*/
    uint64_t table_slots;
    uint64_t postings_words;
    uint64_t hbm_bytes;        /* device bytes held by the handle                */
    uint32_t max_read_kmers;   /* per-read k-mer capacity of the device-buffer entry */
    int32_t device;
    uint32_t format;           /* 0: sorted lists (some node set is not closed under `parent`); 1: split-tree records */
    uint32_t binary_tree;      /* 1: every clade has zero or two children        */
    uint32_t direct_table;     /* 1: 2-bit-code direct table in use (k <= 15); 2: and the index is strand-symmetric
                                * (every k-mer shares its node set with its reverse complement: one lookup per window) */
    uint32_t n_tip_sets;       /* format 1: distinct tip lists (k-mers with the same one share a split tree) */
    uint32_t scratch_slots;    /* per-call scratch workspaces the handle holds right now (a caller that pipelines
                                * batches on ONE stream keeps one; at most 8) */
    uint32_t fat_direct_table; /* 1: k <= 12, the direct table is also kept with the set record inside its 16-byte entries */
} cls_db_info;

/* Number of usable HIP devices (0 if none). */
int cls_device_count(void);

/* Validate + re-encode + upload the index to `device` (-1: current device).
 * The views in `d` are only borrowed for the duration of the call.
 * Replaces nothing in the reference (it keeps the Tree in host hash maps,
 * ports/lib/src/functions/load_database.rs:9-53); called once after it. */
int cls_db_create(const cls_db_desc* d, int device, cls_db** out);
/* Host-only: run the validation + re-encoding of cls_db_create without
 * touching a device (CLS_OK, or the code cls_db_create would return). */
int cls_db_validate(const cls_db_desc* d);
void cls_db_destroy(cls_db* db);
int cls_db_info_get(const cls_db* db, cls_db_info* info);
/* The same for a caller compiled against an OLDER header: copies the first min(info_size, sizeof(cls_db_info))
 * bytes.  The struct only ever grows at its end (88 bytes in round 1, 96 since `fat_direct_table`): a binding that
 * may meet a newer library passes its own sizeof and never has bytes written past its struct. */
int cls_db_info_get2(const cls_db* db, void* info, size_t info_size);

/* Place `n` queries.  `bases` holds the concatenated sequences exactly as
 * `SequenceBody` holds them when place_sequence receives them (after the
 * FASTA stage, sequence.rs:47-56), `offsets[n+1]` their byte ranges.
 * Host pointers; synchronous; `out[n]` caller-allocated, input order.
 * Re-entrant: each call uses its own HIP stream.
 * Replaces: the per-query place_sequence() call, mod.rs:151-159. */
int cls_place_batch(cls_db* db, const char* bases, const uint64_t* offsets, uint32_t n,
                    const cls_params* params, cls_placement* out);

/* Same, with every buffer already resident in the HBM of the handle's
 * device; asynchronous on `hip_stream` (a hipStream_t, NULL = default
 * stream).  `d_stats` may be NULL.  This is the entry bench.py times. */
int cls_place_batch_device(cls_db* db, const void* d_bases, const void* d_offsets, uint32_t n,
                           const cls_params* params, void* d_out, void* d_stats, void* hip_stream);

/* Longest read (bases) cls_place_batch_device() provisions for (the lengths of a device-resident batch are not
 * known to the host).  Default 0: reads of up to 8192 k-mers (4096 + k - 1 bases), every read-length class launched;
 * longer reads are reported CLS_ERR_READ_TOO_LONG.  With n_bases set, the launch follows it: the read-length classes
 * beyond n_bases are not launched (a batch of 150 bp reads then costs one placement kernel, not one per class) and a
 * read longer than n_bases may be reported CLS_ERR_READ_TOO_LONG; a caller with reads beyond 8192 k-mers opts in
 * here.  Reads the LDS-tiled kernel cannot hold keep their per-k-mer state in the workspace: 80 bytes per base and
 * resident workgroup.  At most 2^25. */
int cls_db_set_max_read_len(cls_db* db, uint64_t n_bases);

/* Device time of the DOMINANT placement kernel (the wave-per-read kernel of the
 * 320-k-mer class; for a handle provisioned for long reads, cls_db_set_max_read_len,
 * the LDS-tiled kernel -- all of its launches together; the tuning knob time_class = 2 selects it for gene-length
 * reads too), accumulated over every cls_place_batch_device() launch on this
 * handle since the last reset: HIP events recorded around that kernel on the
 * caller's stream.  Waits for the launches still in flight.  Measurement aid for
 * bench.py's roofline figure; `reset` != 0 clears the accumulators afterwards. */
int cls_db_kernel_time(cls_db* db, double* sum_ms, uint64_t* launches, int reset);
/* Name (template instance, as rocprofv3 prints it without the argument list) of that dominant kernel for this
 * handle, without statistics: lets bench.py tie a committed profile to the kernel it really launches. */
int cls_db_kernel_name(const cls_db* db, char* buf, size_t len);

/* Host-buffer variant that also returns the per-query counters. */
int cls_place_batch_stats(cls_db* db, const char* bases, const uint64_t* offsets, uint32_t n,
                          const cls_params* params, cls_placement* out, cls_query_stats* stats);

/* ---- FASTA input stage (file_or_stdin.rs:76-116, sequence.rs:47-56) ------ */
typedef struct cls_fasta {
    uint32_t n;               /* records                                        */
    uint32_t truncated;       /* 1: stopped at "unexpected sequence without header" (error ignored by the caller, mod.rs:119) */
    char* headers;            /* concatenated header bytes                      */
    uint64_t* header_off;     /* [n+1]                                          */
    char* bases;              /* concatenated filtered (upper-case ACGT) bases  */
    uint64_t* base_off;       /* [n+1]                                          */
} cls_fasta;

int cls_fasta_parse(const char* text, size_t len, cls_fasta* out);
void cls_fasta_free(cls_fasta* f);

/* The same stage as data-parallel passes on the device: `d_text` = the file's bytes in HBM; the filtered bases
 * and their offsets stay in HBM, ready for cls_place_batch_device() (the reads never return to the host); the
 * headers are only needed by the output stage.  Synchronises `hip_stream` (the sizes of the outputs depend on
 * the text).  Buffers hold at least n_bases / n_header_bytes bytes and n + 1 offsets. */
typedef struct cls_fasta_dev {
    uint32_t n;
    uint32_t truncated;
    void* d_headers;          /* concatenated header bytes                      */
    void* d_header_off;       /* uint64 [n+1]                                   */
    void* d_bases;            /* concatenated filtered (upper-case ACGT) bases  */
    void* d_base_off;         /* uint64 [n+1]                                   */
    uint64_t n_header_bytes;
    uint64_t n_bases;
} cls_fasta_dev;
int cls_fasta_scan_device(const void* d_text, uint64_t len, cls_fasta_dev* out, void* hip_stream);
void cls_fasta_dev_free(cls_fasta_dev* f);
/* Host text in, host records out, through the device passes on `device` (-1: current). */
int cls_fasta_parse_gpu(const char* text, size_t len, int device, cls_fasta* out);
/* FASTA text -> placement records without the reads ever returning to the host: H2D of the file, the device
 * FASTA stage, cls_place_batch_device() on its output, D2H of the records (`*records`, free() it) and of the
 * headers (`fa`: n, truncated, headers, header_off; its bases / base_off stay NULL; cls_fasta_free() it).
 * Replaces mod.rs:108-159 (reader thread + channel + per-query place_sequence). */
int cls_place_fasta_text(cls_db* db, const char* text, size_t len, const cls_params* params, cls_fasta* fa,
                         cls_placement** records);

/* Experiment knobs (grid sizes, locality-key definition, kernel family; none changes a result; names in
 * csrc/cls_tuning.h are the CLS_* variables in lower case without the prefix, e.g. "no_order").  Process-global,
 * meant for A/B runs: the library itself never reads the environment.  cls_tuning_from_env() takes every knob
 * from its CLS_* variable, once, when a tool asks for it.  Set knobs before creating handles. */
int cls_set_tuning(const char* name, int value);
void cls_tuning_from_env(void);

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* cls_last_error(void);

/* "classeq2_amd <version> gfx950 abi<N>" */
const char* cls_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CLS_PLACE_H */
