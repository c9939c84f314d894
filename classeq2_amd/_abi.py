"""ctypes mirror of include/cls_place.h and csrc/cls_synth.h (struct layouts only)."""

import ctypes as C

ABI_VERSION = 2
SETS_EXPLICIT, SETS_LEAVES = 0, 1

KIND_ROOT, KIND_NODE, KIND_LEAF = 0, 1, 2
NO_PARENT = (1 << 64) - 1

HAS_MAX_ITERATIONS = 1
HAS_MIN_MATCH_COVERAGE = 2
HAS_REMOVE_INTERSECTION = 4

(
    UNCLASSIFIABLE_NO_MATCH,
    UNCLASSIFIABLE_NO_ROOT,
    UNCLASSIFIABLE_COVERAGE,
    UNCLASSIFIABLE_LEVEL1,
    IDENTITY_FOUND,
    MAX_RESOLUTION,
    INCONCLUSIVE,
    ERR_TOO_FEW_KMERS,
    ERR_MAX_ITER,
    ERR_ROOT_NO_CHILDREN,
    ERR_INVALID_BASE,
    ERR_READ_TOO_LONG,
) = range(12)

STATUS_NAMES = [
    "UNCLASSIFIABLE_NO_MATCH",
    "UNCLASSIFIABLE_NO_ROOT",
    "UNCLASSIFIABLE_COVERAGE",
    "UNCLASSIFIABLE_LEVEL1",
    "IDENTITY_FOUND",
    "MAX_RESOLUTION",
    "INCONCLUSIVE",
    "ERR_TOO_FEW_KMERS",
    "ERR_MAX_ITER",
    "ERR_ROOT_NO_CHILDREN",
    "ERR_INVALID_BASE",
    "ERR_READ_TOO_LONG",
]


class Node(C.Structure):
    _fields_ = [
        ("id", C.c_uint64),
        ("parent", C.c_uint64),
        ("first_child", C.c_uint32),
        ("n_children", C.c_uint32),
        ("kind", C.c_uint8),
        ("has_children", C.c_uint8),
        ("pad_", C.c_uint8 * 6),
    ]


class DbDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_nodes", C.c_uint32),
        ("nodes", C.POINTER(Node)),
        ("k_size", C.c_uint64),
        ("m_size", C.c_uint64),
        ("n_buckets", C.c_uint64),
        ("bucket_key", C.POINTER(C.c_uint64)),
        ("bucket_kmer_off", C.POINTER(C.c_uint64)),
        ("n_kmers", C.c_uint64),
        ("kmer_hash", C.POINTER(C.c_uint64)),
        ("kmer_node_off", C.POINTER(C.c_uint64)),
        ("node_ids", C.POINTER(C.c_uint64)),
        ("node_set_kind", C.c_uint32),
        ("pad_", C.c_uint32),
    ]


class Params(C.Structure):
    _fields_ = [
        ("flags", C.c_uint32),
        ("max_iterations", C.c_int32),
        ("min_match_coverage", C.c_double),
        ("remove_intersection", C.c_uint8),
        ("pad_", C.c_uint8 * 7),
    ]


class Placement(C.Structure):
    _fields_ = [
        ("status", C.c_uint8),
        ("pad_", C.c_uint8 * 3),
        ("one", C.c_int32),
        ("rest", C.c_int32),
        ("levels", C.c_uint32),
        ("clade_id", C.c_uint64),
    ]


class QueryStats(C.Structure):
    _fields_ = [
        ("n_query_kmers", C.c_uint32),
        ("n_matched", C.c_uint32),
        ("n_with_root", C.c_uint32),
        ("index_bytes", C.c_uint32),
        ("leaf_postings", C.c_uint64),
    ]


class DbInfo(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("max_nonleaf_arity", C.c_uint32),
        ("k_size", C.c_uint32),
        ("m_size", C.c_uint32),
        ("n_buckets", C.c_uint32),
        ("n_kmers", C.c_uint64),
        ("n_closed_kmers", C.c_uint64),
        ("table_slots", C.c_uint64),
        ("postings_words", C.c_uint64),
        ("hbm_bytes", C.c_uint64),
        ("max_read_kmers", C.c_uint32),
        ("device", C.c_int32),
        ("format", C.c_uint32),
        ("binary_tree", C.c_uint32),
        ("direct_table", C.c_uint32),
        ("n_tip_sets", C.c_uint32),
        ("scratch_slots", C.c_uint32),
        ("fat_direct_table", C.c_uint32),
    ]


class Fasta(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("truncated", C.c_uint32),
        ("headers", C.POINTER(C.c_char)),
        ("header_off", C.POINTER(C.c_uint64)),
        ("bases", C.POINTER(C.c_char)),
        ("base_off", C.POINTER(C.c_uint64)),
    ]


class ServiceStats(C.Structure):
    _fields_ = [
        ("jobs_submitted", C.c_uint64),
        ("jobs_done", C.c_uint64),
        ("reads_placed", C.c_uint64),
        ("device_batches", C.c_uint64),
        ("max_jobs_in_batch", C.c_uint64),
        ("models", C.c_uint64),
    ]


class SynthCfg(C.Structure):
    _fields_ = [
        ("n_leaves", C.c_uint32),
        ("ref_len", C.c_uint32),
        ("k_size", C.c_uint32),
        ("m_size", C.c_uint32),
        ("seed_tree", C.c_uint64),
        ("seed_refseq", C.c_uint64),
        ("edge_sub_rate", C.c_double),
        ("deep", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("collapse_prob", C.c_double),
        ("id_stride", C.c_uint64),
        ("id_offset", C.c_uint64),
        ("threads", C.c_uint32),
        ("tips_only", C.c_uint32),
    ]


assert C.sizeof(Node) == 32
assert C.sizeof(Placement) == 24
assert C.sizeof(QueryStats) == 24
assert C.sizeof(Params) == 24

import numpy as np  # noqa: E402

NODE_DTYPE = np.dtype(
    [
        ("id", "<u8"),
        ("parent", "<u8"),
        ("first_child", "<u4"),
        ("n_children", "<u4"),
        ("kind", "u1"),
        ("has_children", "u1"),
        ("pad_", "u1", (6,)),
    ]
)
PLACEMENT_DTYPE = np.dtype(
    [("status", "u1"), ("pad_", "u1", (3,)), ("one", "<i4"), ("rest", "<i4"), ("levels", "<u4"), ("clade_id", "<u8")]
)
STATS_DTYPE = np.dtype(
    [("n_query_kmers", "<u4"), ("n_matched", "<u4"), ("n_with_root", "<u4"), ("index_bytes", "<u4"), ("leaf_postings", "<u8")]
)
assert NODE_DTYPE.itemsize == 32 and PLACEMENT_DTYPE.itemsize == 24 and STATS_DTYPE.itemsize == 24
