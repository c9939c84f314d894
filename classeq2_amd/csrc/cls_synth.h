/*
 * cls_synth.h -- deterministic synthetic workload generator (libclssynth.so).
 *
 * Not part of the drop-in boundary: it produces the inputs BASELINE.json's
 * configs name: a random tree, reference sequences evolved down
 * it, the k-mer index a reference `cls build-db` would derive from them
 * (core/src/use_cases/build_database/mod.rs:26-181: every leaf's forward +
 * reverse-complement k-mers, each mapped to the union of root->leaf id paths,
 * bucketed by the hash of the k-mer's first m characters), and query reads.
 * Host C++ only; consumed by tests/, bench.py and the oracle alike through the
 * same `cls_db_desc` view the engine's cls_db_create() takes.
 */
#ifndef CLS_SYNTH_H
#define CLS_SYNTH_H

#include <stdint.h>

#include "cls_place.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cls_synth_cfg {
    uint32_t n_leaves;
    uint32_t ref_len;        /* reference sequence length                        */
    uint32_t k_size;
    uint32_t m_size;
    uint64_t seed_tree;      /* 1 */
    uint64_t seed_refseq;    /* 2 */
    double edge_sub_rate;    /* per-edge per-site substitution probability 0.01  */
    uint32_t deep;           /* 0: Yule (uniform splits); 1: caterpillar-biased  */
    uint32_t max_depth;      /* depth cap (deep mode: 900); 0 = none             */
    double collapse_prob;    /* P(internal node dissolved into its parent) -> polytomies, 0 for the bench configs */
    uint64_t id_stride;      /* clade id = id_offset + id_stride * preorder (1)  */
    uint64_t id_offset;      /* (0); the root keeps id_offset                    */
    uint32_t threads;        /* 0 = all cores                                    */
    uint32_t tips_only;      /* 1: the view lists only the LEAF ids of every node set (CLS_SETS_LEAVES): the explicit
                              * sets of a deep tree (config 5: 50 k leaves at depth 900) would be terabytes and are
                              * never materialised                                */
} cls_synth_cfg;

typedef struct cls_synth_db cls_synth_db; /* owns every array `desc` points at */

int cls_synth_db_create(const cls_synth_cfg* cfg, cls_synth_db** out);
void cls_synth_db_destroy(cls_synth_db* s);
/* Borrowed view, valid until destroy. */
const cls_db_desc* cls_synth_db_desc(const cls_synth_db* s);
uint32_t cls_synth_n_leaves(const cls_synth_db* s);
uint32_t cls_synth_max_depth(const cls_synth_db* s);
/* Leaf i (DFS order): its clade id and its reference sequence (ref_len bytes). */
uint64_t cls_synth_leaf_id(const cls_synth_db* s, uint32_t i);
const char* cls_synth_leaf_seq(const cls_synth_db* s, uint32_t i);

/* Reads: uniform leaf, uniform start, strand flip p=0.5, per-base
 * substitution `err`, a fraction `frac_random` fully random.  Upper-case ACGT.
 * bases[n_reads*read_len], offsets[n_reads+1], truth_leaf[n_reads] (leaf index
 * or UINT32_MAX for random reads; may be NULL).  `first` lets shards generate
 * disjoint slices of ONE global read stream: read i depends only on
 * (seed, first+i). */
int cls_synth_reads(const cls_synth_db* s, uint64_t seed, uint64_t first, uint32_t n_reads,
                    uint32_t read_len, double err, double frac_random, char* bases,
                    uint64_t* offsets, uint32_t* truth_leaf);

const char* cls_synth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
