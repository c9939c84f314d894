/*
 * cls_oracle.c -- flat C port of classeq2's per-query placement algorithm.
 *
 * TEST INFRASTRUCTURE ONLY: the checker of tests/, __graft_entry__.smoke() and
 * the `cpu_baseline` ("port") leg of bench.py.  Nothing the product ships may
 * link, load or call it.
 *
 * It restates, step for step, the reference's
 *   core/src/use_cases/place_sequences/place_sequence.rs:42-602
 *   core/src/use_cases/place_sequences/update_introspection_node.rs:13-91
 *   core/src/domain/dtos/kmers_map.rs (build_kmer_from_string :375-398,
 *     hash_kmer :157-159, get_overlapping_hashed_kmers :273-311,
 *     get_minimized_hashes_with_node :211-229, get_hashed_kmers_with_node :189-203)
 * with sorted arrays + bit sets standing in for Rust's HashMap/HashSet.  The
 * per-level work keeps the reference's shape (every non-LEAF child rescans all
 * matched k-mers; `rest` is the literal union of the siblings' sets); only the
 * per-query deep clone of the whole index (place_sequence.rs:77-80) and the
 * tracing calls are left out, as they change no result.
 *
 * Pin status: MurmurHash3 is pinned by the reference docs' known answers
 * (docs/book/02-build-db.md:181-196, tests/test_oracle_kat.py); the placement
 * decisions are "parity unpinned" by reference outputs (no usable golden
 * fixture exists) -- this port is validated record-for-record
 * against oracle/oracle_literal.py instead (tests/test_oracle_port.py).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "cls_place.h"

/* ---- a3: MurmurHash3_x64_128, seed 0, first half (kmers_map.rs:157-159) -- */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t fmix64(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}
uint64_t cls_oracle_murmur3_h1(const char* key, size_t len) {
    const uint8_t* data = (const uint8_t*)key;
    const size_t nblocks = len / 16;
    uint64_t h1 = 0, h2 = 0;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    for (size_t i = 0; i < nblocks; i++) {
        uint64_t k1, k2;
        memcpy(&k1, data + 16 * i, 8);
        memcpy(&k2, data + 16 * i + 8, 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t* tail = data + nblocks * 16;
    uint64_t k1 = 0, k2 = 0;
    switch (len & 15) {
        case 15: k2 ^= (uint64_t)tail[14] << 48; /* fallthrough */
        case 14: k2 ^= (uint64_t)tail[13] << 40; /* fallthrough */
        case 13: k2 ^= (uint64_t)tail[12] << 32; /* fallthrough */
        case 12: k2 ^= (uint64_t)tail[11] << 24; /* fallthrough */
        case 11: k2 ^= (uint64_t)tail[10] << 16; /* fallthrough */
        case 10: k2 ^= (uint64_t)tail[9] << 8;   /* fallthrough */
        case 9:  k2 ^= (uint64_t)tail[8];
                 k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; /* fallthrough */
        case 8:  k1 ^= (uint64_t)tail[7] << 56; /* fallthrough */
        case 7:  k1 ^= (uint64_t)tail[6] << 48; /* fallthrough */
        case 6:  k1 ^= (uint64_t)tail[5] << 40; /* fallthrough */
        case 5:  k1 ^= (uint64_t)tail[4] << 32; /* fallthrough */
        case 4:  k1 ^= (uint64_t)tail[3] << 24; /* fallthrough */
        case 3:  k1 ^= (uint64_t)tail[2] << 16; /* fallthrough */
        case 2:  k1 ^= (uint64_t)tail[1] << 8;  /* fallthrough */
        case 1:  k1 ^= (uint64_t)tail[0];
                 k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2;
    return h1;
}

/* ---- index ---------------------------------------------------------------- */
typedef struct {
    uint64_t hash;
    uint32_t bucket;
    uint64_t kmer; /* index into the desc's k-mer arrays */
} hent;

struct cls_oracle {
    uint32_t n_nodes;
    cls_node* nodes;
    uint64_t k, m;
    uint64_t n_buckets;
    uint64_t* bucket_key;
    uint64_t n_kmers;
    hent* by_hash;          /* sorted by hash: the HashMap<u64,..> lookups     */
    uint64_t* node_off;     /* per k-mer, into `nodes_sorted`                  */
    uint64_t* nodes_sorted; /* each k-mer's HashSet<u64> as a sorted array     */
    uint32_t* n_leaf_ids;   /* per k-mer: LEAF-kind ids in its set (stats)     */
    uint64_t* bucket_off;   /* [n_buckets+1] k-mer ranges of the buckets (descriptor order) */
    uint64_t* kmer_hash;    /* hashes in descriptor order                       */
    int reference_cost;     /* also pay the reference's per-query overheads (cls_oracle_set_reference_cost) */
    /* CLS_SETS_LEAVES input (deep trees, whose explicit sets do not fit any memory): the node set of a k-mer is the
     * union of the root->leaf paths of its listed leaves (build_database/mod.rs:160-169), kept LAZILY: a clade is a
     * member iff a listed leaf lies in its subtree.  `nodes_sorted` then holds the leaves' DFS entry times and
     * tin/tout the entry / exit time of every row.  tests/test_oracle.py holds this mode to the explicit one. */
    int leaves_only;
    uint32_t *tin, *tout;
};
typedef struct cls_oracle cls_oracle;

static int cmp_u64(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : x > y;
}
static int cmp_hent(const void* a, const void* b) {
    const hent *x = a, *y = b;
    if (x->hash != y->hash) return x->hash < y->hash ? -1 : 1;
    return x->kmer < y->kmer ? -1 : x->kmer > y->kmer;
}
static int set_contains(const uint64_t* s, uint64_t n, uint64_t v) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        if (s[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo < n && s[lo] == v;
}

/* is the clade of row `row` in the node set of k-mer j?  (HashSet<u64>::contains, kmers_map.rs:44-46, :220) */
static int kmer_has_node(const struct cls_oracle* o, uint64_t j, uint32_t row) {
    const uint64_t* s = o->nodes_sorted + o->node_off[j];
    const uint64_t n = o->node_off[j + 1] - o->node_off[j];
    if (!o->leaves_only) return set_contains(s, n, o->nodes[row].id);
    uint64_t lo = 0, hi = n;  /* first listed leaf entered at or after `row` ... */
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        if (s[mid] < o->tin[row]) lo = mid + 1; else hi = mid;
    }
    return lo < n && s[lo] < o->tout[row];  /* ... and before `row` is left: it is below (or is) `row` */
}

void cls_oracle_destroy(cls_oracle* o) {
    if (!o) return;
    free(o->tin); free(o->tout);
    free(o->nodes); free(o->bucket_key); free(o->by_hash); free(o->node_off);
    free(o->nodes_sorted); free(o->n_leaf_ids); free(o->bucket_off); free(o->kmer_hash); free(o);
}

int cls_oracle_create(const cls_db_desc* d, cls_oracle** out) {
    if (!d || !out || d->n_nodes == 0 || d->k_size == 0) return CLS_E_INVALID_ARG;
    cls_oracle* o = calloc(1, sizeof *o);
    if (!o) return CLS_E_NOMEM;
    o->leaves_only = d->abi_version >= 2 && d->node_set_kind == CLS_SETS_LEAVES;
    o->n_nodes = d->n_nodes; o->k = d->k_size; o->m = d->m_size;
    o->n_buckets = d->n_buckets; o->n_kmers = d->n_kmers;
    uint64_t tot = d->n_kmers ? d->kmer_node_off[d->n_kmers] : 0;
    o->nodes = malloc(sizeof(cls_node) * d->n_nodes);
    o->bucket_key = malloc(8 * (d->n_buckets + 1));
    o->by_hash = malloc(sizeof(hent) * (d->n_kmers + 1));
    o->node_off = malloc(8 * (d->n_kmers + 1));
    o->nodes_sorted = malloc(8 * (tot + 1));
    o->n_leaf_ids = calloc(d->n_kmers + 1, 4);
    o->bucket_off = malloc(8 * (d->n_buckets + 1));
    o->kmer_hash = malloc(8 * (d->n_kmers + 1));
    if (!o->nodes || !o->bucket_key || !o->by_hash || !o->node_off || !o->nodes_sorted || !o->n_leaf_ids || !o->bucket_off || !o->kmer_hash) {
        cls_oracle_destroy(o);
        return CLS_E_NOMEM;
    }
    memcpy(o->nodes, d->nodes, sizeof(cls_node) * d->n_nodes);
    if (d->n_buckets) memcpy(o->bucket_key, d->bucket_key, 8 * d->n_buckets);
    if (d->n_buckets) memcpy(o->bucket_off, d->bucket_kmer_off, 8 * (d->n_buckets + 1)); else o->bucket_off[0] = 0;
    if (d->n_kmers) memcpy(o->kmer_hash, d->kmer_hash, 8 * d->n_kmers);
    /* sorted list of LEAF-kind clade ids, for the leaf_postings statistic */
    uint64_t* leaf_ids = malloc(8 * (size_t)d->n_nodes);
    uint64_t n_leaf = 0;
    for (uint32_t r = 0; r < d->n_nodes; r++) if (d->nodes[r].kind == CLS_KIND_LEAF) leaf_ids[n_leaf++] = d->nodes[r].id;
    qsort(leaf_ids, n_leaf, 8, cmp_u64);
    /* leaves-only input: DFS entry / exit times of every row, and id -> row for the listed ids */
    uint64_t* id_row = NULL; /* (id, row) pairs sorted by id */
    if (o->leaves_only) {
        o->tin = malloc(4 * (size_t)d->n_nodes); o->tout = malloc(4 * (size_t)d->n_nodes);
        uint32_t* stack = malloc(8 * (size_t)d->n_nodes + 8);
        id_row = malloc(16 * (size_t)d->n_nodes);
        if (!o->tin || !o->tout || !stack || !id_row) { free(stack); free(id_row); free(leaf_ids); cls_oracle_destroy(o); return CLS_E_NOMEM; }
        uint32_t sp = 0, t = 0;
        stack[sp++] = 0; stack[sp++] = 0; /* (row, next child) */
        o->tin[0] = t++;
        while (sp) {
            uint32_t row = stack[sp - 2], next = stack[sp - 1];
            if (next < d->nodes[row].n_children) {
                stack[sp - 1] = next + 1;
                uint32_t c = d->nodes[row].first_child + next;
                o->tin[c] = t++;
                stack[sp++] = c; stack[sp++] = 0;
            } else { o->tout[row] = t; sp -= 2; }
        }
        free(stack);
        for (uint32_t r = 0; r < d->n_nodes; r++) { id_row[2 * r] = d->nodes[r].id; id_row[2 * r + 1] = r; }
        qsort(id_row, d->n_nodes, 16, cmp_u64);
    }
    uint64_t w = 0;
    for (uint64_t b = 0; b < d->n_buckets; b++)
        for (uint64_t j = d->bucket_kmer_off[b]; j < d->bucket_kmer_off[b + 1]; j++) {
            o->by_hash[j].hash = d->kmer_hash[j];
            o->by_hash[j].bucket = (uint32_t)b;
            o->by_hash[j].kmer = j;
        }
    for (uint64_t j = 0; j < d->n_kmers; j++) {
        uint64_t lo = d->kmer_node_off[j], hi = d->kmer_node_off[j + 1], start = w;
        o->node_off[j] = w;
        if (o->leaves_only) { /* the listed leaves, as DFS entry times */
            for (uint64_t i = lo; i < hi; i++) {
                uint64_t a = 0, b = d->n_nodes, id = d->node_ids[i];
                while (a < b) { uint64_t mid = (a + b) / 2; if (id_row[2 * mid] < id) a = mid + 1; else b = mid; }
                if (a >= d->n_nodes || id_row[2 * a] != id || d->nodes[id_row[2 * a + 1]].kind != CLS_KIND_LEAF) {
                    free(id_row); free(leaf_ids); cls_oracle_destroy(o); return CLS_E_BAD_DB;
                }
                o->nodes_sorted[w + (i - lo)] = o->tin[id_row[2 * a + 1]];
            }
            qsort(o->nodes_sorted + w, hi - lo, 8, cmp_u64);
            for (uint64_t i = 0; i < hi - lo; i++)
                if (i == 0 || o->nodes_sorted[start + i] != o->nodes_sorted[start + i - 1]) o->nodes_sorted[w++] = o->nodes_sorted[start + i];
            o->n_leaf_ids[j] = (uint32_t)(w - start);
            continue;
        }
        memcpy(o->nodes_sorted + w, d->node_ids + lo, 8 * (hi - lo));
        qsort(o->nodes_sorted + w, hi - lo, 8, cmp_u64);
        for (uint64_t i = 0; i < hi - lo; i++) /* a set: drop duplicates */
            if (i == 0 || o->nodes_sorted[start + i] != o->nodes_sorted[start + i - 1]) o->nodes_sorted[w++] = o->nodes_sorted[start + i];
        for (uint64_t i = start; i < w; i++) if (set_contains(leaf_ids, n_leaf, o->nodes_sorted[i])) o->n_leaf_ids[j]++;
    }
    o->node_off[d->n_kmers] = w;
    free(leaf_ids);
    free(id_row);
    qsort(o->by_hash, d->n_kmers, sizeof(hent), cmp_hent);
    *out = o;
    return CLS_OK;
}

/* ---- per-query scratch ------------------------------------------------------ */
typedef struct { void* p; size_t cap; } vec;
typedef struct {
    vec hashes, mins;          /* query k-mers      */
    vec ent_kmer, ent_hidx;    /* M_root entries    */
    vec sets;                  /* per-child bitsets */
    vec child_rows, cand_rows;
    vec buf;
} scratch;

static void* ensure(vec* v, size_t need, size_t elt) {
    if (need > v->cap) {
        size_t n = v->cap ? v->cap : 64;
        while (n < need) n *= 2;
        v->p = realloc(v->p, n * elt);
        v->cap = n;
    }
    return v->p;
}
static size_t uniq_u64(uint64_t* a, size_t n) {
    if (!n) return 0;
    qsort(a, n, 8, cmp_u64);
    size_t w = 1;
    for (size_t i = 1; i < n; i++) if (a[i] != a[w - 1]) a[w++] = a[i];
    return w;
}
static inline int popcnt_and_not(const uint64_t* a, const uint64_t* b, size_t words) { /* |a \ b| */
    int c = 0;
    for (size_t i = 0; i < words; i++) c += __builtin_popcountll(a[i] & ~b[i]);
    return c;
}
static inline int popcnt(const uint64_t* a, size_t words) {
    int c = 0;
    for (size_t i = 0; i < words; i++) c += __builtin_popcountll(a[i]);
    return c;
}

static void place_one(const cls_oracle* o, const char* seq, uint64_t L, int32_t max_iterations,
                      double min_cov, int rm_int, scratch* S, cls_placement* out, cls_query_stats* st) {
    memset(out, 0, sizeof *out);
    if (st) memset(st, 0, sizeof *st);
    const uint64_t k = o->k;
    /* build_kmer_from_string, kmers_map.rs:383-385: shorter than k -> [] -> "<2 k-mers" (:98-102) */
    if (L < k) { out->status = CLS_ERR_TOO_FEW_KMERS; return; }
    /* upper-cased forward string + reverse complement (kmers_map.rs:410, :431-443) */
    char* fwd = ensure(&S->buf, 2 * L + 2, 1); char* rc = fwd + L + 1;
    for (uint64_t i = 0; i < L; i++) {
        char c = seq[i];
        if (c >= 'a' && c <= 'z') c = (char)(c - 32);
        char cc;
        switch (c) { case 'A': cc = 'T'; break; case 'T': cc = 'A'; break; case 'C': cc = 'G'; break; case 'G': cc = 'C'; break;
            default: out->status = CLS_ERR_INVALID_BASE; return; /* the reference panics here */ }
        fwd[i] = c; rc[L - 1 - i] = cc;
    }
    const uint64_t nf = L - k + 1, nk = 2 * nf;
    if (st) st->n_query_kmers = (uint32_t)nk;
    if (nk < 2) { out->status = CLS_ERR_TOO_FEW_KMERS; return; }
    uint64_t* hashes = ensure(&S->hashes, nk, 8); uint64_t* mins = ensure(&S->mins, nk, 8);
    const uint64_t mlen = o->m < k ? o->m : k; /* chars().take(m) */
    for (uint64_t i = 0; i < nf; i++) {
        hashes[i] = cls_oracle_murmur3_h1(fwd + i, k);
        hashes[nf + i] = cls_oracle_murmur3_h1(rc + i, k);
        mins[i] = cls_oracle_murmur3_h1(fwd + i, mlen);
        mins[nf + i] = cls_oracle_murmur3_h1(rc + i, mlen);
    }
    /* get_overlapping_hashed_kmers, kmers_map.rs:273-311: the two HashSets */
    size_t nh = uniq_u64(hashes, nk), nm = uniq_u64(mins, nk);
    /* M = { (bucket, hash) : bucket.key in minimizers, hash in hashes }; keep M_root members */
    uint64_t* ent_kmer = S->ent_kmer.p; uint32_t* ent_hidx = S->ent_hidx.p;
    uint64_t n_m = 0, n_root = 0, leaf_post = 0, n_hidx = 0;
    for (size_t hi = 0; hi < nh; hi++) {
        uint64_t h = hashes[hi], lo = 0, up = o->n_kmers;
        while (lo < up) { uint64_t mid = (lo + up) / 2; if (o->by_hash[mid].hash < h) lo = mid + 1; else up = mid; }
        int any_root = 0;
        for (; lo < o->n_kmers && o->by_hash[lo].hash == h; lo++) {
            if (!set_contains(mins, nm, o->bucket_key[o->by_hash[lo].bucket])) continue;
            uint64_t j = o->by_hash[lo].kmer;
            n_m++;
            leaf_post += o->n_leaf_ids[j];
            /* get_minimized_hashes_with_node(root.id), kmers_map.rs:211-229 */
            if (kmer_has_node(o, j, 0)) {
                ent_kmer = ensure(&S->ent_kmer, n_root + 1, 8); ent_hidx = ensure(&S->ent_hidx, n_root + 1, 4);
                ent_kmer[n_root] = j; ent_hidx[n_root] = (uint32_t)n_hidx; n_root++; any_root = 1;
            }
        }
        if (any_root) n_hidx++; /* get_hashed_kmers_with_node flattens buckets into ONE set of hashes */
    }
    if (st) { st->n_matched = (uint32_t)n_m; st->n_with_root = (uint32_t)n_root; st->leaf_postings = leaf_post; }
    if (n_m == 0) { out->status = CLS_UNCLASSIFIABLE_NO_MATCH; return; }     /* :130-139 */
    if (n_root == 0) { out->status = CLS_UNCLASSIFIABLE_NO_ROOT; return; }   /* :156-164 */
    if (!o->nodes[0].has_children) { out->status = CLS_ERR_ROOT_NO_CHILDREN; return; } /* :199-206 */
    double expected = round((double)n_m * min_cov);                          /* :231-232 */
    uint64_t expected_usize = isnan(expected) ? 0 : (uint64_t)expected;      /* Rust `as usize`: NaN -> 0 */
    if (n_root < expected_usize) {                                           /* :247 */
        out->status = CLS_UNCLASSIFIABLE_COVERAGE; out->one = (int32_t)n_root; return;
    }
    const size_t words = (n_hidx + 63) / 64;
    /* children of the root: ALL of them (leaves filtered at :322-324) */
    uint32_t n_children = o->nodes[0].n_children;
    uint32_t* child_rows = ensure(&S->child_rows, n_children + 1, 4); uint32_t* cand_rows = ensure(&S->cand_rows, n_children + 1, 4);
    for (uint32_t i = 0; i < n_children; i++) child_rows[i] = o->nodes[0].first_child + i;
    uint32_t parent_row = 0;
    int32_t iteration = 0;
    for (;;) {
        iteration++;
        out->levels = (uint32_t)iteration;
        if (iteration > max_iterations) { out->status = CLS_ERR_MAX_ITER; return; } /* :295-301 */
        /* PHASE 1a :319-333 */
        uint32_t n_cand = 0;
        uint64_t* sets = ensure(&S->sets, (size_t)(n_children + 2) * (words ? words : 1), 8);
        for (uint32_t ci = 0; ci < n_children; ci++) {
            const cls_node* c = &o->nodes[child_rows[ci]];
            if (c->kind == CLS_KIND_LEAF) continue;
            uint64_t* set = sets + (size_t)n_cand * words;
            memset(set, 0, 8 * words);
            int any = 0;
            for (uint64_t e = 0; e < n_root; e++) { /* full rescan per child, kmers_map.rs:37-53 */
                uint64_t j = ent_kmer[e];
                if (kmer_has_node(o, j, child_rows[ci])) {
                    set[ent_hidx[e] >> 6] |= 1ULL << (ent_hidx[e] & 63); any = 1;
                }
            }
            if (any) cand_rows[n_cand++] = child_rows[ci];
        }
        /* PHASE 1b :353-418 */
        uint64_t* rest = sets + (size_t)n_cand * words;
        uint32_t n_prop = 0, prop_row = 0; int32_t prop_one = 0, prop_rest = 0;
        int32_t best_diff = 0; uint32_t n_best = 0;
        for (uint32_t a = 0; a < n_cand; a++) {
            const uint64_t* ka = sets + (size_t)a * words;
            int32_t one, rst; uint32_t n_rest_sets = 0;
            memset(rest, 0, 8 * words);
            for (uint32_t b = 0; b < n_cand; b++) {
                if (o->nodes[cand_rows[b]].id == o->nodes[cand_rows[a]].id) continue; /* :361 compares ids */
                const uint64_t* kb = sets + (size_t)b * words;
                for (size_t w = 0; w < words; w++) rest[w] |= kb[w];
                n_rest_sets++;
            }
            if (n_rest_sets == 0) { one = popcnt(ka, words); rst = 0; }           /* :369-375 */
            else if (rm_int) { one = popcnt_and_not(ka, rest, words); rst = popcnt_and_not(rest, ka, words); }
            else { one = popcnt(ka, words); rst = popcnt(rest, words); }
            if (one > rst) {                                                         /* :411-417 */
                int32_t diff = one - rst;
                if (n_prop == 0 || diff > best_diff) { best_diff = diff; n_best = 1; prop_row = cand_rows[a]; prop_one = one; prop_rest = rst; }
                else if (diff == best_diff) n_best++;
                n_prop++;
            }
        }
        /* PHASE 2 :436-600 */
        if (n_prop == 0) {
            if (iteration == 1) { out->status = CLS_UNCLASSIFIABLE_LEVEL1; return; }
            out->status = CLS_MAX_RESOLUTION; out->clade_id = o->nodes[parent_row].id; return;
        }
        if (n_prop > 1 && n_best != 1) {                                            /* :575-598 */
            out->status = CLS_INCONCLUSIVE; out->one = (int32_t)n_prop; out->clade_id = o->nodes[parent_row].id; return;
        }
        /* update_introspection_node.rs:13-91 */
        const cls_node* p = &o->nodes[prop_row];
        uint32_t n_nonleaf = 0;
        if (p->has_children) {
            child_rows = ensure(&S->child_rows, p->n_children + 1, 4); cand_rows = ensure(&S->cand_rows, p->n_children + 1, 4);
            for (uint32_t i = 0; i < p->n_children; i++)
                if (o->nodes[p->first_child + i].kind != CLS_KIND_LEAF) child_rows[n_nonleaf++] = p->first_child + i;
        }
        if (n_nonleaf == 0) {
            out->status = CLS_IDENTITY_FOUND; out->one = prop_one; out->rest = prop_rest; out->clade_id = p->id; return;
        }
        parent_row = prop_row; n_children = n_nonleaf;
    }
}

/* ---- "reference cost" mode ----------------------------------------------------------------
 * What the reference does per QUERY on top of the algorithm (SURVEY.md 3.3 hot spots i-ii), so that a timing of
 * this port can stand in for the unmodified Rust path (an estimate; never used for results):
 *   (i)  place_sequence.rs:77-80  `tree.kmers_map.to_owned()`: a deep clone of the whole index -- every bucket's
 *        HashMap<u64, HashSet<u64>> and every k-mer's HashSet<u64>, one allocation + copy each, dropped at the end;
 *   (ii) kmers_map.rs:58-62       MinimizerValue::get_overlapping_hashed_kmers collects ALL keys of every
 *        touched bucket into a fresh HashSet before intersecting with the query's hashes. */
static uint64_t refcost_overhead(const cls_oracle* o, const char* seq, uint64_t L) {
    uint64_t sink = 0;
    if (o->leaves_only) return 0; /* the explicit sets this mode would clone are never materialised */
    void** sets = malloc(sizeof(void*) * (o->n_kmers + 1));
    void** maps = malloc(sizeof(void*) * (o->n_buckets + 1));
    if (!sets || !maps) { free(sets); free(maps); return 0; }
    for (uint64_t b = 0; b < o->n_buckets; b++) {  /* (i) clone */
        const uint64_t lo = o->bucket_off[b], hi = o->bucket_off[b + 1];
        uint64_t* keys = malloc(16 * (hi - lo) + 16);  /* the bucket's map: key + pointer per entry */
        maps[b] = keys;
        for (uint64_t j = lo; j < hi; j++) {
            const uint64_t n = o->node_off[j + 1] - o->node_off[j];
            uint64_t* set = malloc(8 * n + 16);
            if (set) { memcpy(set, o->nodes_sorted + o->node_off[j], 8 * n); sink += set[n ? n - 1 : 0]; }
            sets[j] = set;
            if (keys) { keys[2 * (j - lo)] = o->kmer_hash[j]; keys[2 * (j - lo) + 1] = (uint64_t)(uintptr_t)set; }
        }
    }
    if (L >= o->k) {  /* (ii) key set of every bucket a query minimizer names */
        const uint64_t nf = L - o->k + 1;
        unsigned char* touched = calloc(o->n_buckets + 1, 1);
        char up[64];
        const uint64_t mm = o->m < o->k ? o->m : o->k;
        for (int strand = 0; strand < 2 && touched && mm <= sizeof up; strand++)
            for (uint64_t p = 0; p < nf; p++) {
                for (uint64_t t = 0; t < mm; t++) {
                    char c = strand == 0 ? seq[p + t] : seq[L - 1 - p - t];
                    if (c >= 'a' && c <= 'z') c -= 32;
                    if (strand) c = c == 'A' ? 'T' : c == 'T' ? 'A' : c == 'C' ? 'G' : c == 'G' ? 'C' : c;
                    up[t] = c;
                }
                const uint64_t mz = mm ? cls_oracle_murmur3_h1(up, mm) : 0;
                for (uint64_t b = 0; b < o->n_buckets; b++) if (o->bucket_key[b] == mz) touched[b] = 1;
            }
        for (uint64_t b = 0; touched && b < o->n_buckets; b++) {
            if (!touched[b]) continue;
            const uint64_t lo = o->bucket_off[b], hi = o->bucket_off[b + 1];
            uint64_t cap = 8;
            while (cap < 2 * (hi - lo)) cap <<= 1;
            uint64_t* tab = calloc(cap, 8);
            if (!tab) continue;
            for (uint64_t j = lo; j < hi; j++) {  /* HashSet<u64>::insert of every key */
                uint64_t h = o->kmer_hash[j] | 1, i = (h * 0x9E3779B97F4A7C15ull) & (cap - 1);
                while (tab[i] && tab[i] != h) i = (i + 1) & (cap - 1);
                tab[i] = h;
            }
            sink += tab[cap / 2];
            free(tab);
        }
        free(touched);
    }
    for (uint64_t j = 0; j < o->n_kmers; j++) free(sets[j]);
    for (uint64_t b = 0; b < o->n_buckets; b++) free(maps[b]);
    free(sets); free(maps);
    return sink;
}

void cls_oracle_set_reference_cost(cls_oracle* o, int on) { if (o) o->reference_cost = on != 0; }

typedef struct {
    const cls_oracle* o; const char* bases; const uint64_t* off; uint32_t lo, hi;
    int32_t max_iter; double cov; int rm; cls_placement* out; cls_query_stats* st;
} job;

static void* worker(void* arg) {
    job* j = arg;
    scratch S; memset(&S, 0, sizeof S);
    volatile uint64_t sink = 0;
    for (uint32_t i = j->lo; i < j->hi; i++) {
        if (j->o->reference_cost) sink += refcost_overhead(j->o, j->bases + j->off[i], j->off[i + 1] - j->off[i]);
        place_one(j->o, j->bases + j->off[i], j->off[i + 1] - j->off[i], j->max_iter, j->cov, j->rm, &S, &j->out[i], j->st ? &j->st[i] : NULL);
    }
    (void)sink;
    free(S.hashes.p); free(S.mins.p); free(S.ent_kmer.p); free(S.ent_hidx.p); free(S.sets.p); free(S.child_rows.p); free(S.cand_rows.p); free(S.buf.p);
    return NULL;
}

int cls_oracle_place_batch(const cls_oracle* o, const char* bases, const uint64_t* offsets, uint32_t n,
                           const cls_params* p, int n_threads, cls_placement* out, cls_query_stats* stats) {
    if (!o || (!bases && n) || !offsets || !out) return CLS_E_INVALID_ARG;
    /* place_sequence.rs:64-75 */
    int rm = (p && (p->flags & CLS_HAS_REMOVE_INTERSECTION)) ? (p->remove_intersection != 0) : 0;
    int32_t max_iter = (p && (p->flags & CLS_HAS_MAX_ITERATIONS)) ? p->max_iterations : 1000;
    double cov = 0.7;
    if (p && (p->flags & CLS_HAS_MIN_MATCH_COVERAGE)) { cov = p->min_match_coverage; if (cov > 1.0) cov = 1.0; else if (cov < 0.0) cov = 0.0; }
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > n) n_threads = n ? (int)n : 1;
    pthread_t* th = malloc(sizeof(pthread_t) * (size_t)n_threads);
    job* jobs = malloc(sizeof(job) * (size_t)n_threads);
    for (int t = 0; t < n_threads; t++) {
        jobs[t] = (job){o, bases, offsets, (uint32_t)((uint64_t)n * t / n_threads), (uint32_t)((uint64_t)n * (t + 1) / n_threads), max_iter, cov, rm, out, stats};
        if (n_threads == 1) worker(&jobs[t]); else pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    if (n_threads > 1) for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
    return CLS_OK;
}
