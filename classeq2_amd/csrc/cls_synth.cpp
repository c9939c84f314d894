// Deterministic synthetic workload generator; see cls_synth.h.
#include "cls_synth.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <parallel/algorithm>
#include <string>
#include <vector>

#include "cls_murmur.h"

namespace {

thread_local std::string g_err;

struct Rng {  // splitmix64-seeded xoshiro256**
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) {
        for (auto& v : s) v = splitmix(seed);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    uint64_t below(uint64_t n) { return (uint64_t)(((__uint128_t)next() * n) >> 64); }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

const char BASES[4] = {'A', 'C', 'G', 'T'};
inline char comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; }
inline char other_base(char c, Rng& r) {
    int cur = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3;
    return BASES[(cur + 1 + (int)r.below(3)) & 3];
}

struct TNode {
    int parent = -1;
    std::vector<int> ch;
    bool leaf = false;
    int depth = 0;
};

struct Rec {
    uint64_t hash;
    uint32_t leaf;
    uint32_t pos_strand;
};

}  // namespace

struct cls_synth_db {
    cls_synth_cfg cfg;
    std::vector<cls_node> nodes;        // BFS rows
    std::vector<uint32_t> leaf_row;     // leaf i (DFS order) -> row
    std::vector<char> leaf_seq;         // n_leaves * ref_len
    std::vector<uint64_t> bucket_key, bucket_kmer_off, kmer_hash, kmer_node_off, node_ids;
    uint32_t max_depth = 0;
    cls_db_desc desc;
};

static void mutate_geometric(std::vector<char>& seq, double p, Rng& rng) {
    if (p <= 0) return;
    const double lg = log(1.0 - p);
    size_t i = 0, n = seq.size();
    for (;;) {
        double u = rng.unit();
        if (u <= 0) u = 1e-300;
        double skip = floor(log(u) / lg);
        if (skip >= (double)(n - i)) break;
        i += (size_t)skip;
        if (i >= n) break;
        seq[i] = other_base(seq[i], rng);
        ++i;
        if (i >= n) break;
    }
}

extern "C" const char* cls_synth_last_error(void) { return g_err.c_str(); }

extern "C" int cls_synth_db_create(const cls_synth_cfg* cfg, cls_synth_db** out) {
    if (!cfg || !out || cfg->n_leaves < 2 || cfg->k_size == 0 || cfg->ref_len < cfg->k_size) {
        g_err = "cls_synth_db_create: invalid configuration";
        return CLS_E_INVALID_ARG;
    }
    try {
        auto* S = new cls_synth_db();
        S->cfg = *cfg;
        const uint32_t NL = cfg->n_leaves, L = cfg->ref_len, K = cfg->k_size;
        const uint32_t M = std::min(cfg->m_size, cfg->k_size);
        int threads = cfg->threads ? (int)cfg->threads : omp_get_max_threads();

        // ---- 1. binary topology by recursive random splits ------------------
        std::vector<TNode> T;
        T.reserve(2 * (size_t)NL);
        T.emplace_back();
        {
            Rng rng(cfg->seed_tree);
            struct Item { int node; uint32_t n; };
            std::vector<Item> st{{0, NL}};
            while (!st.empty()) {
                Item it = st.back();
                st.pop_back();
                if (it.n == 1) { T[it.node].leaf = true; continue; }
                uint32_t n = it.n, a;
                int d = T[it.node].depth;
                uint32_t need = 0;
                while ((1u << need) < n) ++need;  // ceil(log2 n)
                if (cfg->max_depth && (uint32_t)d + need + 2 >= cfg->max_depth) {
                    a = n / 2;  // finish balanced inside the depth cap
                } else if (cfg->deep && (cfg->deep >= 2 || rng.unit() < 0.9)) {  // deep = 2: a pure ladder (depth ~ n/2), for tests
                    uint32_t small = 1 + (uint32_t)rng.below(3);
                    if (small > n - 1) small = n - 1;
                    a = rng.below(2) ? small : n - small;
                } else {
                    a = 1 + (uint32_t)rng.below(n - 1);
                }
                int ca = (int)T.size();
                T.emplace_back(); T.emplace_back();
                T[ca].parent = it.node; T[ca + 1].parent = it.node;
                T[ca].depth = T[ca + 1].depth = d + 1;
                T[it.node].ch = {ca, ca + 1};
                st.push_back({ca + 1, n - a});
                st.push_back({ca, a});
            }
            // ---- 1b. dissolve internal nodes into their parent (polytomies),
            // like Tree::sanitize (core/src/domain/dtos/tree.rs:248-285) does
            // for low-support branches: grandchildren take the child's place.
            if (cfg->collapse_prob > 0) {
                std::vector<char> dissolve(T.size(), 0);
                for (size_t i = 1; i < T.size(); ++i)
                    if (!T[i].leaf && rng.unit() < cfg->collapse_prob) dissolve[i] = 1;
                // post-order so that children are final before the parent splices them
                std::vector<int> order;
                std::vector<int> stk{0};
                while (!stk.empty()) {
                    int v = stk.back(); stk.pop_back(); order.push_back(v);
                    for (int c : T[v].ch) stk.push_back(c);
                }
                for (auto itv = order.rbegin(); itv != order.rend(); ++itv) {
                    int v = *itv;
                    if (T[v].leaf) continue;
                    std::vector<int> nc;
                    for (int c : T[v].ch) {
                        if (!T[c].leaf && dissolve[c]) for (int g : T[c].ch) nc.push_back(g);
                        else nc.push_back(c);
                    }
                    T[v].ch.swap(nc);
                }
                for (size_t i = 1; i < T.size(); ++i) if (dissolve[i]) T[i].ch.clear();
                for (size_t v = 0; v < T.size(); ++v) for (int c : T[v].ch) T[c].parent = (int)v;
            }
        }
        // ---- 2. preorder ids, depths, BFS rows ------------------------------
        std::vector<int> pre(T.size(), -1), order;  // order: DFS preorder list of live nodes
        {
            std::vector<int> stk{0};
            T[0].depth = 0;
            while (!stk.empty()) {
                int v = stk.back(); stk.pop_back();
                pre[v] = (int)order.size(); order.push_back(v);
                for (auto it = T[v].ch.rbegin(); it != T[v].ch.rend(); ++it) {
                    T[*it].depth = T[v].depth + 1; stk.push_back(*it);
                }
            }
        }
        const uint32_t NN = (uint32_t)order.size();
        std::vector<int> row(T.size(), -1), row2node;
        row2node.reserve(NN);
        row2node.push_back(0); row[0] = 0;
        for (size_t r = 0; r < row2node.size(); ++r)
            for (int c : T[row2node[r]].ch) { row[c] = (int)row2node.size(); row2node.push_back(c); }
        S->nodes.resize(NN);
        auto id_of = [&](int v) { return cfg->id_offset + (cfg->id_stride ? cfg->id_stride : 1) * (uint64_t)pre[v]; };
        for (uint32_t r = 0; r < NN; ++r) {
            int v = row2node[r];
            cls_node& n = S->nodes[r];
            memset(&n, 0, sizeof n);
            n.id = id_of(v);
            n.parent = T[v].parent < 0 ? CLS_NO_PARENT : id_of(T[v].parent);
            n.kind = v == 0 ? CLS_KIND_ROOT : (T[v].leaf ? CLS_KIND_LEAF : CLS_KIND_NODE);
            n.n_children = (uint32_t)T[v].ch.size();
            n.first_child = n.n_children ? (uint32_t)row[T[v].ch[0]] : 0;
            n.has_children = n.n_children ? 1 : 0;
            S->max_depth = std::max<uint32_t>(S->max_depth, (uint32_t)T[v].depth);
        }
        // ---- 3. evolve reference sequences down the tree ---------------------
        std::vector<int> leaf_index(T.size(), -1);
        for (int v : order) if (T[v].leaf) { leaf_index[v] = (int)S->leaf_row.size(); S->leaf_row.push_back((uint32_t)row[v]); }
        S->leaf_seq.resize((size_t)NL * L);
        {
            std::vector<std::vector<char>> at_depth(S->max_depth + 1);
            for (int v : order) {  // preorder: the parent's sequence is still at depth-1
                int d = T[v].depth;
                if (v == 0) {
                    Rng rng(cfg->seed_refseq);
                    at_depth[0].resize(L);
                    for (auto& c : at_depth[0]) c = BASES[rng.below(4)];
                } else {
                    at_depth[d] = at_depth[d - 1];
                    uint64_t sd = cfg->seed_refseq * 0x9e3779b97f4a7c15ULL + (uint64_t)pre[v] + 1;
                    Rng rng(sd);
                    mutate_geometric(at_depth[d], cfg->edge_sub_rate, rng);
                }
                if (T[v].leaf) memcpy(&S->leaf_seq[(size_t)leaf_index[v] * L], at_depth[d].data(), L);
            }
        }
        // ---- 4. k-mer records: every leaf, forward + reverse complement ------
        const uint32_t NP = L - K + 1;
        std::vector<Rec> recs((size_t)NL * NP * 2);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16)
        for (uint32_t li = 0; li < NL; ++li) {
            const char* s = &S->leaf_seq[(size_t)li * L];
            Rec* o = &recs[(size_t)li * NP * 2];
            for (uint32_t p = 0; p < NP; ++p) {
                o[p] = {cls::murmur3_h1([&](uint32_t i) { return (uint8_t)s[p + i]; }, K), li, p << 1};
                o[NP + p] = {cls::murmur3_h1([&](uint32_t i) { return (uint8_t)comp(s[L - 1 - p - i]); }, K), li, (p << 1) | 1};
            }
        }
        omp_set_num_threads(threads);
        __gnu_parallel::sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) {
            return a.hash != b.hash ? a.hash < b.hash : a.leaf < b.leaf;
        });
        // ---- 5. groups of equal hash; bucket key from a representative -------
        struct Grp { uint64_t bkey, hash; size_t lo, hi; };
        std::vector<Grp> grps;
        for (size_t i = 0; i < recs.size();) {
            size_t j = i;
            while (j < recs.size() && recs[j].hash == recs[i].hash) ++j;
            grps.push_back({0, recs[i].hash, i, j});
            i = j;
        }
#pragma omp parallel for num_threads(threads)
        for (size_t g = 0; g < grps.size(); ++g) {
            const Rec& r = recs[grps[g].lo];
            const char* s = &S->leaf_seq[(size_t)r.leaf * L];
            uint32_t p = r.pos_strand >> 1;
            if (cfg->m_size == 0) grps[g].bkey = 0;  // MinimizerKey(0), kmers_map.rs:131-134
            else if (r.pos_strand & 1) grps[g].bkey = cls::murmur3_h1([&](uint32_t i) { return (uint8_t)comp(s[L - 1 - p - i]); }, M);
            else grps[g].bkey = cls::murmur3_h1([&](uint32_t i) { return (uint8_t)s[p + i]; }, M);
        }
        // hash-map order is arbitrary in the reference: lay buckets / k-mers out
        // in a scrambled but deterministic order so no consumer leans on sortedness
        __gnu_parallel::sort(grps.begin(), grps.end(), [](const Grp& a, const Grp& b) {
            uint64_t ka = cls::fmix64(a.bkey ^ 0x5bd1e995), kb = cls::fmix64(b.bkey ^ 0x5bd1e995);
            if (ka != kb) return ka < kb;
            return cls::fmix64(a.hash) < cls::fmix64(b.hash);
        });
        const size_t G = grps.size();
        S->kmer_hash.resize(G);
        S->kmer_node_off.assign(G + 1, 0);
        for (size_t g = 0; g < G; ++g) {
            S->kmer_hash[g] = grps[g].hash;
            if (g == 0 || grps[g].bkey != grps[g - 1].bkey) { S->bucket_key.push_back(grps[g].bkey); S->bucket_kmer_off.push_back(g); }
        }
        S->bucket_kmer_off.push_back(G);
        // ---- 6. node set = union of root->leaf paths (build_database/mod.rs:160-169)
        std::vector<uint32_t> parent_row(NN);
        for (uint32_t r = 0; r < NN; ++r) parent_row[r] = T[row2node[r]].parent < 0 ? UINT32_MAX : (uint32_t)row[T[row2node[r]].parent];
        for (int pass = 0; pass < 2; ++pass) {
            if (cfg->tips_only) {  // only the distinct leaves of every k-mer: the node set is the union of their root paths
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4096)
                for (size_t g = 0; g < G; ++g) {
                    uint64_t cnt = 0;
                    uint64_t* dst = pass ? &S->node_ids[S->kmer_node_off[g]] : nullptr;
                    uint32_t prev = UINT32_MAX;
                    for (size_t i = grps[g].lo; i < grps[g].hi; ++i) {  // (sorted by leaf inside a group)
                        if (recs[i].leaf == prev) continue;
                        prev = recs[i].leaf;
                        if (pass) dst[cnt] = S->nodes[S->leaf_row[prev]].id;
                        ++cnt;
                    }
                    if (!pass) S->kmer_node_off[g + 1] = cnt;
                }
                if (!pass) {
                    for (size_t g = 0; g < G; ++g) S->kmer_node_off[g + 1] += S->kmer_node_off[g];
                    S->node_ids.resize(S->kmer_node_off[G]);
                }
                continue;
            }
#pragma omp parallel num_threads(threads)
            {
                std::vector<uint64_t> stamp(NN, 0);
#pragma omp for schedule(dynamic, 1024)
                for (size_t g = 0; g < G; ++g) {
                    uint64_t tag = (uint64_t)g + 1, cnt = 0;
                    uint64_t* dst = pass ? &S->node_ids[S->kmer_node_off[g]] : nullptr;
                    uint32_t prev = UINT32_MAX;
                    for (size_t i = grps[g].lo; i < grps[g].hi; ++i) {
                        if (recs[i].leaf == prev) continue;
                        prev = recs[i].leaf;
                        for (uint32_t r = S->leaf_row[prev]; r != UINT32_MAX && stamp[r] != tag; r = parent_row[r]) {
                            stamp[r] = tag;
                            if (pass) dst[cnt] = S->nodes[r].id;
                            ++cnt;
                        }
                    }
                    if (!pass) S->kmer_node_off[g + 1] = cnt;
                }
            }
            if (!pass) {
                for (size_t g = 0; g < G; ++g) S->kmer_node_off[g + 1] += S->kmer_node_off[g];
                S->node_ids.resize(S->kmer_node_off[G]);
            }
        }
        cls_db_desc& d = S->desc;
        memset(&d, 0, sizeof d);
        d.abi_version = CLS_ABI_VERSION;
        d.n_nodes = NN;
        d.nodes = S->nodes.data();
        d.k_size = cfg->k_size;
        d.m_size = cfg->m_size;
        d.n_buckets = S->bucket_key.size();
        d.bucket_key = S->bucket_key.data();
        d.bucket_kmer_off = S->bucket_kmer_off.data();
        d.n_kmers = G;
        d.kmer_hash = S->kmer_hash.data();
        d.kmer_node_off = S->kmer_node_off.data();
        d.node_ids = S->node_ids.data();
        d.node_set_kind = cfg->tips_only ? CLS_SETS_LEAVES : CLS_SETS_EXPLICIT;
        *out = S;
        return CLS_OK;
    } catch (const std::exception& e) {
        g_err = std::string("cls_synth_db_create: ") + e.what();
        return CLS_E_NOMEM;
    }
}

extern "C" void cls_synth_db_destroy(cls_synth_db* s) { delete s; }
extern "C" const cls_db_desc* cls_synth_db_desc(const cls_synth_db* s) { return &s->desc; }
extern "C" uint32_t cls_synth_n_leaves(const cls_synth_db* s) { return (uint32_t)s->leaf_row.size(); }
extern "C" uint32_t cls_synth_max_depth(const cls_synth_db* s) { return s->max_depth; }
extern "C" uint64_t cls_synth_leaf_id(const cls_synth_db* s, uint32_t i) { return s->nodes[s->leaf_row[i]].id; }
extern "C" const char* cls_synth_leaf_seq(const cls_synth_db* s, uint32_t i) { return &s->leaf_seq[(size_t)i * s->cfg.ref_len]; }

// experiment hook: when set, cls_synth_reads also records each read's start coordinate here
static uint32_t* g_truth_pos = nullptr;
extern "C" void cls_synth_set_truth_pos(uint32_t* p) { g_truth_pos = p; }

extern "C" int cls_synth_reads(const cls_synth_db* s, uint64_t seed, uint64_t first, uint32_t n_reads,
                               uint32_t read_len, double err, double frac_random, char* bases,
                               uint64_t* offsets, uint32_t* truth_leaf) {
    if (!s || !bases || !offsets || read_len == 0 || read_len > s->cfg.ref_len) {
        g_err = "cls_synth_reads: invalid argument";
        return CLS_E_INVALID_ARG;
    }
    const uint32_t L = s->cfg.ref_len, NL = (uint32_t)s->leaf_row.size();
    int threads = s->cfg.threads ? (int)s->cfg.threads : omp_get_max_threads();
#pragma omp parallel for num_threads(threads) schedule(static, 4096)
    for (uint32_t i = 0; i < n_reads; ++i) {
        Rng rng(seed * 0xd1342543de82ef95ULL + (first + i) * 0x9e3779b97f4a7c15ULL + 1);
        char* o = bases + (size_t)i * read_len;
        if (rng.unit() < frac_random) {
            for (uint32_t j = 0; j < read_len; ++j) o[j] = BASES[rng.below(4)];
            if (truth_leaf) truth_leaf[i] = UINT32_MAX;
        } else {
            uint32_t li = (uint32_t)rng.below(NL);
            uint32_t st = (uint32_t)rng.below(L - read_len + 1);
            const char* src = &s->leaf_seq[(size_t)li * L + st];
            if (rng.below(2)) for (uint32_t j = 0; j < read_len; ++j) o[j] = comp(src[read_len - 1 - j]);
            else memcpy(o, src, read_len);
            for (uint32_t j = 0; j < read_len; ++j) if (rng.unit() < err) o[j] = other_base(o[j], rng);
            if (truth_leaf) truth_leaf[i] = li;
            if (g_truth_pos) g_truth_pos[i] = st;
        }
    }
    for (uint64_t i = 0; i <= n_reads; ++i) offsets[i] = i * (uint64_t)read_len;
    return CLS_OK;
}
