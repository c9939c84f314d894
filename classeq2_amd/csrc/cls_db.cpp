// Host-side encoder: borrowed `cls_db_desc` (the reference's Tree + KmersMap,
// flattened) -> the HBM layout of cls_device.h.  Runs once per database at
// cls_db_create(); nothing here is on the timed path.
#include "cls_db.h"

#include "cls_murmur.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>

namespace cls {

namespace {

// An exception inside a worker thread would terminate the process: the workers catch, the caller rethrows
// after the join (encode_db's callers turn it into CLS_E_NOMEM / CLS_E_INTERNAL).
struct WorkerError {
    std::mutex mu;
    std::exception_ptr first;
    void capture() { std::lock_guard<std::mutex> g(mu); if (!first) first = std::current_exception(); }
    void rethrow() { if (first) std::rethrow_exception(first); }
};

void parallel_chunks(uint64_t n, unsigned n_threads, const std::function<void(unsigned, uint64_t, uint64_t)>& fn) {
    if (n_threads <= 1 || n < 4096) {
        fn(0, 0, n);
        return;
    }
    WorkerError we;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < n_threads; ++t)
        th.emplace_back([=, &fn, &we] { try { fn(t, n * t / n_threads, n * (t + 1) / n_threads); } catch (...) { we.capture(); } });
    for (auto& x : th) x.join();
    we.rethrow();
}

// sort `v` with `cmp` on up to n_threads threads: sorted runs, then pairwise merges
template <class T, class Cmp>
void parallel_sort(std::vector<T>& v, unsigned n_threads, Cmp cmp) {
    const uint64_t n = v.size();
    unsigned parts = 1;
    while (parts * 2 <= n_threads && n / (parts * 2) >= 65536) parts *= 2;
    if (parts == 1) { std::sort(v.begin(), v.end(), cmp); return; }
    auto bound = [&](unsigned i) { return v.begin() + (ptrdiff_t)(n * i / parts); };
    WorkerError we;
    {
        std::vector<std::thread> th;
        for (unsigned i = 0; i < parts; ++i) th.emplace_back([&, i] { try { std::sort(bound(i), bound(i + 1), cmp); } catch (...) { we.capture(); } });
        for (auto& x : th) x.join();
    }
    we.rethrow();
    for (unsigned width = 1; width < parts; width *= 2) {
        std::vector<std::thread> th;
        for (unsigned i = 0; i + width < parts; i += 2 * width)
            th.emplace_back([&, i, width] {
                try { std::inplace_merge(bound(i), bound(i + width), bound(std::min(parts, i + 2 * width)), cmp); } catch (...) { we.capture(); }  // (inplace_merge allocates a buffer)
            });
        for (auto& x : th) x.join();
        we.rethrow();
    }
}

}  // namespace

int encode_db(const cls_db_desc* d, EncodedDb& E, std::string& err) {
    // ---- 0. argument checks ---------------------------------------------------
    if (!d) { err = "null descriptor"; return CLS_E_INVALID_ARG; }
    if (d->abi_version != CLS_ABI_VERSION) { err = "cls_db_desc.abi_version mismatch"; return CLS_E_INVALID_ARG; }
    if (d->n_nodes == 0 || !d->nodes) { err = "empty node table"; return CLS_E_BAD_TREE; }
    if (d->k_size == 0 || d->k_size > MAX_K) { err = "kSize must be in [1, " + std::to_string(MAX_K) + "]"; return CLS_E_BAD_DB; }
    if (d->n_buckets >= (1ULL << LOC_BUCKET_BITS)) { err = "too many minimizer buckets (>= 2^24)"; return CLS_E_BAD_DB; }
    if ((d->n_buckets && (!d->bucket_key || !d->bucket_kmer_off)) || (d->n_kmers && (!d->kmer_hash || !d->kmer_node_off)) ) {
        err = "null k-mer map array"; return CLS_E_INVALID_ARG;
    }
    if (d->n_buckets) {
        if (d->bucket_kmer_off[0] != 0 || d->bucket_kmer_off[d->n_buckets] != d->n_kmers) { err = "bucket_kmer_off does not span [0, n_kmers]"; return CLS_E_BAD_DB; }
        for (uint64_t b = 0; b < d->n_buckets; ++b)
            if (d->bucket_kmer_off[b] > d->bucket_kmer_off[b + 1]) { err = "bucket_kmer_off not monotone"; return CLS_E_BAD_DB; }
    } else if (d->n_kmers) { err = "k-mers without buckets"; return CLS_E_BAD_DB; }
    for (uint64_t j = 0; j < d->n_kmers; ++j)
        if (d->kmer_node_off[j] > d->kmer_node_off[j + 1]) { err = "kmer_node_off not monotone"; return CLS_E_BAD_DB; }
    if (d->n_kmers && d->kmer_node_off[d->n_kmers] && !d->node_ids) { err = "null node_ids"; return CLS_E_INVALID_ARG; }

    const uint32_t N = d->n_nodes;
    // ---- 1. validate the row table is a tree rooted at row 0 -------------------
    std::vector<uint32_t> order;  // engine row -> caller row (BFS, non-LEAF children first)
    order.reserve(N);
    {
        std::vector<uint8_t> seen(N, 0);
        order.push_back(0);
        seen[0] = 1;
        for (size_t i = 0; i < order.size(); ++i) {
            const cls_node& n = d->nodes[order[i]];
            if (n.kind > CLS_KIND_LEAF) { err = "node kind out of range"; return CLS_E_BAD_TREE; }
            if (n.n_children == 0) continue;
            if (!n.has_children) { err = "n_children > 0 with has_children == 0"; return CLS_E_BAD_TREE; }
            if ((uint64_t)n.first_child + n.n_children > N || n.first_child == 0) { err = "child rows out of range"; return CLS_E_BAD_TREE; }
            for (int pass = 0; pass < 2; ++pass)  // non-LEAF children first, each group in Clade.children order
                for (uint32_t c = n.first_child; c < n.first_child + n.n_children; ++c) {
                    bool leaf = d->nodes[c].kind == CLS_KIND_LEAF;
                    if (leaf != (pass == 1)) continue;
                    if (seen[c]) { err = "row is the child of two parents (not a tree)"; return CLS_E_BAD_TREE; }
                    seen[c] = 1;
                    order.push_back(c);
                }
        }
        if (order.size() != N) { err = "rows unreachable from the root"; return CLS_E_BAD_TREE; }
    }
    std::vector<uint32_t> new_row(N);
    for (uint32_t r = 0; r < N; ++r) new_row[order[r]] = r;
    E.nodes.assign(N, DNode{});
    for (uint32_t r = 0; r < N; ++r) {
        const cls_node& n = d->nodes[order[r]];
        DNode& o = E.nodes[r];
        o.id = n.id;
        o.split = 0;
        o.flags = (n.has_children ? 1u : 0u) | (std::min<uint32_t>(n.n_children, (1u << 24) - 1) << 8);
        o.n_nonleaf = 0;
        o.first_child = 0;
        if (n.n_children) {
            uint32_t first = UINT32_MAX;
            for (uint32_t c = n.first_child; c < n.first_child + n.n_children; ++c) {
                first = std::min(first, new_row[c]);
                if (d->nodes[c].kind != CLS_KIND_LEAF) o.n_nonleaf++;
            }
            o.first_child = first;
            E.max_nonleaf_arity = std::max(E.max_nonleaf_arity, o.n_nonleaf);
        }
    }
    // ---- 2. DFS pre-order + subtree sizes in the engine's child order ---------
    std::vector<uint32_t> parent_pre(N, UINT32_MAX), size_by_pre(N, 1), row_by_pre(N), depth_by_pre(N, 0);
    std::vector<uint8_t> leaf_by_pre(N, 0);
    {
        struct Fr { uint32_t row, next; };
        std::vector<Fr> st;
        st.push_back({0, 0});
        uint32_t counter = 0;
        E.nodes[0].pre = counter++;
        row_by_pre[0] = 0;
        uint32_t depth = 0;
        while (!st.empty()) {
            Fr& f = st.back();
            DNode& n = E.nodes[f.row];
            if (f.next < d->nodes[order[f.row]].n_children) {
                uint32_t c = n.first_child + f.next++;
                E.nodes[c].pre = counter++;
                row_by_pre[E.nodes[c].pre] = c;
                parent_pre[E.nodes[c].pre] = n.pre;
                depth_by_pre[E.nodes[c].pre] = (uint32_t)st.size();
                st.push_back({c, 0});
                depth = std::max<uint32_t>(depth, (uint32_t)st.size() - 1);
            } else {
                n.size = counter - n.pre;
                size_by_pre[n.pre] = n.size;
                if (d->nodes[order[f.row]].n_children) n.split = E.nodes[n.first_child].pre + E.nodes[n.first_child].size;
                st.pop_back();
            }
        }
        E.max_depth = depth;
        for (uint32_t r = 0; r < N; ++r) leaf_by_pre[E.nodes[r].pre] = d->nodes[order[r]].kind == CLS_KIND_LEAF;
    }
    // ---- 3. clade id -> pre ------------------------------------------------------
    std::vector<std::pair<uint64_t, uint32_t>> id2pre(N);
    for (uint32_t r = 0; r < N; ++r) id2pre[r] = {E.nodes[r].id, E.nodes[r].pre};
    std::sort(id2pre.begin(), id2pre.end());
    for (uint32_t r = 1; r < N; ++r)
        if (id2pre[r].first == id2pre[r - 1].first) { err = "duplicate clade id " + std::to_string(id2pre[r].first); return CLS_E_BAD_TREE; }
    auto pre_of = [&](uint64_t id) -> uint32_t {
        auto it = std::lower_bound(id2pre.begin(), id2pre.end(), std::make_pair(id, (uint32_t)0));
        return (it != id2pre.end() && it->first == id) ? it->second : UINT32_MAX;
    };
    // ---- 4. per k-mer: node set -> sorted pre list -> tips / explicit list -------
    const uint64_t NK = d->n_kmers;
    std::vector<uint32_t> bucket_of(NK);
    for (uint64_t b = 0; b < d->n_buckets; ++b)
        for (uint64_t j = d->bucket_kmer_off[b]; j < d->bucket_kmer_off[b + 1]; ++j) bucket_of[j] = (uint32_t)b;
    unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<std::vector<uint32_t>> chunk_words(nt);
    std::vector<uint64_t> local_off(NK);  // offset of k-mer j inside its chunk
    std::vector<std::pair<uint64_t, uint64_t>> chunk_range(nt, {0, 0});
    std::atomic<uint64_t> n_closed{0};
    parallel_chunks(NK, nt, [&](unsigned t, uint64_t lo, uint64_t hi) {
        chunk_range[t] = {lo, hi};
        std::vector<uint32_t>& W = chunk_words[t];
        std::vector<uint32_t> P, stk;
        uint64_t closed_cnt = 0;
        for (uint64_t j = lo; j < hi; ++j) {
            P.clear();
            for (uint64_t i = d->kmer_node_off[j]; i < d->kmer_node_off[j + 1]; ++i) {
                uint32_t p = pre_of(d->node_ids[i]);
                if (p != UINT32_MAX) P.push_back(p);  // ids that are no clade of this tree can never be asked for
            }
            std::sort(P.begin(), P.end());
            P.erase(std::unique(P.begin(), P.end()), P.end());
            bool has_root = !P.empty() && P[0] == 0;
            bool closed = true;
            stk.clear();
            uint32_t n_leaf = 0;
            for (uint32_t x : P) {
                n_leaf += leaf_by_pre[x];
                if (!closed) continue;
                while (!stk.empty() && x >= stk.back() + size_by_pre[stk.back()]) stk.pop_back();
                if (x != 0 && (stk.empty() || stk.back() != parent_pre[x])) closed = false;
                stk.push_back(x);
            }
            local_off[j] = W.size();
            size_t hdr = W.size();
            W.push_back(0);
            W.push_back(n_leaf);
            uint32_t n_el = 0;
            for (size_t i = 0; i < P.size(); ++i) {
                uint32_t x = P[i];
                if (x == 0) continue;  // the root is carried by POST_HAS_ROOT
                if (closed && i + 1 < P.size() && P[i + 1] < x + size_by_pre[x]) continue;  // has a member below: not a tip
                W.push_back(x);
                ++n_el;
            }
            W[hdr] = n_el | (has_root ? POST_HAS_ROOT : 0) | (closed ? POST_CLOSED : 0);
            closed_cnt += closed;
        }
        n_closed += closed_cnt;
    });
    // chunks that were not run (single-thread fallback) stay empty
    std::vector<uint64_t> chunk_base(nt + 1, 0);
    for (unsigned t = 0; t < nt; ++t) chunk_base[t + 1] = chunk_base[t] + chunk_words[t].size();
    const uint64_t total_words = chunk_base[nt];
    if (total_words >= (1ULL << 40)) { err = "postings exceed 2^40 words"; return CLS_E_BAD_DB; }
    E.postings.resize(total_words + 4);  // small tail pad: speculative reads past an empty list stay in bounds
    for (unsigned t = 0; t < nt; ++t) {
        std::copy(chunk_words[t].begin(), chunk_words[t].end(), E.postings.begin() + chunk_base[t]);
        std::vector<uint32_t>().swap(chunk_words[t]);
    }
    std::vector<uint64_t> kmer_off(NK);
    for (unsigned t = 0; t < nt; ++t)
        for (uint64_t j = chunk_range[t].first; j < chunk_range[t].second; ++j) kmer_off[j] = chunk_base[t] + local_off[j];
    // ---- 4b. split-tree form when the whole index allows it (see cls_device.h) -------
    const uint64_t* d_kmer_hash = d->kmer_hash;
    E.strictly_binary = true;
    for (uint32_t r = 0; r < N; ++r)
        if (d->nodes[order[r]].n_children != 0 && d->nodes[order[r]].n_children != 2) { E.strictly_binary = false; break; }
    // CLS_FORCE_LIST=1 keeps the sorted-list form (A/B experiments only)
    E.format = (n_closed.load() == NK && getenv("CLS_FORCE_LIST") == nullptr) ? FMT_SPLIT : FMT_LIST;
    if (E.format == FMT_SPLIT) {
        // k-mers with the SAME tip list (neighbouring k-mers of a conserved region) share one split tree:
        // group them exactly (signature first, then the lists themselves).
        std::vector<uint64_t> sig(NK);
        parallel_chunks(NK, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
            for (uint64_t j = lo; j < hi; ++j) {
                const uint32_t* w = &E.postings[kmer_off[j]];
                const uint32_t n = w[0] & POST_LEN_MASK;
                uint64_t h = 0x9E3779B97F4A7C15ull ^ n;
                for (uint32_t i = 0; i < n; ++i) h = fmix64(h ^ w[POST_HEADER_WORDS + i]) + 0x632BE59BD9B4E019ull;
                sig[j] = h;
            }
        });
        std::vector<uint32_t> by_set(NK);
        if (NK >= (1ULL << 32)) { err = "more than 2^32 k-mers"; return CLS_E_BAD_DB; }
        for (uint64_t j = 0; j < NK; ++j) by_set[j] = (uint32_t)j;
        auto tips_of = [&](uint32_t j, uint32_t& n) { const uint32_t* w = &E.postings[kmer_off[j]]; n = w[0] & POST_LEN_MASK; return w + POST_HEADER_WORDS; };
        auto cmp = [&](uint32_t a, uint32_t b) {
            if (sig[a] != sig[b]) return sig[a] < sig[b];
            uint32_t na, nb;
            const uint32_t* ta = tips_of(a, na);
            const uint32_t* tb = tips_of(b, nb);
            if (na != nb) return na < nb;
            const int c = na ? memcmp(ta, tb, (size_t)na * 4) : 0;
            return c != 0 ? c < 0 : a < b;
        };
        parallel_sort(by_set, nt, cmp);
        std::vector<uint32_t> set_of(NK), set_rep;  // k-mer -> set, set -> a k-mer that holds its tip list
        std::vector<uint64_t> set_rec;                // set -> first split-node record
        uint64_t n_recs = SPLIT_FIRST_REC + SPLIT_HEADER_RECS * NK;  // records 0/1: the dummy "no k-mer" header; then every k-mer's header
        for (uint64_t i = 0; i < NK; ++i) {
            const uint32_t j = by_set[i];
            bool same = false;
            if (i) {
                const uint32_t p = by_set[i - 1];
                uint32_t na, nb;
                const uint32_t* ta = tips_of(p, na);
                const uint32_t* tb = tips_of(j, nb);
                same = sig[p] == sig[j] && na == nb && (na == 0 || memcmp(ta, tb, (size_t)na * 4) == 0);
            }
            if (!same) {
                uint32_t n;
                (void)tips_of(j, n);
                set_rep.push_back(j);
                set_rec.push_back(n_recs);
                n_recs += n ? n - 1 : 0;
            }
            set_of[j] = (uint32_t)(set_rep.size() - 1);
        }
        std::vector<uint64_t>().swap(sig);
        std::vector<uint32_t>().swap(by_set);
        if (n_recs >= (1ULL << 31)) { err = "split-tree postings exceed 2^31 records"; return CLS_E_BAD_DB; }
        const uint64_t NS = set_rep.size();
        E.n_sets = NS;
        std::vector<uint32_t> recs((n_recs + 1) * 4, 0);
        recs[2] = 0xFFFFFFFFu;  // dummy header {0, 0, first tip = MAX, last tip = 0}: decodes to "inactive"
        std::vector<uint32_t> set_root(NS, 0);
        parallel_chunks(NS, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
            std::vector<uint32_t> d, stk, L, R, pos, span_lo, span_hi;
            for (uint64_t g = lo; g < hi; ++g) {
                uint32_t n;
                const uint32_t* tip = tips_of(set_rep[g], n);
                const uint64_t base = set_rec[g];
                d.assign(n, 0); L.assign(n, 0); R.assign(n, 0);
                for (uint32_t i = 1; i < n; ++i) {  // depth of LCA(tip[i-1], tip[i])
                    uint32_t a = tip[i - 1];
                    while (!(tip[i] < a + size_by_pre[a])) a = parent_pre[a];
                    d[i] = depth_by_pre[a];
                }
                stk.clear();
                for (uint32_t i = 1; i < n; ++i) {  // Cartesian tree, shallowest LCA on top
                    uint32_t last = 0;
                    while (!stk.empty() && d[stk.back()] > d[i]) { last = stk.back(); stk.pop_back(); }
                    L[i] = last;
                    if (!stk.empty()) R[stk.back()] = i;
                    stk.push_back(i);
                }
                const uint32_t root = stk.empty() ? 0 : stk.front();
                // memory order of the split nodes: DFS pre-order, the child that parts MORE tips first
                // (a read's walk follows the heavier side more often, so consecutive steps tend to
                // share a 64-byte line)
                pos.assign(n, 0); span_lo.assign(n, 0); span_hi.assign(n, 0);
                if (root) {
                    uint32_t next = 0;
                    stk.clear();
                    stk.push_back(root);
                    span_lo[root] = 0; span_hi[root] = n;
                    while (!stk.empty()) {
                        const uint32_t x = stk.back(); stk.pop_back();
                        pos[x] = next++;
                        const uint32_t l = L[x], r = R[x];
                        if (l) { span_lo[l] = span_lo[x]; span_hi[l] = x; }
                        if (r) { span_lo[r] = x; span_hi[r] = span_hi[x]; }
                        const bool left_heavy = (x - span_lo[x]) >= (span_hi[x] - x);
                        if (left_heavy) { if (r) stk.push_back(r); if (l) stk.push_back(l); }
                        else { if (l) stk.push_back(l); if (r) stk.push_back(r); }
                    }
                }
                auto at = [&](uint32_t i) { return (uint32_t)(base + pos[i]); };
                set_root[g] = root ? at(root) : 0;
                for (uint32_t i = 1; i < n; ++i) {
                    uint32_t* t = &recs[(size_t)at(i) * 4];
                    t[0] = tip[i - 1];              // descending into the LEFT part: new last tip ...
                    t[1] = L[i] ? at(L[i]) : 0;     // ... and its split
                    t[2] = tip[i];                  // descending into the RIGHT part: new first tip ...
                    t[3] = R[i] ? at(R[i]) : 0;     // ... and its split
                }
            }
        });
        parallel_chunks(NK, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
            for (uint64_t j = lo; j < hi; ++j) {
                const uint32_t* w = &E.postings[kmer_off[j]];
                const uint32_t n = w[0] & POST_LEN_MASK;
                const uint32_t* tip = w + POST_HEADER_WORDS;
                uint32_t* h = &recs[(SPLIT_FIRST_REC + SPLIT_HEADER_RECS * j) * 4];
                h[0] = w[0];
                h[1] = set_root[set_of[j]];
                h[2] = n ? tip[0] : 0xFFFFFFFFu;  // no tip below the root: the "inactive" state {MAX, 0}
                h[3] = n ? tip[n - 1] : 0;
                h[4] = w[1];  // n_leaf_ids (statistics)
                h[5] = (uint32_t)d_kmer_hash[j];
                h[6] = (uint32_t)(d_kmer_hash[j] >> 32);
                h[7] = bucket_of[j];
            }
        });
        E.postings.swap(recs);
        for (uint64_t j = 0; j < NK; ++j) kmer_off[j] = SPLIT_FIRST_REC + SPLIT_HEADER_RECS * j;
    }
    // ---- 5. hash table -------------------------------------------------------------
    uint64_t cap = 16;
    while (cap < 2 * NK) cap <<= 1;
    if (cap >= (1ULL << 32)) { err = "k-mer table exceeds 2^32 slots"; return CLS_E_BAD_DB; }
    E.table.assign(cap, Slot{0, SLOT_EMPTY});
    const uint64_t mask = cap - 1;
    for (uint64_t j = 0; j < NK; ++j) {
        uint64_t h = d->kmer_hash[j];
        uint64_t off = kmer_off[j];
        uint64_t i = h & mask;
        while (E.table[i].loc != SLOT_EMPTY) {
            if (E.table[i].hash == h) {
                // HashMap<MinimizerKey, HashMap<u64,..>>: the same k-mer hash under two buckets (or twice
                // in one bucket) cannot come out of `cls build-db` short of a 64-bit murmur collision.
                err = "k-mer hash " + std::to_string(h) + " occurs more than once in the index (unsupported)";
                return CLS_E_BAD_DB;
            }
            i = (i + 1) & mask;
        }
        E.table[i] = Slot{h, (off << LOC_BUCKET_BITS) | bucket_of[j]};
    }
    // ---- 6. direct table for small k -----------------------------------------------------
    if (E.format == FMT_SPLIT && d->k_size <= DIRECT_MAX_K && N < DIRECT_TIP_MASK) {
        const uint32_t K = (uint32_t)d->k_size, M = (uint32_t)std::min<uint64_t>(d->m_size, d->k_size);
        const uint64_t n_codes = 1ULL << (2 * K);
        E.direct.assign(4 * n_codes, 0);  // {record offset, root split, first tip | bit length << 27, last tip | has_root << 31}
        for (uint64_t c = 0; c < n_codes; ++c) E.direct[4 * c + 2] = 0xFFFFFFFFu;
        std::atomic<bool> foreign{false};
        std::atomic<uint64_t> found{0};
        parallel_chunks(n_codes, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
            static const char LETTER[4] = {'A', 'C', 'T', 'G'};  // code = (ascii >> 1) & 3
            char buf[DIRECT_MAX_K + 1];
            uint64_t cnt = 0;
            for (uint64_t code = lo; code < hi; ++code) {
                for (uint32_t i = 0; i < K; ++i) buf[i] = LETTER[(code >> (2 * i)) & 3];
                const uint64_t h = murmur3_h1_bytes(buf, K);
                for (uint64_t i = h & mask; E.table[i].loc != SLOT_EMPTY; i = (i + 1) & mask) {
                    if (E.table[i].hash != h) continue;
                    const uint64_t bkey = d->bucket_key[E.table[i].loc & LOC_BUCKET_MASK];
                    if (bkey != (M ? murmur3_h1_bytes(buf, M) : 0ull)) foreign = true;
                    const uint32_t off = (uint32_t)(E.table[i].loc >> LOC_BUCKET_BITS);
                    const uint32_t n_tips = E.postings[(size_t)off * 4] & POST_LEN_MASK;
                    uint32_t lg = 0;
                    while (lg < 31 && (1u << lg) <= n_tips) ++lg;  // bit length: small = specific k-mer
                    const uint32_t* hd = &E.postings[(size_t)off * 4];  // {n | flags, root split, first tip, last tip}
                    uint32_t* e = &E.direct[4 * code];
                    e[0] = off;
                    e[1] = hd[1];
                    e[2] = n_tips ? ((lg << DIRECT_TIP_BITS) | hd[2]) : 0xFFFFFFFFu;
                    e[3] = (n_tips ? hd[3] : 0u) | ((hd[0] & POST_HAS_ROOT) ? 0x80000000u : 0u);
                    ++cnt;
                    break;
                }
            }
            found += cnt;
        });
        // an entry filed under a bucket that is not its own prefix's, or a hash no enumerated k-mer
        // produces (k-mers with non-ACGT letters cannot be queried anyway): keep the generic probe path
        if (foreign.load() || found.load() != NK) std::vector<uint32_t>().swap(E.direct);
        if (!E.direct.empty()) {
            // does every k-mer share its state (tip set, root flag) with its reverse complement?  True for an index
            // built from both strands; the fast kernel then looks up one k-mer per window instead of two.
            std::atomic<bool> asym{false};
            parallel_chunks(n_codes, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
                for (uint64_t code = lo; code < hi && !asym.load(std::memory_order_relaxed); ++code) {
                    uint64_t rc = 0;
                    for (uint32_t i = 0; i < K; ++i) rc |= (((code >> (2 * i)) & 3) ^ 2) << (2 * (K - 1 - i));  // complement = code ^ 2 (A0 <-> T2, C1 <-> G3)
                    if (rc <= code) continue;
                    const uint32_t* a = &E.direct[4 * code];
                    const uint32_t* b = &E.direct[4 * rc];
                    if ((a[0] != 0) != (b[0] != 0) || a[1] != b[1] || a[2] != b[2] || a[3] != b[3]) asym = true;
                }
            });
            E.canonical = !asym.load();
        }
    }
    // ---- 7. without a direct table: the hash table once more, with the descent state inside the slots -----
    if (E.format == FMT_SPLIT && E.direct.empty() && N < DIRECT_TIP_MASK && E.postings.size() * 4 < (1ULL << 32)) {
        E.ftable.assign(cap, FSlot{0, 0, 0, 0xFFFFFFFFu, 0, 0, 0});
        parallel_chunks(cap, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; ++i) {
                const Slot& sl = E.table[i];
                if (sl.loc == SLOT_EMPTY) continue;
                const uint32_t off = (uint32_t)(sl.loc >> LOC_BUCKET_BITS);
                const uint32_t* hd = &E.postings[(size_t)off * 4];  // {n | flags, root split, first tip, last tip}
                const uint32_t n_tips = hd[0] & POST_LEN_MASK;
                uint32_t lg = 0;
                while (lg < 31 && (1u << lg) <= n_tips) ++lg;
                FSlot& f = E.ftable[i];
                f.hash = sl.hash;
                f.off = off;
                f.x = hd[1];
                f.vlo_lg = n_tips ? ((lg << DIRECT_TIP_BITS) | hd[2]) : 0xFFFFFFFFu;
                f.vhi_root = (n_tips ? hd[3] : 0u) | ((hd[0] & POST_HAS_ROOT) ? 0x80000000u : 0u);
                f.bucket = (uint32_t)(sl.loc & LOC_BUCKET_MASK);
            }
        });
    }
    E.bucket_key.assign(d->bucket_key, d->bucket_key + d->n_buckets);
    if (E.bucket_key.empty()) E.bucket_key.push_back(0);
    E.k = (uint32_t)d->k_size;
    E.m = (uint32_t)std::min<uint64_t>(d->m_size, UINT32_MAX);
    E.m_eff = (uint32_t)std::min<uint64_t>(d->m_size, d->k_size);
    E.n_kmers = NK;
    E.n_closed = n_closed.load();
    E.root_has_children = d->nodes[0].has_children != 0;
    return CLS_OK;
}

}  // namespace cls
