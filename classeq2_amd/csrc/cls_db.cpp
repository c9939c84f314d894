// Host-side encoder: borrowed `cls_db_desc` (the reference's Tree + KmersMap,
// flattened) -> the HBM layout of cls_device.h.  Runs once per database at
// cls_db_create(); nothing here is on the timed path.
#include "cls_db.h"

#include "cls_murmur.h"
#include "cls_tuning.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
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
    if (d->abi_version != 1 && d->abi_version != CLS_ABI_VERSION) { err = "cls_db_desc.abi_version mismatch"; return CLS_E_INVALID_ARG; }
    if (d->abi_version >= 2 && d->node_set_kind > CLS_SETS_LEAVES) { err = "cls_db_desc.node_set_kind out of range"; return CLS_E_INVALID_ARG; }
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
    // phase times on stderr when the `timing` knob is set (cls_set_tuning): index builds at BASELINE config 5's size take minutes
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!tuning().timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "encode_db: %-28s %8.2f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
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
    E.kids.assign(4 * (size_t)N, 0);
    for (uint32_t r = 0; r < N; ++r) {
        const DNode& n = E.nodes[r];
        const uint32_t nc = d->nodes[order[r]].n_children;
        for (uint32_t j = 2; j < 5; ++j) E.kids[4 * (size_t)r + (j - 2)] = j < nc ? E.nodes[n.first_child + j].pre : n.pre + n.size;
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
    lap("tree");
    // ---- 4. per k-mer: node set -> sorted pre list -> tips / explicit list -------
    const uint64_t NK = d->n_kmers;
    const bool leaves_only = d->abi_version >= 2 && d->node_set_kind == CLS_SETS_LEAVES;
    if (leaves_only) {
        // the ids name LEAF-kind clades; the node set is the union of their root->leaf paths (build_database/mod.rs:160-169).
        // A LEAF-kind clade with children could sit ON such a path without being listed: keep the contract simple.
        for (uint32_t r = 0; r < N; ++r)
            if (d->nodes[order[r]].kind == CLS_KIND_LEAF && d->nodes[order[r]].n_children) {
                err = "leaves-only node sets need childless LEAF clades"; return CLS_E_BAD_DB;
            }
    }
    unsigned nt = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    std::vector<uint32_t> bucket_of(NK);
    parallel_chunks(d->n_buckets, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
        for (uint64_t b = lo; b < hi; ++b)
            for (uint64_t j = d->bucket_kmer_off[b]; j < d->bucket_kmer_off[b + 1]; ++j) bucket_of[j] = (uint32_t)b;
    });
    std::vector<std::vector<uint32_t>> chunk_words(nt);
    std::vector<uint64_t> local_off(NK);  // offset of k-mer j inside its chunk
    std::vector<std::pair<uint64_t, uint64_t>> chunk_range(nt, {0, 0});
    std::atomic<uint64_t> n_closed{0};
    std::atomic<bool> bad_leaf{false};
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
                if (leaves_only && (p == UINT32_MAX || !leaf_by_pre[p])) bad_leaf.store(true, std::memory_order_relaxed);
            }
            std::sort(P.begin(), P.end());
            P.erase(std::unique(P.begin(), P.end()), P.end());
            local_off[j] = W.size();
            if (leaves_only) {  // closed by construction: the tips ARE the leaves, the root is on every path
                W.push_back((uint32_t)P.size() | (P.empty() ? 0u : POST_HAS_ROOT) | POST_CLOSED);
                W.push_back((uint32_t)P.size());
                W.insert(W.end(), P.begin(), P.end());
                ++closed_cnt;
                continue;
            }
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
    if (bad_leaf.load()) { err = "leaves-only node sets hold an id that is no LEAF clade of the tree"; return CLS_E_BAD_DB; }
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
    std::vector<uint64_t>().swap(local_off);
    lap("tips per k-mer");
    // ---- 4b. tip sets + split trees when the whole index allows it (see cls_device.h) -------
    E.strictly_binary = true;
    for (uint32_t r = 0; r < N; ++r)
        if (d->nodes[order[r]].n_children != 0 && d->nodes[order[r]].n_children != 2) { E.strictly_binary = false; break; }
    E.format = (n_closed.load() == NK && !tuning().force_list && N < DIRECT_TIP_MASK) ? FMT_SPLIT : FMT_LIST;
    std::vector<uint32_t> set_of;  // FMT_SPLIT: k-mer -> set id (>= 1)
    std::vector<uint32_t> set_mask;  // MASK halves: per set its tips as bits relative to the first one, 0 if they span more than 32 rows
    if (E.format == FMT_SPLIT) {
        if (NK >= (1ULL << 32) - 1) { err = "more than 2^32 - 2 k-mers"; return CLS_E_BAD_DB; }
        auto words_of = [&](uint32_t j) { return &E.postings[kmer_off[j]]; };
        // k-mers with the SAME node set (same tips, same root flag) share one set: group them exactly -- by a
        // signature first, then by the lists themselves inside a run of equal signatures.
        struct SigIdx { uint64_t sig; uint32_t j; };
        std::vector<SigIdx> by_set(NK);
        parallel_chunks(NK, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
            for (uint64_t j = lo; j < hi; ++j) {
                const uint32_t* w = words_of((uint32_t)j);
                const uint32_t n = w[0] & POST_LEN_MASK;
                uint64_t h = 0x9E3779B97F4A7C15ull ^ n ^ ((uint64_t)(w[0] & POST_HAS_ROOT) << 8);
                for (uint32_t i = 0; i < n; ++i) h = fmix64(h ^ w[POST_HEADER_WORDS + i]) + 0x632BE59BD9B4E019ull;
                by_set[j] = {h, (uint32_t)j};
            }
        });
        auto same_set = [&](uint32_t a, uint32_t b) {
            const uint32_t *wa = words_of(a), *wb = words_of(b);
            const uint32_t n = wa[0] & POST_LEN_MASK;
            return wa[0] == wb[0] && (n == 0 || memcmp(wa + POST_HEADER_WORDS, wb + POST_HEADER_WORDS, (size_t)n * 4) == 0);
        };
        auto less_set = [&](uint32_t a, uint32_t b) {  // total order on node sets (then on the k-mer index)
            const uint32_t *wa = words_of(a), *wb = words_of(b);
            if (wa[0] != wb[0]) return wa[0] < wb[0];
            const uint32_t n = wa[0] & POST_LEN_MASK;
            const int c = n ? memcmp(wa + POST_HEADER_WORDS, wb + POST_HEADER_WORDS, (size_t)n * 4) : 0;
            return c != 0 ? c < 0 : a < b;
        };
        lap("set signatures");
        parallel_sort(by_set, nt, [&](const SigIdx& a, const SigIdx& b) { return a.sig != b.sig ? a.sig < b.sig : a.j < b.j; });
        lap("sort by signature");
        // (runs of equal signature that hold DIFFERENT sets -- a 64-bit collision -- are put in set order)
        for (uint64_t i = 0; i < NK;) {
            uint64_t e = i + 1;
            bool mixed = false;
            while (e < NK && by_set[e].sig == by_set[i].sig) { mixed |= !same_set(by_set[i].j, by_set[e].j); ++e; }
            if (mixed) std::sort(by_set.begin() + (ptrdiff_t)i, by_set.begin() + (ptrdiff_t)e, [&](const SigIdx& a, const SigIdx& b) { return less_set(a.j, b.j); });
            i = e;
        }
        // distinct sets, then numbered in ascending (first tip, last tip, size): reads that the locality order puts
        // next to each other look up neighbouring set records and split trees
        struct SetKey { uint32_t first, last, n, rep; };
        std::vector<SetKey> keys;
        std::vector<uint32_t> tmp_set(NK);  // k-mer -> provisional set number
        {
            // position i opens a new set iff its k-mer's set differs from its predecessor's; numbered by a prefix sum
            std::vector<uint8_t> opens(NK);
            parallel_chunks(NK, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
                for (uint64_t i = lo; i < hi; ++i) opens[i] = (i == 0 || !same_set(by_set[i - 1].j, by_set[i].j)) ? 1 : 0;
            });
            std::vector<uint64_t> chunk_first(nt + 1, 0);  // sets opened before each chunk
            parallel_chunks(NK, nt, [&](unsigned t, uint64_t lo, uint64_t hi) {
                uint64_t c = 0;
                for (uint64_t i = lo; i < hi; ++i) c += opens[i];
                chunk_first[t + 1] = c;
            });
            for (unsigned t = 0; t < nt; ++t) chunk_first[t + 1] += chunk_first[t];
            keys.resize(chunk_first[nt]);
            const bool chunked = !(nt <= 1 || NK < 4096);  // (parallel_chunks runs small inputs as ONE chunk numbered 0)
            parallel_chunks(NK, nt, [&](unsigned t, uint64_t lo, uint64_t hi) {
                uint64_t g = chunked ? chunk_first[t] : 0;
                for (uint64_t i = lo; i < hi; ++i) {
                    const uint32_t j = by_set[i].j;
                    if (opens[i]) {
                        const uint32_t* w = words_of(j);
                        const uint32_t n = w[0] & POST_LEN_MASK;
                        keys[g++] = {n ? w[POST_HEADER_WORDS] : 0xFFFFFFFFu, n ? w[POST_HEADER_WORDS + n - 1] : ((w[0] & POST_HAS_ROOT) ? 1u : 0u), n, j};
                    }
                    tmp_set[j] = (uint32_t)(g - 1);
                }
            });
        }
        std::vector<SigIdx>().swap(by_set);
        const uint64_t NS = keys.size();
        lap("distinct sets");
        std::vector<uint32_t> rank(NS);  // provisional number -> position in the final order
        {
            std::vector<uint32_t> ord(NS);
            for (uint64_t g = 0; g < NS; ++g) ord[g] = (uint32_t)g;
            parallel_sort(ord, nt, [&](uint32_t a, uint32_t b) {
                const SetKey &x = keys[a], &y = keys[b];
                if (x.first != y.first) return x.first < y.first;
                if (x.last != y.last) return x.last < y.last;
                if (x.n != y.n) return x.n < y.n;
                return less_set(x.rep, y.rep);
            });
            std::vector<SetKey> sorted(NS);
            for (uint64_t g = 0; g < NS; ++g) { sorted[g] = keys[ord[g]]; rank[ord[g]] = (uint32_t)g; }
            keys.swap(sorted);
        }
        set_of.resize(NK);
        parallel_chunks(NK, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
            for (uint64_t j = lo; j < hi; ++j) set_of[j] = rank[tmp_set[j]] + 1;  // set 0 = "no such k-mer"
        });
        std::vector<uint32_t>().swap(tmp_set);
        std::vector<uint32_t>().swap(rank);
        E.n_sets = NS;
        lap("set order");
        // split records: set g owns (n - 1) of them from set_rec[g]
        std::vector<uint64_t> set_rec(NS + 1);
        uint64_t n_recs = 1;  // record 0: the dummy
        for (uint64_t g = 0; g < NS; ++g) { set_rec[g] = n_recs; n_recs += keys[g].n ? keys[g].n - 1 : 0; }
        set_rec[NS] = n_recs;
        if (n_recs >= (1ULL << 32)) { err = "split trees exceed 2^32 records"; return CLS_E_BAD_DB; }
        HugeVec<uint32_t> recs((n_recs + 1) * 4, 0);
        recs[2] = 0xFFFFFFFFu;  // dummy {0, 0, first tip = MAX, 0}: decodes to "inactive"
        const bool mask_halves = !tuning().no_mask_halves;
        HugeVec<uint32_t> recs2(mask_halves ? (n_recs + 1) * 4 : 0, 0);
        if (mask_halves) { recs2[2] = 0xFFFFFFFFu; set_mask.assign(NS + 1, 0u); }
        E.sets.assign(NS + 1, SetRec{0u, 0xFFFFFFFFu, 0u, 0u});
        parallel_chunks(NS, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
            std::vector<uint32_t> dd, stk, L, R, pos, span_lo, span_hi;
            for (uint64_t g = lo; g < hi; ++g) {
                const uint32_t* w = words_of(keys[g].rep);
                const uint32_t n = w[0] & POST_LEN_MASK;
                const uint32_t* tip = w + POST_HEADER_WORDS;
                const uint64_t base = set_rec[g];
                dd.assign(n, 0); L.assign(n, 0); R.assign(n, 0);
                for (uint32_t i = 1; i < n; ++i) {  // depth of LCA(tip[i-1], tip[i])
                    uint32_t a = tip[i - 1];
                    while (!(tip[i] < a + size_by_pre[a])) a = parent_pre[a];
                    dd[i] = depth_by_pre[a];
                }
                stk.clear();
                for (uint32_t i = 1; i < n; ++i) {  // Cartesian tree, shallowest LCA on top
                    uint32_t last = 0;
                    while (!stk.empty() && dd[stk.back()] > dd[i]) { last = stk.back(); stk.pop_back(); }
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
                for (uint32_t i = 1; i < n; ++i) {
                    uint32_t* t = &recs[(size_t)at(i) * 4];
                    t[0] = tip[i - 1];              // descending into the LEFT part: new last tip ...
                    t[1] = L[i] ? at(L[i]) : 0;     // ... and its split
                    t[2] = tip[i];                  // descending into the RIGHT part: new first tip ...
                    t[3] = R[i] ? at(R[i]) : 0;     // ... and its split
                    if (mask_halves) {  // the same record with narrow parts as bit masks (cls_device.h: MASK halves)
                        uint32_t* u = &recs2[(size_t)at(i) * 4];
                        u[0] = t[0]; u[1] = t[1]; u[2] = t[2]; u[3] = t[3];
                        const uint32_t l0 = span_lo[i], r1 = span_hi[i];  // left part = tip[l0 .. i), right part = tip[i .. r1)
                        if (tip[i - 1] - tip[l0] < MASK_HALF_SPAN) {
                            uint32_t bits = 0;
                            for (uint32_t j = l0; j < i; ++j) bits |= 1u << (tip[j] - tip[l0]);
                            u[0] = tip[i - 1] | MASK_HALF; u[1] = bits;
                        }
                        if (tip[r1 - 1] - tip[i] < MASK_HALF_SPAN) {
                            uint32_t bits = 0;
                            for (uint32_t j = i; j < r1; ++j) bits |= 1u << (tip[j] - tip[i]);
                            u[2] = tip[i] | MASK_HALF; u[3] = bits;
                        }
                    }
                }
                if (mask_halves && n && tip[n - 1] - tip[0] < MASK_HALF_SPAN) {
                    uint32_t bits = 0;
                    for (uint32_t j = 0; j < n; ++j) bits |= 1u << (tip[j] - tip[0]);
                    set_mask[g + 1] = bits;
                }
                uint32_t lg = 0;
                while (lg < 31 && (1u << lg) <= n) ++lg;  // bit length: small = specific to a small clade
                SetRec& sr = E.sets[g + 1];
                sr.x = root ? at(root) : 0;
                sr.vlo_lg = n ? ((lg << DIRECT_TIP_BITS) | tip[0]) : 0xFFFFFFFFu;
                sr.vhi_root = (n ? tip[n - 1] : 0u) | ((w[0] & POST_HAS_ROOT) ? 0x80000000u : 0u);
                sr.n_leaf = w[1];
            }
        });
        E.postings.swap(recs);
        E.postings2.swap(recs2);
        if (mask_halves) {  // the set records again, narrow sets as bits (the fronts without the fat direct table read these)
            E.sets2.assign(E.sets.begin(), E.sets.end());
            parallel_chunks(NS + 1, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
                for (uint64_t g = lo; g < hi; ++g)
                    if (set_mask[g]) { E.sets2[g].x = set_mask[g]; E.sets2[g].vhi_root |= FAT_X_IS_BITS; }
            });
        }
        lap("split trees");
    }
    // ---- 5. hash table (linear probing; parallel claims, then a duplicate check) -------------------------
    uint64_t cap = 16;
    while (cap < 2 * NK) cap <<= 1;
    if (cap >= (1ULL << 32)) { err = "k-mer table exceeds 2^32 slots"; return CLS_E_BAD_DB; }
    const uint64_t mask = cap - 1;
    const bool split = E.format == FMT_SPLIT;
    // FMT_SPLIT slots are TSlot{hash, set, bucket} (set != 0 <=> occupied); FMT_LIST slots Slot{hash, loc}
    E.table.assign(cap, split ? Slot{0, 0} : Slot{0, SLOT_EMPTY});
    {
        // a slot is claimed by a compare-and-swap on its second word, then its hash is written; lookups only
        // happen after every thread has joined
        uint64_t* raw = reinterpret_cast<uint64_t*>(E.table.data());
        const uint64_t empty = split ? 0ull : SLOT_EMPTY;
        // specificity tier of a set (its locality-key class, cls_device.h): bit length of the tip count <= 3 / 6 / 9 / more
        auto tier_of = [&](uint32_t sid) -> uint64_t { const uint32_t lg = E.sets[sid].vlo_lg >> DIRECT_TIP_BITS; return lg <= 3 ? 0 : lg <= 6 ? 1 : lg <= 9 ? 2 : 3; };
        parallel_chunks(NK, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
            for (uint64_t j = lo; j < hi; ++j) {
                const uint64_t h = d->kmer_hash[j];
                const uint64_t val = split ? ((uint64_t)set_of[j] | (((uint64_t)bucket_of[j] | (tier_of(set_of[j]) << TIER_SHIFT)) << 32))
                                           : ((kmer_off[j] << LOC_BUCKET_BITS) | bucket_of[j]);
                for (uint64_t i = h & mask;; i = (i + 1) & mask) {
                    uint64_t expect = empty;
                    if (__atomic_load_n(&raw[2 * i + 1], __ATOMIC_RELAXED) == empty &&
                        __atomic_compare_exchange_n(&raw[2 * i + 1], &expect, val, false, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED)) {
                        raw[2 * i] = h;
                        break;
                    }
                }
            }
        });
        // HashMap<MinimizerKey, HashMap<u64,..>>: the same k-mer hash under two buckets (or twice in one bucket) cannot
        // come out of `cls build-db` short of a 64-bit murmur collision: every occupied slot must be the FIRST slot
        // of its probe sequence that holds its hash
        std::atomic<bool> has_dup{false};
        std::atomic<uint64_t> dup{0};
        parallel_chunks(cap, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
            for (uint64_t s = lo; s < hi; ++s) {
                if (raw[2 * s + 1] == empty) continue;
                const uint64_t h = raw[2 * s];
                for (uint64_t i = h & mask; i != s; i = (i + 1) & mask)  // (no empty slot between a hash's home and its slot)
                    if (raw[2 * i] == h) { dup.store(h, std::memory_order_relaxed); has_dup.store(true, std::memory_order_relaxed); break; }
            }
        });
        if (has_dup.load()) { err = "k-mer hash " + std::to_string(dup.load()) + " occurs more than once in the index (unsupported)"; return CLS_E_BAD_DB; }
    }
    lap("hash table");
    // ---- 6. direct table for small k: 4^k set ids -----------------------------------------------------
    if (split && d->k_size <= DIRECT_MAX_K && E.n_sets < SET_ID_MASK) {
        const uint32_t K = (uint32_t)d->k_size, M = (uint32_t)std::min<uint64_t>(d->m_size, d->k_size);
        const uint64_t n_codes = 1ULL << (2 * K);
        E.direct.assign(n_codes, 0);
        const TSlot* tab = reinterpret_cast<const TSlot*>(E.table.data());
        std::atomic<bool> foreign{false};
        std::atomic<uint64_t> found{0};
        parallel_chunks(n_codes, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
            static const char LETTER[4] = {'A', 'C', 'T', 'G'};  // code = (ascii >> 1) & 3
            // most of the 4^k probes miss every cache of a table of gigabytes: hash a block of codes, prefetch their
            // home slots, then probe
            constexpr int B = 32;
            char buf[DIRECT_MAX_K + 1];
            uint64_t hs[B];
            uint64_t cnt = 0;
            for (uint64_t c0 = lo; c0 < hi; c0 += B) {
                const int nb = (int)std::min<uint64_t>(B, hi - c0);
                for (int q = 0; q < nb; ++q) {
                    const uint64_t code = c0 + q;
                    for (uint32_t i = 0; i < K; ++i) buf[i] = LETTER[(code >> (2 * i)) & 3];
                    hs[q] = murmur3_h1_bytes(buf, K);
                    __builtin_prefetch(&tab[hs[q] & mask]);
                }
                for (int q = 0; q < nb; ++q) {
                    const uint64_t code = c0 + q, h = hs[q];
                    for (uint64_t i = h & mask; tab[i].set != 0; i = (i + 1) & mask) {
                        if (tab[i].hash != h) continue;
                        for (uint32_t t = 0; t < M; ++t) buf[t] = LETTER[(code >> (2 * t)) & 3];
                        if (d->bucket_key[tab[i].bucket & (uint32_t)LOC_BUCKET_MASK] != (M ? murmur3_h1_bytes(buf, M) : 0ull)) foreign = true;
                        E.direct[code] = tab[i].set | (tab[i].bucket & ~SET_ID_MASK);  // set id | tier << 30
                        ++cnt;
                        break;
                    }
                }
            }
            found += cnt;
        });
        // an entry filed under a bucket that is not its own prefix's, or a hash no enumerated k-mer
        // produces (k-mers with non-ACGT letters cannot be queried anyway): keep the generic probe path
        if (foreign.load() || found.load() != NK) HugeVec<uint32_t>().swap(E.direct);
        if (!E.direct.empty()) {
            // does every k-mer share its set with its reverse complement?  True for an index built from both
            // strands; the fast kernels then look up one k-mer per window instead of two.
            std::atomic<bool> asym{false};
            parallel_chunks(n_codes, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
                for (uint64_t code = lo; code < hi && !asym.load(std::memory_order_relaxed); ++code) {
                    uint64_t rc = 0;
                    for (uint32_t i = 0; i < K; ++i) rc |= (((code >> (2 * i)) & 3) ^ 2) << (2 * (K - 1 - i));  // complement = code ^ 2 (A0 <-> T2, C1 <-> G3)
                    if (rc > code && E.direct[code] != E.direct[rc]) asym = true;
                }
            });
            E.canonical = !asym.load();
            if (K <= FAT_DIRECT_MAX_K && !tuning().no_fat_direct) {  // the denormalised copy for the wave-per-read kernels
                E.direct16.resize(4 * n_codes);
                parallel_chunks(n_codes, nt, [&](unsigned, uint64_t lo, uint64_t hi) {
                    for (uint64_t code = lo; code < hi; ++code) {
                        const uint32_t e = E.direct[code];
                        const SetRec& sr = E.sets[e & SET_ID_MASK];  // (set 0: {0, MAX, 0, 0})
                        uint32_t* o = &E.direct16[4 * code];
                        o[0] = sr.x; o[1] = sr.vlo_lg; o[2] = sr.vhi_root; o[3] = e;
                        // a set that spans at most 32 rows starts its descent as bits (only the wave-per-read kernels read
                        // this table's x: cls_device.h, MASK halves)
                        const uint32_t bits = set_mask.empty() ? 0u : set_mask[e & SET_ID_MASK];
                        if (bits) { o[0] = bits; o[2] |= FAT_X_IS_BITS; }
                    }
                });
            }
        }
    }
    lap("direct table");
    E.bucket_key.assign(d->bucket_key, d->bucket_key + d->n_buckets);
    if (E.bucket_key.empty()) E.bucket_key.push_back(0);
    {   // bucket index by the 2-bit code of a k-mer's first m characters (cls_device.h, MZ_TABLE_MAX_M)
        const uint32_t M = (uint32_t)std::min<uint64_t>(d->m_size, d->k_size);
        std::unordered_map<uint64_t, uint32_t> by_key;
        bool distinct = true;
        for (uint64_t b = 0; b < d->n_buckets; ++b) distinct &= by_key.emplace(d->bucket_key[b], (uint32_t)b).second;
        if (M <= MZ_TABLE_MAX_M && distinct) {
            static const char LETTER[4] = {'A', 'C', 'T', 'G'};  // code = (ascii >> 1) & 3
            E.mz_bucket.assign((size_t)1 << (2 * M), MZ_NO_BUCKET);
            char buf[MZ_TABLE_MAX_M + 1];
            for (uint32_t code = 0; code < E.mz_bucket.size(); ++code) {
                for (uint32_t t = 0; t < M; ++t) buf[t] = LETTER[(code >> (2 * t)) & 3];
                const auto it = by_key.find(M ? murmur3_h1_bytes(buf, M) : 0ull);
                if (it != by_key.end()) E.mz_bucket[code] = it->second;
            }
        }
    }
    E.k = (uint32_t)d->k_size;
    E.m = (uint32_t)std::min<uint64_t>(d->m_size, UINT32_MAX);
    E.m_eff = (uint32_t)std::min<uint64_t>(d->m_size, d->k_size);
    E.n_kmers = NK;
    E.n_closed = n_closed.load();
    E.root_has_children = d->nodes[0].has_children != 0;
    return CLS_OK;
}

}  // namespace cls

// ---- host-side self-check of the MASK halves (tests/test_host_cpu.py; no device involved) ---------------------------
// Every half of the second copy of the split records against the tips its part has ACCORDING TO THE FIRST COPY (walked
// record by record): a MASK half's bits are exactly those tips relative to the part's first one, a plain half is
// unchanged and leads to a part that spans more than 32 rows.  counts[0..2] = records, MASK halves, plain halves.
// CLS_E_INTERNAL on the first mismatch.
extern "C" int cls_db_debug_mask_halves(const cls_db_desc* d, uint64_t* counts) {
    try {
        cls::EncodedDb E;
        std::string err;
        const int rc = cls::encode_db(d, E, err);
        if (rc != CLS_OK) return rc;
        using namespace cls;
        counts[0] = counts[1] = counts[2] = 0;
        if (E.format != FMT_SPLIT || E.postings2.empty()) return CLS_OK;
        if (E.postings2.size() != E.postings.size()) return CLS_E_INTERNAL;
        const TipRec* recs = reinterpret_cast<const TipRec*>(E.postings.data());
        const TipRec* rec2 = reinterpret_cast<const TipRec*>(E.postings2.data());
        const uint64_t n_recs = E.postings.size() / 4;
        // tips of the part a half {end tip, record} leads to, in order (side 0: the part ENDS at the tip, 1: it STARTS there)
        std::function<void(uint32_t, uint32_t, int, std::vector<uint32_t>&)> part = [&](uint32_t end_tip, uint32_t x, int side, std::vector<uint32_t>& out) {
            if (!x) { out.push_back(end_tip); return; }
            const TipRec& t = recs[x];
            part(t.tip_prev, t.l, 0, out);
            part(t.tip, t.r, 1, out);
            (void)side;
        };
        std::vector<uint32_t> tips;
        for (uint64_t x = 1; x + 1 < n_recs; ++x) {  // (record 0: the dummy; the last one: the tail pad)
            const TipRec& t = recs[x];
            const TipRec& u = rec2[x];
            if (t.tip_prev == 0 && t.tip == 0 && t.l == 0 && t.r == 0) continue;  // (unused tail)
            ++counts[0];
            for (int side = 0; side < 2; ++side) {
                const uint32_t end_tip = side ? t.tip : t.tip_prev, sub = side ? t.r : t.l;
                const uint32_t uw = side ? u.tip : u.tip_prev, ux = side ? u.r : u.l;
                tips.clear();
                part(end_tip, sub, side, tips);
                if (!std::is_sorted(tips.begin(), tips.end()) || (side ? tips.front() : tips.back()) != end_tip) return CLS_E_INTERNAL;
                const bool narrow = tips.back() - tips.front() < MASK_HALF_SPAN;
                if (uw & MASK_HALF) {
                    uint32_t bits = 0;
                    for (uint32_t v : tips) bits |= 1u << (v - tips.front());
                    if (!narrow || (uw & ~MASK_HALF) != end_tip || ux != bits) return CLS_E_INTERNAL;
                    ++counts[1];
                } else {
                    if (narrow || uw != end_tip || ux != sub) return CLS_E_INTERNAL;
                    ++counts[2];
                }
            }
        }
        return CLS_OK;
    } catch (const std::bad_alloc&) {
        return CLS_E_NOMEM;
    } catch (...) {
        return CLS_E_INTERNAL;
    }
}

// ---- host-side self-check of the encoder (tests/test_host_cpu.py; no device involved) ----------------------------
// For each probe (k-mer hash, clade id): is the clade in the k-mer's node set ACCORDING TO THE ENCODED INDEX?  Answered
// the way the kernels do: hash table -> set / postings -> tips (every tip of a split tree is collected by walking
// its records) -> pre-order interval test; the root through its flag.  out[i]: 0 = no, 1 = yes, 2 = k-mer not in the
// index, 3 = no such clade.
extern "C" int cls_db_debug_members(const cls_db_desc* d, const uint64_t* hashes, const uint64_t* clade_ids, uint64_t n, uint8_t* out) {
    try {
        cls::EncodedDb E;
        std::string err;
        const int rc = cls::encode_db(d, E, err);
        if (rc != CLS_OK) return rc;
        using namespace cls;
        std::vector<std::pair<uint64_t, uint32_t>> id2row(E.nodes.size());
        for (uint32_t r = 0; r < E.nodes.size(); ++r) id2row[r] = {E.nodes[r].id, r};
        std::sort(id2row.begin(), id2row.end());
        const uint64_t mask = E.table.size() - 1;
        std::vector<uint32_t> tips, stk;
        for (uint64_t i = 0; i < n; ++i) {
            auto it = std::lower_bound(id2row.begin(), id2row.end(), std::make_pair(clade_ids[i], (uint32_t)0));
            if (it == id2row.end() || it->first != clade_ids[i]) { out[i] = 3; continue; }
            const DNode& c = E.nodes[it->second];
            const uint64_t h = hashes[i];
            bool found = false, has_root = false, closed = true;
            tips.clear();
            if (E.format == FMT_SPLIT) {
                const TSlot* tab = reinterpret_cast<const TSlot*>(E.table.data());
                for (uint64_t s = h & mask; tab[s].set != 0; s = (s + 1) & mask) {
                    if (tab[s].hash != h) continue;
                    found = true;
                    const SetRec& sr = E.sets[tab[s].set];
                    has_root = (sr.vhi_root >> 31) != 0;
                    if (sr.vlo_lg != 0xFFFFFFFFu) {
                        const uint32_t lo = sr.vlo_lg & DIRECT_TIP_MASK, hi = sr.vhi_root & 0x7FFFFFFFu;
                        if (!sr.x) { tips.push_back(lo); if (hi != lo) return CLS_E_INTERNAL; }
                        else {  // in-order walk of the split tree: {tip_prev, L, tip, R}
                            const TipRec* recs = reinterpret_cast<const TipRec*>(E.postings.data());
                            std::vector<std::pair<uint32_t, int>> st{{sr.x, 0}};
                            tips.push_back(lo);
                            while (!st.empty()) {
                                auto [x, phase] = st.back();
                                st.pop_back();
                                const TipRec& t = recs[x];
                                if (phase == 0) { st.push_back({x, 1}); if (t.l) st.push_back({t.l, 0}); }
                                else { if (tips.back() != t.tip_prev) return CLS_E_INTERNAL; tips.push_back(t.tip); if (t.r) st.push_back({t.r, 0}); }
                            }
                            if (tips.back() != hi) return CLS_E_INTERNAL;
                        }
                    }
                    break;
                }
            } else {
                for (uint64_t s = h & mask; E.table[s].loc != SLOT_EMPTY; s = (s + 1) & mask) {
                    if (E.table[s].hash != h) continue;
                    found = true;
                    const uint32_t* w = &E.postings[E.table[s].loc >> LOC_BUCKET_BITS];
                    has_root = (w[0] & POST_HAS_ROOT) != 0;
                    closed = (w[0] & POST_CLOSED) != 0;
                    tips.assign(w + POST_HEADER_WORDS, w + POST_HEADER_WORDS + (w[0] & POST_LEN_MASK));
                    break;
                }
            }
            if (!found) { out[i] = 2; continue; }
            bool in = false;
            if (c.pre == 0) in = has_root;
            else if (closed) { auto t = std::lower_bound(tips.begin(), tips.end(), c.pre); in = t != tips.end() && *t < c.pre + c.size; }
            else in = std::binary_search(tips.begin(), tips.end(), c.pre);
            out[i] = in ? 1 : 0;
        }
        return CLS_OK;
    } catch (...) {
        return CLS_E_INTERNAL;
    }
}
