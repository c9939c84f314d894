// HIP kernels of the placement path (gfx950 / CDNA4, wave64).
//
// One WAVEFRONT places one read (reads of up to 64*SLOTS k-mers):
//   A. k-mer extraction (forward + reverse complement, kmers_map.rs:375-398),
//      MurmurHash3 keying (kmers_map.rs:157-159), probe of the HBM-resident
//      k-mer table, minimizer-bucket filter and distinct-hash de-duplication
//      (kmers_map.rs:273-311) -- one k-mer per (lane, slot);
//   B. thresholds (place_sequence.rs:98-139, :156-166, :231-254);
//   C. top-down clade descent (place_sequence.rs:279-601): per level every
//      lane classifies its k-mers against the children's pre-order intervals,
//      __ballot/__popcll give |K_c|, |only_c|, |U| -> the one-vs-rest test.
// Integer set membership only: no MFMA.  See DESIGN.md for the data layout and
// the roofline accounting.
#include <hip/hip_runtime.h>

#include "cls_device.h"
#include "cls_kernels.h"
#include "cls_murmur.h"

namespace cls {

namespace {

constexpr int WAVES_PER_BLOCK = 4;
constexpr uint32_t SET_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ uint32_t popc64(uint64_t m) { return (uint32_t)__popcll(m); }

// first index in [lo, hi) whose value is >= key (hi if none)
__device__ __forceinline__ uint32_t lower_bound_g(const uint32_t* __restrict__ post, uint32_t lo, uint32_t hi, uint32_t key) {
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (post[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Is clade c = pre interval [c0, c1) a member of the k-mer's node set?
// The k-mer's stored elements inside the current parent's interval are
// post[lo, hi), with vlo = post[lo], vhi = post[hi-1] cached in registers.
__device__ __forceinline__ bool member_of(const uint32_t* __restrict__ post, uint32_t lo, uint32_t hi, uint32_t vlo,
                                          uint32_t vhi, bool closed, uint32_t c0, uint32_t c1) {
    if (vhi < c0 || vlo >= c1) return false;
    if (closed) {  // tips: member <=> a tip inside [c0, c1)
        if (vlo >= c0 || vhi < c1) return true;
        uint32_t i = lower_bound_g(post, lo, hi, c0);  // vlo < c0 and vhi >= c1: look inside
        return post[i] < c1;
    }
    // explicit list: member <=> c0 itself is stored
    if (vlo == c0 || vhi == c0) return true;
    if (vlo > c0) return false;
    uint32_t i = lower_bound_g(post, lo, hi, c0);
    return post[i] == c0;
}

struct WaveCtx {
    uint8_t* seq;    // LDS: upper-cased read
    uint32_t* set;   // LDS: distinct-hit set (keys = table slot indices)
    uint32_t* ent;   // LDS: per k-mer, postings offset of its (first-seen) hit or SET_EMPTY
    uint32_t* cnt;   // general path: |K_c| per non-LEAF child
    uint32_t* only;  // general path: |K_c \ R_c| per non-LEAF child
};

__device__ __forceinline__ void write_record(cls_placement* out, uint32_t r, uint32_t status, int32_t one, int32_t rest,
                                             uint32_t levels, uint64_t clade) {
    if ((threadIdx.x & 63) == 0) {
        uint64_t* o = reinterpret_cast<uint64_t*>(out + r);
        o[0] = (uint64_t)(status & 0xFF) | ((uint64_t)(uint32_t)one << 32);
        o[1] = (uint64_t)(uint32_t)rest | ((uint64_t)levels << 32);
        o[2] = clade;
    }
}

template <int SLOTS, int SET_BITS, bool STATS>
__device__ __forceinline__ void place_read(const DbDev db, const PlaceParams prm, const WaveCtx cx, const uint8_t* __restrict__ bases,
                           uint64_t b0, uint64_t b1, uint32_t r, cls_placement* __restrict__ out,
                           cls_query_stats* __restrict__ stats) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t k = db.k;
    const uint64_t L64 = b1 - b0;
    auto put_stats = [&](uint32_t nk, uint32_t nm, uint32_t nr, uint64_t lp) {
        if (STATS && stats && lane == 0) {
            uint64_t* s = reinterpret_cast<uint64_t*>(stats + r);
            s[0] = (uint64_t)nk | ((uint64_t)nm << 32);
            s[1] = (uint64_t)nr;
            s[2] = lp;
        }
    };
    // ---- A0. build_kmer_from_string guards (kmers_map.rs:383-385, place_sequence.rs:98-102)
    if (L64 < k) {
        put_stats(0, 0, 0, 0);
        write_record(out, r, CLS_ERR_TOO_FEW_KMERS, 0, 0, 0, 0);
        return;
    }
    const uint64_t nk64 = 2 * (L64 - k + 1);
    if (nk64 > (uint64_t)(64 * SLOTS)) {
        put_stats(nk64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nk64, 0, 0, 0);
        write_record(out, r, CLS_ERR_READ_TOO_LONG, 0, 0, 0, 0);
        return;
    }
    const uint32_t L = (uint32_t)L64, nf = L - k + 1, nk = 2 * nf;
    // ---- A1. load, upper-case, validate (reverse_complement panics on non-ACGT, kmers_map.rs:440)
    bool bad = false;
    for (uint32_t i = lane; i < L; i += 64) {
        uint8_t c = bases[b0 + i];
        if (c >= 'a' && c <= 'z') c -= 32;
        bad |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
        cx.seq[i] = c;
    }
    for (uint32_t i = lane; i < (1u << SET_BITS); i += 64) cx.set[i] = SET_EMPTY;
    if (__ballot(bad)) {
        put_stats(0, 0, 0, 0);  // the reference dies inside build_kmer_from_string, before any count exists
        write_record(out, r, CLS_ERR_INVALID_BASE, 0, 0, 0, 0);
        return;
    }
    wave_sync();
    // ---- A2. hash every k-mer + its minimizer prefix, probe the table, apply the
    // minimizer-bucket filter and the distinct-hash de-duplication; one k-mer per
    // (slot, lane); the surviving postings offsets are staged in LDS (cx.ent).
    const uint8_t* seq = cx.seq;
    const uint32_t m_eff = db.m_eff;
    const uint32_t* __restrict__ post = db.postings;
    auto hash_at = [&](uint32_t j, uint32_t len) -> uint64_t {
        if (j < nf) {  // forward k-mer at j
            const uint8_t* p = seq + j;
            return murmur3_h1([p](uint32_t i) { return p[i]; }, len);
        }
        const uint8_t* p = seq + (L - 1 - (j - nf));  // k-mer at (j - nf) of the reverse complement
        return murmur3_h1([p](uint32_t i) {
            const uint8_t c = *(p - i);
            return (uint8_t)(c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A');
        }, len);
    };
    uint32_t n_m = 0, n_root = 0;
    uint64_t leafp = 0;
    for (uint32_t base = 0; base < nk; base += 64) {
        const uint32_t j = base + lane;
        bool hit = false;
        uint32_t tidx = 0;
        uint64_t loc = 0, mz = 0;
        if (j < nk) {
            const uint64_t h = hash_at(j, k);
            mz = hash_at(j, m_eff);  // hash("") == 0 when m == 0 (kmers_map.rs:131-134)
            uint64_t idx = h & db.table_mask;
            for (;;) {
                const Slot sl = db.table[idx];
                if (sl.loc == SLOT_EMPTY) break;
                if (sl.hash == h) { hit = true; loc = sl.loc; tidx = (uint32_t)idx; break; }
                idx = (idx + 1) & db.table_mask;
            }
        }
        // the bucket's key must be one of the query's minimizers (kmers_map.rs:295-297)
        bool ok = false;
        uint64_t bk = 0;
        if (hit) { bk = db.bucket_key[loc & LOC_BUCKET_MASK]; ok = (bk == mz); }
        uint64_t pend = __ballot(hit && !ok);
        while (pend) {  // never taken for an index built by `cls build-db`
            const int src = __ffsll((unsigned long long)pend) - 1;
            const uint64_t B = ((uint64_t)__shfl((uint32_t)(bk >> 32), src) << 32) | __shfl((uint32_t)bk, src);
            bool f = false;
            for (uint32_t jj = lane; jj < nk; jj += 64) f |= hash_at(jj, m_eff) == B;
            const bool any = __ballot(f) != 0;
            if ((int)lane == src) ok = any;
            pend &= pend - 1;
        }
        uint32_t ent = SET_EMPTY;
        if (hit && ok) {  // HashSet<u64> of hashes: count each distinct hash once
            uint32_t pos = (tidx * 2654435761u) >> (32 - SET_BITS);
            for (;;) {
                const uint32_t old = atomicCAS(&cx.set[pos], SET_EMPTY, tidx);
                if (old == SET_EMPTY) { ent = (uint32_t)(loc >> LOC_BUCKET_BITS); break; }
                if (old == tidx) break;
                pos = (pos + 1) & ((1u << SET_BITS) - 1);
            }
        }
        cx.ent[j] = ent;  // j < 64*SLOTS always
    }
    wave_sync();
    // ---- A3. per-k-mer state: the stored elements below the root -------------------------
    uint32_t lo[SLOTS], hi[SLOTS], vlo[SLOTS], vhi[SLOTS];
    uint32_t act = 0, closedm = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const uint32_t j = s * 64 + lane;
        const uint32_t off = (j < nk) ? cx.ent[j] : SET_EMPTY;
        const bool is_new = off != SET_EMPTY;
        bool has_root = false;
        lo[s] = hi[s] = vlo[s] = vhi[s] = 0;
        if (is_new) {
            const uint32_t w0 = post[off];
            if (STATS) leafp += post[off + 1];
            has_root = (w0 & POST_HAS_ROOT) != 0;
            const uint32_t len = w0 & POST_LEN_MASK;
            if (has_root && len) {  // only M_root members with something below the root can vote
                lo[s] = off + POST_HEADER_WORDS;
                hi[s] = lo[s] + len;
                vlo[s] = post[lo[s]];
                vhi[s] = post[hi[s] - 1];
                act |= 1u << s;
                if (w0 & POST_CLOSED) closedm |= 1u << s;
            }
        }
        n_m += popc64(__ballot(is_new));
        n_root += popc64(__ballot(is_new && has_root));
    }
    if (STATS) {
        for (int o = 32; o > 0; o >>= 1) leafp += ((uint64_t)__shfl_xor((uint32_t)(leafp >> 32), o) << 32) | __shfl_xor((uint32_t)leafp, o);
        put_stats(nk, n_m, n_root, leafp);
    }
    // ---- B. thresholds -----------------------------------------------------------------
    if (n_m == 0) { write_record(out, r, CLS_UNCLASSIFIABLE_NO_MATCH, 0, 0, 0, 0); return; }      // :130-139
    if (n_root == 0) { write_record(out, r, CLS_UNCLASSIFIABLE_NO_ROOT, 0, 0, 0, 0); return; }     // :156-164
    const DNode* __restrict__ nodes = db.nodes;
    if (!(nodes[0].flags & 1u)) { write_record(out, r, CLS_ERR_ROOT_NO_CHILDREN, 0, 0, 0, 0); return; }  // :199-206
    {
        const double expected = round((double)n_m * prm.min_match_coverage);                      // :231-232
        const uint64_t exp_usize = (expected != expected) ? 0ull : (uint64_t)expected;              // `as usize`
        if ((uint64_t)n_root < exp_usize) { write_record(out, r, CLS_UNCLASSIFIABLE_COVERAGE, (int32_t)n_root, 0, 0, 0); return; }  // :247-254
    }
    // ---- C. descent ----------------------------------------------------------------------
    const bool rm = prm.remove_intersection != 0;
    uint32_t prow = 0;
    int32_t iteration = 0;
    for (;;) {
        ++iteration;
        if (iteration > prm.max_iterations) { write_record(out, r, CLS_ERR_MAX_ITER, 0, 0, (uint32_t)iteration, 0); return; }  // :295-301
        const uint32_t fc = uniform(nodes[prow].first_child);
        const uint32_t m = uniform(nodes[prow].n_nonleaf);
        uint32_t n_pass = 0, n_best = 0, best_row = 0;
        int32_t best_one = 0, best_rest = 0, best_diff = 0;
        if (m <= 2) {
            // ---- binary fast path: everything in registers -----------------------------
            uint32_t a0 = 0, a1 = 0, b0c = 0, b1c = 0;
            if (m >= 1) { a0 = uniform(nodes[fc].pre); a1 = a0 + uniform(nodes[fc].size); }
            if (m == 2) { b0c = uniform(nodes[fc + 1].pre); b1c = b0c + uniform(nodes[fc + 1].size); }
            uint32_t cnt_a = 0, cnt_b = 0, both = 0;
            if (m >= 1) {
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    bool ina = false, inb = false;
                    if (act & (1u << s)) {
                        const bool cl = (closedm >> s) & 1u;
                        ina = member_of(post, lo[s], hi[s], vlo[s], vhi[s], cl, a0, a1);
                        if (m == 2) inb = member_of(post, lo[s], hi[s], vlo[s], vhi[s], cl, b0c, b1c);
                    }
                    cnt_a += popc64(__ballot(ina));
                    cnt_b += popc64(__ballot(inb));
                    both += popc64(__ballot(ina && inb));
                }
            }
            const uint32_t U = cnt_a + cnt_b - both;
            // (one, rest), place_sequence.rs:369-395, with |R_c| = |U| - |only_c| and |R_c \ K_c| = |U| - |K_c|
            for (int c = 0; c < 2; ++c) {
                const uint32_t cn = c ? cnt_b : cnt_a;
                if (cn == 0) continue;  // K_c empty: not a candidate (:329)
                const uint32_t on = cn - both;
                const int32_t one = (int32_t)(rm ? on : cn);
                const int32_t rest = (int32_t)(rm ? U - cn : U - on);
                if (one > rest) {  // :411-417
                    const int32_t diff = one - rest;
                    if (n_pass == 0 || diff > best_diff) { best_diff = diff; n_best = 1; best_row = fc + c; best_one = one; best_rest = rest; }
                    else if (diff == best_diff) ++n_best;
                    ++n_pass;
                }
            }
        } else {
            // ---- general path (polytomies): per-child counters in memory --------------------
            for (uint32_t i = lane; i < m; i += 64) __hip_atomic_store(&cx.only[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t nin[SLOTS], which[SLOTS];
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) { nin[s] = 0; which[s] = 0; }
            for (uint32_t ci = 0; ci < m; ++ci) {
                const uint32_t c0 = uniform(nodes[fc + ci].pre);
                const uint32_t c1 = c0 + uniform(nodes[fc + ci].size);
                uint32_t cn = 0;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    bool in = false;
                    if (act & (1u << s)) in = member_of(post, lo[s], hi[s], vlo[s], vhi[s], (closedm >> s) & 1u, c0, c1);
                    if (in) { if (nin[s] == 0) which[s] = ci; if (nin[s] < 2) ++nin[s]; }
                    cn += popc64(__ballot(in));
                }
                if (lane == 0) __hip_atomic_store(&cx.cnt[ci], cn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            uint32_t U = 0;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                U += popc64(__ballot(nin[s] >= 1));
                if (nin[s] == 1) __hip_atomic_fetch_add(&cx.only[which[s]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            wave_sync();
            for (uint32_t base = 0; base < m; base += 64) {
                const uint32_t ci = base + lane;
                bool pass = false;
                int32_t one = 0, rest = 0;
                if (ci < m) {
                    const uint32_t cn = __hip_atomic_load(&cx.cnt[ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t on = __hip_atomic_load(&cx.only[ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cn) {
                        one = (int32_t)(rm ? on : cn);
                        rest = (int32_t)(rm ? U - cn : U - on);
                        pass = one > rest;
                    }
                }
                uint64_t pm = __ballot(pass);
                while (pm) {  // at most one child can pass; kept general for fidelity with :519-599
                    const int src = __ffsll((unsigned long long)pm) - 1;
                    const int32_t o1 = __shfl(one, src), r1 = __shfl(rest, src);
                    const int32_t diff = o1 - r1;
                    if (n_pass == 0 || diff > best_diff) { best_diff = diff; n_best = 1; best_row = fc + base + src; best_one = o1; best_rest = r1; }
                    else if (diff == best_diff) ++n_best;
                    ++n_pass;
                    pm &= pm - 1;
                }
            }
            wave_sync();
        }
        // ---- PHASE 2 (place_sequence.rs:436-600) -------------------------------------------
        if (n_pass == 0) {
            if (iteration == 1) write_record(out, r, CLS_UNCLASSIFIABLE_LEVEL1, 0, 0, 1, 0);                         // :446-453
            else write_record(out, r, CLS_MAX_RESOLUTION, 0, 0, (uint32_t)iteration, nodes[prow].id);              // :461-464
            return;
        }
        if (n_pass > 1 && n_best != 1) {                                                                             // :575-598
            write_record(out, r, CLS_INCONCLUSIVE, (int32_t)n_pass, 0, (uint32_t)iteration, nodes[prow].id);
            return;
        }
        // update_introspection_node.rs:13-91
        if (uniform(nodes[best_row].n_nonleaf) == 0) {
            write_record(out, r, CLS_IDENTITY_FOUND, best_one, best_rest, (uint32_t)iteration, nodes[best_row].id);
            return;
        }
        prow = best_row;
        const uint32_t n0 = uniform(nodes[prow].pre) + 1;
        const uint32_t n1 = n0 - 1 + uniform(nodes[prow].size);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            if (!(act & (1u << s))) continue;
            bool keep = !(vhi[s] < n0 || vlo[s] >= n1);
            if (keep && vlo[s] < n0) {
                lo[s] = lower_bound_g(post, lo[s], hi[s], n0);  // < hi because vhi >= n0
                vlo[s] = post[lo[s]];
                keep = vlo[s] < n1;
            }
            if (keep && vhi[s] >= n1) {
                hi[s] = lower_bound_g(post, lo[s], hi[s], n1);  // > lo because vlo < n1
                vhi[s] = post[hi[s] - 1];
            }
            if (!keep) act &= ~(1u << s);
        }
    }
}

template <int SLOTS, int SET_BITS, bool STATS>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void place_wave_kernel(DbDev db, PlaceParams prm,
                                                                          const uint8_t* __restrict__ bases,
                                                                          const uint64_t* __restrict__ offsets,
                                                                          uint32_t n_reads, cls_placement* __restrict__ out,
                                                                          cls_query_stats* __restrict__ stats,
                                                                          uint32_t seq_cap, uint32_t* __restrict__ child_ws,
                                                                          uint32_t ws_stride) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t per_wave = seq_cap + (4u << SET_BITS) + 4u * 64 * SLOTS;
    WaveCtx cx;
    cx.seq = smem + wave * per_wave;
    cx.set = reinterpret_cast<uint32_t*>(cx.seq + seq_cap);
    cx.ent = cx.set + (1u << SET_BITS);
    const uint32_t gw = blockIdx.x * WAVES_PER_BLOCK + wave;
    cx.cnt = child_ws ? child_ws + (size_t)gw * 2 * ws_stride : nullptr;
    cx.only = child_ws ? cx.cnt + ws_stride : nullptr;
    const uint32_t n_waves = gridDim.x * WAVES_PER_BLOCK;
    for (uint32_t r = gw; r < n_reads; r += n_waves) {
        const uint64_t b0 = offsets[r], b1 = offsets[r + 1];
        place_read<SLOTS, SET_BITS, STATS>(db, prm, cx, bases, b0, b1, r, out, stats);
        wave_sync();
    }
}

}  // namespace

uint32_t place_ws_words(const DbDev& db, uint32_t grid_blocks) {
    if (db.max_nonleaf_arity <= 2) return 0;
    return grid_blocks * WAVES_PER_BLOCK * 2 * ((db.max_nonleaf_arity + 63) & ~63u);
}

uint32_t place_grid_blocks(uint32_t n_reads, uint32_t n_cu, const DbDev& db) {
    uint32_t want = (n_reads + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    uint32_t cap = n_cu * 8;  // 8 blocks x 4 waves = 32 waves per CU
    if (db.max_nonleaf_arity > 2) {
        // bound the per-wave child-counter workspace to 256 MiB
        uint64_t per_block = (uint64_t)WAVES_PER_BLOCK * 2 * ((db.max_nonleaf_arity + 63) & ~63u) * 4;
        uint64_t fit = (256ull << 20) / per_block;
        if (fit < 1) fit = 1;
        if (cap > fit) cap = (uint32_t)fit;
    }
    return want < cap ? (want ? want : 1) : cap;
}

hipError_t launch_place(const DbDev& db, const PlaceParams& prm, const uint8_t* d_bases, const uint64_t* d_offsets,
                        uint32_t n_reads, cls_placement* d_out, cls_query_stats* d_stats, uint32_t* d_ws,
                        uint32_t grid_blocks, hipStream_t stream) {
    if (n_reads == 0) return hipSuccess;
    constexpr int SLOTS = 5, SET_BITS = 9;  // 320 k-mers per read; 512-entry LDS set
    const uint32_t seq_cap = (64 * SLOTS / 2 + db.k + 15) & ~15u;
    const size_t smem = (size_t)WAVES_PER_BLOCK * (seq_cap + (4u << SET_BITS) + 4u * 64 * SLOTS);
    const uint32_t ws_stride = (db.max_nonleaf_arity + 63) & ~63u;
    if (d_stats)
        hipLaunchKernelGGL((place_wave_kernel<SLOTS, SET_BITS, true>), dim3(grid_blocks), dim3(64 * WAVES_PER_BLOCK), smem, stream,
                           db, prm, d_bases, d_offsets, n_reads, d_out, d_stats, seq_cap, d_ws, ws_stride);
    else
        hipLaunchKernelGGL((place_wave_kernel<SLOTS, SET_BITS, false>), dim3(grid_blocks), dim3(64 * WAVES_PER_BLOCK), smem, stream,
                           db, prm, d_bases, d_offsets, n_reads, d_out, d_stats, seq_cap, d_ws, ws_stride);
    return hipGetLastError();
}

}  // namespace cls
