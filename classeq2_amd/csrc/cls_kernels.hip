// HIP kernels of the placement path (gfx950 / CDNA4, wave64).
//
// The algorithm per read (place_sequence.rs:42-601):
//   A. k-mer extraction (forward + reverse complement, kmers_map.rs:375-398), keying (MurmurHash3,
//      kmers_map.rs:157-159, or the 2-bit code through the direct table), lookup in the HBM-resident index,
//      minimizer-bucket filter and distinct-hash de-duplication (kmers_map.rs:273-311);
//   B. thresholds (place_sequence.rs:98-139, :156-166, :231-254);
//   C. top-down clade descent (place_sequence.rs:279-601): per level |K_c|, |only_c|, |U| over the children's
//      pre-order intervals -> the one-vs-rest test -> narrow to the chosen child.
// Map of this file:
//   match_phase / place_read / place_read_split   generic wave- or workgroup-per-read path, one k-mer per (lane, slot),
//                                                  any k, any index (FMT_LIST and FMT_SPLIT), polytomies by a wave-wide walk
//   fast_front / hash_front / place_read_fast      FMT_SPLIT fast path: direct-table or MurmurHash3 front, k-mers grouped
//                                                  by tip set, descent on {set, weight} groups, per-group polytomy walk
//   order_key_kernel / order_key_half_kernel        locality keys of the reads (sorted by cls_sort.hip)
//   place_long_kernel                              reads beyond 8192 k-mers: state in the workspace
//   classify_kernel, plan_place, launch_place      read-length classes, grids / scratch layout, launches
// Integer set membership only: no MFMA.  See DESIGN.md for the data layout and the roofline accounting.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "cls_device.h"
#include "cls_devutil.h"
#include "cls_kernels.h"
#include "cls_murmur.h"
#include "cls_sort.h"
#include "cls_tuning.h"

namespace cls {

namespace {

constexpr int WAVES_PER_BLOCK = 4;
#ifndef MIN_WAVES_PER_EU
#define MIN_WAVES_PER_EU 4
#endif
#ifndef FAST_MIN_WAVES
#define FAST_MIN_WAVES 5  // 96 VGPRs: 5 workgroups per CU measured best on C3 (4: 9.2 ms, 5: 8.5, 6: 9.1 with more spills;
                          // with MASK halves: 4: 5.48 ms, 5: 5.34, 6: 5.80)
#endif
#ifndef CLS_NARROW_CANON_BITS
#define CLS_NARROW_CANON_BITS 9  // LDS tables of the narrow class on a strand-symmetric index (at most 160 lookups per read): 2^bits entries
#endif
#ifndef FAST_MIN_WAVES_POLY
#define FAST_MIN_WAVES_POLY 4  // the polytomy kernels keep more state: at 96 VGPRs they spill 34 of them (C3s12: 5: 12.3 ms, 4: 11.95, 6: 13.4)
#endif
#ifndef FAST_MIN_WAVES_WIDE
#define FAST_MIN_WAVES_WIDE 4  // the 16-slot class keeps more state per lane: forcing 96 VGPRs on it spills 74 of them
#endif
constexpr uint32_t SET_EMPTY = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t popc64(uint64_t m) { return (uint32_t)__popcll(m); }

// first index in [lo, hi) whose value is >= key (hi if none)
__device__ __forceinline__ uint32_t lower_bound_g(const uint32_t* __restrict__ post, uint32_t lo, uint32_t hi, uint32_t key) {
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (post[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// For every slot s whose bit is set in `need`: move lo[s] (LO_SIDE) or hi[s] (!LO_SIDE) to
// lower_bound(key) inside [lo[s], hi[s]).  All slots advance in lock step, one load per slot in
// flight per round, so the dependent HBM/L2 round trips of a lane's k-mers overlap.
template <int SLOTS, bool LO_SIDE>
__device__ __forceinline__ void multi_lower_bound(const uint32_t* __restrict__ post, uint32_t (&lo)[SLOTS],
                                                  uint32_t (&hi)[SLOTS], uint32_t key, uint32_t need) {
    uint32_t other[SLOTS];  // the end of the range that must survive the search
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) other[s] = LO_SIDE ? hi[s] : lo[s];
    for (;;) {
        uint32_t mid[SLOTS], v[SLOTS];
        bool any = false;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            mid[s] = lo[s] + ((hi[s] - lo[s]) >> 1);
            v[s] = 0;
            if (((need >> s) & 1u) && lo[s] < hi[s]) { v[s] = post[mid[s]]; any = true; }
        }
        if (!any) break;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (((need >> s) & 1u) && lo[s] < hi[s]) { if (v[s] < key) lo[s] = mid[s] + 1; else hi[s] = mid[s]; }
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
        if ((need >> s) & 1u) { if (LO_SIDE) hi[s] = other[s]; else lo[s] = other[s]; }
}

// the same over groups of at most 5 slots (bounds the registers of wide kernels)
template <int SLOTS, bool LO_SIDE>
__device__ __forceinline__ void multi_lower_bound_grouped(const uint32_t* __restrict__ post, uint32_t (&lo)[SLOTS],
                                                          uint32_t (&hi)[SLOTS], uint32_t key, uint32_t need) {
    if constexpr (SLOTS <= 5) {
        multi_lower_bound<SLOTS, LO_SIDE>(post, lo, hi, key, need);
    } else {
        constexpr int G = 4;
        static_assert(SLOTS % G == 0, "SLOTS must be a multiple of 4 above 5");
#pragma unroll
        for (int g0 = 0; g0 < SLOTS; g0 += G) {
            uint32_t l[G], h[G];
#pragma unroll
            for (int i = 0; i < G; ++i) { l[i] = lo[g0 + i]; h[i] = hi[g0 + i]; }
            multi_lower_bound<G, LO_SIDE>(post, l, h, key, (need >> g0) & ((1u << G) - 1));
#pragma unroll
            for (int i = 0; i < G; ++i) { lo[g0 + i] = l[i]; hi[g0 + i] = h[i]; }
        }
    }
}

// rare, kept out of line: the interval has stored elements on both sides of it
__device__ __forceinline__ bool member_search(const uint32_t* __restrict__ post, uint32_t lo, uint32_t hi, bool closed,
                                           uint32_t c0, uint32_t c1) {
    const uint32_t i = lower_bound_g(post, lo, hi, c0);
    const uint32_t v = post[i];
    return closed ? (v < c1) : (v == c0);
}

// Is clade c = pre interval [c0, c1) a member of the k-mer's node set?
// The k-mer's stored elements inside the current parent's interval are
// post[lo, hi), with vlo = post[lo], vhi = post[hi-1] cached in registers.
__device__ __forceinline__ bool member_of(const uint32_t* __restrict__ post, uint32_t lo, uint32_t hi, uint32_t vlo,
                                          uint32_t vhi, bool closed, uint32_t c0, uint32_t c1) {
    if (vhi < c0 || vlo >= c1) return false;
    if (closed) {  // tips: member <=> a tip inside [c0, c1)
        if (vlo >= c0 || vhi < c1) return true;
        return member_search(post, lo, hi, true, c0, c1);  // vlo < c0 and vhi >= c1: look inside
    }
    // explicit list: member <=> c0 itself is stored
    if (vlo == c0 || vhi == c0) return true;
    if (vlo > c0) return false;
    return member_search(post, lo, hi, false, c0, c1);
}

// MurmurHash3_x64_128(bytes[0..len), seed 0).0 of the k-mer, and of its first `mlen` bytes
// (the "minimizer" prefix, kmers_map.rs:10-13; hash("") == 0 covers mSize == 0, :131-134),
// in one pass over the bytes.  Byte-at-a-time on purpose: small register footprint.
__device__ __forceinline__ void hash_kmer_and_prefix(const uint8_t* p, uint32_t len, uint32_t mlen, uint64_t& h_out,
                                                     uint64_t& mz_out) {
    Mur3 m, pm;  // full k-mer / prefix
    uint64_t k1 = 0, k2 = 0;
    bool pm_done = (mlen == 0);
    uint64_t mz = 0;  // murmur3 of the empty message is 0
#pragma unroll 1
    for (uint32_t i = 0; i < len; ++i) {
        const uint32_t pos = i & 15u;
        const uint64_t b = p[i];
        if (pos < 8) k1 |= b << (8 * pos); else k2 |= b << (8 * (pos - 8));
        if (!pm_done && i + 1 == mlen) {  // the prefix ends inside this block: finish its hash from the partial words
            const uint32_t t = mlen & 15u;
            mz = (t == 0) ? (pm.block(k1, k2), pm.finish(0, 0, 0, mlen)) : pm.finish(k1, k2, t, mlen);
            pm_done = true;
        }
        if (pos == 15) {
            m.block(k1, k2);
            if (!pm_done) pm.block(k1, k2);
            k1 = k2 = 0;
        }
    }
    h_out = m.finish(k1, k2, len & 15u, len);
    mz_out = mz;
}

struct WaveCtx {
    uint8_t* seq;    // LDS: upper-cased read
    uint32_t* set;   // LDS: distinct-hit set (keys = table slot indices)
    uint32_t* ent;   // LDS: per k-mer, postings offset of its (first-seen) hit or SET_EMPTY
    uint32_t* cnt;   // general path: |K_c| per non-LEAF child
    uint32_t* only;  // general path: |K_c \ R_c| per non-LEAF child
    bool child_global;  // cnt/only live in global scratch (huge polytomies) rather than LDS: needs agent-scope fences
    uint32_t* red;   // LDS: 3 x 4 words for the cross-wave sums of a multi-wave group (unused when one wave places a read)
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

// ---- group = the WAVES wavefronts that place one read together (1: a wavefront; >1: the workgroup) ----
template <int WAVES> __device__ __forceinline__ void grp_sync() { if constexpr (WAVES == 1) wave_sync(); else __syncthreads(); }
template <int WAVES> __device__ __forceinline__ bool grp_any(bool p) {
    if constexpr (WAVES == 1) return __ballot(p) != 0; else return __syncthreads_or(p ? 1 : 0) != 0;
}
// Sum over the group of up to 4 per-wave values (each wave passes its own partial sums).  `red` = 3
// rotating LDS buffers of 4 words, `round` a counter every thread of the group advances identically:
// one barrier per call (the buffer two rounds ahead is cleared while nobody can still be reading it).
template <int WAVES>
__device__ __forceinline__ void grp_sum4(uint32_t (&v)[4], uint32_t* red, uint32_t& round) {
    if constexpr (WAVES > 1) {
        uint32_t* cur = red + 4 * (round % 3), *nxt = red + 4 * ((round + 1) % 3);
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) if (v[i]) atomicAdd(&cur[i], v[i]);
        }
        if (threadIdx.x < 4) nxt[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = cur[i];
        ++round;
    }
}

// Phases A0-A2 for one read, shared by both postings formats.  Returns false when the
// read's record has already been written (error statuses); otherwise cx.ent[j] holds, for
// query k-mer j < nk, the postings offset (FMT_LIST) or the tip-set id (FMT_SPLIT) of its index entry
// if j is the FIRST query k-mer with that hash and the entry passes the minimizer-bucket filter, else SET_EMPTY.
template <int SLOTS, int SET_BITS, bool STATS, int WAVES = 1, bool SPLIT = false>
__device__ __forceinline__ bool match_phase(const DbDev& db, const WaveCtx& cx, const uint8_t* __restrict__ bases,
                                            uint64_t b0, uint64_t b1, uint32_t r, cls_placement* __restrict__ out,
                                            cls_query_stats* __restrict__ stats, uint32_t& nk_out) {
    const uint32_t lane = threadIdx.x & 63;
    constexpr uint32_t GS = 64 * WAVES;                       // threads that share the read
    const uint32_t tid = WAVES == 1 ? lane : threadIdx.x;
    (void)tid;
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
        return false;
    }
    const uint64_t nk64 = 2 * (L64 - k + 1);
    if (nk64 > (uint64_t)(GS * SLOTS)) {
        put_stats(nk64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nk64, 0, 0, 0);
        write_record(out, r, CLS_ERR_READ_TOO_LONG, 0, 0, 0, 0);
        return false;
    }
    const uint32_t L = (uint32_t)L64, nf = L - k + 1, nk = 2 * nf;
    // ---- A1. load, upper-case, validate (reverse_complement panics on non-ACGT, kmers_map.rs:440);
    // LDS holds the forward string followed by its reverse complement, so that query k-mer j
    // (forward ones first, then those of the reverse complement, kmers_map.rs:387-395) is the
    // k contiguous bytes at kmer_start(j).
    if (WAVES > 1 && tid < 12) cx.red[tid] = 0;
    bool bad = false;
#pragma unroll 1
    for (uint32_t i = tid; i < L; i += GS) {
        uint8_t c = bases[b0 + i];
        if (c >= 'a' && c <= 'z') c -= 32;
        bad |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
        cx.seq[i] = c;
        cx.seq[2 * L - 1 - i] = c ^ ((c & 2) ? 0x04 : 0x15);  // A<->T, C<->G
    }
#pragma unroll 1
    for (uint32_t i = tid; i < (1u << SET_BITS); i += GS) cx.set[i] = SET_EMPTY;
    if (grp_any<WAVES>(bad)) {
        put_stats(0, 0, 0, 0);  // the reference dies inside build_kmer_from_string, before any count exists
        write_record(out, r, CLS_ERR_INVALID_BASE, 0, 0, 0, 0);
        return false;
    }
    grp_sync<WAVES>();
    // ---- A2. hash every k-mer + its minimizer prefix, probe the table, apply the
    // minimizer-bucket filter and the distinct-hash de-duplication; one k-mer per
    // (slot, lane); the surviving postings offsets are staged in LDS (cx.ent).
    const uint8_t* seq = cx.seq;
    const uint32_t m_eff = db.m_eff;
    auto kmer_start = [&](uint32_t j) -> const uint8_t* { return seq + (j < nf ? j : L + (j - nf)); };
#pragma unroll 1
    for (uint32_t base = 0; base < nk; base += GS) {
        const uint32_t j = base + tid;
        bool hit = false;
        uint32_t tidx = 0;
        uint64_t loc = 0, mz = 0;
        if (j < nk) {
            const uint64_t h = murmur3_h1_lds(kmer_start(j), k);  // (eight bytes per LDS read; up to 15 bytes past the k-mer, inside the block's LDS)
            mz = murmur3_h1_lds(kmer_start(j), m_eff);
            uint64_t idx = h & db.table_mask;
#pragma unroll 1
            for (;;) {
                const Slot sl = db.table[idx];  // FMT_SPLIT: TSlot{hash, set | bucket << 32}, set == 0: empty
                if (SPLIT ? (uint32_t)sl.loc == 0u : sl.loc == SLOT_EMPTY) break;
                if (sl.hash == h) { hit = true; loc = sl.loc; tidx = (uint32_t)idx; break; }
                idx = (idx + 1) & db.table_mask;
            }
        }
        // the bucket's key must be one of the query's minimizers (kmers_map.rs:295-297)
        bool ok = false;
        uint64_t bk = 0;
        if (hit) { bk = db.bucket_key[SPLIT ? (uint32_t)(loc >> 32) & (uint32_t)LOC_BUCKET_MASK : (uint32_t)(loc & LOC_BUCKET_MASK)]; ok = (bk == mz); }
        uint64_t pend = __ballot(hit && !ok);
        while (pend) {  // never taken for an index built by `cls build-db`
            const int src = __ffsll((unsigned long long)pend) - 1;
            const uint64_t B = ((uint64_t)__shfl((uint32_t)(bk >> 32), src) << 32) | __shfl((uint32_t)bk, src);
            bool f = false;
#pragma unroll 1
            for (uint32_t jj = lane; jj < nk; jj += 64) {
                uint64_t hh, mm;
                hash_kmer_and_prefix(kmer_start(jj), m_eff, m_eff, hh, mm);
                f |= mm == B;
            }
            const bool any = __ballot(f) != 0;
            if ((int)lane == src) ok = any;
            pend &= pend - 1;
        }
        uint32_t ent = SET_EMPTY;
        if (hit && ok) {  // HashSet<u64> of hashes: count each distinct hash once
            uint32_t pos = (tidx * 2654435761u) >> (32 - SET_BITS);
#pragma unroll 1
            for (;;) {
                const uint32_t old = atomicCAS(&cx.set[pos], SET_EMPTY, tidx);
                if (old == SET_EMPTY) { ent = SPLIT ? ((uint32_t)loc & 0x3FFFFFFFu) : (uint32_t)(loc >> LOC_BUCKET_BITS); break; }
                if (old == tidx) break;
                pos = (pos + 1) & ((1u << SET_BITS) - 1);
            }
        }
        cx.ent[j] = ent;  // j < GS*SLOTS always
    }
    grp_sync<WAVES>();
    nk_out = nk;
    return true;
}

template <int SLOTS, int SET_BITS, bool STATS, bool BINARY, int WAVES = 1>
__device__ __forceinline__ void place_read(const DbDev db, const PlaceParams prm, const WaveCtx cx, const uint8_t* __restrict__ bases,
                           uint64_t b0, uint64_t b1, uint32_t r, cls_placement* __restrict__ out,
                           cls_query_stats* __restrict__ stats) {
    const uint32_t lane = threadIdx.x & 63;
    constexpr uint32_t GS = 64 * WAVES;                       // threads that share the read
    const uint32_t tid = WAVES == 1 ? lane : threadIdx.x;
    (void)tid;
    uint32_t nk = 0;
    if (!match_phase<SLOTS, SET_BITS, STATS, WAVES>(db, cx, bases, b0, b1, r, out, stats, nk)) return;
    auto put_stats = [&](uint32_t nk_, uint32_t nm, uint32_t nr, uint64_t lp) {
        if (STATS && stats && lane == 0) {
            uint64_t* s = reinterpret_cast<uint64_t*>(stats + r);
            s[0] = (uint64_t)nk_ | ((uint64_t)nm << 32);
            s[1] = (uint64_t)nr;
            s[2] = lp;
        }
    };
    const uint32_t* __restrict__ post = db.postings;
    uint32_t n_m = 0, n_root = 0;
    uint64_t leafp = 0;
    // ---- A3. per-k-mer state: the stored elements below the root -------------------------
    uint32_t lo[SLOTS], hi[SLOTS], vlo[SLOTS], vhi[SLOTS];
    uint32_t act = 0, closedm = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const uint32_t j = s * GS + tid;
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
    uint32_t rnd = 0;  // grp_sum4 rounds (identical in every thread of the group)
    if (STATS) for (int o = 32; o > 0; o >>= 1) leafp += ((uint64_t)__shfl_xor((uint32_t)(leafp >> 32), o) << 32) | __shfl_xor((uint32_t)leafp, o);
    {
        uint32_t v[4] = {n_m, n_root, STATS ? (uint32_t)leafp : 0u, STATS ? (uint32_t)(leafp >> 32) : 0u};
        grp_sum4<WAVES>(v, cx.red, rnd);
        n_m = v[0]; n_root = v[1];
        if (WAVES > 1) leafp = ((uint64_t)v[3] << 32) + v[2];  // (per-wave low words summed: < 2^32 in practice)
    }
    if (STATS) put_stats(nk, n_m, n_root, leafp);
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
        if (BINARY || m <= 2) {
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
            { uint32_t v[4] = {cnt_a, cnt_b, both, 0}; grp_sum4<WAVES>(v, cx.red, rnd); cnt_a = v[0]; cnt_b = v[1]; both = v[2]; }
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
        } else if (!BINARY) {
            // ---- general path (polytomies): per-child counters in memory --------------------
            for (uint32_t i = tid; i < m; i += GS) {
                __hip_atomic_store(&cx.only[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (WAVES > 1) __hip_atomic_store(&cx.cnt[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (WAVES > 1) { if (cx.child_global) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); __syncthreads(); }
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
                if (lane == 0) {
                    if (WAVES == 1) __hip_atomic_store(&cx.cnt[ci], cn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else if (cn) __hip_atomic_fetch_add(&cx.cnt[ci], cn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            uint32_t U = 0;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                U += popc64(__ballot(nin[s] >= 1));
                if (nin[s] == 1) __hip_atomic_fetch_add(&cx.only[which[s]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            { uint32_t v[4] = {U, 0, 0, 0}; grp_sum4<WAVES>(v, cx.red, rnd); U = v[0]; }
            if (cx.child_global) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            grp_sync<WAVES>();
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
            grp_sync<WAVES>();
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
        // narrow every k-mer's element range to the chosen clade's interval (n0-1 itself excluded);
        // the binary searches of the SLOTS k-mers a lane owns run interleaved so that their
        // dependent HBM/L2 round trips overlap.
        uint32_t need_lo = 0, need_hi = 0;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            if (!(act & (1u << s))) continue;
            if (vhi[s] < n0 || vlo[s] >= n1) { act &= ~(1u << s); continue; }
            if (vlo[s] < n0) need_lo |= 1u << s;
            if (vhi[s] >= n1) need_hi |= 1u << s;
        }
        if (__ballot(need_lo != 0)) {
            multi_lower_bound_grouped<SLOTS, true>(post, lo, hi, n0, need_lo);  // < hi because vhi >= n0
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) if (need_lo & (1u << s)) vlo[s] = post[lo[s]];
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if ((need_lo & (1u << s)) && vlo[s] >= n1) { act &= ~(1u << s); need_hi &= ~(1u << s); }
        }
        if (__ballot(need_hi != 0)) {
            multi_lower_bound_grouped<SLOTS, false>(post, lo, hi, n1, need_hi);  // > lo because vlo < n1
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) if (need_hi & (1u << s)) vhi[s] = post[hi[s] - 1];
        }
    }
}

// ---- FMT_SPLIT: every node set closed, every clade has 0 or 2 children ---------------------
// Per k-mer only (vlo, vhi, x) live in registers: the smallest / largest tip inside the current
// clade and the record index of the split that parts them at their LCA (cls_device.h).
template <int SLOTS, int SET_BITS, bool STATS, int WAVES = 1, bool POLY = false>
__device__ __forceinline__ void place_read_split(const DbDev db, const PlaceParams prm, const WaveCtx cx,
                                                 const uint8_t* __restrict__ bases, uint64_t b0, uint64_t b1, uint32_t r,
                                                 cls_placement* __restrict__ out, cls_query_stats* __restrict__ stats,
                                                 uint32_t profile_stop) {
    const uint32_t lane = threadIdx.x & 63;
    constexpr uint32_t GS = 64 * WAVES;                       // threads that share the read
    const uint32_t tid = WAVES == 1 ? lane : threadIdx.x;
    (void)tid;
    uint32_t nk = 0;
    if (!match_phase<SLOTS, SET_BITS, STATS, WAVES, true>(db, cx, bases, b0, b1, r, out, stats, nk)) return;
    if (profile_stop == 1) { write_record(out, r, 0xFE, (int32_t)cx.ent[lane], 0, 0, 0); return; }  // profiling aid only
    const uint4* __restrict__ recs = reinterpret_cast<const uint4*>(db.postings);
    const uint4* __restrict__ sets = reinterpret_cast<const uint4*>(db.sets);  // {x, first tip | lg << 27, last tip | root << 31, n_leaf}
    uint32_t vlo[SLOTS], vhi[SLOTS], x[SLOTS];
    uint32_t act = 0, n_m = 0, n_root = 0;
    uint64_t leafp = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const uint32_t j = s * GS + tid;
        const uint32_t sid = (j < nk) ? cx.ent[j] : SET_EMPTY;
        const bool is_new = sid != SET_EMPTY;
        const uint4 sr = sets[is_new ? sid : 0u];  // set 0: "no such k-mer" {0, MAX, 0, 0}
        if (STATS && is_new) leafp += sr.w;
        const bool has_root = is_new && (sr.z >> 31) != 0;
        const bool has_tips = sr.y != 0xFFFFFFFFu;
        vlo[s] = has_tips ? (sr.y & DIRECT_TIP_MASK) : 0xFFFFFFFFu;
        vhi[s] = sr.z & 0x7FFFFFFFu;
        x[s] = sr.x;
        if (has_root && has_tips) act |= 1u << s;
        n_m += popc64(__ballot(is_new));
        n_root += popc64(__ballot(is_new && has_root));
    }
    uint32_t rnd = 0;
    if (STATS) for (int o = 32; o > 0; o >>= 1) leafp += ((uint64_t)__shfl_xor((uint32_t)(leafp >> 32), o) << 32) | __shfl_xor((uint32_t)leafp, o);
    {
        uint32_t v[4] = {n_m, n_root, STATS ? (uint32_t)leafp : 0u, STATS ? (uint32_t)(leafp >> 32) : 0u};
        grp_sum4<WAVES>(v, cx.red, rnd);
        n_m = v[0]; n_root = v[1];
        if (WAVES > 1) leafp = ((uint64_t)v[3] << 32) + v[2];  // (per-wave low words summed: < 2^32 in practice)
    }
    if (STATS && stats && lane == 0) {
        uint64_t* s = reinterpret_cast<uint64_t*>(stats + r);
        s[0] = (uint64_t)nk | ((uint64_t)n_m << 32);
        s[1] = (uint64_t)n_root;
        s[2] = leafp;
    }
    if (profile_stop == 2) { write_record(out, r, 0xFE, (int32_t)(vlo[0] + vhi[1] + x[2] + act), 0, 0, 0); return; }
    // ---- B. thresholds (as in place_read) ------------------------------------------------------
    if (n_m == 0) { write_record(out, r, CLS_UNCLASSIFIABLE_NO_MATCH, 0, 0, 0, 0); return; }
    if (n_root == 0) { write_record(out, r, CLS_UNCLASSIFIABLE_NO_ROOT, 0, 0, 0, 0); return; }
    const DNode* __restrict__ nodes = db.nodes;
    if (!(nodes[0].flags & 1u)) { write_record(out, r, CLS_ERR_ROOT_NO_CHILDREN, 0, 0, 0, 0); return; }
    {
        const double expected = round((double)n_m * prm.min_match_coverage);
        const uint64_t exp_usize = (expected != expected) ? 0ull : (uint64_t)expected;
        if ((uint64_t)n_root < exp_usize) { write_record(out, r, CLS_UNCLASSIFIABLE_COVERAGE, (int32_t)n_root, 0, 0, 0); return; }
    }
    // ---- C. descent ---------------------------------------------------------------------------
    const bool rm = prm.remove_intersection != 0;
    uint32_t prow = 0;
    int32_t iteration = 0;
    for (;;) {
        ++iteration;
        if (iteration > prm.max_iterations) { write_record(out, r, CLS_ERR_MAX_ITER, 0, 0, (uint32_t)iteration, 0); return; }
        const uint32_t fc = uniform(nodes[prow].first_child);
        const uint32_t m = uniform(nodes[prow].n_nonleaf);  // the non-LEAF children come first
        if (POLY && (uniform(nodes[prow].flags) >> 8) != 2) {
            // ---- a clade that does not have exactly two children (polytomy after support collapse) -------
            // The Cartesian tree breaks ties to the left, so the splits between a k-mer's occupied children
            // form a right-going chain: walk the non-LEAF children left to right, stepping a k-mer past a
            // child's end with one record read when it has tips on both sides of it.
            uint32_t s_vlo[SLOTS], s_vhi[SLOTS], s_x[SLOTS], nin[SLOTS], which[SLOTS];
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) { s_vlo[s] = vlo[s]; s_vhi[s] = vhi[s]; s_x[s] = x[s]; nin[s] = 0; which[s] = 0; }
            const uint32_t s_act = act;
            auto step_right = [&](uint32_t c_end, uint32_t& live) {  // restrict every live k-mer to its tips >= c_end
                constexpr int G = SLOTS <= 5 ? SLOTS : 4;
#pragma unroll
                for (int g0 = 0; g0 < SLOTS; g0 += G) {
                    uint4 t[G];
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        const int s = g0 + i;
                        const bool str = s < SLOTS && ((live >> s) & 1u) && vlo[s] < c_end && vhi[s] >= c_end;
                        t[i] = recs[str ? x[s] : 0u];
                    }
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        const int s = g0 + i;
                        if (s >= SLOTS || !((live >> s) & 1u)) continue;
                        if (vhi[s] < c_end) live &= ~(1u << s);                        // nothing at or beyond c_end
                        else if (vlo[s] < c_end) { vlo[s] = t[i].z; x[s] = t[i].w; }   // first tip >= c_end, split of the rest
                    }
                }
            };
            for (uint32_t i = tid; i < m; i += GS) {
                __hip_atomic_store(&cx.only[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (WAVES > 1) __hip_atomic_store(&cx.cnt[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (WAVES > 1) { if (cx.child_global) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); __syncthreads(); }
            uint32_t live = act;
            for (uint32_t ci = 0; ci < m; ++ci) {
                const uint32_t c_end = uniform(nodes[fc + ci].pre) + uniform(nodes[fc + ci].size);
                uint32_t cn = 0;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    const bool in = ((live >> s) & 1u) && vlo[s] < c_end;  // vlo >= start of child ci for a live k-mer
                    if (in) { if (nin[s] == 0) which[s] = ci; if (nin[s] < 2) ++nin[s]; }
                    cn += popc64(__ballot(in));
                }
                if (lane == 0) {
                    if (WAVES == 1) __hip_atomic_store(&cx.cnt[ci], cn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else if (cn) __hip_atomic_fetch_add(&cx.cnt[ci], cn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                step_right(c_end, live);
            }
            uint32_t U = 0;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                U += popc64(__ballot(nin[s] >= 1));
                if (nin[s] == 1) __hip_atomic_fetch_add(&cx.only[which[s]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            { uint32_t v[4] = {U, 0, 0, 0}; grp_sum4<WAVES>(v, cx.red, rnd); U = v[0]; }
            if (cx.child_global) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            grp_sync<WAVES>();
            uint32_t n_pass = 0, n_best = 0, best_row = 0;
            int32_t best_one = 0, best_rest = 0, best_diff = 0;
            for (uint32_t base = 0; base < m; base += 64) {
                const uint32_t ci = base + lane;
                bool pass = false;
                int32_t one = 0, rest = 0;
                if (ci < m) {
                    const uint32_t cn = __hip_atomic_load(&cx.cnt[ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t on = __hip_atomic_load(&cx.only[ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (cn) { one = (int32_t)(rm ? on : cn); rest = (int32_t)(rm ? U - cn : U - on); pass = one > rest; }
                }
                uint64_t pm = __ballot(pass);
                while (pm) {
                    const int src = __ffsll((unsigned long long)pm) - 1;
                    const int32_t o1 = __shfl(one, src), r1 = __shfl(rest, src);
                    const int32_t diff = o1 - r1;
                    if (n_pass == 0 || diff > best_diff) { best_diff = diff; n_best = 1; best_row = fc + base + src; best_one = o1; best_rest = r1; }
                    else if (diff == best_diff) ++n_best;
                    ++n_pass;
                    pm &= pm - 1;
                }
            }
            grp_sync<WAVES>();
            if (n_pass == 0) {
                if (iteration == 1) write_record(out, r, CLS_UNCLASSIFIABLE_LEVEL1, 0, 0, 1, 0);
                else write_record(out, r, CLS_MAX_RESOLUTION, 0, 0, (uint32_t)iteration, nodes[prow].id);
                return;
            }
            if (n_pass > 1 && n_best != 1) { write_record(out, r, CLS_INCONCLUSIVE, (int32_t)n_pass, 0, (uint32_t)iteration, nodes[prow].id); return; }
            if (uniform(nodes[best_row].n_nonleaf) == 0) {
                write_record(out, r, CLS_IDENTITY_FOUND, best_one, best_rest, (uint32_t)iteration, nodes[best_row].id);
                return;
            }
            // back to the state at the top of this level, step past the children before the chosen one, enter it
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) { vlo[s] = s_vlo[s]; vhi[s] = s_vhi[s]; x[s] = s_x[s]; }
            act = s_act;
            for (uint32_t row = fc; row < best_row; ++row) step_right(uniform(nodes[row].pre) + uniform(nodes[row].size), act);
            {
                const uint32_t c0 = uniform(nodes[best_row].pre), c_end = c0 + uniform(nodes[best_row].size);
                constexpr int G = SLOTS <= 5 ? SLOTS : 4;
#pragma unroll
                for (int g0 = 0; g0 < SLOTS; g0 += G) {
                    uint4 t[G];
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        const int s = g0 + i;
                        const bool str = s < SLOTS && ((act >> s) & 1u) && vlo[s] < c_end && vlo[s] != c0 && vhi[s] >= c_end;
                        t[i] = recs[str ? x[s] : 0u];
                    }
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        const int s = g0 + i;
                        if (s >= SLOTS || !((act >> s) & 1u)) continue;
                        if (vlo[s] >= c_end || vlo[s] == c0) act &= ~(1u << s);            // not below the chosen clade
                        else if (vhi[s] >= c_end) { vhi[s] = t[i].x; x[s] = t[i].y; }     // keep the part inside it
                    }
                }
            }
            prow = best_row;
            continue;
        }
        uint32_t a0 = 0, a1 = 0, bend = 0;
        if (m >= 1) { a0 = uniform(nodes[fc].pre); a1 = a0 + uniform(nodes[fc].size); }
        // the second child starts at a1 whatever its kind; it is scored only if it is not a LEAF
        uint32_t cnt_a = 0, cnt_b = 0, both = 0;
        if (m >= 1) {
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const bool on = (act >> s) & 1u;
                const bool ina = on && vlo[s] < a1;             // vlo >= a0 always: the range lies below the parent
                const bool inb = on && m == 2 && vhi[s] >= a1;  // vhi < end of the parent's interval always
                cnt_a += popc64(__ballot(ina));
                cnt_b += popc64(__ballot(inb));
                both += popc64(__ballot(ina && inb));
            }
        }
        (void)bend;
        { uint32_t v[4] = {cnt_a, cnt_b, both, 0}; grp_sum4<WAVES>(v, cx.red, rnd); cnt_a = v[0]; cnt_b = v[1]; both = v[2]; }
        const uint32_t U = cnt_a + cnt_b - both;
        uint32_t n_pass = 0, n_best = 0, best = 0;
        int32_t best_one = 0, best_rest = 0, best_diff = 0;
        for (int c = 0; c < 2; ++c) {  // (one, rest) exactly as in place_read
            const uint32_t cn = c ? cnt_b : cnt_a;
            if (cn == 0) continue;
            const uint32_t on = cn - both;
            const int32_t one = (int32_t)(rm ? on : cn);
            const int32_t rest = (int32_t)(rm ? U - cn : U - on);
            if (one > rest) {
                const int32_t diff = one - rest;
                if (n_pass == 0 || diff > best_diff) { best_diff = diff; n_best = 1; best = c; best_one = one; best_rest = rest; }
                else if (diff == best_diff) ++n_best;
                ++n_pass;
            }
        }
        if (n_pass == 0) {
            if (iteration == 1) write_record(out, r, CLS_UNCLASSIFIABLE_LEVEL1, 0, 0, 1, 0);
            else write_record(out, r, CLS_MAX_RESOLUTION, 0, 0, (uint32_t)iteration, nodes[prow].id);
            return;
        }
        if (n_pass > 1 && n_best != 1) {
            write_record(out, r, CLS_INCONCLUSIVE, (int32_t)n_pass, 0, (uint32_t)iteration, nodes[prow].id);
            return;
        }
        const uint32_t crow = fc + best;
        if (uniform(nodes[crow].n_nonleaf) == 0) {
            write_record(out, r, CLS_IDENTITY_FOUND, best_one, best_rest, (uint32_t)iteration, nodes[crow].id);
            return;
        }
        prow = crow;
        // narrow to the chosen clade: ONE 16-byte read for a k-mer with tips on both sides, none
        // otherwise.  The reads of a lane's SLOTS k-mers are issued back to back (record 0 stands in
        // for "no read") so that their HBM round trips overlap.
        constexpr int G = SLOTS <= 5 ? SLOTS : 4;  // reads in flight per lane
#pragma unroll
        for (int g0 = 0; g0 < SLOTS; g0 += G) {
            uint4 t[G];
            uint32_t need = 0;
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int s = g0 + i;
                if (s >= SLOTS) { t[i] = uint4{0, 0, 0, 0}; continue; }
                const bool on = (act >> s) & 1u;
                const bool gone = best == 0 ? (vlo[s] >= a1 || vlo[s] == a0) : (vhi[s] < a1);
                if (on && gone) act &= ~(1u << s);
                const bool straddles = on && !gone && (best == 0 ? vhi[s] >= a1 : vlo[s] < a1);
                if (straddles) need |= 1u << i;
                t[i] = recs[straddles ? x[s] : 0u];
            }
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int s = g0 + i;
                if (s >= SLOTS || !((need >> i) & 1u)) continue;
                if (best == 0) { vhi[s] = t[i].x; x[s] = t[i].y; }
                else { vlo[s] = t[i].z; x[s] = t[i].w; }
            }
        }
        if (best != 0) {
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (((act >> s) & 1u) && vlo[s] == a1) act &= ~(1u << s);  // the clade itself is the tip: nothing below it
        }
    }
}

template <int SLOTS, int SET_BITS, bool STATS, bool POLY>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, MIN_WAVES_PER_EU) void place_split_kernel(
    DbDev db, PlaceParams prm, const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets,
    const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_len,
    cls_placement* __restrict__ out, cls_query_stats* __restrict__ stats, uint32_t seq_cap, uint32_t profile_stop,
    uint32_t* __restrict__ child_ws, uint32_t ws_stride) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t per_wave = seq_cap + (4u << SET_BITS) + 4u * 64 * SLOTS;
    WaveCtx cx;
    cx.seq = smem + wave * per_wave;
    cx.set = reinterpret_cast<uint32_t*>(cx.seq + seq_cap);
    cx.ent = cx.set + (1u << SET_BITS);
    cx.red = nullptr;
    const uint32_t gw = blockIdx.x * WAVES_PER_BLOCK + wave;
    cx.child_global = child_ws != nullptr;
    cx.cnt = child_ws ? child_ws + (size_t)gw * 2 * ws_stride
                      : reinterpret_cast<uint32_t*>(smem + WAVES_PER_BLOCK * per_wave) + (size_t)wave * 2 * ws_stride;  // LDS (ws_stride may be 0)
    cx.only = cx.cnt + ws_stride;
    const uint32_t n_waves = gridDim.x * WAVES_PER_BLOCK;
    const uint32_t n_list = *list_len;
    for (uint32_t i = gw; i < n_list; i += n_waves) {
        const uint32_t r = list[i];
        const uint64_t b0 = offsets[r], b1 = offsets[r + 1];
        place_read_split<SLOTS, SET_BITS, STATS, 1, POLY>(db, prm, cx, bases, b0, b1, r, out, stats, profile_stop);
        wave_sync();
    }
}

template <int SLOTS, int SET_BITS, bool STATS, bool BINARY>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, MIN_WAVES_PER_EU) void place_wave_kernel(DbDev db, PlaceParams prm,
                                                                          const uint8_t* __restrict__ bases,
                                                                          const uint64_t* __restrict__ offsets,
                                                                          const uint32_t* __restrict__ list,
                                                                          const uint32_t* __restrict__ list_len,
                                                                          cls_placement* __restrict__ out,
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
    cx.red = nullptr;
    const uint32_t gw = blockIdx.x * WAVES_PER_BLOCK + wave;
    cx.child_global = child_ws != nullptr;
    cx.cnt = child_ws ? child_ws + (size_t)gw * 2 * ws_stride
                      : reinterpret_cast<uint32_t*>(smem + WAVES_PER_BLOCK * per_wave) + (size_t)wave * 2 * ws_stride;  // LDS (ws_stride may be 0)
    cx.only = cx.cnt + ws_stride;
    const uint32_t n_waves = gridDim.x * WAVES_PER_BLOCK;
    const uint32_t n_list = *list_len;
    for (uint32_t i = gw; i < n_list; i += n_waves) {
        const uint32_t r = list[i];
        const uint64_t b0 = offsets[r], b1 = offsets[r + 1];
        place_read<SLOTS, SET_BITS, STATS, BINARY>(db, prm, cx, bases, b0, b1, r, out, stats);
        wave_sync();
    }
}

// ---- one WORKGROUP per read: the same per-k-mer code with the read's k-mers spread over WAVES
// wavefronts (reads of up to 64*WAVES*SLOTS k-mers: marker-gene length queries) ----------------------
template <int WAVES, int SLOTS, int SET_BITS, bool STATS, bool BINARY, bool SPLIT>
__global__ __launch_bounds__(64 * WAVES) void place_block_kernel(DbDev db, PlaceParams prm, const uint8_t* __restrict__ bases,
                                                                const uint64_t* __restrict__ offsets,
                                                                const uint32_t* __restrict__ list,
                                                                const uint32_t* __restrict__ list_len,
                                                                cls_placement* __restrict__ out, cls_query_stats* __restrict__ stats,
                                                                uint32_t seq_cap, uint32_t* __restrict__ child_ws, uint32_t ws_stride) {
    extern __shared__ __align__(16) uint8_t smem[];
    WaveCtx cx;
    cx.seq = smem;
    cx.set = reinterpret_cast<uint32_t*>(cx.seq + seq_cap);
    cx.ent = cx.set + (1u << SET_BITS);
    cx.red = cx.ent + 64 * WAVES * SLOTS;
    cx.child_global = child_ws != nullptr;
    cx.cnt = child_ws ? child_ws + (size_t)blockIdx.x * 2 * ws_stride : cx.red + 16;  // LDS (ws_stride may be 0)
    cx.only = cx.cnt + ws_stride;
    const uint32_t n_list = *list_len;
    for (uint32_t i = blockIdx.x; i < n_list; i += gridDim.x) {
        const uint32_t r = list[i];
        const uint64_t b0 = offsets[r], b1 = offsets[r + 1];
        if constexpr (SPLIT) place_read_split<SLOTS, SET_BITS, STATS, WAVES, !BINARY>(db, prm, cx, bases, b0, b1, r, out, stats, 0u);
        else place_read<SLOTS, SET_BITS, STATS, BINARY, WAVES>(db, prm, cx, bases, b0, b1, r, out, stats);
        __syncthreads();
    }
}

// ---- fast path: FMT_SPLIT + direct 2-bit k-mer table ---------------------------------------------
// Same algorithm as place_read_split, written for instruction economy (the split kernel was bound by
// instruction issue, the scalar unit above all): no MurmurHash (the 2-bit code of a k-mer indexes the
// direct table), no probe loop, no per-slot branches (an absent / inactive k-mer is the state
// {vlo = MAX, vhi = 0}), one DPP reduction per level instead of 3*SLOTS ballots, and the two children's
// node records fetched ahead of the counting.
// Where the third, fourth and fifth child of a clade start (DbDev.kids): 16 bytes through the scalar unit.
struct skids_t { uint32_t s[4]; };
__device__ __forceinline__ skids_t load_kids(const uint32_t* kids, uint32_t row) {
    typedef __attribute__((address_space(4))) const uint32_t as4_u32;
    as4_u32* p = (as4_u32*)(uintptr_t)(kids + 4 * (size_t)__builtin_amdgcn_readfirstlane(row));
    skids_t r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.s[i] = p[i];
    return r;
}

struct FastCtx {
    uint8_t* ascii;    // LDS: upper-cased read, L bytes
    uint32_t* packed;  // LDS: the read 2 bits per base, base i at bits 2(i & 15) of word i >> 4
    uint32_t* set;     // LDS: distinct-hit set keyed by header record offset
    uint32_t* gkey;    // LDS: tip-set groups of the read's k-mers: key (root split, or the single tip) ...
    uint32_t* gcnt;    // ... and how many distinct k-mers carry it
    uint4* stage;      // LDS: the groups compacted {first tip, last tip, split, weight}; lies over set/gkey/gcnt
    uint32_t* ccnt;    // LDS (polytomy trees): k-mers under each non-LEAF child of the current clade ...
    uint32_t* conly;   // ... and those under no other one
    uint32_t* cpre;    // ... and where each child's pre-order interval starts (+ the end of the last one)
};
constexpr int FAST_SLOTS_NARROW = 5;       // slots of the narrow wave-per-read class (CLS_SLOTS[0]); the wide one skips its reads
constexpr uint32_t FAST_MAX_ARITY = 256;  // non-LEAF children per clade the fast path keeps counters for

// Front of the fast path, shared with order_key_kernel: the read -> LDS (upper-cased, validated),
// 2 bits per base, then per query k-mer its 2-bit code and the direct-table entry `e` = set id | tier << 30
// (0: the k-mer is not in the index); `key` = what makes the k-mer distinct (its code).
// `canonical`: one lookup per window j < nk = nf, of the smaller of the k-mer and its reverse complement;
// kw = how many distinct query k-mers the lookup stands for (2, or 1 for a palindrome).
// Returns false if the read holds a character other than ACGT.
// FAT: the entry comes from the denormalised 16-byte table (`direct16`) together with its set record `rec` =
// {root split, first tip | lg << 27, last tip | has_root << 31}: no second, dependent read.
struct FatRec { uint32_t x, vlo_lg, vhi_root; };
// The three per-read LDS tables back to empty, sixteen bytes per store (the tables start on 16-byte boundaries: they
// double as the uint4 staging area); table_bits >= 2 whenever SET_BITS > 0.
template <int SET_BITS>
__device__ __forceinline__ void clear_tables(const FastCtx& cx, uint32_t table_bits, uint32_t lane) {
    if constexpr (SET_BITS == 0) {
        if (lane == 0) cx.set[0] = SET_EMPTY;
    } else {
        const uint4 ones = make_uint4(SET_EMPTY, SET_EMPTY, SET_EMPTY, SET_EMPTY), zero = make_uint4(0, 0, 0, 0);
        uint4* const s4 = reinterpret_cast<uint4*>(cx.set);
        uint4* const k4 = reinterpret_cast<uint4*>(cx.gkey);
        uint4* const c4 = reinterpret_cast<uint4*>(cx.gcnt);
#pragma unroll 1
        for (uint32_t i = lane; i < (1u << table_bits) / 4; i += 64) { s4[i] = ones; k4[i] = ones; c4[i] = zero; }
    }
}

template <int SLOTS, int SET_BITS, bool ADDR32, bool FAT = false>
__device__ __forceinline__ bool fast_front(const DbDev& db, const FastCtx& cx, const uint8_t* __restrict__ bases, uint64_t b0,
                                           uint32_t L, uint32_t nf, uint32_t nk, uint32_t (&e)[SLOTS], uint32_t (&key)[SLOTS], uint32_t (&kw)[SLOTS],
                                           FatRec (&rec)[SLOTS], uint32_t sample_shift = 32, bool canonical = false, uint32_t table_bits = SET_BITS) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t k = db.k;
    bool bad = false;
#pragma unroll 1
    for (uint32_t i = lane; i < L; i += 64) {
        uint8_t c = bases[b0 + i];
        if (c >= 'a' && c <= 'z') c -= 32;
        bad |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
        cx.ascii[i] = c;
    }
    clear_tables<SET_BITS>(cx, table_bits, lane);  // (only the part of the tables this read will use)
    if (__ballot(bad)) return false;
    wave_sync();
    {
        const uint32_t n_words = (L + 15) >> 4;
#pragma unroll 1
        for (uint32_t w = lane; w < n_words; w += 64) {
            uint32_t acc = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // 4 ascii bytes -> 8 bits: (c >> 1) & 3 = A0 C1 T2 G3
                uint32_t t = *reinterpret_cast<const uint32_t*>(cx.ascii + 16 * w + 4 * q);  // reads past L stay inside the buffer
                t = (t >> 1) & 0x03030303u;
                t = (t | (t >> 6) | (t >> 12) | (t >> 18)) & 0xFFu;
                acc |= t << (8 * q);
            }
            cx.packed[w] = acc;
        }
    }
    wave_sync();
    const uint32_t* __restrict__ direct = db.direct;
    const uint32_t kmask = (k == 16) ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const uint32_t j = s * 64 + lane;
        const bool valid = j < nk;
        const bool rc = j >= nf;
        const uint32_t p = valid ? (rc ? (nf - 1) - (j - nf) : j) : 0;  // window start; the rc list runs backwards over the windows
        const uint32_t w = p >> 4, sh = (2 * p) & 31;
        const uint32_t d0 = cx.packed[w], d1 = cx.packed[w + 1];
        uint32_t code = (uint32_t)((((uint64_t)d1 << 32) | d0) >> sh) & kmask;
        // reverse complement of the window: complement = code ^ 0b10.., then reverse the 2-bit groups
        uint32_t rcc = __builtin_bitreverse32(code ^ (0xAAAAAAAAu & kmask));
        rcc = ((rcc >> 1) & 0x55555555u) | ((rcc & 0x55555555u) << 1);
        rcc >>= (32 - 2 * k);
        const bool palindrome = code == rcc;
        code = canonical ? (rcc < code ? rcc : code) : (rc ? rcc : code);  // canonical: the same entry whichever strand was read
        // sample_shift < 32: look up only the k-mers whose scrambled code has its top bits clear (a content-
        // based sample, the same k-mers in every read that contains them); 32 = all
        const bool take = valid && (sample_shift >= 32 || ((code * 0x9E3779B1u) >> sample_shift) == 0);
        if constexpr (FAT) {
            const uint4 v = ldx<uint4, ADDR32>(reinterpret_cast<const uint4*>(db.direct16), take ? code : 0u);
            e[s] = take ? v.w : 0u;
            rec[s] = FatRec{v.x, take ? v.y : 0xFFFFFFFFu, take ? v.z : 0u};
        } else {
            const uint32_t v = ldx<uint32_t, ADDR32>(direct, take ? code : 0u);
            e[s] = take ? v : 0u;
        }
        key[s] = code;
        kw[s] = !take ? 0u : (canonical && !palindrome) ? 2u : 1u;  // canonical: the window stands for the k-mer and its reverse complement
    }
    return true;
}

// The same front for an index WITHOUT a direct table (k > 15: the reference's default is k = 35): the read goes to
// LDS as forward ++ reverse-complement ASCII, each k-mer is keyed by MurmurHash3 (kmers_map.rs:157-159), probed in
// the HBM hash table (TSlot: one 16-byte read per probe) and passed through the minimizer-bucket filter
// (kmers_map.rs:295-297; the wave-wide search for a foreign bucket's key is never taken for built indexes).
// `e` = set id | tier << 30 as above, `key` = the table slot (distinct per k-mer hash).  `ib`: index bytes asked for.
template <int SLOTS, int SET_BITS, bool ADDR32>
__device__ __forceinline__ bool hash_front(const DbDev& db, const FastCtx& cx, const uint8_t* __restrict__ bases, uint64_t b0,
                                           uint32_t L, uint32_t nf, uint32_t nk, uint32_t (&e)[SLOTS], uint32_t (&key)[SLOTS], uint32_t (&kw)[SLOTS],
                                           uint32_t table_bits, uint32_t& ib) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t k = db.k, m_eff = db.m_eff;
    bool bad = false;
#pragma unroll 1
    for (uint32_t i = lane; i < L; i += 64) {
        uint8_t c = bases[b0 + i];
        if (c >= 'a' && c <= 'z') c -= 32;
        bad |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
        cx.ascii[i] = c;
        cx.ascii[2 * L - 1 - i] = c ^ ((c & 2) ? 0x04 : 0x15);  // A<->T, C<->G
    }
    clear_tables<SET_BITS>(cx, table_bits, lane);  // (only the part of the tables this read will use)
    if (__ballot(bad)) return false;
    wave_sync();
    const uint8_t* seq = cx.ascii;
    auto kmer_start = [&](uint32_t j) -> const uint8_t* { return seq + (j < nf ? j : L + (j - nf)); };
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const uint32_t j = s * 64 + lane;
        bool hit = false;
        uint64_t mz = 0;
        uint32_t ent = 0, bucket = 0, slot = 0;
        const uint32_t* __restrict__ mzb = db.mz_bucket;  // the bucket keyed by the hash of a k-mer's first m characters, tabulated (cls_device.h)
        if (j < nk) {
            const uint64_t h = murmur3_h1_lds(kmer_start(j), k);
            if (!mzb) mz = murmur3_h1_lds(kmer_start(j), m_eff);  // the "minimizer": the hash of the first m characters (kmers_map.rs:10-13)
            uint64_t idx = h & db.table_mask;
            const uint4* __restrict__ ft = reinterpret_cast<const uint4*>(db.table);  // TSlot = {hash lo, hash hi, set, bucket | tier << 30}
#pragma unroll 1
            for (;;) {
                const uint4 a = ft[idx];
                ib += 16;
                if (a.z == 0) break;  // empty slot
                if ((((uint64_t)a.y << 32) | a.x) == h) {
                    hit = true;
                    ent = a.z | (a.w & ~SET_ID_MASK);
                    bucket = a.w & (uint32_t)LOC_BUCKET_MASK;
                    slot = (uint32_t)idx;
                    break;
                }
                idx = (idx + 1) & db.table_mask;
            }
        }
        bool ok = false;
        uint64_t bk = 0;
        if (hit && mzb) { ok = mzb[lds_prefix_code(kmer_start(j), m_eff)] == bucket; ib += 4; }
        else if (hit) { bk = db.bucket_key[bucket]; ok = (bk == mz); ib += 8; }
        uint64_t pend = __ballot(hit && !ok);
        while (pend) {  // the bucket's key may still be the minimizer of another query k-mer
            const int src = __ffsll((unsigned long long)pend) - 1;
            const uint64_t B = ((uint64_t)__shfl((uint32_t)(bk >> 32), src) << 32) | __shfl((uint32_t)bk, src);
            const uint32_t Bi = __shfl(bucket, src);
            bool f = false;
#pragma unroll 1
            for (uint32_t jj = lane; jj < 2 * nf; jj += 64) {  // every k-mer of the read, both strands
                f |= mzb ? mzb[lds_prefix_code(kmer_start(jj), m_eff)] == Bi : murmur3_h1_lds(kmer_start(jj), m_eff) == B;
            }
            const bool any = __ballot(f) != 0;
            if ((int)lane == src) ok = any;
            pend &= pend - 1;
        }
        e[s] = (hit && ok) ? ent : 0u;
        key[s] = slot;
        kw[s] = (hit && ok) ? 1u : 0u;
    }
    return true;
}

// Inactive group of the descent: {GRP_INACTIVE, 0}.  Below 2^31, so that `a < b` can be taken as the sign of a - b.
constexpr uint32_t GRP_INACTIVE = 0x7FFFFFFFu;
__device__ __forceinline__ uint32_t mask_lt(uint32_t a, uint32_t b) { return (uint32_t)((int32_t)(a - b) >> 31); }  // ~0 if a < b (both < 2^31)
// The same with the shift hidden from the optimiser: it otherwise recognises "sign-extended compare", turns every use of the
// mask back into v_cmp + v_cndmask pairs and the AND / OR of two masks into s_and_b64 / s_or_b64 on the scalar unit.  Kept
// opaque, `(a & m) | (b & ~m)` stays one v_bfi_b32 and three-input mask logic one v_bitop3_b32.
__device__ __forceinline__ uint32_t sign_mask(uint32_t d) {
    uint32_t m;
    asm("v_ashrrev_i32_e32 %0, 31, %1" : "=v"(m) : "v"(d));
    return m;
}
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }  // m ? a : b, bit by bit

// The descent (C) on the read's tip-set groups, staged in cx.stage[0 .. n_sets).  (A two-kernel form -- front writes the groups, this runs as its own kernel at 39
// VGPRs and 8 waves per SIMD -- was measured: 7.4 ms against 6.85 ms fused on C3; more reads in flight do not pay for
// writing and re-reading 1.6 GB of groups.)
// STATS: `ib` accumulates (per lane) the index bytes the descent asks for: 32 per node record, 8 per split half.
template <bool ADDR32, bool POLY, bool STATS, bool MANY, bool REG2>
__device__ __forceinline__ void descend_groups(const DbDev& db, const PlaceParams& prm, const FastCtx& cx, uint32_t n_sets, snode_t P,
                                               uint32_t r, cls_placement* __restrict__ out, uint32_t& ib) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n_chunks = uniform((n_sets + 63) >> 6);  // wave-uniform; >= 1 here (some k-mer has the root and tips... or none: then 0)
    if (n_sets + lane < 64 * n_chunks) cx.stage[n_sets + lane] = uint4{GRP_INACTIVE, 0u, 0u, 0u};  // pad the last chunk with inactive entries
    wave_sync();
    uint32_t vlo, vhi, x, wt;      // chunk 0 and ...
    uint32_t vlo2, vhi2, x2, wt2;  // ... chunk 1 live in registers (a 150 bp read has 70-140 groups): no LDS traffic, no loop for them
    // (REG2: not for the polytomy kernels with the MurmurHash3 front -- C3s35 66.2 against 67.8 M placements/s with it,
    // C3s12 (direct table) 79.5 against 76.7)
    {
        const uint4 g = n_chunks ? cx.stage[lane] : uint4{GRP_INACTIVE, 0u, 0u, 0u};
        vlo = g.x; vhi = g.y; x = g.z; wt = g.w;
        const uint4 g2 = (REG2 && n_chunks > 1) ? cx.stage[64 + lane] : uint4{GRP_INACTIVE, 0u, 0u, 0u};
        vlo2 = g2.x; vhi2 = g2.y; x2 = g2.z; wt2 = g2.w;
    }
    constexpr uint32_t CL = REG2 ? 2u : 1u;  // first chunk that stays in LDS
    // MANY = false: the caller knows that every group is in the register chunks (nearly every 150 bp read): the copy of the
    // descent without the LDS loops and without the three tests a level for them
    const uint32_t n_loop = MANY ? n_chunks : 0u;
    // sum of every group's weight, active or not: the constant of the level decision (bit 31 of a weight word: the group holds bits)
    uint32_t w_all = (wt & 0xFFFFFFu) + (wt2 & 0xFFFFFFu);
#pragma unroll 1
    for (uint32_t c = CL; c < n_loop; ++c) w_all += cx.stage[c * 64 + lane].w & 0xFFFFFFu;
    w_all = wave_sum(w_all);
    // ---- C. descent -----------------------------------------------------------------------------------
    // record x = 8-byte halves 2x (left part), 2x+1 (right part).  The wave-per-read kernels read the copy with MASK halves
    // (cls_device.h): a group whose part has become narrow holds its tips as bits of `x` (bit i = row lo + i; bit 31 of its
    // weight word says so) and is narrowed by bit arithmetic from then on, no read.
    constexpr bool MASKS = true;
    const uint32_t* __restrict__ half = (MASKS && db.postings2) ? db.postings2 : db.postings;
    // One step along a group's chain of occupied children at a polytomy: the first tip at or beyond `bound` (there is one:
    // the caller saw hi >= bound) and what describes the tips from there on -- the right half of the group's split record
    // (the Cartesian tree breaks ties to the left: that half IS "the first tip beyond this child, the split of the rest"),
    // or, for a group that holds bits, a shift.
    auto next_part = [&](uint32_t bound, uint32_t& v, uint32_t& xx, bool& bits) {
        if (MASKS && bits) {
            const uint32_t up = xx >> ((bound - v) & 31u);
            const uint32_t z = (uint32_t)__ffs((int)up) - 1u;
            v = bound + z; xx = up >> (z & 31u);
        } else {
            const uint2 t = ld_half<ADDR32>(half, xx, 1u);
            if (STATS) ib += 8;
            v = t.x & ~MASK_HALF; xx = t.y;
            if (MASKS) bits = (t.x & MASK_HALF) != 0u;
        }
    };
    const bool rm = prm.remove_intersection != 0;
    skids_t K{};  // polytomy trees: where the current clade's third .. fifth child start, fetched with its node record
    if (POLY) K = load_kids(db.kids, 0);
    int32_t iteration = 0;
    for (;;) {
        ++iteration;
        if (iteration > prm.max_iterations) { write_record(out, r, CLS_ERR_MAX_ITER, 0, 0, (uint32_t)iteration, 0); return; }
        const uint32_t fc = P.s[2], m = P.s[3];
        if (POLY && (P.s[7] >> 8) != 2) {
            // ---- a clade that does not have exactly two children (polytomy after support collapse) -----------------
            // The Cartesian tree breaks ties to the left, so the splits between a group's occupied children form a
            // right-going chain: every group walks ITS OWN chain (child under its first tip by binary search over the
            // children's intervals, then one 8-byte read per further occupied child), adding its weight to the
            // per-child counters in LDS.  Cost per group = children it has tips under, whatever the clade's arity.
            const DNode* __restrict__ nodes = db.nodes;
            // (one, rest), place_sequence.rs:369-395, with |R_c| = |U| - |only_c| and |R_c \ K_c| = |U| - |K_c|
            uint32_t n_pass = 0, n_best = 0, best_row = 0;
            int32_t best_one = 0, best_rest = 0, best_diff = 0;
            if (m <= 4) {
                // ---- at most four non-LEAF children (nearly every polytomy of a support-collapsed tree): the children's
                // boundaries are scalars (the node record + its `kids` record), the child under a tip is three compares,
                // the per-child counters are packed per-lane registers reduced like the binary counters: no LDS, no
                // atomics, no extra round trip for the children's records.  Cost per group = children it has tips under.
                const uint32_t B0 = P.s[0] + 1, B1 = P.s[6], B2 = K.s[0], B3 = K.s[1], B4 = K.s[2];
                const uint32_t last_end = m == 0 ? B0 : m == 1 ? B1 : m == 2 ? B2 : m == 3 ? B3 : B4;  // end of the last non-LEAF child
                uint32_t cA = 0, cB = 0, oA = 0, oB = 0, u_lane = 0;  // |K_0| | |K_1| << 16, |K_2| | |K_3| << 16; likewise |only_c|; |U|
                auto walk4 = [&](uint32_t v, uint32_t vh, uint32_t xx, uint32_t wf) {
                    uint32_t nin = 0, which = 0;
                    bool bits = (wf >> 31) != 0u;  // the group holds its tips as bits of xx (MASK halves)
                    const uint32_t w = wf & 0xFFFFFFu;
                    while (v < last_end) {  // v lies under exactly one non-LEAF child
                        const uint32_t c = (v >= B1 ? 1u : 0u) + (v >= B2 ? 1u : 0u) + (v >= B3 ? 1u : 0u);
                        const uint32_t c_end = c == 0 ? B1 : c == 1 ? B2 : c == 2 ? B3 : B4;
                        const uint32_t inc = w << (16 * (c & 1u));
                        if (c < 2) cA += inc; else cB += inc;
                        if (nin == 0) which = c;
                        if (nin < 2) ++nin;
                        if (vh < c_end) break;                                         // no tip beyond this child
                        next_part(c_end, v, xx, bits);                                 // first tip beyond it, the split of the rest
                    }
                    if (nin == 1) { const uint32_t inc = w << (16 * (which & 1u)); if (which < 2) oA += inc; else oB += inc; }
                    u_lane += nin ? w : 0u;
                };
                walk4(vlo, vhi, x, wt);
                if (REG2) walk4(vlo2, vhi2, x2, wt2);
#pragma unroll 1
                for (uint32_t c = CL; c < n_loop; ++c) { const uint4 g = cx.stage[c * 64 + lane]; walk4(g.x, g.y, g.z, g.w); }
                cA = wave_sum(cA); cB = wave_sum(cB); oA = wave_sum(oA); oB = wave_sum(oB);
                const uint32_t U = wave_sum(u_lane);
#pragma unroll
                for (uint32_t ci = 0; ci < 4; ++ci) {
                    if (ci >= m) break;
                    const uint32_t cn = ((ci < 2 ? cA : cB) >> (16 * (ci & 1u))) & 0xFFFFu, on = ((ci < 2 ? oA : oB) >> (16 * (ci & 1u))) & 0xFFFFu;
                    if (!cn) continue;  // K_c empty: not a candidate (:329)
                    const int32_t one = (int32_t)(rm ? on : cn), rest = (int32_t)(rm ? U - cn : U - on);
                    if (one <= rest) continue;  // :411-417
                    const int32_t diff = one - rest;
                    if (n_pass == 0 || diff > best_diff) { best_diff = diff; n_best = 1; best_row = fc + ci; best_one = one; best_rest = rest; }
                    else if (diff == best_diff) ++n_best;
                    ++n_pass;
                }
            } else {
                // ---- any arity up to FAST_MAX_ARITY: children's intervals and per-child counters in LDS
                uint32_t last_end = 0;
                if (m) { const snode_t Lc = load_node(nodes, fc + m - 1); last_end = Lc.s[0] + Lc.s[1]; }  // the non-LEAF children come first
                if (STATS && lane == 0 && m) ib += 32;
                for (uint32_t i = lane; i < m; i += 64) { cx.ccnt[i] = 0; cx.conly[i] = 0; cx.cpre[i] = nodes[fc + i].pre; if (STATS) ib += 4; }  // children tile [pre+1, last_end)
                if (lane == 0) cx.cpre[m] = last_end;
                wave_sync();
                uint32_t u_lane = 0;
                auto walk = [&](uint32_t v, uint32_t vh, uint32_t xx, uint32_t wf) {
                    uint32_t nin = 0, which = 0;
                    bool bits = (wf >> 31) != 0u;
                    const uint32_t w = wf & 0xFFFFFFu;
                    while (v < last_end) {  // v lies under exactly one non-LEAF child: the last one that starts at or before it
                        uint32_t lo_ = 0, hi_ = m;
                        while (hi_ - lo_ > 1) { const uint32_t mid = (lo_ + hi_) >> 1; if (cx.cpre[mid] <= v) lo_ = mid; else hi_ = mid; }
                        const uint32_t c_end = cx.cpre[lo_ + 1];  // (back to back: the next child starts where this one ends)
                        atomicAdd(&cx.ccnt[lo_], w);
                        if (nin == 0) which = lo_;
                        if (nin < 2) ++nin;
                        if (vh < c_end) break;                                         // no tip beyond this child
                        next_part(c_end, v, xx, bits);                                 // first tip beyond it, the split of the rest
                    }
                    if (nin == 1) atomicAdd(&cx.conly[which], w);
                    u_lane += nin ? w : 0u;
                };
                walk(vlo, vhi, x, wt);
                if (REG2) walk(vlo2, vhi2, x2, wt2);
#pragma unroll 1
                for (uint32_t c = CL; c < n_loop; ++c) { const uint4 g = cx.stage[c * 64 + lane]; walk(g.x, g.y, g.z, g.w); }
                const uint32_t U = wave_sum(u_lane);
                wave_sync();
                for (uint32_t base = 0; base < m; base += 64) {
                    const uint32_t ci = base + lane;
                    bool pass = false;
                    int32_t one = 0, rest = 0;
                    if (ci < m) {
                        const uint32_t cn = cx.ccnt[ci], on = cx.conly[ci];
                        if (cn) { one = (int32_t)(rm ? on : cn); rest = (int32_t)(rm ? U - cn : U - on); pass = one > rest; }
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
            const uint64_t pid = ((uint64_t)P.s[5] << 32) | P.s[4];
            if (n_pass == 0) {
                if (iteration == 1) write_record(out, r, CLS_UNCLASSIFIABLE_LEVEL1, 0, 0, 1, 0);
                else write_record(out, r, CLS_MAX_RESOLUTION, 0, 0, (uint32_t)iteration, pid);
                return;
            }
            if (n_pass > 1 && n_best != 1) { write_record(out, r, CLS_INCONCLUSIVE, (int32_t)n_pass, 0, (uint32_t)iteration, pid); return; }
            P = load_node(nodes, best_row);
            K = load_kids(db.kids, best_row);
            if (STATS && lane == 0) ib += 32;
            if (P.s[3] == 0) {
                write_record(out, r, CLS_IDENTITY_FOUND, best_one, best_rest, (uint32_t)iteration, ((uint64_t)P.s[5] << 32) | P.s[4]);
                return;
            }
            // enter the chosen child [c0, c_end): step past the occupied children before it, keep the part inside it
            const uint32_t c0 = P.s[0], c_end = c0 + P.s[1];
            auto enter = [&](uint32_t& v, uint32_t& vh, uint32_t& xx, uint32_t& wf) {
                bool dead = v > vh;
                bool bits = (wf >> 31) != 0u;
                while (!dead && v < c0) {
                    if (vh < c0) { dead = true; break; }
                    next_part(bits ? c0 : 0u, v, xx, bits);  // (a group that holds bits goes straight to its first tip inside the clade)
                }
                if (!dead && v < c_end && v != c0) {  // a tip strictly below the chosen clade
                    if (vh >= c_end) {                // ... and tips beyond it: keep the part inside
                        if (MASKS && bits) { xx &= (1u << ((c_end - v) & 31u)) - 1u; vh = v + 31u - (uint32_t)__clz((int)xx); }
                        else {
                            const uint2 t = ld_half<ADDR32>(half, xx, 0u);
                            if (STATS) ib += 8;
                            vh = t.x & ~MASK_HALF; xx = t.y;
                            if (MASKS) bits = (t.x & MASK_HALF) != 0u;
                        }
                    }
                } else { v = GRP_INACTIVE; vh = 0; }
                if (MASKS) wf = (wf & 0xFFFFFFu) | (bits ? MASK_HALF : 0u);
            };
            enter(vlo, vhi, x, wt);
            if (REG2) enter(vlo2, vhi2, x2, wt2);
#pragma unroll 1
            for (uint32_t c = CL; c < n_loop; ++c) {
                uint4 g = cx.stage[c * 64 + lane];
                enter(g.x, g.y, g.z, g.w);
                cx.stage[c * 64 + lane] = g;
            }
            continue;
        }
        const uint32_t a0 = P.s[0] + 1, a1 = P.s[6];  // first child = [a0, a1), second = [a1, end of the parent)
        // The level's two memory round trips are taken off its dependent chain (waves spent two thirds of their time
        // parked on them): a group of chunk 0 that has tips on both sides of a1 will need one half of its split record
        // whichever child wins, so the whole 16-byte record is requested NOW, before the counting and the reduction ...
        // (one, rest) of place_sequence.rs:369-395 with |R_c| = |U| - |only_c|, |R_c \ K_c| = |U| - |K_c|:
        //   one_a - rest_a = |only_a| - |only_b| = |K_a| - |K_b| = -(one_b - rest_b)   for either remove_intersection,
        // so exactly one child passes `one > rest` when the two differ and none when they tie: a level only
        // needs the SIGN of |K_a| - |K_b|; the three counts themselves go into the record of the level the descent ends at
        // and are taken there (`final_counts`).  With u(v) = 1 if v < a1 else 0 (the sign bit of v - a1: every value is
        // below 2^31, an inactive group is {GRP_INACTIVE, 0}),
        //   |K_a| - |K_b| = sum w [lo < a1] - sum w [hi >= a1] = sum w (u(lo) + u(hi)) - sum w      over ALL groups
        // (an inactive group gives u(lo) + u(hi) = 0 + 1): six vector instructions a group, no lane predicates (`a && b`
        // between two compare results is an s_and_b64 on the ONE scalar unit of the CU, and the level loop was issuing
        // more scalar than vector instructions), one reduction.
        // A LEAF child is not scored (:322-324): m < 2 takes the second child's count away (u(hi) = 1 for everyone), m = 0
        // the first child's too (u(lo) = 0).
        const uint32_t a1_lo = m == 0 ? 0u : a1, a1_hi = m < 2 ? GRP_INACTIVE : a1;
        uint32_t acc = 0;
        auto count = [&](uint32_t lo_, uint32_t hi_, uint32_t w) { acc += __umul24(w, ((lo_ - a1_lo) >> 31) + ((hi_ - a1_hi) >> 31)); };
        count(vlo, vhi, wt);
        if (REG2) count(vlo2, vhi2, wt2);
#pragma unroll 1
        for (uint32_t c = CL; c < n_loop; ++c) { const uint4 g = cx.stage[c * 64 + lane]; count(g.x, g.y, g.w); }
        const int32_t diff_ab = (int32_t)(wave_sum(acc) - w_all);  // |K_a| - |K_b|
        // |K_a|, |K_b|, |K_a ^ K_b| of this level: only the level the descent ends at asks
        auto final_counts = [&](uint32_t& cnt_a, uint32_t& cnt_b, uint32_t& both) {
            uint32_t ca = 0, cb = 0, bo = 0;
            auto count3 = [&](uint32_t lo_, uint32_t hi_, uint32_t w) {
                const uint32_t ina = w & 0xFFFFFFu & mask_lt(lo_, a1);    // lo >= a0 for an active group
                const uint32_t inb = w & 0xFFFFFFu & ~mask_lt(hi_, a1);   // hi < end of the parent for an active one, 0 for an inactive one
                ca += ina; cb += inb; bo += ina & inb;        // (both are 0 or w)
            };
            count3(vlo, vhi, wt);
            if (REG2) count3(vlo2, vhi2, wt2);
#pragma unroll 1
            for (uint32_t c = CL; c < n_loop; ++c) { const uint4 g = cx.stage[c * 64 + lane]; count3(g.x, g.y, g.w); }
            cnt_a = m == 0 ? 0u : wave_sum(ca);
            cnt_b = m < 2 ? 0u : wave_sum(cb);
            both = m < 2 ? 0u : wave_sum(bo);
        };
        const uint64_t pid = ((uint64_t)P.s[5] << 32) | P.s[4];
        if (diff_ab == 0) {
            if (iteration == 1) write_record(out, r, CLS_UNCLASSIFIABLE_LEVEL1, 0, 0, 1, 0);
            else write_record(out, r, CLS_MAX_RESOLUTION, 0, 0, (uint32_t)iteration, pid);
            return;
        }
        const bool right = diff_ab < 0;
        // (Measured and rejected, C3: requesting chunk 0's whole split record before the counting, 6.87 against 6.79 ms; with both
        // children's node records a level ahead as well, 7.18; waiting for the chosen child's record only after the split reads
        // have been requested, 6.66 against 6.15 ms -- each costs more registers and scalar moves than the round trip it hides.)
#ifdef CLS_EXP_HOT_NODES
        P = load_node(db.nodes, (fc + (right ? 1u : 0u)) & 255u);  // (timing experiment: wrong placements)
#else
        P = load_node(db.nodes, fc + (right ? 1u : 0u));
#endif
        if (POLY) K = load_kids(db.kids, fc + (right ? 1u : 0u));
        __builtin_amdgcn_sched_barrier(0);  // keep the request up here (the scheduler sinks scalar loads to their first use)
        if (STATS && lane == 0) ib += 32;
        if (P.s[3] == 0) {  // no non-LEAF child below the chosen clade (update_introspection_node.rs:45-85)
            uint32_t cnt_a, cnt_b, both;
            final_counts(cnt_a, cnt_b, both);
            const uint32_t cn = right ? cnt_b : cnt_a, on = cn - both, U = cnt_a + cnt_b - both;
            write_record(out, r, CLS_IDENTITY_FOUND, (int32_t)(rm ? on : cn), (int32_t)(rm ? U - cn : U - on), (uint32_t)iteration,
                         ((uint64_t)P.s[5] << 32) | P.s[4]);
            return;
        }
        // narrow: a set with tips on both sides of a1 reads 8 bytes of its split node (the half for the side
        // taken); everything else is arithmetic on (lo, hi).  Inactive afterwards = {MAX, 0}.
        // Two steps, so that the reads of the two register chunks are in flight together (as one step per chunk the
        // second read was only requested once the first had come back: two round trips a level instead of one), and one
        // copy of the level per side (`right` is wave-uniform).
        auto narrow_side = [&](auto side) {
            constexpr bool RIGHT = decltype(side)::value;
            // (every value is below 2^31 and an active group has a0 <= lo <= hi: `u < v` is the sign of u - v, `lo != a0` that of a0 - lo)
            // `aux`: left = the group keeps a tip strictly below the first child (known before the read comes back);
            // right = the group has NO tip at or beyond a1.  `strm` = tips on both sides, `rd` = and the part is behind a read.
            auto classify = [&](uint32_t lo_, uint32_t hi_, uint32_t w_, uint32_t& strm, uint32_t& rd, uint32_t& aux) {
                const uint32_t lt = sign_mask(lo_ - a1), nge = sign_mask(hi_ - a1);  // a tip below a1 / NO tip at or beyond a1
                if (!RIGHT) { aux = lt & sign_mask(a0 - lo_); strm = aux & ~nge; }   // tips on both sides, one strictly below the first child
                else { aux = nge; strm = lt & ~nge; }                                  // tips on both sides
                rd = MASKS ? (strm & ~sign_mask(w_)) : strm;
                if (STATS) ib += rd & 8u;
            };
            auto request = [&](uint32_t x_, uint32_t rd) -> uint2 {
                return ld_half<ADDR32>(half, x_ & rd, (RIGHT ? 1u : 0u) & rd);  // (record 0: the dummy)
            };
            auto finish = [&](uint32_t& lo_, uint32_t& hi_, uint32_t& x_, uint32_t& w_, uint32_t strm, uint32_t rd, uint32_t aux, uint2 t) {
                uint32_t end = t.x, xn = t.y;  // what the read gave: the new last (left) / first (right) tip, the part's split or bits
                if (MASKS) {
                    w_ |= t.x & rd & MASK_HALF;       // the part came as bits: the group is narrowed by arithmetic from here on
                    end &= ~MASK_HALF;
                    // a group that already holds bits: rows [lo, a1) are bits [0, d), d = a1 - lo in 1 .. 31 when it has tips on both sides
                    const uint32_t d = a1 - lo_;
                    uint32_t mend, mx;
                    if (!RIGHT) { mx = x_ & ((1u << (d & 31u)) - 1u); mend = lo_ + 31u - (uint32_t)__clz((int)mx); }
                    else { const uint32_t up = x_ >> (d & 31u); const uint32_t z = (uint32_t)__ffs((int)up) - 1u; mx = up >> (z & 31u); mend = a1 + z; }
                    end = bfi(rd, end, mend);
                    xn = bfi(rd, xn, mx);
                }
                x_ = bfi(strm, xn, x_);
                uint32_t keep;
                if (!RIGHT) {
                    keep = aux;
                    hi_ = bfi(strm, end, hi_);
                } else {
                    lo_ = bfi(strm, end, lo_);
                    keep = sign_mask(a1 - lo_) & ~aux;  // something in the second child, and not just the clade itself
                }
                lo_ = bfi(keep, lo_, GRP_INACTIVE);
                hi_ &= keep;
            };
            uint32_t s0 = 0, s1 = 0, r0 = 0, r1 = 0, k0 = 0, k1 = 0;
            uint2 t0{}, t1{};
            classify(vlo, vhi, wt, s0, r0, k0);
            if (REG2) classify(vlo2, vhi2, wt2, s1, r1, k1);
            // (deep in the tree every group with tips on both sides holds bits: no read at all, and nothing to wait for)
            if (!MASKS || __builtin_amdgcn_ballot_w64((r0 | r1) != 0u) != 0ull) {
                t0 = request(x, r0);
                if (REG2) t1 = request(x2, r1);
                __builtin_amdgcn_sched_barrier(0);  // both reads on their way before either is waited for
            }
            finish(vlo, vhi, x, wt, s0, r0, k0, t0);
            if (REG2) finish(vlo2, vhi2, x2, wt2, s1, r1, k1, t1);
#pragma unroll 1
            for (uint32_t c = CL; c < n_loop; ++c) {
                uint4 g = cx.stage[c * 64 + lane];
                uint32_t sm, rm_, km;
                classify(g.x, g.y, g.w, sm, rm_, km);
                const uint2 t = request(g.z, rm_);
                finish(g.x, g.y, g.z, g.w, sm, rm_, km, t);
                cx.stage[c * 64 + lane] = g;
            }
        };
        if (right) narrow_side(std::true_type{}); else narrow_side(std::false_type{});
    }
}

template <int SLOTS, int SET_BITS, bool STATS, bool ADDR32, int MODE, bool POLY>
__device__ __forceinline__ void place_read_fast(const DbDev& db, const PlaceParams& prm, const FastCtx& cx,
                                                const uint8_t* __restrict__ bases, uint64_t b0, uint64_t b1, uint32_t r,
                                                cls_placement* __restrict__ out, cls_query_stats* __restrict__ stats,
                                                uint32_t profile_stop_arg) {
#ifdef CLS_PROFILE_HOOKS
    const uint32_t profile_stop = profile_stop_arg;  // CLS_PROFILE_STOP=1|2: truncate after a phase (timing breakdowns)
#else
    constexpr uint32_t profile_stop = 0;  // (one less live scalar in a kernel that spills SGPRs)
    (void)profile_stop_arg;
#endif
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t k = db.k;
    const uint64_t L64 = b1 - b0;
    auto put_stats = [&](uint32_t nk_, uint32_t nm, uint32_t nr, uint64_t lp) {
        if (STATS && stats && lane == 0) {
            uint64_t* s = reinterpret_cast<uint64_t*>(stats + r);
            s[0] = (uint64_t)nk_ | ((uint64_t)nm << 32);
            s[1] = (uint64_t)nr;
            s[2] = lp;
        }
    };
    if (L64 < k) { put_stats(0, 0, 0, 0); write_record(out, r, CLS_ERR_TOO_FEW_KMERS, 0, 0, 0, 0); return; }
    const uint32_t L = (uint32_t)L64, nf = L - k + 1, nk = 2 * nf;  // nk <= 64*SLOTS by classification
    // ---- A1/A2. load + validate + 2-bit pack; per k-mer: code -> direct table -> tip-set id ------------
    // CANON (an index in which every k-mer and its reverse complement carry the same tip set, i.e. one built from
    // both strands): the k-mers of window p on the two strands are reverse complements of each other, so ONE lookup
    // of the smaller of the two answers for both; half the lookups, half the slots.
    // MODE 0: direct table, both strands looked up; 1: strand-symmetric index, one lookup per window; 2: MurmurHash3 front;
    // 3 / 4: as 0 / 1 through the denormalised 16-byte table (k <= 12): the set record comes with the entry
    constexpr bool CANON = MODE == 1 || MODE == 4, HASHED = MODE == 2, FAT = MODE >= 3;
    constexpr int LS = CANON ? (SLOTS + 1) / 2 : SLOTS;
    uint32_t sid[LS], key[LS], kw[LS];
    FatRec frec[LS];
    // the wide class sizes its LDS tables by the read (clearing 3 x 2048 entries cost more than placing a 250 bp read)
    uint32_t tb = SET_BITS;
    if constexpr (SET_BITS > 9) {
        const uint32_t want = 2 * (CANON ? nf : nk);
        tb = 9;
        while ((1u << tb) < want && tb < (uint32_t)SET_BITS) ++tb;
        tb = uniform(tb);
    }
    bool valid_read;
    uint32_t ib = 0;  // STATS: index bytes this lane asked for (table entries, set records, node records, split halves)
    if constexpr (HASHED) valid_read = hash_front<LS, SET_BITS, ADDR32>(db, cx, bases, b0, L, nf, nk, sid, key, kw, tb, ib);
    else {
        valid_read = fast_front<LS, SET_BITS, ADDR32, FAT>(db, cx, bases, b0, L, nf, CANON ? nf : nk, sid, key, kw, frec, 32, CANON, tb);
        if (STATS) for (int s = 0; s < LS; ++s) ib += kw[s] ? (FAT ? 16u : 4u) : 0u;  // one table entry per looked-up window
    }
    // cls_query_stats.index_bytes: written last, after the counters (put_stats clears the field)
    auto put_index_bytes = [&]() {
        if constexpr (STATS) {
            const uint32_t total = wave_sum(ib);
            if (stats && lane == 0) reinterpret_cast<uint32_t*>(stats + r)[3] = total;
        }
    };
    if (!valid_read) {
        put_stats(0, 0, 0, 0);
        write_record(out, r, CLS_ERR_INVALID_BASE, 0, 0, 0, 0);
        return;
    }
    // distinct hashes: the FIRST k-mer with a given code (table slot) keeps its entry (HashSet<u64> semantics)
    uint32_t nm_lane = 0;
#pragma unroll
    for (int s = 0; s < LS; ++s) {
        sid[s] &= SET_ID_MASK;  // (the tier bits only matter to the locality keys)
        if (sid[s] != 0) {
            uint32_t pos = (key[s] * 2654435761u) >> (32 - tb);
#pragma unroll 1
            for (;;) {
                const uint32_t old = atomicCAS(&cx.set[pos], SET_EMPTY, key[s]);
                if (old == SET_EMPTY) break;
                if (old == key[s]) { sid[s] = 0; kw[s] = 0; break; }
                pos = (pos + 1) & ((1u << tb) - 1);
            }
        }
        nm_lane += sid[s] != 0 ? kw[s] : 0u;
    }
    if (profile_stop == 1) { write_record(out, r, 0xFE, (int32_t)(sid[0] + sid[LS - 1]), 0, 0, 0); return; }  // profiling aid
    // ---- A4. group the k-mers by tip set ----------------------------------------------------------------
    // k-mers with the same set stay together all the way down, so everything below runs on {set, number of k-mers}
    // pairs: a 150 bp read has a few dozen of them.  The first k-mer to claim a set id owns the group; only the
    // owners read the 16-byte set record.
    uint32_t pos[LS];
    uint32_t owner = 0;
#pragma unroll
    for (int s = 0; s < LS; ++s) {
        pos[s] = 0;
        if (sid[s] != 0) {
            uint32_t p = (sid[s] * 2654435761u) >> (32 - tb);
#pragma unroll 1
            for (;;) {
                const uint32_t old = atomicCAS(&cx.gkey[p], SET_EMPTY, sid[s]);
                if (old == SET_EMPTY) { owner |= 1u << s; break; }
                if (old == sid[s]) break;
                p = (p + 1) & ((1u << tb) - 1);
            }
            atomicAdd(&cx.gcnt[p], kw[s]);
            pos[s] = p;
        }
    }
    wave_sync();
#pragma unroll
    for (int s = 0; s < LS; ++s) pos[s] = ((owner >> s) & 1u) ? cx.gcnt[pos[s]] : 0u;  // now: the group's weight
    wave_sync();  // the three tables are dead from here on: the staging area lies over them
    // ---- A3. the owners read their set records; |M_root|; the groups that have tips below the root are compacted
    // into the staging area (64 per chunk; chunk 0 then lives in registers, the others stay in LDS).  A few records
    // in flight at a time: the wide class would otherwise hold 16 of them per lane in registers.
    const uint4* __restrict__ sets = reinterpret_cast<const uint4*>(db.sets2 ? db.sets2 : db.sets);  // (narrow sets as bits: MASK halves)
    uint32_t cnt = nm_lane;  // |M| in bits 0..15, |M_root| in bits 16..31 (per lane, then summed)
    uint64_t leafp = 0;
    uint32_t n_sets = 0;
    constexpr int G = LS <= 4 ? LS : 4;
#pragma unroll
    for (int g0 = 0; g0 < LS; g0 += G) {
        uint4 sr[G];  // {root split, first tip | lg << 27, last tip | has_root << 31, n_leaf}
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int s = g0 + i;
            const bool own = s < LS && ((owner >> s) & 1u);
            if constexpr (FAT) {  // the record came with the table entry (n_leaf, a statistic, still lives in the set record)
                const FatRec fr = frec[s < LS ? s : 0];
                sr[i] = own ? uint4{fr.x, fr.vlo_lg, fr.vhi_root, 0u} : uint4{0u, 0xFFFFFFFFu, 0u, 0u};
                if (STATS && own) sr[i].w = ldx<uint4, ADDR32>(sets, sid[s < LS ? s : 0]).w;
            } else {
                sr[i] = ldx<uint4, ADDR32>(sets, own ? sid[s < LS ? s : 0] : 0u);  // set 0: {0, MAX, 0, 0}
                if (STATS && own) ib += 16;
            }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int s = g0 + i;
            if (s >= LS) continue;
            const uint32_t w = pos[s];  // 0 unless this lane owns the group
            cnt += ((sr[i].z >> 31) * w) << 16;
            if (STATS) leafp += (uint64_t)w * sr[i].w;
            const bool live = ((owner >> s) & 1u) && sr[i].y != 0xFFFFFFFFu;
            const uint64_t m = __ballot(live);
            if (live)
                cx.stage[n_sets + popc64(m & ((1ull << lane) - 1))] = uint4{sr[i].y & DIRECT_TIP_MASK, sr[i].z & DIRECT_TIP_MASK, sr[i].x,
                                                                              w | ((sr[i].z & FAT_X_IS_BITS) << 1)};  // (a narrow set comes as bits: MASK halves)
            n_sets += popc64(m);
        }
    }
    cnt = wave_sum(cnt);
    const uint32_t n_m = cnt & 0xFFFFu, n_root = cnt >> 16;
    if (STATS) {
        for (int o = 32; o > 0; o >>= 1) leafp += ((uint64_t)__shfl_xor((uint32_t)(leafp >> 32), o) << 32) | __shfl_xor((uint32_t)leafp, o);
        put_stats(nk, n_m, n_root, leafp);
    }
    if (profile_stop == 2) { write_record(out, r, 0xFE, (int32_t)n_sets, 0, 0, 0); return; }
    // ---- B. thresholds ------------------------------------------------------------------------------
    if (n_m == 0) { put_index_bytes(); write_record(out, r, CLS_UNCLASSIFIABLE_NO_MATCH, 0, 0, 0, 0); return; }
    if (n_root == 0) { put_index_bytes(); write_record(out, r, CLS_UNCLASSIFIABLE_NO_ROOT, 0, 0, 0, 0); return; }
    // node records through the scalar unit: a DNode is 8 dwords {pre, size, first_child, n_nonleaf, id lo, id hi, split, flags}
    snode_t P = load_node(db.nodes, 0);
    if (STATS && lane == 0) ib += 32;
    if (!(P.s[7] & 1u)) { put_index_bytes(); write_record(out, r, CLS_ERR_ROOT_NO_CHILDREN, 0, 0, 0, 0); return; }
    {
        const double expected = round((double)n_m * prm.min_match_coverage);
        const uint64_t exp_usize = (expected != expected) ? 0ull : (uint64_t)expected;
        if ((uint64_t)n_root < exp_usize) { put_index_bytes(); write_record(out, r, CLS_UNCLASSIFIABLE_COVERAGE, (int32_t)n_root, 0, 0, 0); return; }
    }
    bool in_registers = false;
    if constexpr (!POLY) in_registers = uniform(n_sets) <= 128;  // (the binary descent keeps two chunks of 64 groups in registers)
    constexpr bool REG2 = !POLY || MODE != 2;
    if (in_registers) descend_groups<ADDR32, POLY, STATS, false, REG2>(db, prm, cx, n_sets, P, r, out, ib);
    else descend_groups<ADDR32, POLY, STATS, true, REG2>(db, prm, cx, n_sets, P, r, out, ib);
    put_index_bytes();
}

template <int SLOTS, int SET_BITS, bool STATS, bool ADDR32, int MODE, bool POLY>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, SLOTS > 5 ? FAST_MIN_WAVES_WIDE : POLY ? FAST_MIN_WAVES_POLY : FAST_MIN_WAVES) void place_fast_kernel(
    DbDev db, PlaceParams prm, const uint8_t* __restrict__ bases, const uint64_t* __restrict__ offsets,
    const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_len, uint32_t list_n, uint32_t xcd_chunks,
    cls_placement* __restrict__ out, cls_query_stats* __restrict__ stats, uint32_t ascii_cap, uint32_t profile_stop) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t wave = threadIdx.x >> 6;
    // (MODE 2 keeps forward ++ reverse-complement ASCII in `ascii_cap` bytes and packs nothing)
    const uint32_t packed_words = MODE == 2 ? 0u : ((ascii_cap >> 4) + 2 + 3) & ~3u;  // whole 16-byte units: what follows stays 16-byte aligned
    // the staging area must fit over the three tables (combinations that do not are instantiated by the dispatch
    // macros but never launched: set_bits_of)
    constexpr bool FITS = (12u << SET_BITS) >= 16u * 64 * ((MODE == 1 || MODE == 4) ? (SLOTS + 1) / 2 : SLOTS);
    if constexpr (FITS) {
    const uint32_t per_wave = ascii_cap + 4u * packed_words + (12u << SET_BITS) + (POLY ? 12u * FAST_MAX_ARITY + 16u : 0u);
    FastCtx cx;
    cx.ascii = smem + wave * per_wave;
    cx.packed = reinterpret_cast<uint32_t*>(cx.ascii + ascii_cap);
    cx.set = cx.packed + packed_words;
    cx.gkey = cx.set + (1u << SET_BITS);
    cx.gcnt = cx.gkey + (1u << SET_BITS);
    cx.stage = reinterpret_cast<uint4*>(cx.set);
    cx.ccnt = cx.gcnt + (1u << SET_BITS);
    cx.conly = cx.ccnt + FAST_MAX_ARITY;
    cx.cpre = cx.conly + FAST_MAX_ARITY;
    // Two walks over the class list, one call site (the body is large):
    //  * locality-ordered list: XCD x (workgroups with blockIdx % 8 == x share an L2) walks the x-th eighth of
    //    the list front to back, so that reads processed together share cache lines; the list holds every class
    //  * plain class list: wave w takes entries w, w + n_waves, ...
    if (xcd_chunks && SLOTS > FAST_SLOTS_NARROW && *list_len == 0) return;  // no read of the wide class in the batch
    const uint32_t n_list = xcd_chunks ? list_n : *list_len;
    const uint32_t xcd = blockIdx.x & 7u, bpx = gridDim.x >> 3;
    const uint32_t chunk = xcd_chunks ? (n_list + 7u) >> 3 : n_list;
    const uint32_t first = xcd_chunks ? (blockIdx.x >> 3) * WAVES_PER_BLOCK + wave : blockIdx.x * WAVES_PER_BLOCK + wave;
    const uint32_t stride = (xcd_chunks ? bpx : gridDim.x) * WAVES_PER_BLOCK;
    const uint32_t origin = xcd_chunks ? xcd * chunk : 0u;
    constexpr uint32_t cap = 64 * SLOTS;
    for (uint32_t ql = first; ql < chunk; ql += stride) {
        const uint32_t q = origin + ql;
        if (q >= n_list) break;
        const uint32_t r = list[q];
        const uint64_t b0 = offsets[r], b1 = offsets[r + 1];
        if (xcd_chunks) {
            const uint64_t L64 = b1 - b0;
            if (L64 >= db.k && 2 * (L64 - db.k + 1) > cap) continue;  // another class' read (classify_kernel binned it)
            if (SLOTS > FAST_SLOTS_NARROW && (L64 < db.k || 2 * (L64 - db.k + 1) <= 64 * FAST_SLOTS_NARROW)) continue;
        }
        place_read_fast<SLOTS, SET_BITS, STATS, ADDR32, MODE, POLY>(db, prm, cx, bases, b0, b1, r, out, stats, profile_stop);
        wave_sync();
    }
    }
}

// Locality key of every read for the ordering above.  Among the read's k-mers present in the index, the ones
// specific to a small clade (few tips: tier bits of the table entry; the bound is widened until a handful
// qualify) name a leaf neighbourhood through the MEDIAN of their set ids (sets are numbered in ascending first
// tip; the median is robust against the chance matches of sequencing errors, which land anywhere in the tree),
// and a MinHash over the k-mers' codes names a group of overlapping reads.
//   key_mode 0: {median set id >> block_shift, 16-bit MinHash of the specific k-mers}: leaf neighbourhood first
//   key_mode 1: {20-bit MinHash of ALL present k-mers, median set id}: locus first
// Reads the fast kernel will not place (too short / too long / bad characters) get the last key.
__device__ __forceinline__ uint64_t make_order_key(uint32_t key_mode, uint32_t median, uint32_t mh_spec, uint32_t mh_all, uint32_t block_shift, uint32_t set_bits) {
    if (key_mode == 1) return ((uint64_t)(mh_all >> 12) << set_bits) | median;
    return ((uint64_t)(median >> block_shift) << 16) | (mh_spec >> 16);
}

template <int SLOTS, bool ADDR32, bool FWD, bool HASHED>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void order_key_kernel(DbDev db, const uint8_t* __restrict__ bases,
                                                                       const uint64_t* __restrict__ offsets, uint32_t n_reads,
                                                                       uint64_t* __restrict__ keys, uint32_t* __restrict__ idx,
                                                                       uint32_t ascii_cap, uint32_t key_mode,
                                                                       uint32_t block_shift, uint32_t sample_shift, uint32_t fwd_only,
                                                                       uint32_t key_cap) {
    extern __shared__ __align__(16) uint8_t smem[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t packed_words = HASHED ? 0u : ((ascii_cap >> 4) + 2 + 3) & ~3u;  // whole 16-byte units: what follows stays 16-byte aligned
    const uint32_t per_wave = ascii_cap + 4u * packed_words + 16u;
    FastCtx cx;
    cx.ascii = smem + wave * per_wave;
    cx.packed = reinterpret_cast<uint32_t*>(cx.ascii + ascii_cap);
    cx.set = cx.packed + packed_words;  // 1 dummy entry (SET_BITS = 0)
    cx.gkey = cx.gcnt = nullptr;
    cx.stage = nullptr;
    const uint32_t gw = blockIdx.x * WAVES_PER_BLOCK + wave;
    const uint32_t n_waves = gridDim.x * WAVES_PER_BLOCK;
    for (uint32_t r = gw; r < n_reads; r += n_waves) {
        const uint64_t b0 = offsets[r], L64 = offsets[r + 1] - b0;
        uint64_t key = ~0ull;
        if (L64 >= db.k && 2 * (L64 - db.k + 1) <= (uint64_t)key_cap) {
            // every read is keyed by its first 32*SLOTS windows (64 by default: a third of the lookups of a 150 bp read
            // cost 1.3 ms less than the slightly coarser order costs the placement kernel); reads of the next class
            // (up to key_cap k-mers) too: the same sorted list then orders both wave-per-read kernels
            const uint32_t L = (uint32_t)std::min<uint64_t>(L64, 32 * SLOTS + db.k - 1), nf = L - db.k + 1, nk = 2 * nf;
            constexpr int LS = FWD ? (SLOTS + 1) / 2 : SLOTS;  // one lookup per window needs half the slots
            uint32_t e[LS], code[LS], kw[LS];
            // one lookup per window, of the smaller of the k-mer and its reverse complement: an index built from
            // both strands files the two under the same leaves, and the key then does not depend on the strand read
            bool valid_read;
            // (without a direct table FWD means "the forward k-mers only": the key then depends on the strand read)
            uint32_t ib_unused = 0;
            if constexpr (HASHED) valid_read = hash_front<LS, 0, ADDR32>(db, cx, bases, b0, L, nf, FWD ? nf : nk, e, code, kw, 0, ib_unused);
            else { FatRec unused[LS]; valid_read = fast_front<LS, 0, ADDR32>(db, cx, bases, b0, L, nf, FWD ? nf : nk, e, code, kw, unused, sample_shift, FWD); }
            if (valid_read) {
                uint32_t cand = 0, n_cand = 0;
                for (uint32_t tier = 0; tier < 4 && n_cand < 4; ++tier) {
                    cand = 0; n_cand = 0;
#pragma unroll
                    for (int s = 0; s < LS; ++s) {
                        const bool c = (e[s] & SET_ID_MASK) != 0 && (e[s] >> TIER_SHIFT) <= tier;
                        cand |= (c ? 1u : 0u) << s;
                        n_cand += popc64(__ballot(c));
                    }
                }
                if (n_cand) {
                    // median set id of the candidates: radix select, one bit per round
                    uint32_t rank = n_cand >> 1, prefix = 0;
                    for (int bit = (int)db.set_bits - 1; bit >= 0; --bit) {
                        uint32_t c0 = 0;
#pragma unroll
                        for (int s = 0; s < LS; ++s) {
                            const uint32_t t = e[s] & SET_ID_MASK;
                            const bool z = ((cand >> s) & 1u) && ((t ^ prefix) >> (bit + 1)) == 0 && !((t >> bit) & 1u);
                            c0 += popc64(__ballot(z));
                        }
                        if (rank >= c0) { rank -= c0; prefix |= 1u << bit; }
                    }
                    // groups of overlapping reads: MinHash over the k-mers (reads that share most of them share it)
                    uint32_t mh = 0xFFFFFFFFu, mha = 0xFFFFFFFFu;
#pragma unroll
                    for (int s = 0; s < LS; ++s) {
                        const uint32_t hsh = mix32(code[s]);
                        if ((cand >> s) & 1u) mh = hsh < mh ? hsh : mh;
                        if ((e[s] & SET_ID_MASK) != 0) mha = hsh < mha ? hsh : mha;
                    }
                    for (int o = 32; o > 0; o >>= 1) {
                        const uint32_t o1 = __shfl_xor(mh, o), o2 = __shfl_xor(mha, o);
                        mh = o1 < mh ? o1 : mh;
                        mha = o2 < mha ? o2 : mha;
                    }
                    key = make_order_key(key_mode, prefix, mh, mha, block_shift, db.set_bits);
                }
            }
            wave_sync();
        }
        if (lane == 0) { keys[r] = key; idx[r] = r; }
    }
}

// (Measured, C3 with MASK halves: the first 32 windows instead of 64 take 0.35 ms off this kernel and add 0.40 ms to the placement
// kernel, whose reads are then ordered less well.)
// The same key from the read's first 64 windows, written for throughput: a HALF wavefront per read (two reads in
// flight per wave: the kernel is bound by the latency of its two dependent reads, bases then table), two windows per
// lane, canonical lookups through the direct table.  Ballots are taken wave-wide and split by half.
template <bool ADDR32>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void order_key_half_kernel(DbDev db, const uint8_t* __restrict__ bases,
                                                                            const uint64_t* __restrict__ offsets, uint32_t n_reads,
                                                                            uint64_t* __restrict__ keys, uint32_t* __restrict__ idx,
                                                                            uint32_t key_mode, uint32_t block_shift,
                                                                            uint32_t key_cap) {
    constexpr uint32_t MAXL = 64 + DIRECT_MAX_K - 1 + 1;  // first 64 windows
    constexpr uint32_t WORDS = (MAXL + 15) / 16 + 2;
    __shared__ __align__(16) uint8_t s_ascii[WAVES_PER_BLOCK * 2][(MAXL + 19) & ~3u];
    __shared__ uint32_t s_packed[WAVES_PER_BLOCK * 2][WORDS];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, hl = lane & 31;
    uint8_t* ascii = s_ascii[wave * 2 + half];
    uint32_t* packed = s_packed[wave * 2 + half];
    const uint32_t k = db.k;
    const uint32_t kmask = (1u << (2 * k)) - 1u;  // k <= 15
    const uint32_t* __restrict__ direct = db.direct;
    const uint32_t pair0 = (blockIdx.x * WAVES_PER_BLOCK + wave) * 2, stride = gridDim.x * WAVES_PER_BLOCK * 2;
    auto half_of = [&](uint64_t m) { return (uint32_t)(m >> (32 * half)); };  // this half's 32 bits of a wave-wide ballot
    for (uint32_t r0 = pair0; r0 < n_reads; r0 += stride) {  // (uniform trip count per wave: r0 is the pair's first read)
        const uint32_t r = r0 + half;
        const bool have = r < n_reads;
        const uint64_t b0 = have ? offsets[r] : 0, L64 = have ? offsets[r + 1] - b0 : 0;
        const bool keyed = have && L64 >= k && 2 * (L64 - k + 1) <= (uint64_t)key_cap;
        const uint32_t L = keyed ? (uint32_t)std::min<uint64_t>(L64, 64 + k - 1) : 0, nf = keyed ? L - k + 1 : 0;
        bool bad = false;
        for (uint32_t i = hl; i < L; i += 32) {
            uint8_t c = bases[b0 + i];
            if (c >= 'a' && c <= 'z') c -= 32;
            bad |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
            ascii[i] = c;
        }
        const bool ok = keyed && half_of(__ballot(bad)) == 0;
        wave_sync();
        for (uint32_t w = hl; w < (L + 15) / 16 + 1; w += 32) {
            uint32_t acc = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // 4 ascii bytes -> 8 bits: (c >> 1) & 3 = A0 C1 T2 G3
                uint32_t t = *reinterpret_cast<const uint32_t*>(ascii + 16 * w + 4 * q);  // reads past L stay inside the buffer
                t = (t >> 1) & 0x03030303u;
                t = (t | (t >> 6) | (t >> 12) | (t >> 18)) & 0xFFu;
                acc |= t << (8 * q);
            }
            if (w < WORDS) packed[w] = acc;
        }
        wave_sync();
        uint32_t e[2], hs[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint32_t p = hl + 32 * s;
            const bool valid = ok && p < nf;
            const uint32_t w = valid ? p >> 4 : 0, sh = valid ? (2 * p) & 31 : 0;
            const uint32_t d0 = packed[w], d1 = packed[w + 1];
            uint32_t code = (uint32_t)((((uint64_t)d1 << 32) | d0) >> sh) & kmask;
            uint32_t rcc = __builtin_bitreverse32(code ^ (0xAAAAAAAAu & kmask));
            rcc = ((rcc >> 1) & 0x55555555u) | ((rcc & 0x55555555u) << 1);
            rcc >>= (32 - 2 * k);
            code = rcc < code ? rcc : code;  // the same entry whichever strand was read
            const uint32_t v = ldx<uint32_t, ADDR32>(direct, valid ? code : 0u);
            e[s] = valid ? v : 0u;
            hs[s] = mix32(code);
        }
        // candidates: k-mers present in the index that are specific to a small clade; widen the bound until a few qualify
        uint32_t cand = 0, n_cand = 0;
        for (uint32_t tier = 0; tier < 4; ++tier) {
            const bool need = n_cand < 4;  // (per half; the other half may already be done)
            uint32_t c_new = 0, n_new = 0;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bool c = (e[s] & SET_ID_MASK) != 0 && (e[s] >> TIER_SHIFT) <= tier;
                c_new |= (c ? 1u : 0u) << s;
                n_new += (uint32_t)__popc(half_of(__ballot(c)));
            }
            if (need) { cand = c_new; n_cand = n_new; }
            if (__ballot(n_cand < 4) == 0) break;
        }
        // median set id of the candidates: radix select, one bit per round
        uint32_t rank = n_cand >> 1, prefix = 0;
        for (int bit = (int)db.set_bits - 1; bit >= 0; --bit) {
            uint32_t c0 = 0;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uint32_t t = e[s] & SET_ID_MASK;
                const bool z = ((cand >> s) & 1u) && ((t ^ prefix) >> (bit + 1)) == 0 && !((t >> bit) & 1u);
                c0 += (uint32_t)__popc(half_of(__ballot(z)));
            }
            if (rank >= c0) { rank -= c0; prefix |= 1u << bit; }
        }
        uint32_t mh = 0xFFFFFFFFu, mha = 0xFFFFFFFFu;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if ((cand >> s) & 1u) mh = hs[s] < mh ? hs[s] : mh;
            if ((e[s] & SET_ID_MASK) != 0) mha = hs[s] < mha ? hs[s] : mha;
        }
        for (int o = 16; o > 0; o >>= 1) {
            const uint32_t o1 = __shfl_xor(mh, o), o2 = __shfl_xor(mha, o);
            mh = o1 < mh ? o1 : mh;
            mha = o2 < mha ? o2 : mha;
        }
        if (have && hl == 0) {
            keys[r] = (ok && n_cand) ? make_order_key(key_mode, prefix, mh, mha, block_shift, db.set_bits) : ~0ull;
            idx[r] = r;
        }
        wave_sync();
    }
}

// ---- long reads: one WORKGROUP per read, per-k-mer state in global scratch ---------------------------
// Reads with more k-mers than the register-resident kernels hold (marker genes are a few kb, BASELINE
// config 5 has 10 kb reads).  Same algorithm, written one k-mer per thread and trip: the read is not
// staged (bases are re-read from memory per k-mer), the distinct-hash set, the per-k-mer state and the
// per-child counters live in a per-workgroup slice of the workspace, and after every level the surviving
// k-mers are compacted into the other of two state buffers, so a level costs time in proportion to the
// k-mers still below the current clade.  FMT_SPLIT walks a k-mer's own chain of occupied children (cost
// per k-mer = children it has tips under, whatever the clade's arity); FMT_LIST tests every child.
constexpr int LONG_THREADS = 256;
constexpr int LONG_STATE_WORDS = 5;  // u32 arrays per state buffer: SPLIT {vlo, vhi, x}; LIST {lo, hi, vlo, vhi, closed}

constexpr uint32_t LONG_LDS_ARITY = 256;  // children intervals kept in LDS up to this many non-LEAF children
struct LongSh {
    uint32_t acc[4];
    unsigned long long acc64;
    uint32_t n_next;     // survivors appended to the next state buffer
    uint32_t n_pass, n_best, best_row;
    int32_t best_diff, best_one, best_rest;
    uint32_t cpre[LONG_LDS_ARITY + 1];  // where the current clade's non-LEAF children start (+ the end of the last)
    uint32_t cnt_l[LONG_LDS_ARITY], only_l[LONG_LDS_ARITY];  // per-child counters of clades with at most that many
};

// position of this thread's element in a list that all threads of the workgroup append to (wave-aggregated)
__device__ __forceinline__ uint32_t append_slot(bool keep, uint32_t* counter) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t m = __ballot(keep);
    if (!m) return 0;
    const int leader = __ffsll((unsigned long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
}

template <int N>
__device__ __forceinline__ void long_sum(uint32_t (&v)[N], LongSh& sh) {
    __syncthreads();
    if (threadIdx.x < N) sh.acc[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const uint32_t w = wave_sum(v[i]);
        if ((threadIdx.x & 63) == 0 && w) atomicAdd(&sh.acc[i], w);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = sh.acc[i];
}

template <bool SPLIT, bool STATS>
__global__ __launch_bounds__(LONG_THREADS) void place_long_kernel(DbDev db, PlaceParams prm, const uint8_t* __restrict__ bases,
                                                                  const uint64_t* __restrict__ offsets,
                                                                  const uint32_t* __restrict__ list,
                                                                  const uint32_t* __restrict__ list_len,
                                                                  cls_placement* __restrict__ out, cls_query_stats* __restrict__ stats,
                                                                  uint32_t* __restrict__ ws, uint64_t ws_stride_words, uint32_t cap,
                                                                  uint32_t set_size, uint32_t arity_pad) {
    __shared__ LongSh sh;
    constexpr uint32_t NT = LONG_THREADS;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    uint32_t* const buf0 = ws + (size_t)blockIdx.x * ws_stride_words;
    uint32_t* const buf1 = buf0 + (size_t)LONG_STATE_WORDS * cap;
    uint32_t* const cnt = buf1 + (size_t)LONG_STATE_WORDS * cap;
    uint32_t* const only = cnt + arity_pad;
    uint32_t* const set = buf1;               // phase A only: the distinct-hit set lies over the second state buffer
    uint32_t* const ent = buf0 + 4 * (size_t)cap;  // phase A -> A3: index entry per query k-mer
    const DNode* __restrict__ nodes = db.nodes;
    const uint32_t* __restrict__ post = db.postings;
    const uint4* __restrict__ recs = reinterpret_cast<const uint4*>(db.postings);
    const uint32_t k = db.k, m_eff = db.m_eff;
    const bool rm = prm.remove_intersection != 0;
    const uint32_t n_list = *list_len;
    for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        __syncthreads();  // the previous read's use of `sh` and of the scratch is over
        const uint32_t r = list[li];
        const uint64_t b0 = offsets[r], L64 = offsets[r + 1] - b0;
        auto put_stats = [&](uint32_t nk_, uint32_t nm, uint32_t nr, uint64_t lp) {
            if (STATS && stats && tid == 0) {
                uint64_t* s = reinterpret_cast<uint64_t*>(stats + r);
                s[0] = (uint64_t)nk_ | ((uint64_t)nm << 32);
                s[1] = (uint64_t)nr;
                s[2] = lp;
            }
        };
        auto record = [&](uint32_t status, int32_t one, int32_t rest, uint32_t levels, uint64_t clade) {
            if (tid == 0) {
                uint64_t* o = reinterpret_cast<uint64_t*>(out + r);
                o[0] = (uint64_t)(status & 0xFF) | ((uint64_t)(uint32_t)one << 32);
                o[1] = (uint64_t)(uint32_t)rest | ((uint64_t)levels << 32);
                o[2] = clade;
            }
        };
        if (L64 < k) { put_stats(0, 0, 0, 0); record(CLS_ERR_TOO_FEW_KMERS, 0, 0, 0, 0); continue; }
        const uint64_t nk64 = 2 * (L64 - k + 1);
        if (nk64 > cap) { put_stats(nk64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)nk64, 0, 0, 0); record(CLS_ERR_READ_TOO_LONG, 0, 0, 0, 0); continue; }
        const uint32_t L = (uint32_t)L64, nf = L - k + 1, nk = 2 * nf;
        // ---- A1. validate (reverse_complement panics on non-ACGT, kmers_map.rs:440) -------------------
        bool bad = false;
        for (uint32_t i = tid; i < L; i += NT) {
            uint8_t c = bases[b0 + i];
            if (c >= 'a' && c <= 'z') c -= 32;
            bad |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
        }
        for (uint32_t i = tid; i < set_size; i += NT) set[i] = SET_EMPTY;
        if (__syncthreads_or(bad ? 1 : 0)) { put_stats(0, 0, 0, 0); record(CLS_ERR_INVALID_BASE, 0, 0, 0, 0); continue; }
        // character t of query k-mer j: forward k-mers first, then those of the reverse complement (kmers_map.rs:387-395)
        auto kmer_char = [&](uint32_t j, uint32_t t) -> uint8_t {
            const bool rc = j >= nf;
            uint8_t c = bases[b0 + (rc ? (L - 1 - (j - nf) - t) : (j + t))];
            if (c >= 'a' && c <= 'z') c -= 32;
            return rc ? (uint8_t)(c ^ ((c & 2) ? 0x04 : 0x15)) : c;  // A<->T, C<->G
        };
        // ---- A2. hash, probe, minimizer-bucket filter, distinct hashes ---------------------------------
        const bool use_direct = SPLIT && db.direct != nullptr;  // k <= 15, every entry in the bucket of its own prefix: no hash, no filter
        for (uint32_t j = tid; j < nk; j += NT) {
            if (use_direct) {
                uint32_t code = 0;
                for (uint32_t t = 0; t < k; ++t) code |= (uint32_t)((kmer_char(j, t) >> 1) & 3u) << (2 * t);  // A0 C1 T2 G3, first base in the low bits
                const uint32_t sid = db.direct[code] & SET_ID_MASK;  // tip set of the k-mer, 0 = not in the index
                uint32_t e = SET_EMPTY;
                if (sid) {  // distinct k-mers are distinct codes
                    uint32_t pos = (code * 2654435761u) & (set_size - 1);
                    for (;;) {
                        const uint32_t old = atomicCAS(&set[pos], SET_EMPTY, code);
                        if (old == SET_EMPTY) { e = sid; break; }
                        if (old == code) break;
                        pos = (pos + 1) & (set_size - 1);
                    }
                }
                ent[j] = e;
                continue;
            }
            const uint64_t h = murmur3_h1([&](uint32_t t) { return kmer_char(j, t); }, k);
            const uint64_t mz = murmur3_h1([&](uint32_t t) { return kmer_char(j, t); }, m_eff);
            bool hit = false;
            uint64_t loc = 0;
            uint32_t tidx = 0;
            uint64_t idx = h & db.table_mask;
            for (;;) {
                const Slot sl = db.table[idx];  // FMT_SPLIT: TSlot{hash, set | bucket << 32}, set == 0: empty
                if (SPLIT ? (uint32_t)sl.loc == 0u : sl.loc == SLOT_EMPTY) break;
                if (sl.hash == h) { hit = true; loc = sl.loc; tidx = (uint32_t)idx; break; }
                idx = (idx + 1) & db.table_mask;
            }
            bool ok = false;
            if (hit) {
                const uint64_t bk = db.bucket_key[SPLIT ? (uint32_t)(loc >> 32) & (uint32_t)LOC_BUCKET_MASK : (uint32_t)(loc & LOC_BUCKET_MASK)];
                ok = bk == mz;
                if (!ok) {  // the bucket's key may still be the minimizer of another query k-mer (kmers_map.rs:295-297)
                    for (uint32_t jj = 0; jj < nk && !ok; ++jj)
                        ok = murmur3_h1([&](uint32_t t) { return kmer_char(jj, t); }, m_eff) == bk;
                }
            }
            uint32_t e = SET_EMPTY;
            if (hit && ok) {  // HashSet<u64> of hashes: the first k-mer to claim the entry keeps it
                uint32_t pos = (tidx * 2654435761u) & (set_size - 1);
                for (;;) {
                    const uint32_t old = atomicCAS(&set[pos], SET_EMPTY, tidx);
                    if (old == SET_EMPTY) { e = SPLIT ? (uint32_t)loc : (uint32_t)(loc >> LOC_BUCKET_BITS); break; }
                    if (old == tidx) break;
                    pos = (pos + 1) & (set_size - 1);
                }
            }
            ent[j] = e;
        }
        if (tid == 0) { sh.n_next = 0; sh.acc64 = 0; }
        __syncthreads();
        // ---- A3. state of the k-mers that can vote, compacted into buf1 ----------------------------------
        uint32_t* cur = buf1;
        uint32_t* nxt = buf0;
        uint32_t n_m = 0, n_root = 0;
        {
            uint64_t leafp = 0;
            for (uint32_t base = 0; base < nk; base += NT) {
                const uint32_t j = base + tid;
                const uint32_t off = j < nk ? ent[j] : SET_EMPTY;
                const bool is_new = off != SET_EMPTY;
                bool has_root = false, active = false;
                uint32_t st[LONG_STATE_WORDS] = {0, 0, 0, 0, 0};
                if (is_new) {
                    if constexpr (SPLIT) {
                        const SetRec sr = db.sets[off];  // (`off` is the tip-set id here)
                        if (STATS) leafp += sr.n_leaf;
                        has_root = (sr.vhi_root >> 31) != 0;
                        active = has_root && sr.vlo_lg != 0xFFFFFFFFu;
                        st[0] = sr.vlo_lg & DIRECT_TIP_MASK; st[1] = sr.vhi_root & 0x7FFFFFFFu; st[2] = sr.x;
                    } else {
                        const uint32_t w0 = post[off];
                        if (STATS) leafp += post[off + 1];
                        has_root = (w0 & POST_HAS_ROOT) != 0;
                        const uint32_t len = w0 & POST_LEN_MASK;
                        active = has_root && len != 0;
                        if (active) {
                            st[0] = off + POST_HEADER_WORDS; st[1] = st[0] + len;
                            st[2] = post[st[0]]; st[3] = post[st[1] - 1];
                            st[4] = (w0 & POST_CLOSED) ? 1u : 0u;
                        }
                    }
                }
                n_m += is_new ? 1u : 0u;
                n_root += has_root ? 1u : 0u;
                const uint32_t p = append_slot(active, &sh.n_next);
                if (active) {
#pragma unroll
                    for (int i = 0; i < (SPLIT ? 3 : LONG_STATE_WORDS); ++i) cur[(size_t)i * cap + p] = st[i];
                }
            }
            if (STATS) {
                for (int o = 32; o > 0; o >>= 1) leafp += ((uint64_t)__shfl_xor((uint32_t)(leafp >> 32), o) << 32) | __shfl_xor((uint32_t)leafp, o);
                if (lane == 0 && leafp) atomicAdd(&sh.acc64, (unsigned long long)leafp);
            }
            uint32_t v[2] = {n_m, n_root};
            long_sum<2>(v, sh);
            n_m = v[0]; n_root = v[1];
            if (STATS) put_stats(nk, n_m, n_root, (uint64_t)sh.acc64);
        }
        uint32_t n_act = sh.n_next;
        // ---- B. thresholds (as in place_read) ------------------------------------------------------------
        if (n_m == 0) { record(CLS_UNCLASSIFIABLE_NO_MATCH, 0, 0, 0, 0); continue; }
        if (n_root == 0) { record(CLS_UNCLASSIFIABLE_NO_ROOT, 0, 0, 0, 0); continue; }
        if (!(nodes[0].flags & 1u)) { record(CLS_ERR_ROOT_NO_CHILDREN, 0, 0, 0, 0); continue; }
        {
            const double expected = round((double)n_m * prm.min_match_coverage);
            const uint64_t exp_usize = (expected != expected) ? 0ull : (uint64_t)expected;
            if ((uint64_t)n_root < exp_usize) { record(CLS_UNCLASSIFIABLE_COVERAGE, (int32_t)n_root, 0, 0, 0); continue; }
        }
        // ---- C. descent --------------------------------------------------------------------------------------
        uint32_t prow = 0;
        int32_t iteration = 0;
        for (;;) {
            ++iteration;
            if (iteration > prm.max_iterations) { record(CLS_ERR_MAX_ITER, 0, 0, (uint32_t)iteration, 0); break; }
            const uint32_t fc = nodes[prow].first_child, m = nodes[prow].n_nonleaf;  // the non-LEAF children come first
            __syncthreads();
            // per-child counters: in LDS for ordinary clades (hundreds of same-address atomics per level were serialising
            // in L2), in the workspace slice for huge polytomies
            uint32_t* const cntp = m <= LONG_LDS_ARITY ? sh.cnt_l : cnt;
            uint32_t* const onlyp = m <= LONG_LDS_ARITY ? sh.only_l : only;
            for (uint32_t i = tid; i < m; i += NT) {
                __hip_atomic_store(&cntp[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&onlyp[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid == 0) { sh.n_pass = 0; sh.n_best = 0; sh.best_diff = 0; sh.n_next = 0; }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            __syncthreads();
            const uint32_t last_end = m ? nodes[fc + m - 1].pre + nodes[fc + m - 1].size : 0u;
            const bool lds_children = m <= LONG_LDS_ARITY;  // the children tile [pre + 1, last_end) back to back
            if (lds_children) {
                for (uint32_t i = tid; i < m; i += NT) sh.cpre[i] = nodes[fc + i].pre;
                if (tid == 0) sh.cpre[m] = last_end;
                __syncthreads();
            }
            uint32_t U = 0;
            // One atomic per wavefront and distinct child instead of one per k-mer: on a binary clade every k-mer of
            // the read votes for one of two counters, and 20 000 atomics on two addresses per level was the kernel.
            auto add_grouped = [&](uint32_t* arr, bool pred, uint32_t ci) {  // wave-uniform call
                uint64_t todo = __ballot(pred);
                while (todo) {
                    const int leader = __ffsll((unsigned long long)todo) - 1;
                    const uint32_t c = __shfl(ci, leader);
                    const uint64_t same = __ballot(pred && ci == c);
                    if ((int)lane == leader) __hip_atomic_fetch_add(&arr[c], (uint32_t)__popcll(same), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    todo &= ~same;
                }
            };
            for (uint32_t base = 0; base < n_act; base += NT) {
                const uint32_t j = base + tid;
                const bool valid = j < n_act;
                uint32_t nin = 0, which = 0;
                if constexpr (SPLIT) {
                    uint32_t v = valid ? cur[j] : 0xFFFFFFFFu, xx = valid ? cur[2 * (size_t)cap + j] : 0u;
                    const uint32_t vh = valid ? cur[(size_t)cap + j] : 0u;
                    bool walking = valid && v < last_end;
                    while (__ballot(walking)) {  // v lies under exactly one non-LEAF child: the last one starting at or before it
                        uint32_t lo_ = 0, c_end = 0;
                        if (walking) {
                            uint32_t hi_ = m;
                            if (lds_children) {
                                while (hi_ - lo_ > 1) { const uint32_t mid = (lo_ + hi_) >> 1; if (sh.cpre[mid] <= v) lo_ = mid; else hi_ = mid; }
                                c_end = sh.cpre[lo_ + 1];
                            } else {
                                while (hi_ - lo_ > 1) { const uint32_t mid = (lo_ + hi_) >> 1; if (nodes[fc + mid].pre <= v) lo_ = mid; else hi_ = mid; }
                                c_end = nodes[fc + lo_].pre + nodes[fc + lo_].size;
                            }
                        }
                        add_grouped(cntp, walking, lo_);
                        if (walking) {
                            if (nin == 0) which = lo_;
                            if (nin < 2) ++nin;
                            if (vh < c_end) walking = false;  // no tip beyond this child
                            else {
                                const uint4 t = recs[xx];     // first tip beyond it, and the split of the rest
                                v = t.z; xx = t.w;
                                walking = v < last_end;
                            }
                        }
                    }
                } else {
                    const uint32_t lo = valid ? cur[j] : 0u, hi = valid ? cur[(size_t)cap + j] : 0u;
                    const uint32_t vlo = valid ? cur[2 * (size_t)cap + j] : 0xFFFFFFFFu, vhi = valid ? cur[3 * (size_t)cap + j] : 0u;
                    const bool closed = valid && cur[4 * (size_t)cap + j] != 0;
                    for (uint32_t ci = 0; ci < m; ++ci) {
                        const uint32_t c0 = nodes[fc + ci].pre, c1 = c0 + nodes[fc + ci].size;
                        const bool in = valid && member_of(post, lo, hi, vlo, vhi, closed, c0, c1);
                        const uint64_t mm = __ballot(in);
                        if (mm && lane == (uint32_t)(__ffsll((unsigned long long)mm) - 1))
                            __hip_atomic_fetch_add(&cntp[ci], (uint32_t)__popcll(mm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (in) { if (nin == 0) which = ci; if (nin < 2) ++nin; }
                    }
                }
                add_grouped(onlyp, nin == 1, which);
                U += nin ? 1u : 0u;
            }
            { uint32_t v[1] = {U}; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); long_sum<1>(v, sh); U = v[0]; }
            // (one, rest), place_sequence.rs:369-395, with |R_c| = |U| - |only_c| and |R_c \ K_c| = |U| - |K_c|
            for (int pass_no = 0; pass_no < 2; ++pass_no) {
                for (uint32_t ci = tid; ci < m; ci += NT) {
                    const uint32_t cn = __hip_atomic_load(&cntp[ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!cn) continue;  // K_c empty: not a candidate (:329)
                    const uint32_t on = __hip_atomic_load(&onlyp[ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int32_t one = (int32_t)(rm ? on : cn), rest = (int32_t)(rm ? U - cn : U - on);
                    if (one <= rest) continue;  // :411-417
                    if (pass_no == 0) { atomicAdd(&sh.n_pass, 1u); atomicMax(&sh.best_diff, one - rest); }
                    else if (one - rest == sh.best_diff) {
                        if (atomicAdd(&sh.n_best, 1u) == 0) { sh.best_row = fc + ci; sh.best_one = one; sh.best_rest = rest; }
                    }
                }
                __syncthreads();
            }
            const uint32_t n_pass = sh.n_pass, n_best = sh.n_best, best_row = sh.best_row;
            // ---- PHASE 2 (place_sequence.rs:436-600) ----------------------------------------------------------
            if (n_pass == 0) {
                if (iteration == 1) record(CLS_UNCLASSIFIABLE_LEVEL1, 0, 0, 1, 0);
                else record(CLS_MAX_RESOLUTION, 0, 0, (uint32_t)iteration, nodes[prow].id);
                break;
            }
            if (n_pass > 1 && n_best != 1) { record(CLS_INCONCLUSIVE, (int32_t)n_pass, 0, (uint32_t)iteration, nodes[prow].id); break; }
            if (nodes[best_row].n_nonleaf == 0) {  // update_introspection_node.rs:13-91
                record(CLS_IDENTITY_FOUND, sh.best_one, sh.best_rest, (uint32_t)iteration, nodes[best_row].id);
                break;
            }
            // narrow every k-mer to the chosen clade; survivors go to the other buffer
            const uint32_t c0 = nodes[best_row].pre, c_end = c0 + nodes[best_row].size;
            for (uint32_t base = 0; base < n_act; base += NT) {
                const uint32_t j = base + tid;
                bool keep = false;
                uint32_t st[LONG_STATE_WORDS] = {0, 0, 0, 0, 0};
                if (j < n_act) {
                    if constexpr (SPLIT) {
                        uint32_t v = cur[j], vh = cur[(size_t)cap + j], xx = cur[2 * (size_t)cap + j];
                        bool dead = false;
                        while (v < c0) {  // step past the occupied children before the chosen one
                            if (vh < c0) { dead = true; break; }
                            const uint4 t = recs[xx];
                            v = t.z; xx = t.w;
                        }
                        if (!dead && v < c_end && v != c0) {  // a tip strictly below the chosen clade
                            if (vh >= c_end) { const uint4 t = recs[xx]; vh = t.x; xx = t.y; }  // keep the part inside it
                            keep = true;
                            st[0] = v; st[1] = vh; st[2] = xx;
                        }
                    } else {
                        uint32_t lo = cur[j], hi = cur[(size_t)cap + j], vlo = cur[2 * (size_t)cap + j], vhi = cur[3 * (size_t)cap + j];
                        const uint32_t n0 = c0 + 1;  // the clade itself excluded
                        if (!(vhi < n0 || vlo >= c_end)) {
                            keep = true;
                            if (vlo < n0) { lo = lower_bound_g(post, lo, hi, n0); vlo = post[lo]; if (vlo >= c_end) keep = false; }
                            if (keep && vhi >= c_end) { hi = lower_bound_g(post, lo, hi, c_end); vhi = post[hi - 1]; }
                            st[0] = lo; st[1] = hi; st[2] = vlo; st[3] = vhi; st[4] = cur[4 * (size_t)cap + j];
                        }
                    }
                }
                const uint32_t p = append_slot(keep, &sh.n_next);
                if (keep) {
#pragma unroll
                    for (int i = 0; i < (SPLIT ? 3 : LONG_STATE_WORDS); ++i) nxt[(size_t)i * cap + p] = st[i];
                }
            }
            __syncthreads();
            n_act = sh.n_next;
            { uint32_t* t = cur; cur = nxt; nxt = t; }
            prow = best_row;
        }
    }
}

// ---- read-length classes ----------------------------------------------------------------------
// One thread per read: reads are binned by their k-mer count into the kernel wide enough for
// them (class lists in device memory; nothing returns to the host).  Reads no kernel can hold
// get their record here.
constexpr int CLASSIFY_THREADS = 256, CLASSIFY_PER_THREAD = 4;
// class lists: 0, 1 the wave-per-read kernels; 2 the workgroup-per-read kernel; 3 .. 5 the shared launches of the LDS-tiled
// kernel; 6 its launch that gives a read a whole CU (and the list the shared launches hand reads on to); 7 the workspace
// kernel (and the list the LDS-tiled kernel spills to)
constexpr int N_LISTS = 8;
struct ClassCaps { uint32_t cap[N_LISTS]; };  // class c takes the reads of up to cap[c] k-mers that no class before it takes (0: not in this launch)
__global__ __launch_bounds__(CLASSIFY_THREADS) void classify_kernel(const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ order, uint32_t n_reads, uint32_t k, ClassCaps caps,
                                                                   uint32_t* __restrict__ lists, uint32_t* __restrict__ counts,
                                                                   cls_placement* __restrict__ out, cls_query_stats* __restrict__ stats) {
    // a workgroup bins 1024 reads: positions inside the workgroup from LDS counters, ONE global atomic per class
    __shared__ uint32_t s_cnt[N_LISTS], s_base[N_LISTS];
    if (threadIdx.x < N_LISTS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t first = blockIdx.x * (CLASSIFY_THREADS * CLASSIFY_PER_THREAD) + threadIdx.x;
    int cls_id[CLASSIFY_PER_THREAD];
    uint32_t pos[CLASSIFY_PER_THREAD], rd[CLASSIFY_PER_THREAD];
#pragma unroll
    for (int i = 0; i < CLASSIFY_PER_THREAD; ++i) {
        const uint32_t at = first + i * CLASSIFY_THREADS;
        const uint32_t r = (order && at < n_reads) ? order[at] : at;  // (`order`: the reads in locality order; the class lists then come out in that order too, near enough)
        rd[i] = r;
        cls_id[i] = -1;
        pos[i] = 0;
        if (at < n_reads) {
            const uint64_t L = offsets[r + 1] - offsets[r];
            const uint64_t nk = L < k ? 0 : 2 * (L - k + 1);
            // (class 0 includes L < k: the kernel reports CLS_ERR_TOO_FEW_KMERS)
#pragma unroll
            for (int c = N_LISTS - 1; c >= 0; --c) if (nk <= caps.cap[c]) cls_id[i] = c;
            if (cls_id[i] < 0) {
                uint64_t* o = reinterpret_cast<uint64_t*>(out + r);
                o[0] = CLS_ERR_READ_TOO_LONG; o[1] = 0; o[2] = 0;
                if (stats) {
                    uint64_t* st = reinterpret_cast<uint64_t*>(stats + r);
                    st[0] = nk > 0xFFFFFFFFull ? 0xFFFFFFFFull : nk; st[1] = 0; st[2] = 0;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < N_LISTS; ++c) {
            const uint64_t m = __ballot(cls_id[i] == c);
            if (!m) continue;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt[c], (uint32_t)__popcll(m));
            base = __shfl(base, 0);
            if (cls_id[i] == c) pos[i] = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
        }
    }
    __syncthreads();
    if (threadIdx.x < N_LISTS) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]) : 0u;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CLASSIFY_PER_THREAD; ++i) {
        const int c = cls_id[i];
        if (c >= 0) lists[(size_t)c * n_reads + s_base[c] + pos[i]] = rd[i];
    }
}

}  // namespace

namespace {
constexpr int N_CLASSES = 2;            // wave-per-read classes; class 2 = one workgroup per read
constexpr int BLK_WAVES = 8, BLK_SLOTS = 16, BLK_SET_BITS = 14;  // 8192 k-mers per read
constexpr int ORDER_KEY_BITS_MAX = TIER_SHIFT + 20;  // {MinHash, set id}
int order_key_bits(const DbDev& db) {  // + 1: reads without a key sort last with the all-ones key
    const Tuning& tn = tuning();
    const int bits = tn.order_mode == 1 ? (int)db.set_bits + 20 : std::max(1, (int)db.set_bits - tn.order_block_shift) + 16;
    return std::min<int>(64, bits + 1);
}
constexpr int CLS_SLOTS[N_CLASSES] = {5, 16};      // k-mers per read: 320 / 1024
constexpr int CLS_SET_BITS[N_CLASSES] = {9, 11};   // LDS distinct-hit set: 512 / 2048 entries
constexpr int NARROW_CANON_BITS = CLS_NARROW_CANON_BITS;

bool use_fast(const DbDev& db);
bool use_order(const DbDev& db, uint32_t n_reads) {
    return use_fast(db) && !tuning().no_order && n_reads >= 4096;
}
bool use_fast(const DbDev& db) {
    // with a direct table (k <= 15), or keyed by MurmurHash3 through the hash table
    const bool front = db.direct != nullptr || (db.addr32 && db.k <= 256);
    return db.format == FMT_SPLIT && (db.binary_tree || db.max_nonleaf_arity <= FAST_MAX_ARITY) && front && !tuning().no_fast;
}
int fast_mode(const DbDev& db) { return db.direct == nullptr ? 2 : (db.direct16 && db.addr32) ? (db.canonical ? 4 : 3) : db.canonical ? 1 : 0; }
bool mode_canonical(int mode) { return mode == 1 || mode == 4; }
// table bits of a wave-per-read class: the narrow class needs fewer on a strand-symmetric index (one lookup per window)
int set_bits_of(const DbDev& db, int c) { return (c == 0 && use_fast(db) && mode_canonical(fast_mode(db))) ? NARROW_CANON_BITS : CLS_SET_BITS[c]; }
uint32_t seq_cap_of(const DbDev& db, int c) { return (2 * (64 * CLS_SLOTS[c] / 2 + db.k) + 15) & ~15u; }
// fast path: L <= 32*SLOTS + k - 1 ascii bytes (+ padding so that the 16-byte packer can over-read)
uint32_t ascii_cap_of(const DbDev& db, int c) {
    const uint32_t fwd = 64 * CLS_SLOTS[c] / 2 + db.k + 16;
    return ((db.direct ? fwd : 2 * fwd) + 15) & ~15u;  // without a direct table: forward ++ reverse complement, hashed as ASCII
}
uint32_t child_ws_stride(const DbDev& db);
constexpr uint32_t CHILD_LDS_MAX = 256;  // per-child counters of polytomies up to this arity live in LDS
bool child_in_lds(const DbDev& db) { return child_ws_stride(db) != 0 && child_ws_stride(db) <= CHILD_LDS_MAX; }
size_t smem_of(const DbDev& db, int c) {
    if (use_fast(db)) {
        const uint32_t ac = ascii_cap_of(db, c);
        const uint32_t packed = db.direct ? 4u * (((ac >> 4) + 2 + 3) & ~3u) : 0u;
        return (size_t)WAVES_PER_BLOCK * (ac + packed + (12u << set_bits_of(db, c)) + (db.binary_tree ? 0u : 12u * FAST_MAX_ARITY + 16u));
    }
    return (size_t)WAVES_PER_BLOCK * (seq_cap_of(db, c) + (4u << CLS_SET_BITS[c]) + 4u * 64 * CLS_SLOTS[c]) +
           (child_in_lds(db) ? (size_t)WAVES_PER_BLOCK * 2 * child_ws_stride(db) * 4 : 0);
}

template <int SLOTS, int SET_BITS>
const void* kernel_of_t(const DbDev& db, bool stats) {
    if (use_fast(db)) {
#define CLS_FAST_OF(A32, MD, PO) (stats ? (const void*)place_fast_kernel<SLOTS, SET_BITS, true, A32, MD, PO> : (const void*)place_fast_kernel<SLOTS, SET_BITS, false, A32, MD, PO>)
#define CLS_FAST_OF2(A32, MD) (db.binary_tree ? CLS_FAST_OF(A32, MD, false) : CLS_FAST_OF(A32, MD, true))
        const int mode = fast_mode(db);
        if (mode == 2) return CLS_FAST_OF2(true, 2);  // (the hashed front is only instantiated with 32-bit offsets: use_fast)
        if (mode == 3) return CLS_FAST_OF2(true, 3);  // (the 16-byte table only exists for k <= 12: 32-bit offsets)
        if (mode == 4) return CLS_FAST_OF2(true, 4);
        if (db.addr32) return mode == 1 ? CLS_FAST_OF2(true, 1) : CLS_FAST_OF2(true, 0);
        return mode == 1 ? CLS_FAST_OF2(false, 1) : CLS_FAST_OF2(false, 0);
#undef CLS_FAST_OF2
#undef CLS_FAST_OF
    }
    if (db.format == FMT_SPLIT) {
        if (db.binary_tree) return stats ? (const void*)place_split_kernel<SLOTS, SET_BITS, true, false> : (const void*)place_split_kernel<SLOTS, SET_BITS, false, false>;
        return stats ? (const void*)place_split_kernel<SLOTS, SET_BITS, true, true> : (const void*)place_split_kernel<SLOTS, SET_BITS, false, true>;
    }
    const bool binary = db.max_nonleaf_arity <= 2;  // no node has more than two non-LEAF children
    if (stats) return binary ? (const void*)place_wave_kernel<SLOTS, SET_BITS, true, true> : (const void*)place_wave_kernel<SLOTS, SET_BITS, true, false>;
    return binary ? (const void*)place_wave_kernel<SLOTS, SET_BITS, false, true> : (const void*)place_wave_kernel<SLOTS, SET_BITS, false, false>;
}
const void* kernel_of(const DbDev& db, int c, bool stats) {
    if (c == 0 && set_bits_of(db, 0) != CLS_SET_BITS[0]) return kernel_of_t<CLS_SLOTS[0], NARROW_CANON_BITS>(db, stats);
    return c == 0 ? kernel_of_t<CLS_SLOTS[0], CLS_SET_BITS[0]>(db, stats) : kernel_of_t<CLS_SLOTS[1], CLS_SET_BITS[1]>(db, stats);
}

uint32_t blk_seq_cap(const DbDev& db) { return (2 * (64 * BLK_WAVES * BLK_SLOTS / 2 + db.k) + 15) & ~15u; }
size_t blk_smem(const DbDev& db);
uint32_t child_ws_stride(const DbDev& db) {
    if (db.format == FMT_SPLIT) return db.binary_tree ? 0u : ((std::max(db.max_nonleaf_arity, 1u) + 63) & ~63u);
    return db.max_nonleaf_arity <= 2 ? 0u : ((db.max_nonleaf_arity + 63) & ~63u);
}
size_t blk_smem(const DbDev& db) {
    return (size_t)blk_seq_cap(db) + (4u << BLK_SET_BITS) + 4u * 64 * BLK_WAVES * BLK_SLOTS + 64 +
           (child_in_lds(db) ? (size_t)2 * child_ws_stride(db) * 4 : 0);
}
}  // namespace

std::string dominant_kernel_name(const DbDev& db, bool stats, const PlacePlan* plan) {
    const std::string sl = std::to_string(CLS_SLOTS[0]) + ", " + std::to_string(set_bits_of(db, 0)) + ", " + (stats ? "true" : "false");
    auto b = [](bool v) { return std::string(v ? "true" : "false"); };
    if (plan && plan->time_tile) return tile_kernel_name(db, stats, plan->tile_name_threads);  // a launch provisioned for long reads (or a bench of gene-length reads): the LDS-tiled kernel is the one that is timed
    if (tuning().time_class == 2) {
        const std::string bl = std::to_string(BLK_WAVES) + ", " + std::to_string(BLK_SLOTS) + ", " + std::to_string(BLK_SET_BITS) + ", " + b(stats);
        if (db.format == FMT_SPLIT) return "place_block_kernel<" + bl + ", " + b(db.binary_tree != 0) + ", true>";
        return "place_block_kernel<" + bl + ", " + b(db.max_nonleaf_arity <= 2) + ", false>";
    }
    if (use_fast(db))
        return "place_fast_kernel<" + sl + ", " + b(fast_mode(db) == 2 || db.addr32) + ", " + std::to_string(fast_mode(db)) + ", " + b(!db.binary_tree) + ">";
    if (db.format == FMT_SPLIT) return "place_split_kernel<" + sl + ", " + b(!db.binary_tree) + ">";
    return "place_wave_kernel<" + sl + ", " + b(db.max_nonleaf_arity <= 2) + ">";
}

PlacePlan plan_place(const DbDev& db, uint32_t n_reads, uint32_t n_cu, bool stats, uint32_t long_cap, uint32_t n_long) {
    PlacePlan p{};
    // persistent-style grids: exactly the blocks that are resident at once (every wave then strides
    // over its class list); CLS_BLOCKS_PER_CU overrides it for tuning experiments
    const int forced = tuning().blocks_per_cu;
    const uint32_t want = (n_reads + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    uint64_t child_words = 0;
    for (int c = 0; c < N_CLASSES; ++c) {
        int per_cu = forced;
        if (per_cu <= 0 &&
            (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel_of(db, c, stats), 64 * WAVES_PER_BLOCK, smem_of(db, c)) != hipSuccess || per_cu <= 0))
            per_cu = 1;
        uint32_t cap = n_cu * (uint32_t)per_cu;
        if (child_ws_stride(db) && !child_in_lds(db)) {  // bound the per-wave child-counter workspace to 128 MiB per class
            const uint64_t per_block = (uint64_t)WAVES_PER_BLOCK * 2 * child_ws_stride(db) * 4;
            const uint64_t fit = std::max<uint64_t>(1, (128ull << 20) / per_block);
            if (cap > fit) cap = (uint32_t)fit;
        }
        p.grid[c] = want < cap ? (want ? want : 1) : cap;
        if (!child_in_lds(db)) child_words = std::max<uint64_t>(child_words, (uint64_t)p.grid[c] * WAVES_PER_BLOCK * 2 * child_ws_stride(db));
    }
    p.grid_blk = std::max<uint32_t>(1, std::min<uint32_t>(n_reads, n_cu));  // 1 workgroup per CU (LDS-bound)
    if (!child_in_lds(db)) child_words = std::max<uint64_t>(child_words, (uint64_t)p.grid_blk * 2 * child_ws_stride(db));
    // workspace (u32 words): [counts 16][list0 n][list1 n][list2 n][list3 n][list4 n][keys_in 2n][keys_out 2n][idx_in n][idx_out n][sort temp][child counters][long-read slices]
    p.ordered = use_order(db, n_reads);
    uint64_t w = 16 + (uint64_t)N_LISTS * n_reads;
    w += w & 1;
    if (p.ordered) {
        // every resident workgroup (5 per CU at 96 VGPRs): since k-mers share split trees and the descent runs
        // on tip-set groups, a read's footprint in the XCD's 4 MiB L2 is small enough that more reads in flight
        // keep paying (C3: 2 per CU 15.5 ms, 3: 11.2, 4: 9.3; a grid beyond residency only adds a tail)
        p.grid[0] = std::max<uint32_t>(8, p.grid[0] & ~7u);  // whole octets of workgroups: one slice of the list per XCD
        p.grid[1] = std::max<uint32_t>(8, p.grid[1] & ~7u);
        {   // the key kernel is bound by the latency of random table reads: every wave the CU can hold
            int per_cu = tuning().key_blocks_per_cu;
            const uint32_t ac = ascii_cap_of(db, 0);
            const size_t smem_k = (size_t)WAVES_PER_BLOCK * (ac + (db.direct ? 4u * (((ac >> 4) + 2 + 3) & ~3u) : 0u) + 16u);
            // (the instance the default knobs launch: forward windows only, at most 64 of them -- two slots, not the placement kernel's five)
            const bool two_slots = !tuning().order_both_strands && tuning().order_windows <= 64;
            const void* kfn = fast_mode(db) == 2 ? (two_slots ? (const void*)order_key_kernel<2, true, true, true> : (const void*)order_key_kernel<CLS_SLOTS[0], true, true, true>)
                              : db.addr32 ? (two_slots ? (const void*)order_key_kernel<2, true, true, false> : (const void*)order_key_kernel<CLS_SLOTS[0], true, true, false>)
                                          : (two_slots ? (const void*)order_key_kernel<2, false, true, false> : (const void*)order_key_kernel<CLS_SLOTS[0], false, true, false>);
            if (per_cu <= 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 64 * WAVES_PER_BLOCK, smem_k) != hipSuccess || per_cu <= 0)) per_cu = 4;
            // (MurmurHash3 front: waves finish their reads at very different times; twice the resident grid evens the tail
            // out: C3s35 keys 2.7 -> 2.1 ms)
            if (tuning().key_blocks_per_cu <= 0 && fast_mode(db) == 2) per_cu *= 2;
            p.grid_key = std::max<uint32_t>(1, std::min<uint32_t>(want, n_cu * (uint32_t)per_cu));
        }
        p.keys_off_words = w;
        w += 6 * (uint64_t)n_reads;
        p.sort_off_words = w;
        p.sort_bytes = order_temp_bytes(n_reads);
        w += (p.sort_bytes + 3) / 4 + 2;
        w += w & 1;
    }
    p.child_off_words = w;
    w += child_words;
    w += w & 1;
    // The longest read the launch is provisioned for: the classes beyond it are not launched (a longer read may be refused).
    p.max_kmers = n_long ? std::max<uint32_t>(long_cap, 2u) : MAX_READ_KMERS;
    // LDS-tiled classes (cls_tile.hip): every read beyond the wave-per-read kernels when the index has the shape for it --
    // the workgroup-per-read kernel then only sees reads of an index that has not (measured on 300-leaf indexes, k = 15 / 35,
    // binary / support-collapsed: 600 bp reads 3.3-4.9 M reads/s there against 23-48 M here, 1.9 kb 2.2-3.5 M against 7.6-17.6 M)
    const uint32_t blk_cap = (uint32_t)(64 * BLK_WAVES * BLK_SLOTS);
    p.tile_from = blk_cap;
    if (tile_usable(db)) {
        p.tile_from = tuning().tile_min_kmers > 0 ? std::min<uint32_t>(blk_cap, (uint32_t)tuning().tile_min_kmers) : (uint32_t)(64 * CLS_SLOTS[1]);
        if (p.max_kmers > p.tile_from) {
            p.tile = tile_plan(db, p.tile_from, p.max_kmers, n_reads, n_cu);
            p.tiled = p.tile.whole.cap_kmers > p.tile_from;
        }
    }
    if (p.tiled) {
        p.time_tile = p.max_kmers > MAX_READ_KMERS || tuning().time_class == 2;
        p.tile_name_threads = p.tile.whole.threads;
        for (uint32_t i = p.tile.n_sub; i-- > 0;) if (p.tile.sub[i].cap_kmers >= p.max_kmers) p.tile_name_threads = p.tile.sub[i].threads;
        p.tile_off_words = w;
        w += p.tile.scratch_words;
        w += w & 1;
    } else p.tile_from = blk_cap;
    // long-read class: per workgroup two state buffers (the distinct-hit set shares the second) + child counters
    if (p.max_kmers > MAX_READ_KMERS || p.tiled) {  // (the LDS-tiled kernel spills to this one)
        long_cap = p.max_kmers;
        p.long_cap = long_cap;
        p.long_set = 1;
        while (p.long_set < 2 * (uint64_t)long_cap) p.long_set <<= 1;
        p.long_arity = (std::max(db.max_nonleaf_arity, 1u) + 63) & ~63u;
        p.long_stride_words = 2 * (uint64_t)LONG_STATE_WORDS * long_cap + 2 * (uint64_t)p.long_arity;
        const uint64_t fit = std::max<uint64_t>(1, (2ull << 30) / (p.long_stride_words * 4));  // at most 2 GiB of slices
        // the kernel is bound by the latency of dependent reads and uses little LDS / few registers: several reads per CU
        const uint64_t per_cu_long = (uint64_t)std::max(1, tuning().long_blocks_per_cu);  // (1: 25.5 k reads/s of 10 kb, 2..8: 31 k)
        p.grid_long = (uint32_t)std::min<uint64_t>({(uint64_t)std::max<uint32_t>(1u, n_long ? n_long : n_reads), (uint64_t)n_cu * per_cu_long, fit});
        if (p.tiled && p.tile.whole.cap_kmers >= long_cap) p.grid_long = std::min<uint32_t>(p.grid_long, 8u);  // only reads the tile kernel spills come here
        p.long_off_words = w;
        w += p.long_stride_words * p.grid_long;
    }
    p.ws_bytes = w * 4;
    return p;
}

hipError_t launch_place(const DbDev& db, const PlaceParams& prm, const PlacePlan& plan, const uint8_t* d_bases,
                        const uint64_t* d_offsets, uint32_t n_reads, cls_placement* d_out, cls_query_stats* d_stats,
                        uint32_t* d_ws, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (n_reads == 0) return hipSuccess;
    uint32_t* counts = d_ws;
    uint32_t* lists[N_LISTS];
    for (int c = 0; c < N_LISTS; ++c) lists[c] = d_ws + 16 + (size_t)c * n_reads;
    uint32_t* child_ws = (child_ws_stride(db) && !child_in_lds(db)) ? d_ws + plan.child_off_words : nullptr;
    hipError_t e = hipMemsetAsync(counts, 0, 64, stream);
    if (e != hipSuccess) return e;
    // CLS_PROFILE_STOP=1|2 truncates the split kernel after the match / state-init phase (timing
    // breakdowns only: the records it then writes are meaningless)
    const uint32_t profile_stop = (uint32_t)tuning().profile_stop;
    const uint32_t ws_stride = child_ws_stride(db);
    const bool st = d_stats != nullptr;
    const bool binary = db.max_nonleaf_arity <= 2;
    auto classify = [&](const uint32_t* order) {
        ClassCaps caps{};
        caps.cap[0] = (uint32_t)(64 * CLS_SLOTS[0]);
        if (plan.max_kmers > caps.cap[0]) caps.cap[1] = (uint32_t)(64 * CLS_SLOTS[1]);
        if (plan.max_kmers > (uint32_t)(64 * CLS_SLOTS[1])) caps.cap[2] = plan.tiled ? plan.tile_from : (uint32_t)(64 * BLK_WAVES * BLK_SLOTS);
        if (plan.tiled) {
            for (uint32_t i = 0; i < plan.tile.n_sub; ++i) caps.cap[3 + i] = plan.tile.sub[i].cap_kmers;
            caps.cap[6] = plan.tile.whole.cap_kmers;
        }
        caps.cap[7] = plan.long_cap;
        hipLaunchKernelGGL(classify_kernel, dim3((n_reads + CLASSIFY_THREADS * CLASSIFY_PER_THREAD - 1) / (CLASSIFY_THREADS * CLASSIFY_PER_THREAD)), dim3(CLASSIFY_THREADS), 0, stream,
                           d_offsets, order, n_reads, db.k, caps, lists[0], counts, d_out, d_stats);
    };
    const uint32_t* list0 = lists[0];
    uint32_t list0_n = 0, xcd_chunks = 0;
    if (!plan.ordered) classify(nullptr);
    if (plan.ordered) {
        uint64_t* keys_in = reinterpret_cast<uint64_t*>(d_ws + plan.keys_off_words);
        uint64_t* keys_out = keys_in + n_reads;
        uint32_t* idx_in = reinterpret_cast<uint32_t*>(keys_out + n_reads);
        uint32_t* idx_out = idx_in + n_reads;
        const uint32_t ac = ascii_cap_of(db, 0);
        const Tuning& tn = tuning();
        const uint32_t key_mode = (uint32_t)tn.order_mode;
        const uint32_t fwd_only = tn.order_both_strands ? 0u : 1u;
        const uint32_t sample_shift = (uint32_t)tn.order_sample_shift;
        const uint32_t block_shift = (uint32_t)tn.order_block_shift;
        const size_t smem_k = (size_t)WAVES_PER_BLOCK * (ac + (db.direct ? 4u * (((ac >> 4) + 2 + 3) & ~3u) : 0u) + 16u);
        // windows of a read that make its key (CLS_ORDER_WINDOWS; 64 = one lookup slot per lane, 160 = all of a 150 bp read)
        const uint32_t key_windows = (uint32_t)tn.order_windows;
        // the reads that get a key: those of the two wave-per-read classes and, when the launch has them, of the LDS-tiled classes
        // (by their first windows: reads of one neighbourhood of the tree then share table lines, set records and split halves in L2)
        const uint32_t key_cap = (plan.tiled && !tn.no_tile_order) ? 0xFFFFFFFFu : (uint32_t)(64 * CLS_SLOTS[1]);
#define CLS_LAUNCH_KEY_S(SL, A32, FW, HS)                                                                                                 \
    hipLaunchKernelGGL((order_key_kernel<SL, A32, FW, HS>), dim3(plan.grid_key), dim3(64 * WAVES_PER_BLOCK), smem_k, stream, db,          \
                       d_bases, d_offsets, n_reads, keys_in, idx_in, ac, key_mode, block_shift, sample_shift, fwd_only,                    \
                       key_cap)
#define CLS_LAUNCH_KEY(A32, HS)                                                                                                           \
    do {                                                                                                                                  \
        if (fwd_only && key_windows <= 64) CLS_LAUNCH_KEY_S(2, A32, true, HS);                                                            \
        else if (fwd_only) CLS_LAUNCH_KEY_S(CLS_SLOTS[0], A32, true, HS);                                                                 \
        else CLS_LAUNCH_KEY_S(CLS_SLOTS[0], A32, false, HS);                                                                              \
    } while (0)
        if (fast_mode(db) != 2 && fwd_only && key_windows == 64 && sample_shift >= 32) {
            // (the default: two reads per wavefront)
            const uint32_t grid_half = std::max<uint32_t>(1, std::min<uint32_t>((n_reads + 2 * WAVES_PER_BLOCK - 1) / (2 * WAVES_PER_BLOCK), plan.grid_key));
            if (db.addr32)
                hipLaunchKernelGGL((order_key_half_kernel<true>), dim3(grid_half), dim3(64 * WAVES_PER_BLOCK), 0, stream, db, d_bases, d_offsets,
                                   n_reads, keys_in, idx_in, key_mode, block_shift, key_cap);
            else
                hipLaunchKernelGGL((order_key_half_kernel<false>), dim3(grid_half), dim3(64 * WAVES_PER_BLOCK), 0, stream, db, d_bases, d_offsets,
                                   n_reads, keys_in, idx_in, key_mode, block_shift, key_cap);
        }
        else if (fast_mode(db) == 2) CLS_LAUNCH_KEY(true, true);
        else if (db.addr32) CLS_LAUNCH_KEY(true, false);
        else CLS_LAUNCH_KEY(false, false);
#undef CLS_LAUNCH_KEY
#undef CLS_LAUNCH_KEY_S
        e = order_reads(d_ws + plan.sort_off_words, keys_in, idx_out, n_reads, order_key_bits(db) - 1, stream);  // (- 1: the bit that only the all-ones "no key" sets)
        if (e != hipSuccess) return e;
        list0 = idx_out;
        list0_n = n_reads;
        xcd_chunks = 1;
        classify(idx_out);  // (after the order: the lists of the LDS-tiled classes come out in locality order)
    }
    auto launch_class = [&](auto slots_c, auto bits_c, int c) {
        constexpr int SLOTS = decltype(slots_c)::value, SET_BITS = decltype(bits_c)::value;
        const dim3 grid(plan.grid[c]), block(64 * WAVES_PER_BLOCK);
        const uint32_t seq_cap = seq_cap_of(db, c);
        const size_t smem = smem_of(db, c);
        if (use_fast(db)) {
            const uint32_t ac = ascii_cap_of(db, c);
            const uint32_t* lst = xcd_chunks ? list0 : lists[c];  // ordered: one list for both classes, each skips the other's reads
            const uint32_t ln = list0_n, xc = xcd_chunks;
#define CLS_LAUNCH_FAST(ST, A32, MD, PO)                                                                                              \
    hipLaunchKernelGGL((place_fast_kernel<SLOTS, SET_BITS, ST, A32, MD, PO>), grid, block, smem, stream, db, prm, d_bases, d_offsets, \
                       lst, counts + c, ln, xc, d_out, d_stats, ac, profile_stop)
#define CLS_LAUNCH_FAST3(ST, A32, MD) do { if (db.binary_tree) CLS_LAUNCH_FAST(ST, A32, MD, false); else CLS_LAUNCH_FAST(ST, A32, MD, true); } while (0)
#define CLS_LAUNCH_FAST2(ST, A32) do { if (db.canonical) CLS_LAUNCH_FAST3(ST, A32, 1); else CLS_LAUNCH_FAST3(ST, A32, 0); } while (0)
            if (fast_mode(db) == 2) { if (st) CLS_LAUNCH_FAST3(true, true, 2); else CLS_LAUNCH_FAST3(false, true, 2); }
            else if (fast_mode(db) == 3 && db.addr32) { if (st) CLS_LAUNCH_FAST3(true, true, 3); else CLS_LAUNCH_FAST3(false, true, 3); }
            else if (fast_mode(db) == 4 && db.addr32) { if (st) CLS_LAUNCH_FAST3(true, true, 4); else CLS_LAUNCH_FAST3(false, true, 4); }
            else if (db.addr32) { if (st) CLS_LAUNCH_FAST2(true, true); else CLS_LAUNCH_FAST2(false, true); }
            else { if (st) CLS_LAUNCH_FAST2(true, false); else CLS_LAUNCH_FAST2(false, false); }
#undef CLS_LAUNCH_FAST2
#undef CLS_LAUNCH_FAST3
#undef CLS_LAUNCH_FAST
            return;
        }
        if (db.format == FMT_SPLIT) {
#define CLS_LAUNCH_SPLIT(ST, PO)                                                                                        \
    hipLaunchKernelGGL((place_split_kernel<SLOTS, SET_BITS, ST, PO>), grid, block, smem, stream, db, prm, d_bases, d_offsets, \
                       lists[c], counts + c, d_out, d_stats, seq_cap, profile_stop, child_ws, ws_stride)
            if (db.binary_tree) { if (st) CLS_LAUNCH_SPLIT(true, false); else CLS_LAUNCH_SPLIT(false, false); }
            else { if (st) CLS_LAUNCH_SPLIT(true, true); else CLS_LAUNCH_SPLIT(false, true); }
#undef CLS_LAUNCH_SPLIT
            return;
        }
#define CLS_LAUNCH(ST, BI)                                                                                          \
    hipLaunchKernelGGL((place_wave_kernel<SLOTS, SET_BITS, ST, BI>), grid, block, smem, stream, db, prm, d_bases,  \
                       d_offsets, lists[c], counts + c, d_out, d_stats, seq_cap, child_ws, ws_stride)
        if (st) { if (binary) CLS_LAUNCH(true, true); else CLS_LAUNCH(true, false); }
        else { if (binary) CLS_LAUNCH(false, true); else CLS_LAUNCH(false, false); }
#undef CLS_LAUNCH
    };
    // the timed kernel (cls_db_kernel_time): the LDS-tiled long-read kernel in a launch provisioned for long reads,
    // else the wave-per-read kernel of the <= 320-k-mer class
    const bool time_tile = plan.time_tile, time_blk = !time_tile && tuning().time_class == 2;  // (time_class 2: a bench of gene-length reads times the workgroup-per-read kernel)
    if (ev_start && !time_tile && !time_blk) (void)hipEventRecord(ev_start, stream);
    if (set_bits_of(db, 0) != CLS_SET_BITS[0]) launch_class(std::integral_constant<int, CLS_SLOTS[0]>{}, std::integral_constant<int, NARROW_CANON_BITS>{}, 0);
    else launch_class(std::integral_constant<int, CLS_SLOTS[0]>{}, std::integral_constant<int, CLS_SET_BITS[0]>{}, 0);
    if (ev_stop && !time_tile && !time_blk) (void)hipEventRecord(ev_stop, stream);
    if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
    if (plan.max_kmers > (uint32_t)(64 * CLS_SLOTS[0])) launch_class(std::integral_constant<int, CLS_SLOTS[1]>{}, std::integral_constant<int, CLS_SET_BITS[1]>{}, 1);
    if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
    if (plan.max_kmers > (uint32_t)(64 * CLS_SLOTS[1]) && (!plan.tiled || plan.tile_from > (uint32_t)(64 * CLS_SLOTS[1]))) {   // class 2: one workgroup per read (the generic probe path, whatever the index format)
        const dim3 grid(plan.grid_blk), block(64 * BLK_WAVES);
        const uint32_t seq_cap = blk_seq_cap(db);
        const size_t smem = blk_smem(db);
        if (ev_start && time_blk) (void)hipEventRecord(ev_start, stream);
#define CLS_LAUNCH_BLK(ST, BI, SP)                                                                                              \
    do {                                                                                                                        \
        auto kfn = place_block_kernel<BLK_WAVES, BLK_SLOTS, BLK_SET_BITS, ST, BI, SP>;                                          \
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); /* > 64 KiB of LDS */ \
        hipLaunchKernelGGL(kfn, grid, block, smem, stream, db, prm, d_bases, d_offsets, lists[2], counts + 2, d_out, d_stats,    \
                           seq_cap, child_ws, ws_stride);                                                                        \
    } while (0)
        if (db.format == FMT_SPLIT && db.binary_tree) { if (st) CLS_LAUNCH_BLK(true, true, true); else CLS_LAUNCH_BLK(false, true, true); }
        else if (db.format == FMT_SPLIT) { if (st) CLS_LAUNCH_BLK(true, false, true); else CLS_LAUNCH_BLK(false, false, true); }
        else if (binary) { if (st) CLS_LAUNCH_BLK(true, true, false); else CLS_LAUNCH_BLK(false, true, false); }
        else { if (st) CLS_LAUNCH_BLK(true, false, false); else CLS_LAUNCH_BLK(false, false, false); }
#undef CLS_LAUNCH_BLK
        if (ev_stop && time_blk) (void)hipEventRecord(ev_stop, stream);
    }
    if (plan.tiled) {  // classes 3 .. 6: every state in LDS; reads the kernel cannot hold are appended to class 7's list
        if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
        if (ev_start && time_tile) (void)hipEventRecord(ev_start, stream);
        const uint32_t* sub_lists[TILE_MAX_SUB] = {lists[3], lists[4], lists[5]};
        const uint32_t* sub_lens[TILE_MAX_SUB] = {counts + 3, counts + 4, counts + 5};
        tile_launch(db, prm, plan.tile, st, d_bases, d_offsets, sub_lists, sub_lens, lists[6], counts + 6, d_out, d_stats, lists[7], counts + 7,
                    d_ws + plan.tile_off_words, plan.ordered && !tuning().no_tile_order, stream);
        if (ev_stop && time_tile) (void)hipEventRecord(ev_stop, stream);
    }
    if (plan.grid_long) {  // class 7: reads beyond what the LDS holds (and every long read of the other index shapes): state in the workspace
        if (hipGetLastError() != hipSuccess) return hipErrorLaunchFailure;
        uint32_t* lws = d_ws + plan.long_off_words;
#define CLS_LAUNCH_LONG(SP, ST)                                                                                                 \
    hipLaunchKernelGGL((place_long_kernel<SP, ST>), dim3(plan.grid_long), dim3(LONG_THREADS), 0, stream, db, prm, d_bases, d_offsets, \
                       lists[7], counts + 7, d_out, d_stats, lws, plan.long_stride_words, plan.long_cap, (uint32_t)plan.long_set,  \
                       plan.long_arity)
        if (db.format == FMT_SPLIT) { if (st) CLS_LAUNCH_LONG(true, true); else CLS_LAUNCH_LONG(true, false); }
        else { if (st) CLS_LAUNCH_LONG(false, true); else CLS_LAUNCH_LONG(false, false); }
#undef CLS_LAUNCH_LONG
    }
    return hipGetLastError();
}

}  // namespace cls
