// Long reads, LDS-tiled: one WORKGROUP per read, every per-k-mer state in LDS (gfx950 / CDNA4, wave64).
// BASELINE config 5 (10 kb reads, k = 15, deep tree); binary FMT_SPLIT indexes with a direct table.
//
// The algorithm is place_sequence.rs:42-601 as in cls_kernels.hip (A: k-mers + lookup + distinct hashes,
// B: thresholds, C: descent).  LDS, first the FRONT's: the read packed 2 bits per base, one word per lookup, the set of codes
// that makes the k-mers distinct; then, over the same bytes, the DESCENT's: one ENTRY per run of consecutive windows that
// share a tip set, {first tip << 8 | weight, last tip << 8} + the set's split record = 12 bytes.  The entries pass from the
// one to the other through the workgroup's slot of a small global scratch (L2-resident), so neither phase pays for the other's
// LDS: a 10 kb read's front fits half a CU, and so do the entries of most reads -- two workgroups per CU; a read with more
// entries than that is handed to a second launch that gives a read the whole LDS (tile_plan below).
// What the descent is built around (on a 150-level ladder tree a level costs what its dependent chain costs):
//   * a level is DECIDED by the sign of |K_a| - |K_b| (both `remove_intersection` values): one pass
//     over the entries, two compares each, ONE packed sum per thread, one wave reduction, one barrier; the three counts
//     of the record are taken once, at the level the descent ends at;
//   * narrowing touches only an entry with a tip OUTSIDE the chosen child: it dies, or -- tips on both sides -- reads
//     ONE 8-byte half of its split record (kmers_map.rs:189-203 answered from the split tree); every other entry costs
//     one LDS read and a handful of compares a level.  At 10 kb the level loop is bound by instruction issue (about 28
//     instructions per entry and level), not by latency: a second workgroup per CU buys 6 %, on 5 kb reads 45 %.
// (Measured and rejected, round 3: the entries in registers, 20 per thread of a 512-thread workgroup, two workgroups per
// CU -- the unrolled per-slot code costs more instructions than the second workgroup hides; halves read asynchronously
// and applied a level later, the level decided early when |K_a| - |K_b| exceeds the weight still in flight -- it hides
// the reads (a build that skips them runs within 10 %) but needs a register per entry slot for the loads, and the
// unrolled loop that comes with it is slower than this rolled one.  DESIGN.md 6.)
// A read whose codes overflow a partition of the set (adversarial input only) goes to the workspace kernel through
// the spill list.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <type_traits>

#include "cls_device.h"
#include "cls_devutil.h"
#include "cls_kernels.h"
#include "cls_tuning.h"

namespace cls {

namespace {

constexpr uint32_t RT_TIP_BITS = 24;        // pre-order indices an entry holds (tip << 8 | weight)
constexpr uint32_t RT_TIP_MASK = (1u << RT_TIP_BITS) - 1;
constexpr uint32_t RT_DEAD_LO = 0xFFFFFF00u;
constexpr uint32_t RT_SET_EMPTY = 0xFFFFFFFFu;
#ifndef RT_LOOK_N
#define RT_LOOK_N 5
#endif
constexpr int RT_LOOK = RT_LOOK_N;          // table lookups / set records a thread keeps in flight

struct RtSh {
    uint32_t cnt[3];         // rotating per-level sums: k-mers with a tip before the split | at or after it << 16
    uint32_t fin[3];         // the three counts of the final level
    uint32_t n_groups;
    uint32_t n_m, n_root;
    uint32_t overflow;
    uint32_t ib;
    unsigned long long leafp;
    uint32_t pu, n_pass;     // polytomy levels: running sums over the levels (a level takes the difference to what it saw before)
    uint32_t best_row, n_best;
    int32_t best_one, best_rest;
};

__host__ __device__ inline uint32_t rt_packed_words(uint32_t max_bases) { return ((max_bases + 15) / 16 + 2 + 3) & ~3u; }
__host__ __device__ inline uint32_t rt_seq_words(uint32_t max_bases) { return ((2 * max_bases + 16 + 15) / 16) * 4; }  // ASCII, both strands, 16 bytes of slack for the 8-byte hash reads
// Dynamic LDS: the FRONT holds the packed read, one word per lookup and the set of codes; the DESCENT, over the same
// bytes, 12 bytes per entry.  The entries travel from the one to the other through the workgroup's slot of a global
// scratch (L2-resident: 12 bytes x max_lookups per resident workgroup), so that neither phase pays for the other's LDS.
__host__ __device__ inline uint32_t rt_scratch_words(uint32_t cap_entries) { return (3u * cap_entries + 1u) & ~1u; }  // a workgroup's slot: 12 bytes per entry, 8-byte aligned
__host__ __device__ inline size_t rt_front_bytes(uint32_t max_lookups, uint32_t max_bases, uint32_t set_words, bool hashed) {
    return hashed ? 4ull * rt_seq_words(max_bases) + 4ull * set_words
                  : 4ull * rt_packed_words(max_bases) + 4ull * max_lookups + 4ull * set_words;
}
// (polytomy trees: behind the entries the children's intervals and two counters per non-LEAF child of the clade at hand)
__host__ __device__ inline uint32_t rt_child_words(uint32_t max_arity) { return 3u * max_arity + 4u; }
__host__ __device__ inline size_t rt_smem(uint32_t max_lookups, uint32_t max_bases, uint32_t set_words, uint32_t cap_entries, bool hashed, uint32_t child_words) {
    const size_t f = rt_front_bytes(max_lookups, max_bases, set_words, hashed), d = 12ull * cap_entries + 4ull * child_words;
    return (f > d ? f : d) + 16;
}

// position of this thread's element in a list that all threads of the workgroup append to (wave-aggregated)
__device__ __forceinline__ uint32_t rt_append(bool keep, uint32_t* counter, uint32_t lane) {
    const uint64_t m = __ballot(keep);
    if (!m) return 0;
    const int leader = __ffsll((unsigned long long)m) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
}

// FRONT: 0 = direct table, one lookup per k-mer; 1 = direct table of a strand-symmetric index, one lookup per window;
// 2 = no direct table (k > 15): MurmurHash3 + hash-table probe per k-mer of both strands (kmers_map.rs:157-159, :273-311)
// POLY: the tree has clades that do not have exactly two children (support-collapsed trees): a level at such a clade walks
// every entry's chain of occupied children (below)
template <int THREADS, int FRONT, bool STATS, bool ADDR32, bool POLY>
__global__ __launch_bounds__(THREADS, 4) void place_tile_kernel(DbDev db, PlaceParams prm, const uint8_t* __restrict__ bases,
                                                             const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ list,
                                                             const uint32_t* __restrict__ list_len, cls_placement* __restrict__ out,
                                                             cls_query_stats* __restrict__ stats, uint32_t max_lookups, uint32_t max_bases,
                                                             uint32_t pass_codes, uint32_t set_words, uint32_t* __restrict__ spill_list, uint32_t* __restrict__ spill_len,
                                                             uint32_t cap_entries, uint32_t* __restrict__ gws, uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_len, uint32_t xcd_walk) {
    constexpr bool CANON = FRONT == 1, HASHED = FRONT == 2;
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ RtSh sh;
    // front: [packed read][one word per lookup][set of codes]; descent, over the same bytes: [entries 8 B][their split records 4 B]
    uint32_t* const packed = reinterpret_cast<uint32_t*>(smem);
    uint8_t* const seq = smem;                                    // HASHED: the read upper-cased, then its reverse complement (2 * max_bases + 16 bytes)
    uint32_t* const wsid = packed + rt_packed_words(max_bases);   // per window its tip-set id | bit 31 (the lookup stands for ONE k-mer) if it is the first with its code, else 0
    uint32_t* const cset = HASHED ? packed + rt_seq_words(max_bases) : wsid + max_lookups;  // the set of codes (HASHED: table slots; no word per lookup) that makes the k-mers distinct
    uint2* const ent = reinterpret_cast<uint2*>(smem);            // {LO = first tip << 8 | weight, HI = last tip << 8}; dead: {RT_DEAD_LO, 0}
    uint32_t* const xs = reinterpret_cast<uint32_t*>(ent + cap_entries);  // split record of the entry's set
    uint32_t* const cpre = xs + cap_entries;                      // POLY: where each non-LEAF child of the clade at hand starts (and, one more, where the last ends)
    uint32_t* const ccnt = cpre + db.max_nonleaf_arity + 2;       //       k-mers with a tip under the child
    uint32_t* const conly = ccnt + db.max_nonleaf_arity + 1;      //       ... and under no other
    // this workgroup's slot of the global scratch: the entries as the front makes them
    uint2* const g_ent = reinterpret_cast<uint2*>(gws + (size_t)blockIdx.x * rt_scratch_words(cap_entries));
    uint32_t* const g_xs = reinterpret_cast<uint32_t*>(g_ent + cap_entries);
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t k = db.k;
    const uint32_t kmask = HASHED ? 0u : (1u << (2 * (k & 15u))) - 1u;  // (direct table: k <= 15)
    const uint32_t* __restrict__ direct = db.direct;
    const uint4* __restrict__ sets = reinterpret_cast<const uint4*>(db.sets);
    const uint32_t* __restrict__ half = db.postings;
    const bool rm = prm.remove_intersection != 0;
    const uint32_t n_list = *list_len;
    // a list in locality order (launch_place): workgroup b runs on XCD b mod 8; the list is dealt to the XCDs in blocks of `xcd_walk`
    // consecutive reads, so that the reads in flight on one XCD -- one L2 -- are neighbours in the order while all eight advance
    // through the list together (an eighth of the list per XCD leaves the XCD with the deepest clades working alone at the end)
    const bool dealt = xcd_walk != 0 && (gridDim.x & 7u) == 0;
    const uint32_t q_stride = dealt ? gridDim.x >> 3 : gridDim.x;
    for (uint32_t q = dealt ? blockIdx.x >> 3 : blockIdx.x;; q += q_stride) {
        const uint32_t li = dealt ? (q / xcd_walk) * (8u * xcd_walk) + (blockIdx.x & 7u) * xcd_walk + q % xcd_walk : q;
        if (li >= n_list) { if (!dealt || (q / xcd_walk) * (8u * xcd_walk) >= n_list) break; else continue; }
        __syncthreads();  // the previous read's use of the LDS is over
        const uint32_t r = list[li];
        const uint64_t b0 = offsets[r], L64 = offsets[r + 1] - b0;
        auto put_stats = [&](uint32_t nk_, uint32_t nm, uint32_t nr, uint64_t lp, uint32_t ibytes) {
            if (STATS && stats && tid == 0) {
                uint64_t* s = reinterpret_cast<uint64_t*>(stats + r);
                s[0] = (uint64_t)nk_ | ((uint64_t)nm << 32);
                s[1] = (uint64_t)nr | ((uint64_t)ibytes << 32);
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
        // (classification keeps L >= k and the lookups within max_lookups; checked all the same: never trust a list)
        if (L64 < k || L64 > max_bases) { put_stats(0, 0, 0, 0, 0); record(L64 < k ? CLS_ERR_TOO_FEW_KMERS : CLS_ERR_READ_TOO_LONG, 0, 0, 0, 0); continue; }
        const uint32_t L = (uint32_t)L64, nf = L - k + 1, nk = 2 * nf;
        const uint32_t n_look = CANON ? nf : nk;
        if (n_look > max_lookups) { put_stats(nk, 0, 0, 0, 0); record(CLS_ERR_READ_TOO_LONG, 0, 0, 0, 0); continue; }
        // ---- A1. load, validate (reverse_complement panics on non-ACGT, kmers_map.rs:440); direct table: pack 2 bits per base;
        // hashed: the upper-cased read followed by its reverse complement, so that query k-mer j (the forward ones first, then
        // those of the reverse complement, kmers_map.rs:387-395) is k contiguous bytes ----
        bool bad = false;
        if constexpr (HASHED) {
            for (uint32_t i = tid; i < L; i += THREADS) {
                uint8_t c = bases[b0 + i];
                if (c >= 'a' && c <= 'z') c -= 32;
                bad |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
                seq[i] = c;
                seq[2 * L - 1 - i] = c ^ ((c & 2) ? 0x04 : 0x15);  // A<->T, C<->G
            }
        } else {
            const uint32_t n_words = (L + 15) >> 4;
            for (uint32_t w = tid; w < n_words + 2; w += THREADS) {
                uint32_t acc = 0;
                if (16 * w < L) {
                    const uint8_t* p = bases + b0 + 16 * (uint64_t)w;
                    const uint32_t nq = L - 16 * w < 16 ? L - 16 * w : 16u;
                    for (uint32_t q = 0; q < nq; ++q) {
                        uint8_t c = p[q];
                        if (c >= 'a' && c <= 'z') c -= 32;
                        bad |= !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
                        acc |= (uint32_t)((c >> 1) & 3u) << (2 * q);  // A0 C1 T2 G3
                    }
                }
                packed[w] = acc;
            }
        }
        if (tid == 0) { sh.n_groups = 0; sh.n_m = 0; sh.n_root = 0; sh.overflow = 0; sh.ib = 0; sh.leafp = 0; sh.pu = 0; sh.n_pass = 0; }
        (void)seq; (void)wsid; (void)cpre; (void)ccnt; (void)conly;
        if constexpr (HASHED) for (uint32_t i = tid; i < set_words; i += THREADS) cset[i] = RT_SET_EMPTY;
        if (tid < 3) { sh.cnt[tid] = 0; sh.fin[tid] = 0; }
        if (__syncthreads_or(bad ? 1 : 0)) { put_stats(0, 0, 0, 0, 0); record(CLS_ERR_INVALID_BASE, 0, 0, 0, 0); continue; }
        // code (and palindrome flag) of lookup j: forward windows first, then those of the reverse complement (kmers_map.rs:387-395)
        auto code_of = [&](uint32_t j, bool& palindrome) -> uint32_t {
            const bool rc = j >= nf;
            const uint32_t p = rc ? (nf - 1) - (j - nf) : j;  // window start; the rc list runs backwards over the windows
            const uint32_t w = p >> 4, s2 = (2 * p) & 31;
            const uint32_t d0 = packed[w], d1 = packed[w + 1];
            uint32_t code = (uint32_t)((((uint64_t)d1 << 32) | d0) >> s2) & kmask;
            uint32_t rcc = __builtin_bitreverse32(code ^ (0xAAAAAAAAu & kmask));
            rcc = ((rcc >> 1) & 0x55555555u) | ((rcc & 0x55555555u) << 1);
            rcc >>= (32 - 2 * k);
            palindrome = code == rcc;
            return CANON ? (rcc < code ? rcc : code) : (rc ? rcc : code);
        };
        // ---- A2a. the table lookups, RT_LOOK per thread in flight: wsid[j] = tip-set id | bit 31 when the lookup stands
        // for ONE k-mer (a palindrome, or an index that is not strand-symmetric); 0: the k-mer is not in the index ----
        uint32_t ib = 0;  // per thread; summed at the end
        if constexpr (!HASHED)
        for (uint32_t j0 = tid; j0 < n_look; j0 += RT_LOOK * THREADS) {
            uint32_t sid[RT_LOOK];
            bool pal[RT_LOOK];
#pragma unroll
            for (int q = 0; q < RT_LOOK; ++q) {
                const uint32_t j = j0 + q * THREADS;
                sid[q] = 0; pal[q] = false;
                if (j < n_look) { sid[q] = ldx<uint32_t, ADDR32>(direct, code_of(j, pal[q])) & SET_ID_MASK; if (STATS) ib += 4; }
            }
#pragma unroll
            for (int q = 0; q < RT_LOOK; ++q) {
                const uint32_t j = j0 + q * THREADS;
                if (j < n_look) wsid[j] = sid[q] ? (sid[q] | ((CANON && !pal[q]) ? 0u : 0x80000000u)) : 0u;
            }
        }
        // ---- A2a'. distinct k-mers (HashSet<u64> of hashes, kmers_map.rs:273-311): the codes that are in the index go
        // through an LDS set (ONE pass: the set has two words per possible lookup; `pass_codes` / `set_words` below their
        // defaults make it passes over hash partitions of the codes, and a partition that does not fit spills the read:
        // tests); a later window with the same code drops out
        // (a thread only ever touches its own words of wsid: no barrier between the lookups and the passes) ----
        const uint32_t n_pass = HASHED ? 0u : (n_look + pass_codes - 1) / pass_codes;  // (hashed: the lookups below go through the set as they hit)
        for (uint32_t pass = 0; pass < n_pass; ++pass) {
            if (pass) __syncthreads();  // the previous pass' set is no longer probed
            for (uint32_t i = tid; i < set_words; i += THREADS) cset[i] = RT_SET_EMPTY;
            __syncthreads();
            for (uint32_t j = tid; j < n_look; j += THREADS) {
                if (wsid[j] == 0) continue;
                bool palindrome;
                const uint32_t code = code_of(j, palindrome);
                if (n_pass != 1 && (uint32_t)(((uint64_t)mix32(code) * n_pass) >> 32) != pass) continue;
                uint32_t pos = (uint32_t)(((uint64_t)(code * 2654435761u) * set_words) >> 32);
                for (uint32_t probes = 0;; ++probes) {
                    if (probes == set_words) { sh.overflow = 1; wsid[j] = 0; break; }  // (a partition that does not fit: spill the read)
                    const uint32_t old = atomicCAS(&cset[pos], RT_SET_EMPTY, code);
                    if (old == RT_SET_EMPTY) break;
                    if (old == code) { wsid[j] = 0; break; }
                    pos = pos + 1 == set_words ? 0u : pos + 1;
                }
            }
        }
        if constexpr (!HASHED) {
            __syncthreads();
            if (sh.overflow) {  // hand the read to the workspace kernel
                if (tid == 0) spill_list[atomicAdd(spill_len, 1u)] = r;
                continue;
            }
        }
        // ---- A2b. entries.  Consecutive windows mostly share their tip set (a set's k-mers are the windows between two
        // mutation boundaries of a lineage): runs of equal set ids among a wavefront's 64 consecutive windows become
        // ONE entry weighted by the run, and only the run's head reads the 16-byte set record (RT_LOOK records in
        // flight per thread).  The entries are appended to the workgroup's slot of the global scratch. ----
        uint32_t nm_t = 0, nroot_t = 0;
        uint64_t leafp_t = 0;
        for (uint32_t jb = 0; jb < n_look; jb += RT_LOOK * THREADS) {
            uint32_t v[RT_LOOK];
            if constexpr (HASHED) {
                // MurmurHash3 of the k-mer, linear probing of the 16-byte slots {hash, set, bucket} (RT_LOOK probes in flight);
                // a hit counts if its bucket's key is the k-mer's own minimizer, the hash of its first m characters
                // (kmers_map.rs:10-13) -- the reference also accepts a bucket keyed by ANOTHER query k-mer's minimizer
                // (kmers_map.rs:295-297): no `cls build-db` output has such an entry, and a read that meets one takes the
                // workspace kernel, which checks it the long way.  HashSet<u64> of hashes (kmers_map.rs:273-311): one table
                // slot per distinct hash of the index, so the first lookup to put its slot into the LDS set keeps the hit.
                const TSlot* __restrict__ table = reinterpret_cast<const TSlot*>(db.table);
                const uint64_t tmask = db.table_mask;
                uint64_t h[RT_LOOK], idx[RT_LOOK];
                TSlot sl[RT_LOOK];
#pragma unroll
                for (int q = 0; q < RT_LOOK; ++q) {
                    const uint32_t j = jb + q * THREADS + tid;
                    h[q] = 0; idx[q] = 0; sl[q] = TSlot{0ull, 0u, 0u};
                    if (j < n_look) { h[q] = murmur3_h1_lds(seq + (j < nf ? j : L + (j - nf)), k); idx[q] = h[q] & tmask; }
                }
#pragma unroll
                for (int q = 0; q < RT_LOOK; ++q) if (jb + q * THREADS + tid < n_look) { sl[q] = table[idx[q]]; if (STATS) ib += 16; }
                const uint32_t* __restrict__ mzb = db.mz_bucket;  // the bucket keyed by the hash of a k-mer's first m characters, tabulated (cls_device.h)
                uint64_t bk[RT_LOOK];
#pragma unroll
                for (int q = 0; q < RT_LOOK; ++q) {
                    while (sl[q].set != 0 && sl[q].hash != h[q]) { idx[q] = (idx[q] + 1) & tmask; sl[q] = table[idx[q]]; if (STATS) ib += 16; }  // (set == 0: an empty slot, the k-mer is not in the index)
                    bk[q] = 0;
                    if (sl[q].set != 0 && !mzb) { bk[q] = db.bucket_key[sl[q].bucket & (uint32_t)LOC_BUCKET_MASK]; if (STATS) ib += 8; }
                }
#pragma unroll
                for (int q = 0; q < RT_LOOK; ++q) {
                    v[q] = 0;
                    if (sl[q].set == 0) continue;
                    const uint32_t j = jb + q * THREADS + tid;
                    const uint8_t* p = seq + (j < nf ? j : L + (j - nf));
                    const bool own = mzb ? mzb[lds_prefix_code(p, db.m_eff)] == (sl[q].bucket & (uint32_t)LOC_BUCKET_MASK) : murmur3_h1_lds(p, db.m_eff) == bk[q];
                    if (STATS && mzb) ib += 4;
                    if (!own) { sh.overflow = 1; continue; }
                    const uint32_t code = (uint32_t)idx[q];
                    uint32_t pos = (uint32_t)(((uint64_t)(code * 2654435761u) * set_words) >> 32);
                    for (uint32_t probes = 0;; ++probes) {
                        if (probes == set_words) { sh.overflow = 1; break; }  // (more distinct k-mers than the set holds: spill the read)
                        const uint32_t old = atomicCAS(&cset[pos], RT_SET_EMPTY, code);
                        if (old == RT_SET_EMPTY) { v[q] = (sl[q].set & SET_ID_MASK) | 0x80000000u; break; }  // (every lookup stands for one k-mer)
                        if (old == code) break;
                        pos = pos + 1 == set_words ? 0u : pos + 1;
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < RT_LOOK; ++q) { const uint32_t j = jb + q * THREADS + tid; v[q] = j < n_look ? wsid[j] : 0u; }
            }
            uint4 sr[RT_LOOK];
            uint32_t wq[RT_LOOK];
#pragma unroll
            for (int q = 0; q < RT_LOOK; ++q) {
                const uint32_t sid = v[q] & SET_ID_MASK;
                const uint32_t kw = !sid ? 0u : (v[q] >> 31) ? 1u : 2u;
                const uint32_t prev_sid = __shfl_up(sid, 1);
                const bool member = sid != 0;
                const bool head = member && (lane == 0 || prev_sid != sid);
                // weight of the run that starts at a head: inclusive prefix sums of kw, run end = lane before the next head / non-member
                uint32_t ps = kw;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(ps, o); if ((int)lane >= o) ps += t; }
                const uint64_t stop = __ballot(head || !member);   // lanes at which a run cannot continue
                const uint64_t later = lane == 63 ? 0ull : (stop >> (lane + 1));
                const uint32_t end = later ? lane + (uint32_t)__ffsll((unsigned long long)later) - 1u : 63u;  // last lane of my run (if I am a head)
                const uint32_t ps_end = __shfl(ps, (int)end);
                wq[q] = head ? ps_end - (ps - kw) : 0u;
                sr[q] = uint4{0u, 0xFFFFFFFFu, 0u, 0u};
                if (head) { sr[q] = ldx<uint4, ADDR32>(sets, sid); if (STATS) ib += 16; }
            }
#pragma unroll
            for (int q = 0; q < RT_LOOK; ++q) {
                const uint32_t w = wq[q];
                const bool has_root = (sr[q].z >> 31) != 0, has_tips = sr[q].y != 0xFFFFFFFFu;
                nm_t += w; nroot_t += has_root ? w : 0u;
                if (STATS) leafp_t += (uint64_t)w * sr[q].w;
                const bool live = w != 0 && has_root && has_tips;
                const uint32_t g = rt_append(live, &sh.n_groups, lane);
                if (live && g < cap_entries) {  // (a read with more entries than the launch holds is handed on below)
                    g_ent[g] = uint2{((sr[q].y & RT_TIP_MASK) << 8) | w, (sr[q].z & RT_TIP_MASK) << 8};
                    g_xs[g] = sr[q].x;
                }
            }
        }
        {   // |M|, |M_root| (and the statistics) over the workgroup
            const uint32_t a = wave_sum(nm_t), b = wave_sum(nroot_t);
            if (lane == 0) { if (a) atomicAdd(&sh.n_m, a); if (b) atomicAdd(&sh.n_root, b); }
            if (STATS) {
                for (int o = 32; o > 0; o >>= 1) leafp_t += ((uint64_t)__shfl_xor((uint32_t)(leafp_t >> 32), o) << 32) | __shfl_xor((uint32_t)leafp_t, o);
                if (lane == 0 && leafp_t) atomicAdd(&sh.leafp, (unsigned long long)leafp_t);
            }
        }
        __syncthreads();
        if (HASHED && sh.overflow) {  // hand the read to the workspace kernel
            if (tid == 0) spill_list[atomicAdd(spill_len, 1u)] = r;
            continue;
        }
        const uint32_t n_m = sh.n_m, n_root = sh.n_root, n_groups = sh.n_groups;
        auto finish_stats = [&]() {
            if constexpr (STATS) {
                const uint32_t w = wave_sum(ib);
                if (lane == 0 && w) atomicAdd(&sh.ib, w);
                __syncthreads();
                put_stats(nk, n_m, n_root, (uint64_t)sh.leafp, sh.ib);
            }
        };
        // ---- B. thresholds (place_sequence.rs:120-139, :156-166, :231-254) ---------------------------------------------
        if (n_m == 0) { finish_stats(); record(CLS_UNCLASSIFIABLE_NO_MATCH, 0, 0, 0, 0); continue; }
        if (n_root == 0) { finish_stats(); record(CLS_UNCLASSIFIABLE_NO_ROOT, 0, 0, 0, 0); continue; }
        snode_t P = load_node(db.nodes, 0);
        if (STATS && tid == 0) ib += 32;
        if (!(P.s[7] & 1u)) { finish_stats(); record(CLS_ERR_ROOT_NO_CHILDREN, 0, 0, 0, 0); continue; }
        {
            const double expected = round((double)n_m * prm.min_match_coverage);
            const uint64_t exp_usize = (expected != expected) ? 0ull : (uint64_t)expected;
            if ((uint64_t)n_root < exp_usize) { finish_stats(); record(CLS_UNCLASSIFIABLE_COVERAGE, (int32_t)n_root, 0, 0, 0); continue; }
        }
        // ---- the entries come back from the scratch into the LDS the front has left (a read with more of them than this
        // configuration holds goes to the launch that gives a read the whole LDS) ----
#ifdef RT_EXPERIMENT_FRONT_ONLY
        { record(CLS_MAX_RESOLUTION, 0, 0, n_groups, 0); continue; }  // (timing experiment: wrong placements)
#endif
        if (n_groups > cap_entries) {
            if (tid == 0) big_list[atomicAdd(big_len, 1u)] = r;
            continue;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();  // (every entry is written, every use of the front's LDS is over)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t j = tid; j < n_groups; j += THREADS) {  // (L2-served loads: the lines may sit in this CU's L1 from the read before)
            const unsigned long long e = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(g_ent + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ent[j] = uint2{(uint32_t)e, (uint32_t)(e >> 32)};
            xs[j] = __hip_atomic_load(g_xs + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        // ---- C. descent (place_sequence.rs:279-601) -----------------------------------------------------------------------
        // Every thread owns the same entries at every level (its slots j = tid, tid + THREADS, ...: `my_n` of them): what it
        // lists it settles and counts itself, no barrier between those.
        // (Measured and rejected, C5: four entries per trip of the level loop with their LDS reads in flight together and one
        // "any tip outside" test, and a thread closing the gaps its dead entries leave every 8 / 32 levels -- 4.63 -> 4.78 ms per
        // 4 000 reads at 0.1 scale, 738 k -> 668 k reads/s at full size: the level's time is its dependent chain (reduce,
        // barrier, decide, node record), not the pass over the entries.)  Both children's node records arrive a level ahead (one 64-byte
        // scalar load per level).
        snode_pair_t C{};
        if (!POLY || (P.s[7] >> 8) == 2) {
            C = load_node_pair(db.nodes, P.s[2]);  // (a clade's non-LEAF children first, all in consecutive rows)
            if (STATS && tid == 0) ib += 64;
        }
        // exact share of this thread's settled entries against split a1n: k-mers with a tip before it | at or after it << 16
        uint32_t my_n = tid < n_groups ? (n_groups - tid + THREADS - 1) / THREADS : 0u;
        auto count = [&](uint32_t a1ns) {
            uint32_t da = 0, db_ = 0;
            for (uint32_t i = 0; i < my_n; ++i) {
                const uint32_t j = tid + i * THREADS;
                const uint2 e = ent[j];
                const uint32_t w = e.x & 0xFFu;  // (0 for a dead and for a pending entry)
                da += e.x < a1ns ? w : 0u;
                db_ += e.y >= a1ns ? w : 0u;
            }
            return da | (db_ << 16);
        };
        uint32_t dd = P.s[3] ? count(P.s[6] << 8) : 0u;
        uint32_t pu_seen = 0, n_pass_seen = 0;  // (POLY: sh.pu, sh.n_pass as of the last polytomy level)
        int32_t iteration = 0;
        for (;;) {
            ++iteration;
            if (iteration > prm.max_iterations) { record(CLS_ERR_MAX_ITER, 0, 0, (uint32_t)iteration, 0); break; }
            const uint32_t m = P.s[3];
            if (POLY && (P.s[7] >> 8) != 2) {
                // ---- a clade that does not have exactly two children (polytomy after support collapse) --------------------
                // The Cartesian tree of a set's tips breaks ties to the left, so the splits between an entry's occupied
                // children form a right-going chain: every entry walks ITS OWN chain (the child under its first tip by binary
                // search over the children's intervals, then one 8-byte read -- the right half of its split record: "the first
                // tip beyond this child, the split of the rest" -- per further occupied child), adding its weight to the
                // per-child counters in LDS.  Nothing is written to the entry: whichever child wins, `enter` below walks
                // the (L2-warm) chain up to it.  (one, rest), place_sequence.rs:369-395, with |R_c| = |U| - |only_c| and
                // |R_c \ K_c| = |U| - |K_c|.
                const DNode* __restrict__ nodes = db.nodes;
                const uint32_t fc = P.s[2];
                uint32_t last_end = 0;
                if (m) { const snode_t Lc = load_node(nodes, fc + m - 1); last_end = Lc.s[0] + Lc.s[1]; }  // the non-LEAF children come first, back to back
                for (uint32_t i = tid; i < m; i += THREADS) { ccnt[i] = 0; conly[i] = 0; cpre[i] = nodes[fc + i].pre; if (STATS) ib += 4; }
                if (tid == 0) { cpre[m] = last_end; sh.n_best = 0; if (STATS && m) ib += 32; }
                __syncthreads();
                if (tid == 0) sh.cnt[((uint32_t)iteration + 2u) % 3u] = 0;  // (the binary levels' rotation goes on through this level)
                uint32_t u_t = 0;
                for (uint32_t i = 0; i < my_n; ++i) {
                    const uint32_t j = tid + i * THREADS;
                    const uint2 e = ent[j];
                    uint32_t v = e.x >> 8, xx = xs[j];
                    const uint32_t vh = e.y >> 8, w = e.x & 0xFFu;
                    uint32_t nin = 0, which = 0;
                    while (v < last_end) {  // v lies under exactly one non-LEAF child: the last one that starts at or before it (a dead entry: v = 2^24 - 1)
                        uint32_t lo_ = 0, hi_ = m;
                        while (hi_ - lo_ > 1) { const uint32_t mid = (lo_ + hi_) >> 1; if (cpre[mid] <= v) lo_ = mid; else hi_ = mid; }
                        const uint32_t c_end = cpre[lo_ + 1];  // (back to back: the next child starts where this one ends)
                        atomicAdd(&ccnt[lo_], w);
                        if (nin == 0) which = lo_;
                        if (nin < 2) ++nin;
                        if (vh < c_end) break;  // no tip beyond this child
                        const uint2 h = ld_half<ADDR32>(half, xx, 1u);
                        if (STATS) ib += 8;
                        v = h.x; xx = h.y;
                    }
                    if (nin == 1) atomicAdd(&conly[which], w);
                    u_t += nin ? w : 0u;
                }
                {
                    const uint32_t a = wave_sum(u_t);
                    if (lane == 0 && a) atomicAdd(&sh.pu, a);
                }
                __syncthreads();
                const uint32_t U = sh.pu - pu_seen;
                pu_seen += U;
                for (uint32_t ci = tid; ci < m; ci += THREADS) {
                    const uint32_t cn = ccnt[ci], on = conly[ci];
                    if (!cn) continue;  // K_c empty: not a candidate (:329)
                    const int32_t one = (int32_t)(rm ? on : cn), rest = (int32_t)(rm ? U - cn : U - on);
                    if (one <= rest) continue;  // :411-417
                    if (atomicAdd(&sh.n_pass, 1u) == n_pass_seen) { sh.best_row = fc + ci; sh.best_one = one; sh.best_rest = rest; sh.n_best = 1; }
                }
                __syncthreads();
                const uint32_t n_pass = sh.n_pass - n_pass_seen;
                n_pass_seen += n_pass;
                if (n_pass > 1) {  // (cannot happen: |K_c| + |only_c| > |U| holds for at most one child; kept for fidelity with :519-599)
                    if (tid == 0) {
                        uint32_t np = 0, nb = 0;
                        int32_t bd = 0;
                        for (uint32_t ci = 0; ci < m; ++ci) {
                            const uint32_t cn = ccnt[ci], on = conly[ci];
                            if (!cn) continue;
                            const int32_t one = (int32_t)(rm ? on : cn), rest = (int32_t)(rm ? U - cn : U - on);
                            if (one <= rest) continue;
                            const int32_t diff = one - rest;
                            if (np == 0 || diff > bd) { bd = diff; nb = 1; sh.best_row = fc + ci; sh.best_one = one; sh.best_rest = rest; }
                            else if (diff == bd) ++nb;
                            ++np;
                        }
                        sh.n_best = nb;
                    }
                    __syncthreads();
                }
                const uint64_t pid = ((uint64_t)P.s[5] << 32) | P.s[4];
                if (n_pass == 0) {
                    if (iteration == 1) record(CLS_UNCLASSIFIABLE_LEVEL1, 0, 0, 1, 0);
                    else record(CLS_MAX_RESOLUTION, 0, 0, (uint32_t)iteration, pid);
                    break;
                }
                if (n_pass > 1 && sh.n_best != 1) { record(CLS_INCONCLUSIVE, (int32_t)n_pass, 0, (uint32_t)iteration, pid); break; }
                const uint32_t best_row = sh.best_row;
                const int32_t best_one = sh.best_one, best_rest = sh.best_rest;
                P = load_node(nodes, best_row);
                if (STATS && tid == 0) ib += 32;
                if (P.s[3] == 0) {
                    record(CLS_IDENTITY_FOUND, best_one, best_rest, (uint32_t)iteration, ((uint64_t)P.s[5] << 32) | P.s[4]);
                    break;
                }
                if ((P.s[7] >> 8) == 2) { C = load_node_pair(nodes, P.s[2]); if (STATS && tid == 0) ib += 64; }
                // enter the chosen child [c0, c_end): step past the occupied children before it, keep the part inside it (an
                // entry whose tip IS the child has nothing below it), and count against the split of the child's children
                const uint32_t c0 = P.s[0], c_end = c0 + P.s[1], a1ns = P.s[6] << 8;
                uint32_t da = 0, db_ = 0;
                for (uint32_t i = 0; i < my_n; ++i) {
                    const uint32_t j = tid + i * THREADS;
                    uint2 e = ent[j];
                    if (e.x == RT_DEAD_LO) continue;
                    uint32_t v = e.x >> 8, vh = e.y >> 8, xx = xs[j];
                    const uint32_t w = e.x & 0xFFu;
                    bool dead = false, changed = false;
                    while (v < c0) {
                        if (vh < c0) { dead = true; break; }
                        const uint2 h = ld_half<ADDR32>(half, xx, 1u);
                        if (STATS) ib += 8;
                        v = h.x; xx = h.y; changed = true;
                    }
                    if (!dead && v < c_end && v != c0) {  // a tip strictly below the chosen clade
                        if (vh >= c_end) {                // ... and tips beyond it: keep the part inside
                            const uint2 h = ld_half<ADDR32>(half, xx, 0u);
                            if (STATS) ib += 8;
                            vh = h.x; xx = h.y; changed = true;
                        }
                    } else dead = true;
                    if (dead) { ent[j] = uint2{RT_DEAD_LO, 0u}; continue; }
                    e = uint2{(v << 8) | w, vh << 8};
                    if (changed) { ent[j] = e; xs[j] = xx; }
                    da += e.x < a1ns ? w : 0u;
                    db_ += e.y >= a1ns ? w : 0u;
                }
                dd = da | (db_ << 16);
                continue;
            }
            const uint32_t a0 = P.s[0] + 1, a1 = P.s[6];  // first child = [a0, a1), second = [a1, end of the parent)
            const uint32_t a0s = (a0 << 8) | 0xFFu, a1s = a1 << 8;
            // |K_a| - |K_b| over the workgroup (a LEAF child is not scored, :322-324)
            const uint32_t slot = (uint32_t)iteration % 3u;
            {
                const uint32_t a = wave_sum(dd);
                if (lane == 0 && a) atomicAdd(&sh.cnt[slot], a);
            }
            __syncthreads();
            const uint32_t t_dd = sh.cnt[slot];
            if (tid == 0) sh.cnt[(slot + 2) % 3u] = 0;  // (read by everyone before the barrier just passed; next used two levels on)
            const int32_t dt = (int32_t)(m ? (t_dd & 0xFFFFu) : 0u) - (int32_t)(m >= 2 ? (t_dd >> 16) : 0u);
            const uint64_t pid = ((uint64_t)P.s[5] << 32) | P.s[4];
            // one_a - rest_a = |only_a| - |only_b| = |K_a| - |K_b| = -(one_b - rest_b) for either remove_intersection:
            // exactly one child passes `one > rest` when the two differ, none on a tie
            if (dt == 0) {
                if (iteration == 1) record(CLS_UNCLASSIFIABLE_LEVEL1, 0, 0, 1, 0);
                else record(CLS_MAX_RESOLUTION, 0, 0, (uint32_t)iteration, pid);
                break;
            }
            const bool right = dt < 0;
            snode_t Pn;
#pragma unroll
            for (int i = 0; i < 8; ++i) Pn.s[i] = right ? C.s[8 + i] : C.s[i];
            if (Pn.s[3] == 0) {  // no non-LEAF child below the chosen clade (update_introspection_node.rs:45-85): the record's counts
                uint32_t ca = 0, cb = 0, bo = 0;
                for (uint32_t i = 0; i < my_n; ++i) {
                    const uint2 e = ent[tid + i * THREADS];
                    const uint32_t w = e.x & 0xFFu;
                    const bool ina = e.x < a1s, inb = e.y >= a1s;
                    ca += ina ? w : 0u; cb += inb ? w : 0u; bo += (ina && inb) ? w : 0u;
                }
                ca = wave_sum(ca); cb = wave_sum(cb); bo = wave_sum(bo);
                if (lane == 0) { if (ca) atomicAdd(&sh.fin[0], ca); if (cb) atomicAdd(&sh.fin[1], cb); if (bo) atomicAdd(&sh.fin[2], bo); }
                __syncthreads();
                uint32_t cnt_a = sh.fin[0], cnt_b = sh.fin[1], both = sh.fin[2];
                if (m < 2) { cnt_b = 0; both = 0; }   // the second child is a LEAF (m >= 1 here: d != 0)
                const uint32_t only_a = cnt_a - both, only_b = cnt_b - both, U = cnt_a + cnt_b - both;
                const uint32_t cn = right ? cnt_b : cnt_a, on = right ? only_b : only_a;
                record(CLS_IDENTITY_FOUND, (int32_t)(rm ? on : cn), (int32_t)(rm ? U - cn : U - on), (uint32_t)iteration, ((uint64_t)Pn.s[5] << 32) | Pn.s[4]);
                break;
            }
            if (!POLY || (Pn.s[7] >> 8) == 2) {
                C = load_node_pair(db.nodes, Pn.s[2]);  // its children: looked at after the next barrier
                if (STATS && tid == 0) ib += 64;
            }
            // narrow the thread's entries to the chosen clade -- only one with a tip OUTSIDE it is touched: going left one
            // whose last tip lies at or after a1, going right one whose first tip lies before a1; it dies or, with tips on
            // both sides, becomes pending on its half -- and count them against the split of that clade's children
            const uint32_t a1ns = Pn.s[6] << 8;
            uint32_t da = 0, db_ = 0;
            auto narrow = [&](auto right_c) {
                constexpr bool RIGHT = decltype(right_c)::value;
                // an entry with a tip outside the child taken (or whose first tip IS that child: it has nothing below it)
                auto outside = [&](const uint2& e) { return RIGHT ? e.x <= (a1s | 0xFFu) : (e.y >= a1s || e.x <= a0s); };
                auto fix = [&](uint2& e, uint32_t j) {  // the rare path: such an entry dies or takes the half of its split record
                    if (!outside(e)) return;
                    const bool str = e.x < a1s && e.y >= a1s && (RIGHT || e.x > a0s);  // tips on both sides
                    if (str) {
#ifndef RT_EXPERIMENT_NO_READS
                        const uint2 h = ld_half<ADDR32>(half, xs[j], RIGHT ? 1u : 0u);
#else
                        const uint2 h = uint2{RIGHT ? (a1 + 1) : (a0 + 1), 0u};  // (timing experiment: wrong placements)
#endif
                        if (STATS) ib += 8;
                        if (RIGHT) { e.x = (h.x << 8) | (e.x & 0xFFu); if (!(e.x > (a1s | 0xFFu))) e = uint2{RT_DEAD_LO, 0u}; }  // the first tip at or after a1; a1 itself: the child is the tip
                        else e.y = h.x << 8;                                                                                         // the last tip before a1
                        xs[j] = h.y;
                    } else e = uint2{RT_DEAD_LO, 0u};
                    ent[j] = e;
                };
                auto tally = [&](const uint2& e) {
                    const uint32_t w = e.x & 0xFFu;
                    da += e.x < a1ns ? w : 0u;
                    db_ += e.y >= a1ns ? w : 0u;
                };
                for (uint32_t i = 0; i < my_n; ++i) {
                    const uint32_t j = tid + i * THREADS;
                    uint2 e = ent[j];
                    fix(e, j);
                    tally(e);
                }
            };
            if (right) narrow(std::true_type{}); else narrow(std::false_type{});
            dd = da | (db_ << 16);
            P = Pn;
        }
        finish_stats();
    }
}

const void* tile_kernel(uint32_t threads, int front, bool stats, bool a32, bool poly) {
#define CLS_RT4(TH, FR, ST, A) (poly ? (const void*)place_tile_kernel<TH, FR, ST, A, true> : (const void*)place_tile_kernel<TH, FR, ST, A, false>)
#define CLS_RT3(TH, FR, ST) (a32 ? CLS_RT4(TH, FR, ST, true) : CLS_RT4(TH, FR, ST, false))
#define CLS_RT2(TH, FR) (stats ? CLS_RT3(TH, FR, true) : CLS_RT3(TH, FR, false))
#define CLS_RT1(TH) (front == 2 ? CLS_RT2(TH, 2) : front == 1 ? CLS_RT2(TH, 1) : CLS_RT2(TH, 0))
    return threads == 128 ? CLS_RT1(128) : threads == 256 ? CLS_RT1(256) : threads == 512 ? CLS_RT1(512) : CLS_RT1(1024);
#undef CLS_RT1
#undef CLS_RT2
#undef CLS_RT3
#undef CLS_RT4
}
int tile_front(const DbDev& db) { return db.direct == nullptr ? 2 : db.canonical ? 1 : 0; }

}  // namespace

// FMT_SPLIT index, pre-order indices in 24 bits; without a direct table (k > 15) the hashed front; clades with up to
// RT_MAX_ARITY non-LEAF children (their counters sit in LDS behind the entries)
constexpr uint32_t RT_MAX_ARITY = 1024;
bool tile_usable(const DbDev& db) {
    return db.format == FMT_SPLIT && (db.binary_tree || db.max_nonleaf_arity <= RT_MAX_ARITY) && db.n_nodes < RT_TIP_MASK && !tuning().no_tile;
}

// Configurations of the one kernel.  WHOLE: a read has the whole LDS of a CU (one 1024-thread workgroup per CU; as many
// lookups as 160 KB hold, the code set in one pass).  SHARED: 8, 4 or 2 workgroups per CU (128, 256, 512 threads: sixteen
// wavefronts a CU either way, their dependent chains overlapping -- 2 per CU: 1.45x per read measured on 5 kb reads; 4 per CU
// on 1.9 kb reads: 1.8x over 2), each for the reads whose front fits its share of the LDS -- what is left of the share after
// the read (and the word per lookup) is the code set -- while the descent still holds an entry for every second lookup; a
// read with more entries than that is handed to the WHOLE launch.
TilePlan tile_plan(const DbDev& db, uint32_t from_kmers, uint32_t max_kmers, uint32_t n_reads, uint32_t n_cu) {
    TilePlan p{};
    const bool hashed = tile_front(db) == 2, canon = !hashed && db.canonical != 0;
    const uint32_t per_look = canon ? 2u : 1u;  // k-mers a lookup stands for (canonical: one per window)
    const uint32_t want = std::min<uint32_t>((max_kmers + per_look - 1) / per_look, 32767u);  // lookups of the longest read; weights are summed in 16-bit halves
    auto bases_of = [&](uint32_t look) { return (canon ? look : look / 2) + db.k; };
    const uint32_t cw = db.binary_tree ? 0u : rt_child_words(db.max_nonleaf_arity);
    const size_t lds_max = 160 * 1024 - 512;  // (the kernel's static LDS is 320 bytes: RtSh and the barrier reductions)
    // WHOLE -- as many lookups per read as 160 KB of LDS hold; longer reads: the workspace kernel.  Direct table: a word per
    // lookup and the code set at two words per lookup in the front, 12 bytes per lookup in the descent.  Hashed: the read's
    // two strands in ASCII and a set of at least 1.25 words per lookup; the descent holds as many entries as fit (a read
    // with more of them -- its k-mers' tip sets hardly ever repeating from one window to the next -- spills).
    auto set_of = [&](uint32_t l) { return hashed ? l + l / 4 : 2 * l; };
    auto cap_of = [&](uint32_t l) { return hashed ? std::min<uint32_t>(l, (uint32_t)((lds_max - 16 - 4ull * cw) / 12)) : l; };
    uint32_t look = want;
    if (rt_smem(look, bases_of(look), set_of(look), cap_of(look), hashed, cw) > lds_max) {
        if (!hashed) look = std::min<uint32_t>(look, (uint32_t)((lds_max - 64 - 4ull * 8 - 4ull * cw) / 12));
        while (look > 64 && rt_smem(look, bases_of(look), set_of(look), cap_of(look), hashed, cw) > lds_max) look -= 64;
    }
    TileCfg& W = p.whole;
    W.threads = 1024u;
    W.lookups = look;
    W.bases = bases_of(look);
    W.cap_entries = cap_of(look);
    W.set_words = hashed ? std::min<uint32_t>(2 * look, (uint32_t)((lds_max - 16 - rt_front_bytes(look, W.bases, 0, true)) / 4)) : 2 * look;
    W.smem = rt_smem(look, W.bases, W.set_words, W.cap_entries, hashed, cw);
    W.grid = std::max<uint32_t>(1, std::min<uint32_t>(n_reads, n_cu));
    W.cap_kmers = per_look * look;
    // SHARED
    uint32_t covered = from_kmers;  // reads of up to this many k-mers have their launch
    for (uint32_t per_cu = tuning().tile_one_per_cu ? 0u : 8u; per_cu >= 2 && covered < W.cap_kmers; per_cu /= 2) {
        const size_t share = (160 * 1024) / per_cu - 1024 - 320;  // (their static LDS and the allocation granule inside 160 KB)
        auto fits = [&](uint32_t l) {
            const uint32_t need_set = hashed ? set_of(l) : std::min<uint32_t>(2 * l, 4096u);
            return rt_front_bytes(l, bases_of(l), need_set, hashed) + 16 <= share && 12ull * (l / 2) + 4ull * cw + 16 <= share;
        };
        uint32_t l = look;
        while (l > 32 && !fits(l)) l -= 32;
        if (!fits(l) || per_look * l <= covered) continue;
        TileCfg& c = p.sub[p.n_sub++];
        c.threads = 1024u / per_cu;
        c.lookups = l;
        c.bases = bases_of(l);
        c.set_words = (uint32_t)std::min<size_t>((share - 16 - rt_front_bytes(l, c.bases, 0, hashed)) / 4, 2ull * l + 64);
        c.cap_entries = std::min<uint32_t>(l, (uint32_t)((share - 16 - 4ull * cw) / 12));
        c.smem = rt_smem(l, c.bases, c.set_words, c.cap_entries, hashed, cw);
        c.grid = std::max<uint32_t>(1, std::min<uint32_t>(n_reads, per_cu * n_cu));
        c.cap_kmers = covered = per_look * l;
    }
    // per resident workgroup 12 bytes per entry (L2-resident)
    for (uint32_t i = 0; i <= p.n_sub; ++i) {
        TileCfg& c = i < p.n_sub ? p.sub[i] : p.whole;
        c.scratch_off = p.scratch_words;
        p.scratch_words += (uint64_t)rt_scratch_words(c.cap_entries) * c.grid;
    }
    return p;
}

std::string tile_kernel_name(const DbDev& db, bool stats, uint32_t threads) {
    auto b = [](bool v) { return std::string(v ? "true" : "false"); };
    return "place_tile_kernel<" + std::to_string(threads) + ", " + std::to_string(tile_front(db)) + ", " + b(stats) + ", " + b(db.addr32 != 0) + ", " + b(!db.binary_tree) + ">";
}

void tile_launch(const DbDev& db, const PlaceParams& prm, const TilePlan& p, bool stats, const uint8_t* d_bases, const uint64_t* d_offsets,
                 const uint32_t* const* sub_lists, const uint32_t* const* sub_lens, uint32_t* big_list, uint32_t* big_len,
                 cls_placement* d_out, cls_query_stats* d_stats, uint32_t* spill_list, uint32_t* spill_len, uint32_t* scratch, bool ordered, hipStream_t stream) {
    // `over`: where a read with more entries than the launch holds goes -- from a shared launch to the WHOLE one, from that
    // (hashed front only) to the workspace kernel
    auto launch = [&](const TileCfg& c, const uint32_t* lst, const uint32_t* len, uint32_t* over_list, uint32_t* over_len) {
        const void* kfn = tile_kernel(c.threads, tile_front(db), stats, db.addr32 != 0, !db.binary_tree);
        (void)hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.smem);
        uint32_t max_lookups = c.lookups, max_bases = c.bases, set_words = c.set_words, cap_entries = c.cap_entries;
        uint32_t* gws = scratch + c.scratch_off;
        uint32_t xcd_walk = ordered ? (uint32_t)std::max(1, tuning().tile_deal) : 0u;
        // the code set: every code in one pass at load <= 0.5 (knobs: fewer words / codes per pass -- tests)
        if (tuning().tile_set_words > 0) set_words = std::min<uint32_t>(set_words, (uint32_t)tuning().tile_set_words);
        uint32_t pass_codes = std::max<uint32_t>(1u, set_words / 2);
        if (tuning().tile_pass_codes > 0) pass_codes = (uint32_t)tuning().tile_pass_codes;
        void* args[] = {(void*)&db, (void*)&prm, (void*)&d_bases, (void*)&d_offsets, (void*)&lst, (void*)&len, (void*)&d_out, (void*)&d_stats,
                        (void*)&max_lookups, (void*)&max_bases, (void*)&pass_codes, (void*)&set_words, (void*)&spill_list, (void*)&spill_len,
                        (void*)&cap_entries, (void*)&gws, (void*)&over_list, (void*)&over_len, (void*)&xcd_walk};
        (void)hipLaunchKernel(kfn, dim3(c.grid), dim3(c.threads), args, c.smem, stream);
    };
    for (uint32_t i = 0; i < p.n_sub; ++i) launch(p.sub[i], sub_lists[i], sub_lens[i], big_list, big_len);
    launch(p.whole, big_list, big_len, spill_list, spill_len);
}

}  // namespace cls
