// Long reads, register-tiled: one WORKGROUP per read, every per-k-mer state in REGISTERS (gfx950 / CDNA4, wave64).
// BASELINE config 5 (10 kb reads, k = 15, deep tree); binary FMT_SPLIT indexes with a direct table.
//
// The algorithm is place_sequence.rs:42-601 as in cls_kernels.hip (A: k-mers + lookup + distinct hashes,
// B: thresholds, C: descent).  What this kernel is built around:
//   * a read's state is one ENTRY per run of consecutive windows that share a tip set: {first tip, last tip, weight,
//     split record} in three registers of the thread that owns the run's first window; the LDS only holds the
//     packed read and the set of codes that makes the k-mers distinct (35 KB: several reads per CU, whose dependent
//     reads overlap -- the LDS-resident predecessor held one read per CU and waited out every round trip);
//   * the front issues ALL of a thread's table lookups at once, then the set records four at a time: three or four
//     round trips per read instead of one per 1024 windows;
//   * a level of the descent needs only the SIGN of |K_a| - |K_b| (both `remove_intersection` values):
//     one signed sum per thread, ONE wave reduction, one LDS atomic per wave, one barrier; the three counts of the
//     record are taken once, at the level the descent ends at;
//   * narrowing an entry to the chosen child is one 8-byte split half for an entry with tips on both sides
//     (kmers_map.rs:189-203 answered from the split tree), all of a thread's reads of a level in flight together.
// A read whose codes overflow a partition of the set (adversarial input only) goes to the workspace kernel through
// the spill list, as before.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>

#include "cls_device.h"
#include "cls_devutil.h"
#include "cls_kernels.h"
#include "cls_tuning.h"

namespace cls {

namespace {

#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4
#endif
constexpr uint32_t RT_SET_ENTRIES = 8192;   // LDS set of codes per pass (32 KB)
constexpr uint32_t RT_TIP_BITS = 24;        // pre-order indices an entry holds (tip << 8 | weight)
constexpr uint32_t RT_DEAD_LO = 0xFFFFFF00u;
constexpr uint32_t SET_EMPTY_RT = 0xFFFFFFFFu;

struct RegSh {
    int32_t cnt[3];          // rotating per-level sums of |K_a| - |K_b|
    uint32_t fin[3];         // the three counts of the final level
    uint32_t n_m, n_root;
    uint32_t overflow;
    uint32_t ib;
    unsigned long long leafp;
};

__host__ __device__ inline uint32_t rt_packed_words(uint32_t max_bases) { return ((max_bases + 15) / 16 + 2 + 3) & ~3u; }
__host__ __device__ inline size_t rt_smem(uint32_t max_bases, uint32_t max_lookups) { return 4ull * rt_packed_words(max_bases) + 4ull * RT_SET_ENTRIES + 4ull * max_lookups; }

// 8-byte half of split record x as ONE read (volatile: or the compiler splits it into two dwords and sinks the second
// into a branch with its own wait).
template <bool ADDR32>
__device__ __forceinline__ uint2 ld_half_v(const uint32_t* half, uint32_t x, uint32_t right) {
    uint64_t v;
    if constexpr (ADDR32) v = *reinterpret_cast<const volatile uint64_t*>(reinterpret_cast<const char*>(half) + (uint32_t)((2 * x + right) * 8u));
    else v = *(reinterpret_cast<const volatile uint64_t*>(half) + (2ull * x + right));
    return uint2{(uint32_t)v, (uint32_t)(v >> 32)};
}

// The thread's share of |K_a| - |K_b| against the split a1n of a clade whose entries are narrowed to it (a1ns = a1n << 8;
// wbm = 0xFF when the clade's second child is scored: a LEAF child is not, place_sequence.rs:322-324).
template <int SLOTS>
__device__ __forceinline__ int32_t count_level(const uint32_t (&LO)[SLOTS], const uint32_t (&HI)[SLOTS], uint32_t a1ns, uint32_t wbm) {
    uint32_t da = 0, db = 0;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const uint32_t w = LO[i] & 0xFFu;
        da += LO[i] < a1ns ? w : 0u;
        db += HI[i] >= a1ns ? w : 0u;
    }
    return (int32_t)da - (int32_t)(wbm ? db : 0u);
}

// One level of the descent for a thread's entries: narrow each to the chosen child (RIGHT: [a1, end), else [a0, a1)).
// Only an entry with a tip OUTSIDE the chosen child is touched -- going left one whose last tip lies at or after a1, going
// right one whose first tip lies before a1 (one compare per entry, and a whole wavefront skips an entry slot none of its
// lanes is concerned in); it dies, or, with tips on both sides, reads the 8-byte half of its split record for the side
// taken: {the tip next to a1 on that side, the split of that part}.  All of a thread's reads of a level are in flight together.
// a0s = a0 << 8 | 0xFF, a1s = a1 << 8 (the entries hold tip << 8 | weight).
template <bool RIGHT, int SLOTS, bool ADDR32, bool STATS>
__device__ __forceinline__ void narrow_level(uint32_t (&LO)[SLOTS], uint32_t (&HI)[SLOTS], uint32_t* __restrict__ xs_t, uint32_t stride, const uint32_t* __restrict__ half,
                                             uint32_t a0s, uint32_t a1s, uint32_t& ib) {
    uint2 t[SLOTS];
    auto concerned = [&](int i) { return RIGHT ? LO[i] < a1s : HI[i] >= a1s; };  // (a dead entry {MAX, 0} never is)
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const bool cnd = concerned(i);
        if (__ballot(cnd) == 0) continue;
        // tips on both sides (going left an entry whose first tip is the first child itself dies whatever lies beyond)
        const bool str = cnd && (RIGHT ? HI[i] >= a1s : (LO[i] < a1s && LO[i] > a0s));
#ifdef RT_EXPERIMENT_NO_READS
        if (false) {
#else
        if (str) {
#endif
            t[i] = ld_half_v<ADDR32>(half, xs_t[(uint32_t)i * stride], RIGHT ? 1u : 0u);
            if (STATS) ib += 8;
        }
    }
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const bool cnd = concerned(i);
        if (__ballot(cnd) == 0) continue;
        if (cnd) {
            uint32_t lo = LO[i], hi = HI[i];
            const bool str = RIGHT ? hi >= a1s : (lo < a1s && lo > a0s);
            bool alive;
            if (RIGHT) {
                if (str) lo = (t[i].x << 8) | (lo & 0xFFu);   // the first tip at or after a1
                alive = str && lo > (a1s | 0xFFu);           // a tip strictly below the second child (lo == a1: the child itself is the tip)
            } else {
                if (str) hi = t[i].x << 8;                     // the last tip before a1
                alive = str;                                   // (a0 < lo < a1)
            }
            if (str) xs_t[(uint32_t)i * stride] = t[i].y;
            LO[i] = alive ? lo : RT_DEAD_LO;
            HI[i] = alive ? hi : 0u;
        }
    }
}

template <int THREADS, int SLOTS, bool CANON, bool STATS, bool ADDR32>
__global__ __launch_bounds__(THREADS, RT_MIN_WAVES) void place_regtile_kernel(DbDev db, PlaceParams prm, const uint8_t* __restrict__ bases,
                                                                const uint64_t* __restrict__ offsets, const uint32_t* __restrict__ list,
                                                                const uint32_t* __restrict__ list_len, cls_placement* __restrict__ out,
                                                                cls_query_stats* __restrict__ stats, uint32_t max_bases, uint32_t pass_codes,
                                                                uint32_t* __restrict__ spill_list, uint32_t* __restrict__ spill_len) {
    extern __shared__ __align__(16) uint8_t smem[];
    __shared__ RegSh sh;
    uint32_t* const packed = reinterpret_cast<uint32_t*>(smem);
    uint32_t* const cset = packed + rt_packed_words(max_bases);
    uint32_t* const xs = cset + RT_SET_ENTRIES;  // per lookup: its tip-set id during the front, then the split record of the entry it heads
    const uint32_t k = db.k;
    const uint32_t kmask = (1u << (2 * k)) - 1u;  // k <= 15
    const uint32_t* __restrict__ direct = db.direct;
    const uint4* __restrict__ sets = reinterpret_cast<const uint4*>(db.sets);
    const uint32_t* __restrict__ half = db.postings;
    const bool rm = prm.remove_intersection != 0;
    const uint32_t n_list = *list_len;
    for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        __syncthreads();  // the previous read's use of the LDS is over
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // (opaque per read: or the 2 * SLOTS window indices are hoisted out of this loop and live, spilled, through the descent)
        const uint32_t lane = tid & 63;
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
        // (classification keeps L >= k and the lookups within the slots; checked all the same: never trust a list)
        if (L64 < k || L64 > max_bases) { put_stats(0, 0, 0, 0, 0); record(L64 < k ? CLS_ERR_TOO_FEW_KMERS : CLS_ERR_READ_TOO_LONG, 0, 0, 0, 0); continue; }
        const uint32_t L = (uint32_t)L64, nf = L - k + 1, nk = 2 * nf;
        const uint32_t n_look = CANON ? nf : nk;
        if (n_look > (uint32_t)(THREADS * SLOTS)) { put_stats(nk, 0, 0, 0, 0); record(CLS_ERR_READ_TOO_LONG, 0, 0, 0, 0); continue; }
        // ---- A1. load, validate (reverse_complement panics on non-ACGT, kmers_map.rs:440), pack 2 bits per base ----
        bool bad = false;
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
        if (tid == 0) { sh.n_m = 0; sh.n_root = 0; sh.overflow = 0; sh.ib = 0; sh.leafp = 0; }
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
        // ---- A2a. the thread's lookups, five at a time: window i * THREADS + tid -> its word of xs = tip-set id | bit 31 when the
        // lookup stands for ONE k-mer (a palindrome, or an index that is not strand-symmetric); 0: not in the index ----
        uint32_t ib = 0;  // per thread; summed at the end
#pragma unroll
        for (int c0 = 0; c0 < SLOTS; c0 += 5) {
            __builtin_amdgcn_sched_barrier(0);  // (five lookups in flight per thread, 2560 per workgroup: enough, and few registers)
            uint32_t ws[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const uint32_t j = (uint32_t)(c0 + q) * THREADS + tid;
                ws[q] = 0;
                if (c0 + q < SLOTS && j < n_look) {
                    bool palindrome;
                    const uint32_t code = code_of(j, palindrome);
                    const uint32_t sid = ldx<uint32_t, ADDR32>(direct, code) & SET_ID_MASK;
                    if (STATS) ib += 4;
                    ws[q] = sid ? (sid | ((CANON && !palindrome) ? 0u : 0x80000000u)) : 0u;
                }
            }
#pragma unroll
            for (int q = 0; q < 5; ++q) if (c0 + q < SLOTS) xs[(uint32_t)(c0 + q) * THREADS + tid] = ws[q];  // (a thread only ever touches its own words of xs)
        }
        // ---- A2a'. distinct k-mers (HashSet<u64> of hashes, kmers_map.rs:273-311): the codes that are in the index go
        // through an LDS set, in PASSES over hash partitions of the codes (a read of any length); a later window with
        // the same code drops out ----
        const uint32_t n_pass = (n_look + pass_codes - 1) / pass_codes;
        for (uint32_t pass = 0; pass < n_pass; ++pass) {
            if (pass) __syncthreads();  // the previous pass' set is no longer probed
            for (uint32_t i = tid; i < RT_SET_ENTRIES; i += THREADS) cset[i] = SET_EMPTY_RT;
            __syncthreads();
#pragma unroll 1
            for (uint32_t j = tid; j < n_look; j += THREADS) {
                if (xs[j] == 0) continue;
                bool palindrome;
                const uint32_t code = code_of(j, palindrome);
                if (n_pass != 1 && (uint32_t)(((uint64_t)mix32(code) * n_pass) >> 32) != pass) continue;
                uint32_t pos = (code * 2654435761u) & (RT_SET_ENTRIES - 1);
#pragma unroll 1
                for (uint32_t probes = 0;; ++probes) {
                    if (probes == RT_SET_ENTRIES) { sh.overflow = 1; xs[j] = 0; break; }  // (a partition that does not fit: spill the read)
                    const uint32_t old = atomicCAS(&cset[pos], SET_EMPTY_RT, code);
                    if (old == SET_EMPTY_RT) break;
                    if (old == code) { xs[j] = 0; break; }
                    pos = (pos + 1) & (RT_SET_ENTRIES - 1);
                }
            }
        }
        __syncthreads();
        if (sh.overflow) {  // hand the read to the workspace kernel
            if (tid == 0) spill_list[atomicAdd(spill_len, 1u)] = r;
            continue;
        }
        // ---- A2b. entries.  Consecutive windows mostly share their tip set (a set's k-mers are the windows between two
        // mutation boundaries of a lineage): runs of equal set ids among a wavefront's 64 consecutive windows become
        // ONE entry weighted by the run, held by the thread of the run's first window; only that thread reads the
        // 16-byte set record (four records in flight per thread).  Entry = {LO = first tip << 8 | weight, HI = last
        // tip << 8} in registers + its split record in the thread's word of xs (only an entry with tips on both sides of a
        // split needs it: one in fifty per level); no entry / dead entry = {RT_DEAD_LO, 0}: below no split, above none. ----
        uint32_t LO[SLOTS], HI[SLOTS];
        uint32_t nm_t = 0, nroot_t = 0;
        uint64_t leafp_t = 0;
#pragma unroll
        for (int c0 = 0; c0 < SLOTS; c0 += 4) {
            __builtin_amdgcn_sched_barrier(0);  // (one chunk's records in flight, not all of them: registers)
            uint4 sr[4];
            uint32_t wq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = c0 + q;
                if (i >= SLOTS) break;
                const uint32_t j = (uint32_t)i * THREADS + tid;
                const uint32_t v = j < n_look ? xs[j] : 0u;
                const uint32_t sid = v & SET_ID_MASK;
                const uint32_t kw = !sid ? 0u : (v >> 31) ? 1u : 2u;
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
            for (int q = 0; q < 4; ++q) {
                const int i = c0 + q;
                if (i >= SLOTS) break;
                const uint32_t w = wq[q];
                const bool has_root = (sr[q].z >> 31) != 0, has_tips = sr[q].y != 0xFFFFFFFFu;
                nm_t += w; nroot_t += has_root ? w : 0u;
                if (STATS) leafp_t += (uint64_t)w * sr[q].w;
                const bool live = w != 0 && has_root && has_tips;
                LO[i] = live ? ((sr[q].y & ((1u << RT_TIP_BITS) - 1)) << 8) | w : RT_DEAD_LO;
                HI[i] = live ? (sr[q].z & ((1u << RT_TIP_BITS) - 1)) << 8 : 0u;
                xs[(uint32_t)i * THREADS + tid] = sr[q].x;
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
        const uint32_t n_m = sh.n_m, n_root = sh.n_root;
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
        // ---- C. descent (place_sequence.rs:279-601) -----------------------------------------------------------------------
        // Per level: d = |K_a| - |K_b| over the workgroup (a LEAF child is not scored, :322-324: its side counts 0) decides;
        // then every entry is narrowed to the chosen child and, in the same breath, counted against that child's split.
        // Both children's node records arrive a level ahead (one 64-byte scalar load per level).
        snode_pair_t C = load_node_pair(db.nodes, P.s[2]);  // (a binary tree: the root has its two children in consecutive rows)
        if (STATS && tid == 0) ib += 64;
        int32_t d = P.s[3] ? count_level<SLOTS>(LO, HI, P.s[6] << 8, P.s[3] >= 2 ? 0xFFu : 0u) : 0;
        int32_t iteration = 0;
        for (;;) {
            ++iteration;
            if (iteration > prm.max_iterations) { record(CLS_ERR_MAX_ITER, 0, 0, (uint32_t)iteration, 0); break; }
            const uint32_t slot = (uint32_t)iteration % 3u;
            {
                const int32_t dw = (int32_t)wave_sum((uint32_t)d);
                if (lane == 0 && dw) atomicAdd(&sh.cnt[slot], dw);
            }
            __syncthreads();
            const int32_t dt = sh.cnt[slot];
            if (tid == 0) sh.cnt[(slot + 2) % 3u] = 0;  // (read by everyone before the barrier just passed; next used two levels on)
            const uint32_t m = P.s[3];
            const uint32_t a0 = P.s[0] + 1, a1 = P.s[6];  // first child = [a0, a1), second = [a1, end of the parent)
            const uint32_t a0s = (a0 << 8) | 0xFFu, a1s = a1 << 8;
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
#pragma unroll
                for (int i = 0; i < SLOTS; ++i) {
                    const uint32_t w = LO[i] & 0xFFu;
                    const bool ina = LO[i] < a1s, inb = HI[i] >= a1s;
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
            C = load_node_pair(db.nodes, Pn.s[2]);  // its children: looked at after the next barrier
            if (STATS && tid == 0) ib += 64;
            // narrow every entry to the chosen clade (one 8-byte split half for an entry with tips on both sides of a1)
            // and count it against the split of that clade's children
            if (right) narrow_level<true, SLOTS, ADDR32, STATS>(LO, HI, xs + tid, THREADS, half, a0s, a1s, ib);
            else narrow_level<false, SLOTS, ADDR32, STATS>(LO, HI, xs + tid, THREADS, half, a0s, a1s, ib);
            d = count_level<SLOTS>(LO, HI, Pn.s[6] << 8, Pn.s[3] >= 2 ? 0xFFu : 0u);
            P = Pn;
        }
        finish_stats();
    }
}

// the (threads, slots) instances: 512 x 20 = 10240 lookups (two or three workgroups per CU), 1024 x 20 = 20480
struct RtShape { uint32_t threads, slots; };
constexpr RtShape RT_SHAPES[] = {{512, 20}, {1024, 20}};

template <int TH, int SL>
const void* rt_kernel_of(bool canon, bool stats, bool a32) {
#define CLS_RT(CN, ST, A) (const void*)place_regtile_kernel<TH, SL, CN, ST, A>
    if (canon) return stats ? (a32 ? CLS_RT(true, true, true) : CLS_RT(true, true, false)) : (a32 ? CLS_RT(true, false, true) : CLS_RT(true, false, false));
    return stats ? (a32 ? CLS_RT(false, true, true) : CLS_RT(false, true, false)) : (a32 ? CLS_RT(false, false, true) : CLS_RT(false, false, false));
#undef CLS_RT
}
const void* rt_kernel(uint32_t threads, bool canon, bool stats, bool a32) {
    return threads == 512 ? rt_kernel_of<512, 20>(canon, stats, a32) : rt_kernel_of<1024, 20>(canon, stats, a32);
}

}  // namespace

bool regtile_usable(const DbDev& db) {
    return db.format == FMT_SPLIT && db.binary_tree && db.direct != nullptr && db.n_nodes < (1u << RT_TIP_BITS) - 1 && !tuning().no_tile && !tuning().tile_v1;
}

RegTilePlan regtile_plan(const DbDev& db, uint32_t want_kmers, uint32_t n_long, uint32_t n_cu, bool stats) {
    RegTilePlan p{};
    const bool canon = db.canonical != 0;
    const uint32_t want = canon ? want_kmers / 2 : want_kmers;  // lookups of the longest read (canonical: one per window)
    RtShape shape = RT_SHAPES[sizeof(RT_SHAPES) / sizeof(RT_SHAPES[0]) - 1];
    for (const RtShape& s : RT_SHAPES) if (s.threads * s.slots >= want) { shape = s; break; }
    p.threads = shape.threads;
    p.slots = shape.slots;
    p.lookups = std::min(want, shape.threads * shape.slots);
    p.bases = (canon ? p.lookups : p.lookups / 2) + db.k;
    p.smem = rt_smem(p.bases, p.threads * p.slots);
    p.cap_kmers = canon ? 2 * p.lookups : p.lookups;
    const void* kfn = rt_kernel(p.threads, canon, stats, db.addr32 != 0);
    (void)hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.smem);
    int per_cu = tuning().tile_blocks_per_cu;
    if (per_cu <= 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, (int)p.threads, p.smem) != hipSuccess || per_cu <= 0)) per_cu = 1;
    p.grid = std::max<uint32_t>(1, std::min<uint32_t>(n_long, n_cu * (uint32_t)per_cu));
    return p;
}

std::string regtile_kernel_name(const DbDev& db, bool stats, uint32_t threads, uint32_t slots) {
    auto b = [](bool v) { return std::string(v ? "true" : "false"); };
    return "place_regtile_kernel<" + std::to_string(threads) + ", " + std::to_string(slots) + ", " + b(db.canonical != 0) + ", " + b(stats) + ", " + b(db.addr32 != 0) + ">";
}

void regtile_launch(const DbDev& db, const PlaceParams& prm, const RegTilePlan& p, bool stats, const uint8_t* d_bases, const uint64_t* d_offsets,
                    const uint32_t* list, const uint32_t* list_len, cls_placement* d_out, cls_query_stats* d_stats, uint32_t* spill_list,
                    uint32_t* spill_len, hipStream_t stream) {
    const void* kfn = rt_kernel(p.threads, db.canonical != 0, stats, db.addr32 != 0);
    (void)hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.smem);
    // (the knob counts lookups per pass of a 4096-entry set; this kernel's set has twice the entries)
    uint32_t max_bases = p.bases, pass_codes = (uint32_t)std::min<long long>(2ll * std::max(1, tuning().tile_pass_codes), 0x7fffffffll);
    void* args[] = {(void*)&db, (void*)&prm, (void*)&d_bases, (void*)&d_offsets, (void*)&list, (void*)&list_len, (void*)&d_out, (void*)&d_stats,
                    (void*)&max_bases, (void*)&pass_codes, (void*)&spill_list, (void*)&spill_len};
    (void)hipLaunchKernel(kfn, dim3(p.grid), dim3(p.threads), args, p.smem, stream);
}

}  // namespace cls
