// C-ABI of libclsplace.so (include/cls_place.h): handle management, upload,
// the host-buffer and device-buffer batch entry points.  Nothing unwinds across
// the boundary; every failure leaves a thread-local message for cls_last_error().
#include <hip/hip_runtime.h>
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "cls_db.h"
#include "cls_device.h"
#include "cls_kernels.h"
#include "cls_place.h"
#include "cls_tuning.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define CLS_HIP(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail(CLS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct Workspace {
    uint32_t* ptr = nullptr;
    uint64_t words = 0;
    hipEvent_t done = nullptr;
    hipStream_t stream = nullptr;           // stream of the slot's last user: the next launch on the SAME stream is ordered
                                            // behind it and may take the slot at once
    uint64_t seq = 0;                       // acquisition counter (the oldest busy slot is the one to wait for)
    int launching = 0;                      // callers between acquire and their `done` record
    bool busy = false;
    bool recorded = false;                  // `done` has been recorded for the current user (until then the event still
                                            // shows the PREVIOUS launch as complete: the slot must not be reclaimed)
};
// HIP events around the dominant kernel of one launch (cls_db_kernel_time); pooled, independent of the scratch slots
struct TimedPair {
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool pending = false;                   // recorded, not yet folded into the handle's accumulators
    bool in_use = false;                    // handed to a launch that has not recorded it yet (set and cleared under ws_mu)
};
constexpr size_t MAX_WS_SLOTS = 8;          // scratch slots per handle; beyond it a caller waits for the oldest launch
constexpr size_t MAX_TIMED_PAIRS = 256;     // launches in flight whose kernel time is still to be harvested

// Stream + device staging buffers of one host-buffer call, recycled across calls (a hipMalloc / hipFree / stream
// create per call cost milliseconds -- and hipFree synchronises the device, stalling every other caller).
struct CallSlot {
    hipStream_t stream = nullptr;
    void *d_bases = nullptr, *d_off = nullptr, *d_out = nullptr, *d_stats = nullptr;
    uint64_t cap_bytes = 0;
    uint32_t cap_reads = 0, cap_stats = 0;
    bool busy = false;
};

cls::PlaceParams resolve(const cls_params* p) {
    // place_sequence.rs:64-75
    cls::PlaceParams r;
    r.max_iterations = (p && (p->flags & CLS_HAS_MAX_ITERATIONS)) ? p->max_iterations : 1000;
    r.remove_intersection = (p && (p->flags & CLS_HAS_REMOVE_INTERSECTION)) ? (p->remove_intersection != 0) : 0;
    double c = 0.7;
    if (p && (p->flags & CLS_HAS_MIN_MATCH_COVERAGE)) {
        c = p->min_match_coverage;
        if (c > 1.0) c = 1.0; else if (c < 0.0) c = 0.0;
    }
    r.min_match_coverage = c;
    return r;
}

}  // namespace

struct cls_db {
    int device = 0;
    int n_cu = 0;
    cls::DbDev dev{};
    cls_db_info info{};
    void* d_nodes = nullptr;
    void* d_kids = nullptr;
    void* d_table = nullptr;
    void* d_postings = nullptr;
    void* d_postings2 = nullptr;
    void* d_bucket_key = nullptr;
    void* d_mz_bucket = nullptr;
    void* d_direct = nullptr;
    void* d_direct16 = nullptr;
    void* d_sets = nullptr;
    void* d_sets2 = nullptr;
    std::mutex ws_mu;
    uint64_t max_read_len = 0;  // what the device-buffer entry provisions its long-read slices for (0: none, reads of up to
                                // MAX_READ_KMERS k-mers only; cls_db_set_max_read_len opts in)
    double kernel_ms_sum = 0.0;
    uint64_t kernel_launches = 0;
    std::vector<Workspace> ws;  // per-call scratch (class lists, child counters), recycled once their launch has finished
    std::vector<TimedPair> timed;
    uint64_t ws_seq = 0;
    std::vector<CallSlot> calls;  // host-buffer calls: stream + staging buffers (ws_mu)
};

// ---- experiment knobs (csrc/cls_tuning.h) ------------------------------------------------------------------
namespace cls {
Tuning& tuning() {
    static Tuning t;
    return t;
}
}  // namespace cls
namespace {
struct Knob { const char* name; int cls::Tuning::*field; };
const Knob KNOBS[] = {
    {"no_fast", &cls::Tuning::no_fast}, {"no_order", &cls::Tuning::no_order}, {"force_list", &cls::Tuning::force_list},
    {"no_mask_halves", &cls::Tuning::no_mask_halves}, {"no_fat_direct", &cls::Tuning::no_fat_direct}, {"no_tile", &cls::Tuning::no_tile}, {"tile_pass_codes", &cls::Tuning::tile_pass_codes}, {"tile_set_words", &cls::Tuning::tile_set_words}, {"time_class", &cls::Tuning::time_class}, {"tile_one_per_cu", &cls::Tuning::tile_one_per_cu}, {"tile_min_kmers", &cls::Tuning::tile_min_kmers}, {"no_tile_order", &cls::Tuning::no_tile_order}, {"tile_deal", &cls::Tuning::tile_deal}, {"blocks_per_cu", &cls::Tuning::blocks_per_cu}, {"key_blocks_per_cu", &cls::Tuning::key_blocks_per_cu},
    {"long_blocks_per_cu", &cls::Tuning::long_blocks_per_cu}, {"order_mode", &cls::Tuning::order_mode},
    {"order_windows", &cls::Tuning::order_windows}, {"order_both_strands", &cls::Tuning::order_both_strands},
    {"order_block_shift", &cls::Tuning::order_block_shift}, {"order_sample_shift", &cls::Tuning::order_sample_shift},
    {"profile_stop", &cls::Tuning::profile_stop}, {"timing", &cls::Tuning::timing},
};
}  // namespace

extern "C" int cls_set_tuning(const char* name, int value) {
    if (!name) return fail(CLS_E_INVALID_ARG, "cls_set_tuning: null name");
    for (const Knob& k : KNOBS)
        if (strcmp(k.name, name) == 0) { cls::tuning().*(k.field) = value; return CLS_OK; }
    return fail(CLS_E_INVALID_ARG, std::string("cls_set_tuning: unknown knob ") + name);
}

extern "C" void cls_tuning_from_env(void) {
    for (const Knob& k : KNOBS) {
        std::string var = "CLS_";
        for (const char* c = k.name; *c; ++c) var += (char)toupper((unsigned char)*c);
        if (const char* v = getenv(var.c_str())) cls::tuning().*(k.field) = *v ? atoi(v) : 1;  // (an empty value means "on")
    }
}

extern "C" const char* cls_last_error(void) { return g_err.c_str(); }
extern "C" void cls_internal_set_error(const char* msg) { g_err = msg ? msg : ""; }  // for the library's other translation units

extern "C" const char* cls_version(void) { return "classeq2_amd 0.1.0 gfx950 abi1"; }

extern "C" int cls_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void cls_db_destroy(cls_db* db) {
    if (!db) return;
    int prev = 0;
    bool have_prev = hipGetDevice(&prev) == hipSuccess;
    (void)hipSetDevice(db->device);
    for (auto& w : db->ws) {
        if (w.done) { if (w.recorded) (void)hipEventSynchronize(w.done); (void)hipEventDestroy(w.done); }
        if (w.ptr) (void)hipFree(w.ptr);
    }
    for (auto& t : db->timed) {
        if (t.t0) (void)hipEventDestroy(t.t0);
        if (t.t1) (void)hipEventDestroy(t.t1);
    }
    for (auto& c : db->calls) {
        if (c.stream) { (void)hipStreamSynchronize(c.stream); (void)hipStreamDestroy(c.stream); }
        for (void* p : {c.d_bases, c.d_off, c.d_out, c.d_stats}) if (p) (void)hipFree(p);
    }
    if (db->d_nodes) (void)hipFree(db->d_nodes);
    if (db->d_kids) (void)hipFree(db->d_kids);
    if (db->d_table) (void)hipFree(db->d_table);
    if (db->d_postings) (void)hipFree(db->d_postings);
    if (db->d_postings2) (void)hipFree(db->d_postings2);
    if (db->d_bucket_key) (void)hipFree(db->d_bucket_key);
    if (db->d_mz_bucket) (void)hipFree(db->d_mz_bucket);
    if (db->d_direct) (void)hipFree(db->d_direct);
    if (db->d_direct16) (void)hipFree(db->d_direct16);
    if (db->d_sets) (void)hipFree(db->d_sets);
    if (db->d_sets2) (void)hipFree(db->d_sets2);
    if (have_prev) (void)hipSetDevice(prev);
    delete db;
}

extern "C" int cls_db_create(const cls_db_desc* d, int device, cls_db** out) {
    if (!out) return fail(CLS_E_INVALID_ARG, "cls_db_create: out is null");
    *out = nullptr;
    try {
        cls::EncodedDb E;
        std::string err;
        int rc = cls::encode_db(d, E, err);
        if (rc != CLS_OK) return fail(rc, "cls_db_create: " + err);
        if (E.format == cls::FMT_LIST && E.postings.size() >= (1ULL << 32)) return fail(CLS_E_BAD_DB, "cls_db_create: sorted-list postings exceed 2^32 words");
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
            return fail(CLS_E_NO_DEVICE, "cls_db_create: no HIP device is visible (the placement path has no CPU fallback)");
        if (device < 0) CLS_HIP(hipGetDevice(&device));
        if (device >= n_dev) return fail(CLS_E_INVALID_ARG, "cls_db_create: device ordinal out of range");
        CLS_HIP(hipSetDevice(device));
        hipDeviceProp_t prop;
        CLS_HIP(hipGetDeviceProperties(&prop, device));
        cls_db* db = new cls_db();
        db->device = device;
        db->n_cu = prop.multiProcessorCount;
        auto up = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
            hipError_t e = hipMalloc(dst, bytes + 64);  // (tail pad: the kernels read node records in pairs and 16-byte entries speculatively)
            if (e != hipSuccess) return e;
            return bytes ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
        };
        hipError_t e;
        if ((e = up(&db->d_nodes, E.nodes.data(), E.nodes.size() * sizeof(cls::DNode))) != hipSuccess ||
            (e = up(&db->d_kids, E.kids.data(), E.kids.size() * 4)) != hipSuccess ||
            (e = up(&db->d_table, E.table.data(), E.table.size() * sizeof(cls::Slot))) != hipSuccess ||
            (e = up(&db->d_postings, E.postings.data(), E.postings.size() * 4)) != hipSuccess ||
            (!E.postings2.empty() && (e = up(&db->d_postings2, E.postings2.data(), E.postings2.size() * 4)) != hipSuccess) ||
            (e = up(&db->d_bucket_key, E.bucket_key.data(), E.bucket_key.size() * 8)) != hipSuccess ||
            (!E.mz_bucket.empty() && (e = up(&db->d_mz_bucket, E.mz_bucket.data(), E.mz_bucket.size() * 4)) != hipSuccess) ||
            (!E.direct.empty() && (e = up(&db->d_direct, E.direct.data(), E.direct.size() * 4)) != hipSuccess) ||
            (!E.direct16.empty() && (e = up(&db->d_direct16, E.direct16.data(), E.direct16.size() * 4)) != hipSuccess) ||
            (!E.sets.empty() && (e = up(&db->d_sets, E.sets.data(), E.sets.size() * sizeof(cls::SetRec))) != hipSuccess) ||
            (!E.sets2.empty() && (e = up(&db->d_sets2, E.sets2.data(), E.sets2.size() * sizeof(cls::SetRec))) != hipSuccess)) {
            cls_db_destroy(db);
            return fail(e == hipErrorOutOfMemory ? CLS_E_NOMEM : CLS_E_HIP, std::string("cls_db_create: upload failed: ") + hipGetErrorString(e));
        }
        cls::DbDev& v = db->dev;
        v.nodes = (const cls::DNode*)db->d_nodes;
        v.kids = (const uint32_t*)db->d_kids;
        v.table = (const cls::Slot*)db->d_table;
        v.postings = (const uint32_t*)db->d_postings;
        v.postings2 = (const uint32_t*)db->d_postings2;
        v.bucket_key = (const uint64_t*)db->d_bucket_key;
        v.mz_bucket = (const uint32_t*)db->d_mz_bucket;
        v.direct = (const uint32_t*)db->d_direct;
        v.direct16 = (const uint32_t*)db->d_direct16;
        v.sets = (const cls::SetRec*)db->d_sets;
        v.sets2 = (const cls::SetRec*)db->d_sets2;
        v.table_mask = E.table.size() - 1;
        v.n_nodes = (uint32_t)E.nodes.size();
        v.n_buckets = (uint32_t)E.bucket_key.size();
        v.k = E.k;
        v.m_eff = E.m_eff;
        v.max_nonleaf_arity = E.max_nonleaf_arity;
        v.format = E.format;
        v.binary_tree = E.strictly_binary ? 1u : 0u;
        v.canonical = E.canonical ? 1u : 0u;
        v.n_sets = (uint32_t)E.sets.size();
        v.set_bits = 1;
        while (v.set_bits < 32 && (1ull << v.set_bits) < (uint64_t)E.sets.size()) ++v.set_bits;
        v.addr32 = (E.postings.size() * 4 < (1ull << 32) && E.direct.size() * 4 < (1ull << 32) && E.sets.size() * sizeof(cls::SetRec) < (1ull << 32)) ? 1u : 0u;
        cls_db_info& i = db->info;
        i.n_nodes = v.n_nodes;
        i.max_depth = E.max_depth;
        i.max_nonleaf_arity = E.max_nonleaf_arity;
        i.k_size = E.k;
        i.m_size = E.m;
        i.n_buckets = (uint32_t)d->n_buckets;
        i.n_kmers = E.n_kmers;
        i.n_closed_kmers = E.n_closed;
        i.table_slots = E.table.size();
        i.postings_words = E.postings.size();
        i.hbm_bytes = E.nodes.size() * sizeof(cls::DNode) + E.table.size() * sizeof(cls::Slot) + (E.postings.size() + E.postings2.size()) * 4 + E.bucket_key.size() * 8 + E.mz_bucket.size() * 4 + E.direct.size() * 4 + E.direct16.size() * 4 + (E.sets.size() + E.sets2.size()) * sizeof(cls::SetRec);
        i.max_read_kmers = db->max_read_len ? (uint32_t)std::max<uint64_t>(320, 2 * db->max_read_len) : cls::MAX_READ_KMERS;
        i.device = device;
        i.format = E.format;
        i.binary_tree = E.strictly_binary ? 1u : 0u;
        i.direct_table = E.direct.empty() ? 0u : (E.canonical ? 2u : 1u);
        i.n_tip_sets = (uint32_t)E.n_sets;
        i.fat_direct_table = E.direct16.empty() ? 0u : 1u;
        *out = db;
        return CLS_OK;
    } catch (const std::bad_alloc&) {
        return fail(CLS_E_NOMEM, "cls_db_create: out of host memory");
    } catch (const std::exception& ex) {
        return fail(CLS_E_INTERNAL, std::string("cls_db_create: ") + ex.what());
    } catch (...) {
        return fail(CLS_E_INTERNAL, "cls_db_create: unknown exception");
    }
}

extern "C" int cls_db_validate(const cls_db_desc* d) {
    try {
        cls::EncodedDb E;
        std::string err;
        int rc = cls::encode_db(d, E, err);
        if (rc != CLS_OK) return fail(rc, "cls_db_validate: " + err);
        return CLS_OK;
    } catch (const std::bad_alloc&) {
        return fail(CLS_E_NOMEM, "cls_db_validate: out of host memory");
    } catch (...) {
        return fail(CLS_E_INTERNAL, "cls_db_validate: unknown exception");
    }
}

extern "C" int cls_db_info_get(const cls_db* db, cls_db_info* info) {
    if (!db || !info) return fail(CLS_E_INVALID_ARG, "cls_db_info_get: null argument");
    cls_db* mdb = const_cast<cls_db*>(db);
    std::lock_guard<std::mutex> g(mdb->ws_mu);
    *info = db->info;
    info->scratch_slots = (uint32_t)db->ws.size();
    return CLS_OK;
}

extern "C" int cls_db_info_get2(const cls_db* db, void* info, size_t info_size) {
    if (!db || !info) return fail(CLS_E_INVALID_ARG, "cls_db_info_get2: null argument");
    cls_db_info full;
    const int rc = cls_db_info_get(db, &full);
    if (rc != CLS_OK) return rc;
    memcpy(info, &full, info_size < sizeof(full) ? info_size : sizeof(full));
    return CLS_OK;
}

// fold finished kernel timings into the handle's accumulators (ws_mu held); `wait`: also those still running
static void harvest(cls_db* db, bool wait) {
    for (auto& t : db->timed) {
        if (!t.pending) continue;
        if (wait ? hipEventSynchronize(t.t1) != hipSuccess : hipEventQuery(t.t1) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.t0, t.t1) == hipSuccess) { db->kernel_ms_sum += ms; db->kernel_launches++; }
        t.pending = false;
    }
}

// An event pair for one launch's dominant kernel (ws_mu held); SIZE_MAX: none free, the launch goes untimed.
static size_t acquire_timed(cls_db* db) {
    harvest(db, false);
    for (size_t i = 0; i < db->timed.size(); ++i)
        if (!db->timed[i].pending && !db->timed[i].in_use && db->timed[i].t0) { db->timed[i].in_use = true; return i; }
    if (db->timed.size() >= MAX_TIMED_PAIRS) return SIZE_MAX;
    TimedPair t;
    if (hipEventCreate(&t.t0) != hipSuccess) return SIZE_MAX;
    if (hipEventCreate(&t.t1) != hipSuccess) { (void)hipEventDestroy(t.t0); return SIZE_MAX; }
    t.in_use = true;
    db->timed.push_back(t);
    return db->timed.size() - 1;
}

// Take a scratch workspace for a launch on `stream`:
//  * a slot whose last user ran on the SAME stream is taken at once (stream order makes the reuse safe), so a caller
//    that pipelines many batches on one stream keeps ONE slot however far ahead of the device it runs;
//  * else a slot whose launch has finished;
//  * else a new one, up to MAX_WS_SLOTS; beyond that the caller waits for the oldest launch in flight.
// Idle slots that are too small are freed before a larger one is allocated.
// `*use` = a copy of the slot taken under the lock: the vector may grow (and move) while the caller launches.
static int acquire_ws(cls_db* db, uint64_t words, hipStream_t stream, size_t* slot, Workspace* use) {
    std::unique_lock<std::mutex> g(db->ws_mu);
    for (;;) {
        size_t oldest = SIZE_MAX;
        for (size_t i = 0; i < db->ws.size(); ++i) {
            Workspace& w = db->ws[i];
            if (w.busy && w.recorded && w.launching == 0 && hipEventQuery(w.done) == hipSuccess) w.busy = false;
            // (hipStreamPerThread is ONE handle value that names a different stream in every host thread: never a "same stream")
            const bool same_stream = w.busy && w.recorded && w.launching == 0 && w.stream == stream && stream != hipStreamPerThread;
            if ((!w.busy || same_stream) && w.words >= words) {
                w.busy = true; w.recorded = false; w.launching = 1; w.stream = stream; w.seq = ++db->ws_seq;
                *slot = i; *use = w;
                return CLS_OK;
            }
            if (w.busy && w.recorded && w.launching == 0 && (oldest == SIZE_MAX || w.seq < db->ws[oldest].seq)) oldest = i;
        }
        // nothing fits: drop idle slots (they are too small), then grow or wait
        for (size_t i = 0; i < db->ws.size();) {
            Workspace& w = db->ws[i];
            if (!w.busy) {
                (void)hipEventDestroy(w.done);
                (void)hipFree(w.ptr);
                db->ws.erase(db->ws.begin() + (ptrdiff_t)i);
                oldest = SIZE_MAX;  // (indices moved: recomputed on the next round if needed)
            } else ++i;
        }
        if (db->ws.size() < MAX_WS_SLOTS) break;
        if (oldest == SIZE_MAX) {
            for (size_t i = 0; i < db->ws.size(); ++i)
                if (db->ws[i].busy && db->ws[i].recorded && db->ws[i].launching == 0 && (oldest == SIZE_MAX || db->ws[i].seq < db->ws[oldest].seq)) oldest = i;
        }
        if (oldest == SIZE_MAX) {  // every slot is between acquire and record on another thread: let them get on
            g.unlock();
            std::this_thread::yield();
            g.lock();
            continue;
        }
        hipEvent_t ev = db->ws[oldest].done;
        g.unlock();
        if (hipEventSynchronize(ev) != hipSuccess) return fail(CLS_E_HIP, "hipEventSynchronize failed while waiting for a scratch slot");
        g.lock();
    }
    Workspace w;
    if (hipMalloc((void**)&w.ptr, words * 4) != hipSuccess) return fail(CLS_E_NOMEM, "scratch workspace allocation failed");
    if (hipEventCreateWithFlags(&w.done, hipEventDisableTiming) != hipSuccess) {
        (void)hipFree(w.ptr);
        return fail(CLS_E_HIP, "hipEventCreate failed");
    }
    w.words = words;
    w.busy = true;
    w.recorded = false;
    w.launching = 1;
    w.stream = stream;
    w.seq = ++db->ws_seq;
    db->ws.push_back(w);
    *slot = db->ws.size() - 1;
    *use = w;
    return CLS_OK;
}

// Longest read (bases) any kernel is provisioned for: 2^25 bases = 2^26 k-mers per read.
static constexpr uint64_t HARD_MAX_READ_LEN = 1ull << 25;

extern "C" int cls_db_set_max_read_len(cls_db* db, uint64_t n_bases) {
    if (!db) return fail(CLS_E_INVALID_ARG, "cls_db_set_max_read_len: null handle");
    if (n_bases > HARD_MAX_READ_LEN) return fail(CLS_E_INVALID_ARG, "cls_db_set_max_read_len: at most 2^25 bases per read");
    std::lock_guard<std::mutex> g(db->ws_mu);
    db->max_read_len = n_bases;
    db->info.max_read_kmers = (uint32_t)std::max<uint64_t>(cls::MAX_READ_KMERS, 2 * n_bases);
    return CLS_OK;
}

// `long_cap` = k-mers per read to provision beyond the register-resident kernels (0: none), `n_long` = how many
// such reads the batch can hold at most.
static int place_device(cls_db* db, const void* d_bases, const void* d_offsets, uint32_t n, const cls_params* params,
                        void* d_out, void* d_stats, hipStream_t stream, uint32_t long_cap, uint32_t n_long) {
    const cls::PlaceParams prm = resolve(params);
    const cls::PlacePlan plan = cls::plan_place(db->dev, n, (uint32_t)db->n_cu, d_stats != nullptr, long_cap, n_long);
    size_t slot = 0;
    Workspace use;
    int rc = acquire_ws(db, (plan.ws_bytes + 3) / 4, stream, &slot, &use);
    if (rc != CLS_OK) return rc;
    size_t tp = SIZE_MAX;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    {
        std::lock_guard<std::mutex> g(db->ws_mu);
        tp = acquire_timed(db);
        if (tp != SIZE_MAX) { t0 = db->timed[tp].t0; t1 = db->timed[tp].t1; }
    }
    hipError_t e = cls::launch_place(db->dev, prm, plan, (const uint8_t*)d_bases, (const uint64_t*)d_offsets, n,
                                     (cls_placement*)d_out, (cls_query_stats*)d_stats, use.ptr, stream, t0, t1);
    bool record_failed = false;
    {
        std::lock_guard<std::mutex> g(db->ws_mu);
        if (tp != SIZE_MAX) { db->timed[tp].pending = (e == hipSuccess); db->timed[tp].in_use = false; }
        // (another thread may have grown the vector meanwhile; slots are only erased while idle, never this one)
        for (size_t i = 0; i < db->ws.size(); ++i)
            if (db->ws[i].ptr == use.ptr) { slot = i; break; }
        record_failed = hipEventRecord(db->ws[slot].done, stream) != hipSuccess;
    }
    if (record_failed) (void)hipStreamSynchronize(stream);  // the kernels may still be running: drain before the slot is handed on
    {
        std::lock_guard<std::mutex> g(db->ws_mu);
        for (size_t i = 0; i < db->ws.size(); ++i)
            if (db->ws[i].ptr == use.ptr) { slot = i; break; }
        db->ws[slot].launching = 0;
        db->ws[slot].recorded = !record_failed;
        if (record_failed) db->ws[slot].busy = false;
    }
    if (e != hipSuccess) return fail(CLS_E_HIP, std::string("kernel launch failed: ") + hipGetErrorString(e));
    return CLS_OK;
}

extern "C" int cls_place_batch_device(cls_db* db, const void* d_bases, const void* d_offsets, uint32_t n,
                                      const cls_params* params, void* d_out, void* d_stats, void* hip_stream) {
    if (!db) return fail(CLS_E_INVALID_ARG, "cls_place_batch_device: null handle");
    if (n == 0) return CLS_OK;
    if (!d_offsets || !d_out) return fail(CLS_E_INVALID_ARG, "cls_place_batch_device: null buffer");
    // the handle's device must be current for the scratch allocation, the events and the launches
    int prev = 0;
    CLS_HIP(hipGetDevice(&prev));
    if (prev != db->device) CLS_HIP(hipSetDevice(db->device));
    int rc;
    try {
        uint64_t max_len;
        { std::lock_guard<std::mutex> g(db->ws_mu); max_len = db->max_read_len; }
        // the read lengths are only known on the device: provision for the handle's limit (0: the register-resident
        // kernels only -- the long-read slices are provisioned when the caller opts in, cls_db_set_max_read_len)
        rc = place_device(db, d_bases, d_offsets, n, params, d_out, d_stats, (hipStream_t)hip_stream, (uint32_t)(2 * max_len), max_len ? n : 0);
    } catch (...) {
        rc = fail(CLS_E_INTERNAL, "cls_place_batch_device: unknown exception");
    }
    if (prev != db->device) (void)hipSetDevice(prev);
    return rc;
}

extern "C" int cls_db_kernel_name(const cls_db* db, char* buf, size_t len) {
    if (!db || !buf || !len) return fail(CLS_E_INVALID_ARG, "cls_db_kernel_name: null argument");
    try {
        uint64_t max_len;
        { std::lock_guard<std::mutex> g(const_cast<cls_db*>(db)->ws_mu); max_len = db->max_read_len; }
        // (as cls_place_batch_device would plan a launch: long reads only when the caller opted in)
        const cls::PlacePlan plan = cls::plan_place(db->dev, 4096, (uint32_t)db->n_cu, false, (uint32_t)(2 * max_len), max_len ? 4096 : 0);
        const std::string s = cls::dominant_kernel_name(db->dev, false, &plan);
        snprintf(buf, len, "%s", s.c_str());
        return CLS_OK;
    } catch (...) {
        return fail(CLS_E_INTERNAL, "cls_db_kernel_name: unknown exception");
    }
}

extern "C" int cls_db_kernel_time(cls_db* db, double* sum_ms, uint64_t* launches, int reset) {
    if (!db) return fail(CLS_E_INVALID_ARG, "cls_db_kernel_time: null handle");
    std::lock_guard<std::mutex> g(db->ws_mu);
    harvest(db, true);
    if (sum_ms) *sum_ms = db->kernel_ms_sum;
    if (launches) *launches = db->kernel_launches;
    if (reset) { db->kernel_ms_sum = 0.0; db->kernel_launches = 0; }
    return CLS_OK;
}

static int place_host(cls_db* db, const char* bases, const uint64_t* offsets, uint32_t n, const cls_params* params,
                      cls_placement* out, cls_query_stats* stats) {
    if (!db) return fail(CLS_E_INVALID_ARG, "cls_place_batch: null handle");
    if (n == 0) return CLS_OK;
    if (!offsets || !out || (!bases && offsets[n] != offsets[0])) return fail(CLS_E_INVALID_ARG, "cls_place_batch: null buffer");
    for (uint32_t i = 0; i < n; ++i)
        if (offsets[i] > offsets[i + 1]) return fail(CLS_E_INVALID_ARG, "cls_place_batch: offsets not monotone");
    int prev = 0;
    CLS_HIP(hipGetDevice(&prev));
    CLS_HIP(hipSetDevice(db->device));
    // Large batches go through in chunks on TWO call slots (stream + grow-only staging buffers from the handle's pool):
    // the copy-in of chunk i+1 and the copy-out of chunk i-1 overlap the kernels of chunk i.
    const uint32_t chunk_reads = 256u << 10;       // reads per chunk (each chunk is still ordered as a whole: >= 4096 reads)
    const uint64_t chunk_bytes = 256ull << 20;     // bases per chunk
    const bool two = n > chunk_reads || offsets[n] - offsets[0] > chunk_bytes;
    size_t ci[2] = {0, 0};
    CallSlot cs[2];
    const int n_slots = two ? 2 : 1;
    {
        std::lock_guard<std::mutex> g(db->ws_mu);
        for (int k = 0; k < n_slots; ++k) {
            size_t i = 0;
            while (i < db->calls.size() && db->calls[i].busy) ++i;
            if (i == db->calls.size()) db->calls.emplace_back();
            db->calls[i].busy = true;
            ci[k] = i;
            cs[k] = db->calls[i];
        }
    }
    auto cleanup = [&]() {  // drain, then hand the slots (with whatever they have grown to) back
        for (int k = 0; k < n_slots; ++k) if (cs[k].stream) (void)hipStreamSynchronize(cs[k].stream);
        std::lock_guard<std::mutex> g(db->ws_mu);
        for (int k = 0; k < n_slots; ++k) { cs[k].busy = false; db->calls[ci[k]] = cs[k]; }
        (void)hipSetDevice(prev);
    };
#define CLS_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) { cleanup(); return fail(e_ == hipErrorOutOfMemory ? CLS_E_NOMEM : CLS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)
    try {
        std::vector<uint64_t> rel[2];
        struct Pending { bool on = false; uint32_t first = 0, cnt = 0; } pend[2];
        auto copy_out = [&](int k) -> hipError_t {  // records (and counters) of the chunk slot k last ran
            if (!pend[k].on) return hipSuccess;
            pend[k].on = false;
            hipError_t e = hipMemcpyAsync(out + pend[k].first, cs[k].d_out, (size_t)pend[k].cnt * sizeof(cls_placement), hipMemcpyDeviceToHost, cs[k].stream);
            if (e == hipSuccess && stats)
                e = hipMemcpyAsync(stats + pend[k].first, cs[k].d_stats, (size_t)pend[k].cnt * sizeof(cls_query_stats), hipMemcpyDeviceToHost, cs[k].stream);
            return e;
        };
        int turn = 0;
        for (uint32_t first = 0; first < n; turn ^= (n_slots - 1)) {
            CallSlot& c = cs[turn];
            uint32_t cnt = 0;
            while (first + cnt < n && cnt < chunk_reads && (cnt == 0 || offsets[first + cnt + 1] - offsets[first] <= chunk_bytes)) ++cnt;
            const uint64_t nbytes = offsets[first + cnt] - offsets[first];
            CLS_TRY(copy_out(turn));  // the slot's previous chunk leaves before its buffers are reused (stream order)
            if (!c.stream) CLS_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
            if (nbytes > c.cap_bytes || !c.d_bases) {
                CLS_TRY(hipStreamSynchronize(c.stream));
                if (c.d_bases) { (void)hipFree(c.d_bases); c.d_bases = nullptr; c.cap_bytes = 0; }
                const uint64_t want = std::max<uint64_t>(nbytes + nbytes / 4, 1 << 16);  // (some slack: jobs of similar size reuse it)
                CLS_TRY(hipMalloc(&c.d_bases, want));
                c.cap_bytes = want;
            }
            if (cnt > c.cap_reads) {
                CLS_TRY(hipStreamSynchronize(c.stream));
                for (void** pp : {&c.d_off, &c.d_out, &c.d_stats}) if (*pp) { (void)hipFree(*pp); *pp = nullptr; }
                c.cap_reads = c.cap_stats = 0;
                const uint32_t want = (uint32_t)std::max<uint64_t>((uint64_t)cnt + cnt / 4, 1024);
                CLS_TRY(hipMalloc(&c.d_off, ((size_t)want + 1) * 8));
                CLS_TRY(hipMalloc(&c.d_out, (size_t)want * sizeof(cls_placement)));
                c.cap_reads = want;
            }
            if (stats && c.cap_stats < c.cap_reads) {
                CLS_TRY(hipStreamSynchronize(c.stream));
                if (c.d_stats) { (void)hipFree(c.d_stats); c.d_stats = nullptr; }
                CLS_TRY(hipMalloc(&c.d_stats, (size_t)c.cap_reads * sizeof(cls_query_stats)));
                c.cap_stats = c.cap_reads;
            }
            std::vector<uint64_t>& ro = rel[turn];
            ro.resize((size_t)cnt + 1);
            for (uint32_t i = 0; i <= cnt; ++i) ro[i] = offsets[first + i] - offsets[first];
            // provision exactly what this chunk needs: the classes beyond its longest read are not launched
            uint64_t longest = 1;
            uint32_t n_long = 1;
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint64_t len = ro[i + 1] - ro[i];
                const uint64_t nk = len < db->dev.k ? 0 : 2 * (len - db->dev.k + 1);
                longest = std::max(longest, std::min(len, HARD_MAX_READ_LEN));
                if (nk > cls::MAX_READ_KMERS) ++n_long;
            }
            if (nbytes) CLS_TRY(hipMemcpyAsync(c.d_bases, bases + offsets[first], nbytes, hipMemcpyHostToDevice, c.stream));
            CLS_TRY(hipMemcpyAsync(c.d_off, ro.data(), ((size_t)cnt + 1) * 8, hipMemcpyHostToDevice, c.stream));
            const int rc = place_device(db, c.d_bases, c.d_off, cnt, params, c.d_out, stats ? c.d_stats : nullptr, c.stream, (uint32_t)(2 * longest), n_long);
            if (rc != CLS_OK) { cleanup(); return rc; }
            pend[turn].on = true;
            pend[turn].first = first;
            pend[turn].cnt = cnt;
            first += cnt;
            // while this chunk computes, bring the other slot's finished records home
            if (n_slots == 2) CLS_TRY(copy_out(turn ^ 1));
        }
        for (int k = 0; k < n_slots; ++k) CLS_TRY(copy_out(k));
        for (int k = 0; k < n_slots; ++k) CLS_TRY(hipStreamSynchronize(cs[k].stream));
    } catch (const std::bad_alloc&) {
        cleanup();
        return fail(CLS_E_NOMEM, "cls_place_batch: out of host memory");
    } catch (...) {
        cleanup();
        return fail(CLS_E_INTERNAL, "cls_place_batch: unknown exception");
    }
    cleanup();
    return CLS_OK;
#undef CLS_TRY
}

extern "C" int cls_place_batch(cls_db* db, const char* bases, const uint64_t* offsets, uint32_t n,
                               const cls_params* params, cls_placement* out) {
    return place_host(db, bases, offsets, n, params, out, nullptr);
}

extern "C" int cls_place_batch_stats(cls_db* db, const char* bases, const uint64_t* offsets, uint32_t n,
                                     const cls_params* params, cls_placement* out, cls_query_stats* stats) {
    if (!stats) return fail(CLS_E_INVALID_ARG, "cls_place_batch_stats: stats is null");
    return place_host(db, bases, offsets, n, params, out, stats);
}

// FASTA text -> records, all on the device: H2D of the file bytes, cls_fasta_scan_device(), placement straight
// from the scanned bases, D2H of the 24-byte records and of the headers (the output stage needs those on the
// host).  `fa->bases` / `fa->base_off` come back NULL: the bases never leave the device.
extern "C" int cls_place_fasta_text(cls_db* db, const char* text, size_t len, const cls_params* params, cls_fasta* fa,
                                    cls_placement** records) {
    if (!db || !fa || !records || (!text && len)) return fail(CLS_E_INVALID_ARG, "cls_place_fasta_text: null argument");
    memset(fa, 0, sizeof *fa);
    *records = nullptr;
    int prev = 0;
    CLS_HIP(hipGetDevice(&prev));
    CLS_HIP(hipSetDevice(db->device));
    hipStream_t stream = nullptr;
    void *d_text = nullptr, *d_out = nullptr;
    cls_fasta_dev dv;
    memset(&dv, 0, sizeof dv);
    cls_placement* recs = nullptr;
    bool ok = false;
    auto cleanup = [&]() {
        if (d_text) (void)hipFree(d_text);
        if (d_out) (void)hipFree(d_out);
        cls_fasta_dev_free(&dv);
        if (stream) (void)hipStreamDestroy(stream);
        (void)hipSetDevice(prev);
        if (!ok) { free(recs); cls_fasta_free(fa); }
    };
#define CLS_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) { cleanup(); return fail(e_ == hipErrorOutOfMemory ? CLS_E_NOMEM : CLS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)
    try {
        CLS_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        CLS_TRY(hipMalloc(&d_text, len ? len : 16));
        if (len) CLS_TRY(hipMemcpyAsync(d_text, text, len, hipMemcpyHostToDevice, stream));
        int rc = cls_fasta_scan_device(d_text, len, &dv, stream);
        if (rc != CLS_OK) { cleanup(); return rc; }
        (void)hipFree(d_text);
        d_text = nullptr;
        const uint32_t n = dv.n;
        fa->n = n;
        fa->truncated = dv.truncated;
        fa->headers = (char*)malloc(dv.n_header_bytes + 1);
        fa->header_off = (uint64_t*)malloc(((size_t)n + 1) * 8);
        recs = (cls_placement*)malloc(((size_t)n + 1) * sizeof(cls_placement));
        std::vector<uint64_t> boff((size_t)n + 1);
        if (!fa->headers || !fa->header_off || !recs) { cleanup(); return fail(CLS_E_NOMEM, "cls_place_fasta_text: out of host memory"); }
        if (dv.n_header_bytes) CLS_TRY(hipMemcpyAsync(fa->headers, dv.d_headers, dv.n_header_bytes, hipMemcpyDeviceToHost, stream));
        CLS_TRY(hipMemcpyAsync(fa->header_off, dv.d_header_off, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, stream));
        CLS_TRY(hipMemcpyAsync(boff.data(), dv.d_base_off, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, stream));
        CLS_TRY(hipStreamSynchronize(stream));
        const uint32_t max_reads = 16u << 20;  // bounds the per-call scratch (class lists, sort keys)
        if (n) CLS_TRY(hipMalloc(&d_out, (size_t)std::min(n, max_reads) * sizeof(cls_placement)));
        for (uint32_t first = 0; first < n; first += max_reads) {
            const uint32_t cnt = std::min(max_reads, n - first);
            uint64_t longest = 1;  // (the classes beyond the chunk's longest read are not launched)
            uint32_t n_long = 1;
            for (uint32_t i = 0; i < cnt; ++i) {
                const uint64_t l = boff[first + i + 1] - boff[first + i];
                const uint64_t nk = l < db->dev.k ? 0 : 2 * (l - db->dev.k + 1);
                longest = std::max(longest, std::min(l, HARD_MAX_READ_LEN));
                if (nk > cls::MAX_READ_KMERS) ++n_long;
            }
            rc = place_device(db, dv.d_bases, (const uint64_t*)dv.d_base_off + first, cnt, params, d_out, nullptr, stream, (uint32_t)(2 * longest), n_long);
            if (rc != CLS_OK) { cleanup(); return rc; }
            CLS_TRY(hipMemcpyAsync(recs + first, d_out, (size_t)cnt * sizeof(cls_placement), hipMemcpyDeviceToHost, stream));
            CLS_TRY(hipStreamSynchronize(stream));
        }
        *records = recs;
        ok = true;
        cleanup();
        return CLS_OK;
    } catch (const std::bad_alloc&) {
        cleanup();
        return fail(CLS_E_NOMEM, "cls_place_fasta_text: out of host memory");
    } catch (...) {
        cleanup();
        return fail(CLS_E_INTERNAL, "cls_place_fasta_text: unknown exception");
    }
#undef CLS_TRY
}
