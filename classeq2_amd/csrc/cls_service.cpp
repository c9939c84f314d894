// Resident batching service (include/cls_service.h): models stay on the device, waiting jobs share device batches.
// Replaces the per-job `load_database` + `place_sequences` of the reference's watcher
// (ports/watcher/src/cmds/watch_dir/mod.rs:300-490) for the part that touches the GPU.
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "cls_service.h"

extern "C" void cls_internal_set_error(const char* msg);

namespace {

int sv_fail(int code, const std::string& m) { cls_internal_set_error(m.c_str()); return code; }

struct Job {
    uint64_t ticket = 0;
    std::string model;
    cls_params params{};
    bool has_params = false;
    cls_fasta fa{};
    cls_placement* records = nullptr;
    int status = CLS_OK;
    std::string error;
    bool done = false;
};

bool same_params(const Job& a, const Job& b) {
    if (a.has_params != b.has_params) return false;
    if (!a.has_params) return true;
    return a.params.flags == b.params.flags && a.params.max_iterations == b.params.max_iterations &&
           memcmp(&a.params.min_match_coverage, &b.params.min_match_coverage, sizeof(double)) == 0 &&
           a.params.remove_intersection == b.params.remove_intersection;
}

}  // namespace

struct cls_service {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::map<std::string, cls_db*> models;
    std::deque<std::shared_ptr<Job>> queue;
    std::map<uint64_t, std::shared_ptr<Job>> jobs;  // submitted and not yet waited for
    uint64_t next_ticket = 1;
    std::string in_flight_model;  // the worker is using this model's handle right now
    bool in_flight = false;
    bool stopping = false;
    bool paused = false;  // the worker leaves the queue alone (cls_service_pause)
    cls_service_stats stats{};
    std::thread worker;

    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return stopping || (!paused && !queue.empty()); });
            if (queue.empty()) { if (stopping) return; continue; }
            // everything that waits for the same model with the same parameters as the oldest job: one device batch
            std::vector<std::shared_ptr<Job>> group;
            const std::shared_ptr<Job> head = queue.front();
            uint64_t reads = 0;
            for (auto it = queue.begin(); it != queue.end();) {
                if ((*it)->model == head->model && same_params(**it, *head) && reads + (*it)->fa.n <= 0xFFFF0000ull) {
                    reads += (*it)->fa.n;
                    group.push_back(*it);
                    it = queue.erase(it);
                } else ++it;
            }
            cls_db* db = nullptr;
            auto m = models.find(head->model);
            if (m != models.end()) db = m->second;
            in_flight_model = head->model;
            in_flight = true;
            lk.unlock();
            int rc = CLS_OK;
            std::string err;
            std::vector<cls_placement> out;
            // nothing unwinds out of the worker: an allocation failure fails the group's jobs, not the process
            try {
            out.resize((size_t)reads);
            if (!db) { rc = CLS_E_INVALID_ARG; err = "unknown model id " + head->model; }
            else if (reads) {
                // one concatenated batch: the bases of the jobs back to back, offsets rebased
                uint64_t total = 0;
                for (auto& j : group) total += j->fa.base_off[j->fa.n];
                std::string bases;
                bases.reserve((size_t)total);
                std::vector<uint64_t> off;
                off.reserve((size_t)reads + 1);
                off.push_back(0);
                for (auto& j : group) {
                    const uint64_t base = bases.size();
                    bases.append(j->fa.bases, (size_t)j->fa.base_off[j->fa.n]);
                    for (uint32_t i = 0; i < j->fa.n; ++i) off.push_back(base + j->fa.base_off[i + 1]);
                }
                rc = cls_place_batch(db, bases.data(), off.data(), (uint32_t)reads, head->has_params ? &head->params : nullptr, out.data());
                if (rc != CLS_OK) err = cls_last_error();
            }
            } catch (const std::bad_alloc&) {
                rc = CLS_E_NOMEM;
                try { err = "out of host memory while batching the jobs"; } catch (...) {}
            } catch (...) {
                rc = CLS_E_INTERNAL;
                try { err = "unexpected exception in the service worker"; } catch (...) {}
            }
            lk.lock();
            in_flight = false;
            size_t pos = 0;
            for (auto& j : group) {
                j->status = rc;
                j->error = err;
                if (rc == CLS_OK) {
                    j->records = (cls_placement*)malloc(((size_t)j->fa.n + 1) * sizeof(cls_placement));
                    if (!j->records) { j->status = CLS_E_NOMEM; j->error = "out of host memory"; }
                    else if (j->fa.n) memcpy(j->records, out.data() + pos, (size_t)j->fa.n * sizeof(cls_placement));
                }
                pos += j->fa.n;
                j->done = true;
            }
            stats.jobs_done += group.size();
            stats.reads_placed += reads;
            if (reads) stats.device_batches += 1;
            stats.max_jobs_in_batch = std::max<uint64_t>(stats.max_jobs_in_batch, group.size());
            cv_done.notify_all();
        }
    }
};

extern "C" int cls_service_create(cls_service** out) {
    if (!out) return sv_fail(CLS_E_INVALID_ARG, "cls_service_create: out is null");
    try {
        auto s = std::make_unique<cls_service>();
        s->worker = std::thread([p = s.get()] { p->run(); });
        *out = s.release();
        return CLS_OK;
    } catch (...) {
        return sv_fail(CLS_E_INTERNAL, "cls_service_create: could not start the worker");
    }
}

extern "C" void cls_service_destroy(cls_service* s) {
    if (!s) return;
    {
        std::lock_guard<std::mutex> g(s->mu);
        s->stopping = true;
        s->paused = false;
    }
    s->cv_work.notify_all();
    if (s->worker.joinable()) s->worker.join();
    for (auto& kv : s->jobs) { cls_fasta_free(&kv.second->fa); free(kv.second->records); }
    for (auto& kv : s->models) cls_db_destroy(kv.second);
    delete s;
}

extern "C" int cls_service_add_model(cls_service* s, const char* model_id, cls_db* db) {
    if (!s || !model_id || !db) return sv_fail(CLS_E_INVALID_ARG, "cls_service_add_model: null argument");
    cls_db* old = nullptr;
    {
        std::lock_guard<std::mutex> g(s->mu);
        for (auto& j : s->queue) if (j->model == model_id) return sv_fail(CLS_E_INVALID_ARG, "cls_service_add_model: jobs are queued for this model id");
        if (s->in_flight && s->in_flight_model == model_id) return sv_fail(CLS_E_INVALID_ARG, "cls_service_add_model: a batch of this model id is running");
        auto it = s->models.find(model_id);
        if (it != s->models.end()) { old = it->second; it->second = db; } else s->models[model_id] = db;
        s->stats.models = s->models.size();
    }
    if (old) cls_db_destroy(old);
    return CLS_OK;
}

extern "C" int cls_service_submit(cls_service* s, const char* model_id, const char* fasta_text, size_t len, const cls_params* params,
                                  uint64_t* ticket) {
    if (!s || !model_id || !ticket || (!fasta_text && len)) return sv_fail(CLS_E_INVALID_ARG, "cls_service_submit: null argument");
    try {
        auto j = std::make_shared<Job>();
        j->model = model_id;
        if (params) { j->params = *params; j->has_params = true; }
        int rc = cls_fasta_parse(fasta_text, len, &j->fa);  // a1, on the submitting thread
        if (rc != CLS_OK) return sv_fail(rc, "cls_service_submit: FASTA stage failed");
        {
            std::lock_guard<std::mutex> g(s->mu);
            if (s->stopping) { cls_fasta_free(&j->fa); return sv_fail(CLS_E_INVALID_ARG, "cls_service_submit: the service is shutting down"); }
            if (!s->models.count(j->model)) { cls_fasta_free(&j->fa); return sv_fail(CLS_E_INVALID_ARG, std::string("cls_service_submit: unknown model id ") + model_id); }
            j->ticket = s->next_ticket++;
            s->jobs[j->ticket] = j;
            s->queue.push_back(j);
            s->stats.jobs_submitted++;
            *ticket = j->ticket;
        }
        s->cv_work.notify_one();
        return CLS_OK;
    } catch (...) {
        return sv_fail(CLS_E_INTERNAL, "cls_service_submit: unknown exception");
    }
}

extern "C" int cls_service_wait(cls_service* s, uint64_t ticket, cls_fasta* fa, cls_placement** records) {
    if (!s || !fa || !records) return sv_fail(CLS_E_INVALID_ARG, "cls_service_wait: null argument");
    memset(fa, 0, sizeof *fa);
    *records = nullptr;
    std::shared_ptr<Job> j;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        auto it = s->jobs.find(ticket);
        if (it == s->jobs.end()) return sv_fail(CLS_E_INVALID_ARG, "cls_service_wait: unknown ticket (or already waited for)");
        j = it->second;
        s->cv_done.wait(lk, [&] { return j->done; });
        s->jobs.erase(ticket);
    }
    if (j->status != CLS_OK) {
        cls_fasta_free(&j->fa);
        free(j->records);
        return sv_fail(j->status, "cls_service_wait: " + j->error);
    }
    *fa = j->fa;
    *records = j->records;
    return CLS_OK;
}

extern "C" int cls_service_pause(cls_service* s, int paused) {
    if (!s) return sv_fail(CLS_E_INVALID_ARG, "cls_service_pause: null handle");
    {
        std::lock_guard<std::mutex> g(s->mu);
        s->paused = paused != 0;
    }
    s->cv_work.notify_all();
    return CLS_OK;
}

extern "C" int cls_service_stats_get(cls_service* s, cls_service_stats* out) {
    if (!s || !out) return sv_fail(CLS_E_INVALID_ARG, "cls_service_stats_get: null argument");
    std::lock_guard<std::mutex> g(s->mu);
    *out = s->stats;
    return CLS_OK;
}
