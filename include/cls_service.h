/* Resident batching service around the placement path.
 *
 * The reference's watcher (ports/watcher/src/cmds/watch_dir/mod.rs:156-490) walks the work directories one
 * after the other and, per job, loads the model from disk (:355) and runs place_sequences on the job's query
 * file.  This is the part of it that belongs next to the GPU: models stay resident on the device across jobs,
 * and the queries of many small jobs that wait at the same time are placed by ONE device batch.  Scanning
 * directories, config / status files and logging stay in the caller (control plane).
 *
 * Threading: every function may be called from any thread.  cls_service_submit() parses the job's FASTA text on
 * the calling thread (a1 semantics, cls_fasta_parse) and returns at once; one worker thread per service drains
 * the queue, groups the waiting jobs by (model, parameters) and issues one cls_place_batch() per group.
 */
#ifndef CLS_SERVICE_H
#define CLS_SERVICE_H

#include "cls_place.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cls_service cls_service;

typedef struct cls_service_stats {
    uint64_t jobs_submitted;
    uint64_t jobs_done;
    uint64_t reads_placed;
    uint64_t device_batches;   /* cls_place_batch() calls issued: <= jobs_done, far fewer under load */
    uint64_t max_jobs_in_batch;
    uint64_t models;
} cls_service_stats;

int cls_service_create(cls_service** out);
/* Waits for the queued jobs, stops the worker, destroys the registered handles. */
void cls_service_destroy(cls_service* s);

/* Make a model resident under `model_id` (the watcher's ModelsConfig id).  The service takes ownership of `db`
 * (cls_db_destroy at cls_service_destroy / on replacement by a later call with the same id). */
int cls_service_add_model(cls_service* s, const char* model_id, cls_db* db);

/* Queue one job: the text of its query file and its Option<> parameters (NULL = all None).  -> ticket */
int cls_service_submit(cls_service* s, const char* model_id, const char* fasta_text, size_t len, const cls_params* params,
                       uint64_t* ticket);

/* Block until the job is done; hands over its FASTA records (headers, bases, offsets; cls_fasta_free) and one
 * cls_placement per record (`*records`, free()).  A ticket can be waited for once.  Returns the job's status
 * (CLS_OK, or the error of its batch). */
int cls_service_wait(cls_service* s, uint64_t ticket, cls_fasta* fa, cls_placement** records);

/* paused != 0: the worker leaves the queue alone (jobs keep being accepted) until it is resumed -- e.g. to let a burst
 * of jobs accumulate into one batch per (model, parameters), or around maintenance of the registered models. */
int cls_service_pause(cls_service* s, int paused);

int cls_service_stats_get(cls_service* s, cls_service_stats* out);

#ifdef __cplusplus
}
#endif
#endif /* CLS_SERVICE_H */
