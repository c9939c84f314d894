"""Resident batching service: many small jobs from several threads, two resident models, mixed
parameters -- every job gets exactly the records a direct call gives, and waiting jobs share device batches."""
import threading

import numpy as np
import pytest

from classeq2_amd import engine
from classeq2_amd.synth import SynthDb
from oracle import oracle_port as op
from tests.helpers import records_equal


def _fasta(bases, offsets, tag):
    raw = bytes(bases)
    return b"".join(b">%s_%d\n%s\n" % (tag, i, raw[int(offsets[i]):int(offsets[i + 1])]) for i in range(len(offsets) - 1))


def test_service_refuses_unknown_model_without_gpu():
    with engine.Service() as svc:
        with pytest.raises(engine.ClsError, match="unknown model id"):
            svc.submit("nope", b">a\nACGT\n")
        assert svc.stats()["jobs_submitted"] == 0


@pytest.mark.gpu
def test_jobs_share_batches_and_keep_their_results():
    models = {"m8": SynthDb(120, 400, 8, 4), "m13": SynthDb(90, 500, 13, 4, collapse_prob=0.3)}
    ports = {k: op.OraclePort(v.flat) for k, v in models.items()}
    param_sets = [None, dict(remove_intersection=True), dict(max_iterations=3)]
    jobs = []
    rng = np.random.default_rng(2)
    for j in range(240):
        mid = "m8" if j % 3 else "m13"
        n = int(rng.integers(0, 40))
        bases, offsets, _ = models[mid].reads(n, 100, seed=1000 + j) if n else (np.zeros(0, np.uint8), np.zeros(1, np.uint64), None)
        kw = param_sets[j % len(param_sets)]
        jobs.append((mid, _fasta(bases, offsets, b"j%d" % j), kw, bases, offsets))
    results = [None] * len(jobs)
    with engine.Service() as svc:
        for mid, s in models.items():
            svc.add_model(mid, engine.PlacementDb(s.flat, device=0))

        def client(lo, hi, wait):
            tickets = [(i, svc.submit(jobs[i][0], jobs[i][1], engine.make_params(**jobs[i][2]) if jobs[i][2] else None)) for i in range(lo, hi)]
            if wait:
                for i, t in tickets:
                    results[i] = svc.wait(t)
            return tickets

        # a burst while the worker is held: everything that waits with the same (model, parameters) becomes ONE batch
        svc.pause(True)
        held = client(0, 120, wait=False)
        assert svc.stats()["jobs_done"] == 0
        svc.pause(False)
        for i, t in held:
            results[i] = svc.wait(t)
        st = svc.stats()
        assert st["jobs_done"] == 120 and st["device_batches"] <= 6 and st["max_jobs_in_batch"] >= 120 // 6
        # then concurrent clients against the running worker
        th = [threading.Thread(target=client, args=(120 + k * 30, 120 + (k + 1) * 30, True)) for k in range(4)]
        [t.start() for t in th]
        [t.join() for t in th]
        st = svc.stats()
        with pytest.raises(engine.ClsError, match="unknown ticket"):
            svc.wait(1)
    assert st["jobs_submitted"] == st["jobs_done"] == 240 and st["models"] == 2
    assert st["device_batches"] <= 6 + 120
    for i, (mid, text, kw, bases, offsets) in enumerate(jobs):
        headers, got, truncated = results[i]
        n = len(offsets) - 1
        assert len(headers) == n and not truncated and (n == 0 or headers[0] == b"j%d_0" % i)
        want = ports[mid].place_batch(bases, offsets, op.make_params(**kw) if kw else op.make_params(), threads=2)
        assert len(records_equal(got, want)) == 0, (i, mid, kw)
