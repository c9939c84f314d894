"""Stress: several threads issue large (two-slot, pipelined) and small host-buffer calls on one handle at once."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from classeq2_amd import engine
engine.tuning_from_env()  # CLS_* experiment knobs (the library never reads the environment on its own)
from classeq2_amd.synth import SynthDb
s = SynthDb(400, 1200, 12, 4)
db = engine.PlacementDb(s.flat, device=0)
n = 600_000
bases, offsets, _ = s.reads(n, 100)
ref = db.place_batch(bases, offsets)
errors = []
def big(i):
    for it in range(4):
        got = db.place_batch(bases, offsets)
        if not all((got[f] == ref[f]).all() for f in ("status", "one", "rest", "levels", "clade_id")): errors.append(("big", i, it))
def small(i):
    for it in range(300):
        m = 50 + (i * 37 + it * 11) % 3000
        got = db.place_batch(bases[: 100 * m], offsets[: m + 1])
        if not all((got[f] == ref[:m][f]).all() for f in ("status", "one", "rest", "levels", "clade_id")): errors.append(("small", i, it, m))
t0 = time.time()
th = [threading.Thread(target=big, args=(i,)) for i in range(3)] + [threading.Thread(target=small, args=(i,)) for i in range(4)]
[t.start() for t in th]; [t.join() for t in th]
print("errors", errors[:5], len(errors), "seconds", round(time.time() - t0, 2))
