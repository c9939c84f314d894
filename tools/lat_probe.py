"""Experiment: latency / throughput of host-buffer calls by batch size (what a service job or a Rust caller sees)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from classeq2_amd import engine
engine.tuning_from_env()  # CLS_* experiment knobs (the library never reads the environment on its own)
from classeq2_amd.synth import SynthDb
s = SynthDb(1000, 1500, 12, 4)
db = engine.PlacementDb(s.flat, device=0)
for n in (1, 100, 10000, 50000, 100000, 200000, 300000, 1000000, 100000):
    bases, offsets, _ = s.reads(n, 150)
    for _ in range(2): db.place_batch(bases, offsets)
    t = time.perf_counter()
    reps = 20 if n < 1000000 else 4
    for _ in range(reps): db.place_batch(bases, offsets)
    dt = (time.perf_counter() - t) / reps
    print(f"n={n:8d}: {dt*1e3:9.3f} ms per call -> {n/dt/1e6:8.3f} M reads/s")
