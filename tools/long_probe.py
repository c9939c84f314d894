"""Experiment: 10 kb reads (BASELINE config 5 shape, reduced tree) through the workspace kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from classeq2_amd import engine
engine.tuning_from_env()  # CLS_* experiment knobs (the library never reads the environment on its own)
from classeq2_amd.synth import SynthDb
k = int(sys.argv[1]) if len(sys.argv) > 1 else 15
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
s = SynthDb(300, 12000, k, 4, deep=1)
db = engine.PlacementDb(s.flat, device=0)
print("depth", db.info.max_depth, "kmers", db.info.n_kmers, "direct", db.info.direct_table)
bases, offsets, _ = s.reads(n, 10000)
db.place_batch(bases[:10000 * 50], offsets[:51])
t = time.time(); out = db.place_batch(bases, offsets); dt = time.time() - t
print(f"{n} x 10 kb reads: {dt*1e3:.1f} ms -> {n/dt:.0f} reads/s ({n*10000/dt/1e6:.1f} Mbases/s), levels mean {out['levels'].mean():.1f}")
