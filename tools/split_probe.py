"""Experiment: one 1M-read device call vs two concurrent 500k-read calls on two streams (C3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classeq2_amd import engine
engine.tuning_from_env()  # CLS_* experiment knobs (the library never reads the environment on its own)
from classeq2_amd.synth import SynthDb, CONFIGS
cfg = CONFIGS["C3"]
s = SynthDb(cfg["n_leaves"], cfg["ref_len"], cfg["k_size"], cfg["m_size"])
db = engine.PlacementDb(s.flat, device=0)
n = 1_000_000
bases, offsets, _ = s.reads(n, 150)
d_b = torch.from_numpy(bases).cuda(); d_o = torch.from_numpy(offsets.view(np.int64)).cuda()
d_out = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
def run(parts, reps=10):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    per = n // parts
    def go():
        for p, st in enumerate(streams):
            db.place_batch_device(d_b.data_ptr(), d_o.data_ptr() + 8 * p * per, per, d_out.data_ptr() + 24 * p * per, None, 0, st.cuda_stream)
    for _ in range(2): go()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): go()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3
for parts in (1, 2, 4, 1, 2):
    print(parts, f"{run(parts):.3f} ms")
