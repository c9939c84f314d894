"""Experiment: throughput by read length on the C3 index (which kernel class each length lands in)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classeq2_amd import engine
engine.tuning_from_env()  # CLS_* experiment knobs (the library never reads the environment on its own)
from classeq2_amd.synth import SynthDb
collapse = float(os.environ.get("COLLAPSE", "0"))
s = SynthDb(10000, 4500, 12, 4, collapse_prob=collapse)
db = engine.PlacementDb(s.flat, device=0)
db.set_max_read_len(20000)
st = torch.cuda.current_stream().cuda_stream
for L, n in ((150, 1000000), (250, 600000), (500, 300000), (1000, 100000), (1500, 100000), (3000, 50000), (4400, 30000)):
    bases, offsets, _ = s.reads(n, L)
    d_b = torch.from_numpy(bases).cuda(); d_o = torch.from_numpy(offsets.view(np.int64)).cuda()
    d_out = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    for _ in range(2): db.place_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), None, 0, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): db.place_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), None, 0, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"L={L:5d} n={n:8d}: {ms:8.2f} ms -> {n/ms*1e3/1e6:7.2f} M reads/s, {n*L/ms*1e3/1e9:6.2f} Gbases/s")
