"""Experiment: how much does processing reads in leaf order help (upper bound of locality sorting)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classeq2_amd import engine
engine.tuning_from_env()  # CLS_* experiment knobs (the library never reads the environment on its own)
from classeq2_amd.synth import SynthDb
s = SynthDb(10000, 1500, 12, 4)
db = engine.PlacementDb(s.flat, device=0)
n = 1_000_000
bases, offsets, truth = s.reads(n, 150)
def run(b, tag):
    d_b = torch.from_numpy(b).cuda(); d_o = torch.from_numpy(offsets.view(np.int64)).cuda()
    d_out = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        db.place_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), None, 0, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        db.place_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), None, 0, st)
    e1.record(); torch.cuda.synchronize()
    print(tag, e0.elapsed_time(e1) / 5, "ms", flush=True)
run(bases, "random order")
order = np.argsort(truth, kind="stable")
run(bases.reshape(n, 150)[order].reshape(-1).copy(), "sorted by true leaf")
# sorted by leaf then start position is not available; approximate with leaf-block interleave for XCD locality
blk = order.reshape(8, -1)  # 8 contiguous ranges
inter = blk.T.reshape(-1)   # consecutive reads alternate between the 8 ranges (block b%8 -> range)
run(bases.reshape(n, 150)[inter].reshape(-1).copy(), "sorted, 8-way interleaved (per read)")
# interleave at block granularity (4 reads per block)
blk4 = order.reshape(8, -1, 4).transpose(1, 0, 2).reshape(-1)
run(bases.reshape(n, 150)[blk4].reshape(-1).copy(), "sorted, 8-way interleaved (per 4-read block)")
one = bases[:150 * 64].reshape(64, 150)
run(np.tile(one, (n // 64, 1)).reshape(-1).copy(), "64 distinct reads tiled (all cache-hot)")
for stop in (1, 2):
    pass
