"""Experiment: polytomy tree (support-collapsed), split-tree walk vs sorted lists."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classeq2_amd import engine
engine.tuning_from_env()  # CLS_* experiment knobs (the library never reads the environment on its own)
from classeq2_amd.synth import SynthDb
s = SynthDb(10000, 1500, int(sys.argv[1]) if len(sys.argv) > 1 else 12, 4, collapse_prob=0.3)
db = engine.PlacementDb(s.flat, device=0)
print("format", db.info.format, "binary", db.info.binary_tree, "max arity", db.info.max_nonleaf_arity, "depth", db.info.max_depth)
n = 1_000_000
bases, offsets, _ = s.reads(n, 150)
d_b = torch.from_numpy(bases).cuda(); d_o = torch.from_numpy(offsets.view(np.int64)).cuda()
d_out = torch.zeros(n * 24, dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(2): db.place_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), None, 0, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): db.place_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), None, 0, st)
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1)/3:.2f} ms per 1M reads -> {n/(e0.elapsed_time(e1)/3)*1e3/1e6:.1f} M placements/s")
