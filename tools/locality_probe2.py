"""Experiment: upper bounds of locality orderings (true leaf / true position known)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classeq2_amd import engine, synth
engine.tuning_from_env()  # CLS_* experiment knobs (the library never reads the environment on its own)
from classeq2_amd.synth import SynthDb
s = SynthDb(10000, 1500, 12, 4)
db = engine.PlacementDb(s.flat, device=0)
n = 1_000_000
pos = np.zeros(n, dtype=np.uint32)
synth._lib().cls_synth_set_truth_pos.argtypes = [C.c_void_p]
synth._lib().cls_synth_set_truth_pos(pos.ctypes.data)
bases, offsets, truth = s.reads(n, 150)
synth._lib().cls_synth_set_truth_pos(None)
B = bases.reshape(n, 150)
def run(order, tag):
    b = B[order].reshape(-1).copy() if order is not None else bases
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
    print(f"{tag:60s} {e0.elapsed_time(e1) / 5:7.2f} ms", flush=True)
def xcd(order):  # 8 contiguous ranges interleaved at 4-read-block granularity (block b -> XCD b%8)
    m = (len(order) // 32) * 32
    return np.concatenate([order[:m].reshape(8, -1, 4).transpose(1, 0, 2).reshape(-1), order[m:]])
run(None, "random")
leaf = truth.astype(np.int64); leaf[truth == 0xFFFFFFFF] = 1 << 40
run(xcd(np.argsort(leaf, kind="stable")), "leaf")
run(xcd(np.lexsort((pos, leaf))), "leaf, pos")
run(xcd(np.lexsort((leaf, pos))), "pos, leaf")
run(xcd(np.lexsort((leaf, pos // 32))), "pos/32, leaf")
run(xcd(np.lexsort((pos, leaf // 16))), "leaf/16, pos")
run(xcd(np.lexsort((pos, leaf // 128))), "leaf/128, pos")
run(xcd(np.lexsort((leaf, pos // 150))), "pos/150, leaf")
