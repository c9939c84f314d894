"""Experiment: BASELINE config 5 at a given scale through the device-buffer entry; time per launch as a function of the
iteration cap (1 = the front alone, then the cost per level), and the per-read counters.
usage: c5_probe.py <scale> [n_reads] [read_len]      (CLS_NO_TILE=1: the workspace kernel instead of the LDS-tiled one)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classeq2_amd import _abi, engine
engine.tuning_from_env()
from classeq2_amd.synth import CONFIGS, SynthDb
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
cfg = CONFIGS["C5"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else max(2000, int(cfg["n_reads"] * scale * 0.2))
if len(sys.argv) > 3:
    cfg = dict(cfg, read_len=int(sys.argv[3]))
t0 = time.time()
s = SynthDb(max(64, int(cfg["n_leaves"] * scale)), cfg["ref_len"], cfg["k_size"], cfg["m_size"], deep=1, max_depth=900, tips_only=True)
db = engine.PlacementDb(s.flat, device=0)
db.set_max_read_len(cfg["read_len"])
print(f"setup {time.time()-t0:.1f}s depth {db.info.max_depth} kmers {db.info.n_kmers} sets {db.info.n_tip_sets} kernel {db.kernel_name()}", flush=True)
bases, offsets, _ = s.reads(n, cfg["read_len"], seed=3)
dev = torch.device("cuda:0")
d_b = torch.from_numpy(bases).to(dev); d_o = torch.from_numpy(offsets.astype(np.int64)).to(dev)
d_out = torch.zeros(n * 24, dtype=torch.uint8, device=dev); d_st = torch.zeros(n * 24, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(max_it, stats=False, reps=3):
    p = engine.make_params(max_iterations=max_it) if max_it else None
    db.place_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), p, d_st.data_ptr() if stats else 0, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        db.place_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), p, 0, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for cap in (1, 10, 40, 0):
    ms = run(cap)
    print(f"max_iterations {cap or 'default'}: {ms:9.2f} ms per {n} reads -> {n/ms*1e3:9.0f} reads/s", flush=True)
run(0, stats=True, reps=1)
stt = d_st.cpu().numpy().view(_abi.STATS_DTYPE); rec = d_out.cpu().numpy().view(_abi.PLACEMENT_DTYPE)
print(f"per read: matched {stt['n_matched'].mean():.0f} of {stt['n_query_kmers'].mean():.0f} k-mers, with root {stt['n_with_root'].mean():.0f}, "
      f"index bytes {stt['index_bytes'].mean():.0f}, levels {rec['levels'].mean():.1f} (max {rec['levels'].max()})")
