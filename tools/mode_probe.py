"""Experiment: one read ordering per run (random | leafpos | hot), prints kernel ms."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from classeq2_amd import engine, synth
engine.tuning_from_env()  # CLS_* experiment knobs (the library never reads the environment on its own)
from classeq2_amd.synth import SynthDb
mode = sys.argv[1]
s = SynthDb(10000, 1500, 12, 4)
db = engine.PlacementDb(s.flat, device=0)
n = 1_000_000
pos = np.zeros(n, dtype=np.uint32)
synth._lib().cls_synth_set_truth_pos.argtypes = [C.c_void_p]
synth._lib().cls_synth_set_truth_pos(pos.ctypes.data)
bases, offsets, truth = s.reads(n, 150)
synth._lib().cls_synth_set_truth_pos(None)
B = bases.reshape(n, 150)
def xcd(order):
    m = (len(order) // 32) * 32
    return np.concatenate([order[:m].reshape(8, -1, 4).transpose(1, 0, 2).reshape(-1), order[m:]])
leaf = truth.astype(np.int64); leaf[truth == 0xFFFFFFFF] = 1 << 40
if mode == "random": b = bases
elif mode == "leafpos": b = B[xcd(np.lexsort((pos, leaf)))].reshape(-1).copy()
elif mode == "posleaf": b = B[xcd(np.lexsort((leaf, pos)))].reshape(-1).copy()          # locus-major: all reads of a locus together, leaves inside
elif mode.startswith("posbin"):  # locus bins of N bases, leaves inside a bin
    b = B[xcd(np.lexsort((leaf, pos // int(mode[6:]))))].reshape(-1).copy()
elif mode == "leafpos_noxcd": b = B[np.lexsort((pos, leaf))].reshape(-1).copy()
elif mode.startswith("grp"):
    # groups of overlapping reads (same leaf block, same position bin), groups in RANDOM order
    _, lb, pb = mode.split("_")
    gid = (leaf // int(lb)) * 100000 + pos // int(pb)
    rng = np.random.default_rng(0)
    uniq, inv = np.unique(gid, return_inverse=True)
    perm = rng.permutation(len(uniq))
    b = B[xcd(np.argsort(perm[inv], kind="stable"))].reshape(-1).copy()
    print("groups", len(uniq), "mean size", n / len(uniq))
elif mode == "hot": b = np.tile(B[:64], (n // 64, 1)).reshape(-1).copy()
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
print(f"mode {mode} stop {os.environ.get('CLS_PROFILE_STOP','0')}: {e0.elapsed_time(e1) / 5:7.2f} ms", flush=True)
