"""Long reads (10 kb) on the index shapes the LDS-tiled kernel does not take -- a hashed front (k > 15) and / or a
support-collapsed tree with polytomies: which kernel they run on and at what rate, next to the binary / direct-table shape.
usage: python tools/long_probe.py [n_reads] [read_len] [ref_len]   -> one JSON line per shape (GPU box)
(CLS_TILE_MIN_KMERS=<n> sends shorter reads through the LDS-tiled kernel too)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from classeq2_amd import engine
engine.tuning_from_env()  # CLS_* experiment knobs (the library never reads the environment on its own)
engine.set_tuning("time_class", 2)  # name the kernel of the gene-length classes, not the wave-per-read one
from classeq2_amd.synth import SynthDb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
read_len = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
ref_len = int(sys.argv[3]) if len(sys.argv) > 3 else read_len + 2000
dev = torch.device("cuda:0")
for name, k, collapse in (("binary tree, k=15 (direct table): the LDS-tiled kernel", 15, 0.0), ("binary tree, k=35 (hash table)", 35, 0.0),
                          ("support-collapsed tree (polytomies), k=15", 15, 0.3), ("support-collapsed tree, k=35: the reference's default shape", 35, 0.3)):
    s = SynthDb(300, ref_len, k, 4, collapse_prob=collapse)
    db = engine.PlacementDb(s.flat, device=0)
    db.set_max_read_len(read_len)
    bases, offsets, _ = s.reads(n, read_len)
    d_b = torch.from_numpy(bases).to(dev); d_o = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    d_out = torch.zeros(n * 24, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    db.place_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), None, 0, st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        db.place_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), None, 0, st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(json.dumps({"shape": name, "k": k, "collapse_prob": collapse, "reads": n, "read_len": read_len, "ms": round(ms, 2), "reads_per_s": round(n / ms * 1e3),
                      "gbase_per_s": round(n * read_len / ms / 1e6, 2), "kernel_timed": db.kernel_name(), "binary_tree": int(db.info.binary_tree), "direct_table": int(db.info.direct_table)}), flush=True)
    db.close(); s.close()
