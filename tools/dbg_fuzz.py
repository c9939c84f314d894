"""Debug aid: one seed of tests/test_gpu_fuzz.py, every differing record / counter printed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from classeq2_amd import engine
from classeq2_amd.synth import SynthDb
from oracle import oracle_port as op
from tests.helpers import drop_random_nodes, ragged_reads, truncate_random_sets
seed = int(sys.argv[1]); long_reads = len(sys.argv) > 2 and sys.argv[2] == "long"
rng = np.random.default_rng(1000 + seed)
k = int(rng.choice([3, 4, 5, 7, 8, 11, 14, 15, 16, 17, 21, 31, 32, 33, 40]))
m = int(rng.choice([0, 1, 3, 4, 6, k, k + 3]))
n_leaves = int(rng.choice([2, 3, 5, 17, 64, 150]))
ref_len = int(rng.choice([max(k + 2, 20), 90, 400]))
if long_reads:
    k = max(k, 7); n_leaves = int(rng.choice([5, 17, 40])); ref_len = int(rng.choice([1500, 6200]))
collapse = float(rng.choice([0.0, 0.0, 0.3, 0.7])); deep = int(rng.choice([0, 0, 1, 2]))
s = SynthDb(n_leaves, ref_len, k, m, collapse_prob=collapse, deep=deep, seed_tree=seed + 1, seed_refseq=seed + 2,
            edge_sub_rate=float(rng.choice([0.0, 0.01, 0.05])), id_stride=int(rng.choice([1, 1, 7])), id_offset=int(rng.choice([0, 0, 100])))
mode = int(rng.integers(0, 4)); flat = s.flat
if mode == 1: flat = truncate_random_sets(flat, 0.1, seed=seed)
elif mode == 2: flat = drop_random_nodes(flat, 0.15, seed=seed)
n_reads = int(rng.choice([1, 70, 400, 5000])) if not long_reads else int(rng.choice([3, 40]))
max_len = min(ref_len, int(rng.choice([40, 200, 600]))) if not long_reads else ref_len
bases, offsets = ragged_reads(rng, s, n_reads, 0 if not long_reads else 300, max_len, err=float(rng.choice([0.0, 0.03])), frac_random=0.1, lower_frac=0.1)
kw = {}
if rng.random() < 0.5: kw["remove_intersection"] = bool(rng.random() < 0.5)
if rng.random() < 0.4: kw["max_iterations"] = int(rng.choice([0, 1, 2, 5, 1000]))
if rng.random() < 0.4: kw["min_match_coverage"] = float(rng.choice([0.0, 0.3, 0.7, 1.0, 2.0]))
print(dict(seed=seed, k=k, m=m, leaves=n_leaves, ref=ref_len, collapse=collapse, deep=deep, mode=mode, reads=n_reads, kw=kw))
want, wst = op.OraclePort(flat).place_batch(bases, offsets, op.make_params(**kw), threads=8, want_stats=True)
lens = np.diff(offsets.astype(np.int64))
for stats_on in (True, False):
    with engine.PlacementDb(flat, device=0) as db:
        if stats_on: got, gst = db.place_batch(bases, offsets, engine.make_params(**kw), want_stats=True)
        else: got = db.place_batch(bases, offsets, engine.make_params(**kw))
        print("info", db.info.format, db.info.binary_tree, db.info.direct_table, "stats kernel" if stats_on else "plain kernel")
    for i in range(len(lens)):
        rec_bad = any(got[f][i] != want[f][i] for f in ("status", "one", "rest", "levels", "clade_id"))
        st_bad = stats_on and any(gst[f][i] != wst[f][i] for f in ("n_query_kmers", "n_matched", "n_with_root", "leaf_postings"))
        if rec_bad or st_bad:
            print(i, "len", lens[i], "GOT", got[i], gst[i] if stats_on else "", "WANT", want[i], wst[i])
