"""Differential sweep: small random indexes of every shape the encoder distinguishes (k on both sides of the direct
table limit, m from 0 to beyond k, bushy / ladder / collapsed trees, truncated and non-closed node sets) x ragged reads
x Option<> parameters, HIP path against the oracle.  Seeds are fixed: the sweep is deterministic."""
import numpy as np
import pytest

from classeq2_amd import _abi, engine
from classeq2_amd.synth import SynthDb
from oracle import oracle_port as op
from tests.helpers import drop_random_nodes, ragged_reads, records_equal, stats_equal, truncate_random_sets

pytestmark = pytest.mark.gpu


def _one(seed: int, long_reads: bool = False):
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.choice([3, 4, 5, 7, 8, 11, 14, 15, 16, 17, 21, 31, 32, 33, 40]))
    m = int(rng.choice([0, 1, 3, 4, 6, k, k + 3]))
    n_leaves = int(rng.choice([2, 3, 5, 17, 64, 150]))
    ref_len = int(rng.choice([max(k + 2, 20), 90, 400]))
    if long_reads:  # the workgroup-per-read and workspace kernels: reads of up to 6 kb
        k = max(k, 7)
        n_leaves = int(rng.choice([5, 17, 40]))
        ref_len = int(rng.choice([1500, 6200]))
    collapse = float(rng.choice([0.0, 0.0, 0.3, 0.7]))
    deep = int(rng.choice([0, 0, 1, 2]))
    s = SynthDb(n_leaves, ref_len, k, m, collapse_prob=collapse, deep=deep, seed_tree=seed + 1, seed_refseq=seed + 2,
                edge_sub_rate=float(rng.choice([0.0, 0.01, 0.05])), id_stride=int(rng.choice([1, 1, 7])), id_offset=int(rng.choice([0, 0, 100])))
    mode = int(rng.integers(0, 4))
    flat = s.flat
    if mode == 1:
        flat = truncate_random_sets(flat, 0.1, seed=seed)
    elif mode == 2:
        flat = drop_random_nodes(flat, 0.15, seed=seed)
    n_reads = int(rng.choice([1, 70, 400, 5000])) if not long_reads else int(rng.choice([3, 40]))
    max_len = min(ref_len, int(rng.choice([40, 200, 600]))) if not long_reads else ref_len
    bases, offsets = ragged_reads(rng, s, n_reads, 0 if not long_reads else 300, max_len, err=float(rng.choice([0.0, 0.03])),
                                  frac_random=0.1, lower_frac=0.1)
    kw = {}
    if rng.random() < 0.5:
        kw["remove_intersection"] = bool(rng.random() < 0.5)
    if rng.random() < 0.4:
        kw["max_iterations"] = int(rng.choice([0, 1, 2, 5, 1000]))
    if rng.random() < 0.4:
        kw["min_match_coverage"] = float(rng.choice([0.0, 0.3, 0.7, 1.0, 2.0]))
    with engine.PlacementDb(flat, device=0) as db:
        got, gst = db.place_batch(bases, offsets, engine.make_params(**kw), want_stats=True)
        info = (db.info.format, db.info.binary_tree, db.info.direct_table)
    want, wst = op.OraclePort(flat).place_batch(bases, offsets, op.make_params(**kw), threads=8, want_stats=True)
    ctx = dict(seed=seed, k=k, m=m, leaves=n_leaves, ref=ref_len, collapse=collapse, deep=deep, mode=mode, reads=n_reads, kw=kw, info=info)
    bad = records_equal(got, want)
    assert len(bad) == 0, (ctx, int(bad[0]), got[bad[0]], want[bad[0]])
    sb = stats_equal(gst, wst)
    assert len(sb) == 0, (ctx, int(sb[0]), gst[sb[0]], wst[sb[0]])
    return info


@pytest.mark.parametrize("block", range(5))
def test_random_indexes_reads_and_parameters(block):
    seen = set()
    for seed in range(block * 12, block * 12 + 12):
        seen.add(_one(seed))
    assert len(seen) >= 2  # more than one index layout per block of seeds


def test_random_indexes_with_gene_length_and_long_reads():
    for seed in range(200, 214):
        _one(seed, long_reads=True)
