"""GPU parity: HIP path (through the C-ABI) vs the oracle on the same seeded inputs."""
import numpy as np
import pytest

from classeq2_amd import _abi, engine
from classeq2_amd.synth import CONFIGS, SynthDb
from oracle import oracle_port as op
from tests.helpers import (ODD_PARAM_SETS, PARAM_SETS, describe, drop_random_nodes, ragged_reads, records_equal, stats_equal,
                           truncate_random_sets)

pytestmark = pytest.mark.gpu

CASES = [
    # n_leaves, ref_len, k, m, collapse_prob, id_stride, id_offset, read_len
    (40, 200, 6, 3, 0.0, 1, 0, 60),
    (60, 300, 8, 4, 0.3, 3, 7, 80),
    (100, 300, 10, 4, 0.5, 1, 0, 100),
    (30, 120, 5, 0, 0.5, 5, 100, 40),
    (50, 200, 7, 9, 0.0, 1, 0, 20),
    (200, 500, 12, 4, 0.0, 1, 0, 150),
    (150, 400, 35, 4, 0.1, 2, 1, 150),
    (80, 300, 17, 4, 0.0, 1, 0, 120),
]


def _check(flat, bases, offsets, kw, threads=4):
    with engine.PlacementDb(flat, device=0) as db:
        got, gst = db.place_batch(bases, offsets, engine.make_params(**kw), want_stats=True)
        got2 = db.place_batch(bases, offsets, engine.make_params(**kw))
    want, wst = op.OraclePort(flat).place_batch(bases, offsets, op.make_params(**kw), threads=threads, want_stats=True)
    bad = records_equal(got, want)
    assert len(bad) == 0, f"{len(bad)} records differ, first {bad[0]}: got {describe(got[bad[0]])} want {describe(want[bad[0]])}"
    assert len(records_equal(got, got2)) == 0, "stats and non-stats kernels disagree"
    sb = stats_equal(gst, wst)
    assert len(sb) == 0, f"{len(sb)} stats differ, first {sb[0]}: got {gst[sb[0]]} want {wst[sb[0]]}"
    return got


@pytest.mark.parametrize("case", CASES)
def test_synthetic_parity(case):
    nl, rl, k, m, cp, stride, off, rdlen = case
    s = SynthDb(nl, rl, k, m, collapse_prob=cp, id_stride=stride, id_offset=off)
    bases, offsets, _ = s.reads(1500, rdlen, frac_random=0.05, err=0.02)
    for kw in PARAM_SETS:
        _check(s.flat, bases, offsets, kw)


@pytest.mark.parametrize("frac", [0.05, 0.3])
def test_non_closed_node_sets(frac):
    """Node sets that are not closed under `parent` (explicit-list postings)."""
    s = SynthDb(80, 300, 8, 4, collapse_prob=0.3)
    flat = drop_random_nodes(s.flat, frac, seed=11)
    bases, offsets, _ = s.reads(1500, 90, frac_random=0.05, err=0.02)
    for kw in PARAM_SETS[:3]:
        _check(flat, bases, offsets, kw)


@pytest.mark.parametrize("k,collapse", [(9, 0.0), (12, 0.0), (10, 0.4), (17, 0.0)])
def test_sets_with_nothing_below_the_root(k, collapse):
    """k-mers whose node set is {root} or {} count towards |M| / |M_root| but never vote; every kernel
    family (direct-table fast path for k <= 15 on binary trees, split-tree walk, hash probe)."""
    s = SynthDb(70, 300, k, 4, collapse_prob=collapse)
    flat = truncate_random_sets(s.flat, 0.15, seed=4)
    bases, offsets, _ = s.reads(1200, 110, frac_random=0.05, err=0.02)
    for kw in (dict(), dict(min_match_coverage=1.0), dict(remove_intersection=True)):
        _check(flat, bases, offsets, kw)
    with engine.PlacementDb(flat, device=0) as db:
        assert db.info.format == 1 and db.info.direct_table == (1 if k <= 15 else 0)


def test_ragged_and_edge_reads():
    s = SynthDb(100, 400, 9, 4)
    rng = np.random.default_rng(5)
    bases, offsets = ragged_reads(rng, s, 800, 0, 168)
    got = _check(s.flat, bases, offsets, {})
    lens = np.diff(offsets.astype(np.int64))
    assert (got["status"][lens < 9] == _abi.ERR_TOO_FEW_KMERS).all()
    # invalid characters -> per-read error (the reference panics), only when len >= k
    b2 = bases.copy()
    o = offsets.astype(np.int64)
    victims = [i for i in range(len(lens)) if lens[i] >= 9][:20]
    for i in victims:
        b2[o[i] + lens[i] // 2] = ord("N")
    got2 = _check(s.flat, b2, offsets, {})
    assert (got2["status"][victims] == _abi.ERR_INVALID_BASE).all()


def _device_place(db, bases, offsets, want_stats=True):
    """cls_place_batch_device on torch-owned HBM buffers."""
    import torch

    dev = torch.device("cuda:0")
    n = len(offsets) - 1
    d_b = torch.from_numpy(bases if len(bases) else np.zeros(1, np.uint8)).to(dev)
    d_o = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    d_out = torch.zeros(n * 24, dtype=torch.uint8, device=dev)
    d_st = torch.zeros(n * 24, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    db.place_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), None, d_st.data_ptr() if want_stats else 0, 0)
    torch.cuda.synchronize()
    return d_out.cpu().numpy().view(_abi.PLACEMENT_DTYPE), d_st.cpu().numpy().view(_abi.STATS_DTYPE)


def test_read_too_long_reported():
    """The device-buffer entry provisions for the handle's max_read_len; beyond it reads are refused, not misplaced."""
    s = SynthDb(30, 21000, 9, 4)
    bases, offsets, _ = s.reads(6, 20000)
    with engine.PlacementDb(s.flat, device=0) as db:
        assert 2 * (20000 - 9 + 1) > db.info.max_read_kmers
        got, st = _device_place(db, bases, offsets)
        assert (got["status"] == _abi.ERR_READ_TOO_LONG).all()
        assert (st["n_query_kmers"] == 2 * (20000 - 9 + 1)).all()
        db.set_max_read_len(20000)
        assert db.info.max_read_kmers == 40000
        got2, st2 = _device_place(db, bases, offsets)
        host, hst = db.place_batch(bases, offsets, want_stats=True)  # sizes itself by the batch
    want, wst = op.OraclePort(s.flat).place_batch(bases, offsets, op.make_params(), threads=8, want_stats=True)
    assert len(records_equal(got2, want)) == 0 and len(records_equal(host, want)) == 0
    assert len(stats_equal(st2, wst)) == 0 and len(stats_equal(hst, wst)) == 0


def test_declared_read_length_drops_the_classes_beyond_it():
    """cls_db_set_max_read_len(n): the device-buffer entry launches the read-length classes up to n only -- same records
    for the reads within it, CLS_ERR_READ_TOO_LONG for the one beyond; the host-buffer entry sizes itself by the batch."""
    s = SynthDb(60, 3000, 11, 4, collapse_prob=0.2)
    rng = np.random.default_rng(12)
    parts = [ragged_reads(rng, s, 300, 0, 160), ragged_reads(rng, s, 1, 1000, 1001)]
    bases = np.concatenate([p[0] for p in parts])
    offsets = np.concatenate([parts[0][1], parts[1][1][1:] + parts[0][1][-1]])
    want = op.OraclePort(s.flat).place_batch(bases, offsets, op.make_params(), threads=8)
    with engine.PlacementDb(s.flat, device=0) as db:
        for n_bases, refused in ((0, False), (160, True), (290, True), (1000, False)):
            db.set_max_read_len(n_bases)
            got, _ = _device_place(db, bases, offsets)
            assert len(records_equal(got[:-1], want[:-1])) == 0
            assert (got["status"][-1] == _abi.ERR_READ_TOO_LONG) == refused
            if not refused:
                assert len(records_equal(got, want)) == 0
        assert len(records_equal(db.place_batch(bases, offsets), want)) == 0


@pytest.mark.parametrize("k,collapse,drop,deep", [(15, 0.0, 0.0, 0), (12, 0.4, 0.0, 0), (11, 0.3, 0.2, 0), (15, 0.0, 0.0, 1), (20, 0.5, 0.0, 1)])
def test_long_reads_workspace_kernel(k, collapse, drop, deep):
    """Reads of 4.2..11 kb (BASELINE config 5 has 10 kb reads): more k-mers than the register-resident kernels
    hold, so the per-k-mer state lives in the workspace; every index format, bushy and ladder-like trees."""
    s = SynthDb(90, 11500, k, 4, collapse_prob=collapse, deep=deep)
    flat = drop_random_nodes(s.flat, drop, seed=8) if drop else s.flat
    rng = np.random.default_rng(33)
    bases, offsets = ragged_reads(rng, s, 40, 4200, 11000, lower_frac=0.05)
    got = None
    engine.set_tuning("no_tile", 1)
    try:
        for kw in (dict(), dict(remove_intersection=True, max_iterations=9)):
            got = _check(flat, bases, offsets, kw, threads=16)
    finally:
        engine.set_tuning("no_tile", 0)
    assert (got["status"] != _abi.ERR_READ_TOO_LONG).all()


@pytest.mark.parametrize("k,collapse,deep,leaves", [(15, 0.3, 0, 200), (12, 0.6, 1, 120), (20, 0.5, 1, 90), (35, 0.3, 0, 200), (13, 0.85, 0, 150)])
def test_long_reads_lds_tiled_kernel_polytomy_levels(k, collapse, deep, leaves):
    """Support-collapsed trees (the reference's default build collapses weakly supported clades): at a clade that does not
    have exactly two children the LDS-tiled kernel walks every entry's chain of occupied children and keeps per-child
    counters in LDS (place_sequence.rs:369-417, :519-599); binary levels in between take the two-counter path.  Direct and
    hashed front, both remove_intersection settings, a low iteration cap; same records with the kernel switched off."""
    s = SynthDb(leaves, 11000, k, 4, collapse_prob=collapse, deep=deep)
    rng = np.random.default_rng(70 + k)
    bases, offsets = ragged_reads(rng, s, 48, 4200, 10800, lower_frac=0.05)
    with engine.PlacementDb(s.flat, device=0) as db:
        assert (db.info.format, db.info.binary_tree) == (1, 0)
        assert db.info.max_nonleaf_arity > (4 if collapse > 0.8 else 2)
        db.set_max_read_len(10800)
        assert db.kernel_name().startswith("place_tile_kernel<") and db.kernel_name().endswith(", true>")
    got = {}
    for kw in (dict(), dict(remove_intersection=True), dict(max_iterations=4)):
        got[tuple(kw)] = _check(s.flat, bases, offsets, kw, threads=16)
    assert (got[()]["status"] != _abi.ERR_READ_TOO_LONG).all()
    engine.set_tuning("no_tile", 1)
    try:
        for kw in (dict(), dict(remove_intersection=True)):
            assert len(records_equal(_check(s.flat, bases, offsets, kw, threads=16), got[tuple(kw)])) == 0
    finally:
        engine.set_tuning("no_tile", 0)


@pytest.mark.parametrize("k,collapse", [(9, 0.0), (9, 0.3), (17, 0.0)])
def test_ladder_tree_deeper_than_the_iteration_cap(k, collapse):
    """A ladder-like tree of depth > 1000 (BASELINE config 5 is capped at 900 for this reason): with the default
    cap the deepest reads end in ERR_MAX_ITER at level 1001 (place_sequence.rs:295-301), with a larger one they
    resolve more than a thousand levels down."""
    s = SynthDb(2300, 60, k, 4, deep=2, edge_sub_rate=0.02, collapse_prob=collapse)
    assert s.max_depth > (1050 if collapse == 0.0 else 600)
    bases, offsets, _ = s.reads(300, 60, err=0.0, frac_random=0.02)
    got = _check(s.flat, bases, offsets, {}, threads=8)
    got5k = _check(s.flat, bases, offsets, dict(max_iterations=5000, remove_intersection=True), threads=8)
    if collapse == 0.0:
        assert (got["status"] == _abi.ERR_MAX_ITER).sum() > 10 and (got["levels"][got["status"] == _abi.ERR_MAX_ITER] == 1001).all()
        assert got5k["levels"].max() > 1050 and (got5k["status"] != _abi.ERR_MAX_ITER).all()


@pytest.mark.parametrize("k,collapse,trunc", [(21, 0.0, False), (35, 0.3, False), (12, 0.3, False), (12, 0.0, True), (19, 0.0, True)])
def test_large_batches_take_the_locality_ordered_path(k, collapse, trunc):
    """>= 4096 reads: the key kernel + sort + XCD-ordered walk, for both wave-per-read classes (reads of 60..480 bp),
    with the direct table (k <= 15: strand-symmetric and not) and with the MurmurHash3 front (k > 15)."""
    s = SynthDb(300, 900, k, 4, collapse_prob=collapse)
    flat = truncate_random_sets(s.flat, 0.05, seed=6) if trunc else s.flat
    rng = np.random.default_rng(17)
    bases, offsets = ragged_reads(rng, s, 6000, 60, 480, lower_frac=0.02)
    assert len(offsets) - 1 >= 4096
    for kw in (dict(), dict(remove_intersection=True)):
        _check(flat, bases, offsets, kw, threads=16)


def test_long_and_short_reads_in_one_batch():
    """One batch through all four kernels (320 / 1024 / 8192 k-mers and the workspace kernel), an invalid base
    in a long read, and a long read of random bases."""
    s = SynthDb(120, 9000, 13, 4, collapse_prob=0.2)
    rng = np.random.default_rng(5)
    parts = [ragged_reads(rng, s, 150, 0, 400), ragged_reads(rng, s, 20, 600, 4000), ragged_reads(rng, s, 12, 4300, 8800, frac_random=0.2)]
    bases = np.concatenate([p[0] for p in parts])
    offsets = np.zeros(1, dtype=np.uint64)
    for b, o in parts:
        offsets = np.concatenate([offsets, o[1:] + offsets[-1]])
    bases = bases.copy()
    o = offsets.astype(np.int64)
    bases[o[-2] + 3000] = ord("N")  # inside the last read, a long one
    got = _check(s.flat, bases, offsets, {}, threads=16)
    assert got["status"][-1] == _abi.ERR_INVALID_BASE


@pytest.mark.parametrize("k,collapse,drop", [(12, 0.0, 0.0), (35, 0.0, 0.0), (11, 0.4, 0.0), (10, 0.3, 0.2), (16, 0.0, 0.15)])
def test_gene_length_reads_workgroup_kernel(k, collapse, drop):
    """Reads of 600..3500 bp (marker-gene queries), every index format: the LDS-tiled kernel's shared launches (8, 4, 2
    workgroups per CU by read length) where the index has the shape for it, the workgroup-per-read kernel where it has not
    -- and with the LDS-tiled kernel switched off."""
    s = SynthDb(150, 4000, k, 4, collapse_prob=collapse)
    flat = drop_random_nodes(s.flat, drop, seed=8) if drop else s.flat
    rng = np.random.default_rng(21)
    bases, offsets = ragged_reads(rng, s, 120, 300, 3500, lower_frac=0.05)
    got = {}
    for kw in (dict(), dict(remove_intersection=True, max_iterations=6)):
        got[tuple(kw)] = _check(flat, bases, offsets, kw, threads=16)
    for knob, value in (("no_tile", 1), ("tile_one_per_cu", 1), ("tile_min_kmers", 2000)):
        engine.set_tuning(knob, value)
        try:
            for kw in (dict(), dict(remove_intersection=True, max_iterations=6)):
                assert len(records_equal(_check(flat, bases, offsets, kw, threads=16), got[tuple(kw)])) == 0
        finally:
            engine.set_tuning(knob, 0)


@pytest.mark.parametrize("collapse", [0.0, 0.4])
def test_mixed_read_lengths_use_both_kernels(collapse):
    """Reads of 0..500 bp: the 320-k-mer and the 1024-k-mer kernels in one batch."""
    s = SynthDb(120, 600, 9, 4, collapse_prob=collapse)
    rng = np.random.default_rng(9)
    bases, offsets = ragged_reads(rng, s, 600, 0, 500)
    _check(s.flat, bases, offsets, {})
    _check(s.flat, bases, offsets, dict(remove_intersection=True))
    _check(drop_random_nodes(s.flat, 0.2, seed=4), bases, offsets, {})


def test_huge_polytomy_uses_global_child_counters():
    """A root with 300 internal children (non-LEAF arity > 256): the per-child counters leave LDS for the
    global scratch area; the same shape at arity 40 stays in LDS."""
    from tests.helpers import star_of_cherries
    for n_cherries in (300, 40):
        flat, seqs = star_of_cherries(n_cherries)
        rng = np.random.default_rng(3)
        picks = rng.integers(0, len(seqs), 400)
        reads = [seqs[i][int(a):int(a) + 30] for i, a in zip(picks, rng.integers(0, 25, 400))]
        bases = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
        offsets = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
        with engine.PlacementDb(flat, device=0) as db:
            assert db.info.max_nonleaf_arity == n_cherries and db.info.format == 1 and db.info.binary_tree == 0
        for f in (flat, drop_random_nodes(flat, 0.1, seed=5)):
            for kw in PARAM_SETS[:2]:
                got = _check(f, bases, offsets, kw)
        assert (got["status"] == _abi.IDENTITY_FOUND).sum() > 100


def test_option_corner_values():
    for s in (SynthDb(80, 300, 9, 4), SynthDb(80, 300, 9, 4, collapse_prob=0.4)):
        bases, offsets, _ = s.reads(600, 100, frac_random=0.05, err=0.03)
        for kw in ODD_PARAM_SETS:
            _check(s.flat, bases, offsets, kw)


def test_concurrent_host_calls_are_reentrant():
    """cls_place_batch from several threads on one handle (own stream per call, scratch slots handed out under a lock):
    batches of different sizes, so that slots are added and recycled while other calls are still launching."""
    import threading
    s = SynthDb(200, 500, 10, 4)
    bases, offsets, _ = s.reads(6000, 150)
    want = op.OraclePort(s.flat).place_batch(bases, offsets, threads=8)
    sizes = [6000, 700, 4500, 90, 5200, 2000, 6000, 300]
    errors = []
    with engine.PlacementDb(s.flat, device=0) as db:
        def work(i):
            try:
                for it in range(6):
                    n = sizes[(i + it) % len(sizes)]
                    got = db.place_batch(bases[: 150 * n], offsets[: n + 1])
                    if len(records_equal(got, want[:n])) != 0:
                        errors.append((i, it, n))
            except Exception as e:  # pragma: no cover
                errors.append(e)
        th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
        [t.start() for t in th]
        [t.join() for t in th]
    assert not errors, errors


def test_host_entry_pipelines_large_batches():
    """More than 256k reads through cls_place_batch: chunks alternate between two call slots (copies overlap kernels);
    same records as one device-resident call, and as the oracle on a sample."""
    s = SynthDb(400, 1200, 12, 4)
    n = 700_000
    bases, offsets, _ = s.reads(n, 100)
    with engine.PlacementDb(s.flat, device=0) as db:
        host, hst = db.place_batch(bases, offsets, want_stats=True)
        dev, dst = _device_place(db, bases, offsets)
        host2 = db.place_batch(bases, offsets)
    assert len(records_equal(host, dev)) == 0 and len(stats_equal(hst, dst)) == 0 and len(records_equal(host2, dev)) == 0
    pick = np.random.default_rng(1).choice(n, 20000, replace=False)
    pick.sort()
    sb = bases.reshape(n, 100)[pick].reshape(-1)
    so = np.arange(len(pick) + 1, dtype=np.uint64) * 100
    want = op.OraclePort(s.flat).place_batch(sb, so, threads=16)
    assert len(records_equal(host[pick], want)) == 0


def test_index_format_selection():
    """Which device layout / kernels an index gets."""
    cases = [
        (SynthDb(60, 300, 8, 4), None, (1, 1, 2)),                      # closed sets, binary tree, k <= 15, both strands indexed: fast path, one lookup per window
        (SynthDb(60, 300, 8, 4), -1.0, (1, 1, 1)),                      # the same, not strand-symmetric: fast path, both strands looked up
        (SynthDb(60, 300, 17, 4), None, (1, 1, 0)),                     # k > 15: split records, murmur probe path
        (SynthDb(60, 300, 8, 4, collapse_prob=0.4), None, (1, 0, 2)),   # polytomies: fast path with the per-group child walk
        (SynthDb(60, 300, 17, 4, collapse_prob=0.4), None, (1, 0, 0)),  # polytomies, k > 15: split records, wave-wide child walk
        (SynthDb(60, 300, 8, 4), 0.2, (0, 1, 0)),                       # a node set that is not closed: sorted lists
    ]
    for s, drop, want in cases:
        flat = s.flat if not drop else truncate_random_sets(s.flat, 0.1, seed=2) if drop < 0 else drop_random_nodes(s.flat, drop, seed=2)
        with engine.PlacementDb(flat, device=0) as db:
            assert (db.info.format, db.info.binary_tree, db.info.direct_table) == want


def test_empty_batch_and_offsets_base():
    s = SynthDb(50, 300, 8, 4)
    with engine.PlacementDb(s.flat, device=0) as db:
        out = db.place_batch(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
        assert len(out) == 0
        bases, offsets, _ = s.reads(64, 100)
        a = db.place_batch(bases, offsets)
        pad = np.frombuffer(b"TTTTTTT", dtype=np.uint8)
        b = db.place_batch(np.concatenate([pad, bases]), offsets + np.uint64(len(pad)))
        assert len(records_equal(a, b)) == 0


def test_c2_shape_parity_sample():
    """BASELINE config C2 shape (1k leaves, k=8) on a 20k-read sample."""
    s = SynthDb(1000, 1500, 8, 4)
    bases, offsets, _ = s.reads(20000, 150)
    _check(s.flat, bases, offsets, {}, threads=16)


@pytest.mark.parametrize("case", [(80, 300, 10, 4, 0.0, 0), (100, 300, 12, 4, 0.4, 0), (150, 400, 15, 4, 0.0, 1), (60, 300, 17, 4, 0.2, 0)])
def test_leaves_only_index_input(case):
    """cls_db_desc v2, CLS_SETS_LEAVES: the index lists only the LEAF members of every node set (the union of their
    root paths is implied, build_database/mod.rs:160-169).  Same placements and counters as the explicit index, which
    is what the oracle -- like the reference -- is fed."""
    nl, rl, k, m, cp, deep = case
    s = SynthDb(nl, rl, k, m, collapse_prob=cp, deep=deep)
    t = SynthDb(nl, rl, k, m, collapse_prob=cp, deep=deep, tips_only=True)
    assert t.flat.leaves_only and t.flat.node_ids.size < s.flat.node_ids.size
    bases, offsets, _ = s.reads(1500, min(rl, 150), frac_random=0.05, err=0.02)
    for kw in PARAM_SETS[:3]:
        want, wst = op.OraclePort(s.flat).place_batch(bases, offsets, op.make_params(**kw), threads=8, want_stats=True)
        for flat in (t.flat, s.flat.to_leaves_only()):
            with engine.PlacementDb(flat, device=0) as db:
                got, gst = db.place_batch(bases, offsets, engine.make_params(**kw), want_stats=True)
            assert len(records_equal(got, want)) == 0 and len(stats_equal(gst, wst)) == 0
    with engine.PlacementDb(t.flat, device=0) as a, engine.PlacementDb(s.flat, device=0) as b:
        assert (a.info.format, a.info.n_tip_sets, a.info.n_kmers, a.info.direct_table) == (b.info.format, b.info.n_tip_sets, b.info.n_kmers, b.info.direct_table)


def test_long_reads_lds_tiled_kernel_and_its_spill_path():
    """10 kb reads on a binary tree with a direct table (the shape of BASELINE config 5) take the LDS-tiled kernel:
    strand-symmetric and not, both entries.  With the code set forced into ONE pass a 10 kb read overflows it and is
    handed to the workspace kernel through the spill list: same records."""
    s = SynthDb(200, 11000, 15, 4, deep=1)
    rng = np.random.default_rng(44)
    bases, offsets = ragged_reads(rng, s, 60, 4200, 10800, lower_frac=0.05)
    want = {}
    for flat, tag in ((s.flat, "sym"), (truncate_random_sets(s.flat, 0.02, seed=3), "asym")):
        for kw in (dict(), dict(remove_intersection=True), dict(max_iterations=7)):
            got = _check(flat, bases, offsets, kw, threads=16)
            want[(tag, tuple(kw))] = got
        assert (got["status"] != _abi.ERR_READ_TOO_LONG).all()
    for pass_codes, set_words in ((1 << 30, 4096), (1024, 4096)):  # one pass into a small set: long reads overflow it and spill; passes over hash partitions
        engine.set_tuning("tile_pass_codes", pass_codes)
        engine.set_tuning("tile_set_words", set_words)
        try:
            got = _check(s.flat, bases, offsets, {}, threads=16)
            assert len(records_equal(got, want[("sym", ())])) == 0
        finally:
            engine.set_tuning("tile_pass_codes", 0)
            engine.set_tuning("tile_set_words", 0)
    engine.set_tuning("no_tile", 1)
    try:
        got = _check(s.flat, bases, offsets, {}, threads=16)
        assert len(records_equal(got, want[("sym", ())])) == 0
    finally:
        engine.set_tuning("no_tile", 0)


@pytest.mark.parametrize("k,deep", [(16, 0), (21, 1), (35, 0)])
def test_long_reads_lds_tiled_kernel_hashed_front(k, deep):
    """k > 15 (no direct table; the reference's documented k is 35): the LDS-tiled kernel hashes every k-mer of both
    strands (MurmurHash3, kmers_map.rs:157-159), probes the table and applies the minimizer-bucket filter
    (kmers_map.rs:273-311).  Same records with the kernel switched off (workspace kernel) and against the oracle."""
    s = SynthDb(150, 11000, k, 4, deep=deep)
    rng = np.random.default_rng(45 + k)
    bases, offsets = ragged_reads(rng, s, 48, 4200, 10800, lower_frac=0.05)
    with engine.PlacementDb(s.flat, device=0) as db:
        assert (db.info.format, db.info.binary_tree, db.info.direct_table) == (1, 1, 0)
        db.set_max_read_len(10800)
        assert db.kernel_name().startswith("place_tile_kernel<512, 2,") or db.kernel_name().startswith("place_tile_kernel<1024, 2,")
    for flat in (s.flat, truncate_random_sets(s.flat, 0.02, seed=5)):
        for kw in (dict(), dict(remove_intersection=True), dict(max_iterations=6)):
            got = _check(flat, bases, offsets, kw, threads=16)
        assert (got["status"] != _abi.ERR_READ_TOO_LONG).all()
    engine.set_tuning("no_tile", 1)
    try:
        off = _check(flat, bases, offsets, kw, threads=16)
        assert len(records_equal(off, got)) == 0
    finally:
        engine.set_tuning("no_tile", 0)
    for pass_codes, set_words in ((1 << 30, 4096), (1024, 4096)):  # the spill path and the passes over hash partitions (keys: table slots)
        engine.set_tuning("tile_pass_codes", pass_codes)
        engine.set_tuning("tile_set_words", set_words)
        try:
            assert len(records_equal(_check(flat, bases, offsets, kw, threads=16), got)) == 0
        finally:
            engine.set_tuning("tile_pass_codes", 0)
            engine.set_tuning("tile_set_words", 0)


@pytest.mark.parametrize("k,collapse,trunc", [(12, 0.0, False), (10, 0.4, False), (11, 0.0, True)])
def test_slim_and_denormalised_direct_tables_agree(k, collapse, trunc):
    """For k <= 12 the wave-per-read kernels read a denormalised 16-byte direct table (set record inside the entry);
    with the knob off they go through the 4-byte table + set records like every larger k.  Both against the oracle,
    narrow and wide class, locality-ordered batch."""
    s = SynthDb(300, 900, k, 4, collapse_prob=collapse)
    flat = truncate_random_sets(s.flat, 0.05, seed=6) if trunc else s.flat
    rng = np.random.default_rng(23)
    bases, offsets = ragged_reads(rng, s, 5000, 60, 480, lower_frac=0.02)
    with engine.PlacementDb(flat, device=0) as db:
        assert db.info.fat_direct_table == 1
    for kw in (dict(), dict(remove_intersection=True)):
        _check(flat, bases, offsets, kw, threads=16)
    engine.set_tuning("no_fat_direct", 1)
    try:
        with engine.PlacementDb(flat, device=0) as db:
            assert db.info.fat_direct_table == 0
        for kw in (dict(), dict(remove_intersection=True)):
            _check(flat, bases, offsets, kw, threads=16)
    finally:
        engine.set_tuning("no_fat_direct", 0)


@pytest.mark.parametrize("k,deep,trunc,collapse", [(12, 0, False, 0.0), (12, 2, False, 0.0), (11, 0, True, 0.0), (14, 1, False, 0.0), (17, 0, False, 0.0),
                                                   (12, 0, False, 0.3), (10, 0, True, 0.5), (35, 0, False, 0.3), (12, 1, False, 0.8)])
def test_mask_halves_on_and_off_agree(k, deep, trunc, collapse):
    """The wave-per-read kernels read a second copy of the split records in which parts that span at most 32 rows are bit
    masks (and, for k <= 12, narrow sets start as bits straight from the fat direct table); with the knob off they walk
    the plain records.  Both against the oracle: fat / slim direct table and hashed front, balanced and ladder-like trees,
    binary trees and polytomies (small and large arities), node sets cut back so that some tips are internal clades."""
    s = SynthDb(400, 900, k, 4, collapse_prob=collapse, deep=deep, seed_tree=31, seed_refseq=32)
    flat = truncate_random_sets(s.flat, 0.05, seed=8) if trunc else s.flat
    rng = np.random.default_rng(29)
    bases, offsets = ragged_reads(rng, s, 6000, 60, 480, lower_frac=0.02)
    with engine.PlacementDb(flat, device=0) as db:
        assert db.info.binary_tree == (1 if collapse == 0.0 else 0)
        with_masks = db.info.hbm_bytes
    for kw in (dict(), dict(remove_intersection=True)):
        _check(flat, bases, offsets, kw, threads=16)
    if not trunc:  # the same index handed over as its leaves-only view (the node sets of the internal clades are implied)
        _check(flat.to_leaves_only(), bases, offsets, {}, threads=16)
    engine.set_tuning("no_mask_halves", 1)
    try:
        with engine.PlacementDb(flat, device=0) as db:
            assert db.info.hbm_bytes < with_masks
        for kw in (dict(), dict(remove_intersection=True)):
            _check(flat, bases, offsets, kw, threads=16)
    finally:
        engine.set_tuning("no_mask_halves", 0)


def test_g35_the_references_documented_workload_shape():
    """bench.py --config G35 at test size: the one workload the reference documents (docs/book/06-telemetry-and-benchmark.md:67-70,
    fd7/logging.jsonl:2,4) -- ~1.9 kb queries, a ~590-node support-collapsed tree, k=35, m=4 -- through the
    workgroup-per-read kernel, both remove_intersection values, against the oracle."""
    cfg = CONFIGS["G35"]
    s = SynthDb(cfg["n_leaves"], cfg["ref_len"], cfg["k_size"], cfg["m_size"], collapse_prob=cfg["collapse_prob"])
    bases, offsets, _ = s.reads(600, cfg["read_len"], seed=3)
    for kw in (dict(), dict(remove_intersection=True)):
        got = _check(s.flat, bases, offsets, kw, threads=16)
    assert (got["status"] == _abi.IDENTITY_FOUND).sum() + (got["status"] == _abi.MAX_RESOLUTION).sum() > 500
