"""BASELINE config 5 scaled down to what the oracle can check: a deep caterpillar-biased tree (5 000 leaves, depth
> 500, capped at 900 like the full config), 10 500 bp references, k = 15, 10 kb reads.  The index reaches the engine
in the leaves-only form (cls_db_desc v2, CLS_SETS_LEAVES): its explicit node sets -- tens of billions of ids here,
terabytes at full size -- are never materialised.  The oracle keeps the same leaves-only sets lazily (a clade is a
member iff a listed leaf lies below it), a mode tests/test_oracle.py holds to the explicit-set oracle on every shape
small enough to expand.  Reference path: place_sequence.rs:279-601 on a tree `max_iterations` (1000) deep."""
import numpy as np
import pytest

from classeq2_amd import _abi, engine
from classeq2_amd.synth import SynthDb
from oracle import oracle_port as op
from tests.helpers import describe, device_place, records_equal, stats_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c5s():
    s = SynthDb(5000, 10500, 15, 4, deep=2, max_depth=900, tips_only=True)  # deep=2: a pure ladder down to the depth cap
    oracle = op.OraclePort(s.flat)
    db = engine.PlacementDb(s.flat, device=0)
    yield s, oracle, db
    db.close()
    oracle.close()


def test_scaled_c5_long_reads_against_the_oracle(c5s):
    s, oracle, db = c5s
    assert s.max_depth > 500 and db.info.max_depth == s.max_depth
    assert (db.info.format, db.info.binary_tree, db.info.direct_table) == (1, 1, 2)
    bases, offsets, _ = s.reads(96, 10000, seed=3)
    for kw in (dict(), dict(remove_intersection=True)):
        got, gst = db.place_batch(bases, offsets, engine.make_params(**kw), want_stats=True)
        want, wst = oracle.place_batch(bases, offsets, op.make_params(**kw), threads=16, want_stats=True)
        bad = records_equal(got, want)
        assert len(bad) == 0, f"{kw}: {len(bad)} records differ, first {bad[0]}: got {describe(got[bad[0]])} want {describe(want[bad[0]])}"
        assert len(stats_equal(gst, wst)) == 0
    assert got["levels"].max() > 300 and (got["status"] == _abi.IDENTITY_FOUND).sum() > 60
    # the device-buffer entry, provisioned for 10 kb reads
    db.set_max_read_len(10000)
    dev, dst = device_place(db, bases, offsets, engine.make_params(remove_intersection=True))
    assert len(records_equal(dev, want)) == 0 and len(stats_equal(dst, wst)) == 0
    assert (dst["index_bytes"] > 0).all()


def test_scaled_c5_short_reads_and_the_iteration_cap(c5s):
    """150 bp reads on the same deep index (wave-per-read fast path, locality order), and a lowered iteration cap."""
    s, oracle, db = c5s
    bases, offsets, _ = s.reads(6000, 150, seed=5)
    for kw in (dict(), dict(max_iterations=120)):
        got, gst = db.place_batch(bases, offsets, engine.make_params(**kw), want_stats=True)
        want, wst = oracle.place_batch(bases, offsets, op.make_params(**kw), threads=16, want_stats=True)
        assert len(records_equal(got, want)) == 0 and len(stats_equal(gst, wst)) == 0
    assert (got["status"] == _abi.ERR_MAX_ITER).sum() > 100
