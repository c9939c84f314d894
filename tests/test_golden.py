"""Golden vectors derived from the reference's own fixture files
(tests/golden/make_golden.py): the Colletotrichum database written by a
reference build + the 171 query sequences of the same fixture set."""
import ast
import os

import numpy as np
import pytest

from classeq2_amd import _abi, engine
from classeq2_amd.flatdb import FlatDb
from oracle import oracle_port as op
from tests.helpers import describe, records_equal

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "colletotrichum_k12.npz")


def _load():
    z = np.load(GOLD)
    flat = FlatDb(nodes=z["nodes"], k_size=int(z["k_size"]), m_size=int(z["m_size"]), bucket_key=z["bucket_key"],
                  bucket_kmer_off=z["bucket_kmer_off"], kmer_hash=z["kmer_hash"], kmer_node_off=z["kmer_node_off"],
                  node_ids=z["node_ids"])
    params = [ast.literal_eval(str(p)) for p in z["param_sets"]]
    expected = [z[f"expected_{i}"] for i in range(len(params))]
    return flat, z["bases"], z["offsets"], params, expected


def test_oracle_port_reproduces_golden():
    flat, bases, offsets, params, expected = _load()
    engine.validate(flat)
    port = op.OraclePort(flat)
    for kw, want in zip(params, expected):
        got = port.place_batch(bases, offsets, op.make_params(**kw), threads=2)
        bad = records_equal(got, want)
        assert len(bad) == 0, (kw, describe(got[bad[0]]), describe(want[bad[0]]))


def test_golden_sanity():
    flat, bases, offsets, params, expected = _load()
    assert len(flat.nodes) == 340 and flat.n_kmers == 2158 and flat.k_size == 12
    assert len(offsets) == 172
    # the reads are 206-260 bp: beyond the 320-k-mer kernel, inside the 1024-k-mer one
    lens = np.diff(offsets.astype(np.int64))
    assert lens.min() >= 200 and 2 * (lens.max() - 12 + 1) <= 1024
    st = np.bincount(expected[0]["status"], minlength=12)
    assert st[_abi.IDENTITY_FOUND] + st[_abi.MAX_RESOLUTION] + st[_abi.UNCLASSIFIABLE_LEVEL1] == 171


@pytest.mark.gpu
def test_hip_path_reproduces_golden():
    flat, bases, offsets, params, expected = _load()
    with engine.PlacementDb(flat, device=0) as db:
        for kw, want in zip(params, expected):
            got = db.place_batch(bases, offsets, engine.make_params(**kw))
            bad = records_equal(got, want)
            assert len(bad) == 0, (kw, describe(got[bad[0]]), describe(want[bad[0]]))
