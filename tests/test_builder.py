"""Index builder (`map_kmers_to_tree`) against the one database a reference build wrote."""
import json
import os

import numpy as np
import pytest

from classeq2_amd import engine
from oracle import oracle_literal as lit

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "builder_colletotrichum.json")


@pytest.fixture(scope="module")
def gold(tmp_path_factory):
    g = json.load(open(GOLD))
    p = tmp_path_factory.mktemp("b") / "tree.json"
    p.write_text(g["tree_json"])
    g["tree_path"] = str(p)
    return g


def _as_dict(flat):
    out = {}
    for b in range(len(flat.bucket_key)):
        for j in range(int(flat.bucket_kmer_off[b]), int(flat.bucket_kmer_off[b + 1])):
            ids = flat.node_ids[int(flat.kmer_node_off[j]):int(flat.kmer_node_off[j + 1])]
            out[int(flat.kmer_hash[j])] = (int(flat.bucket_key[b]), sorted(int(x) for x in ids))
    return out


def test_builder_reproduces_reference_built_database(gold):
    t = engine.Tree(gold["tree_path"])
    t.build_kmers_map(gold["msa_fasta"].encode(), gold["k_size"], gold["m_size"], reference_header_shift=True, forward_only=True)
    got = _as_dict(t.flat())
    want = {int(h): (int(v["bucket"]), v["nodes"]) for h, v in gold["expected"].items()}
    assert got.keys() == want.keys()
    assert all(got[h] == want[h] for h in want)
    engine.validate(t.flat())


def test_builder_modes_against_literal(gold):
    """Today's behaviour (forward + reverse complement) and the intended one (no header skew) against the
    literal restatement of build_database/mod.rs."""
    recs = lit.sequence_content_by_channel(gold["msa_fasta"])
    root = json.loads(gold["tree_json"])["root"]
    paths = {}

    def walk(n, path):
        path = path + [n["id"]]
        if n["kind"] == "LEAF":
            paths.setdefault(n["name"], path)
        for c in n.get("children") or []:
            walk(c, path)

    walk(root, [])
    for shift in (True, False):
        km = lit.KmersMap(gold["k_size"], gold["m_size"])
        pairs = [(recs[i + 1][0], recs[i][1]) for i in range(len(recs) - 1)] if shift else [(h, s) for h, s in recs]
        for header, seq in pairs:
            for kmer, h in km.build_kmer_from_string(seq):
                km.insert_or_append_kmer_hash(kmer, h, paths[header])
        want = {h: (key, sorted(nodes)) for key, bucket in km.map.items() for h, nodes in bucket.items()}
        t = engine.Tree(gold["tree_path"])
        t.build_kmers_map(gold["msa_fasta"].encode(), gold["k_size"], gold["m_size"], reference_header_shift=shift)
        got = _as_dict(t.flat())
        assert got == want, shift


def test_builder_rejects_unknown_header(gold):
    t = engine.Tree(gold["tree_path"])
    with pytest.raises(engine.ClsError, match="does not match any tree leaf"):
        t.build_kmers_map(b">nobody\nACGTACGTACGTACGT\n>nobody2\nACGTACGTACGTAAAA\n", 12, 4)


@pytest.mark.gpu
def test_built_index_places_its_own_sequences(gold):
    """End to end: build on the host, upload, place the aligned sequences themselves."""
    from oracle import oracle_port as op
    from tests.helpers import records_equal
    t = engine.Tree(gold["tree_path"])
    t.build_kmers_map(gold["msa_fasta"].encode(), gold["k_size"], gold["m_size"])
    flat = t.flat()
    headers, bases, off, _ = engine.fasta_parse(gold["msa_fasta"].encode())
    with engine.PlacementDb(flat, device=0) as db:
        got = db.place_batch(bases, off)
    want = op.OraclePort(flat).place_batch(bases, off, threads=8)
    assert len(records_equal(got, want)) == 0
