"""Tree / database files: Newick -> tree, load_database forms, `convert database` writers.
Pinned by data files of the reference's own tests (tests/golden/make_treeio_golden.py)."""
import json
import os

import numpy as np
import pytest

from classeq2_amd import engine

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _nodes_of(tree):
    """(id, parent, kind, name, support, length) in document order, from the JSON serialisation."""
    doc = json.loads(tree.dumps(engine.DB_FORMAT_JSON, only_tree=True))
    rows = []

    def walk(c):
        rows.append([c["id"], c["parent"], c["kind"], c.get("name"), c.get("support"), c.get("length")])
        for ch in c.get("children") or []:
            walk(ch)

    walk(doc)
    return rows


def _sanitize_rows(rows, min_support):
    """Tree::sanitize + fix_parent_ids restated on the flat node list (tree.rs:232-291)."""
    kids = {}
    for r in rows:
        kids.setdefault(r[1], []).append(r)

    def build(r):
        out = []
        for ch in kids.get(r[0], []):
            sub = build(ch)
            if ch[4] is None or ch[4] >= min_support or ch[2] == "LEAF":
                out.append((ch, sub))
            else:
                out.extend(sub)
        return out

    flat = []

    def emit(r, sub, parent):
        flat.append([r[0], parent, r[2], r[3], r[4], r[5]])
        for ch, s in sub:
            emit(ch, s, r[0])

    emit(rows[0], build(rows[0]), None)
    return flat


def test_newick_matches_the_reference_built_tree(tmp_path):
    gold = json.load(open(os.path.join(GOLD, "newick_colletotrichum.json")))
    p = tmp_path / gold["tree_name"]
    p.write_text(gold["newick"])
    t = engine.Tree.from_newick_file(str(p), min_branch_support=-2.0)  # the reference build kept every branch (supports down to -1)
    assert _nodes_of(t) == gold["nodes"]
    head = json.loads(t.dumps(engine.DB_FORMAT_JSON))
    assert head["id"] == gold["tree_id"] and head["name"] == gold["tree_name"]  # Uuid::new_v3(NAMESPACE_DNS, file name)
    assert head["minBranchSupport"] == -2.0 and head["kmersMap"] is None


@pytest.mark.parametrize("min_support", [0.0, 70.0, 95.0, 101.0])
def test_sanitize_collapses_low_support_branches(min_support):
    gold = json.load(open(os.path.join(GOLD, "newick_colletotrichum.json")))
    t = engine.Tree.from_newick(gold["newick"], "x.nwk", min_branch_support=min_support)
    got = _nodes_of(t)
    assert got == _sanitize_rows(gold["nodes"], min_support)
    assert sum(r[2] == "LEAF" for r in got) == 171
    if min_support > 100:
        assert all(r[1] == 0 for r in got[1:])  # a star: every leaf hangs off the root


def test_newick_syntax_cases():
    t = engine.Tree.from_newick("((A:0.1,'B b':0.2)95:0.3,[c](C,D)x:1e-3);", None, 50.0)
    rows = _nodes_of(t)
    assert [r[0] for r in rows] == [0, 1, 2, 3, 4, 5, 6]
    assert rows[1][2:] == ["NODE", None, 95.0, 0.3] and rows[3][3] == "B b"
    assert rows[4][4] is None and rows[4][5] == 0.001  # label "x" is no number: support None, so never collapsed
    assert rows[5][3] == "C" and rows[5][5] is None    # no branch length
    assert len(_nodes_of(engine.Tree.from_newick("(A,B,C);"))) == 4  # a trifurcating root is accepted (tree.rs:366-376 does so)
    with pytest.raises(engine.ClsError, match="not rooted"):
        engine.Tree.from_newick("A;")
    with pytest.raises(engine.ClsError):
        engine.Tree.from_newick("((A,B);")


def test_convert_only_tree_yaml_json_byte_identical(tmp_path):
    """The reference's two `--only-tree` exports of one model: reading either and writing the other reproduces
    the reference's file byte for byte (YAML reader, YAML writer, pretty-JSON writer, float formatting)."""
    y = open(os.path.join(GOLD, "bsub_gyrb_tree.cls.yaml"), "rb").read()
    j = open(os.path.join(GOLD, "bsub_gyrb_tree.cls.json"), "rb").read()
    ty = engine.Tree(os.path.join(GOLD, "bsub_gyrb_tree.cls.yaml"))
    tj = engine.Tree(os.path.join(GOLD, "bsub_gyrb_tree.cls.json"))
    assert ty.dumps(engine.DB_FORMAT_JSON) == j
    assert tj.dumps(engine.DB_FORMAT_YAML) == y
    assert ty.dumps(engine.DB_FORMAT_YAML) == y and tj.dumps(engine.DB_FORMAT_JSON) == j


def test_database_round_trip_all_formats(tmp_path):
    """build-db -> .cls (zstd YAML) / .cls.yaml / .cls.json -> load_database: the same index in every form."""
    gold = json.load(open(os.path.join(GOLD, "builder_colletotrichum.json")))
    nw = json.load(open(os.path.join(GOLD, "newick_colletotrichum.json")))
    t = engine.Tree.from_newick(nw["newick"], nw["tree_name"], min_branch_support=-2.0)
    t.build_kmers_map(gold["msa_fasta"].encode(), gold["k_size"], gold["m_size"], reference_header_shift=True, forward_only=True)
    want = t.flat()
    for fmt, ext in ((engine.DB_FORMAT_ZSTD, ".cls"), (engine.DB_FORMAT_YAML, ".cls.yaml"), (engine.DB_FORMAT_JSON, ".cls.json")):
        t.save(str(tmp_path / "db.whatever"), fmt)
        path = str(tmp_path / ("db" + ext))
        assert os.path.exists(path)
        got = engine.Tree(path).flat()
        for f in ("nodes", "bucket_key", "bucket_kmer_off", "kmer_hash", "kmer_node_off", "node_ids"):
            assert np.array_equal(getattr(got, f), getattr(want, f)), (ext, f)
        assert (got.k_size, got.m_size) == (want.k_size, want.m_size)
    raw = open(str(tmp_path / "db.cls"), "rb").read()
    assert raw[:4] == b"\x28\xb5\x2f\xfd" and len(raw) < os.path.getsize(str(tmp_path / "db.cls.yaml")) // 3
    head = open(str(tmp_path / "db.cls.yaml")).read(400)
    assert head.startswith("id: f0e71ef7-2d21-39a2-87de-fc2eeabbec18\nname: Colletotrichum_acutatum_gapdh-PhyML.nwk\nminBranchSupport: -2.0\ninMemorySize: null\nroot:\n  id: 0\n  parent: null\n  kind: ROOT\n")


def test_database_written_before_minimizers(tmp_path):
    """`kmersMap.map` keyed by the k-mer text (what the reference's Colletotrichum build on disk looks like)."""
    (tmp_path / "old.yaml").write_text(
        "id: 00000000-0000-0000-0000-000000000000\nname: t\ninMemorySize: 0.0 Mb\nroot:\n  id: 0\n  kind: ROOT\n  length: 0.0\n  children:\n"
        "  - id: 1\n    name: a\n    kind: LEAF\n    length: 1e-8\n  - id: 2\n    name: b\n    kind: LEAF\nkmersMap:\n  kSize: 4\n  map:\n"
        "    ACGT:\n    - 0\n    - 1\n    TTTT:\n    - 2\n    - 0\n")
    f = engine.Tree(str(tmp_path / "old.yaml")).flat()
    from oracle.oracle_literal import murmurhash3_x64_128

    def h1(b):
        return murmurhash3_x64_128(b, 0)[0]

    assert (f.k_size, f.m_size) == (4, 0) and list(f.bucket_key) == [0]
    assert list(f.kmer_hash) == [h1(b"ACGT"), h1(b"TTTT")] and list(f.node_ids) == [0, 1, 2, 0]
    assert list(f.nodes["kind"]) == [0, 2, 2] and list(f.nodes["id"]) == [0, 1, 2]
