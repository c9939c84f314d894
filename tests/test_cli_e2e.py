"""BASELINE config 1 (plumbing): the bundled reference-built database + its query FASTA through the
`cls place` look-alike (csrc/cls_place_cli.cpp -> cls_place_sequences -> HIP kernels), YAML and JSONL."""
import json
import os
import subprocess

import numpy as np
import pytest
import yaml

from classeq2_amd import _abi, engine
from classeq2_amd.flatdb import FlatDb
from tests.test_golden import _load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "classeq2_amd", "csrc", "cls-place")
KINDS = {0: "ROOT", 1: "NODE", 2: "LEAF"}


def write_db_json(flat: FlatDb, path: str):
    """FlatDb -> the reference's database JSON shape (serde_json of `Tree`, camelCase keys)."""
    n = flat.nodes

    def clade(r):
        d = {"id": int(n[r]["id"]), "parent": None if int(n[r]["parent"]) == _abi.NO_PARENT else int(n[r]["parent"]), "kind": KINDS[int(n[r]["kind"])]}
        if n[r]["kind"] == 2:
            d["name"] = f"leaf_{int(n[r]['id'])}"
        d["length"] = 0.01
        if n[r]["has_children"]:
            d["children"] = [clade(int(n[r]["first_child"]) + i) for i in range(int(n[r]["n_children"]))]
        return d

    km = {}
    for b in range(len(flat.bucket_key)):
        km[str(int(flat.bucket_key[b]))] = {
            str(int(flat.kmer_hash[j])): [int(x) for x in flat.node_ids[int(flat.kmer_node_off[j]):int(flat.kmer_node_off[j + 1])]]
            for j in range(int(flat.bucket_kmer_off[b]), int(flat.bucket_kmer_off[b + 1]))
        }
    doc = {"id": "00000000-0000-0000-0000-000000000000", "name": "golden", "minBranchSupport": 70.0, "inMemorySize": None,
           "root": clade(0), "kmersMap": {"kSize": flat.k_size, "mSize": flat.m_size, "map": km}}
    json.dump(doc, open(path, "w"))


def write_fasta(path, headers, bases, offsets, width=60):
    raw = bytes(bases)
    with open(path, "w") as f:
        for i, h in enumerate(headers):
            s = raw[int(offsets[i]):int(offsets[i + 1])].decode()
            f.write(f">{h}\n")
            for p in range(0, len(s), width):
                f.write(s[p:p + width] + "\n")


def test_host_loader_round_trip(tmp_path):
    """CPU: JSON database -> cls_tree -> flat view == the arrays it was written from."""
    flat, *_ = _load()
    write_db_json(flat, str(tmp_path / "db.json"))
    t = engine.Tree(str(tmp_path / "db.json"))
    got = t.flat()
    for f in ("id", "parent", "first_child", "n_children", "kind", "has_children"):
        assert (got.nodes[f] == flat.nodes[f]).all(), f
    assert (got.k_size, got.m_size) == (flat.k_size, flat.m_size)
    for name in ("bucket_key", "bucket_kmer_off", "kmer_hash", "kmer_node_off", "node_ids"):
        assert (getattr(got, name) == getattr(flat, name)).all(), name
    engine.validate(got)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["yaml", "jsonl"])
def test_cli_places_bundled_fixture(tmp_path, fmt):
    flat, bases, offsets, params, expected = _load()
    headers = [str(h) for h in np.load(os.path.join(ROOT, "tests", "golden", "colletotrichum_k12.npz"))["headers"]]
    db, fa, out = str(tmp_path / "db.json"), str(tmp_path / "q.fasta"), str(tmp_path / "res" / "result.out")
    write_db_json(flat, db)
    if fmt == "jsonl":  # the form `cls build-db` writes: zstd-compressed YAML
        engine.Tree(db).save(str(tmp_path / "db"), engine.DB_FORMAT_ZSTD)
        db = str(tmp_path / "db.cls")
    write_fasta(fa, headers, bases, offsets)
    cmd = [CLI, fa, "-d", db, "-o", out, "--out-format", fmt]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    res = str(tmp_path / "res" / f"result.{fmt}")
    text = open(res).read()
    docs = list(yaml.safe_load_all(text)) if fmt == "yaml" else [json.loads(l) for l in text.splitlines()]
    want = expected[0]
    assert [d["query"] for d in docs] == headers  # input order
    for d, w in zip(docs, want):
        st = int(w["status"])
        if st == _abi.IDENTITY_FOUND:
            assert d["code"] == "IdentityFound" and d["placement"]["clade"]["id"] == int(w["clade_id"])
            assert (d["placement"]["one"], d["placement"]["rest"]) == (int(w["one"]), int(w["rest"]))
        elif st == _abi.MAX_RESOLUTION:
            assert d["code"] == "MaxResolutionReached: LCA Accepted" and d["placement"] == int(w["clade_id"])
        else:
            assert d["code"].startswith("Unclassifiable: ") and "placement" not in d
    assert open(str(tmp_path / "res" / "result.error")).read() == ""
    # overwrite policy (mod.rs:91-106)
    r2 = subprocess.run(cmd, capture_output=True, text=True)
    assert r2.returncode != 0 and "Could not overwrite existing file" in r2.stderr
    r3 = subprocess.run(cmd + ["-f", "-r", "-i", "3"], capture_output=True, text=True)
    assert r3.returncode == 0, r3.stderr
    # -i 3: reads that needed more than 3 levels go to the error file instead
    n_err = open(str(tmp_path / "res" / "result.error")).read().count("The maximum number of iterations has been reached.")
    assert n_err == int((expected[3]["status"] == _abi.ERR_MAX_ITER).sum()) and n_err > 0
