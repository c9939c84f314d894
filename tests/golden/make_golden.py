#!/usr/bin/env python3
"""Generate tests/golden/colletotrichum_k12.npz from the reference's own fixtures.

Run HERE only (needs /root/reference, which does not travel to the GPU box):
    python tests/golden/make_golden.py

Inputs (data files of the reference's test suite, read as data):
  core/src/tests/data/colletotrichum-acutatom-complex/outputs/Colletotrichum_acutatum_gapdh-PhyML.yaml
      a database written by an older build of `cls build-db` (schema
      `kmersMap.map: {KMER_STRING: [node ids]}`, kSize 12): the only index in the
      reference tree that a reference binary produced.
  .../inputs/Colletotrichum_acutatum_gapdh_mafft.fasta      (what it was built from)
  .../inputs/Colletotrichum_acutatum_gapdh_gapsfree.fasta   (used as the 171 queries)

What is pinned by reference OUTPUT (asserted below, so the npz cannot be made
from a restatement that disagrees):
  * every k-mer's node set == union of root->leaf id paths of the leaves it was
    indexed under (build_database/mod.rs:139-169, clade.rs:127-156);
  * the header off-by-one of build_database/mod.rs:93-116 (record i's k-mers are
    filed under record i+1's leaf; the last record is never indexed) -- the
    fixture reproduces it exactly (forward k-mers only in that older build).
What is NOT pinned by reference output: the placement records stored here.  They
are produced by oracle/oracle_literal.py (checked against oracle/cls_oracle.c)
on this database -- a regression vector for the restatement, "parity unpinned"
in the sense of DESIGN.md.

The k-mer strings are re-keyed to today's schema: hash = murmur3_x64_128(kmer,0).0,
bucket = hash of the first mSize=4 characters (kmers_map.rs:125-159).
"""
import collections
import os
import sys

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from classeq2_amd.flatdb import FlatDb  # noqa: E402
from oracle import oracle_literal as lit  # noqa: E402
from oracle import oracle_port as op  # noqa: E402

REF = "/root/reference/core/src/tests/data/colletotrichum-acutatom-complex"
M_SIZE = 4
PARAM_SETS = [dict(), dict(remove_intersection=True), dict(min_match_coverage=1.0), dict(max_iterations=3)]


def read_fasta(path):
    return lit.sequence_content_by_channel(open(path).read())


def main():
    db = yaml.safe_load(open(f"{REF}/outputs/Colletotrichum_acutatum_gapdh-PhyML.yaml"))
    k = int(db["kmersMap"]["kSize"])
    strings = db["kmersMap"]["map"]

    # parent ids are absent from the old schema: restore them like Tree::fix_parent_ids (tree.rs:225-242)
    def fix(n, parent):
        n["parent"] = parent
        for c in n.get("children") or []:
            fix(c, n["id"])

    fix(db["root"], None)
    paths = {}

    def walk(n, path):
        path = path + [n["id"]]
        if n["kind"] == "LEAF":
            paths[n["name"]] = path
        for c in n.get("children") or []:
            walk(c, path)

    walk(db["root"], [])
    # --- the fixture pins the builder semantics (see the module docstring) ---
    msa = read_fasta(f"{REF}/inputs/Colletotrichum_acutatum_gapdh_mafft.fasta")
    rebuilt = collections.defaultdict(set)
    for i in range(len(msa) - 1):
        seq, target = msa[i][1], msa[i + 1][0]
        for p in range(len(seq) - k + 1):
            rebuilt[seq[p:p + k]].update(paths[target])
    assert set(rebuilt) == set(strings), "k-mer key sets differ from the reference fixture"
    assert all(set(strings[s]) == rebuilt[s] for s in strings), "node sets differ from the reference fixture"

    km = {}
    for s, ids in strings.items():
        km.setdefault(lit.build_minimizer_from_string(s, M_SIZE), {})[lit.hash_kmer(s)] = ids
    flat = FlatDb.from_nested(db["root"], k, M_SIZE, km)

    queries = read_fasta(f"{REF}/inputs/Colletotrichum_acutatum_gapdh_gapsfree.fasta")
    headers = [h for h, _ in queries]
    bases = np.frombuffer("".join(s for _, s in queries).encode(), dtype=np.uint8)
    offsets = np.concatenate([[0], np.cumsum([len(s) for _, s in queries])]).astype(np.uint64)

    tree = op.flat_to_literal(flat)
    port = op.OraclePort(flat)
    out = {}
    for i, kw in enumerate(PARAM_SETS):
        want = op.literal_place_batch(tree, bases, offsets, headers=headers, **kw)
        got = port.place_batch(bases, offsets, op.make_params(**kw))
        for f in ("status", "one", "rest", "levels", "clade_id"):
            assert (want[f] == got[f]).all(), (kw, f)
        out[f"expected_{i}"] = want
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "colletotrichum_k12.npz"),
        nodes=flat.nodes, k_size=k, m_size=M_SIZE, bucket_key=flat.bucket_key, bucket_kmer_off=flat.bucket_kmer_off,
        kmer_hash=flat.kmer_hash, kmer_node_off=flat.kmer_node_off, node_ids=flat.node_ids,
        bases=bases, offsets=offsets, headers=np.array(headers), param_sets=np.array([repr(p) for p in PARAM_SETS]), **out,
    )
    st = collections.Counter(out["expected_0"]["status"].tolist())
    print("wrote colletotrichum_k12.npz:", len(flat.nodes), "nodes,", flat.n_kmers, "k-mers,", len(headers), "queries; statuses", dict(st))


if __name__ == "__main__":
    main()
