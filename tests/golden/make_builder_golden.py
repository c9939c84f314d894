#!/usr/bin/env python3
"""Generate tests/golden/builder_colletotrichum.json from the reference's own build fixture.

Run HERE only (needs /root/reference):  python tests/golden/make_builder_golden.py

The reference holds exactly one database that one of its builds wrote:
  core/src/tests/data/colletotrichum-acutatom-complex/outputs/Colletotrichum_acutatum_gapdh-PhyML.yaml
(`kmersMap.map: {KMER_STRING: [node ids]}`, kSize 12), produced by `map_kmers_to_tree` from
  .../inputs/Colletotrichum_acutatum_gapdh_mafft.fasta  and the tree stored in the same YAML.
It pins the builder: node set = union of root->leaf id paths, filed with the record/header
skew of build_database/mod.rs:93-116.  (That older build indexed forward k-mers only and keyed the map by
the k-mer string; today's keys -- murmur3 hash, bucket = hash of the first mSize characters -- are derived
here with the hash function the reference's docs pin.)
"""
import json
import os
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_literal as lit  # noqa: E402

REF = "/root/reference/core/src/tests/data/colletotrichum-acutatom-complex"
M_SIZE = 4


def main():
    db = yaml.safe_load(open(f"{REF}/outputs/Colletotrichum_acutatum_gapdh-PhyML.yaml"))

    def fix(n, parent):  # the old schema stores no parent ids (Tree::fix_parent_ids, tree.rs:225-242)
        n["parent"] = parent
        for key in ("length", "support"):  # PyYAML reads serde_yaml's `1e-8` as a string (YAML 1.1 floats need a dot)
            if isinstance(n.get(key), str):
                n[key] = float(n[key])
        for c in n.get("children") or []:
            fix(c, n["id"])

    fix(db["root"], None)
    expected = {}
    for s, ids in db["kmersMap"]["map"].items():
        expected[str(lit.hash_kmer(s))] = {"bucket": str(lit.build_minimizer_from_string(s, M_SIZE)), "nodes": sorted(ids)}
    out = dict(
        source="reference fixture core/src/tests/data/colletotrichum-acutatom-complex (outputs/*.yaml + inputs/*_mafft.fasta)",
        k_size=int(db["kmersMap"]["kSize"]), m_size=M_SIZE,
        tree_json=json.dumps({"root": db["root"]}),
        msa_fasta=open(f"{REF}/inputs/Colletotrichum_acutatum_gapdh_mafft.fasta").read(),
        expected=expected,
    )
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "builder_colletotrichum.json"), "w"))
    print("wrote builder_colletotrichum.json:", len(expected), "k-mers")


if __name__ == "__main__":
    main()
