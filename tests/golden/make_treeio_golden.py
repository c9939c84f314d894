"""Golden data for tests/test_treeio.py, taken from data files the reference's own tests hold
(run in the build container, where /root/reference is mounted; the GPU box only sees the outputs).

  newick_colletotrichum.json   the reference's Colletotrichum Newick input + the tree a reference build wrote
                               for it (ids, kinds, names, supports, lengths in document order) and its header
  bsub_gyrb_tree.cls.yaml/.json  the reference's `convert database --only-tree` exports of the bsub-gyrB model,
                               one tree in both serialisations: each must convert into the other byte for byte
"""
import json
import os
import shutil

import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
col = os.path.join(REF, "core/src/tests/data/colletotrichum-acutatom-complex")
nwk = open(os.path.join(col, "inputs/Colletotrichum_acutatum_gapdh-PhyML.nwk")).read()
doc = yaml.safe_load(open(os.path.join(col, "outputs/Colletotrichum_acutatum_gapdh-PhyML.yaml")))
rows = []


def walk(c, parent):
    ln = c.get("length")
    rows.append([c["id"], parent, c["kind"], c.get("name"), c.get("support"), float(ln) if ln is not None else None])
    for ch in c.get("children") or []:
        walk(ch, c["id"])


walk(doc["root"], None)
json.dump({"source": "core/src/tests/data/colletotrichum-acutatom-complex (inputs/*.nwk, outputs/*.yaml)",
           "newick": nwk, "tree_id": doc["id"], "tree_name": doc["name"],
           "columns": ["id", "parent", "kind", "name", "support", "length"], "nodes": rows},
          open(os.path.join(HERE, "newick_colletotrichum.json"), "w"))
for ext in ("yaml", "json"):
    shutil.copy(os.path.join(REF, "tests/models/bsub-gyrb-k35.cls." + ext), os.path.join(HERE, "bsub_gyrb_tree.cls." + ext))
    os.chmod(os.path.join(HERE, "bsub_gyrb_tree.cls." + ext), 0o644)
print(len(rows), "nodes")
