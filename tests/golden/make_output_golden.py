#!/usr/bin/env python3
"""Generate tests/golden/output_stage_fd7.json from the reference's own work-dir fixture.

Run HERE only (needs /root/reference):  python tests/golden/make_output_golden.py

Inputs (data files the reference's tests/dev config hold):
  tests/data/public/019051d9-4c7a-7b2d-9dd1-66ef92236fd7/output/result.yaml  -- written by a reference build
  tests/models/bsub-gyrb-k35.cls.json                                        -- the tree (`convert database --only-tree -f json`)
  tests/models/bsub-gyrb-annotations.yaml                                    -- the annotations

result.yaml pins the OUTPUT STAGE (core/src/use_cases/place_sequences/mod.rs:170-248): key order, `code`
strings, the three `placement` shapes, the annotation filter/sort, serde_yaml's float / quoting / tag / block
scalar formatting.  The decisions inside it cannot be recomputed here (its k-mer index is a missing LFS
blob), so each record is reduced to (query, status, clade id, one, rest) and the test re-serialises those
through cls_serialize_results and compares BYTES with the reference's file.
"""
import json
import os

import yaml

REF = "/root/reference/tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "output_stage_fd7.json")


class L(yaml.SafeLoader):
    pass


def _tag(loader, suffix, node):
    return {suffix: loader.construct_scalar(node)}


L.add_multi_constructor("!", _tag)


def main():
    text = open(f"{REF}/data/public/019051d9-4c7a-7b2d-9dd1-66ef92236fd7/output/result.yaml").read()
    records = []
    for doc in yaml.load_all(text, Loader=L):
        code = doc["code"]
        if code == "IdentityFound":
            p = doc["placement"]
            records.append(dict(query=doc["query"], status=4, clade=p["clade"]["id"], one=p["one"], rest=p["rest"]))
        elif code.startswith("MaxResolutionReached"):
            records.append(dict(query=doc["query"], status=5, clade=doc["placement"], one=0, rest=0))
        else:
            raise SystemExit(f"unexpected code {code!r}")
    json.dump(
        dict(
            source="reference fixture tests/data/public/019051d9-4c7a-7b2d-9dd1-66ef92236fd7/output/result.yaml",
            tree_json=open(f"{REF}/models/bsub-gyrb-k35.cls.json").read(),
            annotations_yaml=open(f"{REF}/models/bsub-gyrb-annotations.yaml").read(),
            records=records,
            expected_yaml=text,
        ),
        open(OUT, "w"),
    )
    print("wrote", OUT, len(records), "records")


if __name__ == "__main__":
    main()
