"""End-to-end window of the reference (mod.rs:64-67, 264-267): query FASTA file -> result file, database load excluded.
usage: python tools/e2e.py [--config C3] [--reads N]   (GPU box)"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from classeq2_amd import _abi, engine  # noqa: E402
engine.tuning_from_env()  # CLS_* experiment knobs (the library never reads the environment on its own)
from classeq2_amd.synth import CONFIGS, SynthDb  # noqa: E402

KINDS = ["ROOT", "NODE", "LEAF"]


def tree_json(flat, path):
    n = flat.nodes
    sys.setrecursionlimit(100000)

    def clade(r):
        d = {"id": int(n[r]["id"]), "parent": None if int(n[r]["parent"]) == _abi.NO_PARENT else int(n[r]["parent"]), "kind": KINDS[int(n[r]["kind"])]}
        if n[r]["kind"] == 2:
            d["name"] = f"leaf_{int(n[r]['id'])}"
        else:
            d["support"] = 100.0
        d["length"] = 0.01
        if n[r]["has_children"]:
            d["children"] = [clade(int(n[r]["first_child"]) + i) for i in range(int(n[r]["n_children"]))]
        return d

    json.dump(clade(0), open(path, "w"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--reads", type=int, default=0)
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    n = args.reads or cfg["n_reads"]
    synth = SynthDb(cfg["n_leaves"], cfg["ref_len"], cfg["k_size"], cfg["m_size"], deep=cfg["deep"], max_depth=cfg["max_depth"])
    db = engine.PlacementDb(synth.flat, device=0)
    bases, offsets, _ = synth.reads(n, cfg["read_len"])
    tmp = tempfile.mkdtemp(prefix="cls_e2e_")
    tree_json(synth.flat, os.path.join(tmp, "tree.json"))
    tree = engine.Tree(os.path.join(tmp, "tree.json"))
    b = bases.reshape(n, cfg["read_len"])
    with open(os.path.join(tmp, "q.fasta"), "wb") as f:
        nl = np.full((n, 1), 10, dtype=np.uint8)
        hdr = np.char.add(np.char.add(">r", np.arange(n).astype(str)), "\n").astype("S")
        for lo in range(0, n, 100000):
            f.write(b"".join(h + bytes(row) + b"\n" for h, row in zip(hdr[lo:lo + 100000].tolist(), b[lo:lo + 100000])))
    out = {}
    for fmt, name in ((engine.FORMAT_JSONL, "jsonl"), (engine.FORMAT_YAML, "yaml")):
        engine.place_sequences(db, tree, os.path.join(tmp, "q.fasta"), os.path.join(tmp, "warm"), overwrite=True, fmt=fmt)
        t0 = time.time()
        got, sec = engine.place_sequences(db, tree, os.path.join(tmp, "q.fasta"), os.path.join(tmp, "res"), overwrite=True, fmt=fmt)
        wall = time.time() - t0
        size = os.path.getsize(os.path.join(tmp, "res." + name))
        out[name] = {"reads": got, "window_s": round(sec, 4), "wall_s": round(wall, 4), "reads_per_s": round(got / sec), "result_bytes": size}
    out["query_bytes"] = os.path.getsize(os.path.join(tmp, "q.fasta"))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
