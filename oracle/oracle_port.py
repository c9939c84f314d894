"""ctypes handle on oracle/_build/libclsoracle.so (the flat C port).

TEST INFRASTRUCTURE ONLY -- see oracle/cls_oracle.c.  Also holds the
FlatDb -> literal-oracle object conversion used by the tests.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from classeq2_amd import _abi
from classeq2_amd.flatdb import FlatDb

from . import oracle_literal as lit

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    """Compile the C port (gcc) if needed; returns the .so path."""
    so = os.path.join(_HERE, "_build", "libclsoracle.so")
    src = os.path.join(_HERE, "cls_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "cls_place.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "_build", "libclsoracle.so")
        if not os.path.exists(so):
            so = build()
        L = C.CDLL(so)
        L.cls_oracle_create.argtypes = [C.POINTER(_abi.DbDesc), C.POINTER(C.c_void_p)]
        L.cls_oracle_create.restype = C.c_int
        L.cls_oracle_destroy.argtypes = [C.c_void_p]
        L.cls_oracle_destroy.restype = None
        L.cls_oracle_set_reference_cost.argtypes = [C.c_void_p, C.c_int]
        L.cls_oracle_set_reference_cost.restype = None
        L.cls_oracle_place_batch.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(_abi.Params), C.c_int, C.c_void_p, C.c_void_p,
        ]
        L.cls_oracle_place_batch.restype = C.c_int
        L.cls_oracle_murmur3_h1.argtypes = [C.c_char_p, C.c_size_t]
        L.cls_oracle_murmur3_h1.restype = C.c_uint64
        _LIB = L
    return _LIB


def make_params(max_iterations=None, min_match_coverage=None, remove_intersection=None):
    """Option<i32>, Option<f64>, Option<bool> -> cls_params (None -> flag clear)."""
    p = _abi.Params()
    if max_iterations is not None:
        p.flags |= _abi.HAS_MAX_ITERATIONS
        p.max_iterations = max_iterations
    if min_match_coverage is not None:
        p.flags |= _abi.HAS_MIN_MATCH_COVERAGE
        p.min_match_coverage = min_match_coverage
    if remove_intersection is not None:
        p.flags |= _abi.HAS_REMOVE_INTERSECTION
        p.remove_intersection = 1 if remove_intersection else 0
    return p


class OraclePort:
    def __init__(self, flat: FlatDb):
        self._flat = flat
        self._h = C.c_void_p()
        d = flat.desc()
        rc = lib().cls_oracle_create(C.byref(d), C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"cls_oracle_create failed: {rc}")

    def close(self):
        if self._h:
            lib().cls_oracle_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_reference_cost(self, on: bool) -> None:
        """Also pay the reference's per-query index clone + bucket key-set rebuild (timing estimates only)."""
        lib().cls_oracle_set_reference_cost(self._h, 1 if on else 0)

    def place_batch(self, bases: np.ndarray, offsets: np.ndarray, params=None, threads: int = 1, want_stats=False):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        out = np.zeros(n, dtype=_abi.PLACEMENT_DTYPE)
        stats = np.zeros(n, dtype=_abi.STATS_DTYPE) if want_stats else None
        rc = lib().cls_oracle_place_batch(
            self._h, bases.ctypes.data, offsets.ctypes.data, n,
            C.byref(params) if params is not None else None, threads,
            out.ctypes.data, stats.ctypes.data if want_stats else None,
        )
        if rc != 0:
            raise RuntimeError(f"cls_oracle_place_batch failed: {rc}")
        return (out, stats) if want_stats else out


def flat_to_literal(flat: FlatDb) -> lit.Tree:
    """FlatDb -> oracle_literal.Tree (nested Clade objects + KmersMap dicts)."""
    kinds = {0: lit.ROOT, 1: lit.NODE, 2: lit.LEAF}
    nodes = flat.nodes
    clades = [None] * len(nodes)
    for r in range(len(nodes) - 1, -1, -1):  # children rows are always larger than the parent's
        n = nodes[r]
        if n["has_children"]:
            fc, nc = int(n["first_child"]), int(n["n_children"])
            children = [clades[fc + i] for i in range(nc)]
        else:
            children = None
        par = int(n["parent"])
        clades[r] = lit.Clade(
            id=int(n["id"]), parent=None if par == _abi.NO_PARENT else par, kind=kinds[int(n["kind"])], children=children
        )
    km = lit.KmersMap(flat.k_size, flat.m_size)
    for b in range(len(flat.bucket_key)):
        bucket = km.map.setdefault(int(flat.bucket_key[b]), {})
        for j in range(int(flat.bucket_kmer_off[b]), int(flat.bucket_kmer_off[b + 1])):
            ids = flat.node_ids[int(flat.kmer_node_off[j]) : int(flat.kmer_node_off[j + 1])]
            bucket.setdefault(int(flat.kmer_hash[j]), set()).update(int(x) for x in ids)
    return lit.Tree(root=clades[0], kmers_map=km)


def literal_place_batch(tree: lit.Tree, bases: np.ndarray, offsets: np.ndarray, max_iterations=None,
                        min_match_coverage=None, remove_intersection=None, headers=None) -> np.ndarray:
    n = len(offsets) - 1
    out = np.zeros(n, dtype=_abi.PLACEMENT_DTYPE)
    raw = bytes(np.ascontiguousarray(bases, dtype=np.uint8))
    for i in range(n):
        seq = raw[int(offsets[i]) : int(offsets[i + 1])].decode("latin-1")
        hdr = headers[i] if headers is not None else f"r{i}"
        st, one, rest, levels, clade = lit.place_to_record(hdr, seq, tree, max_iterations, min_match_coverage, remove_intersection)
        out[i] = (st, (0, 0, 0), one, rest, levels, clade)
    return out
