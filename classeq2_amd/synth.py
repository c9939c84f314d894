"""Python handle on the synthetic workload generator (csrc/cls_synth.cpp).

BASELINE.json's configs, SURVEY.md 8(d): seeds tree=1 / refseq=2 / reads=3.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _abi
from .flatdb import FlatDb

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "csrc", "libclssynth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(path)
        lib.cls_synth_db_create.argtypes = [C.POINTER(_abi.SynthCfg), C.POINTER(C.c_void_p)]
        lib.cls_synth_db_create.restype = C.c_int
        lib.cls_synth_db_destroy.argtypes = [C.c_void_p]
        lib.cls_synth_db_destroy.restype = None
        lib.cls_synth_db_desc.argtypes = [C.c_void_p]
        lib.cls_synth_db_desc.restype = C.POINTER(_abi.DbDesc)
        lib.cls_synth_n_leaves.argtypes = [C.c_void_p]
        lib.cls_synth_n_leaves.restype = C.c_uint32
        lib.cls_synth_max_depth.argtypes = [C.c_void_p]
        lib.cls_synth_max_depth.restype = C.c_uint32
        lib.cls_synth_leaf_id.argtypes = [C.c_void_p, C.c_uint32]
        lib.cls_synth_leaf_id.restype = C.c_uint64
        lib.cls_synth_leaf_seq.argtypes = [C.c_void_p, C.c_uint32]
        lib.cls_synth_leaf_seq.restype = C.POINTER(C.c_char)
        lib.cls_synth_reads.argtypes = [
            C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, C.c_double,
            C.c_void_p, C.c_void_p, C.c_void_p,
        ]
        lib.cls_synth_reads.restype = C.c_int
        lib.cls_synth_last_error.restype = C.c_char_p
        _LIB = lib
    return _LIB


# BASELINE.json configs (index = position in `configs`)
CONFIGS = {
    "C2": dict(n_leaves=1000, ref_len=1500, k_size=8, m_size=4, n_reads=100_000, read_len=150, deep=0, max_depth=0),
    "C3": dict(n_leaves=10_000, ref_len=1500, k_size=12, m_size=4, n_reads=1_000_000, read_len=150, deep=0, max_depth=0),
    "C4": dict(n_leaves=10_000, ref_len=1500, k_size=12, m_size=4, n_reads=10_000_000, read_len=150, deep=0, max_depth=0),
    "C5": dict(n_leaves=50_000, ref_len=12_000, k_size=15, m_size=4, n_reads=1_000_000, read_len=10_000, deep=1, max_depth=900,
               tips_only=True),
    # the shapes the reference actually ships: `cls build-db -s 70` collapses low-support branches into polytomies
    # (tree.rs:248-285), and its default is k=35, m=4 (docs/book/02-build-db.md:109-129)
    "C3s12": dict(n_leaves=10_000, ref_len=1500, k_size=12, m_size=4, n_reads=1_000_000, read_len=150, deep=0, max_depth=0,
                  collapse_prob=0.3),
    # the one workload the reference documents (docs/book/06-telemetry-and-benchmark.md:67-70, fd7/logging.jsonl:2,4): gyrB
    # queries of ~1.9 kb on a ~590-node support-collapsed tree with its default k=35, m=4 (scaled up to 100 k queries)
    "G35": dict(n_leaves=300, ref_len=2200, k_size=35, m_size=4, n_reads=100_000, read_len=1900, deep=0, max_depth=0,
                collapse_prob=0.3),
    "C3s35": dict(n_leaves=10_000, ref_len=1500, k_size=35, m_size=4, n_reads=1_000_000, read_len=150, deep=0, max_depth=0,
                  collapse_prob=0.3),
}


class SynthDb:
    """Owns one generated tree + reference sequences + k-mer index."""

    def __init__(self, n_leaves, ref_len, k_size, m_size=4, seed_tree=1, seed_refseq=2, edge_sub_rate=0.01,
                 deep=0, max_depth=0, collapse_prob=0.0, id_stride=1, id_offset=0, threads=0, tips_only=False, **_ignored):
        cfg = _abi.SynthCfg(
            n_leaves=n_leaves, ref_len=ref_len, k_size=k_size, m_size=m_size, seed_tree=seed_tree,
            seed_refseq=seed_refseq, edge_sub_rate=edge_sub_rate, deep=deep, max_depth=max_depth,
            collapse_prob=collapse_prob, id_stride=id_stride, id_offset=id_offset, threads=threads,
            tips_only=1 if tips_only else 0,
        )
        self.cfg = cfg
        self._h = C.c_void_p()
        rc = _lib().cls_synth_db_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"cls_synth_db_create failed ({rc}): {_lib().cls_synth_last_error().decode()}")
        self.ref_len = ref_len
        self.n_leaves = _lib().cls_synth_n_leaves(self._h)
        self.max_depth = _lib().cls_synth_max_depth(self._h)
        self.flat = FlatDb.from_desc(_lib().cls_synth_db_desc(self._h).contents, keepalive=self)

    def close(self):
        if self._h:
            self.flat = None
            _lib().cls_synth_db_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def leaf_id(self, i: int) -> int:
        return _lib().cls_synth_leaf_id(self._h, i)

    def leaf_seq(self, i: int) -> str:
        return C.string_at(_lib().cls_synth_leaf_seq(self._h, i), self.ref_len).decode()

    def reads(self, n_reads, read_len, seed=3, first=0, err=0.01, frac_random=0.01):
        """-> (bases u8[n*read_len], offsets u64[n+1], truth_leaf u32[n])"""
        bases = np.empty(n_reads * read_len, dtype=np.uint8)
        offsets = np.empty(n_reads + 1, dtype=np.uint64)
        truth = np.empty(n_reads, dtype=np.uint32)
        rc = _lib().cls_synth_reads(self._h, seed, first, n_reads, read_len, err, frac_random,
                                    bases.ctypes.data, offsets.ctypes.data, truth.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"cls_synth_reads failed ({rc}): {_lib().cls_synth_last_error().decode()}")
        return bases, offsets, truth
