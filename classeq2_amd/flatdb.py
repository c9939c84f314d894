"""Flat (CSR) host view of a classeq2 `Tree` + `KmersMap`, as numpy arrays.

This is the shape `cls_db_create()` borrows (include/cls_place.h: cls_db_desc):
the reference's nested hash maps
(`KmersMap.map: HashMap<MinimizerKey, HashMap<u64, HashSet<u64>>>`,
core/src/domain/dtos/kmers_map.rs:77-87) flattened to two CSR levels, and the
`Clade` tree (clade.rs:18-38) flattened to a row table whose children are
consecutive rows.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _abi


@dataclass
class FlatDb:
    nodes: np.ndarray  # NODE_DTYPE [n_nodes], row 0 = root
    k_size: int
    m_size: int
    bucket_key: np.ndarray  # u64 [n_buckets]
    bucket_kmer_off: np.ndarray  # u64 [n_buckets+1]
    kmer_hash: np.ndarray  # u64 [n_kmers]
    kmer_node_off: np.ndarray  # u64 [n_kmers+1]
    node_ids: np.ndarray  # u64 [kmer_node_off[-1]]
    _keepalive: object = None

    def __post_init__(self):
        self.nodes = np.ascontiguousarray(self.nodes, dtype=_abi.NODE_DTYPE)
        for name in ("bucket_key", "bucket_kmer_off", "kmer_hash", "kmer_node_off", "node_ids"):
            setattr(self, name, np.ascontiguousarray(getattr(self, name), dtype=np.uint64))

    @property
    def n_nodes(self) -> int:
        return len(self.nodes)

    @property
    def n_kmers(self) -> int:
        return len(self.kmer_hash)

    def desc(self) -> _abi.DbDesc:
        """ctypes cls_db_desc borrowing this object's arrays (keep `self` alive)."""
        d = _abi.DbDesc()
        d.abi_version = _abi.ABI_VERSION
        d.n_nodes = self.n_nodes
        d.nodes = self.nodes.ctypes.data_as(C.POINTER(_abi.Node))
        d.k_size = self.k_size
        d.m_size = self.m_size
        d.n_buckets = len(self.bucket_key)
        u64p = C.POINTER(C.c_uint64)
        d.bucket_key = self.bucket_key.ctypes.data_as(u64p)
        d.bucket_kmer_off = self.bucket_kmer_off.ctypes.data_as(u64p)
        d.n_kmers = self.n_kmers
        d.kmer_hash = self.kmer_hash.ctypes.data_as(u64p)
        d.kmer_node_off = self.kmer_node_off.ctypes.data_as(u64p)
        d.node_ids = self.node_ids.ctypes.data_as(u64p)
        return d

    @classmethod
    def from_desc(cls, d: _abi.DbDesc, keepalive=None, copy: bool = False) -> "FlatDb":
        """Wrap (or copy) the arrays a cls_db_desc points at."""

        def arr(ptr, n, dtype):
            if n == 0:
                return np.zeros(0, dtype=dtype)
            a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dtype).itemsize,)).view(dtype)
            return a.copy() if copy else a

        nb, nk = int(d.n_buckets), int(d.n_kmers)
        kmer_node_off = arr(d.kmer_node_off, nk + 1, np.uint64)
        return cls(
            nodes=arr(d.nodes, int(d.n_nodes), _abi.NODE_DTYPE),
            k_size=int(d.k_size),
            m_size=int(d.m_size),
            bucket_key=arr(d.bucket_key, nb, np.uint64),
            bucket_kmer_off=arr(d.bucket_kmer_off, nb + 1, np.uint64),
            kmer_hash=arr(d.kmer_hash, nk, np.uint64),
            kmer_node_off=kmer_node_off,
            node_ids=arr(d.node_ids, int(kmer_node_off[-1]) if nk else 0, np.uint64),
            _keepalive=None if copy else keepalive,
        )

    @classmethod
    def from_nested(cls, root: dict, k_size: int, m_size: int, kmers_map: dict) -> "FlatDb":
        """Build from the nested shape of the reference's DB file
        (`root:` clade mapping with `children`, `kmersMap.map: {minimizer:
        {hash: [node ids]}}`, docs/book/02-build-db.md:137-196).  `root` nodes
        are dicts with id/parent/kind/children (children None or list)."""
        rows = []
        queue = [root]
        i = 0
        while i < len(queue):
            n = queue[i]
            ch = n.get("children")
            kind = {"ROOT": 0, "NODE": 1, "LEAF": 2}[n["kind"]]
            par = n.get("parent")
            rows.append(
                (
                    n["id"],
                    _abi.NO_PARENT if par is None else par,
                    len(queue) if ch else 0,
                    len(ch) if ch else 0,
                    kind,
                    0 if ch is None else 1,
                    (0,) * 6,
                )
            )
            if ch:
                queue.extend(ch)
            i += 1
        nodes = np.array(rows, dtype=_abi.NODE_DTYPE)
        bucket_key, bucket_off, kmer_hash, node_off, node_ids = [], [0], [], [0], []
        for key, bucket in kmers_map.items():
            bucket_key.append(int(key))
            for h, ids in bucket.items():
                kmer_hash.append(int(h))
                node_ids.extend(int(x) for x in ids)
                node_off.append(len(node_ids))
            bucket_off.append(len(kmer_hash))
        return cls(
            nodes=nodes,
            k_size=k_size,
            m_size=m_size,
            bucket_key=np.array(bucket_key, dtype=np.uint64),
            bucket_kmer_off=np.array(bucket_off, dtype=np.uint64),
            kmer_hash=np.array(kmer_hash, dtype=np.uint64),
            kmer_node_off=np.array(node_off, dtype=np.uint64),
            node_ids=np.array(node_ids, dtype=np.uint64),
        )
