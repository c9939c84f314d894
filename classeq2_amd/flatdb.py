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
    leaves_only: bool = False  # node_ids list only the LEAF-kind members (CLS_SETS_LEAVES, include/cls_place.h)

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
        d.node_set_kind = _abi.SETS_LEAVES if self.leaves_only else _abi.SETS_EXPLICIT
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
            leaves_only=bool(d.abi_version >= 2 and d.node_set_kind == _abi.SETS_LEAVES),
        )

    def to_leaves_only(self) -> "FlatDb":
        """The same index with every node set reduced to its LEAF-kind members (what CLS_SETS_LEAVES carries); only
        equivalent when the sets are unions of root->leaf paths, as `cls build-db` makes them."""
        leaf_ids = np.sort(self.nodes["id"][self.nodes["kind"] == _abi.KIND_LEAF])
        pos = np.searchsorted(leaf_ids, self.node_ids)
        keep = (pos < len(leaf_ids)) & (leaf_ids[np.minimum(pos, len(leaf_ids) - 1)] == self.node_ids)
        csum = np.concatenate([[0], np.cumsum(keep)])
        return FlatDb(nodes=self.nodes.copy(), k_size=self.k_size, m_size=self.m_size, bucket_key=self.bucket_key.copy(),
                      bucket_kmer_off=self.bucket_kmer_off.copy(), kmer_hash=self.kmer_hash.copy(),
                      kmer_node_off=csum[self.kmer_node_off.astype(np.int64)].astype(np.uint64), node_ids=self.node_ids[keep],
                      leaves_only=True)

    def to_explicit(self) -> "FlatDb":
        """A leaves-only index expanded to explicit node sets (union of the root->leaf paths): what the reference's own
        index file holds, and what the oracle is fed."""
        assert self.leaves_only
        ids = self.nodes["id"]
        order = np.argsort(ids)
        row_of = lambda x: order[np.searchsorted(ids[order], x)]  # noqa: E731
        parent_row = np.full(len(ids), -1, dtype=np.int64)
        has_par = self.nodes["parent"] != _abi.NO_PARENT
        parent_row[has_par] = row_of(self.nodes["parent"][has_par])
        off = self.kmer_node_off.astype(np.int64)
        out_ids, out_off = [], [0]
        for j in range(len(off) - 1):
            seen = set()
            for r in row_of(self.node_ids[off[j]:off[j + 1]]):
                r = int(r)
                while r >= 0 and r not in seen:
                    seen.add(r)
                    r = int(parent_row[r])
            out_ids.extend(int(ids[r]) for r in seen)
            out_off.append(len(out_ids))
        return FlatDb(nodes=self.nodes.copy(), k_size=self.k_size, m_size=self.m_size, bucket_key=self.bucket_key.copy(),
                      bucket_kmer_off=self.bucket_kmer_off.copy(), kmer_hash=self.kmer_hash.copy(),
                      kmer_node_off=np.array(out_off, dtype=np.uint64), node_ids=np.array(out_ids, dtype=np.uint64))

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
