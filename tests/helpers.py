"""Shared test helpers (test infrastructure; may use oracle/)."""
import numpy as np

from classeq2_amd import _abi
from classeq2_amd.flatdb import FlatDb

PARAM_SETS = [
    dict(),
    dict(remove_intersection=True),
    dict(min_match_coverage=1.0),
    dict(min_match_coverage=0.0, remove_intersection=False),
    dict(max_iterations=2),
    dict(max_iterations=0),
]
# Option<> corner values: clamping (place_sequence.rs:67-75), NaN coverage (`as usize` -> 0), negative iteration caps
ODD_PARAM_SETS = [
    dict(min_match_coverage=7.5),
    dict(min_match_coverage=-3.0),
    dict(min_match_coverage=float("nan")),
    dict(min_match_coverage=0.5, remove_intersection=True, max_iterations=1000000),
    dict(max_iterations=-4),
    dict(max_iterations=1),
]


def records_equal(a: np.ndarray, b: np.ndarray):
    """Field-wise comparison of cls_placement arrays (padding ignored)."""
    bad = np.zeros(len(a), dtype=bool)
    for f in ("status", "one", "rest", "levels", "clade_id"):
        bad |= a[f] != b[f]
    return np.nonzero(bad)[0]


def stats_equal(a: np.ndarray, b: np.ndarray):
    bad = np.zeros(len(a), dtype=bool)
    for f in ("n_query_kmers", "n_matched", "n_with_root", "leaf_postings"):
        bad |= a[f] != b[f]
    return np.nonzero(bad)[0]


def device_place(db, bases, offsets, params=None, want_stats=True):
    """cls_place_batch_device on torch-owned HBM buffers (cuda:0) -> (records, stats)."""
    import torch

    dev = torch.device("cuda:0")
    n = len(offsets) - 1
    d_b = torch.from_numpy(bases if len(bases) else np.zeros(1, np.uint8)).to(dev)
    d_o = torch.from_numpy(offsets.astype(np.int64)).to(dev)
    d_out = torch.zeros(n * 24, dtype=torch.uint8, device=dev)
    d_st = torch.zeros(n * 24, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    db.place_batch_device(d_b.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), params, d_st.data_ptr() if want_stats else 0, 0)
    torch.cuda.synchronize()
    return d_out.cpu().numpy().view(_abi.PLACEMENT_DTYPE), d_st.cpu().numpy().view(_abi.STATS_DTYPE)


def describe(rec) -> str:
    return f"{_abi.STATUS_NAMES[int(rec['status'])]} one={rec['one']} rest={rec['rest']} levels={rec['levels']} clade={rec['clade_id']}"


def drop_random_nodes(flat: FlatDb, frac: float, seed: int, keep_root_frac: float = 0.9) -> FlatDb:
    """Make node sets that are NOT closed under `parent` (arbitrary DB files are
    allowed at the boundary): delete a random fraction of the ids of every set."""
    rng = np.random.default_rng(seed)
    root_id = int(flat.nodes[0]["id"])
    keep = rng.random(len(flat.node_ids)) >= frac
    is_root = flat.node_ids == root_id
    keep[is_root] = rng.random(int(is_root.sum())) < keep_root_frac
    new_ids = flat.node_ids[keep]
    csum = np.concatenate([[0], np.cumsum(keep)])
    new_off = csum[flat.kmer_node_off.astype(np.int64)].astype(np.uint64)
    return FlatDb(nodes=flat.nodes.copy(), k_size=flat.k_size, m_size=flat.m_size, bucket_key=flat.bucket_key.copy(),
                  bucket_kmer_off=flat.bucket_kmer_off.copy(), kmer_hash=flat.kmer_hash.copy(),
                  kmer_node_off=new_off, node_ids=new_ids)


def truncate_random_sets(flat: FlatDb, frac: float, seed: int) -> FlatDb:
    """Node sets that stay closed under `parent` but hold nothing below the root: a random fraction of the k-mers
    keeps only {root}, another one the empty set (an index file may say so; `cls build-db` never does)."""
    rng = np.random.default_rng(seed)
    root_id = int(flat.nodes[0]["id"])
    off = flat.kmer_node_off.astype(np.int64)
    nk = len(off) - 1
    u = rng.random(nk)
    mode = np.where(u < frac, 1, np.where(u < 2 * frac, 2, 0))  # 1: {root}, 2: {}
    owner = np.repeat(np.arange(nk), np.diff(off))
    is_root = flat.node_ids == root_id
    keep = (mode[owner] == 0) | ((mode[owner] == 1) & is_root)
    new_ids = flat.node_ids[keep]
    csum = np.concatenate([[0], np.cumsum(keep)])
    return FlatDb(nodes=flat.nodes.copy(), k_size=flat.k_size, m_size=flat.m_size, bucket_key=flat.bucket_key.copy(),
                  bucket_kmer_off=flat.bucket_kmer_off.copy(), kmer_hash=flat.kmer_hash.copy(),
                  kmer_node_off=csum[off].astype(np.uint64), node_ids=new_ids)


def ragged_reads(rng, synth, n, min_len, max_len, err=0.02, frac_random=0.05, lower_frac=0.1):
    """Reads of varying length (incl. shorter than k and empty), some lower-case."""
    lens = rng.integers(min_len, max_len + 1, size=n)
    parts, offs = [], [0]
    for i, L in enumerate(lens):
        L = int(L)
        if L == 0:
            offs.append(offs[-1])
            continue
        b, _, _ = synth.reads(1, L, seed=int(rng.integers(1 << 30)), first=i, err=err, frac_random=frac_random)
        if rng.random() < lower_frac:
            b = b | 0x20  # lower-case
        parts.append(b)
        offs.append(offs[-1] + L)
    bases = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
    return bases, np.array(offs, dtype=np.uint64)


def star_of_cherries(n_cherries: int, seq_len: int = 60, k: int = 7, m: int = 3, seed: int = 0):
    """A root with `n_cherries` internal children of two leaves each (a polytomy whose non-LEAF arity is
    n_cherries), indexed like `cls build-db` does.  -> (FlatDb, list of leaf sequences)"""
    from oracle import oracle_literal as lit

    rng = np.random.default_rng(seed)
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    root = dict(id=0, parent=None, kind="ROOT", children=[])
    km = lit.KmersMap(k, m)
    seqs = []
    nid = 1
    for _ in range(n_cherries):
        base = alphabet[rng.integers(0, 4, seq_len)]
        node = dict(id=nid, parent=0, kind="NODE", children=[])
        nid += 1
        for _leaf in range(2):
            sq = base.copy()
            pos = rng.integers(0, seq_len, 2)
            sq[pos] = alphabet[rng.integers(0, 4, 2)]
            seq = bytes(sq).decode()
            seqs.append(seq)
            node["children"].append(dict(id=nid, parent=node["id"], kind="LEAF"))
            for kmer, h in km.build_kmer_from_string(seq):
                km.insert_or_append_kmer_hash(kmer, h, {0, node["id"], nid})
            nid += 1
        root["children"].append(node)
    return FlatDb.from_nested(root, k, m, km.map), seqs
