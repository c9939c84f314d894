"""Literal CPU restatement of classeq2's `place_sequences` hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``classeq2_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may (and there only as the checker).

It follows the reference source structure line by line, with Python ``dict`` /
``set`` standing in for Rust ``HashMap`` / ``HashSet``.  Clarity over speed:
use it on small cases; ``oracle/cls_oracle.c`` is the flat C port used for
large cases and is itself validated against this file.

Parity pin status (see DESIGN.md "Oracle"):
  * MurmurHash3 x64-128 (crate mur3 0.1.0, not vendored under /root/reference)
    is pinned by the known-answer vectors printed in the reference's own docs
    (docs/book/02-build-db.md:181-196) -> tests/test_oracle_kat.py.
  * Placement *decisions*: the reference holds no assertion and no usable
    golden output for this path (the only k-mer index its result fixtures were
    produced with is a missing git-LFS blob) -> **parity unpinned** for
    decisions; they rest on this restatement following the source.

Every function cites the reference file:line it restates (paths relative to
/root/reference).
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Set, Tuple

MASK64 = (1 << 64) - 1

# ---------------------------------------------------------------------------
# a3: mur3::murmurhash3_x64_128(bytes, seed).0  (core/src/domain/dtos/kmers_map.rs:157-159)
# mur3 0.1.0 is Austin Appleby's public-domain MurmurHash3_x64_128.
# ---------------------------------------------------------------------------
_C1 = 0x87C37B91114253D5
_C2 = 0x4CF5AD432745937F


def _rotl64(x: int, r: int) -> int:
    return ((x << r) | (x >> (64 - r))) & MASK64


def _fmix64(k: int) -> int:
    k ^= k >> 33
    k = (k * 0xFF51AFD7ED558CCD) & MASK64
    k ^= k >> 33
    k = (k * 0xC4CEB9FE1A85EC53) & MASK64
    k ^= k >> 33
    return k


def murmurhash3_x64_128(data: bytes, seed: int = 0) -> Tuple[int, int]:
    n = len(data)
    h1 = seed & MASK64
    h2 = seed & MASK64
    nblocks = n // 16
    for b in range(nblocks):
        k1 = int.from_bytes(data[16 * b : 16 * b + 8], "little")
        k2 = int.from_bytes(data[16 * b + 8 : 16 * b + 16], "little")
        k1 = (k1 * _C1) & MASK64
        k1 = _rotl64(k1, 31)
        k1 = (k1 * _C2) & MASK64
        h1 ^= k1
        h1 = _rotl64(h1, 27)
        h1 = (h1 + h2) & MASK64
        h1 = (h1 * 5 + 0x52DCE729) & MASK64
        k2 = (k2 * _C2) & MASK64
        k2 = _rotl64(k2, 33)
        k2 = (k2 * _C1) & MASK64
        h2 ^= k2
        h2 = _rotl64(h2, 31)
        h2 = (h2 + h1) & MASK64
        h2 = (h2 * 5 + 0x38495AB5) & MASK64
    tail = data[16 * nblocks :]
    k1 = 0
    k2 = 0
    t = len(tail)
    if t > 8:
        for i in range(t - 1, 7, -1):
            k2 ^= tail[i] << (8 * (i - 8))
        k2 = (k2 * _C2) & MASK64
        k2 = _rotl64(k2, 33)
        k2 = (k2 * _C1) & MASK64
        h2 ^= k2
    if t > 0:
        for i in range(min(t, 8) - 1, -1, -1):
            k1 ^= tail[i] << (8 * i)
        k1 = (k1 * _C1) & MASK64
        k1 = _rotl64(k1, 31)
        k1 = (k1 * _C2) & MASK64
        h1 ^= k1
    h1 ^= n
    h2 ^= n
    h1 = (h1 + h2) & MASK64
    h2 = (h2 + h1) & MASK64
    h1 = _fmix64(h1)
    h2 = _fmix64(h2)
    h1 = (h1 + h2) & MASK64
    h2 = (h2 + h1) & MASK64
    return h1, h2


def hash_kmer(kmer: str) -> int:
    """KmersMap::hash_kmer, kmers_map.rs:157-159."""
    return murmurhash3_x64_128(kmer.encode("ascii"), 0)[0]


def build_minimizer_from_string(kmer: str, size: int) -> int:
    """MinimizerKey::build_minimizer_from_string, kmers_map.rs:10-13.

    "Minimizer" = hash of the first `size` characters (a prefix bucket)."""
    return hash_kmer(kmer[:size])


# ---------------------------------------------------------------------------
# a11: data model
# ---------------------------------------------------------------------------
ROOT, NODE, LEAF = "ROOT", "NODE", "LEAF"


@dataclass
class Clade:
    """clade.rs:18-38.  LEAF-ness is decided by `kind` only (clade.rs:166-172)."""

    id: int
    parent: Optional[int]
    kind: str
    name: Optional[str] = None
    support: Optional[float] = None
    length: Optional[float] = None
    children: Optional[List["Clade"]] = None

    def is_leaf(self) -> bool:
        return self.kind == LEAF

    def get_node_by_id(self, id_: int) -> Optional["Clade"]:
        """clade.rs:95-109 (first match in DFS order)."""
        if self.id == id_:
            return self
        if self.children is not None:
            for child in self.children:
                node = child.get_node_by_id(id_)
                if node is not None:
                    return node
        return None

    def get_path_to_root(self, root: "Clade") -> Set[int]:
        """clade.rs:111-125."""
        path = {self.id}
        if self.parent is not None:
            path.add(self.parent)
            parent = root.get_node_by_id(self.parent)
            if parent is not None:
                path |= parent.get_path_to_root(root)
        return path

    def get_leaves_with_paths(self, parent_ids=None):
        """clade.rs:127-156."""
        parent_ids = [self.id] if parent_ids is None else parent_ids + [self.id]
        leaves = []
        if self.is_leaf():
            leaves.append((self, parent_ids))
        elif self.children is not None:
            for child in self.children:
                leaves.extend(child.get_leaves_with_paths(list(parent_ids)))
        return leaves


class KmersMap:
    """kmers_map.rs:77-87: {minimizer-key -> {kmer-hash -> {node ids}}}."""

    def __init__(self, k_size: int, m_size: int):
        self.k_size = k_size
        self.m_size = m_size
        self.map: Dict[int, Dict[int, Set[int]]] = {}

    # -- insert side (kmers_map.rs:125-149, :24-35); used by the DB generators
    def insert_or_append_kmer_hash(self, kmer: str, hash_: int, nodes: Iterable[int]) -> None:
        key = 0 if self.m_size == 0 else build_minimizer_from_string(kmer, self.m_size)
        bucket = self.map.setdefault(key, {})
        bucket.setdefault(hash_, set()).update(nodes)

    # -- a2 -----------------------------------------------------------------
    @staticmethod
    def reverse_complement(sequence: str) -> str:
        """kmers_map.rs:431-443 (panics on anything but aAcCgGtT)."""
        comp = {"a": "T", "A": "T", "t": "A", "T": "A", "c": "G", "C": "G", "g": "C", "G": "C"}
        out = []
        for c in reversed(sequence):
            if c not in comp:
                raise ValueError("Invalid character in sequence")
            out.append(comp[c])
        return "".join(out)

    @staticmethod
    def build_kmers_from_sequence(sequence: str, size: int) -> List[Tuple[str, int]]:
        """kmers_map.rs:405-424."""
        s = sequence.upper()
        return [(s[i : i + size], hash_kmer(s[i : i + size])) for i in range(len(s) - size + 1)]

    def build_kmer_from_string(self, sequence: str) -> List[Tuple[str, int]]:
        """kmers_map.rs:375-398: all forward k-mers, then all k-mers of the
        reverse complement; [] if the sequence is shorter than k."""
        if len(sequence) < self.k_size:
            return []
        kmers = self.build_kmers_from_sequence(sequence, self.k_size)
        kmers.extend(self.build_kmers_from_sequence(self.reverse_complement(sequence), self.k_size))
        return kmers

    # -- a5 -----------------------------------------------------------------
    def get_overlapping_hashed_kmers(self, hashed_kmers: List[Tuple[str, int]]) -> "KmersMap":
        """kmers_map.rs:273-311 + MinimizerValue::get_overlapping_hashed_kmers :55-70."""
        out = KmersMap(self.k_size, self.m_size)
        minimizers = {build_minimizer_from_string(kmer, self.m_size) for kmer, _ in hashed_kmers}
        hashes = {h for _, h in hashed_kmers}
        for key, value in self.map.items():
            if key not in minimizers:
                continue
            kept = {h: set(value[h]) for h in (set(value.keys()) & hashes)}
            if kept:
                out.map[key] = kept
        return out

    # -- a6 -----------------------------------------------------------------
    def get_minimized_hashes_with_node(self, node: int) -> Optional[Dict[int, Set[int]]]:
        """kmers_map.rs:211-229 (+ :37-53)."""
        res = {}
        for key, value in self.map.items():
            s = {h for h, nodes in value.items() if node in nodes}
            if s:
                res[key] = s
        return res or None

    def get_overlapping_minimized_hashes(self, hashed: Dict[int, Set[int]]) -> "KmersMap":
        """kmers_map.rs:318-344."""
        out = KmersMap(self.k_size, self.m_size)
        for key, value in self.map.items():
            if key in hashed:
                kept = {h: set(value[h]) for h in (set(value.keys()) & hashed[key])}
                if kept:
                    out.map[key] = kept
        return out

    # -- a8 -----------------------------------------------------------------
    def get_hashed_kmers_with_node(self, node: int) -> Optional[Set[int]]:
        """kmers_map.rs:189-203 (+ :37-53): flattened over buckets into ONE set
        of hashes (so a hash present in two buckets counts once here)."""
        res: Set[int] = set()
        for value in self.map.values():
            res |= {h for h, nodes in value.items() if node in nodes}
        return res or None

    def n_kmers(self) -> int:
        return sum(len(v) for v in self.map.values())


@dataclass
class Tree:
    """tree.rs:9-52 (fields the path reads)."""

    root: Clade
    kmers_map: Optional[KmersMap]
    annotations: Optional[list] = None
    name: str = "tree"


@dataclass
class AdherenceTest:
    """adherence_test.rs:6-17; `clade` is a Clade record or a bare id."""

    clade: object
    one: int
    rest: int


# PlacementStatus, placement_response.rs:7-28 -- modelled as tagged tuples:
#   ("Unclassifiable", msg) | ("IdentityFound", AdherenceTest)
#   ("MaxResolutionReached", id, msg) | ("Inconclusive", [AdherenceTest], msg)
class PlaceError(Exception):
    """use_case_err(...).as_error(): (message, telemetry code or None)."""

    def __init__(self, msg: str, code: Optional[str] = None):
        super().__init__(msg)
        self.msg = msg
        self.code = code


def status_to_string(status) -> str:
    """PlacementStatus::to_string, placement_response.rs:30-42."""
    tag = status[0]
    if tag == "Unclassifiable":
        return f"Unclassifiable: {status[1]}"
    if tag == "IdentityFound":
        return "IdentityFound"
    if tag == "MaxResolutionReached":
        return f"MaxResolutionReached: {status[2]}"
    return f"Inconclusive: {status[2]}"


def rust_as_usize(x: float) -> int:
    """Rust `f64 as usize`: saturating, NaN -> 0."""
    if x != x or x <= 0:
        return 0
    return min(int(x), (1 << 64) - 1)


def rust_round(x: float) -> float:
    """f64::round: half away from zero (x >= 0 on this path); NaN stays NaN."""
    if x != x or x in (float("inf"), float("-inf")):
        return x
    r = math.floor(x)
    if x - r >= 0.5:
        r += 1
    return float(r)


# ---------------------------------------------------------------------------
# a10: update_introspection_node.rs:13-91
# ---------------------------------------------------------------------------
def update_introspection_node(adherence: AdherenceTest):
    if not isinstance(adherence.clade, Clade):
        raise PlaceError("The adherence test does not contain a clade record.")
    parent = adherence.clade
    if parent.children is None:
        return ("Return", ("IdentityFound", adherence))
    non_leaf_children = [c for c in parent.children if not c.is_leaf()]
    if not non_leaf_children:
        return ("Return", ("IdentityFound", adherence))
    return ("Continue", parent, non_leaf_children)


# ---------------------------------------------------------------------------
# a7..a10: place_sequence.rs:42-602
# ---------------------------------------------------------------------------
@dataclass
class Trace:
    """Intermediate quantities (tracing span fields place_sequence.rs:30-41);
    lets tests compare more than the final status."""

    n_query_kmers: int = 0
    query_kmers_len: int = 0
    introspection_coverage: int = 0
    levels: int = 0
    per_level: list = field(default_factory=list)


def place_sequence(
    header: str,
    sequence: str,
    tree: Tree,
    max_iterations: Optional[int] = None,
    min_match_coverage: Optional[float] = None,
    remove_intersection: Optional[bool] = None,
    trace: Optional[Trace] = None,
):
    tr = trace if trace is not None else Trace()
    # :64-75
    remove_intersection = False if remove_intersection is None else remove_intersection
    max_iterations = 1000 if max_iterations is None else max_iterations
    if min_match_coverage is not None:
        v = min_match_coverage
        min_match_coverage = 1.0 if v > 1.0 else (0.0 if v < 0.0 else v)
    else:
        min_match_coverage = 0.7
    # :77-80 (the per-query deep clone has no observable effect)
    kmers_map = tree.kmers_map
    assert kmers_map is not None, "The tree does not have a kmers map."
    # :87-102
    query_kmers = kmers_map.build_kmer_from_string(sequence)
    tr.n_query_kmers = len(query_kmers)
    if len(query_kmers) < 2:
        raise PlaceError("The sequence does not contain enough kmers.", "UCPLACE0005")
    # :118-139
    query_kmers_map = kmers_map.get_overlapping_hashed_kmers(query_kmers)
    query_kmers_len = sum(len(v) for v in query_kmers_map.map.values())
    tr.query_kmers_len = query_kmers_len
    if query_kmers_len == 0:
        # `{query:?}` of SequenceHeader(String) -> SequenceHeader("...")
        return ("Unclassifiable", f"Query sequence {rust_debug_header(header)} may not be related to the phylogeny")
    # :156-166
    with_root = query_kmers_map.get_minimized_hashes_with_node(tree.root.id)
    if with_root is None:
        return ("Unclassifiable", "Query sequence has no overlapping kmers with the reference tree")
    introspection_kmers = query_kmers_map.get_overlapping_minimized_hashes(with_root)
    # :199-206
    if tree.root.children is None:
        raise PlaceError("The root node does not have children. This is unexpected.")
    children = tree.root.children
    iteration = 0
    parent = tree.root  # :220
    # :231-254
    expected_min_clade_coverage = rust_round(query_kmers_len * min_match_coverage)
    introspection_coverage = sum(len(v) for v in introspection_kmers.map.values())
    tr.introspection_coverage = introspection_coverage
    if introspection_coverage < rust_as_usize(expected_min_clade_coverage):
        return ("Unclassifiable", f"Insufficient kmers coverage: {introspection_coverage}")
    # :279-601
    while True:
        iteration += 1
        tr.levels = iteration
        if iteration > max_iterations:
            raise PlaceError("The maximum number of iterations has been reached.", "UCPLACE0010")
        # PHASE 1a :319-333
        children_kmers = []
        for record in children:
            if record.is_leaf():
                continue
            kmers = introspection_kmers.get_hashed_kmers_with_node(record.id)
            if kmers is None:
                continue
            children_kmers.append((kmers, record))
        children_kmers.sort(key=lambda kc: -len(kc[0]))  # :335 (cosmetic)
        # PHASE 1b :353-418
        clade_proposals: List[AdherenceTest] = []
        level_dbg = []
        for kmers, clade in children_kmers:
            # NB :361 compares clade *ids*
            rest = [rk for rk, nested in children_kmers if nested.id != clade.id]
            if not rest:
                adherence = AdherenceTest(clade, len(kmers), 0)
            else:
                rest_len = set().union(*rest)
                if remove_intersection:
                    one_kmers = kmers - rest_len
                    rest_kmers = rest_len - kmers
                else:
                    one_kmers, rest_kmers = kmers, rest_len
                adherence = AdherenceTest(clade, len(one_kmers), len(rest_kmers))
            level_dbg.append((clade.id, adherence.one, adherence.rest))
            if adherence.one > adherence.rest:  # :411-417
                clade_proposals.append(adherence)
        tr.per_level.append((parent.id, level_dbg))
        # PHASE 2 :436-600
        if not clade_proposals:
            if iteration == 1:
                return (
                    "Unclassifiable",
                    "Tree introspection not possible. Query sequence has no overlapping kmers with the reference tree",
                )
            return ("MaxResolutionReached", parent.id, "LCA Accepted")
        if len(clade_proposals) == 1:
            upd = update_introspection_node(clade_proposals[0])
            if upd[0] == "Return":
                return upd[1]
            parent, children = upd[1], upd[2]
            continue
        # :519-599 (set-theoretically unreachable; kept for fidelity)
        fold: Dict[int, List[AdherenceTest]] = {}
        for a in clade_proposals:
            fold.setdefault(a.one - a.rest, []).append(a)
        max_diff_key = max(fold.keys())
        max_diff_value = fold[max_diff_key]
        if len(max_diff_value) == 1:
            upd = update_introspection_node(max_diff_value[0])
            if upd[0] == "Return":
                return upd[1]
            parent, children = upd[1], upd[2]
            continue
        return (
            "Inconclusive",
            [AdherenceTest(a.clade.id if isinstance(a.clade, Clade) else a.clade, a.one, a.rest) for a in clade_proposals],
            "Multiple proposals",
        )


def rust_debug_header(header: str) -> str:
    """`{:?}` of `SequenceHeader(String)` (sequence.rs:4-6): tuple-struct Debug
    around `str` Debug (escapes `"`, `\\`, control chars)."""
    out = ['"']
    for ch in header:
        if ch == '"':
            out.append('\\"')
        elif ch == "\\":
            out.append("\\\\")
        elif ch == "\n":
            out.append("\\n")
        elif ch == "\r":
            out.append("\\r")
        elif ch == "\t":
            out.append("\\t")
        elif ch == "\0":
            out.append("\\0")
        elif ord(ch) < 0x20 or ord(ch) == 0x7F:
            out.append("\\u{%x}" % ord(ch))
        else:
            out.append(ch)
    out.append('"')
    return "SequenceHeader(" + "".join(out) + ")"


# ---------------------------------------------------------------------------
# a1: FASTA input stage
# ---------------------------------------------------------------------------
def remove_non_iupac_from_sequence(sequence: str) -> str:
    """sequence.rs:47-56: upper-case, then DELETE every char not in ACGT."""
    return "".join(c for c in sequence.upper() if c in "ACGT")


def sequence_content_by_channel(text: str) -> List[Tuple[str, str]]:
    """file_or_stdin.rs:76-116.  Returns the records that would have been sent
    down the channel before any error (the caller ignores the error,
    place_sequences/mod.rs:119)."""
    out: List[Tuple[str, str]] = []
    header = ""
    sequence = ""
    # BufRead::lines(): split at '\n'; a line that WAS terminated by '\n' also
    # loses one trailing '\r' (an unterminated last line keeps it)
    parts = text.split("\n")
    lines = [p[:-1] if p.endswith("\r") else p for p in parts[:-1]]
    if parts[-1] != "":
        lines.append(parts[-1])
    for line in lines:
        if line == "":
            continue
        if line.startswith(">"):
            if header != "":
                out.append((header, sequence))
                sequence = ""
            elif sequence != "":
                return out  # Err("unexpected sequence without header"), ignored by caller
            header = line.replace(">", "")
        else:
            sequence += remove_non_iupac_from_sequence(line)
    if header != "" and sequence != "":
        out.append((header, sequence))
    return out


# ---------------------------------------------------------------------------
# Fixed-size record view of a result (the C-ABI's cls_placement, include/cls_place.h)
# ---------------------------------------------------------------------------
ST_UNCLASSIFIABLE_NO_MATCH = 0
ST_UNCLASSIFIABLE_NO_ROOT = 1
ST_UNCLASSIFIABLE_COVERAGE = 2
ST_UNCLASSIFIABLE_LEVEL1 = 3
ST_IDENTITY_FOUND = 4
ST_MAX_RESOLUTION = 5
ST_INCONCLUSIVE = 6
ST_ERR_TOO_FEW_KMERS = 7
ST_ERR_MAX_ITER = 8
ST_ERR_ROOT_NO_CHILDREN = 9
ST_ERR_INVALID_BASE = 10


def place_to_record(header, sequence, tree, max_iterations=None, min_match_coverage=None, remove_intersection=None):
    """Run place_sequence and flatten the outcome to
    (status, one, rest, levels, clade_id) exactly as include/cls_place.h
    documents the cls_placement fields."""
    tr = Trace()
    try:
        st = place_sequence(header, sequence, tree, max_iterations, min_match_coverage, remove_intersection, tr)
    except PlaceError as e:
        if e.code == "UCPLACE0005":
            return (ST_ERR_TOO_FEW_KMERS, 0, 0, 0, 0)
        if e.code == "UCPLACE0010":
            return (ST_ERR_MAX_ITER, 0, 0, tr.levels, 0)
        if e.msg.startswith("The root node does not have children"):
            return (ST_ERR_ROOT_NO_CHILDREN, 0, 0, 0, 0)
        raise
    except ValueError:
        # Rust panics in reverse_complement (kmers_map.rs:440); the C-ABI
        # reports it as a per-read error status instead of aborting.
        return (ST_ERR_INVALID_BASE, 0, 0, 0, 0)
    tag = st[0]
    if tag == "Unclassifiable":
        msg = st[1]
        if msg.endswith("may not be related to the phylogeny"):
            return (ST_UNCLASSIFIABLE_NO_MATCH, 0, 0, 0, 0)
        if msg.startswith("Insufficient kmers coverage"):
            return (ST_UNCLASSIFIABLE_COVERAGE, tr.introspection_coverage, 0, 0, 0)
        if msg.startswith("Tree introspection not possible"):
            return (ST_UNCLASSIFIABLE_LEVEL1, 0, 0, 1, 0)
        return (ST_UNCLASSIFIABLE_NO_ROOT, 0, 0, 0, 0)
    if tag == "IdentityFound":
        a = st[1]
        return (ST_IDENTITY_FOUND, a.one, a.rest, tr.levels, a.clade.id)
    if tag == "MaxResolutionReached":
        return (ST_MAX_RESOLUTION, 0, 0, tr.levels, st[1])
    return (ST_INCONCLUSIVE, len(st[1]), 0, tr.levels, tr.per_level[-1][0])
