"""CPU: pin the oracle (murmur KATs from the reference docs), and check the C
port against the literal restatement record for record."""
import numpy as np
import pytest

from classeq2_amd.synth import SynthDb
from oracle import oracle_literal as lit
from oracle import oracle_port as op
from tests.helpers import ODD_PARAM_SETS, PARAM_SETS, drop_random_nodes, ragged_reads, records_equal

# Known answers printed in the reference's own docs (docs/book/02-build-db.md:181-196):
# minimizer keys + k-mer hashes of the bsub-gyrB model (k=35, m=4); the two
# 35-mers occur in tests/data/public/.../input/*.fasta of the reference.
KATS = [
    (b"CCAA", 10631256518523097406),
    (b"ATAC", 10517626403121597142),
    (b"CCAAAGCCAGGCTTGCAGCGTTATAAAGGTCTTGG", 15277858258350234285),
    (b"ATACAATGACAAGGAGCTTGAAGAACTGTTAAAAA", 13045666331085307167),
    (b"", 0),                          # MinimizerKey(0) for mSize == 0, kmers_map.rs:131-134
    (b"hello", 0xCBD8A7B341BD9B02),    # public MurmurHash3_x64_128 vector
]


@pytest.mark.parametrize("data,want", KATS)
def test_murmur_kat(data, want):
    assert lit.murmurhash3_x64_128(data, 0)[0] == want
    assert op.lib().cls_oracle_murmur3_h1(data, len(data)) == want


def test_murmur_c_vs_literal_all_lengths():
    rng = np.random.default_rng(0)
    for n in range(0, 70):
        data = bytes(rng.integers(65, 91, size=n, dtype=np.uint8))
        assert op.lib().cls_oracle_murmur3_h1(data, n) == lit.murmurhash3_x64_128(data, 0)[0]


def test_rust_round_and_debug():
    assert lit.rust_round(0.5) == 1.0 and lit.rust_round(1.5) == 2.0 and lit.rust_round(2.4999) == 2.0
    assert lit.rust_round(0.49999999999999994) == 0.0
    assert lit.rust_debug_header('a"b\\c\n') == 'SequenceHeader("a\\"b\\\\c\\n")'


CASES = [
    (40, 200, 6, 3, 0.0, 1, 0, 60),
    (60, 300, 8, 4, 0.3, 3, 7, 80),
    (100, 300, 10, 4, 0.5, 1, 0, 100),
    (30, 120, 5, 0, 0.5, 5, 100, 40),
    (50, 200, 7, 9, 0.0, 1, 0, 20),
]


@pytest.mark.parametrize("case", CASES)
def test_port_matches_literal(case):
    nl, rl, k, m, cp, stride, off, rdlen = case
    s = SynthDb(nl, rl, k, m, collapse_prob=cp, id_stride=stride, id_offset=off)
    tree = op.flat_to_literal(s.flat)
    port = op.OraclePort(s.flat)
    bases, offsets, _ = s.reads(150, rdlen, frac_random=0.05, err=0.02)
    for kw in PARAM_SETS + (ODD_PARAM_SETS if nl == 60 else []):
        want = op.literal_place_batch(tree, bases, offsets, **kw)
        got = port.place_batch(bases, offsets, op.make_params(**kw), threads=2)
        assert len(records_equal(got, want)) == 0, kw


def test_port_matches_literal_sets_with_nothing_below_the_root():
    """Node sets {root} and {}: counted in |M| (and |M_root|), never a vote."""
    from tests.helpers import truncate_random_sets

    s = SynthDb(60, 300, 9, 4)
    flat = truncate_random_sets(s.flat, 0.15, seed=4)
    tree = op.flat_to_literal(flat)
    port = op.OraclePort(flat)
    bases, offsets, _ = s.reads(150, 110, frac_random=0.05, err=0.02)
    for kw in (dict(), dict(min_match_coverage=1.0)):
        want = op.literal_place_batch(tree, bases, offsets, **kw)
        got, st = port.place_batch(bases, offsets, op.make_params(**kw), threads=2, want_stats=True)
        assert len(records_equal(got, want)) == 0, kw
    assert (st["n_matched"] > st["n_with_root"]).any()


def test_port_matches_literal_non_closed_and_ragged():
    s = SynthDb(60, 300, 8, 4, collapse_prob=0.3)
    flat = drop_random_nodes(s.flat, 0.2, seed=3)
    tree = op.flat_to_literal(flat)
    port = op.OraclePort(flat)
    bases, offsets = ragged_reads(np.random.default_rng(1), s, 150, 0, 120)
    bases = bases.copy()
    bases[int(offsets[5]) + 3] = ord("N")
    want = op.literal_place_batch(tree, bases, offsets)
    got = port.place_batch(bases, offsets)
    assert len(records_equal(got, want)) == 0
    # trace counters of the literal oracle vs the port's stats
    got, st = port.place_batch(bases, offsets, want_stats=True)
    raw = bytes(bases)
    for i in range(0, 150, 7):
        tr = lit.Trace()
        try:
            lit.place_sequence(f"r{i}", raw[int(offsets[i]):int(offsets[i + 1])].decode(), tree, trace=tr)
        except (lit.PlaceError, ValueError):
            continue
        assert (tr.n_query_kmers, tr.query_kmers_len) == (st[i]["n_query_kmers"], st[i]["n_matched"])


def test_duplicate_hash_across_buckets_semantics():
    """A hash present under two buckets: |M| counts both entries, K_c counts the
    hash once (kmers_map.rs:189-203 flattens into one HashSet)."""
    root = dict(id=0, parent=None, kind="ROOT", children=[
        dict(id=1, parent=0, kind="NODE", children=[dict(id=3, parent=1, kind="LEAF"), dict(id=4, parent=1, kind="NODE", children=[dict(id=6, parent=4, kind="LEAF")])]),
        dict(id=2, parent=0, kind="NODE", children=[dict(id=5, parent=2, kind="LEAF")]),
    ])
    from classeq2_amd.flatdb import FlatDb
    seq = "ACGTTGCA"
    k, m = 4, 2
    km = lit.KmersMap(k, m)
    for kmer, h in km.build_kmer_from_string(seq):
        km.insert_or_append_kmer_hash(kmer, h, {0, 1, 4})
    # plant the hash of "ACGT" under the bucket of "TG" as well, with another node set
    h_acgt = lit.hash_kmer("ACGT")
    km.map.setdefault(lit.hash_kmer("TG"), {})[h_acgt] = {0, 2}
    flat = FlatDb.from_nested(root, k, m, km.map)
    tree = op.flat_to_literal(flat)
    bases = np.frombuffer(seq.encode(), dtype=np.uint8)
    offsets = np.array([0, len(seq)], dtype=np.uint64)
    want = op.literal_place_batch(tree, bases, offsets)
    got, st = op.OraclePort(flat).place_batch(bases, offsets, want_stats=True)
    assert len(records_equal(got, want)) == 0
    tr = lit.Trace()
    lit.place_sequence("q", seq, tree, trace=tr)
    assert st[0]["n_matched"] == tr.query_kmers_len


def test_fasta_literal_semantics():
    txt = ">a b>c\nACGTnnacgt\r\n\n>second\n\n>third\nNNNN\n>z\nGG"
    recs = lit.sequence_content_by_channel(txt)
    assert recs == [("a bc", "ACGTACGT"), ("second", ""), ("third", ""), ("z", "GG")]
    assert lit.sequence_content_by_channel("ACGT\n>h\nAC\n") == []  # sequence before any header -> error, nothing sent


@pytest.mark.parametrize("case", [(60, 300, 8, 4, 0.0, 0), (80, 300, 10, 4, 0.4, 0), (120, 400, 12, 4, 0.0, 1), (50, 200, 17, 4, 0.3, 1)])
def test_leaves_only_oracle_equals_explicit_oracle(case):
    """CLS_SETS_LEAVES input (include/cls_place.h): the C oracle keeps such node sets lazily (a clade is a member iff
    a listed leaf lies below it).  Held to the explicit-set oracle -- the restatement of the reference -- on the same
    indexes: generated both ways, and derived from the explicit one by filtering on kind; records and counters."""
    nl, rl, k, m, cp, deep = case
    s = SynthDb(nl, rl, k, m, collapse_prob=cp, deep=deep)
    t = SynthDb(nl, rl, k, m, collapse_prob=cp, deep=deep, tips_only=True)
    assert t.flat.leaves_only and not s.flat.leaves_only
    derived = s.flat.to_leaves_only()
    assert np.array_equal(t.flat.kmer_hash, s.flat.kmer_hash) and np.array_equal(t.flat.kmer_node_off, derived.kmer_node_off)
    for j in range(0, s.flat.n_kmers, 97):  # same leaves per k-mer (any order)
        a, b = int(derived.kmer_node_off[j]), int(derived.kmer_node_off[j + 1])
        assert sorted(t.flat.node_ids[a:b]) == sorted(derived.node_ids[a:b])
    back = t.flat.to_explicit()
    for j in range(0, s.flat.n_kmers, 211):
        a, b = int(s.flat.kmer_node_off[j]), int(s.flat.kmer_node_off[j + 1])
        c, d = int(back.kmer_node_off[j]), int(back.kmer_node_off[j + 1])
        assert sorted(s.flat.node_ids[a:b]) == sorted(back.node_ids[c:d])
    bases, offsets, _ = s.reads(400, min(rl, 150), frac_random=0.05, err=0.02)
    for kw in (dict(), dict(remove_intersection=True), dict(min_match_coverage=1.0), dict(max_iterations=3)):
        want, wst = op.OraclePort(s.flat).place_batch(bases, offsets, op.make_params(**kw), threads=4, want_stats=True)
        for flat in (t.flat, derived):
            got, gst = op.OraclePort(flat).place_batch(bases, offsets, op.make_params(**kw), threads=4, want_stats=True)
            for f in ("status", "one", "rest", "levels", "clade_id"):
                assert (got[f] == want[f]).all(), (f, kw)
            for f in ("n_query_kmers", "n_matched", "n_with_root", "leaf_postings"):
                assert (gst[f] == wst[f]).all(), (f, kw)
