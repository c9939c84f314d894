"""CPU: the C-ABI library loads and exports what include/cls_place.h declares,
the host-side encoder validates its input, and the FASTA stage matches the
literal oracle.  No compute entry point is called (there is no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from classeq2_amd import _abi, engine
from classeq2_amd.flatdb import FlatDb
from classeq2_amd.synth import SynthDb
from oracle import oracle_literal as lit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    L = engine.lib()
    for header, exports in (("cls_place.h", engine.EXPORTS), ("cls_host.h", engine.HOST_EXPORTS), ("cls_service.h", engine.SERVICE_EXPORTS)):
        hdr = open(os.path.join(ROOT, "include", header)).read()
        declared = set(re.findall(r"^(?:int|void|const char\*)\s+(cls_[a-z0-9_]+)\(", hdr, flags=re.M))
        assert declared == set(exports), (header, declared ^ set(exports))
        for name in declared:
            assert hasattr(L, name), name
    assert b"gfx950" in L.cls_version()


def test_struct_layouts_match_header():
    assert C.sizeof(_abi.Node) == 32 and C.sizeof(_abi.Placement) == 24 and C.sizeof(_abi.QueryStats) == 24
    assert _abi.Placement.clade_id.offset == 16 and _abi.Placement.levels.offset == 12
    assert C.sizeof(_abi.DbDesc) == 96 and _abi.DbDesc.node_set_kind.offset == 88


def test_validate_accepts_generated_dbs():
    for cp in (0.0, 0.4):
        s = SynthDb(80, 300, 9, 4, collapse_prob=cp, id_stride=7, id_offset=3)
        engine.validate(s.flat)


def _flat():
    s = SynthDb(30, 200, 6, 3)
    return FlatDb.from_desc(s.flat.desc(), copy=True)


def test_validate_rejects_bad_input():
    f = _flat()
    f.nodes["first_child"][0] = len(f.nodes) + 5
    with pytest.raises(engine.ClsError) as e:
        engine.validate(f)
    assert e.value.code == -2
    f = _flat()
    f.nodes["id"][3] = f.nodes["id"][4]
    with pytest.raises(engine.ClsError, match="duplicate clade id"):
        engine.validate(f)
    f = _flat()
    f.kmer_hash[1] = f.kmer_hash[0]
    with pytest.raises(engine.ClsError, match="more than once"):
        engine.validate(f)
    f = _flat()
    f.k_size = 0
    with pytest.raises(engine.ClsError):
        engine.validate(f)
    f = _flat()
    f.kmer_node_off[2] = f.kmer_node_off[3] + 1
    with pytest.raises(engine.ClsError, match="monotone"):
        engine.validate(f)
    # two parents claim the same row
    f = _flat()
    r = int(np.nonzero(f.nodes["n_children"] > 0)[0][1])
    f.nodes["first_child"][r] = f.nodes["first_child"][0]
    with pytest.raises(engine.ClsError):
        engine.validate(f)


def test_create_without_gpu_fails_loudly():
    if engine.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(engine.ClsError) as e:
        engine.PlacementDb(_flat())
    assert e.value.code == -4 and "no CPU fallback" in e.value.msg


FASTA_CASES = [
    b">a b>c\nACGTnnacgt\r\n\n>second\n\n>third\nNNNN\n>z\nGG",
    b"ACGT\n>h\nAC\n",
    b">h1\nAC\nGT\n>h2\nacgtnryk\n",
    b">only\n",
    b"",
    b">\nAC\n>x\nGG\n",
    b">h\r\nAC\r\n>g\r\nTT\r",
    b">h\nAC\n>bad\xff\nGG\n>after\nTT\n",
    b">>>x>y\nA-C G.T*\n",
]


@pytest.mark.parametrize("txt", FASTA_CASES)
def test_fasta_parse_matches_literal(txt):
    headers, bases, off, truncated = engine.fasta_parse(txt)
    got = [(headers[i].decode("utf-8"), bytes(bases[int(off[i]):int(off[i + 1])]).decode()) for i in range(len(headers))]
    # the literal works on str; BufRead::lines() stops at the first non-UTF-8 line
    try:
        text = txt.decode("utf-8")
    except UnicodeDecodeError as e:
        cut = txt.rfind(b"\n", 0, e.start) + 1
        text = txt[:cut].decode("utf-8")
        want = lit.sequence_content_by_channel(text)
        # records completed before the bad line were already sent; the pending one is lost
        assert got == [w for w in want][: len(got)] and truncated
        return
    assert got == lit.sequence_content_by_channel(text)


def test_leaves_only_descriptor_validation():
    """cls_db_desc v2 (CLS_SETS_LEAVES): accepted for generated and derived indexes; an id that is not a LEAF clade of
    the tree is refused; a v1 descriptor (no node_set_kind) still validates."""
    s = SynthDb(60, 300, 9, 4, collapse_prob=0.3)
    t = SynthDb(60, 300, 9, 4, collapse_prob=0.3, tips_only=True)
    engine.validate(t.flat)
    engine.validate(s.flat.to_leaves_only())
    f = FlatDb.from_desc(t.flat.desc(), copy=True)
    assert f.leaves_only
    internal = int(f.nodes["id"][f.nodes["kind"] == _abi.KIND_NODE][0])
    f.node_ids[3] = internal
    with pytest.raises(engine.ClsError, match="LEAF"):
        engine.validate(f)
    f = FlatDb.from_desc(t.flat.desc(), copy=True)
    f.node_ids[0] = 2**63 + 12345  # no clade of the tree at all
    with pytest.raises(engine.ClsError, match="LEAF"):
        engine.validate(f)
    d = s.flat.desc()
    d.abi_version = 1
    d.node_set_kind = 77  # ignored by a v1 caller's layout
    assert engine.lib().cls_db_validate(C.byref(d)) == 0
    d.abi_version = 2
    assert engine.lib().cls_db_validate(C.byref(d)) != 0
    d.abi_version = 3
    assert engine.lib().cls_db_validate(C.byref(d)) != 0


@pytest.mark.parametrize("case", [(40, 1500, 21, 0, 0.0, 0, 1), (60, 300, 9, 4, 0.3, 0, 0), (80, 300, 12, 4, 0.0, 1, 1), (50, 200, 8, 4, 0.4, 0, 2),
                                  (64, 400, 11, 4, 0.0, 2, 0), (30, 200, 7, 3, 0.5, 0, 1)])
def test_encoded_index_answers_membership_like_the_node_sets(case):
    """The encoder (tip sets shared between k-mers, split trees, hash table, sorted-list postings) checked on the host,
    no device: for every k-mer of the index and a sample of clades, `clade in nodes(k-mer)` answered from the ENCODED
    index (cls_db_debug_members walks the same tables the kernels read) equals the answer of the node sets themselves."""
    from tests.helpers import drop_random_nodes, truncate_random_sets
    nl, rl, k, m, cp, deep, mode = case
    s = SynthDb(nl, rl, k, m, collapse_prob=cp, deep=deep, seed_tree=203, seed_refseq=204)
    flats = [s.flat, s.flat.to_leaves_only()]
    if mode == 1:
        flats = [truncate_random_sets(s.flat, 0.1, seed=202)]
    elif mode == 2:
        flats = [drop_random_nodes(s.flat, 0.15, seed=5)]
    L = engine.lib()
    L.cls_db_debug_members.argtypes = [C.POINTER(_abi.DbDesc), C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.cls_db_debug_members.restype = C.c_int
    rng = np.random.default_rng(1)
    truth_flat = flats[0]
    ids = truth_flat.nodes["id"]
    off = truth_flat.kmer_node_off.astype(np.int64)
    sample = np.arange(0, truth_flat.n_kmers, max(1, truth_flat.n_kmers // 4000))  # (a few thousand k-mers keep the test in seconds)
    nk = len(sample)
    per = 12
    hashes = np.repeat(truth_flat.kmer_hash[sample], per)
    clades = np.empty(nk * per, dtype=np.uint64)
    want = np.empty(nk * per, dtype=np.uint8)
    for j, kj in enumerate(sample):
        members = truth_flat.node_ids[off[kj]:off[kj + 1]]
        pick = np.concatenate([[ids[0]], rng.choice(ids, per - 1 - min(4, len(members))), rng.choice(members, min(4, len(members))) if len(members) else []])
        clades[j * per:(j + 1) * per] = pick
        want[j * per:(j + 1) * per] = np.isin(pick, members)
    for flat in flats:
        out = np.full(nk * per, 9, dtype=np.uint8)
        d = flat.desc()
        assert L.cls_db_debug_members(C.byref(d), hashes.ctypes.data, clades.ctypes.data, nk * per, out.ctypes.data) == 0
        bad = np.nonzero(out != want)[0]
        assert len(bad) == 0, (len(bad), int(bad[0]), int(out[bad[0]]), int(want[bad[0]]))
    absent = np.array([1, 2, 3], dtype=np.uint64)
    out = np.zeros(3, dtype=np.uint8)
    d = flats[0].desc()
    assert L.cls_db_debug_members(C.byref(d), absent.ctypes.data, ids[:3].copy().ctypes.data, 3, out.ctypes.data) == 0 and (out == 2).all()


@pytest.mark.parametrize("case", [(300, 400, 9, 4, 0.0, 0), (200, 300, 8, 4, 0.0, 1), (120, 300, 12, 4, 0.0, 2), (100, 300, 9, 4, 0.3, 0)])
def test_mask_halves_describe_the_same_tips(case):
    """The second copy of the split records (parts that span at most 32 rows as bit masks) checked on the host against
    the first copy, half by half (cls_db_debug_mask_halves); binary trees and trees with polytomies."""
    nl, rl, k, m, cp, deep = case
    s = SynthDb(nl, rl, k, m, collapse_prob=cp, deep=deep, seed_tree=211, seed_refseq=212)
    L = engine.lib()
    L.cls_db_debug_mask_halves.argtypes = [C.POINTER(_abi.DbDesc), C.c_void_p]
    L.cls_db_debug_mask_halves.restype = C.c_int
    for flat in (s.flat, s.flat.to_leaves_only()):
        counts = np.zeros(3, dtype=np.uint64)
        d = flat.desc()
        assert L.cls_db_debug_mask_halves(C.byref(d), counts.ctypes.data) == 0, counts
        assert counts[0] > 0 and counts[1] > 0 and counts[2] > 0, counts
        assert counts[1] + counts[2] == 2 * counts[0]


def test_tuning_knobs_are_explicit():
    """Experiment knobs go through cls_set_tuning (the library reads no environment variable on its own): known names
    are accepted, unknown ones refused; cls_tuning_from_env is an explicit call."""
    engine.set_tuning("order_mode", 0)
    engine.set_tuning("no_fat_direct", 0)
    with pytest.raises(engine.ClsError, match="unknown knob"):
        engine.set_tuning("no_such_knob", 1)
    os.environ["CLS_ORDER_BLOCK_SHIFT"] = "2"  # the default value: harmless
    engine.tuning_from_env()
    del os.environ["CLS_ORDER_BLOCK_SHIFT"]
    import subprocess
    src = open(os.path.join(os.path.dirname(engine.LIB_PATH), "cls_kernels.hip")).read() + open(os.path.join(os.path.dirname(engine.LIB_PATH), "cls_db.cpp")).read()
    assert "getenv" not in src  # only cls_tuning_from_env (cls_api.cpp) touches the environment
