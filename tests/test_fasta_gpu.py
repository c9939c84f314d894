"""FASTA stage on the device against the literal oracle (`oracle_literal.sequence_content_by_channel`,
file_or_stdin.rs:76-116) directly, and against the host parser: identical records, bases, offsets and truncation
flag, on the fixed corner cases, on random texts built from the characters that matter, and on a file-sized input
placed end to end."""
import numpy as np
import pytest

from classeq2_amd import _abi, engine
from classeq2_amd.synth import SynthDb
from oracle import oracle_literal as lit
from oracle import oracle_port as op
from tests.helpers import records_equal
from tests.test_host_cpu import FASTA_CASES

pytestmark = pytest.mark.gpu


def _same(txt: bytes):
    want = engine.fasta_parse(txt)
    got = engine.fasta_parse(txt, device=0)
    assert got[0] == want[0], (txt[:80], got[0][:3], want[0][:3])
    assert np.array_equal(got[2], want[2]) and np.array_equal(got[1], want[1])
    assert got[3] == want[3]
    # ... and the device stage against the literal oracle itself (no transitive step through the host parser)
    headers, bases, off, truncated = got
    recs = [(headers[i].decode("utf-8"), bytes(bases[int(off[i]):int(off[i + 1])]).decode()) for i in range(len(headers))]
    try:
        text = txt.decode("utf-8")
    except UnicodeDecodeError as e:  # BufRead::lines() stops at the first line that is not UTF-8
        cut = txt.rfind(b"\n", 0, e.start) + 1
        lit_recs = lit.sequence_content_by_channel(txt[:cut].decode("utf-8"))
        assert recs == lit_recs[: len(recs)] and truncated  # records completed before the bad line were sent; the pending one is lost
        return want
    assert recs == lit.sequence_content_by_channel(text), txt[:80]
    return want


@pytest.mark.parametrize("txt", FASTA_CASES + [b"\n", b">", b"A", b"\r\n\r\n", b">h", b">h\n\xc3", b"\xe2\x82\xac\n>h\nAC", b">h\nA\xe2\x82\nC\n>g\nT"])
def test_fixed_cases(txt):
    _same(txt)


def test_random_texts_from_the_alphabet_that_matters():
    rng = np.random.default_rng(12)
    pieces = [b">", b"\n", b"\r", b"\r\n", b"A", b"c", b"G", b"t", b"N", b" ", b">x", b"\n>", b"\n\n", b"ACGTACGT", b"\xc3\xa9", b"\xe2\x82\xac",
              b"\xf0\x9f\x98\x80", b"\xff", b"\xc3", b"\x80", b"\xed\xa0\x80", b"\xc0\xaf", b"\xf4\x90\x80\x80", b"h1", b"-"]
    weights = np.array([6, 10, 2, 3, 8, 4, 8, 4, 2, 1, 3, 4, 2, 6, 0.6, 0.4, 0.3, 0.15, 0.15, 0.15, 0.1, 0.1, 0.1, 3, 1], dtype=float)
    weights /= weights.sum()
    n_trunc = 0
    for trial in range(300):
        n = int(rng.integers(0, 400))
        txt = b"".join(pieces[i] for i in rng.choice(len(pieces), size=n, p=weights))
        n_trunc += _same(txt)[3]
    assert n_trunc > 100  # mostly "sequence without header" / invalid UTF-8 stops
    # line-structured texts: mostly well-formed records, with the odd empty header, blank line, stray '>' or bad byte
    n_ok = n_rec = 0
    for trial in range(300):
        lines = [b">first"] if rng.random() < 0.8 else []
        for _ in range(int(rng.integers(0, 40))):
            u = rng.random()
            if u < 0.35:
                lines.append(b">" + (b"" if rng.random() < 0.03 else b"h%d" % int(rng.integers(100))) + (b">x" if rng.random() < 0.1 else b""))
            elif u < 0.9:
                body = bytes(rng.choice(np.frombuffer(b"ACGTacgtNn-", dtype=np.uint8), size=int(rng.integers(0, 90))))
                lines.append(body + (b"\xe2\x82\xac" if rng.random() < 0.02 else b"") + (b"\xff" if rng.random() < 0.01 else b""))
            else:
                lines.append(b"")
        txt = b"".join(l + (b"\r\n" if rng.random() < 0.3 else b"\n") for l in lines)
        if lines and rng.random() < 0.3:
            txt = txt.rstrip(b"\r\n")
        want = _same(txt)
        n_ok += not want[3]
        n_rec += len(want[0])
    assert n_ok > 100 and n_rec > 500


def test_chunk_boundaries_and_long_lines():
    """Lines longer than a 4096-byte chunk, headers and "\\r\\n" straddling chunk ends."""
    rng = np.random.default_rng(3)
    seq = bytes(rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), size=20000))
    for pad in range(4090, 4100):
        txt = b">" + b"h" * pad + b"\r\n" + seq + b"\n>second line>with>marks\r\n" + seq[:5000] + b"\r\n" + seq[5000:9000] + b"\n\n>tail\n"
        _same(txt)
    _same(b">a\n" + b"ACGT" * 100000)                      # one 400 kb line, no final newline
    _same(b"ACGT" * 3000 + b"\n>late\nAC\n")                 # sequence before the first header: nothing is emitted


def test_fasta_to_placements_on_the_device():
    s = SynthDb(200, 600, 11, 4)
    bases, offsets, _ = s.reads(30000, 150, err=0.02, frac_random=0.03)
    b = bases.reshape(-1, 150)
    rows = []
    for i in range(len(b)):
        rows.append(b">read %d some text\n" % i)
        rows.append(bytes(b[i][:70]) + b"\n" + bytes(b[i][70:]).lower() + (b"\r\n" if i % 3 else b"\n"))
    text = b"".join(rows)
    with engine.PlacementDb(s.flat, device=0) as db:
        headers, got, truncated = db.place_fasta_text(text)
    assert not truncated and len(headers) == len(b) and headers[7] == b"read 7 some text"
    want = op.OraclePort(s.flat).place_batch(bases, offsets, threads=8)
    assert len(records_equal(got, want)) == 0
