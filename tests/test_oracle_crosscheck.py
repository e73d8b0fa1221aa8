"""Cross-checks the C++ oracle against the independent pure-Python restatement (oracle/pyref.py) and the
worked vectors of SURVEY.md Appendix B ([derived], not reference output): words, bucket order, file bytes.

The file format is parity-UNPINNED by the reference's own tests (it has none that touch serialization); these
tests make two independent restatements of the cited Serialize impls agree byte for byte.
"""
import random

import pytest

from oracle import Oracle
from oracle.pyref import PyCBL, params, seq_words


def _rand_seq(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(alphabet) for _ in range(n))


# ---- SURVEY.md Appendix B ------------------------------------------------------------------------------
def test_appendix_b2():
    o = Oracle(7, 14)
    o.insert_seq(b"ACGTACGTAC")
    assert o.seq_words(b"ACGTACGTAC") == [0x38F7, 0x3CF5, 0x3C73, 0x30F9]
    want = bytes.fromhex("0004" "fb0f03" "00010109" "fb8f03" "00010107" "fbc703" "00010103" "fbcf03" "00010105")
    assert o.serialize() == want


def test_appendix_b3():
    seq = b"GATTACAGATTACATTTGGGACCA"
    o = Oracle(7, 14)
    o.insert_seq(seq)
    words = "2657 2655 2653 2651 265d 265b 2659 2657 2455 2553 2551 4ab0 abd2 abf0 abfc 6bfa 5bf8 53f6".split()
    assert o.seq_words(seq) == [int(w, 16) for w in words]
    want = bytes.fromhex(
        "0009fb450200010105fb5502000201030101fb650200070107010501030101010d010b0109fbab0400010100fb3f0500010106"
        "fbbf0500010108fbbf060001010afbbd0a00010102fbbf0a00020100010c"
    )
    assert o.serialize() == want


def test_appendix_b4_trie_bytes():
    """BYTES=2 trie {0x0102, 0x0103, 0x0701} -> 01 | 02 01 07 02 | 02 02 03 00 | 01 01 00 | 03."""
    p = PyCBL(7, 2)  # SUFFIX_BITS = 18-2 = 16 -> BYTES = 2
    assert p.P["BYTES"] == 2
    p.buckets[0] = ["trie", [0x0102, 0x0103, 0x0701]]
    blob = p.serialize()
    assert blob == bytes.fromhex("00" "01" "00" "01" "0201070202020300010100" "03")
    o = Oracle(7, 2)
    o.load(blob)
    assert o.serialize() == blob and o.count() == 3


# ---- words ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k,pb", [(7, 14), (25, 24), (31, 24), (31, 28), (59, 28)])
@pytest.mark.parametrize("canonical", [False, True])
def test_words_match_pyref(k, pb, canonical):
    rng = random.Random(k * 3 + canonical)
    o = Oracle(k, pb, canonical)
    P = params(k, pb)
    for n in (k, k + 1, 150, 400):
        seq = _rand_seq(rng, n)
        want = seq_words(seq, P, canonical)
        assert o.seq_words(seq) == want
        assert o.seq_words(seq, brute_force=True) == want


@pytest.mark.parametrize("canonical", [False, True])
def test_words_multichunk_and_non_acgt(canonical):
    """>2048 k-mers (chunk re-seeding, src/cbl.rs:239-243) and skipped non-ACGT bytes (src/kmer.rs:133-135,
    incl. one inside the first K bytes of a chunk, which yields a short first k-mer)."""
    rng = random.Random(9)
    k, pb = 31, 24
    seq = bytearray(_rand_seq(rng, 5000, b"ACGTacgt"))
    for pos in (3, 40, 41, 2048 + 5, 2048 + 200, 4999):
        seq[pos] = ord("N")
    seq = bytes(seq)
    o = Oracle(k, pb, canonical)
    want = seq_words(seq, params(k, pb), canonical)
    assert o.seq_words(seq) == want
    assert o.seq_words(seq, brute_force=True) == want


# ---- whole index: bucket order, Vec->Trie at 1024, file bytes ------------------------------------------
@pytest.mark.parametrize(
    "k,pb,n,canonical",
    [(7, 14, 3000, False), (9, 4, 40000, False), (9, 4, 40000, True), (25, 24, 4000, False), (31, 24, 3000, True),
     (59, 28, 2500, False), (11, 8, 60000, False)],
)
def test_index_bytes_match_pyref(k, pb, n, canonical):
    rng = random.Random(k + pb)
    o, p = Oracle(k, pb, canonical), PyCBL(k, pb, canonical)
    for _ in range(3):  # several insert_seq calls incl. repeats -> duplicates across calls
        seq = _rand_seq(rng, n // 3)
        for s in (seq, seq[: len(seq) // 2]):
            o.insert_seq(s)
            p.insert_seq(s)
    assert o.count() == p.count()
    blob = o.serialize()
    assert blob == p.serialize()
    kinds = {b[0] for b in p.buckets.values()}
    if (k, pb) in ((9, 4), (11, 8)):
        assert "trie" in kinds  # the threshold path is really exercised
    # load -> serialize is the identity (src/wordset/mod.rs:398-437)
    o2 = Oracle(k, pb, canonical)
    o2.load(blob)
    assert o2.serialize() == blob and o2.count() == o.count()


def test_trailing_bytes_rejected():
    o = Oracle(7, 14)
    o.insert_seq(b"ACGTACGTAC")
    import oracle as _o

    with pytest.raises(_o.OracleError):
        Oracle(7, 14).load(o.serialize() + b"\x00")


def test_insert_after_load_is_resume():
    """`cbl insert` (examples/cbl.rs:230-249): load then insert == inserting everything in one go."""
    rng = random.Random(77)
    s1, s2 = _rand_seq(rng, 20000), _rand_seq(rng, 20000)
    a = Oracle(9, 4)
    a.insert_seq(s1)
    a.insert_seq(s2)
    b = Oracle(9, 4)
    b.insert_seq(s1)
    c = Oracle(9, 4)
    c.load(b.serialize())
    c.insert_seq(s2)
    assert c.serialize() == a.serialize()


# ---- |= (src/wordset/set_ops.rs:123-157, src/trievec/set_ops.rs:43-71) --------------------------------
@pytest.mark.parametrize("k,pb,n", [(7, 14, 2000), (9, 4, 9000), (9, 4, 40000), (11, 8, 60000)])
def test_merge_matches_pyref(k, pb, n):
    rng = random.Random(n + k)
    s1, s2 = _rand_seq(rng, n), _rand_seq(rng, n // 2) + _rand_seq(rng, 50)
    o1, o2, p1, p2 = Oracle(k, pb), Oracle(k, pb), PyCBL(k, pb), PyCBL(k, pb)
    o1.insert_seq(s1), p1.insert_seq(s1)
    o2.insert_seq(s2), p2.insert_seq(s2)
    o1.merge(o2)
    p1.merge(p2)
    assert o1.serialize() == p1.serialize()
    assert set(o1.iter_words()) == set(seq_words(s1, params(k, pb), False)) | set(seq_words(s2, params(k, pb), False))
