"""Pins the CPU oracle (oracle/) against every known-answer test the reference holds for the insert path.

Each test names the reference test it restates (/root/reference/...). Values are literal copies of the
reference's expected answers (data), not of its code.
"""
import ctypes as C
import random

import numpy as np
import pytest

import oracle
from oracle import Oracle, necklace_pos, rev_comp, revert_necklace_pos

L = oracle.lib()


def _queue(bits, width, reverse, word):
    return L.oracle_queue_new(bits, width, int(reverse), word & (2**64 - 1), word >> 64)


def _queue_get(q):
    lo, hi, pos = C.c_uint64(), C.c_uint64(), C.c_uint32()
    L.oracle_queue_get(q, C.byref(lo), C.byref(hi), C.byref(pos))
    return lo.value | (hi.value << 64), pos.value


# ---- src/necklace/queue.rs:133-147 --------------------------------------------------------------------
def test_necklace_queue_kat():
    q = _queue(8, 4, False, 0b10010110)
    assert _queue_get(q) == (0b00101101, 1)
    L.oracle_queue_insert(q, 0)
    assert _queue_get(q) == (0b00001011, 8 - 2)
    L.oracle_queue_free(q)


def test_necklace_queue_rev_kat():
    q = _queue(8, 4, True, 0b10010110)
    assert _queue_get(q) == (0b00101101, 1)
    L.oracle_queue_insert(q, 1)
    assert _queue_get(q) == (0b00101111, 2)
    L.oracle_queue_free(q)


def test_bruteforce_matches_queue_kats():
    assert necklace_pos(0b10010110, 8) == (0b00101101, 1)
    assert necklace_pos(0b00101100, 8) == (0b00001011, 6)
    assert necklace_pos(0b11001011, 8) == (0b00101111, 2)


# ---- src/necklace/minimizer.rs:110-166 ----------------------------------------------------------------
def _min_pos(q):
    out = (C.c_uint32 * 16)()
    n = L.oracle_lmq_min_pos(q, out, 16)
    return list(out[:n])


def test_lex_min_queue_insert_full():
    W = 4
    q = L.oracle_lmq_new(W)
    L.oracle_lmq_insert_full(q, (C.c_uint32 * 4)(2, 1, 2, 1))
    assert _min_pos(q) == [W - 3, W - 1]
    L.oracle_lmq_free(q)


def test_lex_min_queue_insert():
    W = 4
    q = L.oracle_lmq_new(W)
    for val, expect in [(3, [W - 1]), (1, [W - 1]), (2, [W - 2]), (3, [W - 3]), (1, [W - 4, W - 1]), (2, [W - 2])]:
        L.oracle_lmq_insert(q, val)
        assert _min_pos(q) == expect
    L.oracle_lmq_free(q)


# ---- src/necklace/mod.rs:45-98 (properties; N reduced from 1e6, seeded) -------------------------------
def test_necklace_revert():
    rng = random.Random(1)
    for _ in range(20000):
        w = rng.getrandbits(31)
        n, p = necklace_pos(w, 31)
        assert revert_necklace_pos(n, p, 31) == w


def test_same_necklace_fwd_rev_queue():
    rng = random.Random(2)
    for _ in range(5000):
        w = rng.getrandbits(31)
        for rev in (False, True):
            q = _queue(31, 31 - 9 + 1, rev, w)
            assert _queue_get(q) == necklace_pos(w, 31)
            L.oracle_queue_free(q)


def test_same_necklace_periodic_words():
    rng = random.Random(3)
    for _ in range(5000):
        w = rng.getrandbits(30)
        w = (w << 30) | w
        q = _queue(60, 50, False, w)
        assert _queue_get(q) == necklace_pos(w, 60), bin(w)
        L.oracle_queue_free(q)


@pytest.mark.parametrize("bits,width", [(10, 2), (14, 6), (50, 42), (62, 54), (118, 110)])
def test_streaming_queue_equals_bruteforce(bits, width):
    """insert_full + insert2 stream (the path src/cbl.rs:277-287 drives), fwd and REVERSE, incl. degenerate inputs."""
    rng = random.Random(bits)
    mask = (1 << bits) - 1
    for mode in ("rand", "zeros", "period2", "sparse"):
        w = rng.getrandbits(bits)
        qf, qr = _queue(bits, width, False, w), _queue(bits, width, True, w)
        wf = wr = w
        for _ in range(300):
            x = {"rand": rng.getrandbits(2), "zeros": 0, "period2": 1, "sparse": int(rng.random() < 0.05)}[mode]
            L.oracle_queue_insert2(qf, x)
            L.oracle_queue_insert2(qr, x)
            wf = ((wf << 2) & mask) | x
            wr = (wr >> 2) | (x << (bits - 2))
            assert _queue_get(qf) == necklace_pos(wf, bits)
            assert _queue_get(qr) == necklace_pos(wr, bits)
        L.oracle_queue_free(qf)
        L.oracle_queue_free(qr)


# ---- src/kmer.rs:355-413 ------------------------------------------------------------------------------
def _pack(s):
    x = 0
    for ch in s:
        x = (x << 2) | L.oracle_nuc_code(ch)
    return x


def _unpack(x, k):
    return bytes(b"ACTG"[(x >> (2 * (k - 1 - i))) & 3] for i in range(k))


def test_nuc_code_table():  # src/kmer.rs:11-24
    for b in range(256):
        want = {65: 0, 97: 0, 67: 1, 99: 1, 84: 2, 116: 2, 71: 3, 103: 3}.get(b, -1)
        assert L.oracle_nuc_code(b) == want


def test_rc_kats():
    assert _unpack(rev_comp(_pack(b"ATCG"), 4), 4) == b"CGAT"
    assert _unpack(rev_comp(_pack(b"CATAATCCAGC"), 11), 11) == b"GCTGGATTATG"


def test_rc_rc_identity():
    for k in (3, 7, 15, 31, 59):
        rng = random.Random(k)
        for _ in range(2000):
            x = rng.getrandbits(2 * k)
            assert rev_comp(rev_comp(x, k), k) == x


# ---- src/trie.rs:227-261, src/sliced_int.rs:143-169 ---------------------------------------------------
def _be(b):  # 3 big-endian bytes -> int
    return (b[0] << 16) | (b[1] << 8) | b[2]


def test_trie_contains_and_iter_order():
    t = L.oracle_trievec_new()
    L.oracle_trievec_as_trie(t, 3)
    for b in ([9, 9, 9], [1, 1, 1], [1, 2, 4], [1, 2, 3], [7, 7, 7]):
        assert L.oracle_trievec_insert(t, _be(b), 0, 3) == 1
    assert L.oracle_trievec_insert(t, _be([1, 2, 3]), 0, 3) == 0
    for b in ([1, 2, 3], [1, 2, 4], [7, 7, 7]):
        assert L.oracle_trievec_contains(t, _be(b), 0, 3)
    for b in ([1, 2, 1], [3, 3, 3]):
        assert not L.oracle_trievec_contains(t, _be(b), 0, 3)
    lo = np.zeros(8, dtype=np.uint64)
    n = L.oracle_trievec_iter(t, 3, lo.ctypes.data, None, 8)
    assert [int(x) for x in lo[:n]] == [_be(b) for b in ([1, 1, 1], [1, 2, 3], [1, 2, 4], [7, 7, 7], [9, 9, 9])]
    L.oracle_trievec_free(t)


def test_vec_bucket_keeps_first_occurrence_order():  # src/trievec/mod.rs:81-87
    t = L.oracle_trievec_new()
    for x in (442, 631, 363, 631, 777, 123, 442):
        L.oracle_trievec_insert(t, x, 0, 3)
    lo = np.zeros(8, dtype=np.uint64)
    n = L.oracle_trievec_iter(t, 3, lo.ctypes.data, None, 8)
    assert [int(x) for x in lo[:n]] == [442, 631, 363, 777, 123]
    L.oracle_trievec_free(t)


# ---- src/bitvector/mod.rs:148-187, src/ffi.rs:29-39 ---------------------------------------------------
def test_bitvector_rank_and_iter():
    N, BITS = 10000, 20
    b = L.oracle_bv_new(1 << BITS)
    for i in range(0, 2 * N, 2):
        assert L.oracle_bv_insert(b, i) == 1
    assert L.oracle_bv_insert(b, 0) == 0
    for i in range(0, 2 * N, 2):
        assert L.oracle_bv_contains(b, i) and not L.oracle_bv_contains(b, i + 1)
        assert L.oracle_bv_rank(b, i) == i // 2
    L.oracle_bv_free(b)
    b = L.oracle_bv_new(1 << BITS)
    for i in (1, 3, 42, 101010, (1 << BITS) - 1):
        L.oracle_bv_insert(b, i)
    out = (C.c_uint64 * 8)()
    n = L.oracle_bv_iter(b, out, 8)
    assert list(out[:n]) == [1, 3, 42, 101010, (1 << BITS) - 1]
    L.oracle_bv_free(b)


def test_tiered_insert_get():
    t = L.oracle_tv_new()
    for i in range(5):
        L.oracle_tv_insert(t, i, i)
    assert [L.oracle_tv_get(t, i) for i in range(5)] == [0, 1, 2, 3, 4]
    L.oracle_tv_free(t)
    # mid-inserts against a plain list model, across block boundaries
    rng = random.Random(5)
    t, model = L.oracle_tv_new(), []
    for v in range(5000):
        i = rng.randint(0, len(model))
        model.insert(i, v)
        L.oracle_tv_insert(t, i, v)
    assert L.oracle_tv_len(t) == len(model)
    assert [L.oracle_tv_get(t, i) for i in range(len(model))] == model
    L.oracle_tv_free(t)


# ---- src/wordset/mod.rs:451-533 -----------------------------------------------------------------------
def test_wordset_insert_contains():
    N = 20000
    v0 = list(range(0, 2 * N, 2))
    random.Random(42).shuffle(v0)
    w = L.oracle_ws_new(24, 8)
    for i in v0:
        assert L.oracle_ws_insert(w, i) == 1
    assert L.oracle_ws_count(w) == N
    for i in v0:
        assert L.oracle_ws_contains(w, i) and not L.oracle_ws_contains(w, i + 1)
        assert L.oracle_ws_insert(w, i) == 0
    L.oracle_ws_free(w)


def test_wordset_batch():
    N = 20000
    v0 = np.arange(0, 2 * N, 2, dtype=np.uint64)
    w = L.oracle_ws_new(24, 8)
    L.oracle_ws_insert_batch(w, v0.ctypes.data, len(v0))
    assert L.oracle_ws_count(w) == N
    assert all(L.oracle_ws_contains(w, int(i)) for i in v0[::97])
    assert not any(L.oracle_ws_contains(w, int(i) + 1) for i in v0[::97])
    L.oracle_ws_free(w)


def test_wordset_iter_kat():
    SB = 8
    w = L.oracle_ws_new(24, SB)
    vals = [1, 42, (1 << SB) - 1, (1 << SB) + 10, 10 * (1 << SB) + 10]
    for v in vals:
        L.oracle_ws_insert(w, v)
    out = (C.c_uint64 * 8)()
    n = L.oracle_ws_iter(w, out, 8)
    assert list(out[:n]) == vals
    L.oracle_ws_free(w)


# ---- src/cbl.rs:591-773 (N reduced; seeded) -----------------------------------------------------------
def _rand_seq(rng, n):
    return bytes(rng.choice(b"ACGT") for _ in range(n))


def _kmers(seq, k):
    mask = (1 << (2 * k)) - 1
    x, out = 0, []
    for i, ch in enumerate(seq):
        x = ((x << 2) | L.oracle_nuc_code(ch)) & mask
        if i >= k - 1:
            out.append(x)
    return out


@pytest.mark.parametrize("k,pb", [(59, 24), (31, 24), (25, 24), (7, 14)])
def test_batch_equals_single_inserts(k, pb):
    """test_batch_operations: insert_seq (queue path) then per-k-mer contains (brute-force get_word) all true."""
    rng = random.Random(k)
    seq = _rand_seq(rng, 6000)
    a = Oracle(k, pb)
    a.insert_seq(seq)
    kms = _kmers(seq, k)
    assert all(a.contains_kmer(x) for x in kms)
    assert a.count() == len(set(kms))
    b = Oracle(k, pb)
    fresh = [b.insert_kmer(x) for x in kms]
    assert sum(fresh) == len(set(kms))
    assert sorted(a.iter_words()) == sorted(b.iter_words())
    other = _kmers(_rand_seq(rng, 300), k)
    assert [a.contains_kmer(x) for x in other] == [x in set(kms) for x in other]


@pytest.mark.parametrize("k,pb", [(59, 24), (31, 24), (7, 14)])
def test_canonical_batch(k, pb):
    rng = random.Random(100 + k)
    seq = _rand_seq(rng, 5000)
    a = Oracle(k, pb, canonical=True)
    a.insert_seq(seq)
    for x in _kmers(seq, k):
        assert a.contains_kmer(x) and a.contains_kmer(rev_comp(x, k))
    assert a.seq_words(seq) == a.seq_words(seq, brute_force=True)


def test_iter_roundtrip():  # test_iter: recover_kmer . get_word = id
    a = Oracle(59, 24)
    kmers = list(range(0, 1000, 7))
    for x in kmers:
        a.insert_kmer(x)
    assert sorted(a.kmer_of_word(w) for w in a.iter_words()) == kmers


def test_short_sequence_rejected():  # src/cbl.rs:329-334
    with pytest.raises(oracle.OracleError, match="smaller than K"):
        Oracle(31, 24).insert_seq(b"ACGT")


def test_t_too_small_is_impossible_by_construction():
    """src/cbl.rs:87-91 (K=31 needs 68 bits): the oracle picks the 128-bit word type itself, like build.rs:34-41."""
    a = Oracle(31, 24)
    a.insert_seq(b"ACGT" * 20)
    assert max(a.iter_words()).bit_length() <= 68
