"""GPU parity tests proper: the HIP path, called through the C ABI (include/cblx.h via cbl_amd.CBL), against the
CPU oracle on the same seeded inputs. Bit-exact: words, bucket contents and order, serialized index bytes.

Mirrors the reference's own test shapes (/root/reference/src/cbl.rs:664-683 batch == single, :726-761 canonical,
src/wordset/mod.rs:451-533) plus the edge cases of SURVEY.md §7 (non-ACGT bytes, multi-chunk sequences, short
sequences, duplicates, Vec->Trie threshold, huge buckets, incremental inserts, load/insert, merge).
"""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

torch = pytest.importorskip("torch")

import cbl_amd  # noqa: E402
from cbl_amd import synth  # noqa: E402
from oracle import Oracle  # noqa: E402


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")


def _spawn_context():
    """Context for the tests that start several ranks on this one GPU. The parent first gives back what earlier tests left cached
    (contexts not yet collected, torch's caching allocator): with that memory held, five K = 59 ranks at PREFIX_BITS >= 26 spent
    115-157 s in allocation retries where they take 13 s on a free GPU."""
    import gc

    import torch.multiprocessing as mp

    gc.collect()
    torch.cuda.empty_cache()
    return mp.get_context("spawn")


def _rand_seq(rng, n, alphabet=b"ACGT"):
    return bytes(rng.choice(alphabet) for _ in range(n))


def _concat(seqs):
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(s) for s in seqs])
    return bases, offsets


def _hi_tensor(g, n):
    hb = g.consts()["hi_bytes"]
    return None if hb == 0 else torch.zeros(n, dtype=torch.uint8 if hb == 1 else torch.int64, device="cuda")


def _hi_from_words(g, words):
    hb = g.consts()["hi_bytes"]
    if hb == 0:
        return None
    a = np.array([w >> 64 for w in words], dtype=np.uint64)
    return torch.from_numpy(a.astype(np.uint8) if hb == 1 else a.astype(np.int64)).cuda()


def _words_from(lo, hi):
    lo = lo.cpu().numpy().astype(np.uint64)
    if hi is None:
        return [int(a) for a in lo]
    hi = hi.cpu().numpy().astype(np.uint64)
    return [int(a) | (int(b) << 64) for a, b in zip(lo, hi)]


def _gpu_words(g, seqs):
    bases, offsets = _concat(seqs)
    nmax = int(len(bases))
    pad = (-len(bases)) % 16 + 16
    d_b = torch.from_numpy(np.concatenate([bases, np.zeros(pad, np.uint8)])).cuda()
    d_o = torch.from_numpy(offsets.astype(np.int64)).cuda()
    d_lo = torch.zeros(nmax + 1, dtype=torch.int64, device="cuda")
    d_hi = _hi_tensor(g, nmax + 1)
    n = g.seq_words_device(d_b, d_o, len(seqs), d_lo, d_hi, nmax)
    return _words_from(d_lo[:n], None if d_hi is None else d_hi[:n])


def _oracle_words(o, seqs):
    out = []
    for s in seqs:
        out += o.seq_words(s)
    return out


def _check_index(g, o):
    assert g.count() == o.count()
    assert g.num_buckets() == o.n_buckets()
    gb, ob = g.serialize(), o.serialize()
    if gb != ob:
        # locate the first differing bucket for the failure message
        o2 = cbl_amd.CBL(g.k, g.prefix_bits, canonical=g.is_canonical())
        o2.load(ob)
        for (p1, k1, s1), (p2, k2, s2) in zip(g.buckets(), o2.buckets()):
            assert (p1, k1) == (p2, k2), f"bucket header differs: gpu {(p1, k1, len(s1))} oracle {(p2, k2, len(s2))}"
            assert s1 == s2, f"bucket {p1:#x} kind {k1}: contents/order differ (len {len(s1)} vs {len(s2)})"
        assert False, "serialized bytes differ but buckets equal?"


# ---- KRN-1: words ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k,pb", [(7, 14), (25, 24), (29, 24), (31, 24), (31, 28), (33, 24), (59, 28), (31, 2), (31, 5)])
@pytest.mark.parametrize("canonical", [False, True])
def test_words_match_oracle(k, pb, canonical):
    _need_gpu()
    rng = random.Random(k * 7 + canonical)
    seqs = [_rand_seq(rng, n) for n in (k, k + 1, 150, 150, 151, 400, 2 * k, 3000, 150)]
    g = cbl_amd.CBL(k, pb, canonical=canonical)
    o = Oracle(k, pb, canonical)
    assert _gpu_words(g, seqs) == _oracle_words(o, seqs)


@pytest.mark.parametrize("k", [31, 59])
@pytest.mark.parametrize("canonical", [False, True])
def test_words_multichunk_non_acgt_and_degenerate(k, canonical):
    """>2048 k-mers per sequence (chunk re-seeding, src/cbl.rs:239-243), skipped non-ACGT bytes incl. inside the first
    K bytes of a chunk (src/kmer.rs:133-135), lower case, homopolymers and short-period repeats."""
    _need_gpu()
    rng = random.Random(k + 99)
    s1 = bytearray(_rand_seq(rng, 9000, b"ACGTacgt"))
    for pos in (3, 40, 41, 2048 + 5, 2048 + 200, 4096 + k - 2, 8999):
        s1[pos] = ord("N")
    seqs = [bytes(s1), b"A" * 300, b"G" * 200, b"AC" * 150, b"ACGT" * 100, b"N" * k + _rand_seq(rng, 100),
            _rand_seq(rng, 100) + b"NNNN", b"acgtn" * 60, _rand_seq(rng, 5000)]
    g = cbl_amd.CBL(k, 24, canonical=canonical)
    o = Oracle(k, 24, canonical)
    assert _gpu_words(g, seqs) == _oracle_words(o, seqs)


def test_dirty_chunk_next_to_clean_ones_of_the_same_shape():
    """Regression (found by the fuzzer, round 2): the encode kernel's fast path for tiles whose chunks all have one shape
    must not take a tile whose SECOND chunk holds a non-ACGT byte — that chunk is written by the scalar kernel, and counting
    it twice in the fused first-pass histogram corrupted the partition of the NEXT insert's batch."""
    _need_gpu()
    rng = random.Random(9)
    for k, pb in ((15, 25), (31, 24), (25, 12)):
        L = k + 37
        reads = [_rand_seq(rng, L) for _ in range(40)]
        # a dirty read with as many k-mers as a clean one: one N in front of the first K bytes' worth... the k-mer count of a
        # dirty chunk is 1 + valid bytes after the first K, so put the invalid byte among the first K and add one base
        for pos in (1, 5, 9, 17, 33):
            r = bytearray(_rand_seq(rng, L + 1))
            r[2] = ord("N")
            reads[pos] = bytes(r)
        g, o = cbl_amd.CBL(k, pb), Oracle(k, pb)
        for _ in range(3):  # the second and third inserts go through the incremental path
            bases, offsets = _concat(reads)
            g.insert_seqs(bases, offsets)
            o.insert_seqs(bases, offsets)
            _check_index(g, o)
            reads = [_rand_seq(rng, L) for _ in range(10)] + reads[:20]


def test_short_sequence_rejected():
    _need_gpu()
    g = cbl_amd.CBL(31, 24)
    with pytest.raises(cbl_amd.CblxError, match="smaller than K") as e:
        g.insert_seq(b"ACGT")
    assert e.value.code == cbl_amd.ESHORT
    with pytest.raises(cbl_amd.CblxError):
        cbl_amd.CBL(30, 24)  # K must be odd (build.rs:18-24)


def test_bad_offsets_in_a_big_host_batch_enqueue_nothing():
    """cblx_insert_seqs on a batch large enough for the bulk path (>= 1 MiB of bases): the offsets are validated while the
    bases are already on the wire; a short sequence or a descending offset must still leave the index and its pending queue
    exactly as they were (src/cbl.rs:329-334 panics before anything is inserted)."""
    _need_gpu()
    k, L, n = 31, 150, 20_000
    bases, offsets = synth.reads(7, n, L)
    g = cbl_amd.CBL(k, 24)
    o = Oracle(k, 24)
    short = offsets.copy()
    short[n // 2 + 1:] -= np.uint64(L - 10)  # sequence n/2 keeps 10 bases
    with pytest.raises(cbl_amd.CblxError, match="smaller than K") as e:
        g.insert_seqs(bases, short)
    assert e.value.code == cbl_amd.ESHORT
    desc = offsets.copy()
    desc[100] = desc[99] - np.uint64(1)
    with pytest.raises(cbl_amd.CblxError, match="non-decreasing"):
        g.insert_seqs(bases, desc)
    g.flush()
    assert g.count() == 0
    g.insert_seqs(bases, offsets)  # the same buffers, valid offsets: the whole batch goes in
    o.insert_seqs(bases, offsets)
    _check_index(g, o)


# ---- whole path: bucket contents, order, serialized bytes -------------------------------------------------
@pytest.mark.parametrize(
    "k,pb,nreads,L,canonical",
    [
        (25, 24, 2000, 150, False),   # BASELINE cfg 1 shape, scaled
        (31, 24, 2000, 150, False),   # cfg 2 shape
        (31, 24, 1500, 150, True),
        (31, 28, 1500, 150, False),   # cfg 3 shape
        (59, 28, 1000, 250, False),   # cfg 4 shape (wide k-mer, 97-bit suffix)
        (59, 28, 600, 250, True),
        (33, 24, 1000, 150, False),   # wide k-mer, narrow suffix
        (7, 14, 300, 60, False),
        (31, 2, 800, 150, False),     # 68-bit word with a 66-bit suffix: the hi bits are suffix bits, the wide layout takes it
        (31, 3, 800, 150, True),
        (31, 4, 800, 150, False),     # smallest PREFIX_BITS whose first-pass digit still covers the hi bits
        (29, 1, 800, 150, True),
    ],
)
def test_index_matches_oracle(k, pb, nreads, L, canonical):
    _need_gpu()
    bases, offsets = synth.reads(42, nreads, L)
    g = cbl_amd.CBL(k, pb, canonical=canonical)
    o = Oracle(k, pb, canonical)
    g.insert_seqs(bases, offsets)
    o.insert_seqs(bases, offsets)
    _check_index(g, o)


@pytest.mark.parametrize("k,pb,nreads,L,canonical", [(31, 28, 3000, 150, False), (31, 25, 2500, 150, True), (59, 28, 1500, 250, False), (59, 26, 1200, 250, True),
                                                    (33, 27, 2000, 150, True), (27, 26, 2500, 150, False), (31, 26, 2500, 150, False), (15, 25, 4000, 100, False)])
def test_fine_bins_build_on_one_gpu(k, pb, nreads, L, canonical, monkeypatch):
    """PREFIX_BITS > 24 on an empty index: the first partition pass runs on FINE bins (253 intervals: blocks of 2^16 prefixes where the
    necklace prefixes are dense, two passes behind them; blocks of 2^24 above, three) instead of the top 8 prefix bits + three passes + a
    run-by-run split of the last bits. Same bytes either way: reads with N and lower case, a sequence of several chunks, every read
    twice; the second batch meets a non-empty index (the incremental path); CBLX_FINE_BINS=0 keeps the plain build."""
    _need_gpu()
    monkeypatch.setenv("CBLX_FINE_MIN", "0")
    rng = random.Random(k * 100 + pb)
    bases, offsets = synth.reads(42, nreads, L)
    b = bytearray(bases.tobytes())
    for _ in range(nreads // 40):  # a few dirty chunks
        b[rng.randrange(len(b))] = ord("N")
    for i in range(0, len(b), 997):
        b[i] = ord(chr(b[i]).lower()) if chr(b[i]) in "ACGT" else b[i]
    bases = np.frombuffer(bytes(b), dtype=np.uint8)
    long_seq = _rand_seq(rng, 7000)
    o = Oracle(k, pb, canonical)
    o.insert_seqs(bases, offsets)
    o.insert_seqs(bases, offsets)
    o.insert_seq(long_seq)
    for fine in ("1", "0"):
        monkeypatch.setenv("CBLX_FINE_BINS", fine)
        g = cbl_amd.CBL(k, pb, canonical=canonical)
        both = np.concatenate([bases, bases])
        off2 = np.concatenate([offsets, offsets[1:] + offsets[-1]])
        g.insert_seqs(both, off2)
        g.flush()
        assert g.fine_builds() == (1 if fine == "1" else 0)
        g.insert_seq(long_seq)  # on a non-empty index: the incremental path
        assert g.fine_builds() == (1 if fine == "1" else 0)
        _check_index(g, o)
        assert g.validate() == 0
        g.close()


@pytest.mark.parametrize("k,pb,nreads,L,canonical,repeated", [(31, 16, 40000, 150, False, 60), (31, 18, 30000, 150, True, 0), (59, 20, 12000, 250, False, 25), (25, 14, 30000, 150, False, 400),
                                                             (31, 22, 60000, 150, False, 300), (33, 17, 20000, 150, True, 10)])
def test_clean_spans_equal_the_per_bucket_route(k, pb, nreads, L, canonical, repeated, monkeypatch):
    """Clean spans (k_bucket_span): on an empty index, stretches of consecutive short runs (2 .. 512 words) are checked for repeats by ONE workgroup — a
    fingerprint table of (bucket, suffix) — and settled without another kernel when there is none; a span with a repeat is left to the per-bucket
    kernels untouched. Buckets of a few dozen to a few hundred words, `repeated` of the reads inserted twice (their spans must NOT be settled: the
    first occurrence stays, in stream order), 16-byte suffixes, canonical; same bytes as the oracle and as the build without (the default: the
    pre-filter is a measured switch that did not pay, DESIGN_HISTORY.md §3.13)."""
    _need_gpu()
    hb, ho = synth.reads(7 + k, nreads, L)
    if repeated:  # the first `repeated` reads once more at the end
        hb = np.concatenate([hb, hb[: repeated * L]])
        ho = np.concatenate([ho, ho[1: repeated + 1] + ho[-1]])
    o = Oracle(k, pb, canonical)
    o.insert_seqs(hb, ho)
    blobs = []
    for spans in ("1", "0"):
        monkeypatch.setenv("CBLX_SPANS", spans)
        g = cbl_amd.CBL(k, pb, canonical=canonical)
        g.insert_seqs(hb, ho)
        g.flush()
        _check_index(g, o)
        assert g.validate() == 0
        blobs.append(g.serialize())
        g.insert_seqs(hb[: 500 * L], ho[:501])  # on a non-empty index (no spans there): nothing new
        assert g.count() == o.count()
        g.close()
    assert blobs[0] == blobs[1]


@pytest.mark.parametrize(
    "k,pb,n,canonical",
    [(9, 4, 40000, False), (9, 4, 40000, True), (11, 8, 60000, False), (11, 10, 200000, False), (13, 12, 300000, False),
     (35, 6, 60000, False)],
)
def test_threshold_and_big_buckets(k, pb, n, canonical):
    """Small PREFIX_BITS force every bucket class: Vec (<= 1024 distinct, first-occurrence order with duplicates),
    Trie (> 1024), runs of > 8192 words (huge path), and for K=35/PB=6 the 128-bit suffix huge path."""
    _need_gpu()
    rng = random.Random(n + k)
    seqs = []
    for _ in range(3):
        s = _rand_seq(rng, n // 3)
        seqs += [s, s[: len(s) // 2]]  # repeats -> duplicates
    g = cbl_amd.CBL(k, pb, canonical=canonical)
    o = Oracle(k, pb, canonical)
    for s in seqs:
        g.insert_seq(s)
        o.insert_seq(s)
    _check_index(g, o)
    kinds = {kind for _, kind, _ in g.buckets()}
    assert 1 in kinds


@pytest.mark.parametrize("k,pb,nreads,L,canonical", [(31, 10, 30000, 150, False), (31, 6, 20000, 150, True), (59, 12, 12000, 250, False), (25, 8, 30000, 150, False),
                                                    (15, 4, 40000, 100, False)])
def test_deep_buckets_take_the_split_path(k, pb, nreads, L, canonical):
    """Runs of 4097 .. 262144 words (what one rank of an 8-GPU job holds per prefix at PREFIX_BITS = 24): split by the top
    suffix bits in scratch, sub-ranges sorted by the counting-sort kernel, collected in order (k_big_*). Build, incremental
    insert on top, every read twice (a big run that must STAY a Vec falls back to the general kernel), and `|=` of two such
    indexes (Trie |= Trie through the same path) — all byte-identical to the oracle."""
    _need_gpu()
    hb, ho = synth.reads(61, nreads, L)
    g, o = cbl_amd.CBL(k, pb, canonical=canonical), Oracle(k, pb, canonical)
    g.insert_seqs(hb, ho)
    o.insert_seqs(hb, ho)
    _check_index(g, o)
    _p, ln, kind = g.bucket_table_np()
    assert ln.max() > 4096 and kind.max() == 1, "shape does not reach the big path"
    hb2, ho2 = synth.reads(62, nreads // 3, L)
    g.insert_seqs(hb2, ho2)  # resident Tries + new words: big runs again
    o.insert_seqs(hb2, ho2)
    _check_index(g, o)
    g.insert_seqs(hb, ho)    # everything again: no change
    o.insert_seqs(hb, ho)
    _check_index(g, o)
    assert g.validate() == 0
    g2, o2 = cbl_amd.CBL(k, pb, canonical=canonical), Oracle(k, pb, canonical)
    hb3, ho3 = synth.reads(63, nreads, L)
    g2.insert_seqs(hb3, ho3)
    g2.insert_seqs(hb, ho[: nreads // 4 + 1])  # a shared part
    o2.insert_seqs(hb3, ho3)
    o2.insert_seqs(hb, ho[: nreads // 4 + 1])
    g |= g2
    o.merge(o2)
    _check_index(g, o)
    _check_index(g2, o2)
    # few distinct words repeated many times: the run is long, the bucket stays a Vec (first-occurrence order)
    rng = random.Random(5)
    motif = [_rand_seq(rng, k + 40) for _ in range(6)]
    seqs = [motif[i % 6] for i in range(3000)]
    d, od = cbl_amd.CBL(k, 2, canonical=canonical), Oracle(k, 2, canonical)
    for sq in seqs:
        d.insert_seq(sq)
        od.insert_seq(sq)
    _check_index(d, od)


@pytest.mark.parametrize(
    "k,pb,canonical,alphabet,glen,cov",
    [
        (31, 28, False, b"AC", 60000, 12),    # 2^4 buckets per run; runs of every length class, repeats in all of them
        (31, 26, False, b"AC", 120000, 8),    # 2^2 buckets per run: buckets over the Vec threshold inside one-tile runs
        (31, 25, True, b"ACG", 100000, 6),
        (27, 27, False, b"AC", 40000, 20),    # 8-byte records from the start (no hi part to drop)
        (31, 28, False, b"ACGT", 30000, 30),  # plain 30x coverage: every run one value, many times
        (59, 28, False, b"AC", 30000, 10),    # 16-byte records (split straight from the registers)
    ],
)
def test_prefix_split_on_crowded_runs_full_of_repeats(k, pb, canonical, alphabet, glen, cov):
    """PREFIX_BITS > 24: the runs of equal 24-bit prefix are split by their last bits in LDS (k_prefix_split, DESIGN_HISTORY.md §3.11).
    Low-complexity reads crowd the runs (two-letter genomes: 4096 possible 24-bit prefixes, runs of one to several tiles, buckets on
    both sides of the Vec threshold), coverage repeats every word. Build, an incremental batch on top, everything again."""
    _need_gpu()
    rng = random.Random(pb * 1000 + k)
    genome = _rand_seq(rng, glen, alphabet)
    L = max(100, k + 40)

    def reads(n):
        out = []
        for _ in range(n):
            q = rng.randrange(0, len(genome) - L)
            out.append(genome[q : q + L])
        return _concat(out)

    b1, o1 = reads(glen * cov // L)
    g, o = cbl_amd.CBL(k, pb, canonical=canonical), Oracle(k, pb, canonical)
    g.insert_seqs(b1, o1)
    o.insert_seqs(b1, o1)
    _check_index(g, o)
    assert g.validate() == 0
    b2, o2 = reads(glen * cov // (3 * L))
    extra = _concat([_rand_seq(rng, 3000, alphabet) for _ in range(20)])
    for bb, oo in ((b2, o2), extra, (b1, o1)):
        g.insert_seqs(bb, oo)
        o.insert_seqs(bb, oo)
        _check_index(g, o)
    assert g.validate() == 0


def test_duplicate_heavy_reads_keep_first_occurrence_order():
    """30x coverage of a small genome: every k-mer arrives many times; Vec buckets must keep stream order."""
    _need_gpu()
    rng = random.Random(5)
    genome = _rand_seq(rng, 20000)
    reads = []
    for _ in range(4000):
        p = rng.randrange(0, len(genome) - 150)
        reads.append(genome[p : p + 150])
    bases, offsets = _concat(reads)
    for k, pb in ((31, 24), (21, 12)):
        g, o = cbl_amd.CBL(k, pb), Oracle(k, pb)
        g.insert_seqs(bases, offsets)
        o.insert_seqs(bases, offsets)
        _check_index(g, o)


@pytest.mark.parametrize(
    "k,pb,canonical,glen,cov",
    [
        (15, 6, False, 3000, 30), (31, 8, False, 6000, 30), (31, 8, True, 6000, 30), (45, 10, False, 8000, 30), (21, 4, False, 1500, 30),
        # runs longer than one workgroup's sort (> 4096 arrivals): the pre-pass of the long runs (k_big_claim) ...
        (15, 4, False, 2500, 200),   # ... ending as Vecs (<= 1024 distinct per bucket, tens of thousands of arrivals)
        (15, 2, False, 2000, 60),    # ... or shrunk to 1025..2048 distinct words (1981 here), then sorted by one workgroup
        (15, 2, True, 4000, 60),     # ... or passed on untouched (3977 distinct: more than half the table)
        (17, 1, False, 1500, 400),   # ... and runs above 262144 arrivals (the class of the global-memory kernel)
        (31, 3, False, 3000, 120),   # 68-bit words (hi byte dropped after the first pass)
    ],
)
def test_repeat_heavy_runs_take_the_claim_kernel(k, pb, canonical, glen, cov):
    """Runs of a few thousand words in which every suffix occurs dozens of times (one batch at 30x coverage with few
    prefixes): the counting sort gives up on them (crowded sub-buckets) and the claim-table kernel deduplicates them; the
    buckets must keep the first-occurrence order of the stream. Second batch: the buckets that became Tries meanwhile take
    the same route and end in the radix kernel (sorted layout)."""
    _need_gpu()
    rng = random.Random(1000 * k + pb)
    genome = _rand_seq(rng, glen)
    reads = []
    for _ in range(glen * cov // 100):
        p = rng.randrange(0, len(genome) - 100)
        reads.append(genome[p : p + 100])
    bases, offsets = _concat(reads)
    g, o = cbl_amd.CBL(k, pb, canonical=canonical), Oracle(k, pb, canonical)
    g.insert_seqs(bases, offsets)
    o.insert_seqs(bases, offsets)
    _check_index(g, o)
    # grow some buckets past the threshold with distinct words, then the same repeats again plus repeats of a second genome
    extra = [_rand_seq(rng, 400) for _ in range(60 << max(0, pb - 6))]
    b2, o2 = _concat(extra)
    g.insert_seqs(b2, o2)
    o.insert_seqs(b2, o2)
    _check_index(g, o)
    genome2 = _rand_seq(rng, glen // 2)
    reads2 = reads[: len(reads) // 2]
    for _ in range(glen // 2 * cov // 100):
        p = rng.randrange(0, len(genome2) - 100)
        reads2.append(genome2[p : p + 100])
    b3, o3 = _concat(reads2)
    g.insert_seqs(b3, o3)
    o.insert_seqs(b3, o3)
    _check_index(g, o)


def test_incremental_flushes_equal_one_shot():
    """Batch boundaries must not change the result (SURVEY.md §7 hard part 6)."""
    _need_gpu()
    for k, pb, n in ((31, 24, 3000), (9, 4, 30000), (11, 8, 50000)):
        rng = random.Random(k)
        seqs = [_rand_seq(rng, 150) for _ in range(n // 150)]
        seqs += seqs[: len(seqs) // 4]
        one, inc, o = cbl_amd.CBL(k, pb), cbl_amd.CBL(k, pb), Oracle(k, pb)
        for s in seqs:
            one.insert_seq(s)
            o.insert_seq(s)
        for i, s in enumerate(seqs):
            inc.insert_seq(s)
            if i % 37 == 0:
                inc.flush()
        assert inc.serialize() == one.serialize() == o.serialize()


def test_per_record_inserts_across_queue_blocks():
    """The reference's call pattern: one insert_seq per record (/root/reference/examples/cbl.rs:160-163). cblx_insert_seq has a
    short path for the common call (two copies into the current pinned blocks of the queue) and the full path for everything else:
    70 000 records of ragged lengths cross several 4 MiB base blocks and the 64 Ki-record offsets block; a too short record in the
    middle is refused (/root/reference/src/cbl.rs:329-334) and leaves the queue as it was; a batch call and a flush in between."""
    _need_gpu()
    k, pb = 31, 24
    rng = np.random.default_rng(11)
    lens = rng.integers(k, 240, size=70_000)
    offs = np.zeros(len(lens) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(lens)
    bases = rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), size=int(offs[-1]), p=[0.24, 0.24, 0.24, 0.24, 0.01, 0.01, 0.01, 0.005, 0.005])
    raw = bases.tobytes()
    g, o = cbl_amd.CBL(k, pb), Oracle(k, pb)
    for i in range(len(lens)):
        a, b = int(offs[i]), int(offs[i + 1])
        g.insert_seq(raw[a:b])
        if i == 40_000:
            with pytest.raises(cbl_amd.CblxError) as e:
                g.insert_seq(b"ACGTACGT")
            assert e.value.code == cbl_amd.ESHORT
        if i == 50_000:
            g.insert_seqs(bases[: int(offs[300])], offs[:301])  # a batch call in between (records 0..299 again)
        if i == 66_000:
            g.flush()
    o.insert_seqs(bases, offs)
    assert g.serialize() == o.serialize()

@pytest.mark.parametrize("pack", ["1", "0"])
def test_streamed_insert_rejects_bad_offsets_and_leaves_the_index_alone(pack, monkeypatch):
    """A big host batch is inserted slice by slice behind its transfer, and the offsets of a slice are checked just before a kernel
    reads them (the first slices are already through KRN-1 by then). A sequence shorter than K in a late slice, or offsets that
    decrease in the middle, fail as a check in front of everything would (same code, same message: /root/reference/src/cbl.rs:329-334
    for the short sequence) and leave the resident index exactly as it was."""
    _need_gpu()
    monkeypatch.setenv("CBLX_H2D_PACK", pack)
    k, pb, n, L = 31, 24, 470_000, 150  # 70.5 MB of bases: the streamed paths take batches of 64 MiB and more
    bases, offsets = synth.reads(7, n, L)
    g = cbl_amd.CBL(k, pb)
    g.insert_seqs(bases[: 2000 * L], offsets[:2001])
    g.flush()
    before = g.serialize()
    short = offsets.copy()
    short[400_001:] -= np.uint64(130)  # sequence 400 000 has 20 bases now
    with pytest.raises(cbl_amd.CblxError) as e:
        g.insert_seqs(bases, short)
        g.flush()
    assert e.value.code == cbl_amd.ESHORT and "Sequence size (20) is smaller than K (31)" in str(e.value)
    assert g.serialize() == before
    down = offsets.copy()
    down[200_000] = down[200_001] + np.uint64(5)
    with pytest.raises(cbl_amd.CblxError) as e:
        g.insert_seqs(bases, down)
        g.flush()
    assert e.value.code == cbl_amd.EINVAL and "non-decreasing" in str(e.value)
    assert g.serialize() == before
    g.insert_seqs(bases, offsets)  # and the context still works
    g.flush()
    o = Oracle(k, pb)
    o.insert_seqs(bases[: 2000 * L], offsets[:2001])
    o.insert_seqs(bases, offsets)
    assert g.serialize() == o.serialize()

@pytest.mark.parametrize("k,pb,canonical,nreads,L,dirty", [(31, 24, False, 600_000, 150, False), (59, 28, True, 300_000, 250, True), (25, 12, False, 700_000, 120, True),
                                                          (31, 8, False, 500_000, 150, False), (27, 20, True, 600_000, 130, "mixed"), (31, 28, False, 400_000, 150, "mixed")])
def test_streamed_insert_from_pinned_host_memory_equals_one_shot(k, pb, canonical, nreads, L, dirty, monkeypatch):
    """A big batch handed over in PINNED host memory crosses PCIe in slices that land front to back; flush() runs KRN-1 and the
    first partition pass of slice c while the later slices are on the wire (cblx_insert_seqs + cblx_flush). Same index bytes as
    the insert of the same reads from device memory, as the same call with streaming switched off, and on top of a non-empty
    index; reads with N (dirty chunks), a ragged last slice, PREFIX_BITS = 8 (no pass behind pass A: the plain path)."""
    _need_gpu()
    d_b, d_o = synth.reads_torch(99, nreads, L, device="cuda")
    if dirty:
        d_b = d_b.clone()
        d_b[torch.arange(7, d_b.numel(), 9973, device="cuda")] = ord("N")
        if dirty == "mixed":  # lower case (valid), other letters and bytes above 127 (skipped), runs of N
            gen = torch.Generator(device="cuda")
            gen.manual_seed(5)
            pos = torch.randint(0, d_b.numel(), (d_b.numel() // 50,), device="cuda", generator=gen)
            d_b[pos] = d_b[pos] | 0x20
            pos = torch.randint(0, d_b.numel(), (d_b.numel() // 400,), device="cuda", generator=gen)
            d_b[pos] = torch.tensor(list(b"nRYx-\xe9*"), dtype=torch.uint8, device="cuda")[torch.randint(0, 7, (pos.numel(),), device="cuda", generator=gen)]
            d_b[1000:1400] = ord("N")
    hb = torch.empty(d_b.numel(), dtype=torch.uint8, pin_memory=True)
    ho = torch.empty(d_o.numel(), dtype=torch.int64, pin_memory=True)
    hb.copy_(d_b)
    ho.copy_(d_o)
    torch.cuda.synchronize()
    nb_, no_ = hb.numpy(), ho.numpy().view(np.uint64)
    ref = cbl_amd.CBL(k, pb, canonical=canonical)
    ref.insert_seqs_device(d_b, d_o, nreads)
    want = ref.serialize()
    monkeypatch.setenv("CBLX_H2D_PACK", "0")  # the ASCII bytes themselves cross the link, in slices
    for slices in ("12", "5", "1"):
        monkeypatch.setenv("CBLX_H2D_SLICES", slices)
        g = cbl_amd.CBL(k, pb, canonical=canonical)
        g.insert_seqs(nb_, no_)
        g.flush()
        assert g.count() == ref.count()
        assert g.serialize() == want, slices
        # PREFIX_BITS > 24: the slices take the FINE-bins route as they land (round 6), like the resident build
        assert g.fine_builds() == (1 if pb > 24 else 0)
        g.close()
    # the batch as bit planes (host threads pack 3 bits per base, the insert runs right behind the transfer): from the pinned
    # buffer, from a pageable copy, and with offsets that do not start at 0
    monkeypatch.setenv("CBLX_H2D_PACK", "1")
    pageable = np.array(nb_, copy=True)
    shifted = np.concatenate([np.full(37, ord("A"), dtype=np.uint8), pageable])
    for bases, offs in ((nb_, no_), (pageable, no_), (shifted, no_ + np.uint64(37))):
        g = cbl_amd.CBL(k, pb, canonical=canonical)
        g.insert_seqs(bases, offs)
        assert g.count() == ref.count()
        assert g.serialize() == want
        assert g.fine_builds() == (1 if pb > 24 else 0)
        g.close()
    # on top of a resident index (the incremental path behind the pieces), then the same batch again: nothing new
    for pack in ("0", "1"):
        monkeypatch.setenv("CBLX_H2D_PACK", pack)
        monkeypatch.setenv("CBLX_H2D_SLICES", "7")
        half = nreads // 2
        g = cbl_amd.CBL(k, pb, canonical=canonical)
        g.insert_seqs_device(d_b, d_o, half)
        g.insert_seqs(nb_[int(no_[half]):], (no_[half:] - no_[half]))
        assert g.serialize() == want
        g.insert_seqs(nb_, no_)
        assert g.serialize() == want
        g.close()
    ref.close()


def test_load_then_insert_is_cbl_insert():
    """`cbl insert` (examples/cbl.rs:230-249): read_index, insert_seq per record, write_index."""
    _need_gpu()
    rng = random.Random(8)
    for k, pb in ((31, 24), (9, 4)):
        s1, s2 = _rand_seq(rng, 30000), _rand_seq(rng, 30000)
        o = Oracle(k, pb)
        o.insert_seq(s1)
        blob = o.serialize()
        g = cbl_amd.CBL(k, pb)
        g.load(blob)
        assert g.serialize() == blob and g.count() == o.count()
        g.insert_seq(s2)
        o.insert_seq(s2)
        _check_index(g, o)


def test_merge_matches_oracle():
    """`cbl merge` (examples/cbl.rs:270-279): self |= other, incl. the Vec-may-exceed-1024 quirk."""
    _need_gpu()
    rng = random.Random(21)
    for k, pb, n, canonical in ((31, 24, 20000, False), (9, 4, 9000, False), (9, 4, 40000, False), (11, 8, 60000, False), (13, 10, 150000, True),
                                (15, 12, 400000, False), (35, 6, 3000, False), (35, 10, 100000, False), (59, 28, 30000, True), (31, 24, 2000000, False)):
        s1, s2 = _rand_seq(rng, n), _rand_seq(rng, n // 2) + _rand_seq(rng, 64)
        if n >= 100000:  # shared stretch: both-sides buckets with common words
            s2 = s2 + s1[n // 3 : n // 3 + n // 4]
        g1, g2 = cbl_amd.CBL(k, pb, canonical=canonical), cbl_amd.CBL(k, pb, canonical=canonical)
        o1, o2 = Oracle(k, pb, canonical), Oracle(k, pb, canonical)
        g1.insert_seq(s1), o1.insert_seq(s1)
        g2.insert_seq(s2), o2.insert_seq(s2)
        g1 |= g2
        o1.merge(o2)
        _check_index(g1, o1)
        _check_index(g2, o2)  # the reference's |= sorts other's Vec buckets that met a bucket of self (iter_sorted)
        assert g1.validate(strict=False) == 0
        # an oversized Vec left by |= turns into a Trie only when a later insert touches it (src/wordset/mod.rs:213-214)
        s3 = _rand_seq(rng, 3000)
        g1.insert_seq(s3)
        o1.insert_seq(s3)
        _check_index(g1, o1)


@pytest.mark.parametrize("peer", ["0", "1"])
def test_merge_from_is_clone_then_ior_without_the_copy(peer, monkeypatch):
    """cblx_merge_from(dst, a, b): dst = what `a |= b` leaves in a, a untouched, b as `|=` leaves it (its Vecs that met a bucket of a sorted) —
    the bench's merge step. Also with an empty side, on top of a dst that held something else, through the peer-copy path, and the words
    the stages were given add up (cblx_stage_units: what the bench prices the merge kernels on)."""
    _need_gpu()
    monkeypatch.setenv("CBLX_FORCE_PEER_COPY", peer)
    rng = random.Random(23)
    for k, pb, n, canonical in ((31, 24, 20000, False), (11, 8, 60000, False), (13, 10, 150000, True), (59, 28, 30000, True), (31, 10, 1200000, False)):
        s1, s2 = _rand_seq(rng, n), _rand_seq(rng, n // 2) + _rand_seq(rng, 64)
        if n >= 100000:
            s2 = s2 + s1[n // 3: n // 3 + n // 4]
        a, b = cbl_amd.CBL(k, pb, canonical=canonical), cbl_amd.CBL(k, pb, canonical=canonical)
        oa, ob = Oracle(k, pb, canonical), Oracle(k, pb, canonical)
        a.insert_seq(s1), oa.insert_seq(s1)
        b.insert_seq(s2), ob.insert_seq(s2)
        a_before = a.serialize()
        w = cbl_amd.CBL(k, pb, canonical=canonical, profile=True)
        w.insert_seq(_rand_seq(rng, 500))  # dropped by merge_from
        w.stage_times_reset()
        w.merge_from(a, b)
        oa.merge(ob)
        _check_index(w, oa)
        _check_index(b, ob)
        assert a.serialize() == a_before
        assert w.validate(strict=False) == 0
        un = w.stage_units()
        assert sum(un.values()) > 0 and un["merge_gather"] + un["bucket_medium"] + un["bucket_huge"] <= a.count() + b.count()
        w.merge_from(a, b)  # again: b's Vecs are sorted now, the result is the same
        _check_index(w, oa)
        e = cbl_amd.CBL(k, pb, canonical=canonical)
        w.merge_from(e, b)  # an empty self: every bucket of b cloned as stored
        assert w.serialize() == b.serialize()
        w.merge_from(a, e)
        assert w.serialize() == a_before
        with pytest.raises(cbl_amd.CblxError):
            w.merge_from(w, b)
        for x in (a, b, w, e):
            x.close()


def test_merge_through_the_peer_copy_path(monkeypatch):
    """`|=` of indexes on different GPUs copies other's resident index over the fabric (hipMemcpyPeer) and merges on the
    device; CBLX_FORCE_PEER_COPY=1 takes that path on one GPU. Same bytes as the oracle for self and other, quirks included,
    also into an empty self."""
    _need_gpu()
    monkeypatch.setenv("CBLX_FORCE_PEER_COPY", "1")
    rng = random.Random(22)
    for k, pb, n, canonical in ((31, 24, 20000, False), (11, 8, 60000, False), (13, 10, 150000, True), (59, 28, 30000, True), (35, 6, 3000, False)):
        s1, s2 = _rand_seq(rng, n), _rand_seq(rng, n // 2) + _rand_seq(rng, 64)
        if n >= 100000:
            s2 = s2 + s1[n // 3: n // 3 + n // 4]
        g1, g2 = cbl_amd.CBL(k, pb, canonical=canonical), cbl_amd.CBL(k, pb, canonical=canonical)
        o1, o2 = Oracle(k, pb, canonical), Oracle(k, pb, canonical)
        g1.insert_seq(s1), o1.insert_seq(s1)
        g2.insert_seq(s2), o2.insert_seq(s2)
        e, oe = cbl_amd.CBL(k, pb, canonical=canonical), Oracle(k, pb, canonical)
        e |= g2
        oe.merge(o2)
        _check_index(e, oe)
        g1 |= g2
        o1.merge(o2)
        _check_index(g1, o1)
        _check_index(g2, o2)
        assert g1.validate(strict=False) == 0 and g2.validate(strict=False) == 0


def test_contains_seq():
    _need_gpu()
    rng = random.Random(3)
    for k, pb in ((31, 24), (59, 28), (9, 4)):
        s1, s2 = _rand_seq(rng, 20000), _rand_seq(rng, 500)
        g = cbl_amd.CBL(k, pb)
        g.insert_seq(s1)
        assert all(g.contains_seq(s1[100:1000]))
        o = Oracle(k, pb)
        o.insert_seq(s1)
        mask = (1 << (2 * k)) - 1
        x, want = 0, []
        for i, ch in enumerate(s2):
            x = ((x << 2) | b"ACTG".index(ch)) & mask
            if i >= k - 1:
                want.append(o.contains_kmer(x))
        assert g.contains_seq(s2) == want


def test_insert_words_device_is_insert_batch():
    """WordSet::insert_batch on already transformed words == insert_seq on the sequence they came from."""
    _need_gpu()
    rng = random.Random(17)
    for k, pb in ((31, 24), (25, 24), (59, 28)):
        seqs = [_rand_seq(rng, 150) for _ in range(300)]
        o = Oracle(k, pb)
        words = []
        for s in seqs:
            o.insert_seq(s)
            words += o.seq_words(s)
        lo = torch.from_numpy(np.array([w & (2**64 - 1) for w in words], dtype=np.uint64).astype(np.int64)).cuda()
        g = cbl_amd.CBL(k, pb)
        hi = _hi_from_words(g, words)
        g.insert_words_device(lo, hi, len(words))
        _check_index(g, o)
        # and on a non-empty index (copy path): insert the same words again -> unchanged
        g.insert_words_device(lo, hi, len(words))
        _check_index(g, o)


# ---- multi-GPU exchange step (run here on one GPU) ---------------------------------------------------------
@pytest.mark.parametrize("k,pb", [(31, 24), (25, 24), (59, 28)])
def test_partition_words_device_is_stable(k, pb):
    _need_gpu()
    rng = random.Random(k)
    seqs = [_rand_seq(rng, 150) for _ in range(400)]
    g = cbl_amd.CBL(k, pb)
    o = Oracle(k, pb)
    words = _oracle_words(o, seqs)
    sb = g.consts()["suffix_bits"]
    prefixes = sorted(w >> sb for w in words)
    for nd in (1, 2, 5, 8, 16):
        bounds = sorted(prefixes[rng.randrange(len(prefixes))] for _ in range(nd - 1))
        lo = torch.from_numpy(np.array([w & (2**64 - 1) for w in words], dtype=np.uint64).astype(np.int64)).cuda()
        hi = _hi_from_words(g, words)
        olo, ohi = torch.zeros_like(lo), (None if hi is None else torch.zeros_like(hi))
        counts = g.partition_words_device(lo, hi, len(words), bounds, nd, olo, ohi)
        dest = [sum(1 for b in bounds if b <= (w >> sb)) for w in words]
        want = [w for d in range(nd) for w, dd in zip(words, dest) if dd == d]
        got = _words_from(olo, ohi)
        assert counts == [dest.count(d) for d in range(nd)]
        assert got == want


def test_sharded_builder_single_rank_nccl():
    """ShardedBuilder over a 1-rank RCCL group == the direct insert (same code path the N-GPU bench runs)."""
    _need_gpu()
    import os

    import torch.distributed as dist

    from cbl_amd import sharded

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for k, pb in ((31, 24), (25, 24), (59, 28)):
            d_b, d_o = synth.reads_torch(42, 3000, 150, device="cuda")
            a, b = cbl_amd.CBL(k, pb), cbl_amd.CBL(k, pb)
            a.insert_seqs_device(d_b, d_o, 3000)
            sharded.ShardedBuilder(b, dist).insert_seqs_device(d_b, d_o, 3000)  # "sorted" protocol
            blob = sharded.gather_serialized(b.serialize(), dist)
            assert blob == a.serialize()
            w = cbl_amd.CBL(k, pb)
            sharded.ShardedBuilder(w, dist, protocol="words").insert_seqs_device(d_b, d_o, 3000)
            assert w.serialize() == blob
            hb, ho = synth.reads(42, 3000, 150)
            assert (d_b[: 3000 * 150].cpu().numpy() == hb).all()
            o = Oracle(k, pb)
            o.insert_seqs(hb, ho)
            assert blob == o.serialize()
            # fewer reads than slices, then none at all: empty slices still walk the whole protocol
            for proto in ("sorted", "words"):
                t = cbl_amd.CBL(k, pb)
                sbld = sharded.ShardedBuilder(t, dist, protocol=proto)
                sbld.insert_seqs_device(d_b, d_o, 2)
                sbld.insert_seqs_device(d_b, d_o[2:], 0)
                sbld.insert_seqs_device(d_b, d_o[2:], 1)
                o3 = Oracle(k, pb)
                o3.insert_seqs(hb, ho[:4])
                assert t.serialize() == o3.serialize()
    finally:
        if created:
            dist.destroy_process_group()


# ---- BASELINE.json full size, through size-independent properties --------------------------------------------------
def _word_hash(w):
    M = (1 << 64) - 1

    def mix(z):
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    return (mix(((w & M) + 0x9E3779B97F4A7C15) & M) + mix((w >> 64) ^ 0xD1B54A32D192ED03)) & M


def test_checksum_matches_oracle_small():
    _need_gpu()
    rng = random.Random(4)
    for k, pb in ((31, 24), (59, 28), (9, 4)):
        g, o = cbl_amd.CBL(k, pb), Oracle(k, pb)
        for _ in range(20):
            s = _rand_seq(rng, 400)
            g.insert_seq(s), o.insert_seq(s)
        assert g.checksum() == sum(_word_hash(w) for w in o.iter_words()) & ((1 << 64) - 1)
        assert g.validate() == 0


@pytest.mark.parametrize("k,pb,nreads,L", [(31, 24, 10_000_000, 150), (31, 28, 12_500_000, 150), (59, 28, 6_250_000, 250)])
def test_full_size_properties(k, pb, nreads, L):
    """cfg 2 at full size (1.2 G k-mers) and the per-GPU shares of cfg 3 (12.5 M x 150 bp, PREFIX_BITS=28: the fused
    directory path) and cfg 4 (6.25 M x 250 bp, K=59: 16-byte records, 13-byte suffixes): set checksum == checksum of the transformed word stream (no k-mer repeats in
    this stream, checked via count), buckets structurally sound, sampled reads all present, foreign reads absent,
    re-inserting everything is idempotent (count and checksum unchanged)."""
    _need_gpu()
    d_b, d_o = synth.reads_torch(42, nreads, L, device="cuda")
    n_kmers = nreads * (L - k + 1)
    g = cbl_amd.CBL(k, pb)
    g.insert_seqs_device(d_b, d_o, nreads)
    count, cs = g.count(), g.checksum()
    assert g.validate() == 0
    lo = torch.empty(n_kmers + 1, dtype=torch.int64, device="cuda")
    hi = _hi_tensor(g, n_kmers + 1)
    assert g.seq_words_device(d_b, d_o, nreads, lo, hi, n_kmers) == n_kmers
    assert count <= n_kmers
    if count == n_kmers:
        assert cs == g.checksum_words_device(lo, hi, n_kmers)
    del lo, hi
    hb, ho = synth.reads(42, 3, L, first_read=nreads - 3)
    for i in range(3):
        assert all(g.contains_seq(hb[i * L : (i + 1) * L].tobytes()))
    fb, _ = synth.reads(4242, 2, L)
    assert not any(g.contains_seq(fb[:L].tobytes()))
    # the batched query at full size: every k-mer of the indexed reads is found, none of an unrelated read set
    assert g.contains_seqs_device(d_b, d_o, nreads) == (n_kmers, n_kmers)
    d_f = torch.zeros(n_kmers + 8, dtype=torch.uint8, device="cuda")
    assert g.contains_seqs_device(d_b, d_o, nreads, d_f, n_kmers) == (n_kmers, n_kmers) and int(d_f.sum(dtype=torch.int64)) == n_kmers
    del d_f
    f_b, f_o = synth.reads_torch(4242, nreads // 10, L, device="cuda")
    assert g.contains_seqs_device(f_b, f_o, nreads // 10) == (nreads // 10 * (L - k + 1), 0)
    del f_b, f_o
    g.insert_seqs_device(d_b, d_o, nreads)  # idempotence (resident + new through the incremental path)
    assert (g.count(), g.checksum()) == (count, cs)
    assert g.validate() == 0


def test_full_size_merge_properties():
    """cfg 5's per-GPU share at full size: `A |= B` of two indexes of 6.25 M x 150 bp reads (K=31, PB=24) that share the
    k-mers of 1 M reads. checksum(A | B) = checksum(A) + checksum(B) - checksum(A & B) (the checksum is a sum over the set),
    count likewise, buckets structurally sound, `other` unchanged as a set, merging again changes nothing."""
    _need_gpu()
    k, pb, n, L, shared = 31, 24, 6_250_000, 150, 1_000_000
    d_a, o_a = synth.reads_torch(42, n, L, device="cuda")
    d_b, o_b = synth.reads_torch(43, n - shared, L, device="cuda")
    A, B, S = cbl_amd.CBL(k, pb), cbl_amd.CBL(k, pb), cbl_amd.CBL(k, pb)
    A.insert_seqs_device(d_a, o_a, n)
    B.insert_seqs_device(d_b, o_b, n - shared)
    B.insert_seqs_device(d_a, o_a, shared)       # the first `shared` reads of A's stream
    S.insert_seqs_device(d_a, o_a, shared)
    del d_a, d_b
    M = (1 << 64) - 1
    ca, cb, cs_ = A.count(), B.count(), S.count()
    xa, xb, xs = A.checksum(), B.checksum(), S.checksum()
    assert ca == n * (L - k + 1) and cs_ == shared * (L - k + 1) and cb == ca  # no chance repeats in these streams
    A |= B
    assert A.count() == ca + cb - cs_
    assert A.checksum() == (xa + xb - xs) & M
    assert A.validate(strict=False) == 0 and B.validate(strict=False) == 0
    assert (B.count(), B.checksum()) == (cb, xb)
    before = (A.count(), A.checksum())
    A |= B
    assert (A.count(), A.checksum()) == before
    A |= S
    assert (A.count(), A.checksum()) == before
    hb, _ = synth.reads(43, 2, L, first_read=n - shared - 2)
    assert all(A.contains_seq(hb[:L].tobytes()))


def test_sub_batches_equal_one_shot(monkeypatch):
    """An insert larger than one batch is cut at sequence boundaries into sub-batches (cblx.h: a batch takes < 2^32 words);
    forced here with a tiny cut: same bytes as the oracle, for host and device inputs, incl. a sequence longer than the cut."""
    _need_gpu()
    monkeypatch.setenv("CBLX_BATCH_MAX_BASES", "5000")
    rng = random.Random(12)
    for k, pb, canonical in ((31, 24, False), (59, 28, True), (11, 6, False)):
        seqs = [_rand_seq(rng, n) for n in [150] * 200 + [12000, 150, 150, 7000] + [300] * 40]
        bases, offsets = _concat(seqs)
        o = Oracle(k, pb, canonical)
        o.insert_seqs(bases, offsets)
        g = cbl_amd.CBL(k, pb, canonical=canonical)
        pad = (-len(bases)) % 16 + 16
        d_b = torch.from_numpy(np.concatenate([bases, np.zeros(pad, np.uint8)])).cuda()
        d_o = torch.from_numpy(offsets.astype(np.int64)).cuda()
        g.insert_seqs_device(d_b, d_o, len(seqs))
        _check_index(g, o)
        h = cbl_amd.CBL(k, pb, canonical=canonical)
        h.insert_seqs(bases, offsets)
        _check_index(h, o)


def test_full_size_repeat_heavy_batch_properties():
    """SURVEY.md §8d's duplicate-heavy workload at full size: 8 M x 150 bp reads sampled at 30x coverage from a 40 Mbp random
    genome, ONE batch (every k-mer arrives ~24 times: the claim-table kernel carries the bucket stage). Size-independent
    properties against the index of the genome itself: subset, membership of every read k-mer, idempotence, union."""
    _need_gpu()
    k, pb, L, G = 31, 24, 150, 40_000_000
    n = G * 30 // L
    gen, _ = synth.reads_torch(4242, 1, G, device="cuda")
    gtor = torch.Generator(device="cuda")
    gtor.manual_seed(7)
    pos = torch.randint(0, G - L, (n,), device="cuda", dtype=torch.int64, generator=gtor)
    ar = torch.arange(L, device="cuda")
    d_b = torch.cat([gen[(pos[a:a + 1_000_000, None] + ar[None, :]).reshape(-1)] for a in range(0, n, 1_000_000)])
    d_o = torch.arange(0, (n + 1) * L, L, device="cuda", dtype=torch.int64)
    nk = n * (L - k + 1)
    g = cbl_amd.CBL(k, pb)
    g.insert_seqs_device(d_b, d_o, n)
    assert g.validate() == 0
    cnt, cs = g.count(), g.checksum()
    assert cnt <= G - k + 1
    g.insert_seqs_device(d_b, d_o, n)  # the whole batch again: nothing new
    assert (g.count(), g.checksum()) == (cnt, cs)
    # two halves instead of one batch: the same set (the second half meets resident buckets)
    h = cbl_amd.CBL(k, pb)
    h.insert_seqs_device(d_b, d_o, n // 2)
    h.insert_seqs_device(d_b, d_o[n // 2:], n - n // 2)
    assert (h.count(), h.checksum()) == (cnt, cs) and h.validate() == 0
    del h
    ref = cbl_amd.CBL(k, pb)
    ref.insert_seqs_device(gen, torch.tensor([0, G], device="cuda", dtype=torch.int64), 1)
    assert cnt <= ref.count()
    assert ref.contains_seqs_device(d_b, d_o, n) == (nk, nk)
    assert g.contains_seqs_device(d_b, d_o, n) == (nk, nk)
    g |= ref
    assert (g.count(), g.checksum()) == (ref.count(), ref.checksum())


def test_more_than_2_pow_32_words_in_one_index():
    """36 M x 150 bp reads = 4.32 G k-mers (> 2^32) in ONE index on one GPU: three sub-batches through the incremental path;
    then `|=` of two 2.4 G-word indexes. Size-independent properties: count, checksum == checksum of the word stream,
    structure, membership, idempotence of a re-insert of the last batch."""
    _need_gpu()
    k, pb, L = 31, 24, 150
    n = 36_000_000
    n_kmers = n * (L - k + 1)
    assert n_kmers > 1 << 32
    d_b, d_o = synth.reads_torch(42, n, L, device="cuda")
    g = cbl_amd.CBL(k, pb)
    g.insert_seqs_device(d_b, d_o, n)
    count, cs = g.count(), g.checksum()
    # 4.3 G draws from 4^31 k-mers: about two chance repeats are expected ((4.3e9)^2 / (2 * 4^31)), so the count is not exact
    assert n_kmers - 20 <= count <= n_kmers
    # the same reads in two calls (other sub-batch boundaries): the same set
    g2 = cbl_amd.CBL(k, pb)
    g2.insert_seqs_device(d_b, d_o, n // 2)
    g2.insert_seqs_device(d_b, d_o[n // 2:], n - n // 2)
    assert (g2.count(), g2.checksum()) == (count, cs)
    del g2
    # every k-mer of the reads is found (a query batch takes fewer than 2^32 k-mers: two halves)
    half_k = (n // 2) * (L - k + 1)
    assert g.contains_seqs_device(d_b, d_o, n // 2) == (half_k, half_k)
    assert g.contains_seqs_device(d_b, d_o[n // 2:], n - n // 2) == (n_kmers - half_k, n_kmers - half_k)
    assert g.validate() == 0
    hb, _ = synth.reads(42, 2, L, first_read=n - 2)
    assert all(g.contains_seq(hb[:L].tobytes())) and all(g.contains_seq(hb[L:].tobytes()))
    fb, _ = synth.reads(4242, 1, L)
    assert not any(g.contains_seq(fb.tobytes()))
    assert g.serialized_size() > 7 * n_kmers
    g.insert_seqs_device(d_b, d_o[n - 1_000_000:], 1_000_000)  # 1 M reads again: nothing new
    assert (g.count(), g.checksum()) == (count, cs)
    del g
    half = 20_000_000
    A, B = cbl_amd.CBL(k, pb), cbl_amd.CBL(k, pb)
    A.insert_seqs_device(d_b, d_o, half)
    e_b, e_o = synth.reads_torch(43, half, L, device="cuda")
    B.insert_seqs_device(e_b, e_o, half)
    del d_b, e_b
    torch.cuda.empty_cache()
    ca, cb, xa, xb = A.count(), B.count(), A.checksum(), B.checksum()
    assert ca + cb > 1 << 32
    A |= B
    assert ca + cb - 20 <= A.count() <= ca + cb  # chance repeats between the two read sets aside
    if A.count() == ca + cb:
        assert A.checksum() == (xa + xb) & ((1 << 64) - 1)
    assert A.validate(strict=False) == 0 and (B.count(), B.checksum()) == (cb, xb)
    before = (A.count(), A.checksum())
    A |= B
    assert (A.count(), A.checksum()) == before


# ---- the CLI path: FASTA / FASTQ file -> index file, byte-identical to the oracle's ------------------------------------
def test_cli_build_insert_merge_files(tmp_path):
    _need_gpu()
    from cbl_amd.__main__ import main as cli

    k, pb = 31, 24
    b1, o1 = synth.reads(11, 400, 150)
    b2, o2 = synth.reads(12, 300, 150)
    fa = tmp_path / "a.fa"
    fa.write_bytes(synth.fasta_bytes(b1, o1))
    # multi-line FASTA with CRLF and a lower-case stretch; FASTQ for the second set
    ml = tmp_path / "b.fa"
    recs = []
    for i in range(len(o2) - 1):
        s = b2[int(o2[i]) : int(o2[i + 1])].tobytes()
        recs.append(b">r%d desc\r\n" % i + s[:70].lower() + b"\r\n" + s[70:140] + b"\r\n" + s[140:] + b"\r\n")
    ml.write_bytes(b"".join(recs))
    fq = tmp_path / "b.fq"
    fq.write_bytes(b"".join(b"@r%d\n" % i + b2[int(o2[i]) : int(o2[i + 1])].tobytes() + b"\n+\n" + b"I" * 150 + b"\n" for i in range(len(o2) - 1)))

    cli(["-k", "31", "build", str(fa), "-o", str(tmp_path / "a.cbl")])
    oa = Oracle(k, pb)
    oa.insert_seqs(b1, o1)
    assert (tmp_path / "a.cbl").read_bytes() == oa.serialize()

    cli(["-k", "31", "insert", str(tmp_path / "a.cbl"), str(ml), "-o", str(tmp_path / "ab.cbl")])
    oa.insert_seqs(b2, o2)
    assert (tmp_path / "ab.cbl").read_bytes() == oa.serialize()

    cli(["-k", "31", "build", str(fq), "-o", str(tmp_path / "b.cbl")])
    ob = Oracle(k, pb)
    ob.insert_seqs(b2, o2)
    assert (tmp_path / "b.cbl").read_bytes() == ob.serialize()

    cli(["-k", "31", "merge", str(tmp_path / "a.cbl"), str(tmp_path / "b.cbl"), "-o", str(tmp_path / "m.cbl")])
    om = Oracle(k, pb)
    om.insert_seqs(b1, o1)
    om.merge(ob)
    assert (tmp_path / "m.cbl").read_bytes() == om.serialize()

    cli(["-k", "31", "build", "-c", str(fa), "-o", str(tmp_path / "c.cbl")])
    oc = Oracle(k, pb, True)
    oc.insert_seqs(b1, o1)
    assert (tmp_path / "c.cbl").read_bytes() == oc.serialize()

    # list: one k-mer per line in iteration order (examples/cbl.rs:177-203); query: tallies (:205-228)
    cli(["-k", "31", "list", str(tmp_path / "a.cbl"), "-o", str(tmp_path / "a.txt")])
    oa1 = Oracle(k, pb)
    oa1.insert_seqs(b1, o1)
    want = b"".join(bytes(b"ACTG"[(oa1.kmer_of_word(w) >> (2 * (k - 1 - j))) & 3] for j in range(k)) + b"\n" for w in oa1.iter_words()[:3000])
    got = (tmp_path / "a.txt").read_bytes()
    assert got[: len(want)] == want and len(got) == oa1.count() * (k + 1)
    g = cbl_amd.CBL.load_from_file(str(tmp_path / "a.cbl"), k, pb)
    nrec, total, positive = g.query_fastx_file(str(fq))
    exp = [oa1.contains_word(w) for i in range(len(o2) - 1) for w in oa1.seq_words(b2[int(o2[i]) : int(o2[i + 1])].tobytes())]
    assert (nrec, total, positive) == (len(o2) - 1, len(exp), sum(exp))
    assert g.query_fastx_file(str(fa))[1:] == ((len(o1) - 1) * (150 - k + 1),) * 2  # every k-mer of the indexed file is found
    cli(["-k", "31", "query", str(tmp_path / "a.cbl"), str(ml)])


def test_offset_slices_address_the_same_buffer():
    """Device entry points accept offsets[a..b] of a larger batch (offsets[0] != 0, unaligned) against the full base
    buffer — what the sliced multi-GPU exchange feeds them."""
    _need_gpu()
    rng = random.Random(31)
    seqs = [_rand_seq(rng, rng.choice((31, 77, 150, 151, 3000))) for _ in range(60)]
    bases, offsets = _concat(seqs)
    pad = (-len(bases)) % 16 + 16
    d_b = torch.from_numpy(np.concatenate([bases, np.zeros(pad, np.uint8)])).cuda()
    d_o = torch.from_numpy(offsets.astype(np.int64)).cuda()
    for k, pb, canonical in ((31, 24, False), (31, 24, True), (59, 28, False)):
        g, o = cbl_amd.CBL(k, pb, canonical=canonical), Oracle(k, pb, canonical)
        seqs_k = [s for s in seqs]
        for a, b in ((0, 7), (7, 8), (8, 41), (41, 60)):
            if any(len(s) < k for s in seqs_k[a:b]):
                continue
            g.insert_seqs_device(d_b, d_o[a : b + 1].contiguous(), b - a)
            for s in seqs_k[a:b]:
                o.insert_seq(s)
        _check_index(g, o)


def test_seq_words_partitioned_matches_two_step():
    """The fused KRN-1 + destination partition equals seq_words followed by partition_words."""
    _need_gpu()
    rng = random.Random(23)
    seqs = [_rand_seq(rng, rng.choice((150, 151, 400, 3000))) for _ in range(120)]
    bases, offsets = _concat(seqs)
    pad = (-len(bases)) % 16 + 16
    d_b = torch.from_numpy(np.concatenate([bases, np.zeros(pad, np.uint8)])).cuda()
    d_o = torch.from_numpy(offsets.astype(np.int64)).cuda()
    for k, pb, canonical in ((31, 24, False), (31, 24, True), (25, 24, False), (59, 28, False)):
        g = cbl_amd.CBL(k, pb, canonical=canonical)
        words = _gpu_words(g, seqs)
        sb = g.consts()["suffix_bits"]
        prefixes = sorted(w >> sb for w in words)
        for nd in (1, 3, 8):
            bounds = sorted(prefixes[rng.randrange(len(prefixes))] for _ in range(nd - 1))
            cap = len(bases)
            olo = torch.zeros(cap + 1, dtype=torch.int64, device="cuda")
            ohi = _hi_tensor(g, cap + 1)
            nw, counts = g.seq_words_partitioned_device(d_b, d_o, len(seqs), bounds, nd, olo, ohi, cap)
            assert nw == len(words)
            dest = [sum(1 for b in bounds if b <= (w >> sb)) for w in words]
            want = [w for d in range(nd) for w, dd in zip(words, dest) if dd == d]
            assert counts == [dest.count(d) for d in range(nd)]
            assert _words_from(olo[:nw], None if ohi is None else ohi[:nw]) == want


def test_device_serializer_equals_host_serializer(monkeypatch, tmp_path):
    """The index bytes are emitted by kernels (kernels_serde.hpp); the multi-threaded host emitter stays as the path
    for buckets longer than 8192 words. Both must give the same bytes, through serialize() and save_to_file()."""
    _need_gpu()
    kinds = set()
    for k, pb, nreads, L, canonical in ((31, 20, 200_000, 150, False), (59, 22, 60_000, 250, True), (25, 16, 100_000, 100, False), (15, 6, 3000, 100, False)):
        bases, offsets = synth.reads(5, nreads, L)
        g = cbl_amd.CBL(k, pb, canonical=canonical)
        g.insert_seqs(bases, offsets)
        monkeypatch.delenv("CBLX_HOST_SERDE", raising=False)
        dev = g.serialize()
        f = tmp_path / "x.cbl"
        g.save_to_file(str(f))
        assert f.read_bytes() == dev
        monkeypatch.setenv("CBLX_HOST_SERDE", "1")
        host = g.serialize()
        monkeypatch.delenv("CBLX_HOST_SERDE")
        assert len(dev) == len(host)
        assert dev == host
        if nreads <= 100_000:
            kinds |= {kind for _, kind, _ in g.buckets()}
        g2 = cbl_amd.CBL(k, pb, canonical=canonical)
        g2.load(dev)
        assert g2.checksum() == g.checksum() and g2.count() == g.count()
    assert kinds == {0, 1}


def test_fastx_reader_variants(tmp_path):
    """needletail stand-in (examples/cbl.rs:112-115): what bytes reach insert_seq for FASTA / FASTQ, plain / gzip, LF / CRLF,
    wrapped lines, blank lines, a missing final newline, a sequence longer than the read block; and its errors."""
    _need_gpu()
    import gzip

    k, pb = 31, 24
    rng = random.Random(77)
    seqs = [_rand_seq(rng, n) for n in (150, 31, 400, 1000, 33)]
    o = Oracle(k, pb)
    for s_ in seqs:
        o.insert_seq(s_)
    want = o.serialize()

    def wrap(s_, w, eol):
        return eol.join(s_[i : i + w] for i in range(0, len(s_), w)) + eol

    files = {
        "one_line.fa": b"".join(b">r%d\n" % i + s_ + b"\n" for i, s_ in enumerate(seqs)),
        "wrapped_crlf.fa": b"".join(b">r%d some text\r\n" % i + wrap(s_, 60, b"\r\n") for i, s_ in enumerate(seqs)),
        "no_final_newline.fa": b"\n\n" + b"".join(b">r%d\n" % i + wrap(s_, 70, b"\n") for i, s_ in enumerate(seqs))[:-1],
        "blank_lines.fa": b"".join(b">r%d\n" % i + s_[:20] + b"\n\n" + s_[20:] + b"\n\n" for i, s_ in enumerate(seqs)),
        "reads.fq": b"".join(b"@r%d\n" % i + s_ + b"\n+\n" + b"I" * len(s_) + b"\n" for i, s_ in enumerate(seqs)),
        "reads_crlf.fq": b"".join(b"@r%d\r\n" % i + s_ + b"\r\n+r%d\r\n" % i + b"@" * len(s_) + b"\r\n" for i, s_ in enumerate(seqs)),
    }
    files["one_line.fa.gz"] = gzip.compress(files["one_line.fa"])
    files["reads.fq.gz"] = gzip.compress(files["reads.fq"])
    for name, data in files.items():
        f = tmp_path / name
        f.write_bytes(data)
        g = cbl_amd.CBL(k, pb)
        assert g.insert_fastx_file(str(f)) == len(seqs), name
        assert g.serialize() == want, name
    # one sequence much longer than the 16 MiB read block, on a single line and wrapped
    big = _rand_seq(rng, 5_000_000) * 4
    ob = Oracle(k, pb)
    ob.insert_seq(big)
    for name, data in (("big1.fa", b">chr\n" + big + b"\n"), ("bigw.fa.gz", gzip.compress(b">chr\n" + wrap(big, 80, b"\n"), 1))):
        f = tmp_path / name
        f.write_bytes(data)
        g = cbl_amd.CBL(k, pb)
        assert g.insert_fastx_file(str(f)) == 1
        assert g.count() == ob.count() and g.serialize() == ob.serialize(), name
    # errors: a record shorter than K (src/cbl.rs:329-334) keeps the earlier records; truncated FASTQ; not FASTA/FASTQ
    bad = tmp_path / "short.fa"
    bad.write_bytes(b">a\n" + seqs[0] + b"\n>b\nACGT\n>c\n" + seqs[2] + b"\n")
    g = cbl_amd.CBL(k, pb)
    with pytest.raises(cbl_amd.CblxError, match="smaller than K"):
        g.insert_fastx_file(str(bad))
    o1 = Oracle(k, pb)
    o1.insert_seq(seqs[0])
    assert g.serialize() == o1.serialize()
    for name, data, msg in (("trunc.fq", b"@r\n" + seqs[0] + b"\n+\n", "truncated"), ("noplus.fq", b"@r\n" + seqs[0] + b"\nX\nIII\n", "separator"),
                            ("text.txt", b"hello\n", "not a FASTA")):
        f = tmp_path / name
        f.write_bytes(data)
        g = cbl_amd.CBL(k, pb)
        with pytest.raises(cbl_amd.CblxError, match=msg):
            g.insert_fastx_file(str(f))
        assert g.count() == 0
    (tmp_path / "empty.fa").write_bytes(b"")
    g = cbl_amd.CBL(k, pb)
    assert g.insert_fastx_file(str(tmp_path / "empty.fa")) == 0 and g.count() == 0


def test_bounded_ingest_queue_flushes_in_order(tmp_path):
    """The pending-sequence queue is bounded (2 GiB of bases by default): past the bound a batch is inserted and the next
    one goes through the incremental path. A tiny bound (child process: the bound is read once per process) must give the
    same index through insert_seq, insert_seqs (bulk lanes) and the FASTA reader."""
    _need_gpu()
    import subprocess
    import sys

    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import cbl_amd
from cbl_amd import synth
from oracle import Oracle
k, pb = 31, 24
bases, offsets = synth.reads(9, 3000, 150)
o = Oracle(k, pb); o.insert_seqs(bases, offsets); want = o.serialize()
g = cbl_amd.CBL(k, pb)
for i in range(len(offsets) - 1):
    g.insert_seq(bases[int(offsets[i]):int(offsets[i + 1])].tobytes())
assert g.serialize() == want, "insert_seq"
g = cbl_amd.CBL(k, pb)
for a in range(0, 3000, 700):
    b = min(a + 700, 3000)
    g.insert_seqs(bases[int(offsets[a]):int(offsets[b])], (offsets[a:b + 1] - offsets[a]).astype(np.uint64))
assert g.serialize() == want, "insert_seqs"
big_b, big_o = synth.reads(10, 20000, 150)   # 3 MB: the bulk lanes, bound hit inside one call
o2 = Oracle(k, pb); o2.insert_seqs(big_b, big_o)
g = cbl_amd.CBL(k, pb); g.insert_seqs(big_b, big_o.astype(np.uint64)); g.insert_seqs(big_b[:150 * 50], big_o[:51].astype(np.uint64))
assert g.serialize() == o2.serialize(), "bulk"
fa = %r
open(fa, "wb").write(synth.fasta_bytes(bases, offsets))
g = cbl_amd.CBL(k, pb); assert g.insert_fastx_file(fa) == 3000
assert g.serialize() == want, "fasta"
print("ok")
""" % (ROOT, str(tmp_path / "q.fa"))
    env = dict(os.environ, CBLX_INGEST_FLUSH_BYTES="100000")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_sorted_batch_protocol_matches_oracle():
    """cblx_sorted_batch_begin / _export / cblx_insert_sorted_batches_device (the multi-GPU wire protocol): the exported
    batch is the oracle's words stably sorted by prefix (prefixes, counts, packed suffixes, destination splits), and
    inserting batches — several sources, with and without a resident index — equals inserting the reads in batch order."""
    _need_gpu()
    for k, pb, canonical, nreads, L in ((31, 24, False, 900, 150), (59, 28, True, 300, 250), (9, 4, False, 500, 120), (25, 16, False, 2000, 100)):
        P = cbl_amd.CBL(k, pb, canonical=canonical)
        sb_, B = P.consts()["suffix_bits"], P.consts()["bytes"]
        nd = 4
        bounds = np.array(sorted({(1 << pb) // 64, (1 << pb) // 16, (1 << pb) // 3}), dtype=np.uint32)
        assert len(bounds) == nd - 1
        recv = cbl_amd.CBL(k, pb, canonical=canonical)
        o = Oracle(k, pb, canonical)
        # a resident index first (reads of seed 1), then batches of seeds 2, 3, 4 exported by another ctx
        hb, ho = synth.reads(1, nreads, L)
        recv.insert_seqs(hb, ho)
        o.insert_seqs(hb, ho)
        keep = []
        batches = []
        for seed in (2, 3, 4):
            hb, ho = synth.reads(seed, nreads, L)
            d_b = torch.from_numpy(np.concatenate([hb, np.zeros(32, np.uint8)])).cuda()
            d_o = torch.from_numpy(ho.astype(np.int64)).cuda()
            bs, ws = P.sorted_batch_begin(d_b, d_o, nreads, bounds, nd)
            nbk, nw = bs[nd], ws[nd]
            prefix = torch.empty(nbk, dtype=torch.int32, device="cuda")
            count = torch.empty(nbk, dtype=torch.int32, device="cuda")
            suffix = torch.empty(nw * B, dtype=torch.uint8, device="cuda")
            P.sorted_batch_export(prefix, count, suffix)
            # against the oracle's words of the same reads, stably sorted by prefix
            ow = Oracle(k, pb, canonical)
            words = []
            for i in range(nreads):
                words += ow.seq_words(hb[int(ho[i]) : int(ho[i + 1])].tobytes())
            assert nw == len(words)
            words.sort(key=lambda w: w >> sb_)
            from collections import Counter

            pre = [w >> sb_ for w in words]
            tally = Counter(pre)
            uniq = sorted(tally)
            assert prefix.cpu().numpy().astype(np.uint32).tolist() == uniq
            cnt = count.cpu().tolist()
            assert sum(cnt) == nw and cnt == [tally[p] for p in uniq]
            raw = suffix.cpu().numpy().tobytes()
            mask = (1 << sb_) - 1
            assert raw == b"".join((w & mask).to_bytes(B, "little") for w in words)
            for d in range(1, nd):
                kb = int(np.searchsorted(np.array(uniq, dtype=np.int64), int(bounds[d - 1]), side="left"))
                assert bs[d] == kb and ws[d] == int(np.sum(cnt[:kb]))
            # ship it as two batches (destinations 0-1 and 2-3 of the split) to exercise several sources per insert
            cut_b, cut_w = bs[2], ws[2]
            for (b0, b1, w0, w1) in ((0, cut_b, 0, cut_w), (cut_b, nbk, cut_w, nw)):
                batches.append((b1 - b0, w1 - w0, prefix[b0:b1], count[b0:b1], suffix[w0 * B : w1 * B]))
            keep.append((prefix, count, suffix))
            o.insert_seqs(hb, ho)
        recv.insert_sorted_batches_device(batches[:2])   # one source on top of the resident index
        recv.insert_sorted_batches_device(batches[2:])   # then two sources at once
        _check_index(recv, o)
        assert recv.validate() == 0
        # empty index + all batches at once
        fresh = cbl_amd.CBL(k, pb, canonical=canonical)
        fresh.insert_sorted_batches_device(batches)
        o2 = Oracle(k, pb, canonical)
        for seed in (2, 3, 4):
            hb, ho = synth.reads(seed, nreads, L)
            o2.insert_seqs(hb, ho)
        _check_index(fresh, o2)
    # malformed batches are refused and leave the index as it was
    g = cbl_amd.CBL(31, 24)
    g.insert_seq(b"ACGT" * 20)
    before = g.serialize()
    pfx = torch.tensor([5, 5], dtype=torch.int32, device="cuda")
    cnt = torch.tensor([1, 1], dtype=torch.int32, device="cuda")
    sfx = torch.zeros(12, dtype=torch.uint8, device="cuda")
    for bad_p, bad_c in ((pfx, cnt), (torch.tensor([3, 9], dtype=torch.int32, device="cuda"), torch.tensor([2, 0], dtype=torch.int32, device="cuda"))):
        with pytest.raises(cbl_amd.CblxError):
            g.insert_sorted_batches_device([(2, 2, bad_p, bad_c, sfx)])
        assert g.serialize() == before


@pytest.mark.parametrize("planes", ["1", "0"])
def test_parallel_fastx_reader(tmp_path, planes):
    """Large plain FASTA / FASTQ files are read by several threads (regions cut at record starts, two passes). With the
    thresholds lowered (child process) small files take that path: many regions, several flush windows; irregular input
    falls back to the sequential reader and fails exactly as it does. planes = 1: the parser threads pack the lines into bit
    planes and the insert runs behind them (comm.hpp fastx_parallel_planes); 0: ASCII through the transfer lanes."""
    _need_gpu()
    import subprocess
    import sys

    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import cbl_amd
from cbl_amd import synth
from oracle import Oracle
k, pb = 31, 24
tmp = %r
bases, offsets = synth.reads(77, 12000, 150)
rng = np.random.default_rng(5)
hit = rng.random(bases.size)
bases = np.where(hit < 0.0004, ord("N"), np.where(hit < 0.3, bases | 0x20, bases)).astype(np.uint8)  # some N, 30%% lower case
o = Oracle(k, pb); o.insert_seqs(bases, offsets); want = o.serialize()
seqs = [bases[int(offsets[i]):int(offsets[i + 1])].tobytes() for i in range(len(offsets) - 1)]
files = {
  "a.fa": b"".join(b">r%%d\n" %% i + s + b"\n" for i, s in enumerate(seqs)),
  "b.fa": b"\n" + b"".join(b">r%%d x\r\n" %% i + s[:70] + b"\r\n" + s[70:] + b"\r\n\r\n" for i, s in enumerate(seqs))[:-2],
  "c.fq": b"".join(b"@r%%d\n" %% i + s + b"\n+\n" + (b"@" if i %% 3 == 0 else b"I") * len(s) + b"\n" for i, s in enumerate(seqs)),
}
for name, data in files.items():
    path = tmp + "/" + name
    open(path, "wb").write(data)
    g = cbl_amd.CBL(k, pb)
    assert g.insert_fastx_file(path) == len(seqs), name
    assert g.serialize() == want, name
# a short record in the middle: same error and same partial result as the sequential reader
bad = tmp + "/bad.fa"
open(bad, "wb").write(files["a.fa"][:200000].rsplit(b">", 1)[0] + b">short\nACGT\n" + files["a.fa"][200000:].split(b"\n", 1)[1])
g = cbl_amd.CBL(k, pb)
try:
    g.insert_fastx_file(bad)
    raise SystemExit("no error")
except cbl_amd.CblxError as e:
    assert "smaller than K" in str(e)
n_before = files["a.fa"][:200000].rsplit(b">", 1)[0].count(b">")
o2 = Oracle(k, pb); o2.insert_seqs(bases[: n_before * 150], offsets[: n_before + 1]); assert g.serialize() == o2.serialize()
print("ok")
""" % (ROOT, str(tmp_path))
    env = dict(os.environ, CBLX_FASTX_PARALLEL_MIN="1000", CBLX_FASTX_REGION_BYTES="90000", CBLX_INGEST_FLUSH_BYTES="500000", CBLX_H2D_PACK=planes)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_index_bytes_leave_in_chunks(tmp_path):
    """cblx_serialize / cblx_save_to_file hand the emitted bytes over in chunks of consecutive buckets (the download of one runs
    while the next is emitted; indexes of 256 MB and more). With the threshold lowered (child process) small indexes take that
    path: the bytes equal the oracle's, into a buffer and into a file, for Vec-only, Trie-heavy, wide-suffix and split-Trie shapes."""
    _need_gpu()
    import subprocess
    import sys

    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import cbl_amd
from cbl_amd import synth
from oracle import Oracle
tmp = %r
for k, pb, nr, L, canonical in ((31, 24, 3000, 150, False), (31, 8, 6000, 150, False), (59, 12, 1500, 250, True), (21, 6, 9000, 150, False), (31, 24, 0, 150, False)):
    bases, offsets = synth.reads(5 + k, nr, L)
    o = Oracle(k, pb, canonical); g = cbl_amd.CBL(k, pb, canonical=canonical)
    if nr:
        o.insert_seqs(bases, offsets); g.insert_seqs(bases, offsets)
    want = o.serialize()
    assert g.serialize() == want, (k, pb)
    path = tmp + "/i_%%d_%%d.cbl" %% (k, pb)
    g.save_to_file(path)
    assert open(path, "rb").read() == want, (k, pb, "file")
    small = np.empty(max(len(want) - 1, 1), dtype=np.uint8)  # a buffer one byte short: refused, nothing written past it
    try:
        import ctypes as C
        w = C.c_uint64(0)
        rc = g._L.cblx_serialize(g._h, small.ctypes.data_as(C.POINTER(C.c_uint8)), len(want) - 1, C.byref(w))
        assert rc == cbl_amd.ERANGE and w.value == len(want), (rc, w.value)
    finally:
        pass
print("ok")
""" % (ROOT, str(tmp_path))
    env = dict(os.environ, CBLX_SERDE_CHUNK_MIN="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr

# ---- packed k-mers: CBL::insert / contains / iter and the bucket statistics (/root/reference/src/cbl.rs:219-228,358-386;
# the reference's own shape: src/cbl.rs:591-662 insert/contains of random k-mers, :700-724 iter) ------------------------
@pytest.mark.parametrize("k,pb,canonical", [(31, 24, False), (31, 24, True), (25, 12, False), (11, 8, True), (59, 28, False), (45, 20, True), (33, 16, False), (31, 2, True), (31, 3, False)])
def test_single_kmer_insert_contains_iter(k, pb, canonical):
    _need_gpu()
    rng = random.Random(1000 * k + pb + canonical)
    g, o = cbl_amd.CBL(k, pb, canonical=canonical), Oracle(k, pb, canonical)
    pool = [rng.getrandbits(2 * k) for _ in range(700)]
    pool += [0, (1 << (2 * k)) - 1, 1, 1 << (2 * k - 1), int("01" * k, 2), int("10" * k, 2)]
    # a first batch with repeats inside the batch, then a sequence, then a batch that meets resident k-mers
    seq = _rand_seq(rng, 4 * k + 300, b"ACGTN")
    b1 = [rng.choice(pool) for _ in range(1500)]
    b2 = [rng.choice(pool) for _ in range(900)] + [rng.getrandbits(2 * k) for _ in range(300)]
    got1 = g.insert_kmers(b1)
    exp1 = [o.insert_kmer(x) for x in b1]
    assert got1.tolist() == exp1
    assert g.count() == o.count()
    g.insert_seq(seq), o.insert_seq(seq)
    got2 = g.insert_kmers(b2)
    exp2 = [o.insert_kmer(x) for x in b2]
    assert got2.tolist() == exp2
    assert g.count() == o.count()
    assert g.serialize() == o.serialize()
    probe = pool + [rng.getrandbits(2 * k) for _ in range(500)] + b2[-50:]
    assert g.contains_kmers(probe).tolist() == [o.contains_kmer(x) for x in probe]
    assert g.insert(b2[-1]) is False and g.contains(b2[-1]) is True
    # iter: the oracle's words in iteration order, each turned back into its k-mer
    exp_kmers = [o.kmer_of_word(w) for w in o.iter_words()]
    assert list(g.iter()) == exp_kmers
    assert len(exp_kmers) == g.count()
    # every k-mer the iterator yields is a member, and (non-canonical) so was every inserted one
    assert all(g.contains_kmers(exp_kmers[:2000]))
    # contains_all
    assert g.contains_all(seq) is True
    other = _rand_seq(random.Random(5), 5 * k, b"ACGT")
    assert g.contains_all(other) == all(g.contains_seq(other))
    # statistics
    sizes = g.buckets_sizes()
    exp_sizes = {}
    sb = g.consts()["suffix_bits"]
    for w in o.iter_words():
        exp_sizes[w >> sb] = exp_sizes.get(w >> sb, 0) + 1
    assert sizes == sorted(exp_sizes.items())
    assert g.prefix_load() == len(exp_sizes) / float(1 << pb)
    sc = g.buckets_size_count()
    assert sum(s * c for s, c in sc.items()) == g.count() and list(sc) == sorted(sc)
    assert abs(sum(g.buckets_load_repartition().values()) - 1.0) < 1e-9


def test_insert_kmers_rejects_out_of_range_and_keeps_index():
    _need_gpu()
    g = cbl_amd.CBL(15, 10)
    g.insert_kmers([1, 2, 3])
    with pytest.raises(cbl_amd.CblxError):
        g.insert_kmers([5, 1 << 30])  # 2K = 30 bits
    with pytest.raises(cbl_amd.CblxError):
        g.contains_kmers([1 << 64])
    assert g.count() == 3
    assert g.insert_kmers([]).tolist() == []
    e = cbl_amd.CBL(15, 10)
    assert list(e.iter()) == [] and e.buckets_sizes() == [] and e.prefix_load() == 0.0


def test_single_insert_crosses_vec_trie_threshold():
    """One prefix fed by single inserts past 1024 distinct suffixes: first-occurrence order until the conversion, then a
    Trie (/root/reference/src/wordset/mod.rs:97-120,240-244)."""
    _need_gpu()
    k, pb = 15, 4
    rng = random.Random(77)
    g, o = cbl_amd.CBL(k, pb), Oracle(k, pb)
    for n in (600, 500, 2500):
        batch = [rng.getrandbits(2 * k) for _ in range(n)]
        batch += batch[: n // 10]
        assert g.insert_kmers(batch).tolist() == [o.insert_kmer(x) for x in batch]
        assert g.serialize() == o.serialize()
    assert g.validate() == 0


# ---- batched query: the `cbl query` loop (/root/reference/examples/cbl.rs:205-228) ------------------------------------------
@pytest.mark.parametrize("k,pb,canonical", [(31, 24, False), (31, 10, True), (15, 4, False), (59, 28, False), (45, 6, True), (31, 2, False)])
def test_batched_query_matches_oracle(k, pb, canonical, tmp_path, monkeypatch):
    """contains_seqs / query_fastx_file against the oracle's membership of every word, with Vec buckets, big Trie buckets
    (small PREFIX_BITS), misses that share a bucket with hits, and sequences with non-ACGT bytes."""
    _need_gpu()
    rng = random.Random(31 * k + pb)
    g, o = cbl_amd.CBL(k, pb, canonical=canonical), Oracle(k, pb, canonical)
    genome = _rand_seq(rng, 60000)
    g.insert_seq(genome), o.insert_seq(genome)
    assert g.serialize() == o.serialize()
    if pb <= 6:
        assert g.bucket_table_np()[2].max() == 1  # Trie buckets are searched too
    queries = []
    for _ in range(40):
        a = rng.randrange(0, len(genome) - 2000)
        s = bytearray(genome[a : a + rng.randint(k, 1500)])
        for _ in range(rng.randint(0, 6)):  # point mutations: k-mers that miss but fall into populated buckets
            s[rng.randrange(len(s))] = rng.choice(b"ACGTN")
        queries.append(bytes(s))
    queries.append(_rand_seq(rng, 5000))          # multi-chunk miss-heavy sequence
    queries.append(genome[1000 : 1000 + k])       # a single k-mer
    bases, offsets = _concat(queries)
    flags, tot, pos = g.contains_seqs(bases, offsets)
    want = []
    for q in queries:
        want += [o.contains_word(w) for w in o.seq_words(q)]
    assert tot == len(want) and flags.tolist() == want and pos == sum(want)
    # per-sequence entry point gives the same flags
    assert g.contains_seq(queries[0]) == want[: len(o.seq_words(queries[0]))]
    # tallies only, device-resident batch
    pad = (-len(bases)) % 16 + 16
    d_b = torch.from_numpy(np.concatenate([bases, np.zeros(pad, np.uint8)])).cuda()
    d_o = torch.from_numpy(offsets.astype(np.int64)).cuda()
    d_f = torch.zeros(len(bases), dtype=torch.uint8, device="cuda")
    assert g.contains_seqs_device(d_b, d_o, len(queries)) == (tot, pos)
    assert g.contains_seqs_device(d_b, d_o, len(queries), d_f, len(bases)) == (tot, pos)
    assert d_f[:tot].cpu().numpy().astype(bool).tolist() == want
    _, tot2, pos2 = g.contains_seqs(bases, offsets, flags=False)
    assert (tot2, pos2) == (tot, pos)
    # the same tallies through the join (big batches take it by themselves; forced here), Vec and Trie buckets alike
    monkeypatch.setenv("CBLX_QUERY_JOIN_MIN", "1")
    assert g.contains_seqs(bases, offsets, flags=False)[1:] == (tot, pos)
    assert g.contains_seqs_device(d_b, d_o, len(queries)) == (tot, pos)
    jf, jt, jp = g.contains_seqs(bases, offsets)  # per-k-mer flags by join where the word leaves room for the ordinal
    assert (jf.tolist(), jt, jp) == (want, tot, pos)
    assert g.contains_seq(queries[0]) == want[: len(o.seq_words(queries[0]))]
    e = cbl_amd.CBL(k, pb, canonical=canonical)
    assert e.contains_seqs(bases, offsets, flags=False)[1:] == (tot, 0)  # empty index
    # the file loop; the index is untouched and pending inserts are applied first
    fa = tmp_path / "q.fa"
    with open(fa, "wb") as f:
        for i, q in enumerate(queries):
            f.write(b">q%d\n" % i)
            for j in range(0, len(q), 70):
                f.write(q[j : j + 70] + b"\n")
    before = g.count()
    extra = _rand_seq(rng, 3 * k)
    g.insert_seq(extra), o.insert_seq(extra)  # stays pending until the query flushes it
    want2 = []
    for q in queries:
        want2 += [o.contains_word(w) for w in o.seq_words(q)]
    assert g.query_fastx_file(fa) == (len(queries), len(want2), sum(want2))
    assert g.count() == o.count() >= before and g.serialize() == o.serialize()
    with pytest.raises(cbl_amd.CblxError):
        g.contains_seqs(*_concat([queries[0], b"ACG"]))
    assert g.contains_seqs(*_concat([]))[1:] == (0, 0)


@pytest.mark.parametrize("preload", [False, True])
def test_heavily_duplicated_runs_of_33_to_128_words(preload):
    """A run of 33..128 words that are (nearly) all equal overflows a sub-bucket of the counting-sort kernel and comes back
    through its retry list — also when no longer bucket class is populated (regression: the retry count was only read
    when a longer class had work, such buckets were never finalized)."""
    _need_gpu()
    k, pb = 7, 8
    g, o = cbl_amd.CBL(k, pb), Oracle(k, pb)
    if preload:
        s = b"ACGTTGCAAC" * 3
        g.insert_seq(s), o.insert_seq(s)
    batch = [8027, 7897, 1847] * 80 + [8027] * 40
    assert g.insert_kmers(batch).tolist() == [o.insert_kmer(x) for x in batch]
    assert g.count() == o.count() and g.serialize() == o.serialize() and g.validate() == 0
    # the same shape through insert_seq: poly-A stretches inside distinct contexts
    for k2, pb2 in ((31, 24), (15, 10)):
        g2, o2 = cbl_amd.CBL(k2, pb2), Oracle(k2, pb2)
        s = (b"A" * (k2 + 70) + b"C") * 3
        g2.insert_seq(s), o2.insert_seq(s)
        assert g2.serialize() == o2.serialize() and g2.validate() == 0


@pytest.mark.parametrize("k,pb,n,preload", [(11, 8, 45000, 0), (11, 8, 30000, 18000), (7, 8, 45000, 0), (7, 8, 30000, 20000), (13, 10, 170000, 0)])
def test_runs_of_129_to_256_words_take_the_four_slot_class(k, pb, n, preload):
    """Round 6: runs of 129..256 words have a class of their own (k_classify CLS_M32: one wave, four slots per lane; their claim-table and
    counting-sort kernels are instantiations no other length reaches). Random k-mers spread evenly over the 2^pb prefixes put every run
    of the batch into that class — all distinct (k = 11, 13), full of repeats (k = 7: 512 suffixes per prefix, the claim table), and on top
    of the Vec buckets of an earlier batch. Per-k-mer return values, count, bytes: the reference's TrieVec::insert
    (/root/reference/src/trievec/mod.rs:72-99) word by word."""
    _need_gpu()
    rng = np.random.default_rng(1000 * k + pb + preload)
    g, o = cbl_amd.CBL(k, pb), Oracle(k, pb)
    for m in (preload, n):
        if not m:
            continue
        batch = rng.integers(0, 4 ** k, size=m, dtype=np.uint64).tolist()
        assert g.insert_kmers(batch).tolist() == [o.insert_kmer(x) for x in batch]
        assert g.count() == o.count() and g.validate() == 0
    assert g.serialize() == o.serialize()


@pytest.mark.parametrize("k,pb,n1,n2", [(11, 8, 40000, 30000), (13, 10, 100000, 60000), (11, 8, 12000, 9000)])
def test_insert_into_short_tries_of_a_loaded_file(k, pb, n1, n2):
    """A Trie bucket of a few hundred words only comes out of a file (the reference can `remove`; this path cannot): an insert that touches
    it must leave the ascending list — every length class has its sorted route for `res_trie` runs, the 256-slot class of round 6 included.
    The short Tries are made here: words inserted in ascending order leave every Vec bucket ascending, the exported buckets are installed
    elsewhere with kind = Trie, the oracle loads those bytes; then random k-mers go into both (TrieVec::insert on a Trie,
    /root/reference/src/trievec/mod.rs:100-115)."""
    _need_gpu()
    rng = np.random.default_rng(77 * k + pb + n1)
    g, o = cbl_amd.CBL(k, pb), Oracle(k, pb)
    words = sorted({o.word_of_kmer(int(x)) for x in rng.integers(0, 4 ** k, size=n1, dtype=np.uint64)})
    first = [o.kmer_of_word(w) for w in words]
    assert g.insert_kmers(first).all()
    nb, nw, B = g.num_buckets(), g.count(), g.consts()["bytes"]
    prefix = torch.empty(nb, dtype=torch.int32, device="cuda")
    count = torch.empty(nb, dtype=torch.int32, device="cuda")
    kind = torch.empty(nb, dtype=torch.uint8, device="cuda")
    suffix = torch.empty(nw * B, dtype=torch.uint8, device="cuda")
    g.resident_export(prefix, count, kind, suffix)
    vec = kind == 0  # (the lowest prefixes hold more than 1024 words — necklace skew — and are Tries already)
    assert int((vec & (count > 128) & (count <= 256)).sum()) >= 4 and int((vec & (count > 32) & (count <= 128)).sum()) >= 4
    kind.fill_(1)
    g2, o2 = cbl_amd.CBL(k, pb), Oracle(k, pb)
    g2.install_buckets_device([(nb, nw, prefix, count, kind, suffix)])
    blob = g2.serialize()
    assert g2.validate(strict=False) == 0 and g2.bucket_table_np()[2].min() == 1
    o2.load(blob)
    assert o2.serialize() == blob and o2.count() == nw
    batch = rng.integers(0, 4 ** k, size=n2, dtype=np.uint64).tolist() + first[:: 7]
    assert g2.insert_kmers(batch).tolist() == [o2.insert_kmer(x) for x in batch]
    assert g2.count() == o2.count() and g2.serialize() == o2.serialize() and g2.validate(strict=False) == 0
    assert int((g2.bucket_table_np()[2] == 1).sum()) >= nb  # a Trie stays a Trie (the batch's new prefixes are Vecs)


def test_serializer_patches_huge_buckets_into_the_device_body(monkeypatch):
    """Low-complexity reads put > 8192 distinct words under one prefix (a 12-base poly-A run zeroes the whole 24-bit prefix):
    the entries of such buckets are emitted on the host and patched into the body the kernels emit for everything else."""
    _need_gpu()
    k, pb = 31, 24
    rng = random.Random(2024)
    seqs = [_rand_seq(rng, 40) + b"A" * 14 + _rand_seq(rng, 60) for _ in range(3000)] + [_rand_seq(rng, 150) for _ in range(3000)]
    bases, offsets = _concat(seqs)
    g, o = cbl_amd.CBL(k, pb), Oracle(k, pb)
    g.insert_seqs(bases, offsets)
    o.insert_seqs(bases, offsets)
    _, lens, kinds = g.bucket_table_np()
    assert lens.max() > 8192 and (lens <= 8192).sum() > 1000 and kinds[lens.argmax()] == 1
    blob = g.serialize()
    assert blob == o.serialize() and g.serialized_size() == len(blob)
    monkeypatch.setenv("CBLX_HOST_SERDE", "1")
    assert g.serialize() == blob
    monkeypatch.delenv("CBLX_HOST_SERDE")
    h = cbl_amd.CBL(k, pb)
    h.load(blob)
    assert h.serialize() == blob and h.count() == o.count()


def test_merge_into_an_empty_index_is_a_clone():
    """`empty |= other`: every bucket is other-only and is cloned as stored (/root/reference/src/wordset/set_ops.rs:123-157) —
    a device-side deep copy; `other` keeps working and the copy is independent of it."""
    _need_gpu()
    rng = random.Random(8)
    # (the arrays of the longer cases pass 1 MB and take the 16-byte copy kernel, k_copy16, with an 8-byte tail when the arena has an odd
    # number of words; the short ones take hipMemcpyAsync)
    for k, pb, canonical, n1 in ((31, 24, False, 30000), (13, 6, True, 30000), (59, 28, False, 30000), (31, 24, False, 400003), (59, 28, True, 250002), (27, 20, False, 300000)):
        s1, s2 = _rand_seq(rng, n1), _rand_seq(rng, 4000)
        g, o = cbl_amd.CBL(k, pb, canonical=canonical), Oracle(k, pb, canonical)
        g.insert_seq(s1), o.insert_seq(s1)
        e, oe = cbl_amd.CBL(k, pb, canonical=canonical), Oracle(k, pb, canonical)
        e |= g
        oe.merge(o)
        assert e.serialize() == oe.serialize() == g.serialize() and e.validate(strict=False) == 0
        e.insert_seq(s2), oe.insert_seq(s2)   # the clone is a full index of its own
        assert e.serialize() == oe.serialize() and g.serialize() == o.serialize()
        g.insert_seq(s2), o.insert_seq(s2)
        assert g.serialize() == o.serialize() == e.serialize()


@pytest.mark.parametrize("k,pb,canonical,threads", [(31, 24, False, 4), (31, 10, False, 3), (59, 28, True, 5), (15, 6, False, 2), (25, 24, False, 16)])
def test_parallel_index_loader(k, pb, canonical, threads, monkeypatch):
    """Big index files are cut speculatively at recognised entry starts and parsed by several threads (forced here on a
    small file); the result is the sequential loader's, and malformed input still fails like it."""
    _need_gpu()
    bases, offsets = synth.reads(5, 6000, 150)
    g = cbl_amd.CBL(k, pb, canonical=canonical)
    g.insert_seqs(bases, offsets)
    blob = g.serialize()
    monkeypatch.setenv("CBLX_LOAD_THREADS", "1")
    a = cbl_amd.CBL(k, pb)
    a.load(blob)
    monkeypatch.setenv("CBLX_LOAD_THREADS", str(threads))
    b = cbl_amd.CBL(k, pb)
    b.load(blob)
    assert b.serialize() == a.serialize() == blob and b.count() == g.count() and b.is_canonical() == canonical
    assert b.validate(strict=False) == 0
    extra = _rand_seq(random.Random(1), 500)
    b.insert_seq(extra), g.insert_seq(extra)
    assert b.serialize() == g.serialize()
    for bad in (blob[:-1], blob + b"\0", blob[: len(blob) // 2], blob[: len(blob) // 3] + blob[len(blob) // 3 + 1 :]):  # truncated, trailing, a byte lost
        with pytest.raises(cbl_amd.CblxError):
            cbl_amd.CBL(k, pb).load(bad)


# ---- the N-GPU code path with real device steps on ONE GPU: two processes share cuda:0, the collectives go through gloo ----
def _two_rank_worker(rank, world, port, k, pb, canonical, protocol, per, L, q):
    import torch.distributed as dist

    from cbl_amd import sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = cbl_amd.CBL(k, pb, canonical=canonical, device=0)
        sb = sharded.ShardedBuilder(g, sharded.HostStagedGroup(dist), slices=3, protocol=protocol)
        for batch, n in enumerate(per[rank]):  # two batches; the second reuses the first one's splitters
            first = sum(per[r][bb] for r in range(world) for bb in range(batch)) + sum(per[r][batch] for r in range(rank))
            d_b, d_o = synth.reads_torch(23, n, L, first_read=first, device="cuda:0")
            sb.insert_seqs_device(d_b, d_o, n)
        blob = sharded.gather_serialized(g.serialize(), dist)
        if rank == 0:
            q.put((blob, [int(x) for x in sb.bounds], g.count()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,pb,canonical,protocol", [(2, 31, 24, False, "sorted"), (2, 31, 24, True, "words"), (2, 59, 28, False, "sorted"),
                                                         (2, 25, 12, False, "sorted"), (4, 31, 24, False, "sorted"), (3, 31, 24, False, "words"), (8, 31, 24, False, "sorted")])
def test_two_ranks_on_one_gpu_through_a_gloo_shim(world, k, pb, canonical, protocol):
    _need_gpu()
    import socket


    from cbl_amd.sharded import ShardedBuilder

    L = 150 if k < 59 else 250
    per = [(700, 2), (300, 450), (1, 600), (512, 0), (64, 64), (0, 900), (333, 5), (90, 90)][:world]  # reads per batch of every rank (ragged; fewer reads than slices; none)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = _spawn_context()
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, k, pb, canonical, protocol, per, L, q)) for r in range(world)]
    for p in procs:
        p.start()
    blob, bounds, count0 = q.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    one = Oracle(k, pb, canonical)  # stream order of the job: per batch, slice-major then rank-minor
    for batch in range(2):
        first = sum(per[r][bb] for r in range(world) for bb in range(batch))
        starts = [first + sum(per[rr][batch] for rr in range(r)) for r in range(world)]
        sl = [ShardedBuilder.slice_bounds(per[r][batch], 3) for r in range(world)]
        for c in range(3):
            for r in range(world):
                a, b = sl[r][c]
                if b > a:
                    hb, ho = synth.reads(23, b - a, L, first_read=starts[r] + a)
                    one.insert_seqs(hb, ho)
    assert blob == one.serialize()
    assert len(bounds) == world - 1 and 0 < count0 < one.count()  # rank 0 owns part of the index, not all of it


# ---- prefix-range sharded indexes (cfg 5): one rank's share through the C ABI, then N ranks sharing this GPU ---------------
def _bytes_of_shards(shards, canonical):
    """header + the bodies of the given ctxs in order (what ShardedIndex.save_to_file writes)."""
    from cbl_amd.sharded import _varint

    n = sum(g.serialized_body_size()[0] for g in shards)
    out = bytes([int(canonical)]) + _varint(n)
    for g in shards:
        blob = g.serialize()
        ne, nb = g.serialized_body_size()
        assert len(blob) - nb in (2, 4, 6, 10) and ne == g.num_buckets()
        out += blob[len(blob) - nb:]
    return out


@pytest.mark.parametrize("k,pb,canonical,nreads,L", [(31, 24, False, 20000, 150), (11, 8, False, 30, 5000), (59, 28, True, 3000, 250), (15, 6, False, 2000, 150)])
def test_load_shard_export_install_write_body(k, pb, canonical, nreads, L, tmp_path):
    """cblx_load_shard_from_file / cblx_resident_split / _export / cblx_install_buckets_device / cblx_write_body_at on one GPU:
    the shares of W ranks, loaded one after the other, are disjoint, exact, and concatenate to the file; a share exported and
    installed elsewhere serializes to the same bytes (kinds and stored order kept)."""
    _need_gpu()
    hb, ho = synth.reads(9, nreads, L)
    o = Oracle(k, pb, canonical)
    o.insert_seqs(hb, ho)
    blob = o.serialize()
    path = tmp_path / "whole.cbl"
    path.write_bytes(blob)
    for world in (1, 2, 3, 8):
        for bounds_mode in ("bytes", "given"):
            for sequential in (False, True):
                shards, infos, bounds = [], [], None
                if bounds_mode == "given":
                    _offs, first, _ok = cbl_amd.index_shard_cuts(path, k, pb, world, sequential=True)
                    bounds = first[1:world] + (1 if world > 1 else 0)  # between the stored prefixes
                for r in range(world):
                    g = cbl_amd.CBL(k, pb)
                    info, b = g.load_shard_from_file(path, r, world, bounds, sequential)
                    assert info["exact"] == 1 and info["canonical"] == int(canonical) and g.is_canonical() == canonical
                    assert info["local_entries"] == g.num_buckets() and g.validate(strict=False) == 0
                    if bounds is not None:
                        assert (b == bounds).all()
                    shards.append(g)
                    infos.append(info)
                assert sum(i["local_entries"] for i in infos) == infos[0]["header_entries"] == o.n_buckets()
                assert all(infos[i]["end_off"] == infos[i + 1]["begin_off"] for i in range(world - 1)) and infos[-1]["end_off"] == len(blob)
                assert _bytes_of_shards(shards, canonical) == blob
                # rank-ordered save: header by "rank 0", every share at its own offset
                out = tmp_path / "out.cbl"
                from cbl_amd.sharded import _varint

                header = bytes([int(canonical)]) + _varint(o.n_buckets())
                sizes = [g.serialized_body_size()[1] for g in shards]
                with open(out, "wb") as f:
                    f.write(header)
                    f.truncate(len(header) + sum(sizes))
                for r in reversed(range(world)):
                    shards[r].write_body_at(out, len(header) + sum(sizes[:r]))
                assert out.read_bytes() == blob
    # export -> split -> install: a whole index re-cut into 3 pieces, each installed in a fresh ctx
    g = cbl_amd.CBL(k, pb, canonical=canonical)
    g.load(blob)
    nb, nw, B = g.num_buckets(), g.count(), g.consts()["bytes"]
    prefix = torch.empty(nb, dtype=torch.int32, device="cuda")
    count = torch.empty(nb, dtype=torch.int32, device="cuda")
    kind = torch.empty(nb, dtype=torch.uint8, device="cuda")
    suffix = torch.empty(nw * B, dtype=torch.uint8, device="cuda")
    g.resident_export(prefix, count, kind, suffix)
    pv = prefix.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    cuts = np.array([int(pv[nb // 3]), int(pv[2 * nb // 3]) + 1], dtype=np.uint32)
    bs, ws = g.resident_split(cuts, 3)
    assert bs[0] == ws[0] == 0 and bs[3] == nb and ws[3] == nw
    assert bs[1] == int((pv < cuts[0]).sum()) and bs[2] == int((pv < cuts[1]).sum())
    pieces = []
    for d in range(3):
        t = cbl_amd.CBL(k, pb, canonical=canonical)
        t.install_buckets_device([(bs[d + 1] - bs[d], ws[d + 1] - ws[d], prefix[bs[d]: bs[d + 1]], count[bs[d]: bs[d + 1]], kind[bs[d]: bs[d + 1]],
                                   suffix[ws[d] * B: ws[d + 1] * B])])
        assert t.validate(strict=False) == 0
        pieces.append(t)
    assert _bytes_of_shards(pieces, canonical) == blob
    whole = cbl_amd.CBL(k, pb, canonical=canonical)  # several parts in one call
    whole.install_buckets_device([(bs[d + 1] - bs[d], ws[d + 1] - ws[d], prefix[bs[d]: bs[d + 1]], count[bs[d]: bs[d + 1]], kind[bs[d]: bs[d + 1]],
                                   suffix[ws[d] * B: ws[d + 1] * B]) for d in range(3)])
    assert whole.serialize() == blob
    hb2, ho2 = synth.reads(10, 50, L)
    whole.insert_seqs(hb2, ho2)  # an installed index is a full index
    o.insert_seqs(hb2, ho2)
    assert whole.serialize() == o.serialize()
    with pytest.raises(cbl_amd.CblxError):  # parts out of order
        cbl_amd.CBL(k, pb).install_buckets_device([(bs[d + 1] - bs[d], ws[d + 1] - ws[d], prefix[bs[d]: bs[d + 1]], count[bs[d]: bs[d + 1]], kind[bs[d]: bs[d + 1]],
                                                   suffix[ws[d] * B: ws[d + 1] * B]) for d in (1, 0, 2) if bs[d + 1] > bs[d]])


def test_load_rejects_malformed_trie_and_empty_vec():
    """node values out of order / an empty Vec must be CBLX_EFORMAT, not a broken resident index."""
    _need_gpu()
    k, pb = 11, 8
    o = Oracle(k, pb)
    hb, ho = synth.reads(3, 10, 4000)
    o.insert_seqs(hb, ho)
    blob = bytearray(o.serialize())
    g = cbl_amd.CBL(k, pb)
    g.load(bytes(blob))
    kinds = g.bucket_table_np()[2]
    assert kinds.max() == 1  # there is a Trie entry to damage
    # first entry: prefix, tag; find the first Trie entry's root node and swap its first two values
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from shard_standin import _rv, parse_index

    ents = parse_index(bytes(blob), g.consts()["bytes"])[1]
    start = next(e[3] for e in ents if e[1] == 1)
    _p, q = _rv(blob, start)
    _t, q = _rv(blob, q)
    c, q = _rv(blob, q)
    assert c >= 2
    blob[q], blob[q + 1] = blob[q + 1], blob[q]
    with pytest.raises(cbl_amd.CblxError) as e:
        g.load(bytes(blob))
    assert e.value.code == cbl_amd.EFORMAT
    empty_vec = bytes([0, 1, 5, 0, 0])  # one entry: prefix 5, Vec, 0 elements
    with pytest.raises(cbl_amd.CblxError) as e:
        g.load(empty_vec)
    assert e.value.code == cbl_amd.EFORMAT


def _sharded_index_worker(rank, world, port, k, pb, canonical, per, L, slices, tmp, q):
    import sys

    import torch.distributed as dist

    from cbl_amd import sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sharded.MAX_MSG_BYTES = 1 << 16
    try:
        grp = sharded.HostStagedGroup(dist)

        def new():
            return sharded.ShardedIndex(k, pb, grp, canonical=canonical, device=0, slices=slices)

        def feed(idx, seed):
            d_b, d_o = synth.reads_torch(seed, per, L, first_read=rank * per, device="cuda:0")
            idx.insert_seqs_device(d_b, d_o, per)

        A, B = new(), new()
        feed(A, 31)
        nb = np.asarray(A.bounds, dtype=np.uint64) * 3 // 2 + 1
        nb[-1] = max(int(nb[-1]), (1 << pb) - 2)
        B.bounds = np.minimum(nb, (1 << pb) - 1).astype(np.uint32)
        feed(B, 77)
        cA, cB = A.count(), B.count()
        E = A.clone()
        assert E.count() == cA
        F = new()
        feed(F, 5)
        assert F.copy_from(A).count() == cA and np.array_equal(F.bounds, A.bounds)
        R = B.resharded(A.bounds)  # B itself is left alone
        assert R.count() == cB and np.array_equal(R.bounds, A.bounds) and not np.array_equal(B.bounds, A.bounds)
        A.merge_assign(B)
        assert np.array_equal(A.bounds, B.bounds) and B.count() == cB
        pa, pb_, pe = (os.path.join(tmp, n) for n in ("a.cbl", "b.cbl", "e.cbl"))
        sa = A.save_to_file(pa)
        B.save_to_file(pb_)
        E.save_to_file(pe)
        C, D = new(), new()
        C.load_from_file(pa)           # byte-balanced ranges, speculative entry starts
        D.load_from_file(pe, bounds=C.bounds)
        assert C.count() == A.count() and D.count() == cA
        pc = os.path.join(tmp, "c.cbl")
        assert C.save_to_file(pc) == sa
        D.merge_assign(C)
        pd = os.path.join(tmp, "d.cbl")
        D.save_to_file(pd)
        if rank == 0:
            q.put({n: open(os.path.join(tmp, n + ".cbl"), "rb").read() for n in "abecd"})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,pb,canonical,per,L,slices", [(2, 31, 24, False, 3000, 150, 2), (3, 11, 8, False, 6, 7000, 3), (2, 59, 28, True, 600, 250, 1),
                                                              (4, 31, 24, False, 20000, 150, 4), (8, 15, 10, False, 300, 500, 2)])
def test_sharded_index_merge_load_save_on_one_gpu(world, k, pb, canonical, per, L, slices, tmp_path):
    """cfg 5 with `world` ranks sharing this GPU (real device steps, exchange staged through gloo): sharded build of two
    operands at different bounds, clone, re-shard, `A |= B`, rank-ordered save, per-range load, second merge — all files
    byte-identical to the one-process oracle's, quirks of the reference's |= included."""
    _need_gpu()
    import socket


    from cbl_amd.sharded import ShardedBuilder

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = _spawn_context()
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_index_worker, args=(r, world, port, k, pb, canonical, per, L, slices, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    files = q.get(timeout=1200)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0

    def one_process(seed):
        o = Oracle(k, pb, canonical)
        for a, b in ShardedBuilder.slice_bounds(per, slices):
            for r in range(world):
                if b > a:
                    hb, ho = synth.reads(seed, b - a, L, first_read=r * per + a)
                    o.insert_seqs(hb, ho)
        return o

    oa, ob = one_process(31), one_process(77)
    a_before = oa.serialize()
    oa.merge(ob)
    assert files["e"] == a_before
    assert files["a"] == oa.serialize()
    assert files["b"] == ob.serialize()
    assert files["c"] == files["a"]
    od, oc = Oracle(k, pb, canonical), Oracle(k, pb, canonical)
    od.load(files["e"])
    oc.load(files["a"])
    od.merge(oc)
    assert files["d"] == od.serialize()


# ---- one file on N ranks: staged block-cyclic reader, file-order parity -------------------------------------------------------
def _ragged_fasta(path, seed, nrec, fastq=False):
    rng = random.Random(seed)
    recs = []
    with open(path, "wb") as f:
        for i in range(nrec):
            n = rng.choice([64, 150, 151, 300, 777, 2500, 5000])
            s = _rand_seq(rng, n)
            recs.append(s)
            if fastq:
                f.write(b"@r%d\n" % i + s + b"\n+\n" + b"I" * n + b"\n")
            else:
                f.write(b">r%d some text\n" % i)
                w = rng.choice([60, 80, 10_000])
                for a in range(0, n, w):
                    f.write(s[a: a + w] + (b"\r\n" if i % 7 == 3 else b"\n"))
    return recs


@pytest.mark.parametrize("fastq", [False, True])
def test_stage_fastx_blocks(tmp_path, fastq):
    """cblx_stage_fastx_blocks: the staged arrays hold exactly the records block-cyclic dealing gives the rank, in file
    order; nothing is inserted; the ctx keeps working for device inserts and refuses host inserts until the release."""
    _need_gpu()
    path = tmp_path / ("r.fq" if fastq else "r.fa")
    recs = _ragged_fasta(path, 3, 41, fastq)
    g = cbl_amd.CBL(31, 24)
    assert g.count_fastx_records(path) == 41 and g.count() == 0
    from cbl_amd.sharded import GpuEngine

    eng = GpuEngine(g)
    for world, block in ((1, 7), (3, 4), (4, 1), (2, 100)):
        for rank in range(world):
            bases, offsets, n, n_file = eng.stage_fastx_blocks(str(path), block, rank, world)
            mine = [r for i, r in enumerate(recs) if (i // block) % world == rank]
            assert n_file == 41 and n == len(mine)
            off = offsets.cpu().numpy()
            assert off[0] == 0 and list(np.diff(off)) == [len(r) for r in mine]
            assert bases[: int(off[-1])].cpu().numpy().tobytes() == b"".join(mine)
            assert g.count() == 0  # observers do not consume what is staged
            with pytest.raises(cbl_amd.CblxError):
                g.insert_seq(recs[0])
            if n:
                t = cbl_amd.CBL(31, 24)
                t.insert_seqs_device(bases, offsets, n)
                o = Oracle(31, 24)
                for r in mine:
                    o.insert_seq(r)
                assert t.serialize() == o.serialize()
            eng.stage_release()
    g.insert_seq(recs[0])
    assert g.count() == len(recs[0]) - 30


def _file_order_worker(rank, world, port, k, pb, canonical, path, block, protocol, q):
    import torch.distributed as dist

    from cbl_amd import sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        idx = sharded.ShardedIndex(k, pb, sharded.HostStagedGroup(dist), canonical=canonical, device=0, slices=3, protocol=protocol)
        n = idx.insert_fastx_file(path, block)
        n2 = idx.insert_fastx_file(path)  # again, block size chosen by the builder: nothing new
        out = path + ".cbl"
        idx.save_to_file(out)
        if rank == 0:
            q.put((open(out, "rb").read(), n, n2))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,pb,canonical,nrec,block,protocol", [(2, 31, 24, False, 200, 16, "sorted"), (3, 31, 24, True, 61, 1, "words"), (4, 59, 28, False, 90, 7, "sorted"),
                                                                    (2, 25, 12, False, 3, 50, "sorted")])
def test_sharded_build_of_one_file_has_file_order_on_one_gpu(world, k, pb, canonical, nrec, block, protocol, tmp_path):
    """`cbl build file` on N ranks sharing this GPU (exchange staged through gloo): records dealt block-cyclically, so the
    job's stream order is the file's and the saved index is byte-identical to the one-process build in file order."""
    _need_gpu()
    import socket


    path = str(tmp_path / "reads.fa")
    recs = _ragged_fasta(path, nrec, nrec)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = _spawn_context()
    q = ctx.Queue()
    procs = [ctx.Process(target=_file_order_worker, args=(r, world, port, k, pb, canonical, path, block, protocol, q)) for r in range(world)]
    for p in procs:
        p.start()
    blob, n, n2 = q.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert n == n2 == nrec
    one = Oracle(k, pb, canonical)
    for r in recs:
        one.insert_seq(r)
    assert blob == one.serialize()


# ---- the multi-GPU build behind the C ABI: cblx_comm + cblx_sharded_insert_seqs_device ----------------------------------------
def test_native_sharded_insert_single_rank_rccl():
    """One rank, the REAL RCCL communicator (librccl looked up at run time, unique id, comm init, grouped send/recv to self):
    cblx_sharded_insert_seqs_device == the direct insert, slices and repeated batches included."""
    _need_gpu()
    uid = cbl_amd.Comm.unique_id()
    assert len(uid) == 128
    comm = cbl_amd.Comm.rccl(uid, 0, 1, 0)
    for k, pb, canonical, proto in ((31, 24, False, "bins"), (59, 28, True, "bins"), (25, 12, False, "bins"), (31, 24, True, "sorted"), (27, 8, False, "bins"),
                                    (59, 28, False, "sorted"), (31, 28, True, "bins"), (15, 20, False, "bins")):
        comm.set_protocol(proto)  # "bins" at PREFIX_BITS = 8 falls back to "sorted" inside the library
        d_b, d_o = synth.reads_torch(42, 3000, 150, device="cuda")
        a, b = cbl_amd.CBL(k, pb, canonical=canonical), cbl_amd.CBL(k, pb, canonical=canonical)
        a.insert_seqs_device(d_b, d_o, 3000)
        bounds = np.zeros(0, dtype=np.uint32)
        assert b.sharded_insert_seqs_device(comm, d_b, d_o, 3000, [0, 700, 700, 2999, 3000], bounds, False)
        assert b.serialize() == a.serialize()
        b.sharded_insert_seqs_device(comm, d_b, d_o, 3000, [0, 3000], bounds, True)  # again: nothing new
        assert b.serialize() == a.serialize()
        hb, ho = synth.reads(42, 3000, 150)
        o = Oracle(k, pb, canonical)
        o.insert_seqs(hb, ho)
        assert b.serialize() == o.serialize()
    st = comm.stats()
    assert st["sent_bytes"] == 0 and st["recv_bytes"] == 0  # one rank: its own runs only
    comm.close()


def _native_worker(rank, world, port, k, pb, canonical, per, L, path, q, protocol="bins", groups=0):
    import torch.distributed as dist

    from cbl_amd import sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = cbl_amd.Comm.over_group(dist, rank, world, 0)  # host callbacks: the ranks share this GPU
        comm.set_recv_groups(groups)  # 0: the default (4 groups per rank), 1: the ungrouped receiver
        g = cbl_amd.CBL(k, pb, canonical=canonical, device=0)
        sb = sharded.ShardedBuilder(g, dist, slices=3, comm=comm, protocol=protocol)
        used, fine = [], []
        for batch, n in enumerate(per[rank]):  # two batches; the second reuses the first one's splitters (and meets a non-empty index)
            first = sum(per[r][bb] for r in range(world) for bb in range(batch)) + sum(per[r][batch] for r in range(rank))
            d_b, d_o = synth.reads_torch(23, n, L, first_read=first, device="cuda:0")
            sb.insert_seqs_device(d_b, d_o, n)
            used.append(comm.groups_used())
            fine.append(comm.groups_fine())
        blob = sharded.gather_serialized(g.serialize(), dist)
        fblob = None
        if path:
            # one file, the parse shared between the ranks (cblx_stage_fastx_blocks_comm): regions of 3000 bytes so that even this
            # small file is cut into many; a given block size, then the library's choice (block = 0), then a FASTQ file
            os.environ["CBLX_FASTX_REGION_BYTES"] = "3000"
            fblob = []
            for pth, blk in ((path, 5), (path, 0), (path + ".fq", 0), (path + ".short", 4)):
                h = cbl_amd.CBL(k, pb, canonical=canonical, device=0)
                sf = sharded.ShardedBuilder(h, dist, slices=3, comm=comm, protocol=protocol)
                try:
                    nrec = sf.insert_fastx_file(pth, blk)
                    fblob.append((nrec, sharded.gather_serialized(h.serialize(), dist)))
                except cbl_amd.CblxError as e:  # the file with a record shorter than K: ESHORT on the rank that owns it
                    fblob.append(("error", e.code))
                    break
        if rank == 0:
            q.put((blob, [int(x) for x in sb.bounds], g.count(), fblob, sb.stats["sent_bytes"], used, fine))
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,pb,canonical,protocol,groups", [
    (2, 31, 24, False, "bins", 0), (3, 59, 28, True, "bins", 0), (4, 25, 12, False, "bins", 0), (8, 31, 24, False, "bins", 0),
    (3, 31, 28, True, "bins", 0), (2, 21, 16, False, "bins", 0), (3, 33, 25, False, "bins", 0),  # (16-byte records, FINE bins; five ranks sharing the GPU spend 30 - 90 s in their first allocations: one such case is kept, under "auto")
    (2, 31, 24, False, "bins", 1), (3, 59, 28, True, "bins", 1), (8, 31, 24, False, "bins", 1), (3, 31, 28, True, "bins", 1),  # the ungrouped receiver
    (2, 31, 24, False, "bins", 3), (4, 31, 28, False, "bins", 14), (3, 45, 20, True, "bins", 5), (2, 27, 9, False, "bins", 4),
    (4, 31, 25, True, "bins", 3), (2, 33, 27, False, "bins", 5),  # FINE bins where a 65..72-bit word's bins must imply 21 prefix bits; 16-byte records
    (2, 31, 24, False, "sorted", 0), (3, 59, 28, True, "sorted", 0), (8, 31, 24, False, "sorted", 0),
    (2, 31, 28, False, "auto", 0), (4, 31, 24, False, "auto", 0), (5, 31, 24, True, "auto", 0),  # the library's choice: "replicate" on 2 - 3 ranks, "sorted" on 4, "bins" from 5 on
    # "replicate" (round 6): reads cross as bit planes, every rank transforms all of them and keeps its prefix range — ragged and empty shards, canonical,
    # 16-byte records, FINE bins, the ungrouped fallback of the second batch (a non-empty index)
    (2, 31, 24, False, "replicate", 0), (3, 59, 28, True, "replicate", 0), (4, 31, 28, False, "replicate", 3), (2, 31, 26, True, "replicate", 5),
    (3, 25, 12, False, "replicate", 0), (4, 33, 27, False, "replicate", 2)])
def test_native_sharded_insert_on_one_gpu_through_callbacks(world, k, pb, canonical, protocol, groups, tmp_path):
    """The C++ orchestration of the multi-GPU build (slices, splitter choice, count exchange, grouped exchange, batch merge)
    with `world` ranks sharing this GPU and the bytes moved by host callbacks over gloo: byte-identical to the one-process
    oracle in the job's stream order (slice-major, rank-minor), and to the file's order for a file dealt block-cyclically."""
    _need_gpu()
    import socket


    from cbl_amd.sharded import ShardedBuilder

    L = 150 if k < 59 else 250
    per = [(700, 2), (300, 450), (1, 600), (512, 0), (64, 64), (0, 900), (333, 5), (90, 90)][:world]
    path = str(tmp_path / "reads.fa")
    recs = _ragged_fasta(path, 77, 43)
    qrecs = _ragged_fasta(path + ".fq", 78, 29, fastq=True)
    with open(path + ".short", "wb") as f:  # record 9 is shorter than K: every rank falls back to the sequential reader, its owner reports it
        for i, r in enumerate(recs[:12]):
            f.write(b">s%d\n" % i + (r[:20] if i == 9 else r) + b"\n")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = _spawn_context()
    q = ctx.Queue()
    procs = [ctx.Process(target=_native_worker, args=(r, world, port, k, pb, canonical, per, L, path, q, protocol, groups)) for r in range(world)]
    for p in procs:
        p.start()
    blob, bounds, count0, fblob, sent, used, fine = q.get(timeout=420)
    # the grouped receiver (bins protocol, empty index): rank 0 worked its range off in groups; a second batch meets a non-empty index
    # and takes the ungrouped path; groups = 1 switches it off
    assert used[1] == 0
    if protocol == "auto":
        protocol = cbl_amd.Comm.auto_protocol(world)
    if protocol == "sorted" or groups == 1:
        assert used[0] == 0
    elif pb >= 12:  # (a rank whose range is narrower than a few histogram cells gets fewer groups than asked for, down to one)
        assert 1 <= used[0] <= (groups or 4), used
    # PREFIX_BITS > 24: the first pass ran on FINE bins, and rank 0 — the narrowest prefix range — sorted 16 bits in its groups, not 20
    assert fine[1] == 0 and (fine[0] >= 1 if (pb > 24 and protocol in ("bins", "replicate") and groups != 1) else fine[0] == 0), fine
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    one = Oracle(k, pb, canonical)
    for batch in range(2):
        first = sum(per[r][bb] for r in range(world) for bb in range(batch))
        starts = [first + sum(per[rr][batch] for rr in range(r)) for r in range(world)]
        sl = [ShardedBuilder.slice_bounds(per[r][batch], 3) for r in range(world)]
        for c in range(3):
            for r in range(world):
                a, b = sl[r][c]
                if b > a:
                    hb, ho = synth.reads(23, b - a, L, first_read=starts[r] + a)
                    one.insert_seqs(hb, ho)
    assert blob == one.serialize()
    assert len(bounds) == world - 1 and 0 < count0 < one.count() and sent > 0
    of, oq = Oracle(k, pb, canonical), Oracle(k, pb, canonical)
    for r in recs:
        of.insert_seq(r)
    for r in qrecs:
        oq.insert_seq(r)
    assert fblob[0] == (len(recs), of.serialize()) and fblob[1] == (len(recs), of.serialize())
    assert fblob[2] == (len(qrecs), oq.serialize())


@pytest.mark.parametrize("world,k,pb,canonical,groups,slices,gbps,deep", [
    (4, 31, 24, False, 0, 2, 0.0, None), (4, 31, 24, False, 1, 3, 0.0, None), (8, 31, 28, True, 6, 1, 2.0, None), (3, 59, 28, False, 4, 2, 0.0, None),
    (8, 31, 28, False, 4, 3, 0.0, "fine=0"), (2, 31, 26, True, 4, 2, 0.0, None), (8, 31, 28, False, 0, 3, 0.0, None),  # FINE bins off (three passes of 7 + 7 + 6 bits) / on
    (2, 25, 16, False, 0, 1, 1.0, None), (8, 31, 24, False, 1, 4, 2.0, None),
    # the two layouts of a group (last pass into the slot / into the scratch with the long-run path's twin as the slot), forced; short buckets and
    # PREFIX_BITS = 6 buckets of thousands of words (the long-run path proper)
    (4, 31, 24, False, 5, 2, 0.0, "1"), (4, 31, 24, False, 5, 2, 0.0, "0"), (2, 31, 9, False, 3, 1, 0.0, "1"), (2, 31, 9, True, 3, 2, 0.0, "0"), (3, 59, 12, False, 4, 1, 0.0, "1")])
def test_rehearsal_of_rank_0_on_recorded_senders(world, k, pb, canonical, groups, slices, gbps, deep, monkeypatch):
    """cblx_comm_init_sim (the paced one-GPU rehearsal tools/emulate_wire.py times): ranks 1 .. W-1 record what they would send rank 0,
    rank 0 replays it — grouped receiver or not, paced or not — and ends up with exactly its range of the job's index: the entries of
    the one-process oracle's file (stream order slice-major, rank-minor) start with rank 0's entries."""
    _need_gpu()
    from cbl_amd.sharded import ShardedBuilder, _read_varint

    fine_on = deep != "fine=0"
    if not fine_on:
        monkeypatch.setenv("CBLX_FINE_BINS", "0")
        deep = None
    if deep is not None:
        monkeypatch.setenv("CBLX_GROUP_DEEP", deep)
    L, nr, store = (150 if k < 59 else 250), (2500 if pb > 12 else 12000), 77 + world * 16 + groups + (100 if deep else 0) + (0 if fine_on else 1000) + pb
    bounds = np.zeros(world - 1, dtype=np.uint32)
    valid = False
    cuts = [nr * s // slices for s in range(slices + 1)]
    for r in list(range(1, world)) + [0, 0]:  # rank 0 twice: a replay can be repeated
        d_b, d_o = synth.reads_torch(5, nr, L, first_read=r * nr, device="cuda")
        cm = cbl_amd.Comm.sim(r, world, store, gbps if r == 0 else 0.0)
        cm.set_protocol("bins")  # (the default, "auto", is "replicate" / "sorted" up to four ranks)
        cm.set_recv_groups(groups)
        g = cbl_amd.CBL(k, pb, canonical=canonical)
        valid = g.sharded_insert_seqs_device(cm, d_b, d_o, nr, cuts, bounds, valid)
        if r == 0:
            used, blob0, st, fine = cm.groups_used(), g.serialize(), cm.stats(), cm.groups_fine()
        g.close()
        cm.close()
    cbl_amd.Comm.sim_store_free(store)
    assert used == 0 if groups == 1 else 1 <= used <= (groups or 4)
    # PREFIX_BITS > 24: rank 0's groups (the narrowest of the job) sort 16 prefix bits behind the senders' first pass: FINE bins
    assert fine == (used if (pb > 24 and fine_on and groups != 1) else 0), (fine, used)
    assert st["recv_bytes"] > 0 and st["sent_bytes"] > 0
    one = Oracle(k, pb, canonical)
    for c in range(slices):
        for r in range(world):
            if cuts[c + 1] > cuts[c]:
                hb, ho = synth.reads(5, cuts[c + 1] - cuts[c], L, first_read=r * nr + cuts[c])
                one.insert_seqs(hb, ho)
    full = one.serialize()
    n0, p0 = _read_varint(blob0, 1)
    nf, pf = _read_varint(full, 1)
    assert 0 < n0 < nf and blob0[0] == full[0]
    assert full[pf:pf + len(blob0) - p0] == blob0[p0:]
    # ... and the next entry of the job's file belongs to rank 1: its prefix is at or above the bound
    nxt, _ = _read_varint(full, pf + len(blob0) - p0)
    assert nxt >= int(bounds[0])


@pytest.mark.parametrize("world,k,pb,canonical,tgt,slices,gbps", [(2, 31, 28, False, 1, 1, 0.0), (3, 31, 24, True, 0, 2, 1.0), (4, 59, 27, False, 2, 3, 0.0), (2, 25, 12, False, 0, 1, 0.0)])
def test_rehearsal_of_any_rank_on_the_replicate_protocol(world, k, pb, canonical, tgt, slices, gbps, monkeypatch):
    """"replicate" under the one-GPU rehearsal (what tools/emulate_wire.py --protocol replicate times): the other ranks record the planes and offsets
    they would send, the rehearsed rank transforms every rank's reads and keeps its prefix range — exactly the buckets of that range of the
    one-process oracle's index (stream order slice-major, rank-minor), paced wire or not."""
    _need_gpu()
    monkeypatch.setenv("CBLX_SIM_TARGET", str(tgt))
    L, nr, store = (150 if k < 59 else 250), 3000, 88000 + world * 10 + pb
    bounds = np.zeros(world - 1, dtype=np.uint32)
    valid = False
    cuts = [nr * s // slices for s in range(slices + 1)]
    for r in [x for x in range(world) if x != tgt] + [tgt]:
        d_b, d_o = synth.reads_torch(91, nr, L, first_read=r * nr, device="cuda")
        cm = cbl_amd.Comm.sim(r, world, store, gbps if r == tgt else 0.0)
        cm.set_protocol("replicate")
        g = cbl_amd.CBL(k, pb, canonical=canonical)
        valid = g.sharded_insert_seqs_device(cm, d_b, d_o, nr, cuts, bounds, valid)
        if r == tgt:
            mine, used, st, proto = g.buckets(), cm.groups_used(), cm.stats(), cm.protocol_used()
            assert g.validate() == 0
        g.close()
        cm.close()
    cbl_amd.Comm.sim_store_free(store)
    assert proto == "replicate" and used >= 1
    # what crossed: (world - 1) ranks' planes (6 bytes per 16 bases) and offsets — a fraction of a byte per k-mer
    assert 0 < st["recv_bytes"] < (world - 1) * (nr * L * 6 // 16 + nr * 8 + 4096)
    one = Oracle(k, pb, canonical)
    for c in range(slices):
        for r in range(world):
            if cuts[c + 1] > cuts[c]:
                hb, ho = synth.reads(91, cuts[c + 1] - cuts[c], L, first_read=r * nr + cuts[c])
                one.insert_seqs(hb, ho)
    full = cbl_amd.CBL(k, pb, canonical=canonical)
    full.load(one.serialize())
    lo, hi = (int(bounds[tgt - 1]) if tgt else 0), (int(bounds[tgt]) if tgt + 1 < world else 1 << pb)
    want = [b for b in full.buckets() if lo <= b[0] < hi]
    full.close()
    assert len(mine) == len(want) and mine == want


def _reads_with_all_ones(seed, nr, L, first_read, n_poly, n_rich):
    """Random reads, `n_poly` of them replaced by poly-G (every k-mer is the all-ones word: A=0 C=1 T=2 G=3, the one necklace in the upper half of
    the prefix space) and `n_rich` by G-rich reads (a C or a T every dozen bases: necklaces 0111..., prefixes right below 2^(PREFIX_BITS-1))."""
    hb, ho = synth.reads(seed, nr, L, first_read=first_read)
    b = np.array(hb, copy=True).reshape(nr, L)
    rng = random.Random(seed * 7919 + first_read)
    rows = rng.sample(range(nr), n_poly + n_rich)
    for i in rows[:n_poly]:
        b[i, :] = ord("G")
    for i in rows[n_poly:]:
        b[i, :] = ord("G")
        for j in range(rng.randrange(12), L, 12):
            b[i, j] = ord(rng.choice("CT"))
    return np.ascontiguousarray(b.reshape(-1)), ho


@pytest.mark.parametrize("k,pb,n_poly", [(31, 28, 3), (31, 28, 2300), (27, 25, 2300), (59, 28, 1400)])
def test_fine_bins_build_with_the_all_ones_word(k, pb, n_poly, monkeypatch):
    """FINE bins and the all-ones word (bin 255, the only necklace above 2^(PREFIX_BITS-1)): a few copies (a cold segment: its bucket start comes from
    the records) and more than 64 tiles of them (k_dir_gather's table route where the group sorts 16 bits, the fused directory where it sorts 24);
    non-canonical and canonical (62 / 118 one bits: the forward strand is the canonical one). Same bytes as the oracle and as the plain build."""
    _need_gpu()
    monkeypatch.setenv("CBLX_FINE_MIN", "0")
    L, nr = (150 if k < 59 else 250), 6000
    hb, ho = _reads_with_all_ones(5 + k, nr, L, 0, n_poly, 40)
    for canonical in (False, True):
        o = Oracle(k, pb, canonical)
        o.insert_seqs(hb, ho)
        for fine in ("1", "0"):
            monkeypatch.setenv("CBLX_FINE_BINS", fine)
            g = cbl_amd.CBL(k, pb, canonical=canonical)
            g.insert_seqs(hb, ho)
            g.flush()
            assert g.fine_builds() == (1 if fine == "1" else 0)
            _check_index(g, o)
            assert g.validate() == 0
            top = [b for b in g.buckets() if b[0] == (1 << pb) - 1]
            assert len(top) == 1 and len(top[0][2]) == 1  # the all-ones prefix holds the one all-ones suffix
            g.close()


@pytest.mark.parametrize("k,pb,world,n_poly,groups", [(27, 25, 2, 1300, 2), (27, 25, 2, 2, 3), (31, 26, 3, 900, 2), (31, 28, 2, 1300, 0)])
def test_rehearsal_of_the_last_rank_with_a_narrow_top_range(k, pb, world, n_poly, groups, monkeypatch):
    """The LAST rank of a FINE-bins plan whose range is a few blocks of 2^16 prefixes below 2^(PREFIX_BITS-1) (bounds given, not quantiles): every
    one of its groups sorts 16 bits behind the first pass, and the all-ones word rides in bin 255 of its last group — segment 255 of a group whose
    directory comes from the last pass's tables (k_dir_gather). Round 5 gave that segment the all-ones prefix ITSELF as its block base: every row of
    the segment's table then mapped to the one all-ones prefix, one with the bucket start and up to 65 535 with EMPTY, in no order, and a segment of
    more than 64 tiles (not cold) could lose its bucket. Poly-G reads from every rank, few and many; G-rich reads fill the range below."""
    _need_gpu()
    L, nr, store, slices = 150, 4000, 61000 + pb + n_poly, 2
    tgt = world - 1
    monkeypatch.setenv("CBLX_SIM_TARGET", str(tgt))
    half = 1 << (pb - 1)
    bounds = np.array([half - (3 + 2 * (world - 2 - i)) * 65536 + (0 if i else 4096) for i in range(world - 1)], dtype=np.uint32)
    cuts = [nr * s // slices for s in range(slices + 1)]
    reads = [_reads_with_all_ones(300 + k, nr, L, r * nr, n_poly, 600) for r in range(world)]
    for r in [x for x in range(world) if x != tgt] + [tgt]:
        d_b, d_o = torch.from_numpy(reads[r][0]).cuda(), torch.from_numpy(reads[r][1].astype(np.int64)).cuda()
        cm = cbl_amd.Comm.sim(r, world, store, 0.0)
        cm.set_protocol("bins")
        cm.set_recv_groups(groups)
        g = cbl_amd.CBL(k, pb)
        assert g.sharded_insert_seqs_device(cm, d_b, d_o, nr, cuts, bounds, True)
        if r == tgt:
            mine, used, fine = g.buckets(), cm.groups_used(), cm.groups_fine()
            assert g.validate() == 0
        g.close()
        cm.close()
    cbl_amd.Comm.sim_store_free(store)
    one = Oracle(k, pb, False)
    for c in range(slices):
        for r in range(world):
            hb, ho = reads[r]
            one.insert_seqs(hb[cuts[c] * L: cuts[c + 1] * L], ho[: cuts[c + 1] - cuts[c] + 1])
    full = cbl_amd.CBL(k, pb)
    full.load(one.serialize())
    want = [b for b in full.buckets() if b[0] >= int(bounds[-1])]
    full.close()
    assert used >= 1 and fine == used, (used, fine)  # every group of the last rank sorts 16 bits
    assert want and want[-1][0] == (1 << pb) - 1 and len(want) > 1
    assert len(mine) == len(want) and mine == want


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_rehearsal_of_any_rank_on_random_fine_plans(seed, monkeypatch):
    """FINE bins (PREFIX_BITS > 24) under random plans: world size, groups, slices, K, the REHEARSED RANK (CBLX_SIM_TARGET: any rank, not only the
    densest range) and the tail weight drawn at random; the rehearsed rank must end up with exactly the buckets of its prefix range of the
    one-process oracle's index (kinds, stored order), whichever of its groups sort 16 bits and whichever 24."""
    _need_gpu()
    rng = random.Random(9000 + seed)
    world = rng.choice([2, 3, 4, 5, 8])
    k = rng.choice([27, 31, 31, 33, 59])
    pb = rng.randint(25, 28)
    canonical = rng.random() < 0.3
    groups, slices, tgt = rng.choice([0, 2, 3, 5, 6]), rng.randint(1, 3), rng.randrange(world)
    monkeypatch.setenv("CBLX_SIM_TARGET", str(tgt))
    monkeypatch.setenv("CBLX_FINE_TAIL_WEIGHT", str(rng.choice([100, 120, 180])))
    monkeypatch.setenv("CBLX_WIRE_DIGITS", rng.choice(["0", "1"]))
    L, nr, store = (150 if k < 59 else 250), rng.choice([900, 2500, 6000]), 50000 + seed
    bounds = np.zeros(world - 1, dtype=np.uint32)
    valid = False
    cuts = [nr * s // slices for s in range(slices + 1)]
    for r in [x for x in range(world) if x != tgt] + [tgt]:
        d_b, d_o = synth.reads_torch(77 + seed, nr, L, first_read=r * nr, device="cuda")
        cm = cbl_amd.Comm.sim(r, world, store, 0.0)
        cm.set_protocol("bins")
        cm.set_recv_groups(groups)
        g = cbl_amd.CBL(k, pb, canonical=canonical)
        valid = g.sharded_insert_seqs_device(cm, d_b, d_o, nr, cuts, bounds, valid)
        if r == tgt:
            mine, used, fine = g.buckets(), cm.groups_used(), cm.groups_fine()
        g.close()
        cm.close()
    cbl_amd.Comm.sim_store_free(store)
    one = Oracle(k, pb, canonical)
    for c in range(slices):
        for r in range(world):
            if cuts[c + 1] > cuts[c]:
                hb, ho = synth.reads(77 + seed, cuts[c + 1] - cuts[c], L, first_read=r * nr + cuts[c])
                one.insert_seqs(hb, ho)
    full = cbl_amd.CBL(k, pb, canonical=canonical)
    full.load(one.serialize())
    lo, hi = (int(bounds[tgt - 1]) if tgt else 0), (int(bounds[tgt]) if tgt + 1 < world else 1 << pb)
    want = [b for b in full.buckets() if lo <= b[0] < hi]
    full.close()
    desc = f"world {world} k {k} pb {pb} canonical {canonical} groups {groups} slices {slices} rank {tgt}: {used} groups, {fine} fine"
    assert fine <= used, desc  # (used = 0: the plan was refused — coinciding bounds on a tiny job — and the ungrouped receiver ran)
    assert len(mine) == len(want) and mine == want, desc


@pytest.mark.parametrize("k,pb,n,canonical", [(31, 6, 300000, False), (21, 8, 600000, True), (15, 4, 200000, False), (31, 10, 1500000, False), (27, 9, 40000, False),
                                              (31, 16, 900000, False), (59, 12, 200000, True), (35, 14, 500000, False)])
def test_trie_union_by_merge_path_equals_the_sorting_route(k, pb, n, canonical, monkeypatch):
    """Trie |= Trie: two ascending lists are MERGED (k_bucket_union: merge-path rounds of 2048 outputs straight from the two arenas) instead
    of gathered and sorted again (CBLX_MERGE_UNION=0). Buckets of thousands to tens of thousands of words (several rounds per bucket),
    a shared stretch (words both sides hold: other's copy is dropped), one side much longer than the other, and self == other."""
    _need_gpu()
    rng = random.Random(1000 + k + pb)
    s1 = _rand_seq(rng, n)
    s2 = _rand_seq(rng, n // 5) + s1[n // 4: n // 4 + n // 3] + _rand_seq(rng, 100)
    o1, o2 = Oracle(k, pb, canonical), Oracle(k, pb, canonical)
    o1.insert_seq(s1), o2.insert_seq(s2)
    o1.merge(o2)
    want1, want2 = o1.serialize(), o2.serialize()
    o1.merge(o2)  # once more: the set stays, but a Vec that met other's bucket is sorted again as a whole (iter_sorted, src/trievec/mod.rs:209-220)
    want1b, want2b = o1.serialize(), o2.serialize()  # (other's Vecs that now meet a bucket cloned by the first merge get sorted too)
    def same(g, want, what):
        got = g.serialize()
        if got == want:
            return
        w = cbl_amd.CBL(k, pb, canonical=canonical)
        w.load(want)
        for (p1, k1, x1), (p2, k2, x2) in zip(g.buckets(), w.buckets()):
            assert (p1, k1, len(x1)) == (p2, k2, len(x2)), f"{what}: bucket header differs: gpu {(p1, k1, len(x1))} oracle {(p2, k2, len(x2))}"
            if x1 != x2:
                bad = [i for i in range(len(x1)) if x1[i] != x2[i]]
                raise AssertionError(f"{what}: bucket {p1:#x} kind {k1} len {len(x1)}: {len(bad)} positions differ, first {bad[:5]}: gpu {[hex(x1[i]) for i in bad[:3]]} oracle {[hex(x2[i]) for i in bad[:3]]}")
        raise AssertionError(f"{what}: bytes differ, buckets equal")

    # routes: Trie |= Trie merged or sorted again (CBLX_MERGE_UNION), the counting-sort classes read in place or gathered first (CBLX_MERGE_DIRECT)
    for route, direct in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("CBLX_MERGE_UNION", route)
        monkeypatch.setenv("CBLX_MERGE_DIRECT", direct)
        route = f"union={route} direct={direct}"
        g1, g2 = cbl_amd.CBL(k, pb, canonical=canonical), cbl_amd.CBL(k, pb, canonical=canonical)
        g1.insert_seq(s1), g2.insert_seq(s2)
        g1 |= g2
        same(g1, want1, f"route {route}, self after the first merge")
        same(g2, want2, f"route {route}, other after the first merge")
        assert g1.validate(strict=False) == 0
        g1 |= g2  # again: nothing new arrives, every word of other is a duplicate
        same(g1, want1b, f"route {route}, self after the second merge")
        same(g2, want2b, f"route {route}, other after the second merge")
        g1.close(), g2.close()
