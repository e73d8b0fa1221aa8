"""A CPU model of k_bucket_sorted's ranking (cbl_amd/csrc/kernels_bucket.hpp) against a plain sort.

Test infrastructure, not the product: the kernel itself is checked against the oracle by the GPU parity tests. What this file pins is the ARGUMENT
the kernel rests on — after a counting sort on the top suffix bits, a lane that owns `per` consecutive slots of the sub-bucket order finds the final
rank of each of them as  (start of its first slot's sub-bucket, rounded down to an even slot)  +  (entries of ONE walk over the span up to the end of
its last slot's sub-bucket that compare less)  — for every arrival order inside the sub-buckets, ragged lengths, clusters (dozens of entries in one
sub-bucket), spans that cross many sub-buckets, repeats (ties broken by the stream index), the all-ones suffix next to the all-ones slack, and
sub-ranges of long runs (`skip` leading bits shared by the whole run). The result it must give is what TrieVec::insert leaves in a Trie bucket
(/root/reference/src/trievec/mod.rs:100-115): the ascending list of distinct suffixes.
(Not modelled: the kernel hands a run with a sub-bucket of more than MSD_LIMIT entries to the claim-table kernel first — a cost decision, the
ranking argument holds for any sub-bucket size, which is what the repeat cases below exercise.)
"""
import random

import pytest

PK_BITS = 12
ONES = (1 << 128) - 1  # the slack behind the run compares greater than every element


def walk_sort(suffixes, SB, threads, skip=0, rng=None):
    """(sorted distinct suffixes, steps of the longest walk) by the kernel's phases; asserts that the ranks are a permutation."""
    c = len(suffixes)
    assert 0 < c <= threads * 8 and c <= (1 << PK_BITS)
    nbits = max(c - 1, 1).bit_length()  # ceil(log2 c), 1 for c <= 2
    nbits = min(nbits, SB - skip)
    NB, sub_sh = 1 << nbits, SB - skip - nbits
    sub_of = lambda s: (s >> sub_sh) & (NB - 1)  # noqa: E731
    # counting sort into sub-bucket order; the arrival order inside a sub-bucket is whatever the LDS atomics made it
    order = list(range(c))
    if rng:
        rng.shuffle(order)
    cnt = [0] * (NB + 1)
    for e in order:
        cnt[sub_of(suffixes[e])] += 1
    off = [0] * (NB + 1)
    for b in range(NB):
        off[b + 1] = off[b] + cnt[b]
    assert off[NB] == c
    fill = list(off)
    slots = [None] * c
    for e in order:
        b = sub_of(suffixes[e])
        slots[fill[b]] = (suffixes[e] << PK_BITS) | e
        fill[b] += 1
    at = lambda q: slots[q] if q < c else ONES  # noqa: E731
    per = (c + threads - 1) // threads
    out = [None] * c
    longest = 0
    for t in range(threads):
        p0 = t * per
        if p0 >= c:
            continue
        mine = slots[p0: min(p0 + per, c)]
        A = off[sub_of(mine[0] >> PK_BITS)] & ~1
        B = off[sub_of(mine[-1] >> PK_BITS) + 1]
        fin = [0] * len(mine)
        q = A
        while q < B:  # two slots per step; what is read behind B belongs to a later sub-bucket or is the slack: greater than every entry of mine
            for o in (at(q), at(q + 1)):
                for i, me in enumerate(mine):
                    fin[i] += o < me
            q += 2
        longest = max(longest, (B - A + 1) // 2)
        for i, me in enumerate(mine):
            assert out[A + fin[i]] is None, "two elements with one rank"
            out[A + fin[i]] = me
    assert all(x is not None for x in out)
    assert out == sorted(out)
    # heads: the first slot of every suffix value (settled after the sort by looking at the slot in front)
    heads = [x >> PK_BITS for p, x in enumerate(out) if p == 0 or (out[p - 1] >> PK_BITS) != (x >> PK_BITS)]
    return heads, longest


@pytest.mark.parametrize("threads,SB", [(256, 44), (512, 44), (64, 40), (256, 97), (128, 20)])
def test_uniform_and_ragged_runs(threads, SB):
    rng = random.Random(threads * 1000 + SB)
    for c in (1, 2, 3, 63, 64, 65, threads * 4 + 1, threads * 5, threads * 8 - 1, threads * 8):
        if c > (1 << PK_BITS):
            continue
        sfx = [rng.getrandbits(SB) for _ in range(c)]
        heads, _ = walk_sort(sfx, SB, threads, rng=rng)
        assert heads == sorted(set(sfx))


def test_necklace_clusters_and_spans_over_many_sub_buckets():
    """Entries that share their leading bits (consecutive k-mers of a read) crowd ONE sub-bucket while its neighbours stay empty: a lane's eight
    slots then lie in one sub-bucket of dozens, or straddle sub-buckets far apart."""
    rng = random.Random(7)
    SB, threads = 44, 256
    for trial in range(20):
        sfx = []
        while len(sfx) < 1500:
            top = rng.getrandbits(20) << (SB - 20)
            for _ in range(rng.choice((1, 1, 2, 5, 30, 45))):
                sfx.append(top | rng.getrandbits(SB - 20 if trial % 2 else 10))
        sfx = sfx[: 1024 + rng.randrange(1024)]
        heads, longest = walk_sort(sfx, SB, threads, rng=rng)
        assert heads == sorted(set(sfx)) and longest >= 1


def test_repeats_keep_one_copy_and_the_smallest_index_comes_first():
    rng = random.Random(11)
    SB, threads = 40, 256
    base = [rng.getrandbits(SB) for _ in range(300)]
    sfx = [rng.choice(base) for _ in range(2000)]
    heads, _ = walk_sort(sfx, SB, threads, rng=rng)
    assert heads == sorted(set(base) & set(sfx))


def test_all_ones_suffix_next_to_the_slack_and_all_zero():
    SB, threads = 44, 256
    ones = (1 << SB) - 1
    sfx = [ones, 0, ones - 1, 1, ones, 0] + [ones - i for i in range(2, 1100)]
    heads, _ = walk_sort(sfx, SB, threads, rng=random.Random(3))
    assert heads == sorted(set(sfx)) and heads[0] == 0 and heads[-1] == ones


@pytest.mark.parametrize("skip", [1, 5, 8])
def test_sub_range_of_a_long_run_shares_its_leading_bits(skip):
    rng = random.Random(skip)
    SB, threads = 44, 256
    lead = rng.getrandbits(skip) << (SB - skip)
    sfx = [lead | rng.getrandbits(SB - skip) for _ in range(1777)]
    heads, _ = walk_sort(sfx, SB, threads, skip=skip, rng=rng)
    assert heads == sorted(set(sfx))


def test_fewer_suffix_bits_than_sub_bucket_bits():
    """nbits is capped by the suffix width (tiny K): every value has a sub-bucket of its own and the copies of a value fill it."""
    rng = random.Random(5)
    SB, threads = 9, 256
    sfx = [rng.getrandbits(SB) for _ in range(1500)]
    heads, _ = walk_sort(sfx, SB, threads, rng=rng)
    assert heads == sorted(set(sfx))
