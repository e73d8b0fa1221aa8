"""A model of `k_bucket_union`'s rounds (cbl_amd/csrc/kernels_bucket.hpp; Trie |= Trie, /root/reference/src/trievec/set_ops.rs:43-71 merges two
ascending iterators with two pointers and drops other's copy of a word self holds): the index arithmetic of the kernel restated thread by thread in
Python — the two staging RINGS of T slots (word g of a list in slot g mod T, only the slots a round consumed are refilled, the round's outputs pass
through exactly those slots), the co-rank of every thread's last output by binary search on the round's diagonal (ties take self's copy first), the
n + n candidates with all-ones sentinels, min(a[k], b[n-1-k]) + a bitonic network, the comparison with the predecessor (across rounds: the carry)
and the ordered compaction. Checked against sorted(set(A) | set(B)). Runs without a GPU: it pins the ALGORITHM the kernel implements (the kernel
itself is compared with the oracle and with the sorting route in tests/test_gpu_parity.py::test_trie_union_by_merge_path_equals_the_sorting_route)."""
import random

import pytest

INF = (1 << 64) - 1


def pad(i):
    return i + (i >> 3)  # uni_pad: one padding word per 8


def union_rounds(A, B, threads, items):
    T = threads * items
    assert T & (T - 1) == 0
    lds = [None] * (pad(2 * T) + 8)
    ra = lambda g: pad(g & (T - 1))
    rb = lambda g: pad(T + (g & (T - 1)))
    cs, co = len(A), len(B)
    ia = ib = ha = hb = 0
    out, carry, have_carry, loads = [], None, False, 0
    while ia < cs or ib < co:
        na, nb = min(cs - ia, T), min(co - ib, T)
        nout = min(na + nb, T)
        for g in range(ia + ha, ia + na):  # only what the last round consumed is staged again
            lds[ra(g)] = A[g]
            loads += 1
        for g in range(ib + hb, ib + nb):
            lds[rb(g)] = B[g]
            loads += 1
        # co-rank of the end of every thread's outputs
        i1 = []
        for tid in range(threads):
            d1 = min((tid + 1) * items, nout)
            lo, hi = max(d1 - nb, 0), min(d1, na)
            while lo < hi:
                mid = (lo + hi) >> 1
                if lds[ra(ia + mid)] <= lds[rb(ib + d1 - 1 - mid)]:
                    lo = mid + 1
                else:
                    hi = mid
            i1.append(lo)
        iend = i1[-1]
        outs = [None] * T
        for tid in range(threads):
            d0 = min(tid * items, nout)
            i0 = i1[tid - 1] if tid else 0
            j0 = d0 - i0
            a = [lds[ra(ia + i0 + k)] if i0 + k < na else INF for k in range(items)]
            b = [lds[rb(ib + j0 + k)] if j0 + k < nb else INF for k in range(items)]
            o = [min(a[k], b[items - 1 - k]) for k in range(items)]  # the n smallest, a bitonic sequence
            st = items // 2
            while st >= 1:
                for k in range(items):
                    if (k & st) == 0 and o[k] > o[k + st]:
                        o[k], o[k + st] = o[k + st], o[k]
                st >>= 1
            for k in range(items):
                outs[tid * items + k] = o[k]
        # the outputs take the place of what the round consumed: A's ring first (iend slots), then B's
        oslot = lambda q: ra(ia + q) if q < iend else rb(ib + (q - iend))
        for q in range(nout):
            lds[oslot(q)] = outs[q]
        for p in range(nout):
            v = lds[oslot(p)]
            head = (v != lds[oslot(p - 1)]) if p else (not have_carry or v != carry)
            if head:
                out.append(v)
        carry, have_carry = lds[oslot(nout - 1)], True
        ha, hb = na - iend, nb - (nout - iend)
        ia += iend
        ib += nout - iend
    return out, loads


def _lists(rng, na, nb, shared, bits=40):
    pool = rng.sample(range(1 << bits), na + nb)
    a, b = pool[:na], pool[na:]
    b = b[:max(0, nb - shared)] + rng.sample(a, min(shared, na)) if na else b
    return sorted(set(a)), sorted(set(b))


@pytest.mark.parametrize("threads,items", [(8, 4), (2, 4), (16, 2), (128, 4)])
def test_rounds_on_rings_give_the_sorted_union(threads, items):
    rng = random.Random(threads * 100 + items)
    T = threads * items
    shapes = [(0, 5), (5, 0), (1, 1), (T, T), (T - 1, T + 1), (3 * T + 7, 2 * T - 3), (10 * T, 13), (13, 10 * T), (5 * T, 5 * T)]
    for na, nb in shapes:
        for shared in (0, min(na, nb) // 3, min(na, nb)):
            A, B = _lists(rng, na, nb, shared)
            got, loads = union_rounds(A, B, threads, items)
            assert got == sorted(set(A) | set(B)), (threads, items, na, nb, shared)
            assert loads == len(A) + len(B), "every word is staged exactly once"


def test_self_union_and_ties_take_selfs_copy_first():
    rng = random.Random(7)
    A, _ = _lists(rng, 1000, 0, 0)
    got, loads = union_rounds(A, list(A), 8, 4)
    assert got == A and loads == 2 * len(A)


def test_values_at_the_sentinel_boundary():
    # all-ones is the sentinel of the candidate fetch; real suffixes are < 2^SUFFIX_BITS <= 2^64 - 1 only when SUFFIX_BITS < 64 — the kernel masks to
    # SUFFIX_BITS, and a 64-bit suffix of all ones would tie with the sentinel: the model shows the merge still emits it once (it is the maximum)
    A, B = [1, 5, INF - 1], [2, 5, INF - 1]
    got, _ = union_rounds(A, B, 2, 4)
    assert got == [1, 2, 5, INF - 1]
