#!/usr/bin/env python3
"""Generates tests/golden/index_vectors.json: inputs (synthetic read parameters or literal sequences) and expected
outputs (k-mer count, bucket count, index size, SHA-256 of the index bytes, and the full bytes for tiny cases).

The reference cannot be run in this container (no Rust toolchain), so these vectors are produced by the
C++ oracle (oracle/cbl_oracle.hpp) and, where small enough, REQUIRED to equal the independent Python restatement
(oracle/pyref.py) before being written. They pin the oracle against regressions and give the GPU tests a target that does
not depend on running the oracle. Run from the repo root: python tests/golden/make_golden.py"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cbl_amd import synth  # noqa: E402
from oracle import Oracle  # noqa: E402
from oracle.pyref import PyCBL  # noqa: E402

SYNTH = [  # (name, k, pb, canonical, seed, n_reads, read_len, check_with_pyref)
    ("cfg1_shape", 25, 24, False, 42, 2000, 150, True),
    ("cfg2_shape", 31, 24, False, 42, 2000, 150, True),
    ("cfg2_shape_canonical", 31, 24, True, 42, 1500, 150, True),
    ("cfg3_shape", 31, 28, False, 42, 1500, 150, True),
    ("cfg4_shape", 59, 28, False, 42, 1000, 250, True),
    ("cfg1_full_10k_reads", 25, 24, False, 42, 10000, 150, False),   # BASELINE.json configs[0] at full size
    ("tries_k11_pb8", 11, 8, False, 7, 600, 100, True),
]
LITERAL = [  # (name, k, pb, canonical, [sequences])
    ("appendix_b2", 7, 14, False, ["ACGTACGTAC"]),
    ("appendix_b3", 7, 14, False, ["GATTACAGATTACATTTGGGACCA"]),
    ("non_acgt_and_case", 7, 14, False, ["ACGTNNACGTTGCAacgtnACGTAGGCTA", "NNNNNNNACGTACG"]),
    ("canonical_small", 9, 10, True, ["ACGTTGCATGCATGCAAGCTTAGCTAGGATCC", "TTTTTTTTTTTTAAAAAAAAAAAA"]),
]


KMER_CASES = [  # (name, k, pb, canonical, seed, seed sequence length, n single inserts, query length)
    ("kmers_k7_pb14", 7, 14, False, 3, 60, 120, 40),
    ("kmers_k9_pb10_canonical", 9, 10, True, 4, 80, 150, 50),
    ("kmers_k31_pb24", 31, 24, False, 5, 200, 60, 120),
    ("kmers_k35_pb8", 35, 8, True, 6, 150, 60, 100),
]


def kmer_case(name, k, pb, canon, seed, seed_len, n_ins, q_len):
    """Single-k-mer surface (src/cbl.rs:219-228,311-324,358-361): a sequence, then n single inserts drawn from a small
    pool (repeats, k-mers of the sequence, fresh ones), then a query sequence and the iteration order. Expected values
    from the C++ oracle, REQUIRED to equal the Python restatement."""
    import random

    from oracle import pyref

    rng = random.Random(seed)
    seq = "".join(rng.choice("ACGT") for _ in range(seed_len))
    o, p = Oracle(k, pb, canon), PyCBL(k, pb, canon)
    o.insert_seq(seq.encode())
    p.insert_seq(seq.encode())
    mask = (1 << (2 * k)) - 1
    in_seq = []
    x = 0
    for i, ch in enumerate(seq):
        x = ((x << 2) | "ACTG".index(ch)) & mask
        if i >= k - 1:
            in_seq.append(x)
    pool = [rng.getrandbits(2 * k) for _ in range(12)] + rng.sample(in_seq, 6) + [0, mask]
    kmers = [rng.choice(pool) for _ in range(n_ins)]
    P = p.P

    def py_word(km):
        if canon and bin(km).count("1") & 1:
            km = pyref.rev_comp(km, k)
        nk, pos = pyref.necklace_pos(km, P["KB"])
        return (nk << P["POS"]) | pos

    def py_contains(w):
        b = p.buckets.get(w >> P["SB"])
        return b is not None and (w & ((1 << P["SB"]) - 1)) in b[1]

    absent = []
    for km in kmers:
        got = o.insert_kmer(km)
        w = py_word(km)
        exp = not py_contains(w)
        pfx = p._insert_word(w)
        b = p.buckets[pfx]
        if len(b[1]) > 1024 and b[0] == "vec":
            b[0] = "trie"
        if b[0] == "trie":
            b[1].sort()
        assert got == exp, (name, km)
        absent.append(bool(got))
    blob = o.serialize()
    assert p.serialize() == blob, name
    query = "".join(rng.choice("ACGT") for _ in range(q_len // 2)) + seq[5 : 5 + q_len // 2 + k]
    flags = [bool(o.contains_word(w)) for w in o.seq_words(query.encode())]
    assert flags == [py_contains(w) for w in pyref.seq_words(query.encode(), P, canon)], name
    it = [o.kmer_of_word(w) for w in o.iter_words()]
    py_it = []
    for pfx in sorted(p.buckets):
        for sfx in p.buckets[pfx][1]:
            w = (pfx << P["SB"]) | sfx
            nk, pos = w >> P["POS"], w & ((1 << P["POS"]) - 1)
            py_it.append(((nk << (P["KB"] - pos)) & mask) | (nk >> pos))  # src/necklace/mod.rs:29-31
    assert it == py_it, name
    return dict(name=name, k=k, prefix_bits=pb, canonical=canon, sequence=seq, kmers=[hex(v) for v in kmers], was_absent=absent,
                count=o.count(), index_hex=blob.hex(), query=query, query_flags=flags, iter=[hex(v) for v in it])


def main():
    out = {"synthetic": [], "literal": [], "kmers": [kmer_case(*c) for c in KMER_CASES]}
    for name, k, pb, canon, seed, n, L, chk in SYNTH:
        bases, offsets = synth.reads(seed, n, L)
        o = Oracle(k, pb, canon)
        o.insert_seqs(bases, offsets)
        blob = o.serialize()
        if chk:
            p = PyCBL(k, pb, canon)
            raw = bases.tobytes()
            for i in range(n):
                p.insert_seq(raw[i * L : (i + 1) * L])
            assert p.serialize() == blob, name
        out["synthetic"].append(dict(name=name, k=k, prefix_bits=pb, canonical=canon, seed=seed, n_reads=n, read_len=L,
                                     count=o.count(), n_buckets=o.n_buckets(), index_bytes=len(blob),
                                     sha256=hashlib.sha256(blob).hexdigest(),
                                     first_read=raw[:L].decode() if chk else bases[:L].tobytes().decode()))
    for name, k, pb, canon, seqs in LITERAL:
        o, p = Oracle(k, pb, canon), PyCBL(k, pb, canon)
        for s in seqs:
            o.insert_seq(s.encode())
            p.insert_seq(s.encode())
        blob = o.serialize()
        assert p.serialize() == blob, name
        out["literal"].append(dict(name=name, k=k, prefix_bits=pb, canonical=canon, sequences=seqs, count=o.count(),
                                   index_hex=blob.hex()))
    with open(os.path.join(ROOT, "tests", "golden", "index_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["synthetic"]), "+", len(out["literal"]), "+", len(out["kmers"]), "vectors")


if __name__ == "__main__":
    main()
