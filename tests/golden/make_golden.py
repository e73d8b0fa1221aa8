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


def main():
    out = {"synthetic": [], "literal": []}
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
    print("wrote", len(out["synthetic"]), "+", len(out["literal"]), "vectors")


if __name__ == "__main__":
    main()
