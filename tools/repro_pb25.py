#!/usr/bin/env python3
"""Repro hunt: incremental inserts at PREFIX_BITS 17..28 (fused directory on a short last pass) vs the oracle."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cbl_amd
from oracle import Oracle

def rand_seq(rng, n, alphabet): return bytes(rng.choice(alphabet) for _ in range(n))
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for trial in range(400):
    k = rng.choice([13, 15, 21, 25, 31])
    wb = 2 * k + (2 * k - 1).bit_length()
    pb = rng.choice([17, 18, 23, 24, 25, 26, 27, 28])
    if pb >= wb: continue
    alphabet = rng.choice([b"ACGT", b"ACGTacgtN"])
    g, o = cbl_amd.CBL(k, pb), Oracle(k, pb)
    try:
        for step in range(rng.randint(2, 4)):
            seqs = [rand_seq(rng, rng.randint(k, k + rng.choice([0, 9, 400, 2600])), alphabet) for _ in range(rng.randint(1, 25))]
            bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
            offsets = np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64)
            g.insert_seqs(bases, offsets); o.insert_seqs(bases, offsets)
            if g.count() != o.count() or g.serialize() != o.serialize():
                raise AssertionError("mismatch")
    except Exception as e:
        bad += 1
        print("FAIL trial", trial, "k", k, "pb", pb, alphabet, "step", step, "nseq", len(seqs), "bases", len(bases), type(e).__name__, str(e)[:100], flush=True)
print("done, failures:", bad)
