#!/usr/bin/env python3
"""Which kinds meet in `A |= B` at cfg 5's per-GPU share: words of the both-sided buckets by (self kind, other kind), and by merged run length.
Usage: tools/dev_merge_pairs.py [--reads 6250000]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cbl_amd
from cbl_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=6_250_000)
a = ap.parse_args()
tabs = []
for seed in (42, 43):
    b, o = synth.reads_torch(seed, a.reads, 150, device="cuda")
    g = cbl_amd.CBL(31, 24)
    g.insert_seqs_device(b, o, a.reads)
    tabs.append(g.bucket_table_np())
    g.close()
(p1, l1, k1), (p2, l2, k2) = tabs
both, i1, i2 = np.intersect1d(p1, p2, return_indices=True)
ls, lo, ks, ko = l1[i1].astype(np.int64), l2[i2].astype(np.int64), k1[i1], k2[i2]
tot = int(l1.sum() + l2.sum())
out = {"words_total": tot, "both_sided_words": int((ls + lo).sum()), "pairs": {}}
for a_, an in ((0, "Vec"), (1, "Trie")):
    for b_, bn in ((0, "Vec"), (1, "Trie")):
        m = (ks == a_) & (ko == b_)
        out["pairs"][f"{an}|={bn}"] = {"buckets": int(m.sum()), "words": int((ls + lo)[m].sum()), "share_of_all_words": round(float((ls + lo)[m].sum()) / tot, 4)}
c = ls + lo
for name, lo_, hi_ in (("<=512", 0, 512), ("<=1024", 512, 1024), ("<=2048", 1024, 2048), ("<=4096", 2048, 4096), (">4096", 4096, 1 << 40)):
    m = (c > lo_) & (c <= hi_) & ~((ks == 1) & (ko == 1))
    out.setdefault("sorted_route_by_run_length", {})[name] = {"buckets": int(m.sum()), "words": int(c[m].sum())}
print(json.dumps(out))
