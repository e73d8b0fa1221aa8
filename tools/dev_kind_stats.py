"""dev: bucket kind / size distribution at the benchmark's mean bucket load (scaled down)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cbl_amd
from cbl_amd import synth
K, PB, NR, L = 31, 20, 625_000, 150
b, o = synth.reads_torch(42, NR, L, device="cuda:0")
g = cbl_amd.CBL(K, PB, device=0)
g.insert_seqs_device(b, o, NR)
sizes = {0: [], 1: []}
for p, kind, s in g.buckets():
    sizes[kind].append(len(s))
for k in (0, 1):
    a = np.array(sizes[k])
    print("kind", k, "buckets", len(a), "words", int(a.sum()), "max", int(a.max()) if len(a) else 0)
blob = g.serialize()
print("bytes/word", len(blob) / g.count())
a = np.array(sizes[0] + sizes[1]); print("pct words in buckets > 256:", a[a > 256].sum() / a.sum(), " >1024:", a[a > 1024].sum() / a.sum(), "mean", a.mean())
