"""Dev probe: what k_bucket_msd's sub-bucket function does to the buckets of cfg 2 that need a sorted result (> 1024 words).
For a sample of such buckets: the expected number of sub-bucket mates an element ranks itself against, and the largest
sub-bucket, under (a) the top ceil(log2 c) suffix bits (shipping, order-preserving), (b) an ideal range normalisation
(s - min) * NB / (max - min + 1) (order-preserving), (c) a hash of the whole suffix (not order-preserving)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cbl_amd
from cbl_amd import synth

k, pb, n, L = 31, 24, 10_000_000, 150
d_b, d_o = synth.reads_torch(42, n, L, device="cuda")
g = cbl_amd.CBL(k, pb)
g.insert_seqs_device(d_b, d_o, n)
del d_b
P = g.consts()
SB, BYTES = P["suffix_bits"], P["bytes"]
(bs, ws) = g.resident_split(np.zeros(0, dtype=np.uint32), 1)
nb, nw = bs[1], ws[1]
d_p = torch.empty(nb, dtype=torch.int32, device="cuda")
d_c = torch.empty(nb, dtype=torch.int32, device="cuda")
d_k = torch.empty(nb, dtype=torch.uint8, device="cuda")
d_s = torch.empty(nw * BYTES + 16, dtype=torch.uint8, device="cuda")
g.resident_export(d_p, d_c, d_k, d_s)
cnt = d_c.cpu().numpy().astype(np.int64)
start = np.concatenate([[0], np.cumsum(cnt)])
big = np.nonzero(cnt > 1024)[0]
rng = np.random.default_rng(1)
pick = rng.choice(big, size=min(1500, len(big)), replace=False)
res = {"top": [], "range": [], "hash": []}
mx = {"top": 0, "range": 0, "hash": 0}
for b in pick:
    c = int(cnt[b])
    raw = d_s[start[b] * BYTES:(start[b] + c) * BYTES].cpu().numpy().reshape(c, BYTES)
    s = np.zeros(c, dtype=np.uint64)
    for i in range(BYTES):
        s |= raw[:, i].astype(np.uint64) << np.uint64(8 * i)
    nbits = int(np.ceil(np.log2(c)))
    NB = 1 << nbits
    subs = {
        "top": (s >> np.uint64(SB - nbits)).astype(np.int64),
        "range": ((s - s.min()).astype(np.float64) * NB / float(s.max() - s.min() + 1)).astype(np.int64),
        "hash": (((s ^ (s >> np.uint64(32))) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(64 - nbits)).astype(np.int64),
    }
    for name, sub in subs.items():
        occ = np.bincount(sub, minlength=NB)
        res[name].append(float((occ.astype(np.float64) ** 2).sum() / c - 1.0))  # mates per element
        mx[name] = max(mx[name], int(occ.max()))
print("buckets > 1024 words:", len(big), "of", nb, "; sampled", len(pick), "; words in them %.1f %%" % (100.0 * cnt[big].sum() / cnt.sum()))
for name in ("top", "range", "hash"):
    a = np.array(res[name])
    print("%-6s mates per element: mean %.2f  p50 %.2f  p99 %.2f   largest sub-bucket %d" % (name, a.mean(), np.median(a), np.percentile(a, 99), mx[name]))
