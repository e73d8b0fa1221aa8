"""Dev probe: the duplicate-heavy workload of SURVEY.md §8d — reads sampled at 30x coverage from a 40 Mbp random genome
(8 M x 150 bp, K=31, PREFIX_BITS=24): every k-mer arrives ~24 times. Times the build, lists the stage times, and checks
size-independent properties against an index built from the genome itself."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cbl_amd
from cbl_amd import synth

K, PB, L = int(os.environ.get("DUP_K", 31)), int(os.environ.get("DUP_PB", 24)), int(os.environ.get("DUP_L", 150))
COV = int(sys.argv[1]) if len(sys.argv) > 1 else 30  # coverage; the read count stays 8 M
n = 8_000_000
G = n * L // COV
gen, _ = synth.reads_torch(4242, 1, G, device="cuda")  # the genome: one 40 Mbp sequence
torch.manual_seed(7)
pos = torch.randint(0, G - L, (n,), device="cuda", dtype=torch.int64)
chunks = []
for a in range(0, n, 1_000_000):
    p = pos[a:a + 1_000_000]
    chunks.append(gen[(p[:, None] + torch.arange(L, device="cuda")[None, :]).reshape(-1)])
d_b = torch.cat(chunks)
del chunks
d_o = torch.arange(0, (n + 1) * L, L, device="cuda", dtype=torch.int64)
g = cbl_amd.CBL(K, PB, profile=True)
best = 1e9
for rep in range(4):
    g.clear()
    if rep == 1:
        g.stage_times_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.insert_seqs_device(d_b, d_o, n)
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
st = g.stage_times()
nk = n * (L - K + 1)
print("reads", n, "k-mer instances", nk, "distinct", g.count(), "-> %.1f ms, %.2f G k-mers/s" % (best * 1e3, nk / best / 1e9))
print({k_: round(v[0] / 3, 3) for k_, v in st.items() if v[0] > 0})
print("bucket stats", g.bucket_stats() if hasattr(g, "bucket_stats") else "")
# properties
assert g.validate() == 0
cnt, cs = g.count(), g.checksum()
g.insert_seqs_device(d_b, d_o, n)  # everything again: nothing new
assert (g.count(), g.checksum()) == (cnt, cs)
ref = cbl_amd.CBL(K, PB)
ref.insert_seqs_device(gen, torch.tensor([0, G], device="cuda", dtype=torch.int64), 1)
assert cnt <= ref.count()
tot, hit = ref.contains_seqs_device(d_b, d_o, n)
assert tot == hit == nk, (tot, hit, nk)
g |= ref
assert (g.count(), g.checksum()) == (ref.count(), ref.checksum())
print("properties ok; genome k-mers", ref.count(), "covered by the reads %.4f" % (cnt / ref.count()))
