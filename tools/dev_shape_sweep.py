"""Dev probe: throughput of the build on input shapes other than BASELINE's uniform 150-bp reads (same code path, K=31,
PREFIX_BITS=24, ~1 G k-mers each): long sequences, short reads, mixed lengths, reads with N, canonical."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cbl_amd
from cbl_amd import synth

K, PB = 31, 24
dev = "cuda"

def run(name, d_b, d_o, n, canonical=False):
    g = cbl_amd.CBL(K, PB, canonical=canonical, profile=True)
    best = 1e9
    for rep in range(3):
        g.clear()
        if rep == 1:
            g.stage_times_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.insert_seqs_device(d_b, d_o, n)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    lens = (d_o[1:] - d_o[:-1])
    nk = int((lens - (K - 1)).sum().item())
    st = {k_: round(v[0] / 2, 2) for k_, v in g.stage_times().items() if v[0] > 0.05}
    print("%-34s %7.1f ms  %6.2f G k-mers/s  (%d seqs, %d k-mer slots, %d distinct)  %s" % (name, best * 1e3, nk / best / 1e9, n, nk, g.count(), st), flush=True)
    assert g.validate() == 0

tot = 1_230_000_000
bases, _ = synth.reads_torch(42, 1, tot, device=dev)
def offs(lengths):
    o = torch.zeros(len(lengths) + 1, dtype=torch.int64, device=dev)
    o[1:] = torch.cumsum(lengths, 0)
    return o
# uniform 150 (reference point)
n = tot // 150
run("150 bp reads", bases, torch.arange(0, (n + 1) * 150, 150, device=dev, dtype=torch.int64), n)
run("150 bp reads, canonical", bases, torch.arange(0, (n + 1) * 150, 150, device=dev, dtype=torch.int64), n, canonical=True)
# 40 long sequences
n = 40
run("40 sequences of 30.75 Mbp", bases, torch.arange(0, (n + 1) * (tot // n), tot // n, device=dev, dtype=torch.int64), n)
# short reads
for L in (36, 50, 75):
    n = tot // 4 // L if L < 60 else tot // 2 // L
    run("%d bp reads" % L, bases, torch.arange(0, (n + 1) * L, L, device=dev, dtype=torch.int64), n)
# mixed lengths 50..300
torch.manual_seed(3)
lens = torch.randint(50, 301, (tot // 200,), device=dev, dtype=torch.int64)
lens = lens[torch.cumsum(lens, 0) <= tot]
run("mixed lengths 50..300", bases, offs(lens), len(lens))
# 1 % of the reads carry an N
b2 = bases.clone()
n = tot // 150
idx = torch.randint(0, n, (n // 100,), device=dev, dtype=torch.int64) * 150 + torch.randint(0, 150, (n // 100,), device=dev, dtype=torch.int64)
b2[idx] = ord("N")
run("150 bp reads, 1 % with an N", b2, torch.arange(0, (n + 1) * 150, 150, device=dev, dtype=torch.int64), n)
idx = torch.arange(0, n, device=dev, dtype=torch.int64) * 150 + torch.randint(0, 150, (n,), device=dev, dtype=torch.int64)
b2[idx] = ord("N")
run("150 bp reads, every read with an N", b2, torch.arange(0, (n + 1) * 150, 150, device=dev, dtype=torch.int64), n)
