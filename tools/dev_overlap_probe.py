"""Dev: do two independent builds on one GPU overlap (HBM-bound scatter of one under the VALU/LDS-bound bucket kernels of
the other)? Two contexts, two host threads, half-size batches each; sequential vs concurrent wall time."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cbl_amd
from cbl_amd import synth

NR, L, K, PB = 5_000_000, 150, 31, 24
a_b, a_o = synth.reads_torch(42, NR, L, device="cuda:0")
b_b, b_o = synth.reads_torch(43, NR, L, device="cuda:0")
ga, gb = cbl_amd.CBL(K, PB, device=0), cbl_amd.CBL(K, PB, device=0)

def build(g, b, o):
    g.clear()
    g.insert_seqs_device(b, o, NR)

for _ in range(2):
    build(ga, a_b, a_o); build(gb, b_b, b_o)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); build(ga, a_b, a_o); build(gb, b_b, b_o); torch.cuda.synchronize(); ts = time.perf_counter() - t0
    t0 = time.perf_counter()
    th = [threading.Thread(target=build, args=(ga, a_b, a_o)), threading.Thread(target=build, args=(gb, b_b, b_o))]
    [t.start() for t in th]; [t.join() for t in th]; torch.cuda.synchronize(); tp = time.perf_counter() - t0
    print(f"sequential {ts * 1e3:.1f} ms   concurrent {tp * 1e3:.1f} ms   ratio {tp / ts:.3f}")
