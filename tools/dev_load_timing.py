"""dev: where does cblx_load spend its time (perf counters via py-spy are not available; use sizes that isolate parts)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cbl_amd
from cbl_amd import synth
for K, PB, NR in ((31, 24, 10_000_000), (31, 28, 10_000_000)):
    d_b, d_o = synth.reads_torch(42, NR, 150, device="cuda:0")
    g = cbl_amd.CBL(K, PB, device=0)
    g.insert_seqs_device(d_b, d_o, NR)
    blob = g.serialize_np()
    g.close()
    h = cbl_amd.CBL(K, PB, device=0)
    t0 = time.perf_counter()
    h.load(blob)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(K, PB, "bytes %.2f GB  load %.2f s  %.2f GB/s  words %d buckets %d" % (blob.size / 1e9, dt, blob.size / dt / 1e9, h.count(), h.num_buckets()))
    h.close()
