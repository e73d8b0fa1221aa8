import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cbl_amd
from cbl_amd import synth
K, PB, NR = 31, 24, 10_000_000
d_b, d_o = synth.reads_torch(42, NR, 150, device="cuda:0")
g = cbl_amd.CBL(K, PB, device=0)
g.insert_seqs_device(d_b, d_o, NR)
blob = g.serialize_np()
g.close()
for th in (16, 24, 32):
    os.environ["CBLX_LOAD_THREADS"] = str(th)
    ts = []
    for rep in range(2):
        h = cbl_amd.CBL(K, PB, device=0)
        t0 = time.perf_counter(); h.load(blob); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        assert h.count() == 1_200_000_000
        h.close()
    print("threads", th, "load %.2f s = %.2f GB/s" % (min(ts), blob.size / min(ts) / 1e9), flush=True)
