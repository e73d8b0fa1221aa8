import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cbl_amd
from cbl_amd import synth
K, PB, NR = 31, 24, 10_000_000
d_b, d_o = synth.reads_torch(42, NR, 150, device="cuda:0")
g = cbl_amd.CBL(K, PB, device=0)
g.insert_seqs_device(d_b, d_o, NR)
path = "/dev/shm/cblx_lff.cbl"
g.save_to_file(path)
blob = np.fromfile(path, dtype=np.uint8)
g.close()
for rep in range(3):
    t0 = time.perf_counter(); h = cbl_amd.CBL.load_from_file(path, K, PB, device=0); torch.cuda.synchronize(); t1 = time.perf_counter()
    assert h.count() == 1_200_000_000
    h.close()
    h = cbl_amd.CBL(K, PB, device=0)
    t2 = time.perf_counter(); h.load(blob); torch.cuda.synchronize(); t3 = time.perf_counter()
    h.close()
    print("load_from_file %.2f s, load(buffer) %.2f s" % (t1 - t0, t3 - t2), flush=True)
os.remove(path)
