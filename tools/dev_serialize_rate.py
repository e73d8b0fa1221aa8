"""Dev probe: the index bytes of cfg 2 (1.2 G k-mers, 9.4 GB) out of HBM — sizing pass, emitter + download into a FRESH host buffer (first touch
of every page) and into the same buffer again (pages resident), and into a torch pinned buffer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cbl_amd
from cbl_amd import synth

NR = int(os.environ.get("NR", 10_000_000))
d_b, d_o = synth.reads_torch(42, NR, 150, device="cuda")
g = cbl_amd.CBL(31, 24)
g.insert_seqs_device(d_b, d_o, NR)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = g.serialized_size(); t1 = time.perf_counter()
print("size %d bytes, sizing pass %.1f ms" % (n, (t1 - t0) * 1e3))
t0 = time.perf_counter(); blob = g.serialize_np(); t1 = time.perf_counter()
print("fresh numpy buffer: %.1f ms (%.1f GB/s)" % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e9))
for _ in range(2):
    t0 = time.perf_counter(); g.serialize_np(out=blob); t1 = time.perf_counter()
    print("same buffer again: %.1f ms (%.1f GB/s)" % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e9))
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
pn = pin.numpy()
for _ in range(2):
    t0 = time.perf_counter(); g.serialize_np(out=pn); t1 = time.perf_counter()
    print("pinned buffer: %.1f ms (%.1f GB/s)" % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e9))
for _ in range(2):
    path = "/dev/shm/cblx_dev_save.cbl"
    t0 = time.perf_counter(); g.save_to_file(path); t1 = time.perf_counter()
    print("save_to_file (tmpfs): %.1f ms (%.1f GB/s)" % ((t1 - t0) * 1e3, n / (t1 - t0) / 1e9))
    assert os.path.getsize(path) == n
    os.remove(path)
assert bytes(pn[:64]) == bytes(blob[:64]) and int(pn[:n:4097].astype(np.uint64).sum()) == int(blob[:n:4097].astype(np.uint64).sum())
