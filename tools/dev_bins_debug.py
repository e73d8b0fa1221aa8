"""dev: the failing single-rank case of the bins protocol, traced (CBLX_TRACE_SHARDED=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cbl_amd
from cbl_amd import synth

uid = cbl_amd.Comm.unique_id()
comm = cbl_amd.Comm.rccl(uid, 0, 1, 0)
k, pb = int(sys.argv[1]), int(sys.argv[2])
nr = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
d_b, d_o = synth.reads_torch(42, nr, 150, device="cuda")
a, b = cbl_amd.CBL(k, pb), cbl_amd.CBL(k, pb)
a.insert_seqs_device(d_b, d_o, nr)
bounds = np.zeros(0, dtype=np.uint32)
cuts = [0, nr // 4, nr // 4, nr - 1, nr]
b.sharded_insert_seqs_device(comm, d_b, d_o, nr, cuts, bounds, False)
print("equal:", b.serialize() == a.serialize(), b.count(), a.count())
