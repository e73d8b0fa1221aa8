import os, sys
sys.path.insert(0, os.getcwd())
os.environ["CBLX_FINE_MIN"] = "0"
os.environ["CBLX_TRACE_SHARDED"] = "1"
import numpy as np
import cbl_amd
from cbl_amd import synth
from oracle import Oracle
for k, pb, n in ((31, 28, 40), (31, 28, 400), (15, 25, 4000), (15, 25, 40000), (13, 25, 4000)):
    bases, offsets = synth.reads(42, n, 100)
    g = cbl_amd.CBL(k, pb)
    o = Oracle(k, pb, False)
    o.insert_seqs(bases, offsets)
    try:
        g.insert_seqs(bases, offsets); g.flush()
        print(k, pb, n, "fine", g.fine_builds(), "count", g.count(), o.count(), "buckets", g.num_buckets(), o.n_buckets(), "OK" if g.serialize() == o.serialize() else "DIFF", flush=True)
    except Exception as e:
        print(k, pb, n, "EXC", e, flush=True)
