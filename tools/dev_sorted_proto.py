"""dev: wall time of the three calls of the sorted-batch protocol on one GPU (cfg 2 shape, one slice)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cbl_amd
from cbl_amd import synth
K, PB, NR, L = 31, 24, int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 150
d_b, d_o = synth.reads_torch(42, NR, L, device="cuda:0")
torch.cuda.synchronize()
send, recv = cbl_amd.CBL(K, PB, device=0, profile=True), cbl_amd.CBL(K, PB, device=0, profile=True)
B = send.consts()["bytes"]
nd = 8
bounds = np.array([(1 << PB) * (i + 1) // 512 for i in range(nd - 1)], dtype=np.uint32)
for it in range(3):
    recv.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bs, ws = send.sorted_batch_begin(d_b, d_o, NR, bounds, nd)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    prefix = torch.empty(bs[nd], dtype=torch.int32, device="cuda:0"); count = torch.empty(bs[nd], dtype=torch.int32, device="cuda:0")
    suffix = torch.empty(ws[nd] * B, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize(); t2 = time.perf_counter()
    send.sorted_batch_export(prefix, count, suffix)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    batches = [(bs[d + 1] - bs[d], ws[d + 1] - ws[d], prefix[bs[d]:bs[d + 1]], count[bs[d]:bs[d + 1]], suffix[ws[d] * B: ws[d + 1] * B]) for d in range(nd)]
    recv.stage_times_reset()
    recv.insert_sorted_batches_device(batches)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print("begin %.1f ms  export %.1f ms  insert %.1f ms  (words %d, buckets %d)" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, ws[nd], bs[nd]),
          {k: round(v[0], 2) for k, v in recv.stage_times().items() if v[1]})
