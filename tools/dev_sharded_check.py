"""Dev check (1 GPU): ShardedBuilder over a 1-rank RCCL group vs the direct insert, count + checksum, at a given size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import cbl_amd
from cbl_amd import sharded, synth
os.environ.pop("NCCL_DEBUG", None)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d_b, d_o = synth.reads_torch(42, n, 150, device="cuda")
a = cbl_amd.CBL(31, 24); a.insert_seqs_device(d_b, d_o, n)
print("direct  ", a.count(), hex(a.checksum()), flush=True)
for sl in [int(x) for x in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['1','2','4'])]:
    b = cbl_amd.CBL(31, 24)
    t = time.time(); sb = sharded.ShardedBuilder(b, dist, slices=sl); sb.insert_seqs_device(d_b, d_o, n); torch.cuda.synchronize()
    print("slices", sl, b.count(), hex(b.checksum()), "counts", sb.last_counts, round(time.time() - t, 3), "s", flush=True)
    b.close()
dist.destroy_process_group()
