"""Dev check: torch.distributed.all_to_all_single (RCCL) on one rank: does a large self-exchange copy every element?"""
import os, sys
import torch, torch.distributed as dist
os.environ.pop("NCCL_DEBUG", None)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29542")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
for n in (1 << 26, 1 << 28, 300_000_000, 1 << 29, 600_000_000, 1_200_000_000):
    for dt in (torch.int64, torch.uint8):
        src = torch.arange(n, dtype=torch.int64, device="cuda").to(dt)
        for mode in ("sync", "async"):
            dst = torch.full((n,), 7, dtype=dt, device="cuda")
            w = dist.all_to_all_single(dst, src, [n], [n], async_op=(mode == "async"))
            if w is not None: w.wait()
            torch.cuda.synchronize()
            bad = int((dst != src).sum().item())
            print(f"n={n} bytes={n*src.element_size()/1e9:.2f}GB {dt} {mode}: mismatches={bad}", flush=True)
        del src, dst
dist.destroy_process_group()
