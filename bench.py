#!/usr/bin/env python3
"""bench.py — k-mers inserted/sec (build index), K=31 150 bp synthetic reads (BASELINE.json), on N MI355X.

A "step" = one pass of the hot path (insert_seq -> necklace transform -> prefix partition -> bucket insert) over one
batch of synthetic reads already resident in HBM: BASELINE.json configs[1] (K=31, 68-bit word, PREFIX_BITS=24,
10M x 150 bp per GPU). N > 1: one process per GPU (torchrun), reads sharded contiguously by rank, prefix space
sharded by quantile ranges, one RCCL all-to-all of the transformed words (cbl_amd/sharded.py); per-GPU work is fixed
as N grows ("weak").

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     - the dominant kernel group of the step, algorithmic bytes / HIP-event time vs the 8 TB/s HBM peak
  cpu_baseline - the CPU oracle (a C++ port of the reference algorithm, oracle/) timed on a bounded sample, 1 thread
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# ALGORITHMIC bytes per k-mer and step of each kernel group: what the group's contract must move once.
#   R_in = record bytes out of KRN-1 (8 lo + hi part); after the first partition pass a 65..72-bit word keeps only lo.
def stage_alg_bytes(k: int, pb: int, read_len: int):
    kb = 2 * k
    wb = kb + (kb - 1).bit_length()
    hi = 0 if wb <= 64 else (1 if kb <= 64 else 8)
    r_in = 8 + hi
    r_out = 8 if hi == 1 else r_in
    n_a = min(8, pb)
    lsd = (pb - n_a + 7) // 8
    sfx = 8 if wb - pb <= 64 else 16
    # digit side channel: a scatter also writes the next pass's digit (1 B); that pass's histogram then reads 1 B per
    # record instead of the record
    side = list(range(lsd))
    tbl_dir = lsd >= 1 and n_a + 8 * (lsd - 1) <= 16      # bucket directory from the last pass's tables (no record scan)
    return {
        "chunks": read_len / (read_len - k + 1),             # validity scan reads every base once
        "encode": read_len / (read_len - k + 1) + r_in,        # read bases, write one record (+ fused first-pass histogram)
        "radix_hist": (lsd - len(side)) * r_out + len(side),   # pass A's histogram is fused in KRN-1
        "radix_scatter": (r_in + r_out) + lsd * 2 * r_out + len(side),  # every pass reads + writes every record once
        "directory": 0.0 if tbl_dir else r_out,                # boundary detection reads the sorted records only when the tables cannot give it
        "bucket_medium": 2 * sfx,                              # read the run, write the distinct suffixes
        "bucket_small": 2 * sfx,
        "bucket_huge": 2 * sfx,
    }


def survey_b_alg(k: int, pb: int, read_len: int) -> float:
    """SURVEY.md §8d whole-path figure: L/(L-K+1) + 4R + BYTES with R = 16 (K <= 45) or 24."""
    kb = 2 * k
    wb = kb + (kb - 1).bit_length()
    by = (wb - pb + 7) // 8
    R = 16 if k <= 45 else 24
    return read_len / (read_len - k + 1) + 4 * R + by


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--prefix-bits", type=int, default=24)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--canonical", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true", help="dev: run the N-GPU code path on a 1-rank RCCL group")
    args = ap.parse_args()

    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        del os.environ["NCCL_DEBUG"]  # RCCL prints its banner (and WARN lines) on STDOUT; keep stdout to the one JSON line
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torchrun --nproc-per-node {args.gpus}", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the product has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_sharded:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if world > 1:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    import cbl_amd
    from cbl_amd import sharded, synth

    K, PB, L, NR = args.k, args.prefix_bits, args.read_len, args.reads
    kmers_per_rank = NR * (L - K + 1)
    d_bases, d_offsets = synth.reads_torch(42, NR, L, first_read=rank * NR, device=f"cuda:{local_rank}")
    torch.cuda.synchronize()

    cbl = cbl_amd.CBL(K, PB, canonical=args.canonical, device=local_rank, profile=True)
    engine = sharded.ShardedBuilder(cbl, dist) if dist is not None else None

    def step():
        cbl.clear()
        if engine is None:
            cbl.insert_seqs_device(d_bases, d_offsets, NR)
        else:
            engine.insert_seqs_device(d_bases, d_offsets, NR)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    cbl.stage_times_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    count = cbl.count()
    if dist is not None:
        t = torch.tensor([count], dtype=torch.int64, device=f"cuda:{local_rank}")
        dist.all_reduce(t)
        count = int(t.item())

    stages = cbl.stage_times()  # rank 0's stream, HIP events around every launch group, timed steps only
    total_kmers = kmers_per_rank * world * args.steps
    value = total_kmers / dt

    # dominant kernel of the step and its roofline position: ALGORITHMIC bytes per launch / average launch time
    # (HIP events recorded on the ctx's own stream around every launch of that kernel, timed steps only)
    alg = stage_alg_bytes(K, PB, L)
    kernel_of = {"radix_scatter": "k_radix_scatter", "radix_hist": "k_radix_hist", "encode": "k_encode",
                 "bucket_medium": "k_bucket_msd", "directory": "k_boundaries+k_bitvector+k_bucket_table", "chunks": "k_scan_invalid+chunk table"}
    cand = {n: ms for n, (ms, _) in stages.items() if n in alg and ms > 0}
    dom = max(cand, key=cand.get) if cand else None
    roofline = None
    if dom:
        ms_total, launches = stages[dom]
        launches = max(int(launches), 1)
        launch_ms = ms_total / launches
        bytes_per_launch = alg[dom] * kmers_per_rank * args.steps / launches
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        traffic = None
        try:  # HBM bytes per launch from the committed PMC passes of this exact configuration (profiles/)
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            c = tj["config"]
            if (c["k"], c["prefix_bits"], c["reads_per_gpu"], c["read_len"]) == (K, PB, NR, L) and tj["kernel"] == kernel_of.get(dom) and world == 1 and engine is None:
                traffic = tj["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        b_alg = survey_b_alg(K, PB, L)
        roofline = {
            "bound": "hbm", "kernel": kernel_of.get(dom, dom), "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "alg_bytes_per_launch": int(bytes_per_launch), "launch_ms_avg": round(launch_ms, 3),
            "launches_per_step": launches / args.steps,
            "whole_path": {  # SURVEY.md §8d accounting over the full step (per GPU)
                "b_alg_per_kmer": round(b_alg, 2),
                "achieved": round(kmers_per_rank * args.steps / dt * b_alg / 1e9, 1),
                "frac": round(kmers_per_rank * args.steps / dt * b_alg / 1e9 / HBM_PEAK_GBPS, 4),
            },
            "stage_ms_per_step": {n: round(ms / args.steps, 3) for n, (ms, _) in stages.items() if ms > 0},
        }

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import Oracle

        ns = min(args.cpu_sample_reads, NR)
        b, o = synth.reads(42, ns, L)
        orc = Oracle(K, PB, args.canonical)
        secs = orc.insert_seqs(b, o)
        cpu = {"value": round(ns * (L - K + 1) / secs, 1), "unit": "k-mers/s", "cores": 1, "kind": "port",
               "sample": f"first {ns} of the same reads (seed 42), one insert_seq call per read, {secs:.1f} s; "
                         "throughput falls as the index grows, so the full-size CPU figure is lower",
               "host_cores_available": os.cpu_count()}

    out = None
    if rank == 0:
        out = {
            "metric": "k-mers inserted/sec (build index)", "value": round(value, 1), "unit": "k-mers/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic (iid ACGT reads, splitmix64 seed 42, resident in HBM)",
            "config": {"workload": f"K={K} (68-bit word) PREFIX_BITS={PB} {NR}x{L}bp reads per GPU, "
                                   f"{'canonical' if args.canonical else 'non-canonical'}, build from empty index",
                       "k": K, "prefix_bits": PB, "reads_per_gpu": NR, "read_len": L,
                       "parallelism": "1 GPU" if world == 1 else f"{world} GPUs: read-sharded encode, prefix-range all-to-all"},
            "distinct_kmers_in_index": count,
            "roofline": roofline, "cpu_baseline": cpu,
        }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:  # last thing on stdout, on a line of its own
        sys.stdout.flush()
        sys.stdout.write("\n" + json.dumps(out) + "\n")
        sys.stdout.flush()



if __name__ == "__main__":
    main()
