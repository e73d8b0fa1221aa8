#!/usr/bin/env python3
"""What ONE rank of a W-GPU build does at full depth, on one GPU: W shards of reads are encoded + partitioned one after the
other (as W different ranks would), the slice of every sorted batch that belongs to rank `r` is kept, and the receiver step
(cblx_insert_sorted_batches_device) runs on the W batches. Buckets of the rank's prefix range then hold what they would hold
in the real job (W times deeper than a one-GPU build of the rank's own reads). Optionally a second index (seed 43) and `|=`.
Usage: tools/emulate_rank.py [--world 8] [--rank 0] [--reads 6250000] [--k 31] [--prefix-bits 24] [--merge]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cbl_amd
from cbl_amd import sharded, synth


def build_share_words(k, pb, W, rank, NR, L, seed, bounds=None, profile=True):
    """The same share through the WORDS of the rank's range (KRN-1 + one destination pass per sender, the direct pipeline on the
    receiver): passes B / C, the directory and the bucket kernels then see exactly what the receiver of the "bins" protocol sees
    (which gets its words with pass A already done: subtract one scatter pass from radix_scatter)."""
    work = cbl_amd.CBL(k, pb)
    eng = sharded.GpuEngine(work)
    los, his = [], []
    for s in range(W):
        d_b, d_o = synth.reads_torch(seed, NR, L, first_read=s * NR, device="cuda")
        if bounds is None:
            lo, hi = eng.seq_words(d_b, d_o, NR)
            hist = eng.sample_hist(lo, hi)
            bounds = sharded.choose_bounds(hist.cpu().numpy(), W, pb, min(sharded.HIST_BITS, pb))
            del lo, hi
        plo, phi, counts = eng.seq_words_partitioned(d_b, d_o, NR, bounds, W)
        a = int(sum(counts[:rank])); b = a + int(counts[rank])
        los.append(plo[a:b].clone())
        if phi is not None:
            his.append(phi[a:b].clone())
        del plo, phi, d_b, d_o
    work.close()
    lo = torch.cat(los); del los
    hi = torch.cat(his) if his else None
    del his
    torch.cuda.empty_cache()
    g = cbl_amd.CBL(k, pb, profile=profile)
    g.insert_words_device(lo[:4096], hi[:4096] if hi is not None else None, 4096)  # warm the context (first launches, small allocations)
    g.clear(); g.stage_times_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.insert_words_device(lo, hi, int(lo.numel()))
    torch.cuda.synchronize()
    t_recv = time.perf_counter() - t0
    t1 = []
    for _ in range(2):  # warm allocations: the steady-state figure
        g.clear(); g.stage_times_reset(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.insert_words_device(lo, hi, int(lo.numel()))
        torch.cuda.synchronize()
        t1.append(time.perf_counter() - t0)
    return g, bounds, 0.0, min(t1 + [t_recv]), int(lo.numel())


def build_share(k, pb, W, rank, NR, L, seed, bounds=None, profile=True):
    work = cbl_amd.CBL(k, pb)  # the senders' side (one ctx plays all W ranks in turn)
    eng = sharded.GpuEngine(work)
    B = eng.suffix_bytes()
    batches, keep = [], []
    t_send = 0.0
    for s in range(W):
        d_b, d_o = synth.reads_torch(seed, NR, L, first_read=s * NR, device="cuda")
        if bounds is None:
            lo, hi = eng.seq_words(d_b, d_o, NR)
            hist = eng.sample_hist(lo, hi)
            bounds = sharded.choose_bounds(hist.cpu().numpy(), W, pb, min(sharded.HIST_BITS, pb))
            del lo, hi
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bs, ws = eng.sorted_batch_begin(d_b, d_o, NR, bounds, W)
        prefix, count, suffix = eng.sorted_batch_export(int(bs[W]), int(ws[W]))
        torch.cuda.synchronize()
        t_send += time.perf_counter() - t0
        b0, b1, w0, w1 = int(bs[rank]), int(bs[rank + 1]), int(ws[rank]), int(ws[rank + 1])
        p, c, x = prefix[b0:b1].clone(), count[b0:b1].clone(), suffix[w0 * B: w1 * B].clone()
        keep.append((p, c, x))
        batches.append((b1 - b0, w1 - w0, p, c, x))
        del prefix, count, suffix, d_b, d_o
    work.close()
    torch.cuda.empty_cache()
    g = cbl_amd.CBL(k, pb, profile=profile)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.insert_sorted_batches_device(batches)
    torch.cuda.synchronize()
    t_recv = time.perf_counter() - t0
    return g, bounds, t_send, t_recv, sum(b[1] for b in batches)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--reads", type=int, default=6_250_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--prefix-bits", type=int, default=24)
    ap.add_argument("--merge", action="store_true")
    ap.add_argument("--protocol", choices=["sorted", "words"], default="sorted", help="words: the receiver runs the direct pipeline on the words of its range (what the bins protocol's receiver sees, plus pass A)")
    ap.add_argument("--serialize", action="store_true", help="also time the serialized size and the index bytes into a host buffer")
    a = ap.parse_args()
    out = {"config": vars(a)}
    share = build_share_words if a.protocol == "words" else build_share
    g, bounds, ts, tr, nw = share(a.k, a.prefix_bits, a.world, a.rank, a.reads, a.read_len, 42)
    _p, ln, kind = g.bucket_table_np()
    out["build"] = {"words_received": nw, "distinct": g.count(), "buckets": int(len(ln)), "bucket_len_mean": float(ln.mean()), "bucket_len_p99": float(np.percentile(ln, 99)),
                    "bucket_len_max": int(ln.max()), "share_over_4096": float(ln[ln > 4096].sum() / ln.sum()), "share_over_8192": float(ln[ln > 8192].sum() / ln.sum()),
                    "senders_ms_total": ts * 1e3, "receiver_ms": tr * 1e3, "receiver_stage_ms": {n: round(ms, 3) for n, (ms, _) in g.stage_times().items() if ms > 0},
                    "validate": g.validate()}
    if a.serialize:
        t0 = time.perf_counter()
        nbytes = g.serialized_size()
        t1 = time.perf_counter()
        blob = g.serialize_np()
        t2 = time.perf_counter()
        out["serialize"] = {"bytes": int(nbytes), "size_s": round(t1 - t0, 3), "bytes_to_host_s": round(t2 - t1, 3)}
        del blob
    if a.merge:
        h, _, _, _, _ = share(a.k, a.prefix_bits, a.world, a.rank, a.reads, a.read_len, 43, bounds=bounds, profile=False)
        work = cbl_amd.CBL(a.k, a.prefix_bits, profile=True)
        ts_ = []
        for _ in range(3):
            work.clear()
            work |= g
            torch.cuda.synchronize()
            work.stage_times_reset()
            t0 = time.perf_counter()
            work |= h
            torch.cuda.synchronize()
            ts_.append(time.perf_counter() - t0)
        out["merge"] = {"self_words": g.count(), "other_words": h.count(), "union": work.count(), "ms": [round(t * 1e3, 3) for t in ts_],
                        "stage_ms": {n: round(ms, 3) for n, (ms, _) in work.stage_times().items() if ms > 0}, "validate": work.validate(strict=False)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
