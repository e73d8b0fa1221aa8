#!/usr/bin/env python3
"""Rank 0 of a W-GPU sharded build on ONE GPU, with the wire PACED (include/cblx.h: cblx_comm_init_sim): ranks 1 .. W-1 run one after
the other on recording communicators (what each would send rank 0 is kept in device memory), then rank 0 runs the real sharded insert
on a replaying communicator: its exchanges are D2D copies on a side stream held back until `--wire-gbps` per source rank (W - 1 links
in parallel) would have delivered them, while its own KRN-1 + pass A and the receiver's passes / bucket kernels run as in the job.
Reports ms per step of rank 0 with the grouped receiver on / off at every link rate (0 = the copies' own speed).
Usage: tools/emulate_wire.py [--world 8] [--config cfg3|cfg2|cfg4] [--reads N] [--wire-gbps 40,55,75,0] [--groups 8] [--slices 4]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cbl_amd
from cbl_amd import synth

CFG = {"cfg2": (31, 24, 10_000_000, 150), "cfg3": (31, 28, 12_500_000, 150), "cfg4": (59, 28, 6_250_000, 250)}


TAPER = {}


def cuts_of(n, slices):
    if slices in TAPER:  # tapered slices: cumulative fractions
        return [0] + [int(n * f) for f in TAPER[slices][:-1]] + [n]
    return [n * s // slices for s in range(slices + 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--config", choices=sorted(CFG), default="cfg3")
    ap.add_argument("--reads", type=int, default=None, help="reads per rank")
    ap.add_argument("--wire-gbps", default="40,55,75,0")
    ap.add_argument("--groups", default="8", help="groups per rank of the grouped receiver; several values separated by commas")
    ap.add_argument("--slices", type=int, default=4, help="slices of the UNGROUPED run (the grouped one sends nothing before its last slice is through pass A: 1 slice)")
    ap.add_argument("--grouped-slices", default="3", help="slices of the grouped runs (group 0 of every slice but the last crosses under the next slice's kernels); several values separated by commas")
    ap.add_argument("--taper", default="0.5,0.8,1", help="cumulative fractions of the grouped runs' slices instead of equal ones, e.g. 0.5,0.8,1 (the number of values selects the runs with that many slices)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-ungrouped", action="store_true", help="skip the ungrouped receiver's runs")
    ap.add_argument("--protocol", choices=["bins", "sorted", "replicate"], default="bins", help="what crosses the links (sorted: full partition on the sender, packed suffixes; always ungrouped; "
                    "replicate: the reads as bit planes, every rank transforms all of them and keeps its prefix range — grouped receiver, --groups / --grouped-slices apply)")
    ap.add_argument("--no-direct", action="store_true", help="skip the one-GPU build of the same reads")
    ap.add_argument("--rank", type=int, default=0, help="the rank that is rehearsed (CBLX_SIM_TARGET): 0 = the densest prefix range, W - 1 = the sparse tail")
    a = ap.parse_args()
    if a.taper:
        fr = [float(x) for x in a.taper.split(",")]
        TAPER[len(fr)] = fr
    k, pb, nr, L = CFG[a.config]
    nr = a.reads or nr
    W = a.world
    tgt = a.rank
    assert 0 <= tgt < W
    os.environ["CBLX_SIM_TARGET"] = str(tgt)  # read when a rehearsal store is created
    rates = [float(x) for x in a.wire_gbps.split(",")]
    out = {"config": a.config, "k": k, "prefix_bits": pb, "reads_per_rank": nr, "read_len": L, "world": W, "rank": tgt, "kmers_per_rank": nr * (L - k + 1), "runs": []}

    # the one-GPU build of the same share of reads, for reference
    d_b, d_o = synth.reads_torch(42, nr, L, first_read=tgt * nr, device="cuda")
    if not a.no_direct:
        g = cbl_amd.CBL(k, pb)
        g.insert_seqs_device(d_b, d_o, nr)
        ts = []
        for _ in range(a.steps):
            g.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
            g.insert_seqs_device(d_b, d_o, nr); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        out["direct_one_gpu_ms"] = round(min(ts) * 1e3, 3)
        g.close(); del g

    bounds = np.zeros(W - 1, dtype=np.uint32)
    have_bounds = False
    modes = [("grouped", int(x), int(y)) for x in a.groups.split(",") for y in a.grouped_slices.split(",")] + ([] if a.no_ungrouped else [("ungrouped", 1, a.slices)])
    if a.protocol == "sorted":
        modes = [("sorted", 1, a.slices)]
    if a.protocol == "replicate":
        modes = [("replicate", int(x), int(y)) for x in a.groups.split(",") for y in a.grouped_slices.split(",")]
    out["protocol"] = a.protocol
    for mode, groups, slices in modes:
        store = 1000 + groups * 16 + slices + 100000 * tgt
        for r in [x for x in range(W) if x != tgt]:  # the senders: what each would send the rehearsed rank
            rb, ro = synth.reads_torch(42, nr, L, first_read=r * nr, device="cuda")
            cm = cbl_amd.Comm.sim(r, W, store)
            cm.set_protocol(a.protocol)
            cm.set_recv_groups(groups)
            w = cbl_amd.CBL(k, pb)
            have_bounds = w.sharded_insert_seqs_device(cm, rb, ro, nr, cuts_of(nr, slices), bounds, have_bounds)
            w.close(); cm.close()
            del rb, ro
            torch.cuda.empty_cache()
        for gbps in rates:
            cm = cbl_amd.Comm.sim(tgt, W, store, gbps)
            cm.set_protocol(a.protocol)
            cm.set_recv_groups(groups)
            w = cbl_amd.CBL(k, pb, profile=True)
            ts = []
            for it in range(a.steps + 1):
                w.clear(); w.stage_times_reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
                w.sharded_insert_seqs_device(cm, d_b, d_o, nr, cuts_of(nr, slices), bounds, True)
                torch.cuda.synchronize()
                if it:
                    ts.append(time.perf_counter() - t0)
            st = cm.stats()
            run = {"mode": mode, "groups": groups, "groups_used": cm.groups_used(), "groups_fine": cm.groups_fine(), "fine_bins": os.environ.get("CBLX_FINE_BINS", "1") != "0", "slices": slices, "link_gbps": gbps, "ms": [round(t * 1e3, 3) for t in ts], "ms_best": round(min(ts) * 1e3, 3),
                   "words_in_index": w.count(), "recv_bytes_per_step": st["recv_bytes"] // (a.steps + 1), "sent_bytes_per_step": st["sent_bytes"] // (a.steps + 1),
                   "wire_ms_at_rate": round(st["recv_bytes"] / (a.steps + 1) / (W - 1) / (gbps * 1e9) * 1e3, 3) if gbps else None,
                   "stage_ms_last_step": {n: round(ms, 3) for n, (ms, _) in w.stage_times().items() if ms > 0}}
            out["runs"].append(run)
            print(json.dumps(run), file=sys.stderr, flush=True)
            w.close(); cm.close()
        cbl_amd.Comm.sim_store_free(store)
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
