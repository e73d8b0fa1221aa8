#!/usr/bin/env python3
"""bench.py — k-mers inserted/sec (build index), K=31 150 bp synthetic reads (BASELINE.json), on N MI355X.

A "step" = one pass of the hot path (insert_seq -> necklace transform -> prefix partition -> bucket insert) over one
batch of synthetic reads already resident in HBM. Workloads (`--config`, BASELINE.json `configs`):

  cfg2  (default at N = 1)  K=31 (68-bit word) PREFIX_BITS=24, 10 M x 150 bp per GPU — the configuration the metric is quoted on
  cfg3  (default at N > 1)  K=31 PREFIX_BITS=28, 12.5 M x 150 bp per GPU (= 100 M reads on 8 GPUs), prefix-sharded
  cfg4                      K=59 (125-bit word) PREFIX_BITS=28, 6.25 M x 250 bp per GPU (= 50 M reads on 8 GPUs)
  merge                     cfg 5's per-GPU share: `A |= B` of two indexes of 6.25 M x 150 bp reads per GPU each (K=31, PB=24);
                            at N > 1 both operands are prefix-sharded with their own quantile bounds and B is re-sharded
                            to A's bounds through the exchange before the per-rank merge

N > 1: one process per GPU. Started under torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the environment) every process is
one rank; started plainly (`python bench.py --gpus 8`) this process only LAUNCHES the N ranks as fresh child processes
(it never touches a GPU itself) and relays rank 0's line. Reads are sharded contiguously by rank, the prefix space by
quantile ranges, one exchange of the partitioned words over RCCL (cbl_amd/sharded.py); per-GPU work is fixed as N grows
("weak"). `--shared-gpu` runs the N ranks on ONE GPU with the exchange staged through gloo (dry run of the N-rank code).

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     - the dominant kernel group of the step (algorithmic bytes / HIP-event time vs the 8 TB/s HBM peak), and
                 `kernels`: the same figures for every kernel group of the step
  cpu_baseline - the CPU oracle (a C++ port of the reference algorithm, oracle/) timed on a bounded sample, 1 thread
  exchange     - (N > 1) world size as the process group reports it, bytes sent / received per rank, effective GB/s
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CONFIGS = {
    "cfg2": dict(k=31, prefix_bits=24, reads=10_000_000, read_len=150, kind="build"),
    "cfg3": dict(k=31, prefix_bits=28, reads=12_500_000, read_len=150, kind="build"),
    "cfg4": dict(k=59, prefix_bits=28, reads=6_250_000, read_len=250, kind="build"),
    "merge": dict(k=31, prefix_bits=24, reads=6_250_000, read_len=150, kind="merge"),
    # SURVEY.md §8d's duplicate-heavy variant: reads sampled at 30x coverage from a 40 Mbp random genome (every k-mer arrives
    # about 24 times in ONE batch): the deduplication and first-occurrence logic carry the load here, not the partition
    "dup": dict(k=31, prefix_bits=24, reads=8_000_000, read_len=150, kind="build", genome=40_000_000),
}


def word_layout(k: int, pb: int):
    kb = 2 * k
    wb = kb + (kb - 1).bit_length()
    hi = 0 if wb <= 64 else (1 if kb <= 64 else 8)
    sfx = 8 if wb - pb <= 64 else 16
    return kb, wb, hi, sfx, (wb - pb + 7) // 8


# ALGORITHMIC bytes per k-mer and step of each kernel group (DESIGN_HISTORY.md §3/§4): what the group's contract must move once.
#   R_in = record bytes out of KRN-1 (8 lo + hi part); after the first partition pass a 65..72-bit word keeps only lo.
def stage_alg_bytes(k: int, pb: int, read_len: int):
    _, _, hi, sfx, _ = word_layout(k, pb)
    r_in = 8 + hi
    r_out = 8 if hi == 1 else r_in
    n_a = min(8, pb)
    lsd = (pb - n_a + 7) // 8
    split = 0
    if pb > 24:  # two LSD passes on 16 bits, the last pb - 24 bits by k_prefix_split: one more read + write of every record, no histogram
        lsd, split = 2, 1
    # digit side channel: a scatter also writes the next LSD pass's digit (1 B); that pass's histogram then reads 1 B per
    # record instead of the record
    side = lsd
    return {
        "chunks": read_len / (read_len - k + 1),             # validity scan reads every base once
        "encode": read_len / (read_len - k + 1) + r_in,        # read bases, write one record (+ fused first-pass histogram)
        "radix_hist": float(side),                             # pass A's histogram is fused in KRN-1; the others read the side channel
        "radix_scatter": (r_in + r_out) + (lsd + split) * 2 * r_out + side,  # every pass (and the prefix split) reads + writes every record once
        # bucket starts: from the last pass's tables (k_dir_gather) or stored by the last scatter itself (PREFIX_BITS > 24,
        # k_dir_resolve); the sorted records are re-read for them only when there is no LSD pass at all (PREFIX_BITS <= 8)
        "directory": 0.0 if lsd >= 1 else r_out,
        "bucket_medium": 2 * sfx,                              # read the run, write the distinct suffixes
        "bucket_small": 2 * sfx,
        "bucket_huge": 2 * sfx,
        "bucket_big": 2 * sfx,
        "merge_gather": 2 * sfx,
    }


# `A |= B`: bytes per word a stage's kernels were GIVEN (cblx_stage_units: the bucket classes split the words between the stages) —
# the gather copies the one-sided buckets (read + write), the sorting classes read both halves and write the result, the Trie |= Trie
# unions are priced on their OUTPUT words by SURVEY.md §8d's figure (2 BYTES read + BYTES written)
def merge_alg_bytes(k: int, pb: int):
    _, _, _, sfx, by = word_layout(k, pb)
    return {"merge_gather": 2 * sfx, "bucket_medium": 2 * sfx, "bucket_huge": 2 * sfx, "bucket_big": 3 * by, "directory": 0.0}


KERNEL_OF = {"radix_scatter": "k_radix_scatter", "radix_hist": "k_radix_hist_bytes", "radix_scan": "k_colscan_*+k_seg_*", "encode": "k_encode",
             "bucket_medium": "k_bucket_sorted (runs that end up sorted; self |= other) + k_bucket_msd (short runs; + k_bucket_claim for runs full of repeats)", "bucket_small": "k_bucket_small", "bucket_huge": "k_bucket_huge",
             "bucket_big": "long runs: k_radix_scatter on the runs + k_bucket_sorted on the sub-ranges (build); k_bucket_union (Trie |= Trie)",
             "directory": "k_dir_gather/k_dir_resolve+k_bitvector+k_bucket_table", "chunks": "k_scan_invalid+chunk table",
             "merge_gather": "k_merge_gather"}


def survey_b_alg(k: int, pb: int, read_len: int) -> float:
    """SURVEY.md §8d whole-path figure: L/(L-K+1) + 4R + BYTES with R = 16 (K <= 45) or 24."""
    by = word_layout(k, pb)[4]
    R = 16 if k <= 45 else 24
    return read_len / (read_len - k + 1) + 4 * R + by


def source_hash() -> str:
    """sha256 over the kernel / host sources of libcblx: ties a committed counter profile to the code it was taken from."""
    h = hashlib.sha256()
    files = sorted(os.listdir(os.path.join(ROOT, "cbl_amd", "csrc")))
    for f in files:
        if f.endswith((".hpp", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(ROOT, "cbl_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None, help="default: cfg2 on 1 GPU, cfg3 on more")
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--prefix-bits", type=int, default=None)
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--canonical", action="store_true")
    ap.add_argument("--protocol", choices=["auto", "bins", "sorted", "replicate", "words"], default=None,
                    help="N > 1: what crosses the links. native transport: auto (default: the library's choice — replicate on 2 - 3 ranks, where one link per pair "
                         "of GPUs bounds every protocol that ships words, sorted on 4, bins otherwise), bins, sorted or replicate (the reads as bit planes, every rank "
                         "transforms all of them and keeps its prefix range); torch transport: sorted (default) or words")
    ap.add_argument("--slices", type=int, default=None,
                    help="N > 1: slices per rank and step. Default 3 for the native bins protocol, 50 / 30 / 20 % of the reads (its grouped receiver sends "
                         "group-major after the rank's whole first pass; the first group's share of slices 0 and 1 crosses under the next slice's kernels, only "
                         "the short last slice's share of it is exposed), 1 for replicate (one all-gather of the planes up front), 4 otherwise (the exchange of a "
                         "slice overlaps the next slice's kernels)")
    ap.add_argument("--transport", choices=["torch", "native"], default="native",
                    help="N > 1: exchange driven from Python over torch.distributed, or the whole sharded insert inside libcblx on RCCL directly")
    ap.add_argument("--cpu-sample-reads", type=int, default=None,
                    help="reads of the CPU leg (default: about 20-30 s of CPU work: 1 M at PREFIX_BITS <= 24, 125 k at 28, 60 k at K = 59)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="launcher mode (--gpus N without WORLD_SIZE): seconds after which the rank processes are stopped and the run fails")
    ap.add_argument("--cpu-full", action="store_true", help="time the CPU oracle on the whole workload (cfg 2: about 10 minutes) and compare the index bytes of both sides (parity_full_size); "
                    "--config merge: the oracle builds both operands, merges them, and both resulting indexes are compared")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-sample", action="store_true", help="skip parity_sample (the CPU leg's sample built once more on the GPU after the timed region, bytes against the oracle's)")
    ap.add_argument("--no-set-check", action="store_true", help="N > 1: skip set_check (count / checksum / validate / membership of the sharded index against every rank's own word stream)")
    ap.add_argument("--no-serialize", action="store_true", help="skip the serialize leg (index bytes into a host buffer)")
    ap.add_argument("--merge-clone", action="store_true", help="--config merge on one GPU: a step = clone A into the work index, then `|=` (the round-4 step) instead of cblx_merge_from")
    ap.add_argument("--no-h2d", action="store_true", help="skip the PCIe-inclusive measurement (value_h2d_inclusive)")
    ap.add_argument("--no-fasta", action="store_true", help="skip the file-inclusive measurement (fasta_inclusive: the same reads as a FASTA file on tmpfs)")
    ap.add_argument("--no-per-record", action="store_true", help="skip the per-record measurement (per_record: one cblx_insert_seq call per read from a C++ host program)")
    ap.add_argument("--recv-groups", type=int, default=0, help="N > 1, native bins protocol: groups per rank of the receiver (0 = default 4, 1 = ungrouped)")
    ap.add_argument("--force-sharded", action="store_true", help="dev: run the N-GPU code path on a 1-rank RCCL group")
    ap.add_argument("--shared-gpu", action="store_true", help="dry run: all ranks on GPU 0, exchange staged through gloo")
    args = ap.parse_args(argv)
    if args.config is None:
        args.config = "cfg2" if args.gpus == 1 else "cfg3"
    if args.protocol is None:
        args.protocol = "auto" if args.transport == "native" else "sorted"
    # what "auto" resolves to inside the library (cblx.h: CBLX_PROTO_AUTO; restated here because the slice schedule follows it)
    args.protocol_resolved = ("replicate" if 2 <= args.gpus <= 3 else ("sorted" if args.gpus == 4 else "bins")) if args.protocol == "auto" else args.protocol
    cfg = CONFIGS[args.config]
    for name in ("k", "prefix_bits", "reads", "read_len"):
        if getattr(args, name) is None:
            setattr(args, name, cfg[name])
    if args.slices is None:
        args.slices = 3 if (args.transport == "native" and args.protocol_resolved == "bins" and not args.force_sharded) else (1 if args.protocol_resolved == "replicate" else 4)
    args.kind = cfg["kind"]
    args.genome = cfg.get("genome", 0)
    if args.cpu_sample_reads is None:  # the oracle slows down with PREFIX_BITS (2^28-bit Fenwick bitvector) and word width
        args.cpu_sample_reads = 60_000 if args.k > 45 else (125_000 if args.prefix_bits > 24 else 1_000_000)
    return args


def build_in_child():
    """__graft_entry__.build() in a process of its own, under a lock file (torchrun starts N of us at once). Not in this
    process: build() dlopens libcblx.so, which would pull in /opt/rocm's HIP runtime before torch loads its bundled copy —
    two runtimes in one process see no device — and hipcc / g++ children must not inherit a profiler's preload."""
    import fcntl
    from pathlib import Path

    # nothing to do (the usual case, and the only one allowed under a profiler: a process whose GPU runtime a preloaded
    # profiler library has initialised must not start children) -> no child at all
    csrc = Path(ROOT) / "cbl_amd" / "csrc"
    lib = Path(ROOT) / "cbl_amd" / "libcblx.so"
    srcs = list(csrc.glob("*.cpp")) + list(csrc.glob("*.hpp")) + [Path(ROOT) / "include" / "cblx.h"]
    orc = list((Path(ROOT) / "oracle" / "_build").glob("*/liboracle.so"))
    osrc = [Path(ROOT) / "oracle" / "cbl_oracle_capi.cpp", Path(ROOT) / "oracle" / "cbl_oracle.hpp"]
    if lib.exists() and all(lib.stat().st_mtime >= f.stat().st_mtime for f in srcs) and orc and all(orc[0].stat().st_mtime >= f.stat().st_mtime for f in osrc):
        return
    with open(os.path.join(ROOT, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES")}
            subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], check=True, cwd=ROOT, env=env, stdout=sys.stderr)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def cpu_baseline_leg(args, genome_reads=None, keep=None, keep_sample=None):
    """The CPU oracle (C++ port of the reference algorithm, oracle/) on a bounded sample of the same reads, 1 thread: one
    insert_seq call per read. Needs no GPU and no process group: at N > 1 rank 0 runs it BEFORE the group is formed (the other
    ranks wait in the rendezvous), so every line of a scaling run carries it."""
    import numpy as np_

    from cbl_amd import synth
    from oracle import Oracle

    K, PB, L, NR = args.k, args.prefix_bits, args.read_len, args.reads
    ns = NR if args.cpu_full else min(args.cpu_sample_reads, NR)
    orc = Oracle(K, PB, args.canonical)
    secs, done, blk = 0.0, 0, 1_000_000
    curve = []
    while done < ns:  # fed in blocks of 1 M reads so that the decline with index size is on record
        m = min(blk, ns - done)
        if genome_reads is not None:
            b, _ = genome_reads(done, m)
            o = (np_.arange(m + 1, dtype=np_.uint64) * np_.uint64(L))
        else:
            b, o = synth.reads(42, m, L, first_read=done)
        s1 = orc.insert_seqs(b, o)
        secs += s1
        done += m
        curve.append(round(m * (L - K + 1) / s1 / 1e6, 2))
    if keep is not None and ns == NR:  # --cpu-full: the finished oracle index is what the GPU's bytes are compared with (parity_full_size)
        keep.append(orc)
    elif keep_sample is not None and genome_reads is None:  # the default run: the oracle's index of the sample is the checker of `parity_sample`
        keep_sample.append((orc, ns))
    return {"value": round(ns * (L - K + 1) / secs, 1), "unit": "k-mers/s", "cores": 1, "kind": "port",
            "sample": (f"all {ns} reads" if ns == NR else f"first {ns} of rank 0's reads") + f" (seed 42), one insert_seq call per read, {secs:.1f} s"
                      + ("" if ns == NR else "; throughput falls as the index grows, so the full-size CPU figure is lower (profiles/ holds a full run)"),
            "mkmers_per_s_by_1M_read_block": curve if len(curve) > 1 else None,
            "host_cores_available": os.cpu_count()}


def parity_full_size(cbl, orc, kmers):
    """Index bytes of the GPU build against the oracle's, whole workload. Both sides are hashed in 256 MiB chunks."""
    import hashlib

    import numpy as np

    t0 = time.perf_counter()
    g = cbl.serialize_np()
    t1 = time.perf_counter()
    o = orc.serialize_np()
    t2 = time.perf_counter()
    hg, ho = hashlib.sha256(), hashlib.sha256()
    first_diff = None
    step = 256 << 20
    for a in range(0, max(g.size, o.size), step):
        cg, co = g[a:a + step], o[a:a + step]
        hg.update(memoryview(cg))
        ho.update(memoryview(co))
        if first_diff is None and not (cg.size == co.size and np.array_equal(cg, co)):
            m = min(cg.size, co.size)
            d = np.flatnonzero(cg[:m] != co[:m])
            first_diff = int(a + (d[0] if d.size else m))
    equal = first_diff is None and g.size == o.size
    return {"equal": bool(equal), "bytes": int(g.size), "bytes_oracle": int(o.size), "sha256": hg.hexdigest(), "sha256_oracle": ho.hexdigest(),
            "first_difference_at": first_diff, "kmers_inserted": int(kmers), "distinct_kmers": int(cbl.count()), "distinct_kmers_oracle": int(orc.count()),
            "gpu_serialize_s": round(t1 - t0, 2), "oracle_serialize_s": round(t2 - t1, 2), "compare_s": round(time.perf_counter() - t2, 2),
            "what": "cblx_serialize of the timed build's index (all of the workload's reads) vs the CPU oracle's serialize after one insert_seq per read: "
                    "same length, same SHA-256, no differing byte"}


def merge_parity_full_size(args, work, other, kmers_b):
    """cfg 5 at its per-GPU size through the checker (`--config merge --cpu-full`, one GPU): the oracle builds A (seed 42) and B (seed 43)
    with one insert_seq per read, merges them (`A |= B`, /root/reference/src/cbl.rs:433-449), and its bytes are compared with the merged
    index of the last timed step — and B's with the GPU's B, whose both-sided Vec buckets the merge left sorted
    (/root/reference/src/trievec/mod.rs:209-220). Returns (parity record, cpu_baseline of the merge)."""
    from cbl_amd import synth
    from oracle import Oracle

    K, PB, L, NR = args.k, args.prefix_bits, args.read_len, args.reads

    def build(seed):
        o = Oracle(K, PB, args.canonical)
        secs = 0.0
        for done in range(0, NR, 1_000_000):
            m = min(1_000_000, NR - done)
            b, off = synth.reads(seed, m, L, first_read=done)
            secs += o.insert_seqs(b, off)
        return o, secs

    oa, sa = build(42)
    ob, sb = build(43)
    t0 = time.perf_counter()
    oa.merge(ob)
    tm = time.perf_counter() - t0
    rec = parity_full_size(work, oa, kmers_b)
    rec["what"] = ("cblx_serialize of the merged index (A |= B, the last timed step's) vs the CPU oracle's bytes after the same merge of the same two "
                   "indexes: same length, same SHA-256, no differing byte; `other_after_merge`: B itself, whose Vec buckets the merge sorted")
    ro = parity_full_size(other, ob, kmers_b)
    rec["other_after_merge"] = {k: ro[k] for k in ("equal", "bytes", "bytes_oracle", "sha256", "sha256_oracle", "first_difference_at")}
    rec["equal"] = bool(rec["equal"] and ro["equal"])
    rec["oracle_build_s"] = [round(sa, 1), round(sb, 1)]
    rec["oracle_merge_s"] = round(tm, 2)
    cpu = {"value": round(kmers_b / tm, 1), "unit": "k-mers/s", "cores": 1, "kind": "port",
           "sample": f"the oracle's merge of the two full indexes ({NR} reads each, seeds 42 / 43): {tm:.1f} s (their builds, one insert_seq per read: {sa:.0f} + {sb:.0f} s)",
           "mkmers_per_s_by_1M_read_block": None, "host_cores_available": os.cpu_count()}
    return rec, cpu


def parity_sample_leg(args, orc, ns, d_bases, d_offsets, device):
    """The default run's own parity verdict: the reads the CPU leg timed (the first `ns` of rank 0's) built once more on the GPU — after the
    timed region, in a context of their own — and the serialized bytes compared with the oracle's index of exactly those reads, which the CPU
    leg built anyway. Length, SHA-256 of both sides, first differing byte. The oracle is the checker, never the thing measured."""
    import cbl_amd

    K, PB, L = args.k, args.prefix_bits, args.read_len
    g = cbl_amd.CBL(K, PB, canonical=args.canonical, device=device)
    try:
        g.insert_seqs_device(d_bases[: ns * L], d_offsets[: ns + 1], ns)
        rec = parity_full_size(g, orc, ns * (L - K + 1))
    finally:
        g.close()
    rec["reads"] = int(ns)
    rec["what"] = (f"cblx_serialize of a GPU build of the first {ns} of rank 0's reads (the CPU leg's sample, built after the timed region in a context of its own) vs the "
                   "CPU oracle's serialize of the index the cpu_baseline leg built from the same reads: same length, same SHA-256, no differing byte")
    return rec


def set_check_leg(args, cbl, d_bases, d_offsets, rank, world, dev, reduce_sums):
    """An N > 1 line's own correctness verdict on the SHARDED index the timed steps left (every rank holds its prefix range):
      - `count` (all-reduced) against the k-mers inserted: the difference is the repeats of the stream (iid 31-mers: n^2 / 2 / 4^K of them);
      - the set checksum (a sum of word hashes, all-reduced) against the same sum over every rank's OWN word stream (KRN-1 alone): equal when
        the stream holds no repeat — with repeats the index sum is the smaller by their hashes and `checksums_equal` is null;
      - `validate()` of every share (prefix in range and in its bucket, Trie runs ascending, no repeat inside a bucket), all-reduced;
      - membership: every rank's reads (regenerated here, seed 42) queried against THIS rank's share; summed over the shares every inserted
        k-mer must be found exactly once — a word that went to the wrong rank, or was lost on the way, is missing here.
    `reduce_sums(list of ints) -> list of ints` sums over the ranks (values < 2^62)."""
    import torch

    from cbl_amd import synth

    K, L, NR = args.k, args.read_len, args.reads
    n_kmers = NR * (L - K + 1)
    M32 = (1 << 32) - 1
    cnt, cs, bad = cbl.count(), cbl.checksum(), cbl.validate()
    lo = torch.empty(n_kmers + 1, dtype=torch.int64, device=dev)
    hb = cbl.consts()["hi_bytes"]
    hi = None if hb == 0 else torch.zeros(n_kmers + 1, dtype=torch.uint8 if hb == 1 else torch.int64, device=dev)
    nw = cbl.seq_words_device(d_bases, d_offsets, NR, lo, hi, n_kmers)
    cs_w = cbl.checksum_words_device(lo, hi, nw)
    del lo, hi
    found = queried = 0
    for r in range(world):
        if r == rank or args.genome:
            qb, qo = d_bases, d_offsets
        else:
            qb, qo = synth.reads_torch(42, NR, L, first_read=r * NR, device=dev)
        t, f = cbl.contains_seqs_device(qb, qo, NR)
        queried += t
        found += f
        del qb, qo
        if args.genome:
            break
    tot = reduce_sums([cnt, cs & M32, cs >> 32, bad, nw, cs_w & M32, cs_w >> 32, found, queried])
    cs_index = (tot[1] + (tot[2] << 32)) & ((1 << 64) - 1)
    cs_stream = (tot[5] + (tot[6] << 32)) & ((1 << 64) - 1)
    repeats = tot[4] - tot[0]
    expected_found = tot[4] if not args.genome else None  # every share was asked about every rank's reads
    members_ok = None if args.genome else (tot[7] == expected_found and tot[8] == world * tot[4])
    # iid reads: repeats ~ n^2 / (2 * 4^K) (+ the all-ones / low-complexity words count as any other); anything above a generous multiple is a loss
    bound = None if args.genome else int(8 * tot[4] * tot[4] / (2 * 4.0 ** K)) + 64
    ok = (tot[3] == 0 and repeats >= 0 and (bound is None or repeats <= bound) and (members_ok is not False)
          and (repeats != 0 or cs_index == cs_stream))
    return {"ok": bool(ok), "count": int(tot[0]), "kmers_inserted": int(tot[4]), "repeats_implied": int(repeats), "repeats_bound": bound,
            "checksum_index": f"{cs_index:016x}", "checksum_word_streams": f"{cs_stream:016x}",
            "checksums_equal": (cs_index == cs_stream) if repeats == 0 else None,
            "validate_violations": int(tot[3]), "members_found": int(tot[7]), "members_expected": expected_found, "members_ok": members_ok,
            "what": "sharded index after the timed steps: all-reduced cblx_count / cblx_checksum / cblx_validate of the shares; cblx_checksum_words_device over every "
                    "rank's own KRN-1 word stream; every rank's reads (regenerated) through cblx_contains_seqs_device against every share, positives summed"}


# ---- launcher: `python bench.py --gpus N` without torchrun -------------------------------------------------------------
def launch_ranks(args) -> int:
    """Start N fresh rank processes (this process never initialises a GPU), relay rank 0's stdout, return the worst code."""
    build_in_child()  # once, before any rank exists
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", CBLX_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    return supervise_ranks(procs, args.launch_timeout)


def supervise_ranks(procs, deadline_s: float) -> int:
    """Wait for the rank processes (fresh children of a launcher that never touched a GPU). Rank 0's stdout is drained by a thread
    so that it can never block on a full pipe; the children are POLLED: when one exits non-zero, or the deadline passes, the others
    — which would otherwise wait for it in a collective for ever — are terminated (SIGTERM, then SIGKILL after 10 s) and the result
    is non-zero. Returns 0 only if every rank returned 0."""
    import threading

    chunks = []
    rd = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    t_end = time.monotonic() + deadline_s
    rc, why = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            rc, why = max(abs(c) for _, c in bad) or 1, f"rank {bad[0][0]} exited with code {bad[0][1]}"
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > t_end:
            rc, why = 124, f"no result after {deadline_s:.0f} s"
            break
        time.sleep(0.05)
    if why is not None:
        print(f"bench.py launcher: {why}; stopping the other ranks", file=sys.stderr)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(t_kill - time.monotonic(), 0.1))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    rd.join(timeout=5)
    sys.stdout.write(b"".join(chunks).decode(errors="replace"))
    sys.stdout.flush()
    return rc


def per_record_leg(NR, K, PB, L):
    """The reference's own call pattern (/root/reference/examples/cbl.rs:160-163: one insert_seq per record), from a plain C++ host
    program built here against include/cblx.h — Python's per-call cost would be what is measured otherwise. Own process, own context."""
    import subprocess
    import tempfile

    root = os.path.dirname(os.path.abspath(__file__))
    src = os.path.join(root, "tools", "dev_insert_seq_rate.cpp")
    if (K, PB, L) != (31, 24, 150):
        return {"value": None, "error": "the probe is written for cfg 2 (K=31, PREFIX_BITS=24, 150 bp)"}
    try:
        with tempfile.TemporaryDirectory() as d:
            exe = os.path.join(d, "per_record")
            lib = os.path.join(root, "cbl_amd")
            subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "include"), "-o", exe, src, "-L", lib, "-lcblx", "-Wl,-rpath," + lib],
                           check=True, capture_output=True, timeout=120)
            r = subprocess.run([exe, str(NR), "json"], check=True, capture_output=True, text=True, timeout=300)
        out = json.loads(r.stdout.strip().splitlines()[-1])
        out["unit"] = "k-mers/s"
        out["source"] = ("tools/dev_insert_seq_rate.cpp: the step's read shape (iid ACGT, own generator), one cblx_insert_seq call per read from pageable host memory "
                         "+ cblx_flush, best of 2 after 1 warm-up; ms_calls = the call loop (the queue's pinned blocks are DMA'd while it runs), ms_flush = what is left")
        return out
    except Exception as e:
        return {"value": None, "error": f"{type(e).__name__}: {e}"}


def fasta_leg(cbl, h_bases, NR, L, kmers):
    """The reads of the step as a FASTA file (`>r<i>` + one sequence line per record) in /dev/shm, then file -> finished index."""
    import tempfile

    import numpy as np

    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    path = os.path.join(d, f"cblx_bench_{os.getpid()}.fa")
    try:
        with open(path, "wb") as f:
            step = 1_000_000
            for a0 in range(0, NR, step):
                n = min(step, NR - a0)
                rec = np.empty((n, 11 + L + 1), dtype=np.uint8)
                rec[:, 0], rec[:, 1], rec[:, 10], rec[:, -1] = ord(">"), ord("r"), 10, 10
                ids = np.arange(a0, a0 + n)
                for dgt in range(8):
                    rec[:, 9 - dgt] = 48 + (ids // 10 ** dgt) % 10
                rec[:, 11:11 + L] = h_bases[a0 * L:(a0 + n) * L].reshape(n, L)
                f.write(rec.tobytes())
        size = os.path.getsize(path)
        ts = []
        for _ in range(3):
            cbl.clear()
            time.sleep(0.3)  # the previous repetition's mapping of the file is torn down by a helper thread
            t1 = time.perf_counter()
            nrec = cbl.insert_fastx_file(path)
            cbl.flush()
            ts.append(time.perf_counter() - t1)
            assert nrec == NR
        best = min(ts[1:])
        return {"value": round(kmers / best, 1), "unit": "k-mers/s", "ms_per_step": round(best * 1e3, 3), "file_bytes": int(size),
                "file_gbps": round(size / best / 1e9, 2), "distinct_kmers_in_index": int(cbl.count()),
                "source": f"single-line FASTA in {d} (page cache) -> cblx_insert_fastx_file + cblx_flush, best of 2 after 1 warm-up: counting pass, "
                          "parser threads pack the sequence lines into bit planes in place, the sliced insert runs behind them (DESIGN_HISTORY.md §3.10)"}
    except Exception as e:  # no room for the file, or no tmpfs: the line says so
        return {"value": None, "error": f"{type(e).__name__}: {e}"}
    finally:
        try:
            os.remove(path)
        except OSError:
            pass


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        sys.exit(launch_ranks(args))
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.shared_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        del os.environ["NCCL_DEBUG"]  # RCCL prints its banner (and WARN lines) on STDOUT; keep stdout to the one JSON line

    # build before torch / any GPU call (hipcc and g++ children must not inherit an initialised runtime or a profiler's
    # preload); under torchrun local rank 0 builds and the others wait on a lock file for it
    if not os.environ.get("CBLX_BENCH_CHILD"):
        build_in_child()

    # the CPU leg first: no GPU, no process group (rank 0 only; at N > 1 the other ranks wait for it in the rendezvous)
    cpu_early = None
    oracle_full = []  # --cpu-full: the oracle's finished index of ALL of rank 0's reads, the checker of parity_full_size
    oracle_sample = []  # the default run: (oracle, reads) of the CPU leg's sample, the checker of parity_sample
    if rank == 0 and not args.no_cpu_baseline and args.kind == "build" and not args.genome:
        cpu_early = cpu_baseline_leg(args, keep=oracle_full, keep_sample=None if args.no_parity_sample else oracle_sample)

    import torch

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the product has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    dist = None
    transport = "none"
    if world > 1 or args.force_sharded:
        import torch.distributed as tdist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.shared_gpu:
            from cbl_amd.sharded import HostStagedGroup

            tdist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist = HostStagedGroup(tdist)
            transport = "gloo, staged through the host (all ranks share GPU 0: dry run)"
        else:
            tdist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            dist = tdist
            transport = "RCCL grouped send/recv (torch.distributed nccl backend)"

    import cbl_amd
    from cbl_amd import sharded, synth

    K, PB, L, NR = args.k, args.prefix_bits, args.read_len, args.reads
    kmers_per_rank = NR * (L - K + 1)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def allreduce_max(x: float) -> float:
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=tdist.ReduceOp.MAX)
        return float(t.item())

    def allreduce_sum(x: int) -> int:
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        return int(t.item())

    extra = {}
    engine = None
    if args.kind == "build":
        if args.genome:
            # reads = windows of one random genome (splitmix64 seed 4242) at uniform random positions (torch generator seed 7 + rank)
            gen, _ = synth.reads_torch(4242, 1, args.genome, device=dev)
            gtor = torch.Generator(device=dev)
            gtor.manual_seed(7 + rank)
            pos = torch.randint(0, args.genome - L, (NR,), device=dev, dtype=torch.int64, generator=gtor)
            ar = torch.arange(L, device=dev)
            d_bases = torch.cat([gen[(pos[a:a + 1_000_000, None] + ar[None, :]).reshape(-1)] for a in range(0, NR, 1_000_000)])
            d_offsets = torch.arange(0, (NR + 1) * L, L, device=dev, dtype=torch.int64)
            del gen, pos
        else:
            d_bases, d_offsets = synth.reads_torch(42, NR, L, first_read=rank * NR, device=dev)
        torch.cuda.synchronize()
        cbl = cbl_amd.CBL(K, PB, canonical=args.canonical, device=local_rank, profile=True)
        comm = None
        if dist is not None and args.transport == "native":
            if args.shared_gpu:
                comm = cbl_amd.Comm.over_group(tdist, rank, world, local_rank)
                transport = "libcblx sharded insert, host callbacks over gloo (all ranks share GPU 0: dry run)"
            else:
                box = [cbl_amd.Comm.unique_id() if rank == 0 else None]
                tdist.broadcast_object_list(box, src=0)
                comm = cbl_amd.Comm.rccl(box[0], rank, world, local_rank)
                transport = "libcblx sharded insert on RCCL (ncclSend / ncclRecv groups on a side stream)"
        if comm is not None and args.recv_groups:
            comm.set_recv_groups(args.recv_groups)
        # the tapered slices belong to the GROUPED receiver's schedule: not to the ungrouped one (--recv-groups 1), not to "sorted"
        grouped = (comm is not None and args.protocol_resolved == "bins" and args.recv_groups != 1 and args.slices == len(sharded.ShardedBuilder.GROUPED_WEIGHTS)
                   and not args.force_sharded)
        engine = sharded.ShardedBuilder(cbl, dist, slices=args.slices, protocol=args.protocol, comm=comm,
                                        slice_weights=sharded.ShardedBuilder.GROUPED_WEIGHTS if grouped else None) if dist is not None else None

        def step(_i):
            cbl.clear()
            if engine is None:
                cbl.insert_seqs_device(d_bases, d_offsets, NR)
            else:
                engine.insert_seqs_device(d_bases, d_offsets, NR)

        units_per_rank_step = kmers_per_rank
        alg = stage_alg_bytes(K, PB, L)
        timed_ctx = cbl
    else:
        # `A |= B` (src/cbl.rs:433-449): A from the reads of seed 42, B from seed 43 (SURVEY.md §8d cfg 5). Every step merges B
        # into a copy of A (the merge changes self).
        a_bases, a_off = synth.reads_torch(42, NR, L, first_read=rank * NR, device=dev)
        b_bases, b_off = synth.reads_torch(43, NR, L, first_read=rank * NR, device=dev)
        # every step works on ONE reusable copy (`work`): clear, clone A into it (|= into an empty index is a device copy of
        # A's share, a few ms, inside the step), then the merge proper. Fresh contexts per step would time hipMalloc: a new
        # ctx has a cold allocation cache and a 12 GB arena costs more to map than the merge takes.
        if dist is None:
            A = cbl_amd.CBL(K, PB, canonical=args.canonical, device=local_rank)
            A.insert_seqs_device(a_bases, a_off, NR)
            B = cbl_amd.CBL(K, PB, canonical=args.canonical, device=local_rank)
            B.insert_seqs_device(b_bases, b_off, NR)
            work = cbl_amd.CBL(K, PB, canonical=args.canonical, device=local_rank, profile=True)
            count_a, count_b = A.count(), B.count()

            # `work = A; work |= B` without the copy (cblx_merge_from: the device merge writes a new arena anyway; A stays as it is, B
            # as `|=` leaves it). --merge-clone restores the round-4 step (clear, clone A into the work index, `|=`): 2.2 ms more
            if args.merge_clone:
                def step(_i):
                    work.clear()
                    work.__ior__(A)
                    work.__ior__(B)
            else:
                def step(_i):
                    work.merge_from(A, B)

            last = lambda: work  # noqa: E731
        else:
            A = sharded.ShardedIndex(K, PB, dist, canonical=args.canonical, device=local_rank, slices=args.slices)
            A.insert_seqs_device(a_bases, a_off, NR)
            B = sharded.ShardedIndex(K, PB, dist, canonical=args.canonical, device=local_rank, slices=args.slices)
            B.insert_seqs_device(b_bases, b_off, NR)  # its own quantile bounds
            work = sharded.ShardedIndex(K, PB, dist, canonical=args.canonical, device=local_rank, slices=args.slices, profile=True)
            count_a, count_b = A.local_count(), B.local_count()
            engine = B  # exchange accounting of the re-shard

            def step(_i):
                work.copy_from(A)
                work.merge_assign(B.resharded(work.bounds))  # B itself keeps its own bounds: every step pays the exchange

            last = lambda: work.cbl  # noqa: E731
        del a_bases, b_bases
        torch.cuda.synchronize()
        units_per_rank_step = count_b  # k-mers of B inserted into A per step (this rank's share)
        alg = merge_alg_bytes(K, PB)
        extra["merge"] = {"words_self": allreduce_sum(count_a), "words_other": allreduce_sum(count_b)}
        timed_ctx = None

    for i in range(args.warmup):
        step(i)
    fence()
    (cbl if args.kind == "build" else last()).stage_times_reset()
    if engine is not None and hasattr(engine, "reset_stats"):
        engine.reset_stats()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    dt = allreduce_max(time.perf_counter() - t0)

    if args.kind == "build":
        count = allreduce_sum(cbl.count())
        stages = cbl.stage_times()  # rank 0's stream, HIP events around every launch group, timed steps only
        scatter_record_passes = cbl.stage_units().get("radix_scatter", 0)  # records x partition passes over the timed steps (the passes a record takes vary with the route)
        units_alg = kmers_per_rank * args.steps  # k-mers through the kernels of this rank in the timed region
    else:
        count = allreduce_sum(last().count())
        stages = last().stage_times()
        stage_units = last().stage_units()  # words per stage over the timed steps (the bucket classes split them)
        units_alg = (count_a + count_b) * args.steps  # words of the merged runs
        extra["merge"]["words_union"] = count
    total_units = allreduce_sum(units_per_rank_step) * args.steps
    value = total_units / dt

    # every kernel group of the step against the HBM roofline: ALGORITHMIC bytes per launch / average launch time (HIP
    # events recorded on the ctx's own stream around every launch of the group, timed steps only); `dom` = the slowest
    kernels = []
    if args.kind == "build":
        stage_units = {}
        if scatter_record_passes and engine is None:
            # every record-pass reads and writes a record behind the first pass's layout; the first pass reads KRN-1's wider records; every
            # pass but a record's last also writes the next pass's digit byte
            _, _, hi_b, _, _ = word_layout(K, PB)
            r_in = 8 + hi_b
            r_out = 8 if hi_b == 1 else r_in
            alg = dict(alg)
            alg["radix_scatter"] = (scatter_record_passes * 2 * r_out + units_alg * (r_in - r_out) + max(scatter_record_passes - units_alg, 0)) / units_alg
            extra["partition_passes_per_record"] = round(scatter_record_passes / units_alg, 3)
    # the bucket kernels split the words between them by run length and (builds) the split is not reported: the whole word count is
    # priced against the slowest of them (at these workloads it holds > 99 % of the words), the others carry no fraction
    bucket_stages = [n for n in ("bucket_medium", "bucket_small", "bucket_big", "bucket_huge") if stages.get(n, (0, 0))[0] > 0]
    bucket_main = max(bucket_stages, key=lambda n: stages[n][0]) if bucket_stages else None
    for n, (ms, launches) in stages.items():
        if ms <= 0 or n not in alg:
            continue
        launches = max(int(launches), 1)
        row = {"name": KERNEL_OF.get(n, n), "stage": n, "ms_per_step": round(ms / args.steps, 3), "launches_per_step": launches / args.steps,
               "launch_ms_avg": round(ms / launches, 3)}
        if stage_units.get(n, 0) > 0:  # the library counted the words this stage was given
            bpl = alg[n] * stage_units[n] / launches
            ach = bpl / (ms / launches * 1e-3) / 1e9
            row.update({"alg_bytes_per_launch": int(bpl), "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBPS, 4), "words_per_step": int(stage_units[n] / args.steps)})
        elif args.kind != "build" or (n in bucket_stages and n != bucket_main):
            row.update({"alg_bytes_per_launch": None, "achieved": None, "frac": None})
        else:
            bpl = alg[n] * units_alg / launches
            ach = bpl / (ms / launches * 1e-3) / 1e9 if bpl else 0.0
            row.update({"alg_bytes_per_launch": int(bpl), "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBPS, 4)})
        kernels.append(row)
    kernels.sort(key=lambda x: -x["ms_per_step"])
    roofline_error = None
    for row in kernels:  # an algorithmic rate above the peak is a bookkeeping error, never a result: the row loses its fraction, the line says so and the run fails AFTER printing
        if row["frac"] is not None and row["frac"] > 1.0:
            roofline_error = (roofline_error or "") + f"stage {row['stage']}: algorithmic rate {row['achieved']} GB/s above the HBM peak (pricing error); "
            row["frac_error"], row["frac"] = row["frac"], None
    roofline = None
    if kernels:
        dom = kernels[0]
        traffic = None
        try:  # HBM bytes per launch from the committed PMC passes — only if they were taken from exactly these sources
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            c = tj["config"]
            if (c["k"], c["prefix_bits"], c["reads_per_gpu"], c["read_len"], c.get("kind", "build")) == (K, PB, NR, L, args.kind) and \
                    tj["kernel"] == dom["name"] and tj.get("src_sha") == source_hash() and world == 1 and engine is None:
                traffic = tj["hbm_bytes_per_launch"]
                for row in kernels:  # the other kernel groups the counter passes covered: HBM bytes per launch of the group
                    kt = tj.get("kernels", {}).get(row["stage"])
                    if kt and row["launches_per_step"]:
                        row["traffic"] = int(kt["hbm_bytes_per_step"] / row["launches_per_step"])
        except (OSError, KeyError, ValueError):
            pass
        roofline = {"bound": "hbm", "kernel": dom["name"], "achieved": dom["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": dom["frac"],
                    "traffic": traffic, "alg_bytes_per_launch": dom["alg_bytes_per_launch"], "launch_ms_avg": dom["launch_ms_avg"],
                    "launches_per_step": dom["launches_per_step"], "kernels": kernels,
                    "stage_ms_per_step": {n: round(ms / args.steps, 3) for n, (ms, _) in stages.items() if ms > 0}}
        if args.kind == "build":
            b_alg = survey_b_alg(K, PB, L)
            roofline["whole_path"] = {  # SURVEY.md §8d accounting over the full step (per GPU)
                "b_alg_per_kmer": round(b_alg, 2),
                "achieved": round(kmers_per_rank * args.steps / dt * b_alg / 1e9, 1),
                "frac": round(kmers_per_rank * args.steps / dt * b_alg / 1e9 / HBM_PEAK_GBPS, 4)}
        else:
            by = word_layout(K, PB)[4]
            roofline["whole_path"] = {  # 2 BYTES read + BYTES written per output suffix
                "b_alg_per_output_word": 3 * by,
                "achieved": round(count / max(world, 1) * args.steps / dt * 3 * by / 1e9, 1),
                "frac": round(count / max(world, 1) * args.steps / dt * 3 * by / 1e9 / HBM_PEAK_GBPS, 4)}

    # BASELINE.md §3's gate at FULL size (`--cpu-full`, one GPU): the bytes `cbl build` would write (/root/reference/examples/cbl.rs:147-167,
    # 132-142) from the timed build's index against the CPU oracle's bytes for the same reads — length, SHA-256 of both, and a
    # chunk-by-chunk comparison. The oracle is the checker here, never the thing measured.
    if args.kind == "build" and oracle_full and world == 1 and engine is None:
        extra["parity_full_size"] = parity_full_size(cbl, oracle_full.pop(), kmers_per_rank)

    # the line's own correctness evidence: the sharded index against every rank's word stream (N > 1), and the CPU leg's sample
    # rebuilt on the GPU and byte-compared with the oracle's index of it (rank 0)
    if args.kind == "build" and engine is not None and world > 1 and not args.no_set_check:
        def reduce_sums(vals):
            t = torch.tensor(vals, dtype=torch.int64, device=dev)
            dist.all_reduce(t)
            return [int(x) for x in t.cpu().tolist()]
        extra["set_check"] = set_check_leg(args, cbl, d_bases, d_offsets, rank, world, dev, reduce_sums)
    if args.kind == "build" and oracle_sample:
        orc_s, ns_s = oracle_sample.pop()
        try:
            extra["parity_sample"] = parity_sample_leg(args, orc_s, ns_s, d_bases, d_offsets, local_rank)
        except Exception as e:  # (the line must come out: the verdict is then "not equal")
            extra["parity_sample"] = {"equal": False, "error": f"{type(e).__name__}: {e}"}
        del orc_s

    # what the N-GPU code path costs a rank over the direct build of the same reads (no exchange, no slices): the direct steps
    # run AFTER the timed region, each rank on its own reads
    if args.kind == "build" and engine is not None:
        cbl.clear()
        cbl.insert_seqs_device(d_bases, d_offsets, NR)  # warm-up of the direct path's allocations
        fence()
        cbl.stage_times_reset()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            cbl.clear()
            cbl.insert_seqs_device(d_bases, d_offsets, NR)
        torch.cuda.synchronize()
        d_dt = time.perf_counter() - t1
        d_stages = cbl.stage_times()
        sh_ms, di_ms = dt / args.steps * 1e3, d_dt / args.steps * 1e3
        if world > 1:
            # the denominator of a scaling figure: THIS configuration on one GPU (rank 0 building its own share directly, on the same box,
            # right after the timed steps) — the driver's N = 1 line is another configuration (cfg 2)
            one = kmers_per_rank / (d_dt / args.steps)
            extra["one_gpu_same_config"] = {"ms_per_step": round(di_ms, 3), "value": round(one, 1), "unit": "k-mers/s",
                                            "what": f"rank 0's {NR} reads through cblx_insert_seqs_device on its own GPU, no exchange, {args.steps} steps after the timed region"}
            extra["scaling_vs_one_gpu_same_config"] = round(value / one, 3)
        extra["sharded_overhead"] = {
            "direct_ms": round(di_ms, 3), "sharded_ms": round(sh_ms, 3), "ratio": round(sh_ms / di_ms, 4),
            "direct_stage_ms": {n: round(ms / args.steps, 3) for n, (ms, _) in d_stages.items() if ms > 0},
            "sharded_stage_ms": {n: round(ms / args.steps, 3) for n, (ms, _) in stages.items() if ms > 0},
            "note": "rank 0, same reads: cblx_insert_seqs_device (direct) against the sharded insert; at N = 1 the exchange is a 1-rank group, "
                    "so sharded - direct = slices, bins / batches, receive arena, piece tables"}
        fence()

    # SURVEY.md §8d "also with serialization": the index bytes of this rank's share — size pass, then into a host buffer
    if args.kind == "build" and not args.no_serialize:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nbytes = cbl.serialized_size()
        t2 = time.perf_counter()
        blob = cbl.serialize_np()
        t3 = time.perf_counter()
        cbl.serialize_np(out=blob)  # the same buffer again: its pages are resident now
        t4 = time.perf_counter()
        extra["serialize"] = {"bytes": int(nbytes), "size_ms": round((t2 - t1) * 1e3, 3), "to_host_ms": round((t3 - t2) * 1e3, 3),
                              "to_host_again_ms": round((t4 - t3) * 1e3, 3), "gbps_to_host": round(nbytes / max(t3 - t2, 1e-9) / 1e9, 2),
                              "gbps_to_host_again": round(nbytes / max(t4 - t3, 1e-9) / 1e9, 2),
                              "note": "rank 0's share: device emitter (kernels_serde.hpp; chunks of buckets are downloaded while the next chunk is emitted) + pinned "
                                      "download lanes into a pageable host buffer; to_host_ms writes a FRESH buffer (first touch of every page), to_host_again_ms the "
                                      "same buffer once more; load_ms = cblx_load of those bytes into a fresh context, the first load of the process (host threads parse, the suffixes go up as they are decoded, the arena is allocated by a helper thread meanwhile); not part of `value`"}
        # the same bytes back into a fresh context (`cbl insert` / `merge` / `query` start with this)
        try:
            import cbl_amd as _ca

            h = _ca.CBL(K, PB, canonical=bool(args.canonical), device=local_rank)
            torch.cuda.synchronize()
            t5 = time.perf_counter()
            h.load(blob)
            n_loaded = h.count()
            t6 = time.perf_counter()
            h.close()
            extra["serialize"].update({"load_ms": round((t6 - t5) * 1e3, 3), "gbps_load": round(nbytes / max(t6 - t5, 1e-9) / 1e9, 2), "loaded_kmers": int(n_loaded)})
        except Exception as e:
            extra["serialize"]["load_error"] = f"{type(e).__name__}: {e}"
        del blob

    exchange = None
    if dist is not None and engine is not None and hasattr(engine, "stats"):
        st = engine.stats
        vec = torch.tensor([st["sent_bytes"], st["recv_bytes"]], dtype=torch.int64, device=dev)
        parts = [torch.zeros_like(vec) for _ in range(world)]
        dist.all_gather(parts, vec)
        parts = [p.cpu().tolist() for p in parts]
        out_s = allreduce_max(st["outstanding_s"])
        sent = [p[0] // args.steps for p in parts]
        recv = [p[1] // args.steps for p in parts]
        peers = max(world - 1, 1)
        exchange = {"world_size": dist.get_world_size(), "transport": transport,
                    # what crossed the links: the library's own report where it chose ("auto": sorted on 2 - 4 ranks, bins otherwise)
                    "protocol": (comm.protocol_used() if (args.kind == "build" and comm is not None) else args.protocol_resolved), "protocol_asked": args.protocol,
                    "slice_weights": (list(sharded.ShardedBuilder.GROUPED_WEIGHTS) if (args.kind == "build" and grouped) else None),
                    # groups rank 0's receiver worked its range off in while the later ones were still on the wire (0: the ungrouped
                    # receiver — everything waits for the last record; DESIGN_HISTORY.md §5.6)
                    "recv_groups_used": (comm.groups_used() if (args.kind == "build" and comm is not None) else None),
                    "sent_bytes_per_rank_step": sent, "recv_bytes_per_rank_step": recv,
                    "outstanding_ms_per_step": round(out_s / args.steps * 1e3, 3), "wait_ms_per_step": round(allreduce_max(st["wait_s"]) / args.steps * 1e3, 3),
                    # bytes one rank pushes to ONE peer / the time its exchanges were in flight (they overlap the kernels of the next slice)
                    "effective_gbps_per_link": round(max(sent) / peers / max(out_s / args.steps, 1e-9) / 1e9, 2) if out_s > 0 else None}

    # PCIe-inclusive figure ("from bases resident in pinned host memory"): never `value`
    h2d = None
    if args.kind == "build" and world == 1 and dist is None and not args.no_h2d:
        import numpy as np

        hb = torch.empty(NR * L, dtype=torch.uint8, pin_memory=True)
        ho = torch.empty(NR + 1, dtype=torch.int64, pin_memory=True)
        hb.copy_(d_bases[: NR * L])
        ho.copy_(d_offsets)
        torch.cuda.synchronize()
        nb_, no_ = hb.numpy(), ho.numpy().view(np.uint64)
        ts = []
        for _ in range(3):
            cbl.clear()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            cbl.insert_seqs(nb_, no_)
            cbl.flush()
            ts.append(time.perf_counter() - t1)
        # the wire alone: the same pinned bytes copied to the device and nothing else (best of 3)
        wire = []
        sink_b, sink_o = torch.empty_like(d_bases[: NR * L]), torch.empty_like(d_offsets)
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sink_b.copy_(hb, non_blocking=True)
            sink_o.copy_(ho, non_blocking=True)
            torch.cuda.synchronize()
            wire.append(time.perf_counter() - t1)
        del sink_b, sink_o
        # the same from PAGEABLE host memory (a plain numpy copy of the pinned buffer)
        pb_ = np.array(nb_, copy=True)
        po_ = np.array(no_, copy=True)
        tp = []
        for _ in range(2):
            cbl.clear()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            cbl.insert_seqs(pb_, po_)
            cbl.flush()
            tp.append(time.perf_counter() - t1)
        del pb_, po_
        best, w = min(ts[1:]), min(wire)
        packed = os.environ.get("CBLX_H2D_PACK", "") != "0" and (os.cpu_count() or 1) >= 16
        plane_bytes = 4  # per 16 bases: a dword of code planes; the synthetic reads hold no invalid base, so no validity word is sent
        res_ms = dt / args.steps * 1e3
        h2d = {"value": round(kmers_per_rank / best, 1), "unit": "k-mers/s", "ms_per_step": round(best * 1e3, 3),
               "ms_per_step_pageable": round(min(tp) * 1e3, 3),
               "kernels_ms_resident": round(res_ms, 3), "exposed_transfer_ms": round(best * 1e3 - res_ms, 3),
               "wire_ms": round((-(-hb.numel() // 16) * plane_bytes + ho.numel() * 8 if packed else hb.numel() + ho.numel() * 8) / ((hb.numel() + ho.numel() * 8) / w) * 1e3, 3),
               "ascii_copy_ms": round(w * 1e3, 3), "ascii_copy_gbps": round((hb.numel() + ho.numel() * 8) / w / 1e9, 2),
               "bytes_on_the_wire": int(-(-hb.numel() // 16) * plane_bytes + ho.numel() * 8) if packed else int(hb.numel() + ho.numel() * 8),
               "mode": ("bit planes: host threads pack 2 code bits + 1 validity bit per base; the validity plane of a transfer unit without invalid bases "
                        "(all of them here) is filled in on the device instead of sent; units (1, 3, 4, 4, 3, 1 sixteenths) are copied as they are packed, "
                        "the sliced insert runs right behind them"
                        if packed else "ASCII bytes in slices that land front to back, KRN-1 + the first pass of a slice under the wire"),
               "source": "pinned host memory (torch pin_memory) -> cblx_insert_seqs + cblx_flush, best of 2 after 1 warm-up; ascii_copy_ms = one plain copy of the "
                         "same pinned ASCII bytes (what the wire alone would cost if the bases crossed it as they are); wire_ms = bytes_on_the_wire at that rate; "
                         "exposed_transfer_ms = ms_per_step - the step on resident data (packing, wire and slicing that the kernels do not hide)"}
        # the same reads as a single-line FASTA file (tmpfs) -> cblx_insert_fastx_file + flush: never `value`
        if not args.no_fasta:
            extra["fasta_inclusive"] = fasta_leg(cbl, nb_, NR, L, kmers_per_rank)
        # the reference's call granularity — one insert_seq per record: never `value`
        if not args.no_per_record and not args.genome:
            extra["per_record"] = per_record_leg(NR, K, PB, L)
        del hb, ho

    cpu = cpu_early
    if args.kind == "merge" and args.cpu_full and world == 1 and dist is None and not args.no_cpu_baseline:
        extra["parity_full_size"], cpu = merge_parity_full_size(args, work, B, count_b)
    if cpu is None and rank == 0 and not args.no_cpu_baseline and args.kind == "build" and args.genome:
        cpu = cpu_baseline_leg(args, lambda done, m: (d_bases[done * L:(done + m) * L].cpu().numpy(), None))  # the same reads the GPU got

    out = None
    if rank == 0:
        wb = word_layout(K, PB)[1]
        if args.kind == "build":
            workload = (f"{args.config}: K={K} ({wb}-bit word) PREFIX_BITS={PB} {NR}x{L}bp reads per GPU, "
                        + (f"{NR * L // args.genome}x coverage of a {args.genome} bp genome, " if args.genome else "")
                        + f"{'canonical' if args.canonical else 'non-canonical'}, build from empty index")
            par = "1 GPU" if world == 1 else f"{world} GPUs: read-sharded encode + partition, prefix-range exchange ({args.protocol_resolved} protocol), per-range bucket insert"
        else:
            workload = (f"merge (cfg 5 per-GPU share): K={K} ({wb}-bit word) PREFIX_BITS={PB}, A |= B with A, B = indexes of {NR}x{L}bp reads per GPU each "
                        f"(seeds 42 / 43); a step = " + ("device copy of A into a work index + the merge" if args.merge_clone or world > 1 else "cblx_merge_from(work, A, B): work = A | B as `A |= B` would leave A, A untouched (no copy)") + "; value = k-mers of B merged per second")
            par = "1 GPU" if world == 1 else f"{world} GPUs: both operands prefix-range sharded, B re-sharded to A's bounds, per-rank merge"
        out = {
            "metric": "k-mers inserted/sec (build index)" if args.kind == "build" else "k-mers inserted/sec (merge: self |= other)",
            "value": round(value, 1), "unit": "k-mers/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
            "data": (f"synthetic (reads = windows at uniform random positions of one iid ACGT genome of {args.genome} bp, resident in HBM)" if args.genome
                     else "synthetic (iid ACGT reads, splitmix64 seed 42, resident in HBM)"),
            "config": {"workload": workload, "name": args.config, "k": K, "prefix_bits": PB, "reads_per_gpu": NR, "read_len": L, "parallelism": par},
            "distinct_kmers_in_index": count,
            "value_h2d_inclusive": h2d["value"] if h2d else None, "h2d_inclusive": h2d,
            "roofline": roofline, "cpu_baseline": cpu, "exchange": exchange,
        }
        if roofline_error:
            out["roofline_error"] = roofline_error
        out.update(extra)
    if dist is not None:
        dist.barrier()
        tdist.destroy_process_group()
    if out is not None:  # last thing on stdout, on a line of its own
        sys.stdout.flush()
        sys.stdout.write("\n" + json.dumps(out) + "\n")
        sys.stdout.flush()
    bad = roofline_error is not None or (extra.get("parity_sample") or {}).get("equal") is False or (extra.get("set_check") or {}).get("ok") is False
    if bad:  # the measurements are on stdout; a pricing slip, a differing byte or a failed set check still fails the run
        sys.exit(3)


if __name__ == "__main__":
    main()
