"""Secondary measurements SURVEY.md §8d asks to report beside the headline metric (never `value` of bench.py):
host-buffer (PCIe-inclusive) build rate, serialization, duplicate-heavy reads, `|=` of two indexes (cfg 5 scaled to one
GPU), load + insert on a non-empty index, batched contains_seq. One JSON object on stdout; copy it to profiles/.

    python tools/bench_extras.py [--reads 10000000] [--skip-serialize]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--prefix-bits", type=int, default=24)
    ap.add_argument("--skip-serialize", action="store_true")
    a = ap.parse_args()
    import numpy as np
    import torch

    import __graft_entry__ as ge

    ge.build()
    import cbl_amd
    from cbl_amd import synth

    K, PB, L, NR = a.k, a.prefix_bits, a.read_len, a.reads
    per_read = L - K + 1
    out = {"config": {"k": K, "prefix_bits": PB, "reads": NR, "read_len": L}}
    dev = "cuda:0"

    def timed(f, reps=3, pre=None):
        best = None
        for _ in range(reps):
            if pre:
                pre()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            f()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best

    d_bases, d_offsets = synth.reads_torch(42, NR, L, device=dev)
    torch.cuda.synchronize()
    g = cbl_amd.CBL(K, PB, device=0)

    # 1. device-resident build (the headline path, for reference inside this file)
    dt = timed(lambda: g.insert_seqs_device(d_bases, d_offsets, NR), pre=g.clear)
    out["build_device_resident"] = {"ms": dt * 1e3, "kmers_per_s": NR * per_read / dt}

    # 2. host buffers through cblx_insert_seqs (PCIe-inclusive)
    h_bases = d_bases.cpu().numpy()
    h_offsets = d_offsets.cpu().numpy().astype(np.uint64)
    dt = timed(lambda: (g.insert_seqs(h_bases, h_offsets), g.flush()), pre=g.clear)
    out["build_host_buffers"] = {"ms": dt * 1e3, "kmers_per_s": NR * per_read / dt, "note": "pageable numpy buffers -> cblx_insert_seqs + flush"}

    # 2b. FASTA file -> index (examples/cbl.rs build): single-line FASTA on tmpfs, parse + PCIe + insert
    fa = "/dev/shm/cblx_extras.fa"
    with open(fa, "wb") as f:
        step = 1_000_000
        for a0 in range(0, NR, step):
            n = min(step, NR - a0)
            rec = np.empty((n, 11 + L + 1), dtype=np.uint8)  # ">r%08d\n" + bases + "\n"
            rec[:, 0], rec[:, 1], rec[:, 10], rec[:, -1] = ord(">"), ord("r"), 10, 10
            ids = np.arange(a0, a0 + n)
            for d in range(8):
                rec[:, 9 - d] = 48 + (ids // 10**d) % 10
            rec[:, 11 : 11 + L] = h_bases[a0 * L : (a0 + n) * L].reshape(n, L)
            f.write(rec.tobytes())
    fsize = os.path.getsize(fa)
    def _file():
        assert g.insert_fastx_file(fa) == NR
        g.flush()
    dt = timed(_file, reps=2, pre=g.clear)
    os.remove(fa)
    out["build_from_fasta_file"] = {"ms": dt * 1e3, "kmers_per_s": NR * per_read / dt, "file_bytes": fsize, "file_GB_per_s": fsize / dt / 1e9}

    # 3. serialization of the resident index (host, multi-threaded) -- Appendix A bytes
    if not a.skip_serialize:
        t0 = time.perf_counter()
        nbytes = g.serialized_size()
        dt_size = time.perf_counter() - t0
        buf = np.empty(nbytes, dtype=np.uint8)
        buf[::4096] = 0  # touch the pages: the caller's buffer exists before the call
        t0 = time.perf_counter()
        blob = g.serialize_np(buf)
        dt = time.perf_counter() - t0
        path = "/dev/shm/cblx_extras.cbl"
        t0 = time.perf_counter()
        g.save_to_file(path)
        dt_file = time.perf_counter() - t0
        os.remove(path)
        out["serialize"] = {"size_pass_ms": dt_size * 1e3, "to_host_buffer_ms": dt * 1e3, "save_to_file_tmpfs_ms": dt_file * 1e3, "bytes": len(blob),
                            "GB_per_s": len(blob) / dt / 1e9, "words": g.count()}
        blob = blob.tobytes()
        # 6. load + insert on a non-empty index (cbl insert): parse + upload, then the device path with resident words
        g2 = cbl_amd.CBL(K, PB, device=0)
        t0 = time.perf_counter()
        g2.load(blob)
        torch.cuda.synchronize()
        dt_load = time.perf_counter() - t0
        del blob
        nr2 = max(NR // 10, 1)
        e_bases, e_offsets = synth.reads_torch(43, nr2, L, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g2.insert_seqs_device(e_bases, e_offsets, nr2)
        torch.cuda.synchronize()
        dt_ins = time.perf_counter() - t0
        out["load_then_insert"] = {"load_ms": dt_load * 1e3, "insert_ms": dt_ins * 1e3, "new_reads": nr2, "resident_words": g.count(),
                                   "words_after": g2.count()}
        g2.close()
        del e_bases, e_offsets

    # 5. contains_seq on one long sequence (first 2 M reads' bases as one sequence)
    q = h_bases[: min(NR, 2_000_000) * L].tobytes()
    t0 = time.perf_counter()
    hits = g.contains_seq_np(q)
    dt = time.perf_counter() - t0
    out["contains_seq"] = {"ms": dt * 1e3, "kmers": len(hits), "kmers_per_s": len(hits) / dt, "hit_fraction": float(np.mean(hits))}
    # 5b. batched query (cbl query): all reads of the indexed set (hits) and an unrelated read set (misses), device-resident
    dt = timed(lambda: g.contains_seqs_device(d_bases, d_offsets, NR), reps=2)
    tot, pos = g.contains_seqs_device(d_bases, d_offsets, NR)
    out["query_batched_hits"] = {"ms": dt * 1e3, "kmers": tot, "positive": pos, "kmers_per_s": tot / dt}
    m_bases, m_offsets = synth.reads_torch(77, NR, L, device=dev)
    torch.cuda.synchronize()
    dt = timed(lambda: g.contains_seqs_device(m_bases, m_offsets, NR), reps=2)
    tot, pos = g.contains_seqs_device(m_bases, m_offsets, NR)
    out["query_batched_misses"] = {"ms": dt * 1e3, "kmers": tot, "positive": pos, "kmers_per_s": tot / dt}
    d_flags = torch.zeros(NR * per_read + 8, dtype=torch.uint8, device=dev)
    dt = timed(lambda: g.contains_seqs_device(d_bases, d_offsets, NR, d_flags, NR * per_read), reps=2)
    out["query_batched_hits_with_flags"] = {"ms": dt * 1e3, "kmers_per_s": NR * per_read / dt, "flags_set": int(d_flags.sum(dtype=torch.int64))}
    dt = timed(lambda: g.contains_seqs_device(m_bases, m_offsets, NR, d_flags, NR * per_read), reps=2)
    out["query_batched_misses_with_flags"] = {"ms": dt * 1e3, "kmers_per_s": NR * per_read / dt, "flags_set": int(d_flags.sum(dtype=torch.int64))}
    del m_bases, m_offsets, d_flags
    g.clear()
    del h_bases, h_offsets

    # 4. duplicate-heavy: every read emitted twice (NR/2 distinct reads)
    half = NR // 2
    db = d_bases[: half * L]
    dup_bases = torch.cat([db, db])
    dup_offsets = torch.arange(0, 2 * half + 1, dtype=torch.int64, device=dev) * L
    dt = timed(lambda: g.insert_seqs_device(dup_bases, dup_offsets, 2 * half), pre=g.clear)
    out["build_duplicate_heavy"] = {"ms": dt * 1e3, "kmers_per_s": 2 * half * per_read / dt, "distinct_words": g.count()}
    del dup_bases, dup_offsets
    g.clear()

    # 7. self |= other: two independent read sets of NR/2 reads each (seeds 42, 43), cfg 5 scaled to one GPU
    o_bases, o_offsets = synth.reads_torch(43, half, L, device=dev)
    a_off = d_offsets[: half + 1]
    ga, gb = cbl_amd.CBL(K, PB, device=0), cbl_amd.CBL(K, PB, device=0)
    best = None
    for _ in range(2):
        ga.clear(), gb.clear()
        ga.insert_seqs_device(d_bases[: half * L], a_off, half)
        gb.insert_seqs_device(o_bases, o_offsets, half)
        ga.flush(), gb.flush()
        na, nb = ga.count(), gb.count()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ga |= gb
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    nout = ga.count()
    C = g.consts()
    byts = C["bytes"]
    out["merge_assign"] = {"ms": best * 1e3, "self_words": na, "other_words": nb, "out_words": nout, "out_words_per_s": nout / best,
                           "alg_GB_per_s": nout * 3 * byts / best / 1e9, "alg_bytes_per_out_word": 3 * byts}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
