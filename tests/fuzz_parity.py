"""Randomised differential test: HIP path (through the C ABI) vs the CPU oracle on random configurations and operation
sequences (insert_seq / insert_seqs / incremental flushes / |= / serialize-load round trips / sorted batches). Not part of
the default suites (minutes of run time); run on a GPU box:  python tests/fuzz_parity.py [--cases 150] [--seed 1]"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cbl_amd
from oracle import Oracle


def rand_seq(rng, n, alphabet):
    return bytes(rng.choice(alphabet) for _ in range(n))


def main():
    os.environ["CBLX_QUERY_JOIN_MIN"] = "1"  # tallies-only queries take the join path whatever their size
    os.environ.setdefault("CBLX_FINE_MIN", "0")  # PREFIX_BITS > 24: the first batch into an empty index takes the FINE-bins build whatever its size
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    for case in range(a.cases):
        k = rng.choice([5, 7, 9, 11, 13, 15, 21, 25, 27, 29, 31, 33, 35, 45, 59])
        wb = 2 * k + (2 * k - 1).bit_length()
        pb = rng.randint(1, min(28, wb - 1, 2 * k))
        if wb - pb > 128:
            continue
        canonical = rng.random() < 0.3
        alphabet = b"ACGT" if rng.random() < 0.7 else b"ACGTacgtN"
        if rng.random() < 0.2:
            alphabet = b"AAAC"  # low complexity: heavy duplication, long runs
        try:
            g, o = cbl_amd.CBL(k, pb, canonical=canonical), Oracle(k, pb, canonical)
        except Exception as e:  # parameter combinations both sides refuse
            continue
        desc = f"case {case}: k={k} pb={pb} canonical={canonical} alphabet={alphabet!r}"
        ops = []
        print("START", desc, file=sys.stderr, flush=True)
        for step in range(rng.randint(1, 5)):
            op = rng.choice(["seq", "seqs", "seqs", "merge", "roundtrip", "sorted", "kmers", "query", "file", "shards", "repeats"])
            ops.append(op)
            prev_blob = o.serialize() if os.environ.get("CBLX_FUZZ_DIAG") else None
            try:
                if op == "seq":
                    for _ in range(rng.randint(1, 5)):
                        s = rand_seq(rng, rng.randint(k, k + rng.choice([0, 1, 5, 300, 5000])), alphabet)
                        g.insert_seq(s), o.insert_seq(s)
                elif op == "seqs":
                    seqs = [rand_seq(rng, rng.randint(k, k + rng.choice([0, 3, 100, 2500])), alphabet) for _ in range(rng.randint(1, 60))]
                    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
                    offsets = np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64)
                    g.insert_seqs(bases, offsets), o.insert_seqs(bases, offsets)
                    if rng.random() < 0.5:
                        g.flush()
                elif op == "repeats":  # one batch at high coverage of a short "genome": runs full of repeats (claim-table kernels)
                    genome = rand_seq(rng, rng.randint(k + 50, k + rng.choice([200, 1500, 6000])), alphabet)
                    rl = rng.randint(k, min(len(genome), k + 120))
                    seqs = []
                    for _ in range(len(genome) * rng.choice([8, 30, 100]) // rl):
                        p0 = rng.randrange(0, len(genome) - rl + 1)
                        seqs.append(genome[p0:p0 + rl])
                    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
                    offsets = np.cumsum([0] + [len(s_) for s_ in seqs]).astype(np.uint64)
                    g.insert_seqs(bases, offsets), o.insert_seqs(bases, offsets)
                elif op == "merge":
                    g2, o2 = cbl_amd.CBL(k, pb, canonical=canonical), Oracle(k, pb, canonical)
                    for _ in range(rng.randint(1, 20)):
                        s = rand_seq(rng, rng.randint(k, k + 2000), alphabet)
                        g2.insert_seq(s), o2.insert_seq(s)
                    g |= g2
                    o.merge(o2)
                    assert g2.serialize() == o2.serialize(), desc + " (other after |=) " + str(ops)
                elif op == "roundtrip":
                    blob = g.serialize()
                    assert blob == o.serialize(), desc + " (before round trip) " + str(ops)
                    g = cbl_amd.CBL(k, pb, canonical=canonical)
                    g.load(blob)
                elif op == "kmers":  # single-k-mer inserts: return values and final state
                    pool = [rng.getrandbits(2 * k) for _ in range(rng.randint(1, 200))]
                    batch = [rng.choice(pool) for _ in range(rng.randint(1, 400))]
                    got = g.insert_kmers(batch).tolist()
                    assert got == [o.insert_kmer(x) for x in batch], desc + " (insert_kmers) " + str(ops)
                    probe = batch[:50] + [rng.getrandbits(2 * k) for _ in range(50)]
                    assert g.contains_kmers(probe).tolist() == [o.contains_kmer(x) for x in probe], desc + " (contains_kmers) " + str(ops)
                elif op == "query":
                    seqs = [rand_seq(rng, rng.randint(k, k + rng.choice([0, 7, 500, 3000])), alphabet) for _ in range(rng.randint(1, 30))]
                    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
                    offsets = np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64)
                    flags, tot, pos = g.contains_seqs(bases, offsets)
                    want = [o.contains_word(w) for s in seqs for w in o.seq_words(s)]
                    assert flags.tolist() == want and tot == len(want) and pos == sum(want), desc + " (query) " + str(ops)
                    assert g.contains_seqs(bases, offsets, flags=False)[1:] == (tot, pos), desc + " (query by join) " + str(ops)
                    if rng.random() < 0.3:
                        assert [o.kmer_of_word(w) for w in o.iter_words()] == list(g.iter()), desc + " (iter) " + str(ops)
                elif op == "file":  # FASTA in (multi-line, mixed case), index file out and back in, the file query loop
                    import tempfile

                    seqs = [rand_seq(rng, rng.randint(k, k + rng.choice([0, 9, 400, 2600])), alphabet) for _ in range(rng.randint(1, 25))]
                    with tempfile.TemporaryDirectory() as td:
                        fa, ix = os.path.join(td, "in.fa"), os.path.join(td, "x.cbl")
                        width = rng.choice([60, 70, 10_000])
                        with open(fa, "wb") as f:
                            for i, sq in enumerate(seqs):
                                f.write(b">r%d some text\n" % i)
                                for j in range(0, len(sq), width):
                                    f.write(sq[j : j + width] + (b"\r\n" if width == 70 else b"\n"))
                        assert g.insert_fastx_file(fa) == len(seqs), desc + " (records) " + str(ops)
                        for sq in seqs:
                            o.insert_seq(sq)
                        g.save_to_file(ix)
                        with open(ix, "rb") as f:
                            assert f.read() == o.serialize(), desc + " (index file) " + str(ops)
                        g = cbl_amd.CBL.load_from_file(ix, k, pb)
                        want = [o.contains_word(w) for sq in seqs for w in o.seq_words(sq)]
                        assert g.query_fastx_file(fa) == (len(seqs), len(want), sum(want)), desc + " (file query) " + str(ops)
                        assert g.contains_all(seqs[0]) == all(o.contains_word(w) for w in o.seq_words(seqs[0])), desc + " (contains_all) " + str(ops)
                    sizes = {}
                    sbits = g.consts()["suffix_bits"]
                    for w in o.iter_words():
                        sizes[w >> sbits] = sizes.get(w >> sbits, 0) + 1
                    assert g.buckets_sizes() == sorted(sizes.items()), desc + " (buckets_sizes) " + str(ops)
                elif op == "shards":  # the index as W prefix-range shares of its file, and as re-cut bucket batches
                    import tempfile

                    from cbl_amd.sharded import _varint

                    blob = g.serialize()
                    assert blob == o.serialize(), desc + " (before shards) " + str(ops)
                    if g.num_buckets():
                        W = rng.randint(1, 5)
                        with tempfile.TemporaryDirectory() as td:
                            ix = os.path.join(td, "x.cbl")
                            with open(ix, "wb") as f:
                                f.write(blob)
                            bounds = None
                            if rng.random() < 0.5 and W > 1:
                                bounds = np.sort(np.array([rng.randrange(1 << pb) for _ in range(W - 1)], dtype=np.uint32))
                            seqm = rng.random() < 0.3
                            shares, ents = [], 0
                            for r in range(W):
                                sh = cbl_amd.CBL(k, pb)
                                info, _b = sh.load_shard_from_file(ix, r, W, bounds, seqm)
                                assert info["exact"] == 1, desc + " (shard not exact) " + str(ops)
                                ents += info["local_entries"]
                                shares.append(sh)
                            assert ents == g.num_buckets(), desc + " (shard entries) " + str(ops)
                            body = b"".join(s_.serialize()[len(s_.serialize()) - s_.serialized_body_size()[1]:] for s_ in shares)
                            assert bytes([blob[0]]) + _varint(ents) + body == blob, desc + " (shares != file) " + str(ops)
                        nb, nw, B = g.num_buckets(), g.count(), g.consts()["bytes"]
                        pfx = torch.empty(nb, dtype=torch.int32, device="cuda")
                        cnt = torch.empty(nb, dtype=torch.int32, device="cuda")
                        knd = torch.empty(nb, dtype=torch.uint8, device="cuda")
                        sfx = torch.empty(nw * B, dtype=torch.uint8, device="cuda")
                        g.resident_export(pfx, cnt, knd, sfx)
                        nd = rng.randint(1, 4)
                        cuts = np.sort(np.array([rng.randrange(1 << pb) for _ in range(nd - 1)], dtype=np.uint32))
                        bs, ws = g.resident_split(cuts, nd)
                        parts = [(bs[d + 1] - bs[d], ws[d + 1] - ws[d], pfx[bs[d]: bs[d + 1]], cnt[bs[d]: bs[d + 1]], knd[bs[d]: bs[d + 1]], sfx[ws[d] * B: ws[d + 1] * B])
                                 for d in range(nd) if bs[d + 1] > bs[d]]
                        g2 = cbl_amd.CBL(k, pb, canonical=canonical)
                        g2.install_buckets_device(parts)
                        assert g2.serialize() == blob, desc + " (export -> install) " + str(ops)
                        g = g2  # carry on with the installed copy
                elif op == "sorted":
                    seqs = [rand_seq(rng, rng.randint(k, k + 800), b"ACGT") for _ in range(rng.randint(1, 40))]
                    hb = np.frombuffer(b"".join(seqs), dtype=np.uint8)
                    ho = np.cumsum([0] + [len(s) for s in seqs]).astype(np.int64)
                    d_b = torch.from_numpy(np.concatenate([hb, np.zeros(32, np.uint8)])).cuda()
                    d_o = torch.from_numpy(ho).cuda()
                    nd = rng.randint(1, 5)
                    bounds = np.sort(np.array([rng.randrange(1 << pb) for _ in range(nd - 1)], dtype=np.uint32))
                    snd = cbl_amd.CBL(k, pb, canonical=canonical)
                    bs, ws = snd.sorted_batch_begin(d_b, d_o, len(seqs), bounds, nd)
                    B = snd.consts()["bytes"]
                    pfx = torch.empty(max(bs[nd], 1), dtype=torch.int32, device="cuda")
                    cnt = torch.empty(max(bs[nd], 1), dtype=torch.int32, device="cuda")
                    sfx = torch.empty(max(ws[nd] * B, 1), dtype=torch.uint8, device="cuda")
                    snd.sorted_batch_export(pfx, cnt, sfx)
                    batches = [(bs[d + 1] - bs[d], ws[d + 1] - ws[d], pfx[bs[d] : bs[d + 1]], cnt[bs[d] : bs[d + 1]], sfx[ws[d] * B : ws[d + 1] * B])
                               for d in range(nd) if ws[d + 1] > ws[d]]
                    g.insert_sorted_batches_device(batches)
                    for s in seqs:
                        o.insert_seq(s)
            except cbl_amd.CblxError as e:  # report what led here, and whether the index was still sound before the failing call
                print("FAIL", desc, ops, "error:", e, file=sys.stderr, flush=True)
                if prev_blob is not None:
                    try:
                        print("  state before the op == oracle's:", g.serialize() == prev_blob, "validate:", g.validate(strict=False), "count", g.count(), file=sys.stderr, flush=True)
                    except Exception as e2:  # noqa: BLE001
                        print("  (state not readable:", e2, ")", file=sys.stderr)
                raise
            assert g.count() == o.count(), desc + f" count after {ops}"
        assert g.serialize() == o.serialize(), desc + " " + str(ops)
        assert g.validate(strict=False) == 0, desc
        if case % 10 == 0:
            print(desc, ops, "ok", flush=True)
    print("fuzz ok")


if __name__ == "__main__":
    main()
