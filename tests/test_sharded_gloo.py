"""N > 1 path on CPU: world_size-2 (and 3) `gloo` runs of cbl_amd.sharded.ShardedBuilder — the orchestration the GPU
ranks execute (splitter choice, stable partition by destination, count exchange, all_to_all_single, source-rank
ordered insert, rank-ordered serialization) — with the device steps stood in by the CPU oracle (test-only engine).
The sharded result must be byte-identical to the one-process build of the same reads in the same order."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


from shard_standin import OracleEngine  # noqa: E402  (CPU stand-in for cbl_amd.sharded.GpuEngine)


def _worker(rank, world, port, k, pb, nreads, L, slices, protocol, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    from cbl_amd import sharded, synth
    from oracle import Oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    if slices == 3:
        sharded.MAX_MSG_BYTES = 64  # force the multi-message path of the exchange (8 words per message)
    try:
        per = nreads // world
        bases, offsets = synth.reads(7, per, L, first_read=rank * per)
        orc = Oracle(k, pb)

        class _Cbl:  # the two attributes ShardedBuilder reads from a CBL
            prefix_bits = pb

        sb = sharded.ShardedBuilder(_Cbl(), dist, engine=OracleEngine(orc, k, pb), slices=slices, protocol=protocol)
        assert sb.protocol == protocol
        # two batches: the second reuses the first batch's splitters
        h = per // 2
        for a, b in ((0, h), (h, per)):
            bb = torch.from_numpy(bases[a * L : b * L].copy())
            oo = torch.from_numpy((offsets[a : b + 1] - offsets[a]).astype(np.int64))
            sb.insert_seqs_device(bb, oo, b - a)
        blob = sharded.gather_serialized(orc.serialize(), dist)
        path = os.path.join(os.environ.get("CBLX_TEST_TMP", "/tmp"), "cblx_sharded_%d.cbl" % port)
        size = sharded.save_serialized(orc.serialize(), dist, path)  # every rank writes its entries in place
        if rank == 0:
            with open(path, "rb") as f:
                on_disk = f.read()
            os.remove(path)
            assert size == len(on_disk) and on_disk == blob
            q.put((blob, sb.bounds.tolist(), sb.last_counts))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("protocol", ["sorted", "words"])
@pytest.mark.parametrize("world,k,pb,nreads,L,slices", [(2, 31, 24, 240, 150, 1), (2, 31, 24, 240, 150, 3), (3, 9, 4, 600, 100, 4),
                                                        (2, 59, 28, 120, 250, 2)])
def test_sharded_build_equals_single_process(world, k, pb, nreads, L, slices, protocol):
    import torch.multiprocessing as mp

    from cbl_amd import synth
    from oracle import Oracle

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, pb, nreads, L, slices, protocol, q)) for r in range(world)]
    for p in procs:
        p.start()
    blob, bounds, counts = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # one-process build. Stream order of the sharded job: per batch, slice-major then rank-minor.
    from cbl_amd.sharded import ShardedBuilder

    per = nreads // world
    h = per // 2
    one = Oracle(k, pb)
    for a, b in ((0, h), (h, per)):
        for sa, sb_ in ShardedBuilder.slice_bounds(b - a, slices):
            for r in range(world):
                bases, offsets = synth.reads(7, sb_ - sa, L, first_read=r * per + a + sa)
                one.insert_seqs(bases, offsets)
    assert blob == one.serialize()
    assert len(bounds) == world - 1 and bounds == sorted(bounds)
    assert sum(counts[0]) > 0


def test_choose_bounds_balances_skewed_histogram():
    from cbl_amd.sharded import choose_bounds

    hist = np.zeros(1 << 16, dtype=np.int64)
    x = np.arange(1 << 16) / (1 << 16)
    hist[:] = (1e6 * 62 * (1 - x) ** 61).astype(np.int64)  # SURVEY.md F6 density
    b = choose_bounds(hist, 8, 24, 16)
    assert len(b) == 7 and all(b[i] <= b[i + 1] for i in range(6))
    cells = (b >> 8).astype(np.int64)
    cum = np.cumsum(hist)
    parts = np.diff(np.concatenate([[0], cum[cells - 1], [cum[-1]]]))
    assert parts.max() / parts.mean() < 1.05
    assert (choose_bounds(np.zeros(16, dtype=np.int64), 4, 4, 4) == [4, 8, 12]).all()


def _worker_ragged(rank, world, port, k, pb, L, counts, protocol, q):
    """Ranks with very different read counts, fewer reads than slices, and a batch in which one rank has nothing."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist

    from cbl_amd import sharded, synth
    from oracle import Oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = Oracle(k, pb)

        class _Cbl:
            prefix_bits = pb

        sb = sharded.ShardedBuilder(_Cbl(), dist, engine=OracleEngine(orc, k, pb), slices=4, protocol=protocol)
        first = 0
        for batch in counts:  # batch = reads per rank
            n = batch[rank]
            bases, offsets = synth.reads(11, n, L, first_read=first + sum(batch[:rank]))
            sb.insert_seqs_device(torch.from_numpy(bases.copy()), torch.from_numpy(offsets.astype(np.int64)), n)
            first += sum(batch)
        blob = sharded.gather_serialized(orc.serialize(), dist)
        if rank == 0:
            q.put(blob)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("protocol", ["sorted", "words"])
def test_sharded_build_with_ragged_and_empty_shards(protocol):
    import torch.multiprocessing as mp

    from cbl_amd import synth
    from cbl_amd.sharded import ShardedBuilder
    from oracle import Oracle

    k, pb, L, world = 31, 24, 120, 2
    counts = [(9, 2), (5, 0), (0, 3)]  # per batch: reads of rank 0, rank 1 (fewer than the 4 slices; none at all)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged, args=(r, world, port, k, pb, L, counts, protocol, q)) for r in range(world)]
    for p in procs:
        p.start()
    blob = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    one = Oracle(k, pb)  # stream order: per batch, slice-major then rank-minor
    first = 0
    for batch in counts:
        starts = [first + sum(batch[:r]) for r in range(world)]
        sl = [ShardedBuilder.slice_bounds(batch[r], 4) for r in range(world)]
        for c in range(4):
            for r in range(world):
                a, b = sl[r][c]
                if b > a:
                    bases, offsets = synth.reads(11, b - a, L, first_read=starts[r] + a)
                    one.insert_seqs(bases, offsets)
        first += sum(batch)
    assert blob == one.serialize()


# ---- one FASTA file, all ranks: block-cyclic dealing gives the stream order of the file ---------------------------------
def _write_fasta(path, seed, nrec):
    """ragged records (some multi-line, CRLF here and there), deterministic"""
    import random

    rng = random.Random(seed)
    recs = []
    with open(path, "wb") as f:
        for i in range(nrec):
            n = rng.choice([40, 64, 150, 151, 300, 777, 2500])
            s = bytes(rng.choice(b"ACGT") for _ in range(n))
            recs.append(s)
            f.write(b">r%d some text\n" % i)
            w = rng.choice([60, 80, 10_000])
            for a in range(0, n, w):
                f.write(s[a : a + w] + (b"\r\n" if i % 7 == 3 else b"\n"))
    return recs


def _worker_file(rank, world, port, k, pb, path, block, protocol, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from cbl_amd import sharded
    from oracle import Oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = Oracle(k, pb)

        class _Cbl:
            prefix_bits = pb

        sb = sharded.ShardedBuilder(_Cbl(), dist, engine=OracleEngine(orc, k, pb), slices=3, protocol=protocol)
        n = sb.insert_fastx_file(path, block)
        n2 = sb.insert_fastx_file(path, 0)  # again (nothing new), block size chosen by the builder
        blob = sharded.gather_serialized(orc.serialize(), dist)
        if rank == 0:
            q.put((blob, n, n2))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,pb,nrec,block,protocol", [(2, 31, 24, 57, 5, "sorted"), (3, 15, 8, 40, 1, "words"), (2, 31, 24, 9, 100, "sorted"), (3, 25, 12, 2, 1, "sorted")])
def test_sharded_build_from_one_file_has_file_order(world, k, pb, nrec, block, protocol, tmp_path):
    import torch.multiprocessing as mp

    from oracle import Oracle

    path = str(tmp_path / "reads.fa")
    recs = _write_fasta(path, 5 + nrec, nrec)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_file, args=(r, world, port, k, pb, path, block, protocol, q)) for r in range(world)]
    for p in procs:
        p.start()
    blob, n, n2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert n == n2 == nrec
    one = Oracle(k, pb)
    for s in recs:  # FILE order, one insert_seq per record: what `cbl build` does
        one.insert_seq(s)
    assert blob == one.serialize()
