"""The multi-GPU build on a REAL multi-rank RCCL group: one process per GPU, torch.distributed nccl backend (and the native
cblx_comm on RCCL), byte-identical to the one-process oracle. Needs >= 2 GPUs in one box: on the one-GPU pool this repo was
built on these tests SKIP (RCCL refuses two ranks on one GPU) — the same workers run there with the exchange staged through
gloo (tests/test_gpu_parity.py: *_through_a_gloo_shim, *_through_callbacks). DESIGN_HISTORY.md §5 states the multi-GPU path as
unverified on hardware until these have run."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import cbl_amd  # noqa: E402
from cbl_amd import synth  # noqa: E402
from oracle import Oracle  # noqa: E402


def _ngpu():
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, k, pb, canonical, protocol, native, per, L, q, groups=0):
    import torch.distributed as dist

    from cbl_amd import sharded

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    sharded.MAX_MSG_BYTES = 1 << 16  # force message splitting
    try:
        comm = None
        if native:
            box = [cbl_amd.Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = cbl_amd.Comm.rccl(box[0], rank, world, rank)
            comm.set_recv_groups(groups)  # 0: the default (grouped receiver, 4 groups per rank), 1: everything waits for the last record
        g = cbl_amd.CBL(k, pb, canonical=canonical, device=rank)
        sb = sharded.ShardedBuilder(g, dist, slices=3, protocol=protocol, comm=comm)
        for batch, n in enumerate(per[rank]):
            first = sum(per[r][bb] for r in range(world) for bb in range(batch)) + sum(per[r][batch] for r in range(rank))
            d_b, d_o = synth.reads_torch(23, n, L, first_read=first, device=f"cuda:{rank}")
            sb.insert_seqs_device(d_b, d_o, n)
        blob = sharded.gather_serialized(g.serialize(), dist)
        # cfg 5: a second index at other bounds, re-shard + merge
        A = sharded.ShardedIndex(k, pb, dist, canonical=canonical, device=rank, slices=2)
        B = sharded.ShardedIndex(k, pb, dist, canonical=canonical, device=rank, slices=2)
        a_b, a_o = synth.reads_torch(31, 400, L, first_read=rank * 400, device=f"cuda:{rank}")
        A.insert_seqs_device(a_b, a_o, 400)
        nb = np.asarray(A.bounds, dtype=np.uint64) * 3 // 2 + 1
        B.bounds = np.minimum(nb, (1 << pb) - 1).astype(np.uint32)
        b_b, b_o = synth.reads_torch(77, 400, L, first_read=rank * 400, device=f"cuda:{rank}")
        B.insert_seqs_device(b_b, b_o, 400)
        A.merge_assign(B)
        mblob = sharded.gather_serialized(A.cbl.serialize(), dist)
        if rank == 0:
            q.put((blob, mblob, sb.stats["sent_bytes"]))
        if comm is not None:
            comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,k,pb,canonical,protocol,native,groups", [
    (2, 31, 24, False, "sorted", False, 0), (2, 31, 24, True, "words", False, 0), (2, 59, 28, False, "sorted", True, 0),
    (4, 31, 24, False, "sorted", True, 0), (8, 31, 28, False, "sorted", False, 0),
    (2, 31, 24, False, "bins", True, 0), (4, 59, 28, True, "bins", True, 0), (8, 31, 28, False, "bins", True, 0),
    (2, 31, 24, False, "bins", True, 1), (8, 31, 28, False, "bins", True, 1), (4, 31, 24, True, "bins", True, 3),
    (2, 31, 28, False, "auto", True, 0), (8, 31, 26, True, "auto", True, 0),  # the library's choice ("replicate" on 2 - 3 ranks, "sorted" on 4); FINE bins at PREFIX_BITS > 24 on 8
    # "replicate" (round 6): the reads cross as bit planes (one grouped exchange of planes and offsets), every rank transforms all of them
    (2, 31, 24, False, "replicate", True, 0), (3, 59, 28, True, "replicate", True, 0), (4, 31, 28, False, "replicate", True, 3), (8, 31, 26, False, "replicate", True, 0)])
def test_sharded_build_and_merge_on_real_rccl(world, k, pb, canonical, protocol, native, groups):
    if _ngpu() < world:
        pytest.skip(f"needs {world} GPUs in one box, {_ngpu()} visible (RCCL refuses two ranks on one GPU)")
    import torch.multiprocessing as mp

    from cbl_amd.sharded import ShardedBuilder

    L = 150 if k < 59 else 250
    per = [(700, 2), (300, 450), (1, 600), (512, 0), (64, 64), (0, 900), (333, 5), (90, 90)][:world]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, pb, canonical, protocol, native, per, L, q, groups)) for r in range(world)]
    for p in procs:
        p.start()
    blob, mblob, sent = q.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    one = Oracle(k, pb, canonical)
    for batch in range(2):
        first = sum(per[r][bb] for r in range(world) for bb in range(batch))
        starts = [first + sum(per[rr][batch] for rr in range(r)) for r in range(world)]
        sl = [ShardedBuilder.slice_bounds(per[r][batch], 3) for r in range(world)]
        for c in range(3):
            for r in range(world):
                a, b = sl[r][c]
                if b > a:
                    hb, ho = synth.reads(23, b - a, L, first_read=starts[r] + a)
                    one.insert_seqs(hb, ho)
    assert blob == one.serialize() and sent > 0

    def one_process(seed):
        o = Oracle(k, pb, canonical)
        for a, b in ShardedBuilder.slice_bounds(400, 2):
            for r in range(world):
                hb, ho = synth.reads(seed, b - a, L, first_read=r * 400 + a)
                o.insert_seqs(hb, ho)
        return o

    oa, ob = one_process(31), one_process(77)
    oa.merge(ob)
    assert mblob == oa.serialize()
