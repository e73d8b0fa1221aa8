"""Dev probe: host time per rank of staging ONE FASTA file for a W-rank sharded build, W = 1, 2, 4, 8 ranks sharing this GPU (gloo
callbacks): the parse shared between the ranks (cblx_stage_fastx_blocks_comm: every rank scans 1 / W of the file and reads only
its own blocks) against every rank parsing the whole file (cblx_stage_fastx_blocks). cfg-2 reads, single-line records on tmpfs.
Usage: python tools/dev_fasta_shared.py [reads]"""
import json, os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

NR, L, K = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 150, 31
FA = "/dev/shm/cblx_shared.fa"


def worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    import cbl_amd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = cbl_amd.Comm.over_group(dist, rank, world, 0)
    g = cbl_amd.CBL(K, 24, device=0)
    res = {}
    for name in ("shared", "whole_file"):
        best = 1e9
        for rep in range(2):
            dist.barrier()
            t0 = time.perf_counter()
            if name == "shared":
                _pb, _po, n, n_file, block = g.stage_fastx_blocks_comm(comm, FA, 0, 4)
            else:
                block = -(-NR // (world * 4))
                _pb, _po, n, n_file = g.stage_fastx_blocks(FA, block, rank, world)
            dt = time.perf_counter() - t0
            g.stage_release()
            assert n_file == NR, (n_file, NR)
            best = min(best, dt)
        t = torch.tensor([best], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res[name] = float(t.item())
        cnt = torch.tensor([n], dtype=torch.int64)
        dist.all_reduce(cnt)
        assert int(cnt.item()) == NR
    if rank == 0:
        q.put(res)
    comm.close()
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    from cbl_amd import synth
    h_bases, _ = synth.reads(42, NR, L)
    with open(FA, "wb") as f:
        step = 1_000_000
        for a0 in range(0, NR, step):
            n = min(step, NR - a0)
            rec = np.empty((n, 11 + L + 1), dtype=np.uint8)
            rec[:, 0], rec[:, 1], rec[:, 10], rec[:, -1] = ord(">"), ord("r"), 10, 10
            ids = np.arange(a0, a0 + n)
            for d in range(8):
                rec[:, 9 - d] = 48 + (ids // 10**d) % 10
            rec[:, 11:11 + L] = h_bases[a0 * L:(a0 + n) * L].reshape(n, L)
            f.write(rec.tobytes())
    del h_bases
    out = {"file_bytes": os.path.getsize(FA), "reads": NR, "host_cores": os.cpu_count(), "rows": []}
    ctx = mp.get_context("spawn")
    for W in (1, 2, 4, 8):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        q = ctx.Queue()
        ps = [ctx.Process(target=worker, args=(r, W, port, q)) for r in range(W)]
        for p in ps: p.start()
        res = q.get(timeout=900)
        for p in ps: p.join(timeout=120)
        out["rows"].append({"world": W, "shared_ms_max_over_ranks": round(res["shared"] * 1e3, 1), "whole_file_ms_max_over_ranks": round(res["whole_file"] * 1e3, 1)})
        print(out["rows"][-1], file=sys.stderr, flush=True)
    os.remove(FA)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
