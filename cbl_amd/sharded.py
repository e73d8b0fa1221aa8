"""Multi-GPU build: prefix-range sharding of one CBL index over the ranks of a torch.distributed group.

The reference is single-process (no distributed layer); this is the path BASELINE.json's north_star
adds: buckets are independent by prefix, so the 2^PREFIX_BITS space is cut into `world` contiguous ranges, one per
GPU, and what the k-mers turn into crosses xGMI once. Two wire protocols (ShardedBuilder(protocol=...)):

  "sorted" (default)
    rank r: its contiguous shard of the reads  --KRN-1 + the FULL stable partition (cblx_sorted_batch_begin)-->
            a prefix-sorted batch; per destination a contiguous slice of it: non-empty prefixes, words per prefix, and
            the suffixes alone, packed (6 B per word at K=31 / PREFIX_BITS=24; a full word is 9 B)
            --grouped send/recv over RCCL-->  the batches of all ranks for MY prefix range
            --bucket-by-bucket merge + KRN-3 (cblx_insert_sorted_batches_device)-->  resident sub-index of my range
    Nothing is partitioned twice, and a third fewer bytes cross the links (the exchange is what bounds the job).
  "words"
    rank r: KRN-1 --> words --stable partition by destination (cblx_seq_words_partitioned_device)--> exchange of the
            9-byte words --> the whole single-GPU pipeline on the receiver (cblx_insert_words_device).

Global stream order is slice-major, rank-minor and both protocols deliver by (slice, source rank), so the
first-occurrence order inside Vec buckets (/root/reference/src/trievec/mod.rs:81-87) is that of the one-process
build. Ranges are balanced by quantiles of a sampled prefix histogram because necklace prefixes are heavily skewed
toward small values: equal-width ranges would put nearly all work on rank 0.

The serialized index of the whole job = header (canonical byte, total bucket count) + the per-rank bucket entries
concatenated in rank order (`gather_serialized`).
"""
from __future__ import annotations

import time

import numpy as np

MAX_MSG_BYTES = 256 << 20  # cap of one point-to-point message of the exchange
MAX_DEST = 16  # cblx.h: the destination partition takes at most 16 prefix ranges
SLICE_MAX_BASES = 1 << 31  # bases per slice of a rank's shard: one slice = one batch of fewer than 2^32 words
HIST_BITS = 16      # resolution of the splitter histogram (top bits of the prefix)
SAMPLE_STRIDE = 61  # every 61st word feeds the histogram


def choose_bounds(hist: np.ndarray, world: int, prefix_bits: int, hist_bits: int) -> np.ndarray:
    """world-1 ascending prefix values cutting the histogram mass into `world` near-equal parts."""
    cum = np.cumsum(hist.astype(np.float64))
    total = cum[-1] if len(cum) else 0.0
    shift = prefix_bits - hist_bits
    bounds = []
    for d in range(1, world):
        cell = int(np.searchsorted(cum, total * d / world, side="left")) + 1 if total > 0 else d * len(hist) // world
        cell = min(max(cell, 1), len(hist))
        bounds.append(min(cell << shift, (1 << prefix_bits) - 1) if shift >= 0 else cell >> -shift)
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return np.asarray(bounds, dtype=np.uint32)


class _DeviceArray:
    """Device memory owned by libcblx, viewed through __cuda_array_interface__ (torch.as_tensor makes a no-copy view)."""

    def __init__(self, ptr: int, n: int, typestr: str):
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": (n,), "typestr": typestr, "version": 2, "strides": None}


class GpuEngine:
    """The three device steps of the sharded build, on libcblx (torch tensors only carry the device memory)."""

    def __init__(self, cbl):
        import torch

        self.cbl = cbl
        self.torch = torch
        c = cbl.consts()
        self.sb, self.pb = c["suffix_bits"], cbl.prefix_bits
        # element type of the hi array follows from K (cblx_consts.hi_bytes): none, uint8 (K = 31) or 64-bit
        self.hi_dtype = {0: None, 1: torch.uint8, 8: torch.int64}[c["hi_bytes"]]
        self.device = None

    def seq_words(self, d_bases, d_offsets, n):
        torch = self.torch
        self.device = d_bases.device
        cap = int(d_offsets[n] - d_offsets[0])  # >= number of k-mers of the slice
        lo = torch.empty(cap + 1, dtype=torch.int64, device=self.device)
        hi = torch.empty(cap + 1, dtype=self.hi_dtype, device=self.device) if self.hi_dtype is not None else None
        nw = self.cbl.seq_words_device(d_bases, d_offsets, n, lo, hi, cap)
        return lo[:nw], (hi[:nw] if hi is not None else None)

    def sample_hist(self, lo, hi):
        """Histogram of the top HIST_BITS prefix bits over a strided sample (int64 tensor on the device)."""
        torch = self.torch
        hb = min(HIST_BITS, self.pb)
        shift = self.sb + self.pb - hb  # bit position of the histogram key inside the word
        slo = lo[::SAMPLE_STRIDE]
        shi = hi[::SAMPLE_STRIDE].to(torch.int64) if hi is not None else None
        if shift >= 64:
            key = (shi >> (shift - 64)) & ((1 << hb) - 1)
        else:
            key = (slo >> shift) & ((1 << (64 - shift)) - 1)
            if shi is not None and shift + hb > 64:
                key = key | (shi << (64 - shift))
            key = key & ((1 << hb) - 1)
        return torch.bincount(key, minlength=1 << hb)

    def partition(self, lo, hi, bounds, nd):
        torch = self.torch
        n = int(lo.numel())
        out_lo = torch.empty(n + 1, dtype=torch.int64, device=lo.device)
        out_hi = torch.empty(n + 1, dtype=hi.dtype, device=lo.device) if hi is not None else None
        counts = self.cbl.partition_words_device(lo, hi, n, bounds, nd, out_lo, out_hi)
        return out_lo[:n], (out_hi[:n] if out_hi is not None else None), counts

    def seq_words_partitioned(self, d_bases, d_offsets, n, bounds, nd):
        """KRN-1 and the destination partition in one library call (words are read once on the way out)."""
        torch = self.torch
        cap = int(d_offsets[n] - d_offsets[0])
        lo = torch.empty(cap + 1, dtype=torch.int64, device=d_bases.device)
        hi = torch.empty(cap + 1, dtype=self.hi_dtype, device=d_bases.device) if self.hi_dtype is not None else None
        nw, counts = self.cbl.seq_words_partitioned_device(d_bases, d_offsets, n, bounds, nd, lo, hi, cap)
        return lo[:nw], (hi[:nw] if hi is not None else None), counts

    def insert_words(self, lo, hi):
        self.cbl.insert_words_device(lo, hi, int(lo.numel()))

    # ---- one file, file-order parity ------------------------------------------------------------------------------------
    def count_fastx_records(self, path):
        return self.cbl.count_fastx_records(path)

    def stage_fastx_blocks(self, path, block, rank, world):
        """This rank's block-cyclic share of the file's records, staged in HBM by libcblx, as torch views (no copy)."""
        torch = self.torch
        pb, po, n, n_file = self.cbl.stage_fastx_blocks(path, block, rank, world)
        dev = torch.device("cuda", torch.cuda.current_device())
        offsets = torch.as_tensor(_DeviceArray(po, n + 1, "<i8"), device=dev)
        nbytes = int(offsets[n]) if n else 0
        bases = torch.as_tensor(_DeviceArray(pb, max(nbytes, 1), "|u1"), device=dev)
        self.device = dev
        return bases, offsets, n, n_file

    def stage_fastx_blocks_comm(self, comm, path, block, slices):
        """The same with the parse shared between the ranks of a native communicator: every rank reads 1 / world of the file."""
        torch = self.torch
        pb, po, n, n_file, block = self.cbl.stage_fastx_blocks_comm(comm, path, block, slices)
        dev = torch.device("cuda", torch.cuda.current_device())
        offsets = torch.as_tensor(_DeviceArray(po, n + 1, "<i8"), device=dev)
        nbytes = int(offsets[n]) if n else 0
        bases = torch.as_tensor(_DeviceArray(pb, max(nbytes, 1), "|u1"), device=dev)
        self.device = dev
        return bases, offsets, n, n_file, block

    def stage_release(self):
        self.cbl.stage_release()

    def empty_like(self, t, n):
        return self.torch.empty(n, dtype=t.dtype, device=t.device)

    # ---- sorted-batch protocol ----------------------------------------------------------------------------------------
    def suffix_bytes(self):
        return self.cbl.consts()["bytes"]

    def sorted_batch_begin(self, d_bases, d_offsets, n, bounds, nd):
        self.device = d_bases.device
        return self.cbl.sorted_batch_begin(d_bases, d_offsets, n, bounds, nd)

    def sorted_batch_export(self, n_buckets, n_words):
        torch = self.torch
        prefix = torch.empty(max(n_buckets, 1), dtype=torch.int32, device=self.device)
        count = torch.empty(max(n_buckets, 1), dtype=torch.int32, device=self.device)
        suffix = torch.empty(max(n_words * self.suffix_bytes(), 1), dtype=torch.uint8, device=self.device)
        self.cbl.sorted_batch_export(prefix, count, suffix)
        return prefix[:n_buckets], count[:n_buckets], suffix[: n_words * self.suffix_bytes()]

    def insert_sorted_batches(self, batches):
        self.cbl.insert_sorted_batches_device(batches)


class _Wire:
    """The personalised exchange both the sharded build and the re-shard of a sharded index run over the process group,
    with its accounting."""

    def _wire_init(self, dist):
        self.dist = dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        if self.world > MAX_DEST:
            raise ValueError(f"at most {MAX_DEST} ranks (the destination partition of libcblx takes <= {MAX_DEST} ranges), got {self.world}")
        # exchange accounting since the last reset_stats(): bytes that left / reached this rank (own run excluded), the
        # wall time during which an exchange was outstanding (first issue -> last wait of each call) and the time this
        # rank spent blocked in those waits
        self.stats = {"sent_bytes": 0, "recv_bytes": 0, "outstanding_s": 0.0, "wait_s": 0.0, "messages": 0}
        self._t_first_issue = None

    def _exchange(self, src, dst, send_l, recv_l):
        """Personalised exchange: src holds the runs for rank 0..W-1 back to back (send_l), dst receives the runs of
        source rank 0..W-1 back to back (recv_l). Grouped point-to-point (what RCCL's all-to-all is built from), every
        message capped at MAX_MSG_BYTES: torch/RCCL all_to_all_single was measured here to drop data once a message
        reaches 2^31 bytes or 2^30 elements (tools/dev_a2a_check.py). The rank's own run is a local copy.
        Returns the outstanding requests."""
        dist, W, me = self.dist, self.world, self.rank
        step = max(1, MAX_MSG_BYTES // src.element_size())
        soff = [0] * (W + 1)
        roff = [0] * (W + 1)
        for r in range(W):
            soff[r + 1] = soff[r] + send_l[r]
            roff[r + 1] = roff[r] + recv_l[r]
        dst[roff[me] : roff[me + 1]].copy_(src[soff[me] : soff[me + 1]])
        es = src.element_size()
        self.stats["sent_bytes"] += (soff[W] - send_l[me]) * es
        self.stats["recv_bytes"] += (roff[W] - recv_l[me]) * es
        if self._t_first_issue is None:
            self._t_first_issue = time.perf_counter()
        ops = []
        for d in range(1, W):  # ring order keeps the pairing of sends and receives symmetric across ranks
            to, frm = (me + d) % W, (me - d) % W
            for o in range(0, send_l[to], step):
                ops.append(dist.P2POp(dist.isend, src[soff[to] + o : soff[to] + min(o + step, send_l[to])], to))
            for o in range(0, recv_l[frm], step):
                ops.append(dist.P2POp(dist.irecv, dst[roff[frm] + o : roff[frm] + min(o + step, recv_l[frm])], frm))
        self.stats["messages"] += len(ops)
        return list(dist.batch_isend_irecv(ops)) if ops else []

    def _drain(self, inflight, probe):
        """Wait for every outstanding exchange of this call, then hand the received buffers over to libcblx's own stream
        (on the host: the library does not run on torch's stream)."""
        t0 = time.perf_counter()
        for works, *_ in inflight:
            for x in works:
                x.wait()
        if probe is not None and getattr(probe, "is_cuda", False):
            import torch

            torch.cuda.current_stream(probe.device).synchronize()
        t1 = time.perf_counter()
        self.stats["wait_s"] += t1 - t0
        if self._t_first_issue is not None:
            self.stats["outstanding_s"] += t1 - self._t_first_issue
            self._t_first_issue = None

    def reset_stats(self):
        for k in self.stats:
            self.stats[k] = 0 if isinstance(self.stats[k], int) else 0.0

    def _host_tensor_device(self):
        return "cuda" if self.dist.get_backend() == "nccl" else "cpu"

    def _all_reduce_ints(self, vals, op="sum"):
        import torch

        t = torch.tensor(list(vals), dtype=torch.int64, device=self._host_tensor_device())
        R = self.dist.ReduceOp
        self.dist.all_reduce(t, op={"sum": R.SUM, "min": R.MIN, "max": R.MAX}[op])
        return [int(x) for x in t.cpu().tolist()]


class ShardedBuilder(_Wire):
    """`insert_seqs_device` over a process group: every rank passes ITS contiguous shard of the reads.

    The shard is consumed in `slices` slices so that the all-to-all of slice c overlaps the encode + partition of slice
    c+1 (the exchange is the only step that leaves the GPU: ~7/8 of the words cross xGMI on 8 ranks). The job's stream
    order is therefore slice-major, rank-minor: (slice 0 of rank 0, slice 0 of rank 1, ..., slice 1 of rank 0, ...),
    i.e. file order when the file is dealt to the ranks block-cyclically; with slices=1 it is plain rank order.
    """

    def __init__(self, cbl, dist, engine=None, slices: int = 4, slack: float = 1.3, protocol: str | None = None, comm=None, slice_weights=None):
        """comm: a cbl_amd.Comm — the whole insert then runs inside libcblx (cblx_sharded_insert_seqs_device, exchange on
        RCCL directly; protocol "bins" (the exchange sits between the first and the second partition pass, nothing is
        partitioned twice or copied), "sorted", "replicate" (reads cross the links as bit planes, every rank transforms all of
        them and keeps its prefix range) or "auto"); `dist` is still used for the few host-side agreements (slice
        counts). Without comm the device steps are driven from here over torch.distributed: "sorted" (default) or "words"."""
        self.cbl = cbl
        self.engine = engine or GpuEngine(cbl)
        if protocol is None:
            protocol = "auto" if comm is not None else "sorted"
        if protocol not in (("sorted", "bins", "auto", "replicate") if comm is not None else ("sorted", "words")):
            raise ValueError("protocol must be 'sorted', 'bins', 'replicate' or 'auto' with a native communicator, 'sorted' or 'words' without")
        self.protocol = protocol if (comm is not None or hasattr(self.engine, "sorted_batch_begin")) else "words"
        self.comm = comm
        if comm is not None:
            comm.set_protocol(self.protocol)
        self._wire_init(dist)
        self.slices, self.slack = max(1, slices), slack
        self.slice_weights = tuple(slice_weights) if slice_weights else None  # relative slice lengths (len == slices), else the default cut
        self.bounds = None  # fixed by the first batch so later batches land on the same owners
        self.last_counts = None

    # Slices of the GROUPED receiver's schedule (native "bins" protocol on an empty index, DESIGN_HISTORY.md §5.6): the senders run all their
    # slices before the data crosses group-major, and only the first group's share of the LAST slice is exposed (the earlier slices'
    # shares of it cross under the next slice's kernels) — so the last slice is the short one. Measured against a paced wire
    # (cfg 3, 55 GB/s per link, 4 groups): 2 equal slices 56.6 ms, these three 54.9, four (45 / 30 / 17 / 8 %) 55.3.
    GROUPED_WEIGHTS = (5, 3, 2)

    @staticmethod
    def slice_bounds(n: int, slices: int, weights=None):
        """Read ranges [a, b) of the slices of an n-read shard (same formula on every rank). `weights`: relative lengths (one per
        slice). Default: with three or more slices the first and the last are half as long as the others: the first slice's
        encode + partition is the only work no exchange hides (it bounds the job when the links do), the last slice's exchange
        the only exchange no kernel hides (it bounds the job when the kernels do)."""
        slices = max(1, slices)  # every rank walks the same number of slices whatever its n (an empty slice still takes part in the exchange)
        if weights is not None and len(weights) == slices:
            w = list(weights)
        elif slices < 3:
            return [(n * c // slices, n * (c + 1) // slices) for c in range(slices)]
        else:
            w = [1] + [2] * (slices - 2) + [1]
        tot, acc, cuts = sum(w), 0, [0]
        for x in w:
            acc += x
            cuts.append(n * acc // tot)
        return [(cuts[c], cuts[c + 1]) for c in range(slices)]

    def insert_seqs_device(self, d_bases, d_offsets, n, slice_list=None):
        """slice_list: explicit read ranges [(a, b), ...] of the slices (the same NUMBER of slices on every rank) instead of
        the default cut of the shard into `slices` pieces."""
        if slice_list is None:
            # one slice goes through the kernels as ONE batch, and a batch takes fewer than 2^32 words: more slices for a
            # big shard (the same number on every rank: a slice is also a round of the exchange)
            nbases = int(d_offsets[n] - d_offsets[0]) if n else 0
            need = -(-nbases // SLICE_MAX_BASES)
            slice_list = self.slice_bounds(n, max(self.slices, self._all_reduce_ints([need], "max")[0]), self.slice_weights)
        self._slice_list = list(slice_list)
        if self.comm is not None:
            return self._insert_native(d_bases, d_offsets, n)
        if self.protocol == "sorted":
            return self._insert_sorted(d_bases, d_offsets, n)
        return self._insert_words(d_bases, d_offsets, n)

    def insert_fastx_file(self, path, block: int = 0) -> int:
        """Sharded build from ONE FASTA / FASTQ(.gz) file every rank can read, with the stream order of the file: the
        records are dealt to the ranks in blocks of `block` records, block-cyclically (block j -> rank j % W, where it is
        that rank's slice j // W), so slice-major / rank-minor order IS file order and the gathered index is byte-identical
        to `cbl build` of the file (/root/reference/examples/cbl.rs:154-166). block = 0: sized so that every rank gets
        about `slices` blocks. Returns the number of records in the file."""
        eng, W = self.engine, self.world
        if self.comm is not None and hasattr(eng, "stage_fastx_blocks_comm"):
            # the parse is shared: every rank scans 1 / W of the file and reads only the bytes of its own blocks
            bases, offsets, n, n_file, block = eng.stage_fastx_blocks_comm(self.comm, path, max(block, 0), self.slices)
        else:
            if block <= 0:
                n_file = eng.count_fastx_records(path)
                block = max(1, -(-n_file // (W * self.slices)))
            bases, offsets, n, n_file = eng.stage_fastx_blocks(path, block, self.rank, W)
        try:
            nblocks = -(-n_file // block)
            rounds = -(-nblocks // W)  # the blocks of the file are dealt W at a time
            sl = [(min(c * block, n), min((c + 1) * block, n)) for c in range(max(rounds, 1))]
            self.insert_seqs_device(bases, offsets, n, slice_list=sl)
        finally:
            eng.stage_release()
        return n_file

    def _insert_native(self, d_bases, d_offsets, n):
        sl = self._slice_list
        cuts = [sl[0][0]] + [b for _a, b in sl]
        assert all(sl[i][1] == sl[i + 1][0] for i in range(len(sl) - 1)), "slices must be contiguous"
        have = self.bounds is not None
        bounds = np.ascontiguousarray(self.bounds, dtype=np.uint32).copy() if have else np.zeros(max(self.world - 1, 0), dtype=np.uint32)
        before = self.comm.stats()
        t0 = time.perf_counter()
        self.cbl.sharded_insert_seqs_device(self.comm, d_bases, d_offsets, n, cuts, bounds, have)
        dt = time.perf_counter() - t0
        after = self.comm.stats()
        self.bounds = bounds
        for k in ("sent_bytes", "recv_bytes", "messages"):
            self.stats[k] += after[k] - before[k]
        self.stats["outstanding_s"] += dt  # the exchange overlaps the kernels inside the call: the whole call is the window

    def _choose_bounds_from(self, d_bases, off, n):
        """First batch only: quantile ranges from a sampled, all-reduced prefix histogram of the first slice."""
        eng, dist = self.engine, self.dist
        lo, hi = eng.seq_words(d_bases, off, n)
        hist = eng.sample_hist(lo, hi)
        dist.all_reduce(hist)
        hb = min(HIST_BITS, self.cbl.prefix_bits)
        self.bounds = choose_bounds(hist.cpu().numpy(), self.world, self.cbl.prefix_bits, hb)

    def _insert_sorted(self, d_bases, d_offsets, n):
        import torch

        dist, eng, W = self.dist, self.engine, self.world
        B = eng.suffix_bytes()
        slices_in = []  # (recv_b, recv_w, prefix_r, count_r, suffix_r) per slice
        inflight = []
        send_tot, recv_tot = [0] * W, [0] * W
        for a, b in self._slice_list:
            off = d_offsets[a : b + 1]  # offsets stay absolute: no copy of the bases
            if self.bounds is None:
                self._choose_bounds_from(d_bases, off, b - a)
            bs, ws = eng.sorted_batch_begin(d_bases, off, b - a, self.bounds, W)
            send_b = [int(bs[d + 1] - bs[d]) for d in range(W)]
            send_w = [int(ws[d + 1] - ws[d]) for d in range(W)]
            prefix, count, suffix = eng.sorted_batch_export(int(bs[W]), int(ws[W]))
            send = torch.tensor([send_b, send_w], dtype=torch.int64, device=prefix.device).t().contiguous()  # [W, 2]
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send)
            recv_h = recv.cpu().tolist()
            recv_b, recv_w = [int(x[0]) for x in recv_h], [int(x[1]) for x in recv_h]
            prefix_r = eng.empty_like(prefix, max(sum(recv_b), 1))
            count_r = eng.empty_like(count, max(sum(recv_b), 1))
            suffix_r = eng.empty_like(suffix, max(sum(recv_w) * B, 1))
            works = self._exchange(prefix, prefix_r, send_b, recv_b)
            works += self._exchange(count, count_r, send_b, recv_b)
            works += self._exchange(suffix, suffix_r, [w * B for w in send_w], [w * B for w in recv_w])
            inflight.append((works, prefix, count, suffix))
            slices_in.append((recv_b, recv_w, prefix_r, count_r, suffix_r))
            send_tot = [x + y for x, y in zip(send_tot, send_w)]
            recv_tot = [x + y for x, y in zip(recv_tot, recv_w)]
        self._drain(inflight, slices_in[0][2] if slices_in else None)
        inflight = []
        self.last_counts = (send_tot, recv_tot)
        batches = []  # stream order: slice-major, source-rank-minor
        for recv_b, recv_w, prefix_r, count_r, suffix_r in slices_in:
            bo = wo = 0
            for r in range(W):
                if recv_w[r]:
                    batches.append((recv_b[r], recv_w[r], prefix_r[bo : bo + recv_b[r]], count_r[bo : bo + recv_b[r]], suffix_r[wo * B : (wo + recv_w[r]) * B]))
                bo += recv_b[r]
                wo += recv_w[r]
        if batches:
            eng.insert_sorted_batches(batches)

    def _insert_words(self, d_bases, d_offsets, n):
        import torch

        dist, eng, W = self.dist, self.engine, self.world
        k = self.cbl.k if hasattr(self.cbl, "k") else None
        recv_lo = recv_hi = None
        filled, cap = 0, 0
        inflight = []  # (works, send buffers kept alive)
        send_tot = [0] * W
        recv_tot = [0] * W
        for a, b in self._slice_list:
            off = d_offsets[a : b + 1]  # offsets stay absolute: no copy of the bases
            if self.bounds is None or not hasattr(eng, "seq_words_partitioned"):
                lo, hi = eng.seq_words(d_bases, off, b - a)
                if self.bounds is None:
                    hist = eng.sample_hist(lo, hi)
                    dist.all_reduce(hist)
                    hb = min(HIST_BITS, self.cbl.prefix_bits)
                    self.bounds = choose_bounds(hist.cpu().numpy(), W, self.cbl.prefix_bits, hb)
                plo, phi, counts = eng.partition(lo, hi, self.bounds, W)
                del lo, hi
            else:
                plo, phi, counts = eng.seq_words_partitioned(d_bases, off, b - a, self.bounds, W)
            n_words = int(plo.numel())
            send = torch.tensor(counts, dtype=torch.int64, device=plo.device)
            recv = torch.empty_like(send)
            dist.all_to_all_single(recv, send)
            send_l, recv_l = [int(x) for x in counts], [int(x) for x in recv.cpu().tolist()]
            n_recv = sum(recv_l)
            if recv_lo is None:  # one receive buffer for the whole batch: slices land back to back, no gather copy
                est_total = int(n_words * (n / max(b - a, 1)) * self.slack) + 4096
                cap = max(est_total, n_recv)
                recv_lo = eng.empty_like(plo, cap)
                recv_hi = eng.empty_like(phi, cap) if phi is not None else None
            if filled + n_recv > cap:  # quantile ranges drifted: grow (rare)
                for wk, *_ in inflight:
                    for x in wk:
                        x.wait()
                inflight = []
                cap = int((filled + n_recv) * 1.5) + 4096
                nlo = eng.empty_like(plo, cap)
                nlo[:filled] = recv_lo[:filled]
                recv_lo = nlo
                if recv_hi is not None:
                    nhi = eng.empty_like(phi, cap)
                    nhi[:filled] = recv_hi[:filled]
                    recv_hi = nhi
            works = self._exchange(plo, recv_lo[filled : filled + n_recv], send_l, recv_l)
            if phi is not None:
                works += self._exchange(phi, recv_hi[filled : filled + n_recv], send_l, recv_l)
            inflight.append((works, plo, phi))
            filled += n_recv
            send_tot = [x + y for x, y in zip(send_tot, send_l)]
            recv_tot = [x + y for x, y in zip(recv_tot, recv_l)]
        self._drain(inflight, recv_lo)
        inflight = []
        self.last_counts = (send_tot, recv_tot)
        if filled:
            eng.insert_words(recv_lo[:filled], recv_hi[:filled] if recv_hi is not None else None)

    def reset(self):
        self.bounds = None


class GpuShard:
    """One rank's share of a sharded index on libcblx: the device steps ShardedIndex drives (torch tensors only carry the
    device memory that crosses the wire)."""

    def __init__(self, cbl, device=None):
        import torch

        self.cbl, self.torch = cbl, torch
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())

    def new_like(self, profile: bool = False) -> "GpuShard":
        from . import CBL

        dev = self.device.index if self.device.index is not None else -1
        return GpuShard(CBL(self.cbl.k, self.cbl.prefix_bits, canonical=self.cbl.is_canonical(), device=dev, profile=profile), self.device)

    def builder_engine(self):
        return GpuEngine(self.cbl)

    def clear(self):
        self.cbl.clear()

    def count(self) -> int:
        return self.cbl.count()

    def is_canonical(self) -> bool:
        return self.cbl.is_canonical()

    def suffix_bytes(self) -> int:
        return self.cbl.consts()["bytes"]

    def load_shard(self, path, rank, world, bounds, sequential):
        return self.cbl.load_shard_from_file(path, rank, world, bounds, sequential)

    def split(self, bounds, nd):
        return self.cbl.resident_split(bounds, nd)

    def export(self):
        """(prefix int32, count int32, kind uint8, packed suffixes uint8) of the whole resident share, on the device."""
        t = self.torch
        nb, nw, B = self.cbl.num_buckets(), self.cbl.count(), self.suffix_bytes()
        prefix = t.empty(max(nb, 1), dtype=t.int32, device=self.device)
        count = t.empty(max(nb, 1), dtype=t.int32, device=self.device)
        kind = t.empty(max(nb, 1), dtype=t.uint8, device=self.device)
        suffix = t.empty(max(nw * B, 1), dtype=t.uint8, device=self.device)
        self.cbl.resident_export(prefix, count, kind, suffix)
        return prefix[:nb], count[:nb], kind[:nb], suffix[: nw * B]

    def empty_like(self, t, n):
        return self.torch.empty(n, dtype=t.dtype, device=t.device)

    def install(self, parts):
        self.cbl.install_buckets_device(parts)

    def merge_assign(self, other: "GpuShard"):
        self.cbl |= other.cbl

    def body_size(self):
        return self.cbl.serialized_body_size()

    def write_body_at(self, path, off):
        self.cbl.write_body_at(path, off)


class ShardedIndex(_Wire):
    """One CBL index cut into `world` contiguous prefix ranges, one per rank (rank order = ascending prefixes).

    Carries the reference's build / insert / merge / save surface to N GPUs:
      insert_seqs_device  the sharded build (ShardedBuilder) into this index
      load_from_file      every rank reads ITS prefix range of an index file (read_index, /root/reference/examples/cbl.rs:117-130)
      merge_assign        `self |= other` (/root/reference/src/cbl.rs:433-449) range by range; an operand cut at other bounds
                          is first re-sharded: each rank exports its share as bucket batches (kinds and stored order kept),
                          one grouped exchange, install on the new owner
      save_to_file        header by rank 0, every rank writes its entries at its own offset (write_index, examples/cbl.rs:132-142)
    `bounds`: world-1 ascending prefix values, rank r owns bounds[r-1] <= prefix < bounds[r]; None while the index is empty.
    """

    def __init__(self, k=None, prefix_bits=24, dist=None, canonical=False, device=-1, slices=4, protocol="sorted", shard=None, profile=False):
        if shard is None:
            from . import CBL

            shard = GpuShard(CBL(k, prefix_bits, canonical=canonical, device=device, profile=profile))
        self.shard = shard
        self._wire_init(dist)
        self.slices, self.protocol = slices, protocol
        self.bounds = None
        self._builder = None

    @property
    def cbl(self):
        return self.shard.cbl

    # ---- build ---------------------------------------------------------------------------------------------------------
    def insert_seqs_device(self, d_bases, d_offsets, n):
        if self._builder is None:
            self._builder = ShardedBuilder(self.shard.cbl, self.dist, engine=self.shard.builder_engine(), slices=self.slices, protocol=self.protocol)
            self._builder.stats = self.stats  # one account for the index
        self._builder.bounds = self.bounds
        self._builder.insert_seqs_device(d_bases, d_offsets, n)
        self.bounds = self._builder.bounds

    def insert_fastx_file(self, path, block: int = 0) -> int:
        """`cbl build` / `cbl insert` of one file on all ranks, stream order = file order (ShardedBuilder.insert_fastx_file)."""
        if self._builder is None:
            self._builder = ShardedBuilder(self.shard.cbl, self.dist, engine=self.shard.builder_engine(), slices=self.slices, protocol=self.protocol)
            self._builder.stats = self.stats
        self._builder.bounds = self.bounds
        n = self._builder.insert_fastx_file(path, block)
        self.bounds = self._builder.bounds
        return n

    def local_count(self) -> int:
        return self.shard.count()

    def count(self) -> int:
        return self._all_reduce_ints([self.shard.count()])[0]

    def _like(self, shard, bounds):
        o = ShardedIndex(dist=self.dist, slices=self.slices, protocol=self.protocol, shard=shard)
        o.bounds = None if bounds is None else np.asarray(bounds, dtype=np.uint32).copy()
        return o

    def clone(self, profile: bool = False) -> "ShardedIndex":
        sh = self.shard.new_like(profile=profile)
        sh.merge_assign(self.shard)  # |= into an empty index clones every bucket as stored (src/trievec/set_ops.rs:43-71)
        return self._like(sh, self.bounds)

    def copy_from(self, other: "ShardedIndex") -> "ShardedIndex":
        """Make this index a copy of `other` (same bounds), reusing this rank's context and its cached device memory."""
        self.shard.clear()
        self.shard.merge_assign(other.shard)
        self.bounds = None if other.bounds is None else np.asarray(other.bounds, dtype=np.uint32).copy()
        self._builder = None
        return self

    # ---- files ---------------------------------------------------------------------------------------------------------
    def load_from_file(self, path, bounds=None):
        """Replace the contents by the index file at `path` (visible to every rank). bounds=None cuts it into ranges of
        about equal byte length; passing another index's bounds makes the two mergeable without an exchange."""
        W = self.world
        b_in = None if bounds is None else np.asarray(bounds, dtype=np.uint32)
        for sequential in (False, True):
            info, b = self.shard.load_shard(path, self.rank, W, b_in, sequential)
            ok, = self._all_reduce_ints([int(info["exact"])], "min")
            total, = self._all_reduce_ints([int(info["local_entries"])])
            if ok and total == info["header_entries"]:
                break
        else:
            raise ValueError(f"{path}: the entries of the {W} prefix ranges do not add up to the header's count (corrupt index file?)")
        # every rank cut the file with the same deterministic procedure; keep rank 0's word for it anyway
        bb = self._all_reduce_ints([int(x) for x in b] if self.rank == 0 else [0] * len(b)) if W > 1 else []
        self.bounds = np.asarray(bb, dtype=np.uint32)
        self._builder = None
        return info

    def save_to_file(self, path) -> int:
        """One index file from the shares of all ranks (one node, one file system). Returns its size."""
        import torch

        ne, nbytes = self.shard.body_size()
        canon = int(self.shard.is_canonical())
        mine = torch.tensor([ne, nbytes, canon], dtype=torch.int64, device=self._host_tensor_device())
        parts = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        parts = [p.cpu().tolist() for p in parts]
        if len({p[2] for p in parts}) != 1:
            raise ValueError("One of the index is canonical while the other isn't")
        header = bytes([canon]) + _varint(sum(p[0] for p in parts))
        offset = len(header) + sum(p[1] for p in parts[: self.rank])
        total = len(header) + sum(p[1] for p in parts)
        if self.rank == 0:
            with open(path, "wb") as f:
                f.write(header)
                f.truncate(total)
        self.dist.barrier()
        self.shard.write_body_at(path, offset)
        self.dist.barrier()
        return total

    # ---- re-shard + merge ----------------------------------------------------------------------------------------------
    def resharded(self, bounds) -> "ShardedIndex":
        """A copy of this index cut at `bounds` instead of self.bounds (self is left as it is)."""
        import torch

        W, sh = self.world, self.shard
        bounds = np.asarray(bounds, dtype=np.uint32)
        B = sh.suffix_bytes()
        bs, ws = sh.split(bounds, W)
        prefix, count, kind, suffix = sh.export()
        send_b = [int(bs[d + 1] - bs[d]) for d in range(W)]
        send_w = [int(ws[d + 1] - ws[d]) for d in range(W)]
        send = torch.tensor([send_b, send_w], dtype=torch.int64, device=prefix.device).t().contiguous()  # [W, 2]
        recv = torch.empty_like(send)
        self.dist.all_to_all_single(recv, send)
        recv_h = recv.cpu().tolist()
        recv_b, recv_w = [int(x[0]) for x in recv_h], [int(x[1]) for x in recv_h]
        prefix_r = sh.empty_like(prefix, max(sum(recv_b), 1))
        count_r = sh.empty_like(count, max(sum(recv_b), 1))
        kind_r = sh.empty_like(kind, max(sum(recv_b), 1))
        suffix_r = sh.empty_like(suffix, max(sum(recv_w) * B, 1))
        works = self._exchange(prefix, prefix_r, send_b, recv_b)
        works += self._exchange(count, count_r, send_b, recv_b)
        works += self._exchange(kind, kind_r, send_b, recv_b)
        works += self._exchange(suffix, suffix_r, [w * B for w in send_w], [w * B for w in recv_w])
        self._drain([(works,)], prefix_r)
        parts, bo, wo = [], 0, 0
        for r in range(W):  # the pieces of my new range, from the ranks that held them, in ascending prefix order
            if recv_b[r]:
                parts.append((recv_b[r], recv_w[r], prefix_r[bo : bo + recv_b[r]], count_r[bo : bo + recv_b[r]], kind_r[bo : bo + recv_b[r]],
                              suffix_r[wo * B : (wo + recv_w[r]) * B]))
            bo += recv_b[r]
            wo += recv_w[r]
        out = sh.new_like()
        out.install(parts)
        return self._like(out, bounds)

    def merge_assign(self, other: "ShardedIndex") -> "ShardedIndex":
        """`self |= other`. When the two are cut at different bounds `other` is re-sharded IN PLACE to self's bounds first
        (it stays the same set; like the reference's |=, which sorts other's Vec buckets it walks, the call may change how
        `other` is laid out, never what it contains)."""
        if other.bounds is None and other.count() == 0:
            return self
        if self.bounds is None:
            self.bounds = None if other.bounds is None else np.asarray(other.bounds, dtype=np.uint32).copy()
        if self.world > 1 and not np.array_equal(np.asarray(self.bounds), np.asarray(other.bounds)):
            moved = other.resharded(self.bounds)
            other.shard, other.bounds, other._builder = moved.shard, moved.bounds, None
        self.shard.merge_assign(other.shard)
        self._builder = None
        return self

    __ior__ = merge_assign


class HostStagedGroup:
    """torch.distributed look-alike whose collectives stage device tensors through the host and a `gloo` group.

    RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), so this is how several ranks of the sharded build run on
    ONE GPU (dry runs of `bench.py --gpus N --shared-gpu`, the -m gpu shim tests): every device step is the real one,
    only the wire is gloo instead of RCCL. Not a production transport."""

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    class _Work:
        def __init__(self, w, dst=None, host=None):
            self.w, self.dst, self.host = w, dst, host

        def wait(self):
            self.w.wait()
            if self.dst is not None:
                self.dst.copy_(self.host)

    isend, irecv = "isend", "irecv"

    class ReduceOp:
        SUM, MAX, MIN = "sum", "max", "min"

    def __init__(self, d):
        self.d = d

    def get_world_size(self):
        return self.d.get_world_size()

    def get_rank(self):
        return self.d.get_rank()

    def get_backend(self):
        return "gloo"

    def barrier(self):
        self.d.barrier()

    def all_reduce(self, t, op=None):
        h = t.cpu()
        rop = {None: self.d.ReduceOp.SUM, "sum": self.d.ReduceOp.SUM, "max": self.d.ReduceOp.MAX, "min": self.d.ReduceOp.MIN}.get(op, op)
        self.d.all_reduce(h, op=rop)
        t.copy_(h)

    def all_gather(self, outs, t):
        hs = [o.cpu() for o in outs]
        self.d.all_gather(hs, t.cpu())
        for o, h in zip(outs, hs):
            o.copy_(h)

    def all_gather_object(self, outs, obj):
        self.d.all_gather_object(outs, obj)

    def all_to_all_single(self, out, inp):
        ho, hi = out.cpu(), inp.cpu()
        self.d.all_to_all_single(ho, hi)
        out.copy_(ho)

    def batch_isend_irecv(self, ops):
        import torch

        works = []
        for o in ops:
            if o.op == "isend":
                works.append(self._Work(self.d.isend(o.tensor.cpu().contiguous(), o.peer)))
            else:
                h = torch.empty(o.tensor.shape, dtype=o.tensor.dtype)
                works.append(self._Work(self.d.irecv(h, o.peer), o.tensor, h))
        return works


def _read_varint(b: bytes, pos: int):
    t = b[pos]
    if t <= 250:
        return t, pos + 1
    nb = {0xFB: 2, 0xFC: 4, 0xFD: 8}[t]
    return int.from_bytes(b[pos + 1 : pos + 1 + nb], "little"), pos + 1 + nb


def _varint(v: int) -> bytes:
    if v <= 250:
        return bytes([v])
    if v < 1 << 16:
        return b"\xfb" + v.to_bytes(2, "little")
    if v < 1 << 32:
        return b"\xfc" + v.to_bytes(4, "little")
    return b"\xfd" + v.to_bytes(8, "little")


def gather_serialized(local_blob: bytes, dist, dst: int = 0):
    """Index file of the whole job from the per-rank serializations (rank order = ascending prefix ranges).

    File = canonical u8 | varint(n_buckets) | entries (/root/reference/src/wordset/mod.rs:388-394); the per-rank
    entries are disjoint ascending prefix runs, so the job's file is one header + their concatenation."""
    nb, pos = _read_varint(local_blob, 1)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, (local_blob[0], nb, local_blob[pos:]))
    if dist.get_rank() != dst:
        return None
    assert len({p[0] for p in parts}) == 1
    return bytes([parts[0][0]]) + _varint(sum(p[1] for p in parts)) + b"".join(p[2] for p in parts)


def save_serialized(local_blob, dist, path) -> int:
    """The same file as `gather_serialized`, written in place: every rank writes its entries at its own offset of `path`
    (one node, one file system), so no rank ever holds another rank's bytes — `gather_serialized` ships every part to
    every rank through pickles, which is fine for tests and hopeless for 9 GB parts. `local_blob`: bytes or a numpy
    uint8 array (`CBL.serialize_np()`). Returns the file size."""
    import numpy as np
    import torch

    blob = np.frombuffer(local_blob, dtype=np.uint8) if isinstance(local_blob, (bytes, bytearray, memoryview)) else local_blob
    head = bytes(blob[:16].tobytes())
    nb, pos = _read_varint(head, 1)
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor([nb, len(blob) - pos, head[0]], dtype=torch.int64, device=dev)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    parts = [p.cpu().tolist() for p in parts]
    assert len({p[2] for p in parts}) == 1, "One of the index is canonical while the other isn't"
    header = bytes([head[0]]) + _varint(sum(p[0] for p in parts))
    offset = len(header) + sum(p[1] for p in parts[:rank])
    total = len(header) + sum(p[1] for p in parts)
    if rank == 0:
        with open(path, "wb") as f:
            f.write(header)
            f.truncate(total)
    dist.barrier()
    with open(path, "r+b") as f:
        f.seek(offset)
        blob[pos:].tofile(f)
    dist.barrier()
    return total
