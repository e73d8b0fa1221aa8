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

import numpy as np

MAX_MSG_BYTES = 256 << 20  # cap of one point-to-point message of the exchange
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


class ShardedBuilder:
    """`insert_seqs_device` over a process group: every rank passes ITS contiguous shard of the reads.

    The shard is consumed in `slices` slices so that the all-to-all of slice c overlaps the encode + partition of slice
    c+1 (the exchange is the only step that leaves the GPU: ~7/8 of the words cross xGMI on 8 ranks). The job's stream
    order is therefore slice-major, rank-minor: (slice 0 of rank 0, slice 0 of rank 1, ..., slice 1 of rank 0, ...),
    i.e. file order when the file is dealt to the ranks block-cyclically; with slices=1 it is plain rank order.
    """

    def __init__(self, cbl, dist, engine=None, slices: int = 4, slack: float = 1.3, protocol: str = "sorted"):
        self.cbl, self.dist = cbl, dist
        self.engine = engine or GpuEngine(cbl)
        if protocol not in ("sorted", "words"):
            raise ValueError("protocol must be 'sorted' or 'words'")
        self.protocol = protocol if hasattr(self.engine, "sorted_batch_begin") else "words"
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.slices, self.slack = max(1, slices), slack
        self.bounds = None  # fixed by the first batch so later batches land on the same owners
        self.last_counts = None

    @staticmethod
    def slice_bounds(n: int, slices: int):
        """Read ranges [a, b) of the slices of an n-read shard (same formula on every rank). With three or more slices the
        first and the last are half as long as the others: the first slice's encode + partition is the only work no
        exchange hides (it bounds the job when the links do), the last slice's exchange the only exchange no kernel hides
        (it bounds the job when the kernels do)."""
        slices = max(1, slices)  # every rank walks the same number of slices whatever its n (an empty slice still takes part in the exchange)
        if slices < 3:
            return [(n * c // slices, n * (c + 1) // slices) for c in range(slices)]
        w = [1] + [2] * (slices - 2) + [1]
        tot, acc, cuts = sum(w), 0, [0]
        for x in w:
            acc += x
            cuts.append(n * acc // tot)
        return [(cuts[c], cuts[c + 1]) for c in range(slices)]

    def insert_seqs_device(self, d_bases, d_offsets, n):
        if self.protocol == "sorted":
            return self._insert_sorted(d_bases, d_offsets, n)
        return self._insert_words(d_bases, d_offsets, n)

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
        for a, b in self.slice_bounds(n, self.slices):
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
        for works, *_ in inflight:
            for x in works:
                x.wait()
        if slices_in and slices_in[0][2].is_cuda:
            torch.cuda.current_stream(slices_in[0][2].device).synchronize()  # libcblx runs on its own stream: hand over on the host
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
        for a, b in self.slice_bounds(n, self.slices):
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
        for works, *_ in inflight:
            for x in works:
                x.wait()
        if recv_lo is not None and recv_lo.is_cuda:
            torch.cuda.current_stream(recv_lo.device).synchronize()  # libcblx runs on its own stream: hand over on the host
        inflight = []
        self.last_counts = (send_tot, recv_tot)
        if filled:
            eng.insert_words(recv_lo[:filled], recv_hi[:filled] if recv_hi is not None else None)

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
        ops = []
        for d in range(1, W):  # ring order keeps the pairing of sends and receives symmetric across ranks
            to, frm = (me + d) % W, (me - d) % W
            for o in range(0, send_l[to], step):
                ops.append(dist.P2POp(dist.isend, src[soff[to] + o : soff[to] + min(o + step, send_l[to])], to))
            for o in range(0, recv_l[frm], step):
                ops.append(dist.P2POp(dist.irecv, dst[roff[frm] + o : roff[frm] + min(o + step, recv_l[frm])], frm))
        return list(dist.batch_isend_irecv(ops)) if ops else []

    def reset(self):
        self.bounds = None


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
