"""CPU stand-ins for the device steps of cbl_amd.sharded (tests only): the CPU oracle behind the engine interfaces
ShardedBuilder / ShardedIndex drive, on torch CPU tensors. The world-2/3 gloo tests run the real orchestration (splitter
choice, exchanges, source-rank ordered installs, load agreement, rank-ordered save) over these."""
import numpy as np


# ---- index bytes <-> entries, independent of oracle/ and of libcblx ---------------------------------
def _rv(b, p):
    t = b[p]
    if t <= 250:
        return t, p + 1
    n = {0xFB: 2, 0xFC: 4, 0xFD: 8}[t]
    return int.from_bytes(b[p + 1 : p + 1 + n], "little"), p + 1 + n


def _wv(v):
    if v <= 250:
        return bytes([v])
    if v < 1 << 16:
        return b"\xfb" + v.to_bytes(2, "little")
    if v < 1 << 32:
        return b"\xfc" + v.to_bytes(4, "little")
    return b"\xfd" + v.to_bytes(8, "little")


def _parse_trie(b, p, depth, nbytes, acc, out):
    c, p = _rv(b, p)
    vals = b[p : p + c]
    p += c
    nc, p = _rv(b, p)
    shift = 8 * (nbytes - 1 - depth)
    if depth + 1 == nbytes:
        assert nc == 0
        out.extend(acc | (v << shift) for v in vals)
        return p
    assert nc == c
    for v in vals:
        p = _parse_trie(b, p, depth + 1, nbytes, acc | (v << shift), out)
    return p


def parse_index(blob: bytes, nbytes: int):
    """(canonical, [(prefix, kind, [suffix, ...], start offset, end offset)]) of a serialized index."""
    canonical = blob[0]
    n, p = _rv(blob, 1)
    entries = []
    for _ in range(n):
        start = p
        prefix, p = _rv(blob, p)
        tag, p = _rv(blob, p)
        items = []
        if tag == 0:
            cnt, p = _rv(blob, p)
            for _ in range(cnt):
                ln, p = _rv(blob, p)
                items.append(int.from_bytes(blob[p : p + ln], "little"))
                p += ln
        else:
            p = _parse_trie(blob, p, 0, nbytes, 0, items)
            ln, p = _rv(blob, p)
            assert ln == len(items)
        entries.append((prefix, tag, items, start, p))
    assert p == len(blob)
    return canonical, entries


def _trie_bytes(items, depth, nbytes):
    shift = 8 * (nbytes - 1 - depth)
    groups = {}
    for s in items:
        groups.setdefault((s >> shift) & 0xFF, []).append(s)
    keys = sorted(groups)
    out = _wv(len(keys)) + bytes(keys)
    if depth == nbytes - 1:
        return out + _wv(0)
    out += _wv(len(keys))
    for k in keys:
        out += _trie_bytes(groups[k], depth + 1, nbytes)
    return out


def entry_bytes(prefix, kind, items, nbytes):
    if kind == 0:
        return _wv(prefix) + _wv(0) + _wv(len(items)) + b"".join(_wv(nbytes) + s.to_bytes(nbytes, "little") for s in items)
    return _wv(prefix) + _wv(1) + _trie_bytes(sorted(items), 0, nbytes) + _wv(len(items))


def build_index(canonical, entries, nbytes):
    return bytes([canonical]) + _wv(len(entries)) + b"".join(entry_bytes(p, k, it, nbytes) for p, k, it, *_ in entries)


class OracleEngine:
    """CPU stand-in for cbl_amd.sharded.GpuEngine: same steps on torch CPU tensors."""

    def __init__(self, orc, k, pb):
        import torch
        from oracle.pyref import params

        self.torch, self.o = torch, orc
        P = params(k, pb)
        self.sb, self.pb = P["SB"], pb

    def _to_ints(self, lo, hi):
        lo = lo.numpy().astype(np.uint64)
        hi = hi.numpy().astype(np.uint64)
        return [int(a) | (int(b) << 64) for a, b in zip(lo, hi)]

    def _from_ints(self, words):
        t = self.torch
        lo = np.array([w & (2**64 - 1) for w in words], dtype=np.uint64).astype(np.int64)
        hi = np.array([w >> 64 for w in words], dtype=np.uint64).astype(np.int64)
        return t.from_numpy(lo), t.from_numpy(hi)

    def seq_words(self, bases, offsets, n):
        b = bases.numpy().tobytes()
        off = offsets.numpy()
        words = []
        for i in range(n):
            words += self.o.seq_words(b[int(off[i]) : int(off[i + 1])])
        return self._from_ints(words)

    def sample_hist(self, lo, hi):
        from cbl_amd.sharded import HIST_BITS, SAMPLE_STRIDE

        hb = min(HIST_BITS, self.pb)
        keys = [(w >> (self.sb + self.pb - hb)) & ((1 << hb) - 1) for w in self._to_ints(lo, hi)[::SAMPLE_STRIDE]]
        return self.torch.from_numpy(np.bincount(np.array(keys, dtype=np.int64), minlength=1 << hb).astype(np.int64))

    def partition(self, lo, hi, bounds, nd):
        words = self._to_ints(lo, hi)
        dest = [int(np.searchsorted(bounds, (w >> self.sb), side="right")) for w in words]
        order = sorted(range(len(words)), key=lambda i: dest[i])  # Python's sort is stable
        plo, phi = self._from_ints([words[i] for i in order])
        return plo, phi, [dest.count(d) for d in range(nd)]

    def insert_words(self, lo, hi):
        self.o.insert_words(self._to_ints(lo, hi))

    # ---- one file, file-order parity: same contract as cblx_stage_fastx_blocks (include/cblx.h) ----------------------------
    @staticmethod
    def _fasta_records(path):
        recs, cur = [], None
        for line in open(path, "rb").read().split(b"\n"):
            line = line.rstrip(b"\r")
            if line.startswith(b">"):
                if cur is not None:
                    recs.append(b"".join(cur))
                cur = []
            elif line and cur is not None:
                cur.append(line)
        if cur is not None:
            recs.append(b"".join(cur))
        return recs

    def count_fastx_records(self, path):
        return len(self._fasta_records(path))

    def stage_fastx_blocks(self, path, block, rank, world):
        t = self.torch
        recs = self._fasta_records(path)
        mine = [r for i, r in enumerate(recs) if (i // block) % world == rank]
        offs = np.zeros(len(mine) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(r) for r in mine])
        raw = b"".join(mine)
        return (t.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()) if raw else t.empty(0, dtype=t.uint8), t.from_numpy(offs), len(mine), len(recs))

    def stage_release(self):
        pass

    def empty_like(self, t, n):
        return self.torch.empty(n, dtype=t.dtype)

    # ---- sorted-batch protocol: same contracts as cblx_sorted_batch_* / cblx_insert_sorted_batches_device ------------
    def suffix_bytes(self):
        return (self.sb + 7) // 8

    def sorted_batch_begin(self, bases, offsets, n, bounds, nd):
        lo, hi = self.seq_words(bases, offsets, n)
        words = self._to_ints(lo, hi)
        order = sorted(range(len(words)), key=lambda i: words[i] >> self.sb)  # stable: stream order inside a prefix
        self._batch = [words[i] for i in order]
        from collections import Counter

        tally = Counter(w >> self.sb for w in self._batch)
        uniq = sorted(tally)
        self._prefix = uniq
        self._count = [tally[p] for p in uniq]
        bs, ws = [0], [0]
        for d in range(1, nd):
            k = int(np.searchsorted(np.array(uniq, dtype=np.int64), int(bounds[d - 1]), side="left"))
            bs.append(k)
            ws.append(sum(self._count[:k]))
        bs.append(len(uniq))
        ws.append(len(words))
        return bs, ws

    def sorted_batch_export(self, n_buckets, n_words):
        t = self.torch
        B = self.suffix_bytes()
        assert n_buckets == len(self._prefix) and n_words == len(self._batch)
        mask = (1 << self.sb) - 1
        raw = b"".join((w & mask).to_bytes(B, "little") for w in self._batch)
        return (t.tensor(self._prefix, dtype=t.int32), t.tensor(self._count, dtype=t.int32),
                t.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()) if raw else t.empty(0, dtype=t.uint8))

    def insert_sorted_batches(self, batches):
        B = self.suffix_bytes()
        for nb, nw, prefix, count, suffix in batches:
            raw = suffix.numpy().tobytes()
            words, k = [], 0
            for p, c in zip(prefix.tolist(), count.tolist()):
                for _ in range(c):
                    words.append(((p & 0xFFFFFFFF) << self.sb) | int.from_bytes(raw[k * B : (k + 1) * B], "little"))
                    k += 1
            assert k == nw
            self.o.insert_words(words)


class _CblAttrs:  # what ShardedBuilder reads from a CBL
    def __init__(self, k, pb):
        self.k, self.prefix_bits = k, pb


class OracleShard:
    """CPU stand-in for cbl_amd.sharded.GpuShard: one rank's share held by a CPU oracle. Bucket batches go through the
    serialized form (kinds and stored order are exactly what that form carries)."""

    def __init__(self, k, pb, canonical=False):
        import torch
        from oracle import Oracle
        from oracle.pyref import params

        self.torch = torch
        self.k, self.pb, self.canonical = k, pb, canonical
        self.P = params(k, pb)
        self.o = Oracle(k, pb, canonical)
        self.cbl = _CblAttrs(k, pb)

    def new_like(self, profile=False):
        return OracleShard(self.k, self.pb, self.canonical)

    def builder_engine(self):
        return OracleEngine(self.o, self.k, self.pb)

    def clear(self):
        from oracle import Oracle

        self.o = Oracle(self.k, self.pb, self.canonical)

    def count(self):
        return self.o.count()

    def is_canonical(self):
        return self.canonical

    def suffix_bytes(self):
        return self.P["BYTES"]

    def _entries(self):
        return parse_index(self.o.serialize(), self.P["BYTES"])[1]

    def _set_entries(self, entries):
        from oracle import Oracle

        self.o = Oracle(self.k, self.pb, self.canonical)
        self.o.load(build_index(int(self.canonical), entries, self.P["BYTES"]))

    def load_shard(self, path, rank, world, bounds, sequential):
        """Sequential semantics of cblx_load_shard_from_file (include/cblx.h)."""
        blob = open(path, "rb").read()
        canonical, entries = parse_index(blob, self.P["BYTES"])
        self.canonical = bool(canonical)
        body = 1 + len(_wv(len(entries)))
        length = len(blob) - body
        none = 1 << self.pb
        starts, firsts = [len(entries)] * (world + 1), [none] * (world + 1)
        starts[0] = 0
        r = 1
        for i, (p, _k, _it, s, _e) in enumerate(entries):
            while r < world and (p >= bounds[r - 1] if bounds is not None else (s - body) >= length // world * r):
                starts[r], firsts[r] = i, p
                r += 1
        mine = entries[starts[rank] : starts[rank + 1]]
        self._set_entries(mine)
        info = {"header_entries": len(entries), "local_entries": len(mine), "exact": 1, "canonical": canonical,
                "first_prefix": mine[0][0] if mine else 0, "last_prefix": mine[-1][0] if mine else 0}
        b = np.asarray(bounds if bounds is not None else firsts[1:world], dtype=np.uint32)
        return info, b

    def split(self, bounds, nd):
        ents = self._entries()
        bs, ws = [0], [0]
        for d in range(1, nd):
            k = sum(1 for e in ents if e[0] < int(bounds[d - 1]))
            bs.append(k)
            ws.append(sum(len(e[2]) for e in ents[:k]))
        bs.append(len(ents))
        ws.append(sum(len(e[2]) for e in ents))
        return bs, ws

    def export(self):
        t, B = self.torch, self.P["BYTES"]
        ents = self._entries()
        raw = b"".join(s.to_bytes(B, "little") for e in ents for s in e[2])
        return (t.tensor([e[0] for e in ents], dtype=t.int32), t.tensor([len(e[2]) for e in ents], dtype=t.int32),
                t.tensor([e[1] for e in ents], dtype=t.uint8), t.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()) if raw else t.empty(0, dtype=t.uint8))

    def empty_like(self, t, n):
        return self.torch.empty(n, dtype=t.dtype)

    def install(self, parts):
        B = self.P["BYTES"]
        ents = []
        for nb, nw, prefix, count, kind, suffix in parts:
            raw = suffix.numpy().tobytes()
            k = 0
            for p, c, kd in zip(prefix.tolist(), count.tolist(), kind.tolist()):
                ents.append((p & 0xFFFFFFFF, kd, [int.from_bytes(raw[(k + j) * B : (k + j + 1) * B], "little") for j in range(c)]))
                k += c
            assert k == nw
        assert all(ents[i][0] < ents[i + 1][0] for i in range(len(ents) - 1))
        self._set_entries(ents)

    def merge_assign(self, other):
        self.o.merge(other.o)

    def _body(self):
        blob = self.o.serialize()
        n, p = _rv(blob, 1)
        return n, blob[p:]

    def body_size(self):
        n, body = self._body()
        return n, len(body)

    def write_body_at(self, path, off):
        _n, body = self._body()
        with open(path, "r+b") as f:
            f.seek(off)
            f.write(body)
