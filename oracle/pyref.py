"""Independent pure-Python restatement of the reference's bulk-insert path — TEST INFRASTRUCTURE ONLY.

Second, deliberately naive implementation used to cross-check oracle/cbl_oracle.hpp on small inputs
(tests/test_oracle_crosscheck.py). It follows the *normative* definitions only:
  * necklace/pos by brute force           /root/reference/src/necklace/mod.rs:13-25
  * nucleotide code, rolling pack, rc     /root/reference/src/kmer.rs:11-24,61-72,94-96,293-348
  * chunking + stream order               /root/reference/src/cbl.rs:239-289,328-339
  * prefix/suffix split, Vec->Trie @1024  /root/reference/src/wordset/mod.rs:63-71,187-216,240-244
  * file format                           /root/reference/src/wordset/mod.rs:382-396 + derived Serialize impls,
                                          bincode 1.3 DefaultOptions (varint)  -- parity unpinned by reference tests
  * `|=`                                  /root/reference/src/wordset/set_ops.rs:123-157, src/trievec/set_ops.rs:43-71
Pure-Python loops: small cases only.
"""
from __future__ import annotations

CHUNK = 2048
THRESHOLD = 1024
_CODE = {ord("A"): 0, ord("a"): 0, ord("C"): 1, ord("c"): 1, ord("T"): 2, ord("t"): 2, ord("G"): 3, ord("g"): 3}


def params(k: int, pb: int):
    kb = 2 * k
    pos_bits = (kb - 1).bit_length()  # ilog2(next_power_of_two(kb)) for kb >= 2
    word_bits = kb + pos_bits
    sb = max(word_bits - pb, 0)
    return dict(K=k, PB=pb, KB=kb, POS=pos_bits, WB=word_bits, SB=sb, BYTES=(sb + 7) // 8)


def necklace_pos(x: int, bits: int):
    mask = (1 << bits) - 1
    best, bestp = None, 0
    for p in range(bits):
        r = ((x << p) & mask) | (x >> (bits - p))
        if best is None or r < best:
            best, bestp = r, p
    return best, bestp


def rev_comp(x: int, k: int) -> int:
    r = 0
    for _ in range(k):
        r = (r << 2) | ((x & 3) ^ 2)
        x >>= 2
    return r


def chunk_words(chunk: bytes, P, canonical: bool):
    k, kb = P["K"], P["KB"]
    mask = (1 << kb) - 1
    x = 0
    for b in chunk[:k]:
        c = _CODE.get(b)
        if c is not None:
            x = (x << 2) | c
    fwd, rc = [], []

    def emit(x):
        if canonical and bin(x).count("1") % 2 == 1:
            n, p = necklace_pos(rev_comp(x, k), kb)
            rc.append((n << P["POS"]) | p)
        else:
            n, p = necklace_pos(x & mask, kb)
            fwd.append((n << P["POS"]) | p)

    emit(x)
    for b in chunk[k:]:
        c = _CODE.get(b)
        if c is None:
            continue
        x = ((x << 2) | c) & mask
        emit(x)
    return fwd + rc


def seq_words(seq: bytes, P, canonical: bool):
    k = P["K"]
    if len(seq) < k:
        raise ValueError("Sequence size (%d) is smaller than K (%d)" % (len(seq), k))
    out = []
    for start in range(0, len(seq) - k + 1, CHUNK):
        out += chunk_words(seq[start : min(start + CHUNK + k - 1, len(seq))], P, canonical)
    return out


class PyCBL:
    """buckets: prefix -> ("vec", [suffix,...]) in first-occurrence order | ("trie", sorted list)."""

    def __init__(self, k: int, pb: int = 24, canonical: bool = False):
        self.P = params(k, pb)
        self.canonical = canonical
        self.buckets = {}

    def _insert_word(self, w: int):
        sb = self.P["SB"]
        p, s = w >> sb, w & ((1 << sb) - 1)
        kind, items = self.buckets.setdefault(p, ["vec", []])
        if s not in items:
            items.append(s)
        return p

    def insert_seq(self, seq: bytes):
        k = self.P["K"]
        if len(seq) < k:
            raise ValueError("Sequence size (%d) is smaller than K (%d)" % (len(seq), k))
        for start in range(0, len(seq) - k + 1, CHUNK):
            words = chunk_words(seq[start : min(start + CHUNK + k - 1, len(seq))], self.P, self.canonical)
            sb = self.P["SB"]
            i = 0
            while i < len(words):
                j = i
                p = words[i] >> sb
                while j < len(words) and (words[j] >> sb) == p:
                    self._insert_word(words[j])
                    j += 1
                b = self.buckets[p]
                if len(b[1]) > THRESHOLD and b[0] == "vec":  # threshold check after each group
                    b[0] = "trie"
                if b[0] == "trie":
                    b[1].sort()
                i = j

    def count(self) -> int:
        return sum(len(b[1]) for b in self.buckets.values())

    def merge(self, other: "PyCBL"):
        assert self.canonical == other.canonical
        for p in sorted(other.buckets):
            okind, oitems = other.buckets[p]
            if p not in self.buckets:  # other-only: cloned as stored (no sort)
                self.buckets[p] = [okind, list(oitems)]
                continue
            if okind == "vec":
                oitems.sort()  # iter_sorted sorts other's Vec in place (both-sides buckets only)
            b = self.buckets[p]
            if b[0] == "vec":
                b[1].sort()  # ... and self's
                mine = set(b[1])
                b[1] += [s for s in oitems if s not in mine]  # pushed at the end, no threshold check
            else:
                b[1] = sorted(set(b[1]) | set(oitems))

    # ---------------------------------------------------------------- file format
    @staticmethod
    def _varint(v: int) -> bytes:
        if v <= 250:
            return bytes([v])
        if v < 1 << 16:
            return b"\xfb" + v.to_bytes(2, "little")
        if v < 1 << 32:
            return b"\xfc" + v.to_bytes(4, "little")
        return b"\xfd" + v.to_bytes(8, "little")

    def _trie(self, items, depth: int) -> bytes:
        by = self.P["BYTES"]
        shift = 8 * (by - 1 - depth)
        groups = {}
        for s in items:
            groups.setdefault((s >> shift) & 0xFF, []).append(s)
        keys = sorted(groups)
        out = self._varint(len(keys)) + bytes(keys)
        if depth == by - 1:
            return out + self._varint(0)
        out += self._varint(len(keys))
        for key in keys:
            out += self._trie(groups[key], depth + 1)
        return out

    def serialize(self) -> bytes:
        by = self.P["BYTES"]
        out = bytearray([1 if self.canonical else 0])
        out += self._varint(len(self.buckets))
        for p in sorted(self.buckets):
            kind, items = self.buckets[p]
            out += self._varint(p)
            if kind == "vec":
                out += self._varint(0) + self._varint(len(items))
                for s in items:
                    out += self._varint(by) + s.to_bytes(by, "little")
            else:
                out += self._varint(1) + self._trie(sorted(items), 0) + self._varint(len(items))
        return bytes(out)
