"""CPU oracle for the bulk k-mer insertion path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package. The product
(cbl_amd/, include/cblx.h) never does; it fails loudly when its HIP library is missing.

`Oracle` wraps liboracle.so (oracle/cbl_oracle.hpp, a C++ restatement of the reference's CPU algorithm, see
the citations in that header). The library is built with `-O3 -march=native` for the machine it runs on
(keyed by the CPU's flag set, so a copy built in the authoring container is never executed on a GPU box with
a different host CPU).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
from pathlib import Path

_HERE = Path(__file__).resolve().parent
_LIB = None


def _cpu_key() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return hashlib.sha1(line.encode()).hexdigest()[:12]
    except OSError:
        pass
    return "generic"


def lib_path() -> Path:
    return _HERE / "_build" / _cpu_key() / "liboracle.so"


def build(force: bool = False) -> Path:
    out = lib_path()
    srcs = [_HERE / "cbl_oracle_capi.cpp", _HERE / "cbl_oracle.hpp"]
    if not force and out.exists() and all(out.stat().st_mtime >= s.stat().st_mtime for s in srcs):
        return out
    out.parent.mkdir(parents=True, exist_ok=True)
    tmp = out.with_suffix(".tmp%d" % os.getpid())
    cmd = ["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-shared", "-o", str(tmp), str(srcs[0])]
    subprocess.run(cmd, check=True, cwd=str(_HERE))
    os.replace(tmp, out)
    return out


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        L = C.CDLL(str(build()))
        u64, u32, vp, i32 = C.c_uint64, C.c_uint32, C.c_void_p, C.c_int
        pu64, pu8 = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
        sig = {
            "oracle_last_error": (C.c_char_p, []),
            "oracle_create": (vp, [u32, u32, i32]),
            "oracle_destroy": (None, [vp]),
            "oracle_insert_seq": (i32, [vp, C.c_char_p, u64]),
            "oracle_insert_seqs": (i32, [vp, vp, vp, u64, C.POINTER(C.c_double)]),
            "oracle_insert_words": (i32, [vp, vp, vp, u64]),
            "oracle_count": (u64, [vp]),
            "oracle_n_buckets": (u64, [vp]),
            "oracle_serialize": (i32, [vp, C.POINTER(pu8), pu64]),
            "oracle_free": (None, [vp]),
            "oracle_load": (i32, [vp, C.c_char_p, u64]),
            "oracle_merge": (i32, [vp, vp]),
            "oracle_seq_words": (C.c_int64, [vp, C.c_char_p, u64, i32, vp, vp, u64]),
            "oracle_insert_kmer": (i32, [vp, u64, u64]),
            "oracle_contains_kmer": (i32, [vp, u64, u64]),
            "oracle_contains_word": (i32, [vp, u64, u64]),
            "oracle_word_of_kmer": (None, [vp, u64, u64, pu64, pu64]),
            "oracle_kmer_of_word": (None, [vp, u64, u64, pu64, pu64]),
            "oracle_iter_words": (u64, [vp, vp, vp, u64]),
            "oracle_necklace_pos": (None, [u64, u64, u32, pu64, pu64, C.POINTER(u32)]),
            "oracle_revert_necklace_pos": (None, [u64, u64, u32, u32, pu64, pu64]),
            "oracle_rev_comp": (None, [u64, u64, u32, pu64, pu64]),
            "oracle_nuc_code": (i32, [u32]),
            "oracle_queue_new": (vp, [u32, u32, i32, u64, u64]),
            "oracle_queue_free": (None, [vp]),
            "oracle_queue_insert": (None, [vp, u32]),
            "oracle_queue_insert2": (None, [vp, u32]),
            "oracle_queue_get": (None, [vp, pu64, pu64, C.POINTER(u32)]),
            "oracle_lmq_new": (vp, [u32]),
            "oracle_lmq_free": (None, [vp]),
            "oracle_lmq_insert": (None, [vp, u32]),
            "oracle_lmq_insert2": (None, [vp, u32, u32]),
            "oracle_lmq_insert_full": (None, [vp, C.POINTER(u32)]),
            "oracle_lmq_min_pos": (u32, [vp, C.POINTER(u32), u32]),
            "oracle_bv_new": (vp, [u64]),
            "oracle_bv_free": (None, [vp]),
            "oracle_bv_insert": (i32, [vp, u64]),
            "oracle_bv_contains": (i32, [vp, u64]),
            "oracle_bv_rank": (u64, [vp, u64]),
            "oracle_bv_iter": (u64, [vp, pu64, u64]),
            "oracle_tv_new": (vp, []),
            "oracle_tv_free": (None, [vp]),
            "oracle_tv_insert": (None, [vp, u64, u32]),
            "oracle_tv_get": (u32, [vp, u64]),
            "oracle_tv_len": (u64, [vp]),
            "oracle_trievec_new": (vp, []),
            "oracle_trievec_free": (None, [vp]),
            "oracle_trievec_insert": (i32, [vp, u64, u64, u32]),
            "oracle_trievec_as_trie": (None, [vp, u32]),
            "oracle_trievec_contains": (i32, [vp, u64, u64, u32]),
            "oracle_trievec_iter": (u64, [vp, u32, vp, vp, u64]),
            "oracle_ws_new": (vp, [u32, u32]),
            "oracle_ws_free": (None, [vp]),
            "oracle_ws_insert": (i32, [vp, u64]),
            "oracle_ws_insert_batch": (None, [vp, vp, u64]),
            "oracle_ws_contains": (i32, [vp, u64]),
            "oracle_ws_count": (u64, [vp]),
            "oracle_ws_iter": (u64, [vp, vp, u64]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


class OracleError(RuntimeError):
    pass


def _split(x: int):
    return x & 0xFFFFFFFFFFFFFFFF, (x >> 64) & 0xFFFFFFFFFFFFFFFF


class Oracle:
    """CPU restatement of `CBL<K, T, PREFIX_BITS>` (src/cbl.rs) — same method names as the reference."""

    def __init__(self, k: int, prefix_bits: int = 24, canonical: bool = False):
        self._L = lib()
        self.k, self.prefix_bits, self.canonical = k, prefix_bits, canonical
        self._h = self._L.oracle_create(k, prefix_bits, int(canonical))
        if not self._h:
            raise OracleError(self._L.oracle_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.oracle_destroy(self._h)
            self._h = None

    def _chk(self, rc):
        if rc != 0:
            raise OracleError(self._L.oracle_last_error().decode())

    def insert_seq(self, seq: bytes):
        self._chk(self._L.oracle_insert_seq(self._h, seq, len(seq)))

    def insert_seqs(self, bases, offsets) -> float:
        """bases: numpy uint8 array; offsets: numpy uint64 array of n+1 starts. Returns seconds spent."""
        import numpy as np

        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        secs = C.c_double(0.0)
        self._chk(self._L.oracle_insert_seqs(self._h, bases.ctypes.data, offsets.ctypes.data, len(offsets) - 1, C.byref(secs)))
        return secs.value

    def insert_words(self, words):
        """WordSet::insert_batch on already transformed words (list of Python ints)."""
        import numpy as np

        lo = np.array([w & 0xFFFFFFFFFFFFFFFF for w in words], dtype=np.uint64)
        hi = np.array([w >> 64 for w in words], dtype=np.uint64)
        self._chk(self._L.oracle_insert_words(self._h, lo.ctypes.data, hi.ctypes.data, len(words)))

    def count(self) -> int:
        return self._L.oracle_count(self._h)

    def n_buckets(self) -> int:
        return self._L.oracle_n_buckets(self._h)

    def serialize(self) -> bytes:
        buf = C.POINTER(C.c_uint8)()
        n = C.c_uint64(0)
        self._chk(self._L.oracle_serialize(self._h, C.byref(buf), C.byref(n)))
        try:
            return C.string_at(buf, n.value)
        finally:
            self._L.oracle_free(buf)

    def serialize_np(self):
        """serialize() as a numpy uint8 view of the library's own buffer (no second copy: a full-size index is gigabytes)."""
        import numpy as np
        import weakref

        buf = C.POINTER(C.c_uint8)()
        n = C.c_uint64(0)
        self._chk(self._L.oracle_serialize(self._h, C.byref(buf), C.byref(n)))
        arr = np.ctypeslib.as_array(buf, shape=(max(n.value, 1),))[: n.value]
        weakref.finalize(arr.base if arr.base is not None else arr, self._L.oracle_free, buf)
        return arr

    def load(self, data: bytes):
        self._chk(self._L.oracle_load(self._h, data, len(data)))

    def merge(self, other: "Oracle"):
        """`self |= other` (src/cbl.rs:433-449)."""
        self._chk(self._L.oracle_merge(self._h, other._h))

    def seq_words(self, seq: bytes, brute_force: bool = False):
        """Words of get_seq_words over every chunk of `seq` (list of Python ints)."""
        import numpy as np

        cap = max(len(seq), 1) * 2
        lo = np.zeros(cap, dtype=np.uint64)
        hi = np.zeros(cap, dtype=np.uint64)
        n = self._L.oracle_seq_words(self._h, seq, len(seq), int(brute_force), lo.ctypes.data, hi.ctypes.data, cap)
        if n < 0:
            raise OracleError(self._L.oracle_last_error().decode())
        return [int(lo[i]) | (int(hi[i]) << 64) for i in range(n)]

    def insert_kmer(self, kmer: int) -> bool:
        return bool(self._L.oracle_insert_kmer(self._h, *_split(kmer)))

    def contains_kmer(self, kmer: int) -> bool:
        return bool(self._L.oracle_contains_kmer(self._h, *_split(kmer)))

    def contains_word(self, word: int) -> bool:
        return bool(self._L.oracle_contains_word(self._h, *_split(word)))

    def word_of_kmer(self, kmer: int) -> int:
        lo, hi = C.c_uint64(), C.c_uint64()
        self._L.oracle_word_of_kmer(self._h, *_split(kmer), C.byref(lo), C.byref(hi))
        return lo.value | (hi.value << 64)

    def kmer_of_word(self, word: int) -> int:
        lo, hi = C.c_uint64(), C.c_uint64()
        self._L.oracle_kmer_of_word(self._h, *_split(word), C.byref(lo), C.byref(hi))
        return lo.value | (hi.value << 64)

    def iter_words(self):
        import numpy as np

        cap = self.count()
        lo = np.zeros(max(cap, 1), dtype=np.uint64)
        hi = np.zeros(max(cap, 1), dtype=np.uint64)
        n = self._L.oracle_iter_words(self._h, lo.ctypes.data, hi.ctypes.data, cap)
        return [int(lo[i]) | (int(hi[i]) << 64) for i in range(n)]


def necklace_pos(word: int, bits: int):
    lo, hi, pos = C.c_uint64(), C.c_uint64(), C.c_uint32()
    lib().oracle_necklace_pos(*_split(word), bits, C.byref(lo), C.byref(hi), C.byref(pos))
    return lo.value | (hi.value << 64), pos.value


def revert_necklace_pos(necklace: int, pos: int, bits: int) -> int:
    lo, hi = C.c_uint64(), C.c_uint64()
    lib().oracle_revert_necklace_pos(*_split(necklace), pos, bits, C.byref(lo), C.byref(hi))
    return lo.value | (hi.value << 64)


def rev_comp(x: int, k: int) -> int:
    lo, hi = C.c_uint64(), C.c_uint64()
    lib().oracle_rev_comp(*_split(x), k, C.byref(lo), C.byref(hi))
    return lo.value | (hi.value << 64)
