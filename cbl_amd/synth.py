"""Deterministic synthetic reads: iid uniform ACGT, splitmix64, 2 bits per base.

Base i of the stream comes from 64-bit output j = i // 32 of splitmix64(seed): bits [2*(i%32), 2*(i%32)+2)
select "ACTG"[code] (the reference's code order, /root/reference/src/kmer.rs:11). Output j is
mix(seed + (j+1)*0x9E3779B97F4A7C15), so any slice of the stream can be generated independently
(ranks generate their own shard). Read r is stream bases [r*L, (r+1)*L).
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_LUT = np.frombuffer(b"ACTG", dtype=np.uint8)


def splitmix64_at(seed: int, j: np.ndarray) -> np.ndarray:
    """j-th output (j >= 0, uint64 array) of splitmix64 seeded with `seed`."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (j.astype(np.uint64) + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def stream_bases(seed: int, start: int, count: int) -> np.ndarray:
    """ASCII bases [start, start+count) of the stream, as a uint8 array."""
    if count == 0:
        return np.zeros(0, dtype=np.uint8)
    j0, j1 = start // 32, (start + count + 31) // 32
    out = np.empty((j1 - j0) * 32, dtype=np.uint8)
    step = 1 << 20
    shifts = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    for a in range(j0, j1, step):
        b = min(a + step, j1)
        z = splitmix64_at(seed, np.arange(a, b, dtype=np.uint64))
        codes = ((z[:, None] >> shifts) & np.uint64(3)).astype(np.uint8)
        out[(a - j0) * 32 : (b - j0) * 32] = _LUT[codes.reshape(-1)]
    off = start - j0 * 32
    return out[off : off + count]


def reads(seed: int, n_reads: int, read_len: int, first_read: int = 0):
    """(bases uint8[n*L], offsets uint64[n+1]) for reads [first_read, first_read+n_reads)."""
    bases = np.ascontiguousarray(stream_bases(seed, first_read * read_len, n_reads * read_len))
    offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
    return bases, offsets


def fasta_bytes(bases: np.ndarray, offsets: np.ndarray) -> bytes:
    """Single-line FASTA (`>r<i>\\n<bases>\\n`)."""
    parts = []
    for i in range(len(offsets) - 1):
        parts.append(b">r%d\n" % i)
        parts.append(bases[int(offsets[i]) : int(offsets[i + 1])].tobytes())
        parts.append(b"\n")
    return b"".join(parts)


def reads_torch(seed: int, n_reads: int, read_len: int, first_read: int = 0, device="cuda"):
    """Same stream as `reads`, generated on `device` with torch integer ops (bench start-up: no 1.5 GB H2D copy).

    Returns (bases uint8[n*L (+16 pad)], offsets int64[n+1]); the pad keeps 16-byte loads in bounds.
    """
    import torch

    def lsr(z, s):  # logical shift right on int64
        return (z >> s) & ((1 << (64 - s)) - 1)

    def s64(v):  # python int -> signed 64-bit
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >= (1 << 63) else v

    start, count = first_read * read_len, n_reads * read_len
    j0, j1 = start // 32, (start + count + 31) // 32
    out = torch.empty((j1 - j0) * 32 + 16, dtype=torch.uint8, device=device)
    out[-16:] = 0
    lut = torch.tensor(list(b"ACTG"), dtype=torch.uint8, device=device)
    shifts = (torch.arange(32, dtype=torch.int64, device=device) * 2)[None, :]
    step = 1 << 22
    for a in range(j0, j1, step):
        b = min(a + step, j1)
        j = torch.arange(a, b, dtype=torch.int64, device=device)
        z = (j + 1) * s64(0x9E3779B97F4A7C15) + s64(seed)
        z = (z ^ lsr(z, 30)) * s64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * s64(0x94D049BB133111EB)
        z = z ^ lsr(z, 31)
        codes = (z[:, None] >> shifts) & 3
        out[(a - j0) * 32 : (b - j0) * 32] = lut[codes.reshape(-1)]
    off = start - j0 * 32
    if off:
        out = out[off:].clone()
    offsets = torch.arange(n_reads + 1, dtype=torch.int64, device=device) * read_len
    return out[: count + 16], offsets
