// kernels_serde.hpp — the serde/bincode index bytes emitted on the device.
//
// Reference: `WordSet::serialize` /root/reference/src/wordset/mod.rs:382-396 (map of prefix -> TrieVec, ascending
// prefixes), `TrieOrVec` derive src/trievec/mod.rs:8-15, `Trie/TrieNode` derive src/trie.rs:8-9,53-57, `TinyBitvector`
// src/bitvector/tiny/mod.rs:97-105 (popcount + set indices), `SlicedInt::serialize` src/sliced_int.rs:110-114, bincode
// 1.3 varint options src/cbl.rs:132-135. A bucket entry is a byte range that depends on that bucket only, so: pass 1
// sizes every entry, an exclusive scan places them, pass 2 writes them.
//
// Trie entry as a data-parallel computation over the bucket's sorted suffixes x[0..n):
//   lcp[j]  = number of leading (big-endian) bytes x[j] shares with x[j-1]  (lcp[0] = -1)
//   a node at level d starts at j   <=>  lcp[j] <  d      (x[j] is the first suffix with its d-byte prefix)
//   x[j] adds a child at level d    <=>  lcp[j] <= d      (first suffix with its (d+1)-byte prefix)
//   so the nodes of level d are the children of level d-1, in the same order, and with X_d = exclusive rank of the
//   level-d children: c(node k) = X_d[start of node k+1] - X_d[start of node k].
//   Pre-order = nodes sorted by (start j, level d): position = sum of the header sizes of all earlier (j, d) pairs;
//   header = varint(c) | c child bytes | varint(#children) where child bytes are written by the children themselves.
#pragma once
#include <type_traits>

#include "kernels_bucket.hpp"

namespace cblx {

__device__ __forceinline__ u32 vlen(u64 v) { return v <= 250 ? 1u : v < (1ull << 16) ? 3u : v < (1ull << 32) ? 5u : 9u; }
__device__ __forceinline__ void put_varint(u8* p, u64 v) {
    if (v <= 250) { p[0] = (u8)v; return; }
    const int nb = v < (1ull << 16) ? 2 : v < (1ull << 32) ? 4 : 8;
    p[0] = nb == 2 ? 0xFB : nb == 4 ? 0xFC : 0xFD;
    for (int i = 0; i < nb; ++i) p[1 + i] = (u8)(v >> (8 * i));
}

template <bool WS> __device__ __forceinline__ Sfx<WS> arena_sfx(const u64* __restrict__ a_lo, const u64* __restrict__ a_hi, u64 i, u32 SB) {
    // arena slots of buckets a rebuild did not touch still carry the full word: always mask to SB bits
    Sfx<WS> s;
    if constexpr (WS) {
        s.lo = a_lo[i];
        s.hi = a_hi[i] & ((1ull << (SB - 64)) - 1ull);
    } else {
        s.lo = SB >= 64 ? a_lo[i] : (a_lo[i] & ((1ull << SB) - 1ull));
    }
    return s;
}
template <bool WS> __device__ __forceinline__ u32 sfx_byte_le(const Sfx<WS>& s, u32 k) {
    if constexpr (WS) return k < 8 ? (u32)(s.lo >> (8 * k)) & 255u : (u32)(s.hi >> (8 * (k - 8))) & 255u;
    else return (u32)(s.lo >> (8 * k)) & 255u;
}
// little-endian index of the most significant byte where a and b differ (a != b)
template <bool WS> __device__ __forceinline__ u32 top_diff_byte(const Sfx<WS>& a, const Sfx<WS>& b) {
    if constexpr (WS) {
        const u64 xh = a.hi ^ b.hi;
        if (xh) return 8u + (63u - (u32)__builtin_clzll(xh)) / 8u;
    }
    return (63u - (u32)__builtin_clzll(a.lo ^ b.lo)) / 8u;
}

enum { SER_C64 = 0, SER_C256 = 1, SER_C1024 = 2, SER_HOST = 3, SER_SPLIT = 4, SER_NCLS = 5 };
// The lists of the three workgroup classes are kept per CHUNK of consecutive buckets (bucket r belongs to chunk r / per): the
// emitter runs group of chunks by group of chunks (groups of about equal bytes, a small one first) so that the download of a
// group's bytes hides the emission of the next. Sub-list (class, chunk k)
// starts at lists[class * nb + k * per] (a chunk holds at most `per` buckets), its length is list_n[class * SER_CHUNKS + k]; the
// host and split classes use chunk 0 only.
static const u32 SER_CHUNKS = 32;
// A Trie longer than one workgroup takes (SER_CAP1024) is cut at its ROOT: the children of the root are the distinct top bytes,
// the sub-trie under each is the trie of the (contiguous, sorted) sub-range that shares the byte, and pre-order puts the
// sub-tries one after the other behind the root's header — so every sub-range of <= SER_CAP1024 words is emitted by the same
// workgroup kernel started at level 1 ("virtual bucket"), and the entry is header | root | sub-tries | varint(len).
static const u32 SER_SPLIT_MAX = 1u << 21;
static const u32 SER_CAP64 = 64 * 16, SER_CAP256 = 256 * 16, SER_CAP1024 = 1024 * 8;  // bucket lengths per workgroup shape
static const u32 SER_TINY = 32;  // Vec buckets up to this length are written by one thread each

// Pass over all buckets, one thread each: tiny Vec buckets are sized (EMIT = false) or written (EMIT = true) here; the
// others are appended to the list of their size class (sizing pass only; unordered).
template <bool WS, bool EMIT>
__global__ __launch_bounds__(CLASSIFY_THREADS) void k_serde_tiny(u64 nb, const u32* __restrict__ prefix, const u64* __restrict__ start, const u32* __restrict__ cnt,
                             const u8* __restrict__ kind, const u64* __restrict__ a_lo, const u64* __restrict__ a_hi, u32 SB, u32 BYTES,
                             u32* __restrict__ size, const u64* __restrict__ off, u8* __restrict__ out, u32* __restrict__ lists,
                             u32* __restrict__ list_n, u64 per /* buckets per chunk */) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = r < nb;
    const u32 n = live ? cnt[r] : 0;
    const bool tiny = live && kind[r] == KIND_VEC && n <= SER_TINY;
    if (tiny) {
        const u32 pfx = prefix[r];
        const u32 hdr = vlen(pfx) + 1 + vlen(n);
        if constexpr (!EMIT) {
            size[r] = hdr + n * (1 + BYTES);
        } else {
            u8* o = out + off[r];
            put_varint(o, pfx);
            o[vlen(pfx)] = 0;  // TrieOrVec::Vec
            put_varint(o + vlen(pfx) + 1, n);
            o += hdr;
            const u64 s0 = start[r];
            for (u32 j = 0; j < n; ++j) {
                const Sfx<WS> s = arena_sfx<WS>(a_lo, a_hi, s0 + j, SB);
                o[0] = (u8)BYTES;
                for (u32 k = 0; k < BYTES; ++k) o[1 + k] = (u8)sfx_byte_le<WS>(s, k);
                o += 1 + BYTES;
            }
        }
    }
    if constexpr (!EMIT) {
        int cls = -1;
        if (live && !tiny)
            cls = n <= SER_CAP64 ? SER_C64 : n <= SER_CAP256 ? SER_C256 : n <= SER_CAP1024 ? SER_C1024 : (kind[r] == KIND_TRIE && n <= SER_SPLIT_MAX && BYTES > 1) ? SER_SPLIT : SER_HOST;
        const u32 ch = (cls == SER_C64 || cls == SER_C256 || cls == SER_C1024) ? (u32)(r / per) : 0u;
        const u32 slot = block_append<CLASSIFY_THREADS, SER_NCLS * SER_CHUNKS>(cls >= 0 ? cls * (int)SER_CHUNKS + (int)ch : -1, list_n);
        if (cls >= 0) lists[(u64)cls * nb + (u64)ch * per + slot] = (u32)r;
    }
}

// One workgroup per listed bucket (n <= THREADS * ITEMS).
// SUB: the listed ids are sub-ranges of split Tries (start / cnt = the sub-range's arena start and length, `size` / `off` its
// own byte size and absolute output offset): levels 1.. of the sub-range only, no entry header, no length field.
template <int THREADS, int ITEMS, bool WS, bool EMIT, bool SUB = false>
__global__ __launch_bounds__(THREADS) void k_serde_bucket(const u32* __restrict__ list, const u32* __restrict__ list_n, const u32* __restrict__ prefix,
                                                          const u64* __restrict__ start, const u32* __restrict__ cnt, const u8* __restrict__ kind,
                                                          const u64* __restrict__ a_lo, const u64* __restrict__ a_hi, u32 SB, u32 BYTES,
                                                          u32* __restrict__ size, const u64* __restrict__ off, u8* __restrict__ out) {
    constexpr int NW = THREADS / 64, EPW = 64 * ITEMS, CAP = THREADS * ITEMS;
    constexpr u32 D0 = SUB ? 1u : 0u;            // first level this workgroup emits
    // EMIT: the entry is assembled in LDS and leaves in aligned 16-byte stores (the bytes of an entry are written one by one, a
    // node header here, a child byte there: as global stores that is one memory transaction per byte — 86 ms for the 9.4 GB of
    // cfg 2's index). 13 bytes per element cover a Trie of random suffixes (12.4: three bytes per node on the sparse levels); a longer entry is written
    // to its place directly, as before.
    constexpr u32 STAGE = EMIT ? (u32)CAP * 13u + 64u : 0u;
    __shared__ u32 s_wtot[NW + 1];
    __shared__ u16 s_ns[CAP + 2];                // X_d at the start of node k (+ sentinel)
    __shared__ u32 s_np[EMIT ? CAP : 1];         // where node k's child bytes start (relative to the entry)
    __shared__ __attribute__((aligned(16))) u8 s_stage[EMIT ? STAGE + 16 : 16];
    if (blockIdx.x >= *list_n) return;
    const u32 r = list[blockIdx.x];
    const u32 n = cnt[r], pfx = SUB ? 0u : prefix[r];
    const u64 s0 = start[r];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const u32 hdr = SUB ? 0u : vlen(pfx) + 1;
    u8* const gdst = EMIT ? out + off[r] : nullptr;
    const u32 esz = EMIT ? size[r] : 0u;                                   // the sizing pass's figure for this entry
    const u32 mis = EMIT ? (u32)(reinterpret_cast<uintptr_t>(gdst) & 15u) : 0u;  // staged at the same misalignment as its place
    const bool staged = EMIT && esz + mis <= STAGE;
    // one byte of the entry (position relative to its start): into the staging buffer (an LDS store: a flat pointer that may or
    // may not point into LDS makes every later LDS read wait for the stores before it), or to its place
    auto st = [&](u32 pos, u32 v) {
        if constexpr (EMIT) { if (staged) s_stage[mis + pos] = (u8)v; else gdst[pos] = (u8)v; }
    };
    auto st_varint = [&](u32 pos, u32 v) {  // bincode varint of a value below 2^32
        if (v <= 250u) { st(pos, v); return; }
        if (v < 65536u) { st(pos, 0xFBu); st(pos + 1, v & 255u); st(pos + 2, v >> 8); return; }
        st(pos, 0xFCu);
        for (u32 i = 0; i < 4; ++i) st(pos + 1 + i, (v >> (8 * i)) & 255u);
    };
    auto flush_stage = [&]() {  // (every thread of the workgroup)
        if constexpr (EMIT) {
            if (!staged) return;
            __syncthreads();
            u8* const gbase = gdst - mis;
            const u32 span = mis + esz, nch = (span + 15u) >> 4;
            for (u32 k = tid; k < nch; k += THREADS) {
                const u32 b0 = k << 4;
                if (b0 >= mis && b0 + 16u <= span) {
                    *reinterpret_cast<uint4*>(gbase + b0) = *reinterpret_cast<const uint4*>(s_stage + b0);
                } else {  // the first / last chunk: only the entry's own bytes (the neighbours' are theirs)
                    for (u32 b = b0 < mis ? mis : b0; b < b0 + 16u && b < span; ++b) gbase[b] = s_stage[b];
                }
            }
        }
    };
    if (!SUB && kind[r] == KIND_VEC) {           // varint(n) then n x (varint(BYTES) | BYTES little-endian bytes), stored order
        if constexpr (!EMIT) {
            if (tid == 0) size[r] = hdr + vlen(n) + n * (1 + BYTES);
        } else {
            if (tid == 0) { st_varint(0, pfx); st(hdr - 1, 0); st_varint(hdr, n); }
            const u32 body = hdr + vlen(n);
            for (u32 j = tid; j < n; j += THREADS) {
                const Sfx<WS> s = arena_sfx<WS>(a_lo, a_hi, s0 + j, SB);
                const u32 p = body + j * (1 + BYTES);
                st(p, BYTES);
                for (u32 k = 0; k < BYTES; ++k) st(p + 1 + k, sfx_byte_le<WS>(s, k));
            }
            flush_stage();
        }
        return;
    }
    // ---- Trie(root, len): pre-order nodes then varint(n)
    Sfx<WS> x[ITEMS];
    int lcp[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const u32 e = w * EPW + i * 64 + lane;
        lcp[i] = 127;  // slots past the end never flag
        x[i].lo = 0;
        if constexpr (WS) x[i].hi = 0;
        if (e < n) {
            x[i] = arena_sfx<WS>(a_lo, a_hi, s0 + e, SB);
            if (e == 0) lcp[i] = (int)D0 - 1;   // the first word opens a node at every level this workgroup emits
            else {
                const Sfx<WS> p = arena_sfx<WS>(a_lo, a_hi, s0 + e - 1, SB);
                lcp[i] = (int)(BYTES - 1) - (int)top_diff_byte<WS>(x[i], p);
            }
        }
    }
    // exclusive rank (element order) of a flag; returns the number of flags
    auto rank_flags = [&](const bool (&f)[ITEMS], u32 (&xr)[ITEMS]) -> u32 {
        u32 run = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const u64 bal = __ballot(f[i]);
            xr[i] = run + mbcnt(bal);
            run += (u32)__builtin_popcountll(bal);
        }
        if constexpr (NW == 1) return run;
        __syncthreads();
        if (lane == 0) s_wtot[w] = run;
        __syncthreads();
        u32 base = 0, tot = 0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) { const u32 t = s_wtot[ww]; if ((u32)ww < w) base += t; tot += t; }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) xr[i] += base;
        return tot;
    };
    // one walk over the levels; EM = false accumulates header bytes per element into acc, EM = true writes them at base + acc
    auto levels = [&](auto em_tag, u32 (&acc)[ITEMS], const u32 (&base)[ITEMS]) {
        constexpr bool EM = decltype(em_tag)::value;
        bool tprev[ITEMS], tcur[ITEMS];
        u32 xprev[ITEMS], xcur[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) tprev[i] = lcp[i] < (int)D0;
        u32 nodes = rank_flags(tprev, xprev);  // level D0: the root (of the sub-range)
        for (u32 d = D0; d < BYTES; ++d) {
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) tcur[i] = lcp[i] <= (int)d;
            const u32 tot = rank_flags(tcur, xcur);
#pragma unroll
            for (int i = 0; i < ITEMS; ++i)
                if (tprev[i]) s_ns[xprev[i]] = (u16)xcur[i];
            if (tid == 0) s_ns[nodes] = (u16)tot;
            __syncthreads();
            const bool leaf = d + 1 == BYTES;
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) {
                if (tprev[i]) {
                    const u32 c = (u32)s_ns[xprev[i] + 1] - (u32)s_ns[xprev[i]];
                    const u32 hs = vlen(c) + c + (leaf ? 1u : vlen(c));
                    if constexpr (EM) {
                        const u32 p = base[i] + acc[i];
                        st_varint(p, c);
                        st_varint(p + vlen(c) + c, leaf ? 0u : c);
                        s_np[xprev[i]] = p + vlen(c);
                    }
                    acc[i] += hs;
                }
            }
            if constexpr (EM) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < ITEMS; ++i) {
                    if (tcur[i]) {
                        const u32 kp = xprev[i] + (tprev[i] ? 1u : 0u) - 1u;  // the node this child belongs to
                        st(s_np[kp] + (xcur[i] - (u32)s_ns[kp]), sfx_byte_le<WS>(x[i], BYTES - 1 - d));
                    }
                }
            }
            __syncthreads();  // s_ns / s_np are rewritten by the next level
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) { tprev[i] = tcur[i]; xprev[i] = xcur[i]; }
            nodes = tot;
        }
    };
    u32 E[ITEMS], base[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) { E[i] = 0; base[i] = 0; }
    levels(std::false_type(), E, base);
    // element-order exclusive scan of the per-element header bytes
    u32 run = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const u32 inc = wave_inclusive_scan(E[i]);
        base[i] = run + inc - E[i];
        run += __shfl(inc, 63, 64);
    }
    __syncthreads();
    if (lane == 0) s_wtot[w] = run;
    __syncthreads();
    u32 wbase = 0, total = 0;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) { const u32 t = s_wtot[ww]; if ((u32)ww < w) wbase += t; total += t; }
    if constexpr (!EMIT) {
        if (tid == 0) size[r] = SUB ? total : hdr + total + vlen(n);
    } else {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { base[i] += wbase + hdr; E[i] = 0; }
        if constexpr (!SUB) if (tid == 0) { st_varint(0, pfx); st(hdr - 1, 1); st_varint(hdr + total, n); }  // TrieOrVec::Trie(.., len)
        levels(std::true_type(), E, base);
        flush_stage();
    }
}

// ---- split Tries: plan (children of the root = sub-ranges), size, emit of header / root / length ----------------------------
// One workgroup per split bucket, thread t = top byte value t: the sub-range [lower_bound(t), lower_bound(t + 1)). Sub-ranges
// go to the list of their workgroup shape; a bucket with a sub-range no shape takes is handed to the host emitter.
template <bool WS>
__global__ __launch_bounds__(256) void k_serde_split_plan(const u32* __restrict__ list, const u32* __restrict__ list_n, const u64* __restrict__ start,
                                                          const u32* __restrict__ cnt, const u64* __restrict__ a_lo, const u64* __restrict__ a_hi, u32 SB,
                                                          u32 BYTES, u64 nb, u64* __restrict__ vstart, u32* __restrict__ vcnt, u32* __restrict__ vsize,
                                                          u32* __restrict__ vlists, u64 vcap, u32* __restrict__ vlist_n, u32* __restrict__ bad,
                                                          u32* __restrict__ host_list, u32* __restrict__ host_n) {
    __shared__ u32 s_lb[257];
    __shared__ u32 s_bad;
    const u32 i = blockIdx.x, t = threadIdx.x;
    int cls = -1;
    if (i < *list_n) {
        const u32 r = list[i], n = cnt[r];
        const u64 s0 = start[r];
        u32 lo = 0, hi = n;  // first element whose top byte is >= t
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (sfx_byte_le<WS>(arena_sfx<WS>(a_lo, a_hi, s0 + mid, SB), BYTES - 1) < t) lo = mid + 1; else hi = mid;
        }
        s_lb[t] = lo;
        if (t == 0) { s_lb[256] = n; s_bad = 0; }
        __syncthreads();
        const u32 c = s_lb[t + 1] - s_lb[t];
        const u64 v = (u64)i * 256 + t;
        vstart[v] = s0 + s_lb[t];
        vcnt[v] = c;
        vsize[v] = 0;
        if (c > SER_CAP1024) atomicOr(&s_bad, 1u);
        __syncthreads();
        if (s_bad) {
            if (t == 0) { bad[i] = 1; host_list[atomicAdd(host_n, 1u)] = r; }
        } else {
            if (t == 0) bad[i] = 0;
            if (c) cls = c <= SER_CAP64 ? 0 : c <= SER_CAP256 ? 1 : 2;
        }
    }
    const u32 slot = block_append<256, 3>(cls, vlist_n);
    if (cls >= 0) vlists[(u64)cls * vcap + slot] = (u32)((u64)i * 256 + t);
    (void)nb;
}
// size of a split entry: header | varint(c0) c0 bytes varint(c0) | sub-tries | varint(n)
__global__ __launch_bounds__(256) void k_serde_split_size(const u32* __restrict__ list, const u32* __restrict__ list_n, const u32* __restrict__ prefix,
                                                          const u32* __restrict__ cnt, const u32* __restrict__ vcnt, const u32* __restrict__ vsize,
                                                          const u32* __restrict__ bad, u32* __restrict__ size) {
    __shared__ u32 s_scan[256 / 64 + 1];
    const u32 i = blockIdx.x, t = threadIdx.x;
    if (i >= *list_n || bad[i]) return;  // a bad one is sized by the host emitter
    const u64 v = (u64)i * 256 + t;
    u32 tot_c, tot_s;
    (void)block_exclusive_scan<256, u32>(vcnt[v] ? 1u : 0u, s_scan, &tot_c);
    (void)block_exclusive_scan<256, u32>(vsize[v], s_scan, &tot_s);
    if (t == 0) {
        const u32 r = list[i];
        size[r] = vlen(prefix[r]) + 1 + vlen(tot_c) + tot_c + vlen(tot_c) + tot_s + vlen(cnt[r]);
    }
}
// header, root node and length field of a split entry; absolute output offset of every sub-trie
__global__ __launch_bounds__(256) void k_serde_split_emit(const u32* __restrict__ list, const u32* __restrict__ list_n, const u32* __restrict__ prefix,
                                                          const u32* __restrict__ cnt, const u32* __restrict__ vcnt, const u32* __restrict__ vsize,
                                                          const u32* __restrict__ bad, const u64* __restrict__ off, u64* __restrict__ voff,
                                                          u8* __restrict__ out) {
    __shared__ u32 s_scan[256 / 64 + 1];
    const u32 i = blockIdx.x, t = threadIdx.x;
    if (i >= *list_n || bad[i]) return;
    const u32 r = list[i];
    const u64 v = (u64)i * 256 + t;
    const bool present = vcnt[v] != 0;
    u32 c0, tot_s;
    const u32 rk = block_exclusive_scan<256, u32>(present ? 1u : 0u, s_scan, &c0);
    const u32 so = block_exclusive_scan<256, u32>(vsize[v], s_scan, &tot_s);
    const u32 pfx = prefix[r], hdr = vlen(pfx) + 1;
    u8* o = out + off[r];
    const u32 root = vlen(c0) + c0 + vlen(c0);
    if (t == 0) {
        put_varint(o, pfx);
        o[hdr - 1] = 1;  // TrieOrVec::Trie
        put_varint(o + hdr, c0);
        put_varint(o + hdr + vlen(c0) + c0, c0);
        put_varint(o + hdr + root + tot_s, cnt[r]);
    }
    if (present) o[hdr + vlen(c0) + rk] = (u8)t;
    voff[v] = off[r] + hdr + root + so;
}

}  // namespace cblx
