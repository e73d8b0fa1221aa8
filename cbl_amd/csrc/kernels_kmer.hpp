// kernels_kmer.hpp — packed k-mers in and out of the index: CBL::insert / contains on an IntKmer
// (/root/reference/src/cbl.rs:199-228) and CBL::iter (:208-214,358-361). A packed k-mer is IntKmer::to_int(): 2K bits,
// first base most significant (/root/reference/src/kmer.rs:200-202); lo = low 64 bits, hi = the rest (K >= 33).
#pragma once
#include "kernels_bucket.hpp"

namespace cblx {

// get_word (src/cbl.rs:199-206): canonical() when the index is canonical (even popcount keeps the k-mer, src/kmer.rs:94-106),
// then necklace_pos + merge_necklace_pos. bad[0] counts k-mers with bits set above 2K (not an IntKmer<K>).
template <bool WIDE, typename HiT>
__global__ void k_kmers_to_words(const u64* __restrict__ k_lo, const u64* __restrict__ k_hi, u64 n, Consts P, u64* __restrict__ w_lo,
                                 HiT* __restrict__ w_hi, u32* __restrict__ bad) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef typename KmerT<WIDE>::type T;
    T x;
    bool oob;
    if constexpr (WIDE) {
        x = ((u128)(k_hi ? k_hi[i] : 0ull) << 64) | (u128)k_lo[i];  // K = 31 with a > 64-bit suffix runs this layout too
        oob = (x >> P.KB) != 0;
    } else {
        x = k_lo[i];
        oob = (x >> P.KB) != 0 || (k_hi && k_hi[i] != 0);
    }
    if (oob) { atomicAdd(bad, 1u); x = 0; }
    u64 lo, hi;
    kmer_word<WIDE>(x, P, P.canonical && !kmer_is_fwd<WIDE>(x), lo, hi);
    w_lo[i] = lo;
    st_hi<HiT>(w_hi, i, hi);
}

// ---- "was this word absent so far?" for a batch of single inserts (WordSet::insert returns it, src/wordset/mod.rs:97-120):
// absent = not in the resident index AND no equal word earlier in the batch. The second half is a first-occurrence
// test: an open-addressing table of word INDICES (slot = i + 1, 0 = empty). Equal words follow the same probe sequence
// and stop at the first slot that is empty or holds an equal word, so every distinct word owns exactly one slot, which
// atomicMin turns into the smallest index of its class. -----------------------------------------------------------------
__device__ __forceinline__ u32 first_slot(u64 lo, u64 hi, u32 mask) { return (u32)(word_hash(lo, hi) >> 29) & mask; }

template <typename HiT>
__global__ void k_first_claim(const u64* __restrict__ w_lo, const HiT* __restrict__ w_hi, u64 n, u32* __restrict__ table, u32 mask) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 lo = w_lo[i], hi = ld_hi<HiT>(w_hi, i);
    u32 h = first_slot(lo, hi, mask);
    for (;;) {
        const u32 cur = atomicCAS(&table[h], 0u, (u32)i + 1u);
        if (cur == 0) return;  // claimed an empty slot
        const u32 j = cur - 1u;  // any member of the slot's class will do for the comparison
        if (w_lo[j] == lo && ld_hi<HiT>(w_hi, j) == hi) { atomicMin(&table[h], (u32)i + 1u); return; }
        h = (h + 1u) & mask;
    }
}
// flag[i] = (i is the first occurrence of its word) && !flag_in[i]   (flag_in = membership in the resident index)
template <typename HiT>
__global__ void k_first_flag(const u64* __restrict__ w_lo, const HiT* __restrict__ w_hi, u64 n, const u32* __restrict__ table, u32 mask,
                             u8* __restrict__ flag) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 lo = w_lo[i], hi = ld_hi<HiT>(w_hi, i);
    u32 h = first_slot(lo, hi, mask);
    for (;;) {
        const u32 j = table[h] - 1u;  // never empty along the probe sequence of an inserted word
        if (w_lo[j] == lo && ld_hi<HiT>(w_hi, j) == hi) { flag[i] = (j == (u32)i && !flag[i]) ? 1 : 0; return; }
        h = (h + 1u) & mask;
    }
}

// zeros[0] += number of zero bytes (contains_all = no zero among the membership flags)
__global__ void k_count_zero_u8(const u8* __restrict__ v, u64 n, u32* __restrict__ zeros) {
    u32 z = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) z += v[i] == 0;
    z = wave_reduce_sum(z);
    if ((threadIdx.x & 63) == 0 && z) atomicAdd(zeros, z);
}

// ---- membership tallies by JOIN (the `cbl query` loop at scale, examples/cbl.rs:205-228): the query words go through the
// same stable partition as a batch of new words, so the queries of one prefix sit in one run; one workgroup then joins that
// run with the resident bucket of the prefix — the bucket is read once per workgroup instead of once per query:
//   a resident bucket of <= JOIN_TAB_MAX elements, Vec or Trie: an open-addressing table of (tag, element index) in LDS;
//   a longer Trie (ascending): binary search per query, the probes of the run's queries share the cache;
//   a longer Vec (only `|=` makes them): scanned per query.
// per_run[b] = queries of run b found. No per-query flags: the partition does not keep the queries' positions. -------------------
// hi64[i] = i << 32 | hi[i]: the ordinal of a query word rides through the partition in the unused upper half of a 64-bit
// hi part (words of at most 96 bits), so the join can write per-query flags in query order
template <typename HiT>
__global__ void k_query_tag(u64 n, const HiT* __restrict__ hi, u64* __restrict__ hi64) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) hi64[i] = (i << 32) | ld_hi<HiT>(hi, i);
}

static const u32 JOIN_THREADS = 256, JOIN_FULL_MAX = 2730, JOIN_TAB_MAX = 4095, JOIN_SLOTS = 8192;
template <bool WS, typename HiT>
__global__ __launch_bounds__(JOIN_THREADS) void k_query_join(u64 nbq, u64 b0, const u32* __restrict__ q_prefix, const u64* __restrict__ q_start,
                                                             const u64* __restrict__ q_lo, const HiT* __restrict__ q_hi, u32 SB, DirView dir,
                                                             const u64* __restrict__ a_lo, const u64* __restrict__ a_hi,
                                                             u32* __restrict__ per_run /* zero-filled, one per run */,
                                                             u8* __restrict__ flags = nullptr /* zero-filled; ordinal = q_hi >> 32 */) {
    // 32 KB of LDS, one of two tables: FULL = 4096 slots holding the suffix itself (narrow suffixes, up to JOIN_FULL_MAX
    // elements: a probe never leaves LDS — a verifying read per hit cost 40 ps, 4x the rest of the join), or 8192 slots of
    // 20-bit tag << 12 | element index (0xFFFFFFFF = empty; index 4095 is never used), verified in the bucket on a tag match
    __shared__ u64 s_full[JOIN_SLOTS / 2];
    u32* s_tab = reinterpret_cast<u32*>(s_full);
    const u64 b = b0 + blockIdx.x;
    if (b >= nbq) return;
    u64 r;
    if (!dir_lookup(dir, q_prefix[b], r)) return;  // no resident bucket: every query of the run misses
    const u64 qs = q_start[b], qe = q_start[b + 1];
    const u64 rs = dir.start[r];
    const u32 rc = dir.count[r];
    const bool trie = dir.kind[r] == KIND_TRIE;
    const u32 tid = threadIdx.x;
    auto res_at = [&](u32 j) -> Sfx<WS> { return load_sfx<WS, u64>(a_lo + rs, WS ? a_hi + rs : a_hi, j, SB); };
    auto less = [&](const Sfx<WS>& x, const Sfx<WS>& y) -> bool {
        if constexpr (WS) { if (x.hi != y.hi) return x.hi < y.hi; }
        return x.lo < y.lo;
    };
    // Up to JOIN_TAB_MAX elements (Vec or Trie alike: the elements of a bucket are distinct) go into an open-addressing
    // table in LDS, at most half full; the tag keeps the probes off global memory until a candidate matches.
    const bool full = !WS && SB < 64 && rc <= JOIN_FULL_MAX;
    const bool table = !full && rc <= JOIN_TAB_MAX;
    if (full) {
        for (u32 i = tid; i < JOIN_SLOTS / 2; i += JOIN_THREADS) s_full[i] = ~0ull;
        __syncthreads();
        for (u32 j = tid; j < rc; j += JOIN_THREADS) {
            const Sfx<WS> e = res_at(j);
            u32 h = sfx_hash_bits<WS>(e, 12);
            while (atomicCAS((unsigned long long*)&s_full[h], ~0ull, (unsigned long long)e.lo) != ~0ull) h = (h + 1u) & (JOIN_SLOTS / 2 - 1u);
        }
        __syncthreads();
    }
    if (table) {
        for (u32 i = tid; i < JOIN_SLOTS; i += JOIN_THREADS) s_tab[i] = 0xFFFFFFFFu;
        __syncthreads();
        for (u32 j = tid; j < rc; j += JOIN_THREADS) {
            const u32 hv = sfx_hash_bits<WS>(res_at(j), 32);
            u32 h = hv >> 19;
            const u32 e = ((hv & 0xFFFFFu) << 12) | j;
            while (atomicCAS(&s_tab[h], 0xFFFFFFFFu, e) != 0xFFFFFFFFu) h = (h + 1u) & (JOIN_SLOTS - 1u);
        }
        __syncthreads();
    }
    u32 found = 0;
    for (u64 q = qs + tid; q < qe; q += JOIN_THREADS) {
        const Sfx<WS> key = load_sfx<WS, HiT>(q_lo, q_hi, q, SB);
        bool hit = false;
        if (full) {
            u32 h = sfx_hash_bits<WS>(key, 12);
            for (;;) {
                const u64 e = s_full[h];
                if (e == key.lo) { hit = true; break; }
                if (e == ~0ull) break;
                h = (h + 1u) & (JOIN_SLOTS / 2 - 1u);
            }
        } else if (table) {
            const u32 hv = sfx_hash_bits<WS>(key, 32);
            u32 h = hv >> 19;
            const u32 tag = hv & 0xFFFFFu;
            for (;;) {
                const u32 e = s_tab[h];
                if (e == 0xFFFFFFFFu) break;
                if ((e >> 12) == tag && res_at(e & 0xFFFu) == key) { hit = true; break; }
                h = (h + 1u) & (JOIN_SLOTS - 1u);
            }
        } else if (trie) {  // ascending: binary search, the probes of the run's queries share the cache
            u32 l = 0, h = rc;
            while (l < h) {
                const u32 mid = (l + h) >> 1;
                if (less(res_at(mid), key)) l = mid + 1; else h = mid;
            }
            hit = l < rc && res_at(l) == key;
        } else {  // a Vec this long only comes out of `|=`
            for (u32 j = 0; j < rc && !hit; ++j) hit = res_at(j) == key;
        }
        found += hit ? 1u : 0u;
        if constexpr (std::is_same<HiT, u64>::value && !WS) {  // per-query flags: the query's ordinal rides in the upper half of hi
            if (flags && hit) flags[q_hi[q] >> 32] = 1;
        }
    }
    // one result per run (summed afterwards): 19 M wave-level atomics on ONE counter cost 47 ms at cfg 2
    found = wave_reduce_sum(found);
    __syncthreads();  // every probe of the table is done: its first word takes the tally
    if (tid == 0) s_tab[0] = 0;
    __syncthreads();
    if ((tid & 63u) == 0 && found) atomicAdd(&s_tab[0], found);
    __syncthreads();
    if (tid == 0) per_run[b] = s_tab[0];
}

// CBL::iter: element e of the index in iteration order (prefixes ascending, bucket order as stored: a Vec in
// first-occurrence order, a Trie ascending — src/wordset/mod.rs:298-309, src/trievec/mod.rs:198-206) -> word ->
// recover_kmer (src/cbl.rs:208-214, revert_necklace_pos src/necklace/mod.rs:29-31).
__global__ void k_export_kmers(u64 e0, u64 nelem, u64 nb, const u64* __restrict__ res_off, const u32* __restrict__ bucket_prefix,
                               const u64* __restrict__ start, const u64* __restrict__ a_lo, const u64* __restrict__ a_hi, Consts P,
                               u64* __restrict__ out_lo, u64* __restrict__ out_hi) {
    const u64 e = e0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nelem) return;
    u64 l = 0, h = nb;
    while (h - l > 1) {
        const u64 mid = (l + h) >> 1;
        if (res_off[mid] <= e) l = mid; else h = mid;
    }
    const u64 j = e - res_off[l];
    u128 sfx = (u128)a_lo[start[l] + j];
    if (a_hi) sfx |= (u128)a_hi[start[l] + j] << 64;
    sfx &= (((u128)1) << P.SB) - 1;
    const u128 word = ((u128)bucket_prefix[l] << P.SB) | sfx;
    const u128 necklace = word >> P.POS;
    const u32 pos = (u32)word & ((1u << P.POS) - 1u);
    const u128 MASK = (((u128)1) << P.KB) - 1;  // KB <= 118
    const u128 kmer = ((necklace << (P.KB - pos)) & MASK) | (necklace >> pos);
    out_lo[e] = (u64)kmer;
    if (out_hi) out_hi[e] = (u64)(kmer >> 64);
}

}  // namespace cblx
