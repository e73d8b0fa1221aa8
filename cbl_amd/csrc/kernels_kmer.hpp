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

// CBL::iter: element e of the index in iteration order (prefixes ascending, bucket order as stored: a Vec in
// first-occurrence order, a Trie ascending — src/wordset/mod.rs:298-309, src/trievec/mod.rs:198-206) -> word ->
// recover_kmer (src/cbl.rs:208-214, revert_necklace_pos src/necklace/mod.rs:29-31).
__global__ void k_export_kmers(u64 nelem, u64 nb, const u64* __restrict__ res_off, const u32* __restrict__ bucket_prefix,
                               const u64* __restrict__ start, const u64* __restrict__ a_lo, const u64* __restrict__ a_hi, Consts P,
                               u64* __restrict__ out_lo, u64* __restrict__ out_hi) {
    const u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x;
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
