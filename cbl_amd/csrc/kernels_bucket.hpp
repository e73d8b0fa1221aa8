// kernels_bucket.hpp — KRN-3 (per-bucket dedup-keep-first / sort) and KRN-4 (prefix bitvector, rank directory,
// bucket table) over the prefix-sorted record array.
//
// Replaces (reference, CPU):
//   KRN-4: Bitvector::insert/rank over RankBV (/root/reference/src/bitvector/mod.rs:35-47, cxx/rank_bv.h:30-33) and the
//          rank -> bucket id TieredVec32 (/root/reference/cxx/tiered_vec.h:39-45)  -> static bitvector + popcount
//          scan + bucket table indexed by rank.
//   KRN-3: TrieVec::insert / insert_iter (/root/reference/src/trievec/mod.rs:72-115): Vec bucket = linear `contains`
//          then push (first-occurrence order, unsorted); adapt_container_grow
//          (/root/reference/src/wordset/mod.rs:240-244): > 1024 distinct -> Trie (canonical, order-free; we keep the
//          sorted distinct list, of which the serialized trie is a pure function).
//
// Input to KRN-3: for bucket r the run rec[raw_start[r] .. raw_start[r+1]) holds its words in STREAM order (stable
// partition), resident entries (if any) first. Output, in place at the start of the run: the distinct suffixes,
// in first-occurrence order if their number is <= 1024 (and the resident bucket was not already a Trie), else
// ascending; count[r], kind[r].
#pragma once
#include <type_traits>

#include "kernels_radix.hpp"

namespace cblx {

static const u32 EMPTY32 = 0xFFFFFFFFu;
enum { KIND_VEC = 0, KIND_TRIE = 1 };

// ---- KRN-4 ---------------------------------------------------------------------------------------------
// start_dense[p] = index of the first record with prefix p (array pre-filled with EMPTY32).
// Each thread owns 4 consecutive records (32 B of lo per lane), the predecessor's prefix comes from the lane below.
template <typename HiT>
__global__ __launch_bounds__(256) void k_boundaries(const u64* __restrict__ lo, const HiT* __restrict__ hi, u64 n, u32 SB, u32 PB,
                                                    u32* __restrict__ start_dense) {
    const u64 i0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    u32 p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u64 i = i0 + k < n ? i0 + k : n - 1;
        p[k] = get_bits(lo[i], ld_hi<HiT>(hi, i), SB, PB);
    }
    u32 q = __shfl_up(p[3], 1, 64);
    if ((threadIdx.x & 63) == 0) q = i0 > 0 ? get_bits(lo[i0 - 1], ld_hi<HiT>(hi, i0 - 1), SB, PB) : EMPTY32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < n && p[k] != q) start_dense[p[k]] = (u32)(i0 + k);
        q = p[k];
    }
}
// Same when the hi part was dropped by the first (most significant digit) partition pass: the top `nA` prefix bits of
// record i are the segment that contains position i (seg_start[257]), the low R bits are still in lo.
__global__ __launch_bounds__(256) void k_boundaries_seg(const u64* __restrict__ lo, u64 n, u32 SB, u32 R, const u32* __restrict__ seg_start,
                                                        u32* __restrict__ start_dense) {
    __shared__ u32 s_seg[257];
    for (u32 i = threadIdx.x; i < 257; i += blockDim.x) s_seg[i] = seg_start[i];
    __syncthreads();
    const u64 i0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 >= n) return;
    u32 cur = 0xFFFFFFFFu;  // segment of the previous lookup: neighbours almost always share it
    auto prefix_at = [&](u64 i) -> u32 {
        if (cur == 0xFFFFFFFFu || (u32)i < s_seg[cur] || (u32)i >= s_seg[cur + 1]) {
            u32 l = 0, h = 256;  // last segment with seg_start[s] <= i
            while (h - l > 1) {
                const u32 mid = (l + h) >> 1;
                if (s_seg[mid] <= (u32)i) l = mid; else h = mid;
            }
            cur = l;
        }
        const u32 low = R ? get_bits(lo[i], 0, SB, R) : 0u;
        return (cur << R) | low;
    };
    u32 p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = prefix_at(i0 + k < n ? i0 + k : n - 1);
    u32 q = __shfl_up(p[3], 1, 64);
    if ((threadIdx.x & 63) == 0) q = i0 > 0 ? prefix_at(i0 - 1) : EMPTY32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < n && p[k] != q) start_dense[p[k]] = (u32)(i0 + k);
        q = p[k];
    }
}
// one lane per prefix, one wave per bitvector word
__global__ void k_bitvector(const u32* __restrict__ start_dense, u64 nprefix, u64* __restrict__ bv, u32* __restrict__ popc) {
    u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    bool has = p < nprefix && start_dense[p] != EMPTY32;
    u64 bal = __ballot(has);
    if ((threadIdx.x & 63) == 0 && p < nprefix) {
        bv[p >> 6] = bal;
        popc[p >> 6] = (u32)__builtin_popcountll(bal);
    }
}
__global__ void k_bucket_table(const u32* __restrict__ start_dense, u64 nprefix, const u64* __restrict__ bv,
                               const u64* __restrict__ rank_dir, u32* __restrict__ bucket_prefix, u64* __restrict__ raw_start,
                               u32 prefix_base = 0 /* the arrays describe prefixes [prefix_base, prefix_base + nprefix), a multiple of 64 */) {
    u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nprefix) return;
    u32 s = start_dense[p];
    if (s == EMPTY32) return;
    u64 w = bv[p >> 6];
    u64 r = rank_dir[p >> 6] + (u64)__builtin_popcountll(w & ((1ull << (p & 63)) - 1ull));
    bucket_prefix[r] = (u32)p + prefix_base;
    raw_start[r] = s;
}

// Directory of a resident index, for device-side lookups by prefix.
struct DirView {
    const u64* bv;        // 2^PB bits
    const u64* rank_dir;  // exclusive popcount prefix per bv word
    const u32* count;     // per rank
    const u8* kind;       // per rank
    const u64* start;     // per rank: first arena slot
    u64 nb;
};
__device__ __forceinline__ bool dir_lookup(const DirView& d, u32 p, u64& rank) {
    if (!d.bv) return false;
    u64 w = d.bv[p >> 6];
    if (!((w >> (p & 63)) & 1ull)) return false;
    rank = d.rank_dir[p >> 6] + (u64)__builtin_popcountll(w & ((1ull << (p & 63)) - 1ull));
    return true;
}

// ---- classification of buckets by run length -----------------------------------------------------------
enum { CLS_M16 = 0, CLS_M64 = 1, CLS_M128 = 2, CLS_M256 = 3, CLS_M512 = 4, CLS_M1024 = 5, CLS_HUGE = 6, CLS_S16 = 7, CLS_S32 = 8, CLS_BIG = 9, CLS_M32 = 10, CLS_N = 11 };
// CLS_M32 (round 6, the build only): runs of 129 - 256 words as 64 threads x FOUR slots. In the 512-slot class they paid for eight slots per lane
// — the phases are unrolled over the slots — and at PREFIX_BITS = 28 nearly all of that class is this short (cfg 3: 3.97 M of its 4.02 M runs):
// 3.51 -> 2.53 ms on the same runs (tools/dev_msd_bench.cpp -DMSD_BENCH_SPLIT). 0 = the classes of rounds 1 - 5.
#ifndef CBLX_CLASS_256
#define CBLX_CLASS_256 1
#endif
// Runs longer than one workgroup's LDS sort takes (> 4096) and up to BIG_MAX: split by the top suffix bits into sub-ranges of
// about a thousand words in scratch, each sorted + deduplicated by k_bucket_msd, then collected in order (k_big_*). Such
// buckets are the rule, not the exception, once an index holds tens of millions of reads at PREFIX_BITS = 24 (one rank of an
// 8-GPU cfg 5 job owns 70 716 buckets of 10 635 words on average).
static const u32 BIG_MAX = 1u << 18, BIG_SUB = 1024, BIG_VCAP = 2048, BIG_SENT = 0xFFFFFFFFu;
__host__ __device__ inline u32 big_bits(u32 c) {  // sub-ranges of a big run: 2^bits, about BIG_SUB words each
    u32 b = 1;
    while (b < 8 && ((c - 1) >> b) >= BIG_SUB) ++b;
    return b;
}
static const u32 SMALL_MAX = 32;  // all-pairs in a slice of a wave up to here; counting-sort kernels above
static const u32 MED_ITEMS = 8;

// One 16-byte descriptor per bucket and class: a bucket kernel starts from a single load instead of a chain of three.
struct BDesc {
    u64 start;  // first record of the run
    u32 c;      // run length; bit 31 = the resident bucket is already a Trie
    u32 r;      // bucket rank
};
static const u32 BDESC_TRIE = 0x80000000u;
// sub-ranges of a big run (k_big_split) tell the sort kernel how many leading suffix bits all their words share: bits 27..30
static const u32 BDESC_SKIP_SHIFT = 27, BDESC_SKIP_MASK = 15u << 27, BDESC_LEN_MASK = (1u << 27) - 1u;

// Slot of this thread in the list of class `cls` (cls < 0: none). The lanes of a class take consecutive slots; ONE atomic
// per (workgroup, class) reserves them (per-wave atomics on the handful of list counters were the whole cost of the
// classification kernels at 2^28 prefixes). Every thread of the workgroup must call it.
template <int THREADS, int NCLS> __device__ __forceinline__ u32 block_append(int cls, u32* __restrict__ list_n) {
    constexpr int NW = THREADS / 64;
    __shared__ u32 s_cnt[NCLS * NW];
    __shared__ u32 s_base[NCLS];
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32 my_rank = 0;
#pragma unroll
    for (int k = 0; k < NCLS; ++k) {
        const u64 bal = __ballot(cls == k);
        if (cls == k) my_rank = mbcnt(bal);
        if (lane == 0) s_cnt[k * NW + w] = (u32)__builtin_popcountll(bal);
    }
    __syncthreads();
    if (threadIdx.x < NCLS) {
        const u32 k = threadIdx.x;
        u32 run = 0;
        for (int ww = 0; ww < NW; ++ww) { const u32 t = s_cnt[k * NW + ww]; s_cnt[k * NW + ww] = run; run += t; }
        s_base[k] = run ? atomicAdd(&list_n[k], run) : 0u;
    }
    __syncthreads();
    return cls >= 0 ? s_base[cls] + s_cnt[cls * NW + w] + my_rank : 0u;
}
static const int CLASSIFY_THREADS = 1024;

__global__ __launch_bounds__(CLASSIFY_THREADS) void k_classify(u64 nb, u32 lds_max /* longest run one workgroup sorts in LDS */, const u32* __restrict__ bucket_prefix, const u64* __restrict__ raw_start,
                                                                DirView old, u32* __restrict__ res_count, u8* __restrict__ res_kind, u32* __restrict__ out_count,
                                                                u8* __restrict__ out_kind, BDesc* __restrict__ lists /* [CLS_N][nb] */, u32* __restrict__ list_n,
                                                                const u8* __restrict__ span_done = nullptr /* buckets a clean span settled already (k_bucket_span) */) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1;
    u64 c = 0;
    u8 rk = KIND_VEC;
    if (r < nb) {
        c = raw_start[r + 1] - raw_start[r];
        u32 rc = 0;
        u64 orank;
        if (dir_lookup(old, bucket_prefix[r], orank)) { rc = old.count[orank]; rk = old.kind[orank]; }
        res_count[r] = rc;
        res_kind[r] = rk;
        if (span_done && span_done[r]) {  // no repeat in the whole span it belongs to: count and kind are written, the words stay where they are
        } else if (rc != 0 && c == rc) {  // untouched by this batch: keep as is (src/wordset/mod.rs:213-214 only re-checks touched buckets)
            out_count[r] = rc;
            out_kind[r] = rk;
        } else if (c == 1 && rc == 0) {  // a single new word: nothing to deduplicate, it already sits in its slot
            out_count[r] = 1;
            out_kind[r] = KIND_VEC;
        } else if (c <= SMALL_MAX && rk != KIND_TRIE) cls = c <= 16 ? CLS_S16 : CLS_S32;  // lanes per bucket: 16 / 32
        else if (c <= 16 * MED_ITEMS) cls = CLS_M16;
        else if (CBLX_CLASS_256 && c <= 32 * MED_ITEMS) cls = CLS_M32;  // one wave, four slots per lane
        else if (c <= 64 * MED_ITEMS) cls = CLS_M64;    // workgroup size follows the run length: THREADS = CAP / 8
        else if (c <= 128 * MED_ITEMS) cls = CLS_M128;
        else if (c <= 256 * MED_ITEMS && c <= lds_max) cls = CLS_M256;
        else if (c <= 512 * MED_ITEMS && c <= lds_max) cls = CLS_M512;
        else if (c <= BIG_MAX) cls = CLS_BIG;
        else cls = CLS_HUGE;
    }
    const u32 slot = block_append<CLASSIFY_THREADS, CLS_N>(cls, list_n);
    if (cls >= 0) lists[(u64)cls * nb + slot] = BDesc{raw_start[r], (u32)c | (rk == KIND_TRIE ? BDESC_TRIE : 0u), (u32)r};
}

// ---- `self |= other` on the device: /root/reference/src/wordset/set_ops.rs:123-157 ------------------------
// The records are self's words (bucket order, stored order inside a bucket) followed by other's; after the stable
// partition the run of a prefix is [self part][other part]. Per bucket (src/trievec/set_ops.rs:43-71, :118-136):
//   other part empty -> untouched;  self part empty -> other's bucket cloned as stored (kind and order kept);
//   both, self is a Vec  -> sorted(self) ++ sorted(other \ self), stays a Vec whatever its size (no threshold check);
//   both, self is a Trie -> sorted union, Trie.
// Side effect of the reference kept: other's Vec buckets that also exist in self end up sorted (iter_sorted sorts in place).
// ---- `self |= other` without re-partitioning: both indexes are already grouped by prefix ----------------------------
// (src/wordset/set_ops.rs:123-157 walks the union of the two prefix bitvectors; here: OR of the bitvector words, rank
// by popcount scan, one merged run of cs + co slots per prefix, then the per-bucket |= rules of src/trievec/set_ops.rs)
struct MergeArgs {
    const u32* cs; const u64* ostart; const u8* okind; u64* o_lo; u64* o_hi;
    // DIRECT mode (k_bucket_msd's merge instantiation; null = the run was gathered): the bucket's two halves are read where they are
    // stored — self's arena from sstart[r], other's from ostart[r] — and only the result is written to the run
    const u64* s_lo = nullptr; const u64* s_hi = nullptr; const u64* sstart = nullptr;
};

__global__ void k_bv_or(u64 nwords, const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ out, u32* __restrict__ popc) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nwords) return;
    const u64 v = a[w] | b[w];
    out[w] = v;
    popc[w] = (u32)__builtin_popcountll(v);
}
// one thread per prefix slot of the merged bitvector: both sides' (count, kind, arena start) and the merged run length
__global__ void k_merge_table(u64 nprefix, const u64* __restrict__ bv, const u64* __restrict__ rank_dir, DirView self, DirView other,
                              u32* __restrict__ bucket_prefix, u32* __restrict__ raw_cnt, u32* __restrict__ m_cs, u64* __restrict__ m_sstart,
                              u64* __restrict__ m_ostart, u8* __restrict__ m_skind, u8* __restrict__ m_okind) {
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nprefix) return;
    const u64 w = bv[p >> 6];
    if (!((w >> (p & 63)) & 1ull)) return;
    const u64 r = rank_dir[p >> 6] + (u64)__builtin_popcountll(w & ((1ull << (p & 63)) - 1ull));
    u32 cs = 0, co = 0;
    u8 ks = KIND_VEC, ko = KIND_VEC;
    u64 rank, ss = 0, os = 0;
    if (dir_lookup(self, (u32)p, rank)) { cs = self.count[rank]; ks = self.kind[rank]; ss = self.start[rank]; }
    if (dir_lookup(other, (u32)p, rank)) { co = other.count[rank]; ko = other.kind[rank]; os = other.start[rank]; }
    bucket_prefix[r] = (u32)p;
    raw_cnt[r] = cs + co;
    m_cs[r] = cs;
    m_sstart[r] = ss;
    m_ostart[r] = os;
    m_skind[r] = ks;
    m_okind[r] = ko;
}
// LPB lanes per merged bucket: self's stored suffixes then other's, copied into the merged run
template <bool WS, int LPB>
__global__ __launch_bounds__(256) void k_merge_gather(u64 nb, const u64* __restrict__ start, const u32* __restrict__ m_cs, const u64* __restrict__ m_sstart,
                                                      const u64* __restrict__ m_ostart, const u64* __restrict__ s_lo, const u64* __restrict__ s_hi,
                                                      const u64* __restrict__ o_lo, const u64* __restrict__ o_hi, u64* __restrict__ out_lo,
                                                      u64* __restrict__ out_hi, const u8* __restrict__ skip_skind = nullptr, const u8* __restrict__ skip_okind = nullptr,
                                                      u32 skip_upto = 0 /* both-sided buckets of up to this many words are read in place by their kernel */) {
    const u64 r = ((u64)blockIdx.x * 256 + threadIdx.x) / LPB;
    if (r >= nb) return;
    const u32 lane = threadIdx.x & (LPB - 1);
    const u64 d0 = start[r];
    const u32 c = (u32)(start[r + 1] - d0), cs = m_cs[r];
    // Trie |= Trie: k_bucket_union merges the two stored lists straight into the run — nothing to copy here
    if (skip_skind && cs != 0 && cs != c && skip_skind[r] == KIND_TRIE && skip_okind[r] == KIND_TRIE) return;
    if (cs != 0 && cs != c && c <= skip_upto) return;
    const u64 ss = m_sstart[r], os = m_ostart[r];
    for (u32 j = lane; j < c; j += LPB) {
        const bool from_self = j < cs;
        const u64 src = from_self ? ss + j : os + (j - cs);
        out_lo[d0 + j] = from_self ? s_lo[src] : o_lo[src];
        if constexpr (WS) out_hi[d0 + j] = from_self ? s_hi[src] : o_hi[src];
    }
}
// the same for the buckets of a list (a both-sided bucket that was read in place gave up in its kernel and goes to the radix kernel, which works on the run)
template <bool WS>
__global__ __launch_bounds__(256) void k_merge_gather_list(const BDesc* __restrict__ list, const u32* __restrict__ list_n, const u32* __restrict__ m_cs, const u64* __restrict__ m_sstart,
                                                           const u64* __restrict__ m_ostart, const u64* __restrict__ s_lo, const u64* __restrict__ s_hi, const u64* __restrict__ o_lo,
                                                           const u64* __restrict__ o_hi, u64* __restrict__ out_lo, u64* __restrict__ out_hi) {
    if (blockIdx.x >= *list_n) return;
    const BDesc d = list[blockIdx.x];
    const u32 c = d.c & BDESC_LEN_MASK, cs = m_cs[d.r];
    const u64 ss = m_sstart[d.r], os = m_ostart[d.r];
    for (u32 j = threadIdx.x; j < c; j += 256) {
        const bool from_self = j < cs;
        const u64 src = from_self ? ss + j : os + (j - cs);
        out_lo[d.start + j] = from_self ? s_lo[src] : o_lo[src];
        if constexpr (WS) out_hi[d.start + j] = from_self ? s_hi[src] : o_hi[src];
    }
}
// one-sided buckets are final after the gather (self-only: untouched; other-only: cloned as stored); both-sided ones
// go to the merge epilogue of the bucket kernel of their length class
static const int CLS_UNION = CLS_S16;  // `self |= other` only (its classification has no small classes): Trie |= Trie by merge path (k_bucket_union)
__global__ __launch_bounds__(CLASSIFY_THREADS) void k_classify_merge(u64 nb, u32 med_max_threads, const u64* __restrict__ raw_start, const u32* __restrict__ m_cs,
                                 const u8* __restrict__ m_skind, const u8* __restrict__ m_okind, u32* __restrict__ out_count,
                                 u8* __restrict__ out_kind, BDesc* __restrict__ lists, u32* __restrict__ list_n, bool union_path = false,
                                 unsigned long long* __restrict__ cls_words = nullptr /* profiling: [CLS_N + 1] words per class, [CLS_N] = one-sided */) {
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1;
    u64 c = 0;
    u8 ks = KIND_VEC;
    if (r < nb) {
        c = raw_start[r + 1] - raw_start[r];
        const u32 cs = m_cs[r], co = (u32)c - cs;
        ks = m_skind[r];
        if (co == 0) { out_count[r] = cs; out_kind[r] = ks; }
        else if (cs == 0) { out_count[r] = co; out_kind[r] = m_okind[r]; }
        else if (union_path && ks == KIND_TRIE && m_okind[r] == KIND_TRIE) cls = CLS_UNION;  // two ascending lists, any length: merged, not sorted again
        else if (c <= 16 * MED_ITEMS) cls = CLS_M16;   // workgroup size follows the run length, as in k_classify
        else if (c <= 64 * MED_ITEMS) cls = CLS_M64;
        else if (c <= 128 * MED_ITEMS) cls = CLS_M128;
        else if (c <= 256 * MED_ITEMS) cls = CLS_M256;
        else if (c <= 512 * MED_ITEMS) cls = CLS_M512;
        else if (ks == KIND_TRIE && m_okind[r] == KIND_TRIE && c <= BIG_MAX) cls = CLS_BIG;  // Trie |= Trie: the sorted union, nothing else
        else if (c <= 1024 * MED_ITEMS && med_max_threads >= 1024) cls = CLS_M1024;
        else cls = CLS_HUGE;
    }
    const u32 slot = block_append<CLASSIFY_THREADS, CLS_N>(cls, list_n);
    if (cls >= 0) lists[(u64)cls * nb + slot] = BDesc{raw_start[r], (u32)c | (ks == KIND_TRIE ? BDESC_TRIE : 0u), (u32)r};
    if (cls_words) {  // (profiling only: a wave reduction per class, summed per workgroup in LDS, ONE global atomic per class and workgroup —
                      //  one per class and WAVE, as first written, serialised half a million atomics on eleven addresses: 22 -> 737 us at cfg 5's share)
        __shared__ unsigned long long s_cw[CLS_N + 1];
        if (threadIdx.x <= CLS_N) s_cw[threadIdx.x] = 0ull;
        __syncthreads();
#pragma unroll
        for (int k = -1; k < CLS_N; ++k) {
            const u64 s = wave_reduce_sum(cls == k && r < nb ? c : 0ull);
            if ((threadIdx.x & 63) == 0 && s) atomicAdd(&s_cw[k < 0 ? CLS_N : k], (unsigned long long)s);
        }
        __syncthreads();
        if (threadIdx.x <= CLS_N && s_cw[threadIdx.x]) atomicAdd(&cls_words[threadIdx.x], s_cw[threadIdx.x]);
    }
}

// ---- Trie |= Trie: the union of two ASCENDING lists is a merge, not a sort (/root/reference/src/trievec/set_ops.rs:43-71 merges two
// sorted iterators with two pointers; src/trievec/mod.rs:118-136 inserts what self lacks) ------------------------------------------------
// One workgroup per bucket, any length. Rounds of UNI_TILE outputs: the next <= UNI_TILE words of either list are staged in LDS
// (coalesced loads from the two indexes' own arenas — the merged run is written, never read), every thread finds the co-rank of the
// END of its UNI_ITEMS consecutive outputs by a binary search on the round's diagonal (merge path; ties take self's copy first) and gets
// its start from its neighbour, loads n + n candidates into registers and merges them with a fixed network — min(a[k], b[n-1-k]) leaves
// the n smallest as a bitonic sequence, log2 n compare-exchange stages sort it — so no lane follows data-dependent control flow. The
// outputs go back to LDS (one padding word per 8: a lane's 64-byte row would otherwise hit the banks of its neighbours'), are
// compared with their predecessor there (equal = other's copy of a word self holds: dropped) and leave compacted in order. The
// co-rank of the round's last output says how far either list was consumed. HBM traffic: every word read once (the staging areas are
// rings since round 5: a round refills only the slots it consumed — see the kernel), the union written once.
// Workgroup shape, measured on cfg 5's share (`bench.py --config merge`, stage bucket_big) / at the 8-GPU depth of cfg 5 (`tools/emulate_rank.py --merge`:
// 752 M + 752 M words in buckets of 10 635 each): 256 threads x 8 outputs 3.18 / 9.4 ms, 128 x 8 3.09, 256 x 4 2.14 / 6.3, **128 x 4 2.07 / 6.2**, 256 x 2 2.53 — half the registers
// (a[], b[], o[]), half the merge network and a shorter search per round buy more than the extra rounds cost; padding one word per 4 instead of 8: 2.33 / 6.7.
static const int UNI_THREADS = 128, UNI_ITEMS = 4, UNI_TILE = UNI_THREADS * UNI_ITEMS;
__device__ __forceinline__ u32 uni_pad(u32 i) { return i + (i >> 3); }
// An element of a union: the suffix alone — one u64, or (hi, lo) for suffixes wider than 64 bits (round 5: those took the sorting
// classes whatever their kinds; the rings hold their two halves in two arrays of the same shape).
template <bool WS> struct UniE;
template <> struct UniE<false> { u64 lo; };
template <> struct UniE<true> { u64 lo, hi; };
template <bool WS> __device__ __forceinline__ bool uni_lt(const UniE<WS>& a, const UniE<WS>& b) {
    if constexpr (WS) return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo);
    else return a.lo < b.lo;
}
template <bool WS> __device__ __forceinline__ bool uni_eq(const UniE<WS>& a, const UniE<WS>& b) {
    if constexpr (WS) return a.lo == b.lo && a.hi == b.hi;
    else return a.lo == b.lo;
}
template <bool WS> __device__ __forceinline__ UniE<WS> uni_inf() {
    UniE<WS> e;
    e.lo = ~0ull;
    if constexpr (WS) e.hi = ~0ull;
    return e;
}
// (selects word by word: a select between two structs sends the arrays they sit in to scratch memory — 224 bytes per lane and 1.4 TB/s
//  as first written)
template <bool WS> __device__ __forceinline__ UniE<WS> uni_sel(bool take_a, const UniE<WS>& a, const UniE<WS>& b) {
    UniE<WS> e;
    e.lo = take_a ? a.lo : b.lo;
    if constexpr (WS) e.hi = take_a ? a.hi : b.hi;
    return e;
}
template <bool WS> __device__ __forceinline__ void uni_cmpx(UniE<WS>& a, UniE<WS>& b) {
    const bool sw = uni_lt<WS>(b, a);
    const UniE<WS> lo = uni_sel<WS>(sw, b, a), hi = uni_sel<WS>(sw, a, b);
    a = lo;
    b = hi;
}
template <bool WS>
__global__ __launch_bounds__(UNI_THREADS) void k_bucket_union(const BDesc* __restrict__ list, const u32* __restrict__ list_n, const u32* __restrict__ m_cs,
                                                              const u64* __restrict__ m_sstart, const u64* __restrict__ m_ostart, const u64* __restrict__ s_lo,
                                                              const u64* __restrict__ s_hi, const u64* __restrict__ o_lo, const u64* __restrict__ o_hi,
                                                              u64* __restrict__ out_lo, u64* __restrict__ out_hi, u32 SB, u32* __restrict__ out_count,
                                                              u8* __restrict__ out_kind) {
    typedef UniE<WS> E;
    constexpr int NW = UNI_THREADS / 64;
    constexpr u32 T = UNI_TILE;
    static_assert((T & (T - 1)) == 0, "the staging rings index by g mod T");
    constexpr u32 SLOTS = T * 2 + (T * 2) / 8 + 8;
    __shared__ u64 s_in[SLOTS];  // A's ring at logical [0, T), B's at [T, 2T); the round's outputs in the slots it consumed
    __shared__ u64 s_inh[WS ? SLOTS : 1];  // (wide suffixes: the high halves, same slots)
    __shared__ u32 s_split[NW + 1];
    __shared__ u32 s_wtot[NW + 1];
    if (blockIdx.x >= *list_n) return;
    const BDesc dsc = list[blockIdx.x];
    const u32 r = dsc.r, c = dsc.c & BDESC_LEN_MASK, cs = m_cs[r], co = c - cs;
    const u64 a_self = m_sstart[r], a_oth = m_ostart[r];
    const u64* __restrict__ A = s_lo + a_self;
    const u64* __restrict__ B = o_lo + a_oth;
    const u64* __restrict__ Ah = WS ? s_hi + a_self : nullptr;
    const u64* __restrict__ Bh = WS ? o_hi + a_oth : nullptr;
    u64* __restrict__ dst = out_lo + dsc.start;
    u64* __restrict__ dsth = WS ? out_hi + dsc.start : nullptr;
    // narrow: suffix = lo & mask; wide: (hi & mask(SB - 64), lo)
    const u64 mask = WS ? ((1ull << (SB - 64)) - 1ull) : (SB >= 64 ? ~0ull : ((1ull << SB) - 1ull));
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    auto get = [&](u32 slot) { E e; e.lo = s_in[slot]; if constexpr (WS) e.hi = s_inh[slot]; return e; };
    auto put = [&](u32 slot, const E& e) { s_in[slot] = e.lo; if constexpr (WS) s_inh[slot] = e.hi; };
    // The two staging areas are RINGS of T slots (round 5): word g of a list lives in slot g mod T of its ring, a round consumes nout
    // words — iend from A, the rest from B — and only those nout slots are refilled in front of the next round; the round's outputs pass
    // through exactly the slots it freed. As first written every round staged the next T words of BOTH lists again and consumed T in
    // all: every word crossed the L2 twice, and with 16 workgroups per CU holding 8 KB each the second read missed — 6.3 GB fetched for
    // 4.0 GB of lists by the TCC counters (profiles/r05_merge_hbm_traffic.md).
    u32 ia = 0, ib = 0, written = 0;
    u32 ha = 0, hb = 0;  // words of A from ia / of B from ib already in the rings
    E carry = uni_inf<WS>();
    bool have_carry = false;
    auto ra = [&](u32 g) { return uni_pad(g & (T - 1)); };
    auto rb = [&](u32 g) { return uni_pad(T + (g & (T - 1))); };
    while (ia < cs || ib < co) {
        const u32 na = cs - ia < T ? cs - ia : T, nb = co - ib < T ? co - ib : T, nout = na + nb < T ? na + nb : T;
        for (u32 g = ia + ha + tid; g < ia + na; g += UNI_THREADS) {
            if constexpr (WS) { s_in[ra(g)] = A[g]; s_inh[ra(g)] = Ah[g] & mask; }
            else s_in[ra(g)] = A[g] & mask;
        }
        for (u32 g = ib + hb + tid; g < ib + nb; g += UNI_THREADS) {
            if constexpr (WS) { s_in[rb(g)] = B[g]; s_inh[rb(g)] = Bh[g] & mask; }
            else s_in[rb(g)] = B[g] & mask;
        }
        __syncthreads();
        // co-rank of the end of this thread's outputs: how many of the first d1 outputs come from A
        const u32 d0 = tid * UNI_ITEMS < nout ? tid * UNI_ITEMS : nout, d1 = (tid + 1) * UNI_ITEMS < nout ? (tid + 1) * UNI_ITEMS : nout;
        u32 lo = d1 > nb ? d1 - nb : 0u, hi = d1 < na ? d1 : na;
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (!uni_lt<WS>(get(rb(ib + d1 - 1 - mid)), get(ra(ia + mid)))) lo = mid + 1; else hi = mid;  // A[mid] <= B[d1 - 1 - mid]: ties take self's copy first
        }
        const u32 i1 = lo;
        if (lane == 63) s_split[w + 1] = i1;
        if (tid == 0) s_split[0] = 0;
        __syncthreads();
        u32 i0 = __shfl_up(i1, 1, 64);
        if (lane == 0) i0 = s_split[w];
        const u32 iend = s_split[NW];  // co-rank of nout (the last thread's end)
        const u32 j0 = d0 - i0;
        E a[UNI_ITEMS], b[UNI_ITEMS];
#pragma unroll
        for (int k = 0; k < UNI_ITEMS; ++k) {
            const u32 x = i0 + k, y = j0 + k;
            a[k] = uni_sel<WS>(x < na, get(ra(ia + (x < na ? x : 0u))), uni_inf<WS>());
            b[k] = uni_sel<WS>(y < nb, get(rb(ib + (y < nb ? y : 0u))), uni_inf<WS>());
        }
        E o[UNI_ITEMS];
#pragma unroll
        for (int k = 0; k < UNI_ITEMS; ++k) o[k] = uni_sel<WS>(uni_lt<WS>(a[k], b[UNI_ITEMS - 1 - k]), a[k], b[UNI_ITEMS - 1 - k]);
#pragma unroll
        for (int st = UNI_ITEMS / 2; st >= 1; st >>= 1)
#pragma unroll
            for (int k = 0; k < UNI_ITEMS; ++k)
                if ((k & st) == 0) uni_cmpx<WS>(o[k], o[k + st]);
        __syncthreads();  // every read of the chunks is done: the outputs take the place of what the round consumed
        // output q of the round sits in the q-th freed slot: A's ring first (iend of them), then B's
        auto oslot = [&](u32 q) { return q < iend ? ra(ia + q) : rb(ib + (q - iend)); };
#pragma unroll
        for (int k = 0; k < UNI_ITEMS; ++k) {
            const u32 q = tid * UNI_ITEMS + k;
            if (q < nout) put(oslot(q), o[k]);  // (nothing past nout: those slots hold words of the next round)
        }
        __syncthreads();
        // ordered compaction of the outputs that differ from their predecessor (wave-contiguous slices keep the order)
        E v[UNI_ITEMS];
        bool head[UNI_ITEMS];
        u32 wh = 0;
#pragma unroll
        for (int j = 0; j < UNI_ITEMS; ++j) {
            const u32 p = w * (64 * UNI_ITEMS) + j * 64 + lane;
            const bool live = p < nout;
            v[j] = get(oslot(live ? p : 0u));
            const E u = get(oslot((live && p) ? p - 1 : 0u));
            head[j] = live && (p ? !uni_eq<WS>(v[j], u) : (!have_carry || !uni_eq<WS>(v[j], carry)));
            wh += (u32)__builtin_popcountll(__ballot(head[j]));
        }
        const E last = get(oslot(nout - 1));
        if (lane == 0) s_wtot[w] = wh;
        __syncthreads();
        u32 run = 0, tot = 0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) { const u32 t = s_wtot[ww]; if ((u32)ww < w) run += t; tot += t; }
#pragma unroll
        for (int j = 0; j < UNI_ITEMS; ++j) {
            const u64 bal = __ballot(head[j]);
            if (head[j]) {
                dst[written + run + mbcnt(bal)] = v[j].lo;
                if constexpr (WS) dsth[written + run + mbcnt(bal)] = v[j].hi;
            }
            run += (u32)__builtin_popcountll(bal);
        }
        carry = last;
        have_carry = true;
        written += tot;
        ha = na - iend;
        hb = nb - (nout - iend);
        ia += iend;
        ib += nout - iend;
        __syncthreads();  // the next round refills the freed slots and rewrites the split table
    }
    if (tid == 0) { out_count[r] = written; out_kind[r] = KIND_TRIE; }
}

// ---- suffix access --------------------------------------------------------------------------------------
// narrow suffix (SB <= 64): suffix = lo & mask. wide (SB > 64): (hi & mask(SB-64), lo).
template <bool WS> struct Sfx;
template <> struct Sfx<false> {
    u64 lo;
    __device__ __forceinline__ bool operator==(const Sfx& o) const { return lo == o.lo; }
    __device__ __forceinline__ bool operator!=(const Sfx& o) const { return lo != o.lo; }
    __device__ __forceinline__ u32 digit(u32 pass) const { return (u32)(lo >> (8 * pass)) & 255u; }
};
template <> struct Sfx<true> {
    u64 lo, hi;
    __device__ __forceinline__ bool operator==(const Sfx& o) const { return lo == o.lo && hi == o.hi; }
    __device__ __forceinline__ bool operator!=(const Sfx& o) const { return lo != o.lo || hi != o.hi; }
    __device__ __forceinline__ u32 digit(u32 pass) const { return pass < 8 ? (u32)(lo >> (8 * pass)) & 255u : (u32)(hi >> (8 * (pass - 8))) & 255u; }
};
template <bool WS, typename HiT>
__device__ __forceinline__ Sfx<WS> load_sfx(const u64* lo, const HiT* hi, u64 i, u32 SB) {
    Sfx<WS> s;
    if constexpr (WS) {
        s.lo = lo[i];
        s.hi = ld_hi<HiT>(hi, i) & ((1ull << (SB - 64)) - 1ull);
    } else {
        s.lo = SB >= 64 ? lo[i] : (lo[i] & ((1ull << SB) - 1ull));
    }
    return s;
}
template <bool WS, typename HiT> __device__ __forceinline__ void store_sfx(u64* lo, HiT* hi, u64 i, const Sfx<WS>& s) {
    lo[i] = s.lo;
    if constexpr (WS) st_hi<HiT>(hi, i, s.hi);
}
template <bool WS> __device__ __forceinline__ Sfx<WS> shfl_sfx(const Sfx<WS>& s, int src) {
    Sfx<WS> r;
    r.lo = __shfl(s.lo, src, 64);
    if constexpr (WS) r.hi = __shfl(s.hi, src, 64);
    return r;
}

// ---- sorted batches (multi-GPU build): a batch = ascending non-empty prefixes, words per prefix, and the suffixes alone,
// packed to BYTES little-endian bytes each, bucket-major and in stream order inside a bucket. A rank partitions its own
// words completely, ships per destination a slice of that batch (6 B per word at K=31 / PREFIX_BITS=24 instead of 9), and
// the receiver merges the batches of all ranks bucket by bucket: nothing is partitioned twice. ---------------------------
__device__ __forceinline__ u64 rank_of(const u64* __restrict__ bv, const u64* __restrict__ rank_dir, u32 p) {
    const u64 w = bv[p >> 6];
    return rank_dir[p >> 6] + (u64)__builtin_popcountll(w & ((1ull << (p & 63)) - 1ull));
}
// sorted records -> packed suffixes. A workgroup packs a tile of PACK_TILE words into LDS (BYTES bytes each, 16-bit pieces
// when BYTES is even) and writes the tile out as aligned dwords: one thread per word storing its bytes straight to global
// memory ran at byte-store speed (BYTES partial stores per word).
static const u32 PACK_THREADS = 256, PACK_ITEMS = 8, PACK_TILE = PACK_THREADS * PACK_ITEMS;
template <bool WS, typename HiT>
__global__ __launch_bounds__(PACK_THREADS) void k_batch_pack(u64 n, const u64* __restrict__ lo, const HiT* __restrict__ hi, u32 SB, u32 BYTES, u8* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) u8 s_b[];  // PACK_TILE * BYTES bytes (launch parameter): 12 KB at K=31 / PB=24 keeps 32 waves per CU
    const u64 t0 = (u64)blockIdx.x * PACK_TILE;
    const u32 nt = (u32)((n - t0) < (u64)PACK_TILE ? (n - t0) : (u64)PACK_TILE);
#pragma unroll
    for (u32 j = 0; j < PACK_ITEMS; ++j) {
        const u32 i = j * PACK_THREADS + threadIdx.x;
        if (i < nt) {
            const Sfx<WS> s = load_sfx<WS, HiT>(lo, hi, t0 + i, SB);
            if ((BYTES & 1u) == 0) {
                u16* o = reinterpret_cast<u16*>(s_b + i * BYTES);
                for (u32 k = 0; k < BYTES / 2; ++k) {
                    u32 v;
                    if constexpr (WS) v = k < 4 ? (u32)(s.lo >> (16 * k)) : (u32)(s.hi >> (16 * (k - 4)));
                    else v = (u32)(s.lo >> (16 * k));
                    o[k] = (u16)v;
                }
            } else {
                u8* o = s_b + i * BYTES;
                for (u32 k = 0; k < BYTES; ++k) {
                    u32 v;
                    if constexpr (WS) v = k < 8 ? (u32)(s.lo >> (8 * k)) : (u32)(s.hi >> (8 * (k - 8)));
                    else v = (u32)(s.lo >> (8 * k));
                    o[k] = (u8)v;
                }
            }
        }
    }
    __syncthreads();
    const u32 nbytes = nt * BYTES;            // the tile starts at byte t0 * BYTES: a multiple of 16 (PACK_TILE is)
    u8* g = out + t0 * BYTES;
    const uint4* sw = reinterpret_cast<const uint4*>(s_b);
    uint4* gw = reinterpret_cast<uint4*>(g);
    for (u32 w = threadIdx.x; w < nbytes / 16; w += PACK_THREADS) gw[w] = sw[w];
    for (u32 b = (nbytes & ~15u) + threadIdx.x; b < nbytes; b += PACK_THREADS) g[b] = s_b[b];
}
__global__ void k_batch_counts(u64 nb, const u64* __restrict__ start, u32* __restrict__ cnt) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < nb) cnt[r] = (u32)(start[r + 1] - start[r]);
}
// first bucket with prefix >= bounds[d] (d < nbounds), and the words in front of it
__global__ void k_batch_split(u64 nb, const u32* __restrict__ prefix, const u64* __restrict__ start, u32 nbounds, const u32* __restrict__ bounds,
                              u64* __restrict__ bucket_split, u64* __restrict__ word_split) {
    const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= nbounds) return;
    u64 l = 0, h = nb;
    while (l < h) {
        const u64 mid = (l + h) >> 1;
        if (prefix[mid] < bounds[d]) l = mid + 1; else h = mid;
    }
    bucket_split[d] = l;
    word_split[d] = start[l];  // start has nb + 1 entries
}
__global__ void k_batch_bits(u64 nbk, const u32* __restrict__ prefix, const u32* __restrict__ cnt, u64 nprefix, u64* __restrict__ bv, u32* __restrict__ bad) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbk) return;
    const u32 p = prefix[i];
    // out of range / not strictly ascending / an empty bucket (would leave a set bit without words)
    if (p >= nprefix || (i > 0 && prefix[i - 1] >= p) || cnt[i] == 0) { atomicAdd(bad, 1u); return; }
    // The prefixes of a batch ascend, so the buckets of one bitvector word are neighbours: the first of them ORs the bits of
    // all (a plain read-modify-write: the batches of a call are applied by successive launches, and within a launch a word
    // has one writer — unless the batch is malformed, which `bad` reports and the caller rejects). One atomic per bucket
    // cost 1 ms per batch at PREFIX_BITS = 28.
    if (i > 0 && (prefix[i - 1] >> 6) == (p >> 6)) return;
    u64 bits = 1ull << (p & 63);
    for (u64 j = i + 1; j < nbk && j < i + 64; ++j) {
        const u32 q = prefix[j];
        if ((q >> 6) != (p >> 6)) break;
        bits |= 1ull << (q & 63);
    }
    bv[p >> 6] |= bits;
}
__global__ void k_popc_words(u64 nwords, const u64* __restrict__ bv, u32* __restrict__ popc) {
    const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nwords) popc[w] = (u32)__builtin_popcountll(bv[w]);
}
// Batches are applied one after the other (stream order): where in the merged run of its prefix does this batch's part go.
// Prefixes are unique inside a batch, so run_len has no concurrent writers.
__global__ void k_batch_offsets(u64 nbk, const u32* __restrict__ prefix, const u32* __restrict__ cnt, const u64* __restrict__ bv,
                                const u64* __restrict__ rank_dir, u32* __restrict__ run_len, u32* __restrict__ off_in_run) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbk) return;
    const u64 r = rank_of(bv, rank_dir, prefix[i]);
    const u32 o = run_len[r];
    off_in_run[i] = o;
    run_len[r] = o + cnt[i];
}
// where in the arena each bucket of a batch goes (start of its merged run + its offset inside the run): computed once per
// bucket here so that the gather below starts from three independent loads instead of a chain of five
__global__ void k_batch_dst(u64 nbk, const u32* __restrict__ prefix, const u32* __restrict__ off_in_run, const u64* __restrict__ bv,
                            const u64* __restrict__ rank_dir, const u64* __restrict__ start, u64* __restrict__ dst) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nbk) dst[i] = start[rank_of(bv, rank_dir, prefix[i])] + off_in_run[i];
}
// resident suffixes (stored order) to the front of their merged runs; LPB lanes per merged bucket
template <bool WS, int LPB>
__global__ __launch_bounds__(256) void k_gather_resident(u64 nb, const u64* __restrict__ start, const u32* __restrict__ m_cs, const u64* __restrict__ m_sstart,
                                                         const u64* __restrict__ s_lo, const u64* __restrict__ s_hi, u64* __restrict__ out_lo,
                                                         u64* __restrict__ out_hi) {
    const u64 r = ((u64)blockIdx.x * 256 + threadIdx.x) / LPB;
    if (r >= nb) return;
    const u32 lane = threadIdx.x & (LPB - 1), cs = m_cs[r];
    const u64 d0 = start[r], ss = m_sstart[r];
    for (u32 j = lane; j < cs; j += LPB) {
        out_lo[d0 + j] = s_lo[ss + j];
        if constexpr (WS) out_hi[d0 + j] = s_hi[ss + j];
    }
}
// one batch's packed suffixes into the merged runs; LPB lanes per bucket of the batch (a wave per bucket leaves most lanes
// idle once the buckets are short: PREFIX_BITS = 28 spreads a slice over buckets of a few dozen words)
template <bool WS, int LPB>
__global__ __launch_bounds__(256) void k_gather_packed(u64 nbk, const u32* __restrict__ cnt, const u64* __restrict__ src_off, const u64* __restrict__ dst,
                                                       const u8* __restrict__ packed, u64 packed_bytes, u32 BYTES, u64* __restrict__ out_lo,
                                                       u64* __restrict__ out_hi) {
    const u64 i = ((u64)blockIdx.x * 256 + threadIdx.x) / LPB;
    if (i >= nbk) return;
    const u32 lane = threadIdx.x & (LPB - 1), c = cnt[i];
    const u64 d0 = dst[i];
    const u64 e0 = src_off[i];
    const u64 lo_mask = BYTES >= 8 ? ~0ull : ((1ull << (8 * BYTES)) - 1ull);
    const u64 hi_mask = BYTES >= 16 ? ~0ull : (BYTES > 8 ? ((1ull << (8 * (BYTES - 8))) - 1ull) : 0ull);
    for (u32 j = lane; j < c; j += LPB) {
        const u64 byte0 = (e0 + j) * BYTES;
        const u8* q = packed + byte0;
        u64 lo = 0, hi = 0;
        if (byte0 + 16 <= packed_bytes) {  // two (unaligned) 8-byte loads; only the last element or two of a batch go byte by byte
            lo = *reinterpret_cast<const u64*>(q) & lo_mask;
            if constexpr (WS) hi = *reinterpret_cast<const u64*>(q + 8) & hi_mask;
        } else {
            for (u32 k = 0; k < BYTES; ++k) {
                const u64 b = q[k];
                if (k < 8) lo |= b << (8 * k); else hi |= b << (8 * (k - 8));
            }
        }
        out_lo[d0 + j] = lo;
        if constexpr (WS) out_hi[d0 + j] = hi;
    }
}

// ---- KRN-3 small: WIDTH lanes per bucket (run <= WIDTH <= 64; 64 / WIDTH buckets share a wave), all-pairs "seen
// before?" (the reference's own `vec.contains(x)` semantics, src/trievec/mod.rs:81-87), ordered compaction by ballot ---
template <int WIDTH, bool WS, typename HiT>
__global__ __launch_bounds__(256) void k_bucket_small(const BDesc* __restrict__ list, const u32* __restrict__ list_n,
                                                      u64* __restrict__ lo, HiT* __restrict__ hi, u32 SB,
                                                      u32* __restrict__ out_count, u8* __restrict__ out_kind) {
    const u32 g = (blockIdx.x * blockDim.x + threadIdx.x) / WIDTH, lane = threadIdx.x & 63, gl = lane & (WIDTH - 1);
    const bool live = g < *list_n;
    BDesc dsc{0, 0, 0};
    if (live) dsc = list[g];
    const u32 r = dsc.r;
    const u64 s0 = dsc.start;
    const u32 c = dsc.c & ~BDESC_TRIE;  // 0 for the idle groups of the last wave
    Sfx<WS> mine;
    mine.lo = 0;
    if constexpr (WS) mine.hi = 0;
    if (gl < c) mine = load_sfx<WS, HiT>(lo, hi, s0 + gl, SB);
    // the groups of a wave loop together: up to the longest run among them
    u32 cmax = c;
#pragma unroll
    for (int o = WIDTH; o < 64; o <<= 1) { const u32 t = (u32)__shfl_xor((int)cmax, o, 64); cmax = t > cmax ? t : cmax; }
    bool dup = false;
    const u32 gbase = lane & ~(u32)(WIDTH - 1);
    for (u32 j = 0; j + 1 < cmax; ++j) {
        Sfx<WS> o = shfl_sfx<WS>(mine, (int)(gbase + j));
        if (j < gl && j < c && o == mine) dup = true;
    }
    const bool keep = gl < c && !dup;
    const u64 bal = (__ballot(keep) >> gbase) & (WIDTH == 64 ? ~0ull : ((1ull << WIDTH) - 1ull));
    if (keep) store_sfx<WS, HiT>(lo, hi, s0 + (u32)__builtin_popcountll(bal & ((1ull << gl) - 1ull)), mine);
    if (gl == 0 && live) {
        out_count[r] = (u32)__builtin_popcountll(bal);
        out_kind[r] = KIND_VEC;
    }
}

// ---- KRN-3 medium: one workgroup per bucket, run <= THREADS*8, stable LSD radix sort in LDS -----------------
template <int THREADS, bool WS, typename HiT>
__global__ __launch_bounds__(THREADS) void k_bucket_medium(const BDesc* __restrict__ list, const u32* __restrict__ list_n,
                                                           u64* __restrict__ lo, HiT* __restrict__ hi, u32 SB,
                                                           u32* __restrict__ out_count, u8* __restrict__ out_kind, MergeArgs mg) {
    constexpr int ITEMS = MED_ITEMS, NW = THREADS / 64, CAP = THREADS * ITEMS;
    __shared__ u64 s_klo[CAP];
    __shared__ u64 s_khi[WS ? CAP : 1];
    __shared__ u16 s_idx[CAP];
    __shared__ u32 s_wcnt[NW * 256];
    __shared__ u32 s_dbase[256];
    __shared__ u32 s_scan[NW + 1];
    __shared__ u32 s_bm[CAP / 32];
    __shared__ u32 s_wtot[NW + 1];

    if (blockIdx.x >= *list_n) return;
    const BDesc dsc = list[blockIdx.x];
    const u32 r = dsc.r;
    const u64 s0 = dsc.start;
    const u32 c = dsc.c & ~BDESC_TRIE;
    const bool res_trie = (dsc.c & BDESC_TRIE) != 0;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const u32 R = (c + THREADS - 1) / THREADS;  // rounds actually needed (<= ITEMS)
    const u32 EPW = 64 * R;                     // elements per wave slice

    Sfx<WS> key[ITEMS];
    u32 idx[ITEMS], digit[ITEMS], pos[ITEMS];
    bool valid[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 e = w * EPW + j * 64 + lane;
        valid[j] = (u32)j < R && e < c;
        idx[j] = e;
        key[j].lo = 0;
        if constexpr (WS) key[j].hi = 0;
        if (valid[j]) key[j] = load_sfx<WS, HiT>(lo, hi, s0 + e, SB);
    }
    const u32 npass = (SB + 7) / 8;
    for (u32 pass = 0; pass < npass; ++pass) {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) digit[j] = valid[j] ? key[j].digit(pass) : 255u;
        tile_rank<THREADS, ITEMS>(digit, pos, s_wcnt, s_dbase, s_scan, R);
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if ((u32)j < R) {  // tail slots (digit 255 in every pass) stay in [c, THREADS*R)
                s_klo[pos[j]] = key[j].lo;
                if constexpr (WS) s_khi[pos[j]] = key[j].hi;
                s_idx[pos[j]] = (u16)idx[j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 e = w * EPW + j * 64 + lane;
            if (valid[j]) {
                key[j].lo = s_klo[e];
                if constexpr (WS) key[j].hi = s_khi[e];
                idx[j] = s_idx[e];
            }
        }
        __syncthreads();
    }
    // heads of equal-suffix runs; equal suffixes are in stream order (stable), so a head is a first occurrence
    bool head[ITEMS];
    u32 wave_heads = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 e = w * EPW + j * 64 + lane;
        head[j] = false;
        if (valid[j]) {
            if (e == 0) head[j] = true;
            else {
                Sfx<WS> prev;
                prev.lo = s_klo[e - 1];
                if constexpr (WS) prev.hi = s_khi[e - 1];
                head[j] = prev != key[j];
            }
        }
        wave_heads += (u32)__builtin_popcountll(__ballot(head[j]));
    }
    if (lane == 0) s_wtot[w] = wave_heads;
    for (u32 i = tid; i < CAP / 32; i += THREADS) s_bm[i] = 0;
    __syncthreads();
    if (tid == 0) {
        u32 run = 0;
        for (int ww = 0; ww < NW; ++ww) { u32 t = s_wtot[ww]; s_wtot[ww] = run; run += t; }
        s_wtot[NW] = run;
    }
    __syncthreads();
    const u32 d = s_wtot[NW];
    if (mg.cs) {  // `self |= other`: slots hold the run sorted by (suffix, index); index < cs = came from self
        const u32 cs = mg.cs[r];
        // ordered compaction of the slots selected by `sel` to dst[base + rank]
        auto compact = [&](const bool (&sel)[ITEMS], u64* dlo, u64* dhi, u64 base) {
            u32 wk = 0;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) wk += (u32)__builtin_popcountll(__ballot(sel[j]));
            __syncthreads();
            if (lane == 0) s_wtot[w] = wk;
            __syncthreads();
            u32 run = 0;
            for (u32 ww = 0; ww < w; ++ww) run += s_wtot[ww];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const u64 bal = __ballot(sel[j]);
                if (sel[j]) {
                    dlo[base + run + mbcnt(bal)] = key[j].lo;
                    if constexpr (WS) dhi[base + run + mbcnt(bal)] = key[j].hi;
                }
                run += (u32)__builtin_popcountll(bal);
            }
        };
        bool sel[ITEMS];
        if (mg.okind[r] == KIND_VEC) {  // the reference's iter_sorted leaves other's Vec sorted
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) sel[j] = valid[j] && idx[j] >= cs;
            compact(sel, mg.o_lo, mg.o_hi, mg.ostart[r]);
        }
        u64* slo = lo;
        u64* shi = reinterpret_cast<u64*>(hi);
        if (res_trie) {  // Trie |= anything: sorted union
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) sel[j] = head[j];
            compact(sel, slo, shi, s0);
        } else {         // Vec |= anything: sorted(self) ++ sorted(other \ self); every self element is a head
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) sel[j] = head[j] && idx[j] < cs;
            compact(sel, slo, shi, s0);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) sel[j] = head[j] && idx[j] >= cs;
            compact(sel, slo, shi, s0 + cs);
        }
        if (tid == 0) {
            out_count[r] = d;
            out_kind[r] = res_trie ? KIND_TRIE : KIND_VEC;
        }
        return;
    }
    const bool trie = d > VEC_THRESHOLD || res_trie;
    if (trie) {
        u32 run = s_wtot[w];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u64 bal = __ballot(head[j]);
            if (head[j]) store_sfx<WS, HiT>(lo, hi, s0 + run + mbcnt(bal), key[j]);
            run += (u32)__builtin_popcountll(bal);
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITEMS; ++j)
            if (head[j]) atomicOr(&s_bm[idx[j] >> 5], 1u << (idx[j] & 31));
        __syncthreads();
        // original (stream) order compaction: re-read the untouched run
        Sfx<WS> orig[ITEMS];
        bool keep[ITEMS];
        u32 wk = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 e = w * EPW + j * 64 + lane;
            keep[j] = valid[j] && ((s_bm[e >> 5] >> (e & 31)) & 1u);
            if (valid[j]) orig[j] = load_sfx<WS, HiT>(lo, hi, s0 + e, SB);
            wk += (u32)__builtin_popcountll(__ballot(keep[j]));
        }
        if (lane == 0) s_wtot[w] = wk;
        __syncthreads();  // every read of the run is done before any in-place write
        u32 run = 0;
        for (u32 ww = 0; ww < w; ++ww) run += s_wtot[ww];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u64 bal = __ballot(keep[j]);
            if (keep[j]) store_sfx<WS, HiT>(lo, hi, s0 + run + mbcnt(bal), orig[j]);
            run += (u32)__builtin_popcountll(bal);
        }
    }
    if (tid == 0) {
        out_count[r] = d;
        out_kind[r] = trie ? KIND_TRIE : KIND_VEC;
    }
}

// ---- clean spans (round 6; a measured switch, OFF by default: CBLX_SPANS=1): many SHORT runs settled by one workgroup, when none holds a repeat ----
// At PREFIX_BITS = 28 a bucket holds 90 words: 14 M one-wave workgroups at cfg 3 on one GPU, each a chain of launch -> scalar loads -> the
// HBM round trip of 600 bytes -> a dozen LDS phases (3.2 us of wave life at 96 % occupancy, DESIGN_HISTORY.md §3.13) — for runs that, in a
// batch of distinct k-mers, need NOTHING: no repeat means count = run length, kind = Vec, the words stay where they are. A SPAN is a maximal
// stretch of consecutive buckets that are all short new runs (2 .. SPAN_MAX_RUN words, empty index) and start inside one aligned block of
// SPAN_BLOCK arena positions: at most SPAN_CAP words. One workgroup loads the span's words coalesced and inserts a 64-bit fingerprint of
// (bucket number inside the span, suffix) of every word into an LDS table (compare-and-swap, linear probing, two slots per word): if no
// fingerprint was there already, no two words of any bucket of the span are equal — exact — and all its buckets are final. Any hit (a
// repeat, or two words sharing a fingerprint: 2^-43 per span) leaves the span to the kernels below, bucket by bucket, untouched: this kernel
// writes counts and kinds only, never a word. Spans of fewer than SPAN_MIN_RUNS buckets are left alone too (one bucket per workgroup is what
// the other kernels do). k_classify then skips the settled buckets.
static const u32 SPAN_BLOCK = 1024, SPAN_MAX_RUN = 512, SPAN_CAP = SPAN_BLOCK + SPAN_MAX_RUN, SPAN_MIN_RUNS = 3, SPAN_THREADS = 256, SPAN_SLOTS = 4096;
__device__ __forceinline__ bool span_eligible(const u64* __restrict__ raw_start, u64 nb, u64 r) {
    if (r >= nb) return false;
    const u64 c = raw_start[r + 1] - raw_start[r];
    return c >= 2 && c <= SPAN_MAX_RUN;
}
// span heads: an eligible bucket whose predecessor is not eligible or starts in another block; cont[r] = bucket r continues its predecessor's span
__global__ __launch_bounds__(CLASSIFY_THREADS) void k_span_heads(u64 nb, const u64* __restrict__ raw_start, u32* __restrict__ heads, u32* __restrict__ nheads, u8* __restrict__ cont) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1;
    if (r < nb) {
        const bool el = span_eligible(raw_start, nb, r);
        const bool head = el && (r == 0 || !span_eligible(raw_start, nb, r - 1) || raw_start[r - 1] / SPAN_BLOCK != raw_start[r] / SPAN_BLOCK);
        cont[r] = el && !head ? 1 : 0;
        if (head) cls = 0;
    }
    const u32 slot = block_append<CLASSIFY_THREADS, 1>(cls, nheads);
    if (cls == 0) heads[slot] = (u32)r;
}
// (32-bit fingerprints: two words of a span share one with probability n^2 / 2^33 — 3 spans in 10 000 go the long way for nothing — and the table
// is 16 KB: eight workgroups per CU. The chain of dependent HBM round trips is what a span costs, so everything a workgroup needs is requested at
// once: the continuation flags and starts of the buckets behind the head, and SPAN_CAP words from the head's first word on, before it knows where
// the span ends.)
template <bool WS, typename HiT>
__global__ __launch_bounds__(SPAN_THREADS) void k_bucket_span(const u32* __restrict__ heads, const u32* __restrict__ nheads, u64 nb, const u64* __restrict__ raw_start,
                                                            const u8* __restrict__ cont, const u64* __restrict__ lo, const HiT* __restrict__ hi, u32 SB, u32* __restrict__ out_count,
                                                            u8* __restrict__ out_kind, u8* __restrict__ span_done) {
    constexpr int ITEMS = (SPAN_CAP + SPAN_THREADS - 1) / SPAN_THREADS;  // 6
    constexpr int LOOK = (SPAN_BLOCK / 2 + 2 + SPAN_THREADS - 1) / SPAN_THREADS;  // buckets behind the head, per thread (runs of two words: 513 of them) = 3
    constexpr u32 HW = SPAN_CAP / 64 + 1;
    __shared__ u32 s_tab[SPAN_SLOTS];
    __shared__ u64 s_heads[HW + 1];  // bit q: a bucket starts at word q of the span
    __shared__ u32 s_hpre[HW + 1];
    __shared__ u32 s_r1, s_end, s_hit;
    if (blockIdx.x >= *nheads) return;
    const u32 tid = threadIdx.x;
    const u64 r0 = heads[blockIdx.x];
    const u64 total = raw_start[nb];
    const u64 a0 = raw_start[r0];
    if (tid == 0) { s_r1 = 0xFFFFFFFFu; s_end = 0; s_hit = 0; }
    for (u32 i = tid; i < HW + 1; i += SPAN_THREADS) s_heads[i] = 0;
    for (u32 i = tid * 4; i < SPAN_SLOTS; i += SPAN_THREADS * 4) *reinterpret_cast<uint4*>(&s_tab[i]) = uint4{~0u, ~0u, ~0u, ~0u};
    // -- one batch of requests: flags + starts of the buckets behind the head, the words
    u32 ct[LOOK];
    u64 st[LOOK];
#pragma unroll
    for (int k = 0; k < LOOK; ++k) {
        const u64 r = r0 + 1 + tid + (u32)k * SPAN_THREADS;
        ct[k] = r < nb ? cont[r] : 0u;
        st[k] = raw_start[r <= nb ? r : nb];
    }
    const u64* __restrict__ wl = lo + a0;
    const HiT* __restrict__ wh = WS ? hi + a0 : nullptr;
    Sfx<WS> key[ITEMS];
    const u32 nmax = total - a0 < SPAN_CAP ? (u32)(total - a0) : SPAN_CAP;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 e = j * SPAN_THREADS + tid;
        key[j] = load_sfx<WS, HiT>(wl, wh, e < nmax ? e : 0u, SB);
    }
    __syncthreads();
    // -- the span's members: buckets r0 .. r0 + s_r1 (s_r1 = the first bucket behind the head that does not continue it)
#pragma unroll
    for (int k = 0; k < LOOK; ++k)
        if (!ct[k]) { atomicMin(&s_r1, tid + (u32)k * SPAN_THREADS); break; }
    __syncthreads();
    const u32 nruns = s_r1 + 1u;
    if (nruns < SPAN_MIN_RUNS) return;
#pragma unroll
    for (int k = 0; k < LOOK; ++k) {
        const u32 i = tid + (u32)k * SPAN_THREADS;
        if (i < s_r1) atomicOr(reinterpret_cast<unsigned long long*>(&s_heads[(st[k] - a0) >> 6]), 1ull << ((st[k] - a0) & 63));
        else if (i == s_r1) s_end = (u32)(st[k] - a0);  // where the bucket behind the span starts
    }
    __syncthreads();
    const u32 n = s_end;  // <= SPAN_CAP
    if (tid < 64) {  // exclusive prefix of the bucket-start bits per 64-bit word
        const u32 v = tid < HW ? (u32)__builtin_popcountll(s_heads[tid]) : 0u;
        const u32 inc = wave_inclusive_scan(v);
        if (tid < HW) s_hpre[tid] = inc - v;
    }
    __syncthreads();
    bool hit = false;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 e = j * SPAN_THREADS + tid;
        if (e < n && !*reinterpret_cast<volatile u32*>(&s_hit)) {
            // bucket number of word e inside the span = bucket starts at or before e (the head's own start carries no bit)
            const u32 b = s_hpre[e >> 6] + (u32)__builtin_popcountll(s_heads[e >> 6] & ((e & 63) == 63 ? ~0ull : ((2ull << (e & 63)) - 1ull)));
            u64 f64 = key[j].lo * 0x9E3779B97F4A7C15ull + (u64)b * 0xD6E8FEB86659FD93ull;
            if constexpr (WS) f64 ^= (key[j].hi + 0x2545F4914F6CDD1Dull) * 0xBF58476D1CE4E5B9ull;
            f64 ^= f64 >> 29; f64 *= 0x94D049BB133111EBull; f64 ^= f64 >> 32;
            u32 f = (u32)f64;
            if (f == ~0u) f = 0;  // (the empty mark)
            u32 h = (u32)(f64 >> 40) & (SPAN_SLOTS - 1);
            for (;;) {
                const u32 old = atomicCAS(&s_tab[h], ~0u, f);
                if (old == ~0u) break;
                if (old == f) { hit = true; break; }
                h = (h + 1u) & (SPAN_SLOTS - 1);
            }
            if (hit) s_hit = 1u;  // (the others stop early: a batch full of repeats pays for a fraction of its spans)
        }
    }
    if (__syncthreads_or((hit || s_hit) ? 1 : 0)) return;  // a repeat (or a shared fingerprint): the span's buckets take the kernels below
    // (the members' lengths from the starts this thread holds: bucket r0 + 1 + i runs from st[k] to its successor's start — re-read, L2-warm)
    for (u32 i = tid; i < nruns; i += SPAN_THREADS) {
        const u64 r = r0 + i;
        out_count[r] = (u32)(raw_start[r + 1] - raw_start[r]);
        out_kind[r] = KIND_VEC;
        span_done[r] = 1;
    }
}

// ---- KRN-3 fast path: one workgroup per bucket, run <= CAP. Suffixes of one prefix are close to uniformly spread,
// so one counting-sort step on their top ceil(log2 c) bits leaves sub-buckets of ~1 element; each element then ranks
// itself inside its sub-bucket by (suffix, stream index) with a handful of compares. O(c) LDS work instead of
// SUFFIX_BITS/8 radix passes. A bucket whose largest sub-bucket exceeds MSD_LIMIT (heavy duplication / repeats) is
// handed to the radix kernel through `retry` untouched. -------------------------------------------------------------
#ifndef CBLX_MSD_LIMIT
#define CBLX_MSD_LIMIT 48
#endif
// (suffixes wider than 64 bits — K >= 57 or so — come with longer reads and clusters of up to K mates: their fallback is a radix sort
// of 13 passes, so they give up later: cfg 4 at the bucket depth of an 8-GPU job 87 -> 73 ms)
static const u32 MSD_LIMIT = CBLX_MSD_LIMIT, MSD_LIMIT_WIDE = 2 * CBLX_MSD_LIMIT, MSD_LIMIT_HASHED = 16;

template <bool WS> __device__ __forceinline__ u32 sfx_top_bits(const Sfx<WS>& k, u32 SB, u32 nbits) {
    if constexpr (WS) {
        const u128 v = ((u128)k.hi << 64) | k.lo;
        return (u32)(v >> (SB - nbits));
    } else {
        return (u32)(k.lo >> (SB - nbits));
    }
}
// the nbits below the top `skip` suffix bits (skip + nbits <= SB, nbits <= 16)
template <bool WS> __device__ __forceinline__ u32 sfx_bits_below(const Sfx<WS>& k, u32 SB, u32 skip, u32 nbits) {
    const u32 sh = SB - skip - nbits;
    if constexpr (WS) {
        const u128 v = ((u128)k.hi << 64) | k.lo;
        return (u32)(v >> sh) & ((1u << nbits) - 1u);
    } else {
        return (u32)(k.lo >> sh) & ((1u << nbits) - 1u);
    }
}
// Sub-bucket of a bucket that can only end up a Vec (run length <= threshold, not a Trie yet): any function of the suffix
// will do, and a hash of ALL its bits spreads what the top bits do not — consecutive k-mers of a read tend to share the
// leading bits of their necklace, prefix and top suffix bits alike (the locality CBL is built on).
template <bool WS> __device__ __forceinline__ u32 sfx_hash_bits(const Sfx<WS>& k, u32 nbits) {
    u64 v = k.lo;
    if constexpr (WS) v ^= k.hi * 0xD6E8FEB86659FD93ull;
    v ^= v >> 32;
    v *= 0x9E3779B97F4A7C15ull;
    return nbits ? (u32)(v >> (64 - nbits)) : 0u;
}
template <bool WS> __device__ __forceinline__ bool sfx_less(const Sfx<WS>& a, u32 ia, const Sfx<WS>& b, u32 ib) {
    if constexpr (WS) {
        if (a.hi != b.hi) return a.hi < b.hi;
    }
    if (a.lo != b.lo) return a.lo < b.lo;
    return ia < ib;
}

// PACKED (SUFFIX_BITS + 12 <= 64, narrow suffix): an element is ONE u64 = suffix << 12 | stream index, so every LDS
// access, compare and move handles key and index together (the kernel is LDS-throughput-bound).
static const u32 PK_BITS = 12;  // CAP <= 4096
#ifndef CBLX_MSD_TRIP
#define CBLX_MSD_TRIP 3  // measured at cfg 2: 2 -> 6.87 ms, 3 -> 6.66 ms, 4 -> 6.86 ms, 8 -> +0.8 ms
#endif
static const int MSD_TRIP = CBLX_MSD_TRIP;  // sub-bucket entries read per trip of the ranking loop (<= the 4 slack entries)
#ifndef CBLX_MSD_REUSE_BASE
#define CBLX_MSD_REUSE_BASE 0  // measured (profiles/r02_variants.md): +0.15 ms — eight more live registers cost more than the LDS read
#endif
#ifndef CBLX_MSD_SKIP_SELF
#define CBLX_MSD_SKIP_SELF 0  // measured (cfg 2): 6.73 -> 8.23 ms — per-read lane masks cost more issue time than the bank conflicts they avoid
#endif
#ifndef CBLX_MSD_SKIP_SINGLE
#define CBLX_MSD_SKIP_SINGLE 1  // a lane alone in its sub-bucket does not enter the ranking loop
#endif
#ifndef CBLX_MSD_PROBE
#define CBLX_MSD_PROBE 0  // > 0: timing probes that leave phases out (tools/variants.sh); never in the product build
#endif
#ifndef CBLX_MSD_PROBE_MERGE
#define CBLX_MSD_PROBE_MERGE 0  // 1: probes 1 - 3 only touch `self |= other` launches (the two indexes in front of a merge bench stay right); 4 / 5 are merge-only anyway
#endif
#if (CBLX_MSD_PROBE || CBLX_ENC_PROBE) && !defined(CBLX_TIMING_PROBES)
#error "CBLX_MSD_PROBE / CBLX_ENC_PROBE leave phases out and produce wrong results: timing builds only (-DCBLX_TIMING_PROBES)"
#endif
#ifndef CBLX_MSD_MERGE_REUSE
#define CBLX_MSD_MERGE_REUSE 0
#endif
#ifndef CBLX_MSD_MERGE_WAVES
#define CBLX_MSD_MERGE_WAVES 7
#endif
#ifndef CBLX_MSD_PRECHECK
#define CBLX_MSD_PRECHECK 0  // measured (round 6, tools/dev_msd_bench.cpp at PREFIX_BITS = 28): 4.203 against 4.209 ms — the class is not bound by what follows the loads
#endif
#ifndef CBLX_MSD_WAVES
#define CBLX_MSD_WAVES 7  // waves per SIMD the register allocation aims at (LDS allows 7 workgroups of the 256-thread class; 76 -> 72 VGPRs: 7.28 -> 7.10 ms)
#endif

// MERGE: the instantiation `self |= other` launches (its epilogue keeps a dozen more registers live, which the build's
// instantiation must not pay for: 6.8 -> 7.0 ms at cfg 2 when the two shared one kernel)
// WIDE PACKED (suffixes wider than 64 bits, SUFFIX_BITS + 12 <= 128): an element is ONE 16-byte LDS slot = suffix << 12 | stream
// index, read with one ds_read_b128 and compared as a 128-bit number — instead of three arrays (lo, hi, 16-bit index) and three
// LDS reads per comparison. 16 instead of 18 bytes per slot also lets two 4096-slot workgroups share a CU instead of one.
struct __attribute__((aligned(16))) W128 { u64 lo, hi; };
__device__ __forceinline__ W128 w128_pack(const Sfx<true>& k, u32 idx) { return W128{(k.lo << 12u) | idx, (k.hi << 12u) | (k.lo >> (64 - 12u))}; }
__device__ __forceinline__ Sfx<true> w128_sfx(const W128& w) { Sfx<true> s; s.lo = (w.lo >> 12u) | (w.hi << (64 - 12u)); s.hi = w.hi >> 12u; return s; }
__device__ __forceinline__ bool w128_less(const W128& a, const W128& b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
__device__ __forceinline__ bool w128_same_sfx(const W128& a, const W128& b) { return a.hi == b.hi && ((a.lo ^ b.lo) >> 12u) == 0; }
template <bool WS> __host__ __device__ inline bool msd_takes(u32 SB) { return !WS || SB + 12u <= 128; }  // else: the LDS radix kernel

// The occupancy every instantiation is compiled FOR = what the register allocator can meet (round 5: all of them asked for 7 and 26 of
// them got 4 - 6 with a warning each, so a register regression in one of them would have gone unnoticed): 16-byte elements (WS) take 4
// waves per SIMD (5 in the shortest build class that has them), the unpacked 2048- and 4096-slot classes and the packed 4096-slot one
// 5 - 6, everything else CBLX_MSD_WAVES.
template <int CAP, bool PACKED, bool WS, bool MERGE> constexpr int msd_waves() {
    if (WS) return CAP <= 128 ? CBLX_MSD_WAVES : ((CAP <= 512 && !MERGE) ? 5 : 4);
    if (CAP >= 4096) return PACKED ? 6 : 5;
    if (CAP >= 2048 && !PACKED) return MERGE ? 5 : 6;
    if (MERGE && CAP > 128) return CBLX_MSD_MERGE_WAVES;
    return CBLX_MSD_WAVES;
}
template <int THREADS, int CAP, bool PACKED, bool WS, typename HiT, bool MERGE = false>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(msd_waves<CAP, PACKED, WS, MERGE>(), 8))) void k_bucket_msd(const BDesc* __restrict__ list, const u32* __restrict__ list_n,
                                                        u64* __restrict__ lo, HiT* __restrict__ hi, u32 SB,
                                                        u32* __restrict__ out_count, u8* __restrict__ out_kind,
                                                        BDesc* __restrict__ retry, u32* __restrict__ retry_n, MergeArgs mg = MergeArgs{},
                                                        u8* __restrict__ bail_flag = nullptr, u32* __restrict__ bail_any = nullptr) {
    static_assert(!(PACKED && WS), "packed elements need a narrow suffix");
    static_assert(CAP <= (1 << PK_BITS), "stream index must fit PK_BITS");
    constexpr int ITEMS = CAP / THREADS, NW = THREADS / 64;
    constexpr bool WP = WS;  // wide suffixes: 16-byte packed elements (callers route SUFFIX_BITS > 116 to the radix kernel: msd_takes)
    __shared__ u64 s_klo[WP ? 1 : CAP + 4];  // + slack: the ranking loop reads up to 3 entries past a sub-bucket
    __shared__ W128 s_kw[WP ? CAP + 4 : 1];
    __shared__ u16 s_idx[(PACKED || WP) ? 1 : CAP + 4];
    // sub-bucket counts, then exclusive offsets: 16-bit entries (values <= CAP), counted with 32-bit LDS atomics on the
    // containing dword (LDS is what bounds residency here)
    __shared__ u32 s_off32[CAP / 2 + 2];
    u16* s_off = reinterpret_cast<u16*>(s_off32);
    __shared__ u32 s_scan[NW + 1];
    __shared__ u32 s_wtot[NW + 1];
    __shared__ u32 s_max;
    __shared__ u32 s_sel[MERGE ? NW : 1];  // merge epilogue: per-wave counts of its three selections (10-bit fields)
    static_assert(!MERGE || 64 * ITEMS < 1024, "the merge epilogue counts a wave's selections in 10-bit fields: a wave owns fewer than 1024 slots");
    // distinctness pre-check of the shortest class (below): 2^15 bits
    constexpr bool PRECHECK = CBLX_MSD_PRECHECK && CAP <= 128 && !MERGE;
    __shared__ u32 s_bits[PRECHECK ? 1024 : 1];

    if (blockIdx.x >= *list_n) return;
    const BDesc dsc = list[blockIdx.x];
    const u32 r = dsc.r;
    const u64 s0 = dsc.start;
    const u32 c = dsc.c & BDESC_LEN_MASK;
    const u32 skip = (dsc.c & BDESC_SKIP_MASK) >> BDESC_SKIP_SHIFT;  // leading suffix bits shared by the whole run (sub-range of a big run)
    const bool res_trie = (dsc.c & BDESC_TRIE) != 0;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    if (c == 0) {  // an empty sub-range of a big run (k_big_split)
        if (tid == 0) { out_count[r] = 0; out_kind[r] = KIND_TRIE; }
        return;
    }
    const u32 R = (c + THREADS - 1) / THREADS;
    const u32 EPW = 64 * R;  // wave-contiguous slices: ballots then compact in stream order
    // The lane's first slot; slot j is + 64 j. In the MERGE instantiation every phase takes it through an empty asm, so that the compiler
    // adds 64 j again where it needs a slot index instead of keeping the eight sums alive from the loads to the epilogue — with the
    // merge epilogue on top they were what spilled (8 registers, 36 bytes of scratch per lane, 3.7 GB of scratch writes per merge by the
    // TCC counters). The build's instantiations keep their code.
    const u32 slot_first = w * EPW + lane;
    auto slot0 = [&]() { u32 v = slot_first; if constexpr (MERGE) asm volatile("" : "+v"(v)); return v; };
    u32 nbits = 32 - __builtin_clz(c - 1 > 0 ? c - 1 : 1);  // ceil(log2 c)
    if (nbits > SB - skip) nbits = SB - skip;
    const u32 NB = 1u << nbits;  // <= CAP because c <= CAP
    // `self |= other` (mg.cs set): the run is [self's suffixes][other's], both parts distinct; every outcome needs the sorted
    // order, so the sub-buckets are always by the top bits
    const bool merging = MERGE && mg.cs != nullptr;
    // no sorted output needed: sub-buckets by hash. Never in the MERGE instantiation: its other caller (the sub-ranges of a long run,
    // k_big_vlist) always asks for the sorted list — as a run-time value there, the hashed variants of the sub-bucket index and of the
    // ranking loop sat behind a branch per slot in the kernel `self |= other` spends its time in
    const bool vec_only = !MERGE && c <= VEC_THRESHOLD && !res_trie;
    // Largest sub-bucket the ranking loop is worth running on. Hashed sub-buckets of distinct suffixes stay below 10
    // entries, so more than MSD_LIMIT_HASHED means repeats (every copy of a value lands in its sub-bucket): such runs are
    // deduplicated far cheaper by the claim table. Top-bit sub-buckets reach 45 entries without a single repeat (necklace
    // clusters, DESIGN_HISTORY.md §3.7) and give up later. A compile-time constant per length class: the runs of the classes up to
    // 1024 words are the hashed ones (a Trie that short only comes out of a loaded file; as a run-time value the limit cost
    // the 2048-slot instantiation four spilled registers and cfg 2 0.3 ms).
    constexpr u32 crowd = (CAP <= (int)VEC_THRESHOLD && !MERGE) ? MSD_LIMIT_HASHED : (WS ? MSD_LIMIT_WIDE : MSD_LIMIT);

    for (u32 i = tid; i < NB / 2 + 1; i += THREADS) s_off32[i] = 0;
    if (tid == 0) s_max = 0;
    __syncthreads();
    Sfx<WS> key[ITEMS];
    u32 sub[ITEMS], arr[ITEMS];
    bool valid[ITEMS];
    // all loads first, unconditionally (slots past the run re-read its first element): eight independent global loads in
    // flight per lane instead of eight load -> wait -> use round trips inside per-slot branches
    u32 cs_m = 0;     // merge: self's length (index < cs_m = the element came from self)
    if constexpr (MERGE) if (merging) cs_m = mg.cs[r];
    if constexpr (MERGE) {
        // One address form for the three cases: element e is word e of `pa` (e < split) or of `pb`. Gathered run / no merge: both are the
        // run. Direct merge: self's part where self stores it, other's where other does, the latter's pointer moved back by cs so that
        // the element index needs no subtraction (two selects and one 64-bit multiply-add per slot; the case distinction per slot that
        // stood here cost twice that). (HiT is u64 whenever the suffix is wide.)
        static_assert(!WS || sizeof(HiT) == 8, "wide suffixes keep their high part in 64-bit words");
        const u64* pa = lo + s0;
        const u64* pb = pa;
        const u64* ha = WS ? reinterpret_cast<const u64*>(hi) + s0 : nullptr;
        const u64* hb = ha;
        u32 split = 0;
        if (merging && mg.s_lo) {
            split = cs_m;
            const u64 a_self = mg.sstart[r], a_oth = mg.ostart[r];
            pa = mg.s_lo + a_self;
            pb = reinterpret_cast<const u64*>(reinterpret_cast<uintptr_t>(mg.o_lo + a_oth) - 8ull * split);
            if constexpr (WS) {
                ha = mg.s_hi + a_self;
                hb = reinterpret_cast<const u64*>(reinterpret_cast<uintptr_t>(mg.o_hi + a_oth) - 8ull * split);
            }
        }
        const u32 sl1 = slot0();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 e = sl1 + j * 64;
            valid[j] = (u32)j < R && e < c;
            const u32 ee = valid[j] ? e : 0u;
            const bool in_a = ee < split;
            key[j] = load_sfx<WS, u64>(in_a ? pa : pb, in_a ? ha : hb, ee, SB);
        }
    } else {
        const u32 sl2 = slot0();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 e = sl2 + j * 64;
            valid[j] = (u32)j < R && e < c;
            const u32 ee = valid[j] ? e : 0u;
            key[j] = load_sfx<WS, HiT>(lo, hi, s0 + ee, SB);
        }
    }
    if constexpr (PRECHECK) {
        // (measured switch, off) A run of at most 128 words that can only end up a Vec (cfg 3 on one GPU: 9.8 M of them, 77 words each, a wave
        // per run) and holds no repeat needs none of the phases below: nothing is written, the count is the run length. A test that can only err
        // on the safe side: every word sets the bit of a 15-bit hash of its suffix in a 4 KB LDS table (atomicOr returns the old word); if no bit
        // was already set, no two suffixes are equal (91 % of the runs of 77 distinct words) and the run is done; otherwise the counting sort
        // decides exactly as before. Bit-identical, and worth nothing: the wave's life (3.2 us at 96 % occupancy) is its launch, its scalar loads
        // and the HBM round trip of its words, not the 400 instructions this saves (DESIGN_HISTORY.md §3.13).
        if (vec_only) {
            for (u32 i = tid * 4; i < 1024; i += THREADS * 4) *reinterpret_cast<uint4*>(&s_bits[i]) = uint4{0, 0, 0, 0};
            __syncthreads();
            bool hit = false;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const u32 h = sfx_hash_bits<WS>(key[j], 15);
                if (valid[j]) hit |= (atomicOr(&s_bits[h >> 5], 1u << (h & 31u)) >> (h & 31u)) & 1u;
            }
            if (!__syncthreads_or(hit ? 1 : 0)) {
                if (tid == 0) { out_count[r] = c; out_kind[r] = KIND_VEC; }
                return;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        sub[j] = vec_only ? sfx_hash_bits<WS>(key[j], nbits) : sfx_bits_below<WS>(key[j], SB, skip, nbits);
        arr[j] = 0;
        if (valid[j]) arr[j] = atomicAdd(&s_off32[sub[j] >> 1], 1u << ((sub[j] & 1u) * 16u));  // raw dword; the field is cut out below
    }
    bool crowded = false;  // an arrival number of `crowd` = a sub-bucket of more than `crowd` entries
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        arr[j] = (arr[j] >> ((sub[j] & 1u) * 16u)) & 0xFFFFu;
        crowded |= arr[j] >= crowd;  // (slots past the run keep arr = 0)
    }
    if (crowded) s_max = crowd + 1u;  // benign race: every writer stores the same value
    __syncthreads();
    if (s_max > crowd) {  // crowded sub-bucket: the run goes to the claim-table kernel (build) / the radix kernel (merge, sub-ranges)
        // (a batch at high coverage sends nearly EVERY run this way: a million appends to one list counter serialise in the
        // L2 — 3 ms — so the build marks the list entry instead and the next kernel walks the same list)
        if (tid == 0) {
            if (bail_flag) { bail_flag[blockIdx.x] = 1; *bail_any = 1u; }
            else retry[atomicAdd(retry_n, 1u)] = dsc;
        }
        return;
    }
    {   // exclusive scan of the NB counts; each thread owns `per` consecutive entries
        const u32 per = (NB + THREADS - 1) / THREADS;  // <= ITEMS
        const u32 b0 = tid * per;
        u32 sum = 0, cnt[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {  // counts stay in registers: the write-back below does not re-read them
            cnt[k] = ((u32)k < per && b0 + k < NB) ? s_off[b0 + k] : 0u;
            sum += cnt[k];
        }
        u32 ex = block_exclusive_scan<THREADS, u32>(sum, s_scan, nullptr);
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            if ((u32)k < per && b0 + k < NB) { s_off[b0 + k] = (u16)ex; ex += cnt[k]; }
        }
    }
    __syncthreads();
#if CBLX_MSD_PROBE == 3  // timing probe only: loads, counting atomics and scan alone
    if (!CBLX_MSD_PROBE_MERGE || merging) {
        if (tid == 0) { out_count[r] = c; out_kind[r] = KIND_VEC; }
        if (key[0].lo != 0x1234567ull) return;
    }
#endif
    if (tid == 0) s_off[NB] = (u16)c;
    if constexpr (PACKED) { if (tid < 4) s_klo[c + tid] = ~0ull; }  // the slack the ranking loop may read compares greater than every element
    if constexpr (WP) { if (tid < 4) s_kw[c + tid] = W128{~0ull, ~0ull}; }
    // scatter into sub-bucket order (arrival order inside a sub-bucket is arbitrary); the offsets are fetched for all
    // slots at once, outside the per-slot branches (sub[] is in range for unused slots too)
    u32 sbase[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) sbase[j] = s_off[sub[j]];
    const u32 sl3 = slot0();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
#if CBLX_MSD_SKIP_SELF
        arr[j] += sbase[j];  // the element's own slot, kept for the ranking loop (same register as the arrival order)
#endif
        if (valid[j]) {
            const u32 e = sl3 + j * 64;
#if CBLX_MSD_SKIP_SELF
            const u32 p = arr[j];
#else
            const u32 p = sbase[j] + arr[j];
#endif
            if constexpr (PACKED) {
                s_klo[p] = (key[j].lo << PK_BITS) | e;
            } else if constexpr (WP) {
                s_kw[p] = w128_pack(key[j], e);
            } else {
                s_klo[p] = key[j].lo;
                s_idx[p] = (u16)e;
            }
        }
    }
    __syncthreads();
    // every element looks at its sub-bucket: rank by (suffix, stream index); head = no equal suffix with a smaller index
    u32 fin[ITEMS];
    bool head[ITEMS];
    u32 wave_heads = 0;
    u32 sa[ITEMS], sb[ITEMS];  // sub-bucket bounds of every item, fetched in one batch of independent LDS reads
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if constexpr (CBLX_MSD_REUSE_BASE || (MERGE && CBLX_MSD_MERGE_REUSE)) sa[j] = sbase[j];  // the offsets did not change since the scatter read them: one LDS read per item less
        else sa[j] = s_off[sub[j]];
        sb[j] = s_off[sub[j] + 1];
    }
    const u32 sl4 = slot0();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        head[j] = false;
        fin[j] = 0;
        if (valid[j]) {
            const u32 e = sl4 + j * 64;
#if CBLX_MSD_PROBE >= 1 && CBLX_MSD_PROBE <= 2  // timing probe only (wrong order): no ranking reads
            const u32 b = sb[j], a = (!CBLX_MSD_PROBE_MERGE || merging) ? b : ((b - sa[j] > 1u) ? sa[j] : b);
            const u32 a0 = sa[j];
#elif CBLX_MSD_SKIP_SINGLE
            const u32 b = sb[j], a = (b - sa[j] > 1u) ? sa[j] : b;  // alone: rank 0, no duplicate, nothing to read
            const u32 a0 = sa[j];
#else
            const u32 a = sa[j], b = sb[j], a0 = a;
#endif
            u32 rank = 0;
            bool dup = false;
            // four entries per trip: the reads are independent, so a sub-bucket (1.5 elements on average, 4-5 for the
            // slowest lane of a wave) costs one LDS round trip instead of one per element
            if constexpr (PACKED) {
                const u64 me = (key[j].lo << PK_BITS) | e;
#if CBLX_MSD_SKIP_SELF
                // only the OTHER entries of the sub-bucket are read, under the lane's own mask: a lane alone in its
                // sub-bucket (about half of them) issues no LDS access at all, one with a single mate issues one — the
                // kernel is bound by the bank conflicts of these random reads (profiles/r02_sq_counters.md)
                for (u32 q = a; q < b; q += MSD_TRIP) {
                    u64 o[MSD_TRIP];
#pragma unroll
                    for (int k = 0; k < MSD_TRIP; ++k) {
                        const u32 ix = q + k;
                        o[k] = ~0ull;  // compares greater than every element
                        if (ix < b && ix != arr[j]) o[k] = s_klo[ix];
                    }
#pragma unroll
                    for (int k = 0; k < MSD_TRIP; ++k) {
                        const bool less = o[k] < me;
                        rank += less ? 1u : 0u;
                        dup |= less && ((o[k] >> PK_BITS) == key[j].lo);
                    }
                }
#else
                if (vec_only) {  // hashed sub-buckets: what follows a sub-bucket is unrelated, entries past b are masked
                    for (u32 q = a; q < b; q += MSD_TRIP) {
                        u64 o[MSD_TRIP];
#pragma unroll
                        for (int k = 0; k < MSD_TRIP; ++k) o[k] = s_klo[q + k];  // s_klo has 4 slack entries
#pragma unroll
                        for (int k = 0; k < MSD_TRIP; ++k) {
                            const bool less = (q + k < b) && o[k] < me;
                            rank += less ? 1u : 0u;
                            dup |= less && ((o[k] >> PK_BITS) == key[j].lo);
                        }
                    }
                } else {  // sub-buckets by the top suffix bits are in ascending order: whatever follows b (the next
                          // sub-bucket, or the all-ones slack) is greater than `me` and needs no bound check
                    for (u32 q = a; q < b; q += MSD_TRIP) {
                        u64 o[MSD_TRIP];
#pragma unroll
                        for (int k = 0; k < MSD_TRIP; ++k) o[k] = s_klo[q + k];
#pragma unroll
                        for (int k = 0; k < MSD_TRIP; ++k) {
                            const bool less = o[k] < me;
                            rank += less ? 1u : 0u;
                            dup |= less && ((o[k] >> PK_BITS) == key[j].lo);
                        }
                    }
                }
#endif
            } else if constexpr (WP) {
                const W128 me = w128_pack(key[j], e);
                if (vec_only) {  // hashed sub-buckets: what follows a sub-bucket is unrelated, entries past b are masked
                    for (u32 q = a; q < b; q += 2) {
                        W128 o[2];
#pragma unroll
                        for (int k = 0; k < 2; ++k) o[k] = s_kw[q + k];  // 4 slack entries
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const bool less = (q + k < b) && w128_less(o[k], me);
                            rank += less ? 1u : 0u;
                            dup |= less && w128_same_sfx(o[k], me);
                        }
                    }
                } else {  // ascending sub-buckets: whatever follows b (the next sub-bucket, or the all-ones slack) is greater than `me`
                    for (u32 q = a; q < b; q += 2) {
                        W128 o[2];
#pragma unroll
                        for (int k = 0; k < 2; ++k) o[k] = s_kw[q + k];
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const bool less = w128_less(o[k], me);
                            rank += less ? 1u : 0u;
                            dup |= less && w128_same_sfx(o[k], me);
                        }
                    }
                }
            } else {
                for (u32 q = a; q < b; q += 2) {
                    Sfx<WS> o[2];
                    u32 oi[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        o[k].lo = s_klo[q + k];
                        oi[k] = (u32)s_idx[q + k];
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const bool less = (q + k < b) && sfx_less<WS>(o[k], oi[k], key[j], e);
                        rank += less ? 1u : 0u;
                        dup |= less && o[k] == key[j];
                    }
                }
            }
            fin[j] = a0 + rank;
            head[j] = !dup;
        }
        wave_heads += (u32)__builtin_popcountll(__ballot(head[j]));
    }
    if constexpr (MERGE) if (merging) {
        // Every element goes to its rank by (suffix, index): the slots then hold the run sorted, self's copy of a suffix in front of
        // other's (index < cs = came from self), and the rules of src/trievec/set_ops.rs:43-71 are ordered compactions of slot subsets,
        // as in k_bucket_medium's merge epilogue. What a slot has to say about its element is two bits, both known here: which side it
        // came from, and whether it is the first of its suffix (the ranking loop's `dup`: an equal suffix with a smaller index — the
        // other side's copy of a word self holds). Round 5: the slots carried the 12-bit index instead, every thread read its slot AND
        // the one in front of it to find the heads again, and the (up to) three compactions each had their own pair of barriers.
#if CBLX_MSD_PROBE == 4  // timing probe only: no merge epilogue at all
        if (tid == 0) { out_count[r] = c; out_kind[r] = KIND_VEC; }
        if (fin[0] != 0x12345u) return;
#endif
        __syncthreads();  // every read of the sub-bucket order is done
        const u32 sl5 = slot0();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if (valid[j]) {
                const u32 e = sl5 + j * 64;
                const u32 fl = (e < cs_m ? 2u : 0u) | (head[j] ? 1u : 0u);
                if constexpr (PACKED) {
                    s_klo[fin[j]] = (key[j].lo << PK_BITS) | fl;
                } else if constexpr (WP) {
                    s_kw[fin[j]] = w128_pack(key[j], fl);
                } else {
                    s_klo[fin[j]] = key[j].lo;
                    s_idx[fin[j]] = (u16)fl;
                }
            }
        }
        __syncthreads();
        const bool o_vec = mg.okind[r] == KIND_VEC;  // the reference's iter_sorted leaves other's Vec sorted
        // three ordered selections of the slots, counted and written together:
        //   O = other's elements (back to other's arena, sorted)            A = Trie |= x: the heads (sorted union); Vec |= x: self's elements
        //   B = Vec |= x only: other's heads = other \ self, behind self's   (every self element is a head)
        u32 fl[ITEMS];
        u32 nO = 0;
        const u32 sl6 = slot0();
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 p = sl6 + j * 64;
            const bool live = (u32)j < R && p < c;
            const u32 pc = live ? p : 0u;
            u32 f;
            if constexpr (PACKED) {
                const u64 v = s_klo[pc];
                key[j].lo = v >> PK_BITS;
                f = (u32)v & 3u;
            } else if constexpr (WP) {
                const W128 v = s_kw[pc];
                key[j] = w128_sfx(v);
                f = (u32)v.lo & 3u;
            } else {
                key[j].lo = s_klo[pc];
                f = s_idx[pc];
            }
            // bit 0: in O; bit 10: in A; bit 20: in B — a lane's eight slots, and then the wave's 512, are counted by plain additions of
            // these words (three 10-bit fields) and ONE wave reduction; a ballot per selection and slot here kept 24 lane masks alive for
            // the write loop below (the compiler re-used them) and spilled half the scalar registers
            const bool self_el = (f & 2u) != 0, hd = (f & 1u) != 0;
            const bool inO = live && !self_el && o_vec, inA = live && (res_trie ? hd : self_el), inB = live && !res_trie && !self_el && hd;
            fl[j] = (inO ? 1u : 0u) | (inA ? (1u << 10) : 0u) | (inB ? (1u << 20) : 0u);
            nO += fl[j];
        }
        nO = wave_reduce_sum(nO);  // (every lane gets the sum) <= 512 per field
        if (lane == 0) s_sel[w] = nO;
        __syncthreads();
        u32 runO = 0, runA = 0, runB = 0, totA = 0, totB = 0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            const u32 t = s_sel[ww], tO = t & 1023u, tA = (t >> 10) & 1023u, tB = t >> 20;
            if ((u32)ww < w) { runO += tO; runA += tA; runB += tB; }
            totA += tA;
            totB += tB;
        }
        u64* slo = lo;
        u64* shi = reinterpret_cast<u64*>(hi);
        const u64 obase = mg.ostart[r];
        runB += cs_m;  // B follows self's cs elements
#if CBLX_MSD_PROBE == 5  // timing probe only: the merge epilogue without its global stores
        if (tid == 0) { out_count[r] = c; out_kind[r] = KIND_VEC; }
        if (runO + runA + runB + fl[0] != 0x12345u) return;
#endif
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u64 balO = __ballot((fl[j] & 1u) != 0), balA = __ballot((fl[j] & (1u << 10)) != 0), balB = __ballot((fl[j] & (1u << 20)) != 0);
            if (fl[j] & 1u) {
                mg.o_lo[obase + runO + mbcnt(balO)] = key[j].lo;
                if constexpr (WS) mg.o_hi[obase + runO + mbcnt(balO)] = key[j].hi;
            }
            if (fl[j] & ((1u << 10) | (1u << 20))) {  // A and B are disjoint
                const u32 q = (fl[j] & (1u << 10)) ? runA + mbcnt(balA) : runB + mbcnt(balB);
                slo[s0 + q] = key[j].lo;
                if constexpr (WS) shi[s0 + q] = key[j].hi;
            }
            runO += (u32)__builtin_popcountll(balO);
            runA += (u32)__builtin_popcountll(balA);
            runB += (u32)__builtin_popcountll(balB);
        }
        if (tid == 0) {
            out_count[r] = res_trie ? totA : cs_m + totB;
            out_kind[r] = res_trie ? KIND_TRIE : KIND_VEC;
        }
        return;
    }
    if (lane == 0) s_wtot[w] = wave_heads;
    __syncthreads();  // also: every read of the sub-bucket order is done
    if (tid == 0) {
        u32 run = 0;
        for (int ww = 0; ww < NW; ++ww) { u32 t = s_wtot[ww]; s_wtot[ww] = run; run += t; }
        s_wtot[NW] = run;
    }
    __syncthreads();
    const u32 d = s_wtot[NW];
#if CBLX_MSD_PROBE == 2  // timing probe only: no sorted write-back (and no ranking reads)
    const bool trie = false;
#else
    const bool trie = d > VEC_THRESHOLD || res_trie;
#endif
    if (!trie) {
        // Vec: first occurrences in stream order, straight from registers (slots are in stream order). Without a
        // duplicate (d == c) every element already sits in its slot and nothing is written.
        u32 run = s_wtot[w];
        if (d != c) {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const u64 bal = __ballot(head[j]);
                if (head[j]) store_sfx<WS, HiT>(lo, hi, s0 + run + mbcnt(bal), key[j]);
                run += (u32)__builtin_popcountll(bal);
            }
        }
    } else {
        // Trie: ascending order. Put every element at its final rank with its head flag, then compact slot by slot.
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if (valid[j]) {
                if constexpr (PACKED) {
                    s_klo[fin[j]] = (key[j].lo << 1) | (head[j] ? 1ull : 0ull);
                } else if constexpr (WP) {
                    s_kw[fin[j]] = w128_pack(key[j], head[j] ? 1u : 0u);
                } else {
                    s_klo[fin[j]] = key[j].lo;
                    s_idx[fin[j]] = head[j] ? 1 : 0;
                }
            }
        }
        __syncthreads();
        u32 wh = 0;
        // slot reads for all rounds at once (clamped index, no per-slot branch around the LDS read)
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 p = w * EPW + j * 64 + lane;
            const bool live = (u32)j < R && p < c;
            const u32 pc = live ? p : 0u;
            if constexpr (PACKED) {
                const u64 v = s_klo[pc];
                head[j] = live && (v & 1ull);
                key[j].lo = v >> 1;
            } else if constexpr (WP) {
                const W128 v = s_kw[pc];
                head[j] = live && (v.lo & 1ull);
                key[j] = w128_sfx(v);
            } else {
                key[j].lo = s_klo[pc];
                head[j] = live && s_idx[pc] != 0;
            }
        }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) wh += (u32)__builtin_popcountll(__ballot(head[j]));
        if (lane == 0) s_wtot[w] = wh;
        __syncthreads();
        u32 run = 0;
        for (u32 ww = 0; ww < w; ++ww) run += s_wtot[ww];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u64 bal = __ballot(head[j]);
            if (head[j]) store_sfx<WS, HiT>(lo, hi, s0 + run + mbcnt(bal), key[j]);
            run += (u32)__builtin_popcountll(bal);
        }
    }
    if (tid == 0) {
        out_count[r] = d;
        out_kind[r] = trie ? KIND_TRIE : KIND_VEC;
    }
}

// ---- KRN-3, runs that end up SORTED (more than 1024 words, a bucket that is a Trie already, `self |= other`), packed elements (round 6) ----
// k_bucket_msd ranks every element inside its sub-bucket by reading the whole sub-bucket: s reads per element, and under the top
// suffix bits — the only kind of sub-bucket a sorted result allows — an element shares its sub-bucket with 6.5 others on average and
// up to 45 (necklace clusters, DESIGN_HISTORY.md §3.7): 60 random 8-byte LDS reads per lane of eight slots, issued slot by slot in loops whose
// trip count is the LARGEST sub-bucket among the wave's 64 lanes (162 LDS instructions per wave, the LDS array 74 % busy, 42 % of it
// bank conflicts: profiles/r05_sq_counters.md). Here the ranking is turned around: after the same counting sort into sub-bucket order,
// a lane OWNS eight consecutive positions of that order — entries of the same one or two sub-buckets — and walks ONCE over the span from
// the start of its first entry's sub-bucket to the end of its last entry's, comparing every entry it reads with all eight of its own
// (registers). Everything in front of the span is smaller than the lane's entries and everything behind it greater (the sub-buckets are
// ascending), so final rank = span start + entries of the span that compare less: one LDS read serves eight comparisons, 13 reads per
// lane on average (29 for the wave's longest span) instead of 60, and a lane's reads step through consecutive slots. Repeats are settled
// AFTER the sort — an element is the head of its value iff the slot in front of it holds another suffix — so the inner loop is one
// compare and one add per pair. The kernel is bound by those comparisons (VALU), no longer by the LDS.
// Outcome as k_bucket_msd's: the ascending distinct list, KIND_TRIE. A run that turns out to stay a Vec (a run of more than 1024 words
// with at most 1024 distinct ones, not a Trie yet — repeats) has written nothing yet and is handed on like a crowded one (claim table).
// MERGE: `self |= other` (mg.cs set): the run is [self's suffixes][other's], read where the two indexes store them (mg.s_lo) or from the
// gathered run; both parts hold distinct words, so a repeat is other's copy of a word self holds and sits right behind it in the sorted
// order. The rules of /root/reference/src/trievec/set_ops.rs:43-71 are then ordered selections of the sorted slots, as in k_bucket_msd's
// merge epilogue: what a slot says about its element is whether it came from self (index < cs) and whether it is the first of its value.
#ifndef CBLX_SORTED_WAVES
#define CBLX_SORTED_WAVES 7
#endif
#ifndef CBLX_SORTED_BALANCE
#define CBLX_SORTED_BALANCE 0  // the chunks of a workgroup dealt to its lanes in order of their span (measured: see DESIGN_HISTORY.md §3.13)
#endif
#ifndef CBLX_SORTED_PERBODY
#define CBLX_SORTED_PERBODY 0  // every phase instantiated per slots-per-lane, not the walk alone (measured: 3.471 against 3.449 ms, DESIGN_HISTORY.md §3.13)
#endif
#ifndef CBLX_SORTED_PROBE
#define CBLX_SORTED_PROBE 0  // > 0: timing probes that leave phases out (wrong results); never in the product build
#endif
#if CBLX_SORTED_PROBE && !defined(CBLX_TIMING_PROBES)
#error "CBLX_SORTED_PROBE leaves phases out and produces wrong results: timing builds only (-DCBLX_TIMING_PROBES)"
#endif
// The element of the walk: suffix << 12 | stream index. Narrow (SUFFIX_BITS + 12 <= 64): one u64 in an array padded by one slot per
// eight — the 64 lanes' slots of one step are 8 slots apart, 9 with the padding: every bank pair is hit four times, the best a
// 64 x 8-byte access can do. Wide (SUFFIX_BITS + 12 <= 128): one 16-byte slot (ds_read_b128), slot index XORed with its block-of-eight
// number instead (eight lanes cover the 32 banks; no LDS spent on padding: the wide classes are bound by LDS residency).
template <bool WS> struct WalkEl;
template <> struct WalkEl<false> {
    typedef u64 T;
    static constexpr u32 slots(u32 cap) { return cap + cap / 8 + 6; }
    static __device__ __forceinline__ u32 phys(u32 p) { return p + (p >> 3); }
    static __device__ __forceinline__ T pack(const Sfx<false>& k, u32 e) { return (k.lo << PK_BITS) | e; }
    static __device__ __forceinline__ T ones() { return ~0ull; }
    static __device__ __forceinline__ bool less(const T& a, const T& b) { return a < b; }
    static __device__ __forceinline__ bool same_sfx(const T& a, const T& b) { return (a >> PK_BITS) == (b >> PK_BITS); }
    static __device__ __forceinline__ u32 idx(const T& a) { return (u32)a & ((1u << PK_BITS) - 1u); }
    static __device__ __forceinline__ Sfx<false> sfx(const T& a) { Sfx<false> s; s.lo = a >> PK_BITS; return s; }
    static __device__ __forceinline__ u32 sub(const T& a, u32 sh, u32 nbits) { return (u32)(a >> (PK_BITS + sh)) & ((1u << nbits) - 1u); }
};
template <> struct WalkEl<true> {
    typedef W128 T;
    static constexpr u32 slots(u32 cap) { return cap + 8; }
    static __device__ __forceinline__ u32 phys(u32 p) { return p ^ ((p >> 3) & 7u); }
    static __device__ __forceinline__ T pack(const Sfx<true>& k, u32 e) { return w128_pack(k, e); }
    static __device__ __forceinline__ T ones() { return W128{~0ull, ~0ull}; }
    static __device__ __forceinline__ bool less(const T& a, const T& b) { return w128_less(a, b); }
    static __device__ __forceinline__ bool same_sfx(const T& a, const T& b) { return w128_same_sfx(a, b); }
    static __device__ __forceinline__ u32 idx(const T& a) { return (u32)a.lo & ((1u << PK_BITS) - 1u); }
    static __device__ __forceinline__ Sfx<true> sfx(const T& a) { return w128_sfx(a); }
    static __device__ __forceinline__ u32 sub(const T& a, u32 sh, u32 nbits) {
        const u128 v = ((u128)a.hi << 64) | a.lo;
        return (u32)(v >> (PK_BITS + sh)) & ((1u << nbits) - 1u);
    }
};
template <int CAP, bool WS> constexpr int sorted_waves() { return WS ? (CAP <= 128 ? CBLX_SORTED_WAVES : 4) : (CAP >= 4096 ? 6 : CBLX_SORTED_WAVES); }
template <int THREADS, int CAP, bool WS, typename HiT, bool MERGE = false>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(sorted_waves<CAP, WS>(), 8))) void k_bucket_sorted(
    const BDesc* __restrict__ list, const u32* __restrict__ list_n, u64* __restrict__ lo, HiT* __restrict__ hi, u32 SB, u32* __restrict__ out_count, u8* __restrict__ out_kind,
    BDesc* __restrict__ retry, u32* __restrict__ retry_n, u8* __restrict__ bail_flag = nullptr, u32* __restrict__ bail_any = nullptr, MergeArgs mg = MergeArgs{}) {
    static_assert(CAP <= (1 << PK_BITS), "stream index must fit PK_BITS");
    static_assert(!WS || sizeof(HiT) == 8, "wide suffixes keep their high part in 64-bit words");
    typedef WalkEl<WS> EL;
    typedef typename EL::T E;
    constexpr int ITEMS = CAP / THREADS, NW = THREADS / 64;
    static_assert(!MERGE || 64 * ITEMS < 1024, "the merge epilogue counts a wave's selections in 10-bit fields");
    __shared__ E s_k[EL::slots(CAP)];       // slot p lives at EL::phys(p); + the all-ones slots behind the run
    __shared__ u32 s_off32[CAP / 2 + 2];    // sub-bucket counts, then exclusive offsets: 16-bit entries, counted with 32-bit atomics on the containing dword
    u16* s_off = reinterpret_cast<u16*>(s_off32);
    __shared__ u32 s_scan[NW + 1];
    __shared__ u32 s_wtot[NW + 1];
    __shared__ u32 s_max;
    __shared__ u32 s_sel[MERGE ? NW : 1];

    if (blockIdx.x >= *list_n) return;
    const BDesc dsc = list[blockIdx.x];
    const u32 r = dsc.r;
    const u64 s0 = dsc.start;
    const u32 c = dsc.c & BDESC_LEN_MASK;
    const u32 skip = (dsc.c & BDESC_SKIP_MASK) >> BDESC_SKIP_SHIFT;  // leading suffix bits shared by the whole run (sub-range of a long run)
    const bool res_trie = (dsc.c & BDESC_TRIE) != 0;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    if (c == 0) {  // an empty sub-range of a long run
        if (tid == 0) { out_count[r] = 0; out_kind[r] = KIND_TRIE; }
        return;
    }
    u32 nbits = 32 - __builtin_clz(c - 1 > 0 ? c - 1 : 1);  // ceil(log2 c)
    if (nbits > SB - skip) nbits = SB - skip;
    const u32 NB = 1u << nbits, sub_sh = SB - skip - nbits;
    constexpr u32 crowd = WS ? MSD_LIMIT_WIDE : MSD_LIMIT;
    const bool merging = MERGE && mg.cs != nullptr;
    u32 cs_m = 0;  // merge: self's length (index < cs_m = the element came from self)
    if constexpr (MERGE) if (merging) cs_m = mg.cs[r];
    auto give_up = [&]() {
        if (tid == 0) {
            if (bail_flag) { bail_flag[blockIdx.x] = 1; *bail_any = 1u; }
            else retry[atomicAdd(retry_n, 1u)] = dsc;
        }
    };
    for (u32 i = tid; i < NB / 2 + 1; i += THREADS) s_off32[i] = 0;
    if (tid == 0) s_max = 0;
    __syncthreads();
    // (CBLX_SORTED_PERBODY=1 instantiates everything below per `per` = slots a lane owns, uniform over the workgroup — a run of 1 300 words on 256
    // threads then issues six slots per lane in every phase, not eight; the masked slots turned out to cost nothing outside the walk)
    const u32 per = (c + THREADS - 1) / THREADS;  // <= ITEMS
    auto body = [&](auto per_tag) {
    constexpr int PI = decltype(per_tag)::value;
    {
        Sfx<WS> key[PI];
        u32 sub[PI], arr[PI];
        // all loads first, unconditionally (slots past the run re-read its first word): eight independent global loads in flight per lane
        if constexpr (MERGE) {
            // element e is word e of `pa` (e < split) or of `pb` (other's pointer moved back by the split: no subtraction per slot)
            const u64* pa = lo + s0;
            const u64* pb = pa;
            const u64* ha = WS ? reinterpret_cast<const u64*>(hi) + s0 : nullptr;
            const u64* hb = ha;
            u32 split = 0;
            if (merging && mg.s_lo) {
                split = cs_m;
                const u64 a_self = mg.sstart[r], a_oth = mg.ostart[r];
                pa = mg.s_lo + a_self;
                pb = reinterpret_cast<const u64*>(reinterpret_cast<uintptr_t>(mg.o_lo + a_oth) - 8ull * split);
                if constexpr (WS) {
                    ha = mg.s_hi + a_self;
                    hb = reinterpret_cast<const u64*>(reinterpret_cast<uintptr_t>(mg.o_hi + a_oth) - 8ull * split);
                }
            }
#pragma unroll
            for (int j = 0; j < PI; ++j) {
                const u32 e = j * THREADS + tid, ee = e < c ? e : 0u;
                const bool in_a = ee < split;
                key[j] = load_sfx<WS, u64>(in_a ? pa : pb, in_a ? ha : hb, ee, SB);
            }
        } else {
            const u64* __restrict__ run_lo = lo + s0;
            const HiT* __restrict__ run_hi = WS ? hi + s0 : nullptr;
#pragma unroll
            for (int j = 0; j < PI; ++j) {
                const u32 e = j * THREADS + tid;
                key[j] = load_sfx<WS, HiT>(run_lo, run_hi, e < c ? e : 0u, SB);
            }
        }
#pragma unroll
        for (int j = 0; j < PI; ++j) {
            const u32 e = j * THREADS + tid;
            sub[j] = sfx_bits_below<WS>(key[j], SB, skip, nbits);
            arr[j] = 0;
            if (e < c) arr[j] = atomicAdd(&s_off32[sub[j] >> 1], 1u << ((sub[j] & 1u) * 16u));  // raw dword; the field is cut out below
        }
        bool crowded = false;  // an arrival number of `crowd` = a sub-bucket of more than `crowd` entries
#pragma unroll
        for (int j = 0; j < PI; ++j) {
            arr[j] = (arr[j] >> ((sub[j] & 1u) * 16u)) & 0xFFFFu;
            crowded |= arr[j] >= crowd;
        }
        if (crowded) s_max = crowd + 1u;  // benign race: every writer stores the same value
        __syncthreads();
        if (s_max > crowd) { give_up(); return; }  // repeats (or a cluster beyond the limit): claim table (build) / radix kernel (merge, sub-ranges)
#if CBLX_SORTED_PROBE == 1  // timing probe only: loads and counting atomics
        if (tid == 0) { out_count[r] = c; out_kind[r] = KIND_TRIE; }
        if (s_off32[tid] != 0x12345678u) return;
#endif
        {   // exclusive scan of the NB counts; each thread owns `per` consecutive entries
            const u32 per_nb = (NB + THREADS - 1) / THREADS;  // <= ITEMS
            const u32 b0 = tid * per_nb;
            u32 sum = 0, cnt[ITEMS];
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {
                cnt[k] = ((u32)k < per_nb && b0 + k < NB) ? s_off[b0 + k] : 0u;
                sum += cnt[k];
            }
            u32 ex = block_exclusive_scan<THREADS, u32>(sum, s_scan, nullptr);
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {
                if ((u32)k < per_nb && b0 + k < NB) { s_off[b0 + k] = (u16)ex; ex += cnt[k]; }
            }
        }
        __syncthreads();
        if (tid == 0) s_off[NB] = (u16)c;
        if (tid < 4) s_k[EL::phys(c + tid)] = EL::ones();  // what a walk may read behind the run compares greater than every element
        u32 sbase[PI];
#pragma unroll
        for (int j = 0; j < PI; ++j) sbase[j] = s_off[sub[j]];
#pragma unroll
        for (int j = 0; j < PI; ++j) {
            const u32 e = j * THREADS + tid;
            if (e < c) s_k[EL::phys(sbase[j] + arr[j])] = EL::pack(key[j], e);
        }
    }
    __syncthreads();
#if CBLX_SORTED_PROBE == 2  // timing probe only: everything up to the scatter
    if (tid == 0) { out_count[r] = c; out_kind[r] = KIND_TRIE; }
    if (s_off32[tid] != 0x1234567u) return;
#endif
    // -- the walk: lane t owns slots [t per, t per + per) of the sub-bucket order
    u32 chunk = tid;
#if CBLX_SORTED_BALANCE
    if constexpr (NW > 1 && THREADS <= 256) {
        // The wave's longest span is what its walk costs (2.2 x the mean): the chunks are dealt to the lanes in order of their span, so that a wave's
        // lanes walk spans of about one length (a counting sort of the THREADS chunks on min(span, 127))
        __shared__ u32 s_bins32[64];
        __shared__ u8 s_perm[THREADS];
        u16* s_bins = reinterpret_cast<u16*>(s_bins32);
        if (tid < 64) s_bins32[tid] = 0;
        const u32 q0 = tid * per;
        u32 span = 0;
        if (q0 < c) {
            const u32 q1 = (q0 + per < c ? q0 + per : c) - 1u;
            const E f = s_k[EL::phys(q0)], l = s_k[EL::phys(q1)];
            span = s_off[EL::sub(l, sub_sh, nbits) + 1u] - (s_off[EL::sub(f, sub_sh, nbits)] & ~1u);
        }
        const u32 bin = span < 127u ? span : 127u;
        __syncthreads();
        const u32 arr = (atomicAdd(&s_bins32[bin >> 1], 1u << ((bin & 1u) * 16u)) >> ((bin & 1u) * 16u)) & 0xFFFFu;
        __syncthreads();
        if (tid < 64) {  // exclusive scan of the 128 counts, two per lane
            const u32 a = s_bins[2 * tid], b = s_bins[2 * tid + 1];
            const u32 inc = wave_inclusive_scan(a + b);
            s_bins[2 * tid] = (u16)(inc - a - b);
            s_bins[2 * tid + 1] = (u16)(inc - b);
        }
        __syncthreads();
        s_perm[s_bins[bin] + arr] = (u8)tid;
        __syncthreads();
        chunk = s_perm[tid];
    }
#endif
    const u32 p0 = chunk * per;
    const u32 n_own = p0 < c ? (c - p0 < per ? c - p0 : per) : 0u;
    E me[PI];
    u32 fin[PI];
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const u32 p = p0 + i;
        me[i] = s_k[EL::phys(p < c ? p : c)];  // (slots past the lane's share read the all-ones slot or a neighbour's entry: never written back)
        fin[i] = 0;
    }
    u32 A = 0, B = 0;
    if (n_own) {
        E last = me[0];
#pragma unroll
        for (int i = 1; i < PI; ++i) if ((u32)i < n_own) last = me[i];
        // (rounded down to an even slot: the entry in front of the sub-bucket is smaller than every entry of mine, it counts like the rest in front)
        A = s_off[EL::sub(me[0], sub_sh, nbits)] & ~1u;
        B = s_off[EL::sub(last, sub_sh, nbits) + 1u];
    }
#ifdef CBLX_SORTED_STATS  // dev: span statistics into retry_n[0..7] as u64 (tools/dev_msd_bench.cpp)
    {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(retry_n);
        const u32 span = B - A;
        u32 mx = span;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const u32 t = __shfl_xor(mx, o, 64); mx = t > mx ? t : mx; }
        const u32 sm = wave_reduce_sum(span), na = wave_reduce_sum(n_own ? 1u : 0u);
        if (lane == 0) { atomicAdd(&st[0], (unsigned long long)sm); atomicAdd(&st[1], (unsigned long long)mx); atomicAdd(&st[2], (unsigned long long)na); atomicAdd(&st[3], 1ull); }
    }
#endif
    // (an entry read behind B belongs to a later sub-bucket, or is all ones: greater than every entry of mine.) The comparisons are
    // what this kernel is made of (two VALU instructions per pair, 28.8 steps of the wave's longest span at cfg 2): the loop is
    // instantiated per `per` — uniform over the workgroup — so that a run of 1300 words (per = 6) does not pay for eight slots
    auto walk = [&](auto walk_tag) {
        constexpr int PER = decltype(walk_tag)::value;
        for (u32 q = A; q < B; q += 2) {
            E o0, o1;
            if constexpr (WS) {
                o0 = s_k[EL::phys(q)];
                o1 = s_k[EL::phys(q + 1)];
            } else {  // (A is even: the two slots of a step are neighbours in the padded array too, one address and an immediate offset)
                const E* at = &s_k[EL::phys(q)];
                o0 = at[0];
                o1 = at[1];
            }
#pragma unroll
            for (int i = 0; i < PER; ++i) fin[i] += (EL::less(o0, me[i]) ? 1u : 0u) + (EL::less(o1, me[i]) ? 1u : 0u);
        }
    };
#if CBLX_SORTED_PERBODY
    walk(std::integral_constant<int, PI>());
#else
    if constexpr (PI == 8) {
        switch (per) {
            case 8: walk(std::integral_constant<int, 8>()); break;
            case 7: walk(std::integral_constant<int, 7>()); break;
            case 6: walk(std::integral_constant<int, 6>()); break;
            case 5: walk(std::integral_constant<int, 5>()); break;
            default: walk(std::integral_constant<int, 4>()); break;
        }
    } else {
        walk(std::integral_constant<int, PI>());
    }
#endif
#if CBLX_SORTED_PROBE == 3  // timing probe only: everything up to the walk
    if (tid == 0) { out_count[r] = c; out_kind[r] = KIND_TRIE; }
    { u32 t = 0;
#pragma unroll
      for (int i = 0; i < PI; ++i) t += fin[i];
      if (t != 0x12345678u) return; }
#endif
    __syncthreads();  // every read of the sub-bucket order is done
#pragma unroll
    for (int i = 0; i < PI; ++i)
        if ((u32)i < n_own) s_k[EL::phys(A + fin[i])] = me[i];
    __syncthreads();
    const u32 EPW = 64 * per;  // wave-contiguous slices of the sorted slots: ballots then compact in order
    Sfx<WS> val[PI];
    if constexpr (MERGE) if (merging) {
        // three ordered selections of the sorted slots, counted and written together (k_bucket_msd's merge epilogue):
        //   O = other's elements (back to other's arena, sorted)            A = Trie |= x: the heads (sorted union); Vec |= x: self's elements
        //   B = Vec |= x only: other's heads = other \ self, behind self's   (every self element is a head)
        const bool o_vec = mg.okind[r] == KIND_VEC;  // the reference's iter_sorted leaves other's Vec sorted
        u32 fl[PI];
        u32 nO = 0;
#pragma unroll
        for (int j = 0; j < PI; ++j) {
            const u32 p = w * EPW + j * 64 + lane;
            const bool live = (u32)j < per && p < c;
            const u32 pc = live ? p : 0u;
            const E v = s_k[EL::phys(pc)], pv = s_k[EL::phys(pc ? pc - 1u : 0u)];
            const bool hd = pc == 0 || !EL::same_sfx(v, pv), self_el = EL::idx(v) < cs_m;
            val[j] = EL::sfx(v);
            const bool inO = live && !self_el && o_vec, inA = live && (res_trie ? hd : self_el), inB = live && !res_trie && !self_el && hd;
            fl[j] = (inO ? 1u : 0u) | (inA ? (1u << 10) : 0u) | (inB ? (1u << 20) : 0u);
            nO += fl[j];
        }
        nO = wave_reduce_sum(nO);  // three 10-bit fields, <= 512 each
        if (lane == 0) s_sel[w] = nO;
        __syncthreads();
        u32 runO = 0, runA = 0, runB = 0, totA = 0, totB = 0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            const u32 t = s_sel[ww], tO = t & 1023u, tA = (t >> 10) & 1023u, tB = t >> 20;
            if ((u32)ww < w) { runO += tO; runA += tA; runB += tB; }
            totA += tA;
            totB += tB;
        }
        u64* __restrict__ self_lo = lo + s0;
        u64* __restrict__ self_hi = WS ? reinterpret_cast<u64*>(hi) + s0 : nullptr;
        const u64 obase = mg.ostart[r];
        runB += cs_m;  // B follows self's cs elements
#pragma unroll
        for (int j = 0; j < PI; ++j) {
            const u64 balO = __ballot((fl[j] & 1u) != 0), balA = __ballot((fl[j] & (1u << 10)) != 0), balB = __ballot((fl[j] & (1u << 20)) != 0);
            if (fl[j] & 1u) {
                const u32 q = runO + mbcnt(balO);
                mg.o_lo[obase + q] = val[j].lo;
                if constexpr (WS) mg.o_hi[obase + q] = val[j].hi;
            }
            if (fl[j] & ((1u << 10) | (1u << 20))) {  // A and B are disjoint
                const u32 q = (fl[j] & (1u << 10)) ? runA + mbcnt(balA) : runB + mbcnt(balB);
                self_lo[q] = val[j].lo;
                if constexpr (WS) self_hi[q] = val[j].hi;
            }
            runO += (u32)__builtin_popcountll(balO);
            runA += (u32)__builtin_popcountll(balA);
            runB += (u32)__builtin_popcountll(balB);
        }
        if (tid == 0) {
            out_count[r] = res_trie ? totA : cs_m + totB;
            out_kind[r] = res_trie ? KIND_TRIE : KIND_VEC;
        }
        return;
    }
    // -- heads (the first slot of every suffix value), counted, then compacted slot by slot
    bool head[PI];
    u32 wh = 0;
#pragma unroll
    for (int j = 0; j < PI; ++j) {
        const u32 p = w * EPW + j * 64 + lane;
        const bool live = (u32)j < per && p < c;
        const u32 pc = live ? p : 0u;
        const E v = s_k[EL::phys(pc)], pv = s_k[EL::phys(pc ? pc - 1u : 0u)];
        head[j] = live && (pc == 0 || !EL::same_sfx(v, pv));
        val[j] = EL::sfx(v);
        wh += (u32)__builtin_popcountll(__ballot(head[j]));
    }
    if (lane == 0) s_wtot[w] = wh;
    __syncthreads();
    u32 run = 0, d = 0;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
        const u32 t = s_wtot[ww];
        if ((u32)ww < w) run += t;
        d += t;
    }
    if (d <= VEC_THRESHOLD && !res_trie) { give_up(); return; }  // stays a Vec (repeats): stream order needed — nothing was written yet
    u64* __restrict__ out_lo = lo + s0;
    HiT* __restrict__ out_hi = WS ? hi + s0 : nullptr;
    if (d == c) {  // no repeat in the run (the usual case): every slot is a head and keeps its place
#pragma unroll
        for (int j = 0; j < PI; ++j) {
            const u32 p = w * EPW + j * 64 + lane;
            if ((u32)j < per && p < c) store_sfx<WS, HiT>(out_lo, out_hi, p, val[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < PI; ++j) {
            const u64 bal = __ballot(head[j]);
            if (head[j]) store_sfx<WS, HiT>(out_lo, out_hi, run + mbcnt(bal), val[j]);
            run += (u32)__builtin_popcountll(bal);
        }
    }
    if (tid == 0) {
        out_count[r] = d;
        out_kind[r] = KIND_TRIE;
    }
    };
#if CBLX_SORTED_PERBODY
    if constexpr (ITEMS == 8) {
        switch (per) {
            case 8: body(std::integral_constant<int, 8>()); break;
            case 7: body(std::integral_constant<int, 7>()); break;
            case 6: body(std::integral_constant<int, 6>()); break;
            case 5: body(std::integral_constant<int, 5>()); break;
            default: body(std::integral_constant<int, 4>()); break;  // (per <= 4: a short run of a class above its length, e.g. a sub-range of a long run)
        }
    } else {
        body(std::integral_constant<int, ITEMS>());
    }
#else
    body(std::integral_constant<int, ITEMS>());
#endif
}

// ---- KRN-3 for runs full of repeats (one batch at high coverage: every k-mer arrives dozens of times). The counting sort
// above gives up on them — a sub-bucket holds ALL copies of its values and every element would read them all — and the LDS
// radix sort it used to hand them to costs 5x the time per word. Here equal suffixes meet in ONE slot of an open-addressing
// table of element indices (twice as many slots as elements: short probe sequences whatever the data), atomicMin leaves
// the smallest index = the first occurrence (TrieVec::insert in Vec mode keeps exactly that one,
// /root/reference/src/trievec/mod.rs:72-99), and the distinct elements are compacted in stream order. A run that needs
// the sorted layout (more than 1024 distinct suffixes, or a bucket that is a Trie already) is listed in `retry` with its
// new length: distinct words now, which the counting sort takes in its stride.
template <int THREADS, int CAP, bool WS, typename HiT>
__global__ __launch_bounds__(THREADS) void k_bucket_claim(const BDesc* __restrict__ list, const u32* __restrict__ list_n, u64* __restrict__ lo, HiT* __restrict__ hi, u32 SB,
                                                          u32* __restrict__ out_count, u8* __restrict__ out_kind, BDesc* __restrict__ retry, u32* __restrict__ retry_n,
                                                          const u8* __restrict__ only /* null: every list entry; else only the marked ones */) {
    constexpr int ITEMS = CAP / THREADS, NW = THREADS / 64;
    constexpr u32 TS = 2 * CAP;
    __shared__ u64 s_klo[CAP];
    __shared__ u64 s_khi[WS ? CAP : 1];
    __shared__ u32 s_tab[TS];
    __shared__ u32 s_wtot[NW + 1];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const u32 nlist = *list_n;
    // `only` set (the marked entries of a class list, possibly very few of them): one workgroup looks at 64 entries — every
    // wave loads the same 64 marks — and takes the marked ones one after the other. (One workgroup per entry that exits when
    // its entry is not marked costs 0.2 ms of launches per half a million entries.)
    const u32 first = only ? blockIdx.x * 64u : blockIdx.x;
    u64 todo = 1;
    if (only) todo = __ballot(first + lane < nlist && only[first + lane] != 0);
    else if (first >= nlist) todo = 0;
    while (todo) {
    const u32 li = first + (u32)__builtin_ctzll(todo);
    todo &= todo - 1;
    const BDesc dsc = list[li];
    const u32 r = dsc.r;
    const u64 s0 = dsc.start;
    const u32 c = dsc.c & BDESC_LEN_MASK;
    const bool res_trie = (dsc.c & BDESC_TRIE) != 0;
    const u32 R = (c + THREADS - 1) / THREADS;
    const u32 EPW = 64 * R;  // wave-contiguous slices: ballots then compact in stream order
    u32 tbits = 33 - __builtin_clz(c > 1 ? c - 1 : 1);  // 2^tbits >= 2 c
    if (tbits < 6) tbits = 6;
    const u32 mask = (1u << tbits) - 1u;  // <= TS - 1 because c <= CAP
    for (u32 i = tid; i <= mask; i += THREADS) s_tab[i] = EMPTY32;
    Sfx<WS> key[ITEMS];
    bool valid[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {  // all loads first (slots past the run re-read its first element)
        const u32 e = w * EPW + j * 64 + lane;
        valid[j] = (u32)j < R && e < c;
        key[j] = load_sfx<WS, HiT>(lo, hi, s0 + (valid[j] ? e : 0u), SB);
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 e = w * EPW + j * 64 + lane;
        if (valid[j]) {
            s_klo[e] = key[j].lo;
            if constexpr (WS) s_khi[e] = key[j].hi;
        }
    }
    __syncthreads();
    u32 slot[ITEMS], old[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {  // the first probe of every item at once: independent atomics in flight instead of one round trip after the other
        const u32 e = w * EPW + j * 64 + lane;
        slot[j] = sfx_hash_bits<WS>(key[j], tbits);
        old[j] = valid[j] ? atomicCAS(&s_tab[slot[j]], EMPTY32, e) : EMPTY32;
    }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 e = w * EPW + j * 64 + lane;
        if (old[j] != EMPTY32) {  // the slot was taken (EMPTY32: the value's first arrival, the slot is its own from now on)
            u32 h = slot[j], o_e = old[j];
            for (;;) {
                Sfx<WS> o;
                o.lo = s_klo[o_e];
                if constexpr (WS) o.hi = s_khi[o_e];
                if (o == key[j]) { atomicMin(&s_tab[h], e); break; }  // whoever holds the slot has this suffix: keep the earlier index
                h = (h + 1u) & mask;
                o_e = atomicCAS(&s_tab[h], EMPTY32, e);
                if (o_e == EMPTY32) break;
            }
            slot[j] = h;
        }
    }
    __syncthreads();
    bool head[ITEMS];
    u32 wave_heads = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const u32 e = w * EPW + j * 64 + lane;
        head[j] = valid[j] && s_tab[slot[j]] == e;
        wave_heads += (u32)__builtin_popcountll(__ballot(head[j]));
    }
    if (lane == 0) s_wtot[w] = wave_heads;
    __syncthreads();
    if (tid == 0) {
        u32 run = 0;
        for (int ww = 0; ww < NW; ++ww) { const u32 t = s_wtot[ww]; s_wtot[ww] = run; run += t; }
        s_wtot[NW] = run;
    }
    __syncthreads();
    const u32 d = s_wtot[NW];
    u32 run = s_wtot[w];
    if (d != c) {  // the first occurrences, in stream order, over the front of the run (every load is done)
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u64 bal = __ballot(head[j]);
            if (head[j]) store_sfx<WS, HiT>(lo, hi, s0 + run + mbcnt(bal), key[j]);
            run += (u32)__builtin_popcountll(bal);
        }
    }
    if (tid == 0) {
        if (d > VEC_THRESHOLD || res_trie) {  // sorted layout needed: the run, now d DISTINCT words, goes on to the counting sort
            retry[atomicAdd(retry_n, 1u)] = BDesc{s0, d | (res_trie ? BDESC_TRIE : 0u), r};
        } else {
            out_count[r] = d;
            out_kind[r] = KIND_VEC;
        }
    }
    __syncthreads();  // the table and the staged keys are reused by the next entry
    }
}

// ---- runs LONGER than one workgroup's LDS that are full of repeats (one batch at very high coverage: a bucket of ten
// thousand arrivals holds a few dozen distinct suffixes). Sorting such a run — split by the top suffix bits, sub-ranges
// through k_bucket_msd, which finds every sub-bucket crowded, then the global-memory radix kernel — cost 25..38 ms per
// 960 M words at 300x..1000x coverage. This pre-pass shrinks the run to its first occurrences instead, in place and in
// stream order, before anything is sorted: an LDS table of SUFFIXES (64-bit compare-and-swap claims a slot for a value)
// with the smallest index seen per slot (atomicMin); the run streams through in tiles twice — claim, then emit the heads.
// A run with more distinct suffixes than half the table (or whose first 512 words are mostly distinct) is passed on untouched
// to the sort it would have had. Narrow suffixes below 64 bits only (the all-ones pattern marks a free slot).
static const u32 BCL_THREADS = 256, BCL_ITEMS = 8, BCL_TILE = BCL_THREADS * BCL_ITEMS, BCL_SLOTS = 4096, BCL_PROBE = 512, BCL_PROBE_MAX = 358;
template <typename HiT>
__global__ __launch_bounds__(BCL_THREADS) void k_big_claim(const BDesc* __restrict__ list, const u32* __restrict__ list_n, u64* __restrict__ lo, u32 SB,
                                                           u32* __restrict__ out_count, u8* __restrict__ out_kind, BDesc* __restrict__ next, u32* __restrict__ next_n,
                                                           BDesc* __restrict__ sorted, u32* __restrict__ sorted_n) {
    constexpr u32 NW = BCL_THREADS / 64, MASK = BCL_SLOTS - 1;
    constexpr u64 FREE = ~0ull;
    __shared__ u64 s_key[BCL_SLOTS];
    __shared__ u32 s_idx[BCL_SLOTS];
    __shared__ u32 s_w[NW + 1];
    __shared__ u32 s_new;
    if (blockIdx.x >= *list_n) return;
    const BDesc dsc = list[blockIdx.x];
    const u32 r = dsc.r;
    const u64 s0 = dsc.start;
    const u32 c = dsc.c & ~BDESC_TRIE;
    const bool res_trie = (dsc.c & BDESC_TRIE) != 0;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const u64 smask = (1ull << SB) - 1ull;
    for (u32 i = tid; i < BCL_SLOTS; i += BCL_THREADS) { s_key[i] = FREE; s_idx[i] = EMPTY32; }
    if (tid == 0) s_new = 0;
    __syncthreads();
    auto slot_hash = [&](u64 key) -> u32 {
        u64 v = key ^ (key >> 32);
        v *= 0x9E3779B97F4A7C15ull;
        return (u32)(v >> 52);  // 12 bits = BCL_SLOTS
    };
    static_assert(BCL_SLOTS == 4096, "slot_hash takes 12 bits");
    auto claim = [&](u64 key, u32 e) -> u32 {  // 1: the value is new
        u32 h = slot_hash(key), fresh = 0;
        for (u32 probes = 0; probes < BCL_SLOTS; ++probes) {
            const u64 old = atomicCAS((unsigned long long*)&s_key[h], (unsigned long long)FREE, (unsigned long long)key);
            if (old == FREE) { fresh = 1; break; }
            if (old == key) break;
            h = (h + 1u) & MASK;
        }
        atomicMin(&s_idx[h], e);
        return fresh;
    };
    {   // a look at the first BCL_PROBE words decides whether this is a run of repeats at all (a long run of distinct words
        // — the receiving side of an 8-GPU build has 70 000 of them — leaves after 512 claims instead of 2048 .. 4096)
        u32 fresh = 0;
        for (u32 e = tid; e < BCL_PROBE && e < c; e += BCL_THREADS) fresh += claim(lo[s0 + e] & smask, e);
        if (fresh) atomicAdd(&s_new, fresh);
        __syncthreads();
        const u32 distinct = s_new;
        __syncthreads();
        if (distinct > BCL_PROBE_MAX) {
            if (tid == 0) next[atomicAdd(next_n, 1u)] = dsc;
            return;
        }
    }
    // phase 1: every suffix claims its slot; the slot remembers the smallest index that asked for it
    for (u32 t0 = 0; t0 < c; t0 += BCL_TILE) {
        u64 key[BCL_ITEMS];
#pragma unroll
        for (int j = 0; j < (int)BCL_ITEMS; ++j) {
            const u32 e = t0 + j * BCL_THREADS + tid;
            key[j] = e < c ? (lo[s0 + e] & smask) : FREE;
        }
        u32 fresh = 0;
#pragma unroll
        for (int j = 0; j < (int)BCL_ITEMS; ++j) {
            const u32 e = t0 + j * BCL_THREADS + tid;
            if (e < c) fresh += claim(key[j], e);
        }
        if (fresh) atomicAdd(&s_new, fresh);
        __syncthreads();
        const u32 distinct = s_new;
        __syncthreads();
        if (distinct > BCL_SLOTS / 2) {  // too many distinct words for the table: the sort it would have had
            if (tid == 0) next[atomicAdd(next_n, 1u)] = dsc;
            return;
        }
    }
    // phase 2: the first occurrences in stream order, written over the front of the run (never past the tile being read)
    u32 base = 0;
    for (u32 t0 = 0; t0 < c; t0 += BCL_TILE) {
        u64 key[BCL_ITEMS];
        bool head[BCL_ITEMS];
        u32 wave_heads = 0;
#pragma unroll
        for (int j = 0; j < (int)BCL_ITEMS; ++j) {
            const u32 e = t0 + w * (64 * BCL_ITEMS) + j * 64 + lane;  // wave-contiguous slices: ballots compact in stream order
            key[j] = e < c ? (lo[s0 + e] & smask) : FREE;
        }
#pragma unroll
        for (int j = 0; j < (int)BCL_ITEMS; ++j) {
            const u32 e = t0 + w * (64 * BCL_ITEMS) + j * 64 + lane;
            head[j] = false;
            if (e < c) {
                u32 h = slot_hash(key[j]);
                while (s_key[h] != key[j]) h = (h + 1u) & MASK;  // it is there
                head[j] = s_idx[h] == e;
            }
            wave_heads += (u32)__builtin_popcountll(__ballot(head[j]));
        }
        if (lane == 0) s_w[w] = wave_heads;
        __syncthreads();  // every load of the tile is done: the stores below may land inside it
        u32 run = base;
        for (u32 ww = 0; ww < w; ++ww) run += s_w[ww];
        u32 tile_heads = 0;
        for (u32 ww = 0; ww < NW; ++ww) tile_heads += s_w[ww];
#pragma unroll
        for (int j = 0; j < (int)BCL_ITEMS; ++j) {
            const u64 bal = __ballot(head[j]);
            if (head[j]) lo[s0 + run + mbcnt(bal)] = key[j];
            run += (u32)__builtin_popcountll(bal);
        }
        base += tile_heads;
        __syncthreads();
    }
    if (tid == 0) {
        if (base <= VEC_THRESHOLD && !res_trie) {
            out_count[r] = base;
            out_kind[r] = KIND_VEC;
        } else {  // distinct now, at most BCL_SLOTS / 2 words: one workgroup sorts it
            sorted[atomicAdd(sorted_n, 1u)] = BDesc{s0, base | (res_trie ? BDESC_TRIE : 0u), r};
        }
    }
}

// ---- KRN-3 big: runs of 4097 .. BIG_MAX words -------------------------------------------------------------------------
// The long runs get ONE more pass of the partition kernels (k_radix_hist / k_radix_scatter with the runs as segments): a stable
// sort on the top suffix bits, out of the arena into a TWIN buffer at the same positions. A run then is a sequence of
// sub-ranges sharing their leading suffix bits; k_bucket_msd sorts + deduplicates every sub-range in place in the twin; and
// k_big_finish adds up a run's distinct counts (closing the gaps duplicates left). The finished runs are never copied back:
// the arena becomes whichever of the two buffers holds more words and the other side's buckets move over (pipeline.hpp,
// finish_twin) — on the receiving side of a many-GPU build, where every run is long, nothing is copied at all. The arena run
// stays untouched, so a run that must stay a Vec (<= 1024 distinct words and no resident Trie: a few words repeated
// thousands of times) or whose sort gave up still finds its original there (k_bucket_huge).
__host__ __device__ inline u32 big_digit_bits(u32 SB) { return SB < 8 ? SB : 8; }  // width of the pass's digit: the top suffix bits
// per list entry: its tiles and its number of sub-ranges (2^bits, about BIG_SUB words each; never finer than the pass's digit)
__global__ void k_big_plan(const BDesc* __restrict__ list, u32 n, u32 SB, u32* __restrict__ ntile, u32* __restrict__ nv) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 c = list[i].c & ~BDESC_TRIE, DB = big_digit_bits(SB), B = big_bits(c) < DB ? big_bits(c) : DB;
    ntile[i] = (c + RDX_TILE - 1) / RDX_TILE;
    nv[i] = 1u << B;
}
__global__ void k_big_tile_table(const BDesc* __restrict__ list, u32 nruns, const u32* __restrict__ tile_first /* nruns + 1 */, u64* __restrict__ t_start,
                                 u32* __restrict__ t_count, u32* __restrict__ t_seg, u64* __restrict__ run_start) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nruns) run_start[t] = list[t].start;
    if (t >= tile_first[nruns]) return;
    u32 lo = 0, hi = nruns;  // last run with tile_first[i] <= t (every run has a tile)
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (tile_first[mid] <= t) lo = mid; else hi = mid;
    }
    const u32 off = (t - tile_first[lo]) * RDX_TILE, c = list[lo].c & ~BDESC_TRIE;
    t_start[t] = list[lo].start + off;
    t_count[t] = c - off < (u32)RDX_TILE ? c - off : (u32)RDX_TILE;
    t_seg[t] = lo;
}
// One descriptor per sub-range (a "virtual bucket" for k_bucket_msd, always asked for the sorted distinct list): sub-range v of
// run i = the digits [v << (DB - B), (v + 1) << (DB - B)) of the pass, contiguous in the twin from rel[i][v << (DB - B)]
// (k_seg_adjust's grp_start, relative to the run), listed by length class for the sort kernel. A sub-range that outgrows the
// sort kernel marks itself BIG_SENT - 1 in v_count and is not listed: the whole run then takes the general kernel.
// (A workgroup takes BIG_VLIST_RUNS runs and reserves its list slots with ONE atomic per class: one workgroup per run meant 70 000 atomics on the same two
// counters on the receiving side of an 8-GPU job, 0.8 ms of nothing else.)
static const int BIG_VLIST_RUNS = 8;
__global__ __launch_bounds__(256) void k_big_vlist(const BDesc* __restrict__ list, u32 nruns, const u64* __restrict__ vb_, const u32* __restrict__ rel /* [nruns][256] */, u32 SB,
                                                   BDesc* __restrict__ vlist, u32* __restrict__ v_count, BDesc* __restrict__ cls_lists /* [2][vtot]: <= 1024 words, <= BIG_VCAP */,
                                                   u32* __restrict__ cls_n, u64 vtot) {
    __shared__ u32 s_n[2], s_base[2];
    const u32 v = threadIdx.x;
    if (v < 2) s_n[v] = 0;
    __syncthreads();
    int cls[BIG_VLIST_RUNS];
    BDesc d[BIG_VLIST_RUNS];
    u32 slot[BIG_VLIST_RUNS];
    const u32 DB = big_digit_bits(SB);
#pragma unroll
    for (int rr = 0; rr < BIG_VLIST_RUNS; ++rr) {
        cls[rr] = -1;
        d[rr] = BDesc{0, 0, 0};
        slot[rr] = 0;
        const u32 i = blockIdx.x * BIG_VLIST_RUNS + rr;
        if (i >= nruns) continue;
        const BDesc dsc = list[i];
        const u32 c = dsc.c & ~BDESC_TRIE, B = big_bits(c) < DB ? big_bits(c) : DB, V = 1u << B, sh = DB - B;
        if (v < V) {
            const u32 a = rel[(u64)i * 256 + (v << sh)], b = v + 1 < V ? rel[(u64)i * 256 + ((v + 1) << sh)] : c;
            const u32 hv = b - a;
            const bool fits = hv <= BIG_VCAP;
            const u64 vb = vb_[i];
            d[rr] = BDesc{dsc.start + a, hv | BDESC_TRIE | (B << BDESC_SKIP_SHIFT), (u32)(vb + v)};
            vlist[vb + v] = d[rr];
            v_count[vb + v] = fits ? (hv ? BIG_SENT : 0u) : BIG_SENT - 1;  // an empty sub-range is finished
            if (fits && hv) {
                cls[rr] = hv <= 1024 ? 0 : 1;  // the sort kernel's workgroup follows the sub-range's length
                slot[rr] = atomicAdd(&s_n[cls[rr]], 1u);
            }
        }
    }
    __syncthreads();
    if (v < 2) s_base[v] = s_n[v] ? atomicAdd(&cls_n[v], s_n[v]) : 0u;
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < BIG_VLIST_RUNS; ++rr)
        if (cls[rr] >= 0) cls_lists[(u64)cls[rr] * vtot + s_base[cls[rr]] + slot[rr]] = d[rr];
}
// One workgroup per big run: its distinct count = the sum over its sorted sub-ranges, which move down over the gaps the
// duplicates left (in the twin). A run that must stay a Vec or whose sort gave up goes to `fb` for the general kernel.
template <bool WS>
__global__ __launch_bounds__(256) void k_big_finish(const BDesc* __restrict__ list, const u32* __restrict__ list_n, const u64* __restrict__ vb_,
                                                    const BDesc* __restrict__ vlist, const u32* __restrict__ v_count, u32 SB, u64* __restrict__ t_lo,
                                                    u64* __restrict__ t_hi, u32* __restrict__ out_count, u8* __restrict__ out_kind, u8* __restrict__ in_twin,
                                                    BDesc* __restrict__ fb, u32* __restrict__ fb_n) {
    __shared__ u32 s_scan[256 / 64 + 1];
    __shared__ u32 s_off[257];
    __shared__ u32 s_fail;
    if (blockIdx.x >= *list_n) return;
    const BDesc dsc = list[blockIdx.x];
    const u64 s0 = dsc.start, vb = vb_[blockIdx.x];
    const u32 c = dsc.c & ~BDESC_TRIE;
    const bool res_trie = (dsc.c & BDESC_TRIE) != 0;
    const u32 DB = big_digit_bits(SB), B = big_bits(c) < DB ? big_bits(c) : DB, V = 1u << B, tid = threadIdx.x;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    u32 cnt = 0;
    if (tid < V) {
        cnt = v_count[vb + tid];
        if (cnt >= BIG_SENT - 1) { atomicOr(&s_fail, 1u); cnt = 0; }
    }
    u32 d;
    const u32 ex = block_exclusive_scan<256, u32>(cnt, s_scan, &d);
    s_off[tid] = ex;
    if (tid == 0) s_off[256] = d;
    __syncthreads();
    if (s_fail || (!res_trie && d <= VEC_THRESHOLD)) {
        if (tid == 0) fb[atomicAdd(fb_n, 1u)] = dsc;
        return;
    }
    if (d != c) {
        for (u32 v = 0; v < V; ++v) {
            const u32 n = s_off[v + 1] - s_off[v];
            const u64 src = vlist[vb + v].start, dst = s0 + s_off[v];  // dst <= src: chunk by chunk, every chunk read before it is written
            if (src == dst) continue;
            for (u32 e0 = 0; e0 < n; e0 += 256) {
                const u32 e = e0 + tid;
                u64 a = 0, b = 0;
                if (e < n) { a = t_lo[src + e]; if constexpr (WS) b = t_hi[src + e]; }
                __syncthreads();
                if (e < n) { t_lo[dst + e] = a; if constexpr (WS) t_hi[dst + e] = b; }
                __syncthreads();
            }
        }
    }
    if (tid == 0) {
        out_count[dsc.r] = d;
        out_kind[dsc.r] = KIND_TRIE;
        in_twin[dsc.r] = 1;
    }
}
// the buckets whose in_twin flag equals `want`: their count[r] words from src to dst at the same positions; LPB lanes per bucket
template <bool WS, int LPB>
__global__ __launch_bounds__(256) void k_copy_buckets(u64 nb, const u64* __restrict__ start, const u32* __restrict__ count, const u8* __restrict__ in_twin, u32 want,
                                                      const u64* __restrict__ src_lo, const u64* __restrict__ src_hi, u64* __restrict__ dst_lo, u64* __restrict__ dst_hi) {
    const u64 r = ((u64)blockIdx.x * 256 + threadIdx.x) / LPB;
    if (r >= nb || in_twin[r] != want) return;
    const u32 lane = threadIdx.x & (LPB - 1), c = count[r];
    const u64 s0 = start[r];
    for (u32 j = lane; j < c; j += LPB) {
        dst_lo[s0 + j] = src_lo[s0 + j];
        if constexpr (WS) dst_hi[s0 + j] = src_hi[s0 + j];
    }
}

// ---- KRN-3 huge: one workgroup per bucket, any run length; the same stable LSD radix passes, tile by tile in
// global scratch (a/b ping-pong of key + stream index). Pathological buckets (poly-A ...) only. -------------------
template <bool WS, typename HiT>
__global__ __launch_bounds__(256) void k_bucket_huge(const BDesc* __restrict__ list, const u32* __restrict__ list_n,
                                                     const u64* __restrict__ scratch_off, u64* __restrict__ lo,
                                                     HiT* __restrict__ hi, u32 SB, u64* __restrict__ a_lo,
                                                     u64* __restrict__ a_hi, u32* __restrict__ a_idx, u64* __restrict__ b_lo,
                                                     u64* __restrict__ b_hi, u32* __restrict__ b_idx,
                                                     u32* __restrict__ out_count, u8* __restrict__ out_kind, MergeArgs mg) {
    constexpr int THREADS = 256, ITEMS = 8, TILE = THREADS * ITEMS;
    __shared__ u32 s_wcnt[(THREADS / 64) * 256];
    __shared__ u32 s_dbase[256];
    __shared__ u32 s_scan[THREADS / 64 + 1];
    __shared__ u32 s_hist[256];
    __shared__ u32 s_run[256];
    __shared__ u32 s_cnt;
    if (blockIdx.x >= *list_n) return;
    const BDesc dsc = list[blockIdx.x];
    const u32 r = dsc.r;
    const u64 s0 = dsc.start;
    const u32 c = dsc.c & ~BDESC_TRIE;
    const bool res_trie = (dsc.c & BDESC_TRIE) != 0;
    const u64 so = scratch_off[blockIdx.x];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    u64 *src_lo = a_lo + so, *dst_lo = b_lo + so, *src_hi = WS ? a_hi + so : nullptr, *dst_hi = WS ? b_hi + so : nullptr;
    u32 *src_idx = a_idx + so, *dst_idx = b_idx + so;
    // stage the run into scratch A with its stream index
    for (u32 e = tid; e < c; e += THREADS) {
        Sfx<WS> s = load_sfx<WS, HiT>(lo, hi, s0 + e, SB);
        src_lo[e] = s.lo;
        if constexpr (WS) src_hi[e] = s.hi;
        src_idx[e] = e;
    }
    __syncthreads();
    const u32 npass = (SB + 7) / 8;
    const u32 ntile = (c + TILE - 1) / TILE;
    for (u32 pass = 0; pass < npass; ++pass) {
        // digit histogram of the whole run
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        for (u32 e = tid; e < c; e += THREADS) {
            Sfx<WS> s;
            s.lo = src_lo[e];
            if constexpr (WS) s.hi = src_hi[e];
            atomicAdd(&s_hist[s.digit(pass)], 1u);
        }
        __syncthreads();
        u32 hv = tid < 256 ? s_hist[tid] : 0;
        u32 ex = block_exclusive_scan<THREADS, u32>(hv, s_scan, nullptr);
        if (tid < 256) s_run[tid] = ex;  // running global start of each digit
        __syncthreads();
        for (u32 t = 0; t < ntile; ++t) {
            Sfx<WS> key[ITEMS];
            u32 idx[ITEMS], digit[ITEMS], pos[ITEMS];
            bool valid[ITEMS];
            const u32 tb = t * TILE;
            const u32 n_tile = c - tb < (u32)TILE ? c - tb : (u32)TILE;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const u32 e = w * (64 * ITEMS) + j * 64 + lane;
                valid[j] = e < n_tile;
                digit[j] = 255;  // tail slots: last in every pass
                idx[j] = 0;
                key[j].lo = 0;
                if constexpr (WS) key[j].hi = 0;
                if (valid[j]) {
                    key[j].lo = src_lo[tb + e];
                    if constexpr (WS) key[j].hi = src_hi[tb + e];
                    idx[j] = src_idx[tb + e];
                    digit[j] = key[j].digit(pass);
                }
            }
            tile_rank<THREADS, ITEMS>(digit, pos, s_wcnt, s_dbase, s_scan, ITEMS);
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                if (valid[j]) {
                    const u32 dst = s_run[digit[j]] + (pos[j] - s_dbase[digit[j]]);
                    dst_lo[dst] = key[j].lo;
                    if constexpr (WS) dst_hi[dst] = key[j].hi;
                    dst_idx[dst] = idx[j];
                }
            }
            __syncthreads();
            // advance the running starts by this tile's digit counts (= next digit base - this digit base)
            if (tid < 256) {
                u32 nxt = tid == 255 ? n_tile : s_dbase[tid + 1];
                s_run[tid] += nxt - s_dbase[tid];
            }
            __syncthreads();
        }
        u64* tl = src_lo; src_lo = dst_lo; dst_lo = tl;
        u64* th = src_hi; src_hi = dst_hi; dst_hi = th;
        u32* ti = src_idx; src_idx = dst_idx; dst_idx = ti;
        __threadfence_block();
        __syncthreads();
    }
    // src_* now sorted by suffix, stable. Count heads.
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    u32 mine = 0;
    for (u32 e = tid; e < c; e += THREADS) {
        bool h = e == 0 || src_lo[e] != src_lo[e - 1];
        if constexpr (WS) h = h || (e > 0 && src_hi[e] != src_hi[e - 1]);
        mine += h ? 1u : 0u;
    }
    mine = wave_reduce_sum(mine);
    if (lane == 0) atomicAdd(&s_cnt, mine);
    __syncthreads();
    const u32 d = s_cnt;
    if (mg.cs) {  // `self |= other` (same rules as k_bucket_medium's merge epilogue); the sorted run lives in scratch
        const u32 cs = mg.cs[r];
        // mode 0: other's elements -> other's arena; 1: heads; 2: heads from self; 3: heads from other
        auto emit = [&](int mode, u64 at) {
            u32 run = 0;
            for (u32 e0 = 0; e0 < c; e0 += THREADS) {
                const u32 e = e0 + tid;
                bool k = false;
                if (e < c) {
                    bool h = e == 0 || src_lo[e] != src_lo[e - 1];
                    if constexpr (WS) h = h || (e > 0 && src_hi[e] != src_hi[e - 1]);
                    const bool from_other = src_idx[e] >= cs;
                    k = mode == 0 ? from_other : mode == 1 ? h : mode == 2 ? (h && !from_other) : (h && from_other);
                }
                u32 tot;
                u32 ex = block_exclusive_scan<THREADS, u32>(k ? 1u : 0u, s_scan, &tot);
                if (k) {
                    Sfx<WS> s;
                    s.lo = src_lo[e];
                    if constexpr (WS) s.hi = src_hi[e];
                    if (mode == 0) {
                        mg.o_lo[at + run + ex] = s.lo;
                        if constexpr (WS) mg.o_hi[at + run + ex] = s.hi;
                    } else {
                        store_sfx<WS, HiT>(lo, hi, at + run + ex, s);
                    }
                }
                run += tot;
            }
        };
        if (mg.okind[r] == KIND_VEC) emit(0, mg.ostart[r]);
        if (res_trie) {
            emit(1, s0);
        } else {
            emit(2, s0);
            emit(3, s0 + cs);
        }
        if (tid == 0) {
            out_count[r] = d;
            out_kind[r] = res_trie ? KIND_TRIE : KIND_VEC;
        }
        return;
    }
    const bool trie = d > VEC_THRESHOLD || res_trie;
    // ordered compaction, chunk of THREADS elements at a time
    u32 base = 0;
    if (trie) {
        for (u32 e0 = 0; e0 < c; e0 += THREADS) {
            const u32 e = e0 + tid;
            bool h = false;
            if (e < c) {
                h = e == 0 || src_lo[e] != src_lo[e - 1];
                if constexpr (WS) h = h || (e > 0 && src_hi[e] != src_hi[e - 1]);
            }
            u32 tot;
            u32 ex = block_exclusive_scan<THREADS, u32>(h ? 1u : 0u, s_scan, &tot);
            if (h) {
                Sfx<WS> s;
                s.lo = src_lo[e];
                if constexpr (WS) s.hi = src_hi[e];
                store_sfx<WS, HiT>(lo, hi, s0 + base + ex, s);
            }
            base += tot;
        }
    } else {
        // mark first occurrences by stream index in dst_idx (reused as a flag array), then compact the original run
        for (u32 e = tid; e < c; e += THREADS) dst_idx[e] = 0;
        __syncthreads();
        for (u32 e = tid; e < c; e += THREADS) {
            bool h = e == 0 || src_lo[e] != src_lo[e - 1];
            if constexpr (WS) h = h || (e > 0 && src_hi[e] != src_hi[e - 1]);
            if (h) dst_idx[src_idx[e]] = 1;
        }
        __threadfence_block();
        __syncthreads();
        // the original run must be read before it is overwritten: d <= 1024 outputs, stage them in dst_lo/dst_hi
        for (u32 e0 = 0; e0 < c; e0 += THREADS) {
            const u32 e = e0 + tid;
            const bool k = e < c && dst_idx[e] != 0;
            u32 tot;
            u32 ex = block_exclusive_scan<THREADS, u32>(k ? 1u : 0u, s_scan, &tot);
            if (k) {
                Sfx<WS> s = load_sfx<WS, HiT>(lo, hi, s0 + e, SB);
                dst_lo[base + ex] = s.lo;
                if constexpr (WS) dst_hi[base + ex] = s.hi;
            }
            base += tot;
        }
        __threadfence_block();
        __syncthreads();
        for (u32 e = tid; e < d; e += THREADS) {
            Sfx<WS> s;
            s.lo = dst_lo[e];
            if constexpr (WS) s.hi = dst_hi[e];
            store_sfx<WS, HiT>(lo, hi, s0 + e, s);
        }
    }
    if (tid == 0) {
        out_count[r] = d;
        out_kind[r] = trie ? KIND_TRIE : KIND_VEC;
    }
}

// dense gather of the resident suffixes (for export / serialization): out[res_off[r] + j] = arena[start[r] + j]
__global__ void k_gather_dense(u64 e0, u64 nelem, u64 nb, const u64* __restrict__ res_off, const u64* __restrict__ start,
                               const u64* __restrict__ a_lo, const u64* __restrict__ a_hi, u32 SB, u64* __restrict__ out_lo,
                               u64* __restrict__ out_hi) {
    u64 e = e0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nelem) return;
    u64 l = 0, h = nb;
    while (h - l > 1) {
        u64 mid = (l + h) >> 1;
        if (res_off[mid] <= e) l = mid; else h = mid;
    }
    const u64 j = e - res_off[l];
    const u64 mlo = SB >= 64 ? ~0ull : ((1ull << SB) - 1ull);
    out_lo[e] = a_lo[start[l] + j] & mlo;
    if (out_hi) out_hi[e] = a_hi[start[l] + j] & ((1ull << (SB - 64)) - 1ull);
}

// membership of words in a resident index: WordSet::contains_batch (/root/reference/src/wordset/mod.rs:163-185).
// QL lanes share a query: the directory lookup is a broadcast, a Vec bucket (first-occurrence order, so nothing to
// search by) is scanned QL suffixes (64 coalesced bytes) per step, a Trie bucket (ascending) by a (QL + 1)-ary search —
// QL probes per step instead of one — until at most QL candidates are left for the equality test.
static const u32 QL = 8;
template <typename HiT>
__global__ __launch_bounds__(256) void k_contains(const u64* __restrict__ w_lo, const HiT* __restrict__ w_hi, u64 n, u32 SB, u32 PB, DirView dir,
                                                   const u64* __restrict__ a_lo, const u64* __restrict__ a_hi, u8* __restrict__ out) {
    const u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / QL;
    if (i >= n) return;  // whole groups leave together (QL divides the workgroup size)
    const u32 lane = threadIdx.x & 63u, sub = lane & (QL - 1u), gsh = lane & ~(QL - 1u);
    const u64 lo = w_lo[i], hi = ld_hi<HiT>(w_hi, i);
    const u32 p = get_bits(lo, hi, SB, PB);
    u64 r;
    bool found = false;
    if (dir_lookup(dir, p, r)) {
        const u128 M = (((u128)1) << SB) - 1;
        const u128 key = (((u128)hi << 64) | lo) & M;
        const u64 s0 = dir.start[r];
        const u32 c = dir.count[r];
        auto at = [&](u32 j) -> u128 {
            u128 v = a_lo[s0 + j];
            if (a_hi) v |= (u128)a_hi[s0 + j] << 64;
            return v & M;
        };
        auto group_any = [&](bool b) -> bool { return ((__ballot(b) >> gsh) & ((1ull << QL) - 1ull)) != 0; };
        u32 l = 0, h = c;  // candidates [l, h)
        if (dir.kind[r] == KIND_TRIE) {
            while (h - l > QL) {  // group-uniform
                const u32 width = h - l;
                const u32 m = l + (u32)(((u64)(sub + 1u) * width) / (QL + 1u));  // l < m < h, ascending in sub
                const bool less = at(m) < key;
                const u32 cnt = (u32)__builtin_popcountll((__ballot(less) >> gsh) & ((1ull << QL) - 1ull));  // probes below the key: the first cnt
                const u32 nl = cnt ? l + (u32)(((u64)cnt * width) / (QL + 1u)) + 1u : l;
                const u32 nh = cnt < QL ? l + (u32)(((u64)(cnt + 1u) * width) / (QL + 1u)) + 1u : h;
                l = nl;
                h = nh;
            }
        }
        for (u32 j0 = l; j0 < h; j0 += QL) {  // group-uniform trip count; a Vec is scanned whole unless the key turns up
            const u32 j = j0 + sub;
            found = j < h && at(j) == key;
            if (group_any(found)) { found = true; break; }
        }
    }
    if (sub == 0) out[i] = found ? 1 : 0;
}

// ---- order-independent checksum and structural validation (full-size parity properties) ------------------------------
__device__ __forceinline__ u64 mix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ u64 word_hash(u64 lo, u64 hi) { return mix64(lo + 0x9E3779B97F4A7C15ull) + mix64(hi ^ 0xD1B54A32D192ED03ull); }
// sum of word_hash over n words given as lo/hi arrays (hi may be null)
template <typename HiT>
__global__ void k_checksum_words(const u64* __restrict__ lo, const HiT* __restrict__ hi, u64 n, u64* __restrict__ out) {
    u64 s = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) s += word_hash(lo[i], ld_hi<HiT>(hi, i));
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long*)out, (unsigned long long)s);
}
// same sum over the resident index: word = prefix << SB | suffix
__global__ void k_checksum_index(u64 nelem, u64 nb, const u64* __restrict__ res_off, const u32* __restrict__ bucket_prefix,
                                 const u64* __restrict__ start, const u64* __restrict__ a_lo, const u64* __restrict__ a_hi, u32 SB,
                                 u64* __restrict__ out) {
    u64 s = 0;
    for (u64 e = (u64)blockIdx.x * blockDim.x + threadIdx.x; e < nelem; e += (u64)gridDim.x * blockDim.x) {
        u64 l = 0, h = nb;
        while (h - l > 1) {
            u64 mid = (l + h) >> 1;
            if (res_off[mid] <= e) l = mid; else h = mid;
        }
        const u64 j = e - res_off[l];
        u128 sfx = (u128)a_lo[start[l] + j];
        if (a_hi) sfx |= (u128)a_hi[start[l] + j] << 64;
        sfx &= (((u128)1) << SB) - 1;
        const u128 word = ((u128)bucket_prefix[l] << SB) | sfx;
        s += word_hash((u64)word, (u64)(word >> 64));
    }
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long*)out, (unsigned long long)s);
}
// one wave per bucket: Trie buckets strictly ascending (hence distinct), Vec buckets pairwise distinct, kinds consistent
// with the insert-only rule when `strict` (Vec <= 1024 < Trie). bad[0] += violations.
__global__ __launch_bounds__(256) void k_validate(u64 r0, u64 nb, const u64* __restrict__ start, const u32* __restrict__ count,
                                                  const u8* __restrict__ kind, const u64* __restrict__ a_lo,
                                                  const u64* __restrict__ a_hi, u32 SB, u32 strict, u64* __restrict__ bad) {
    const u64 r = r0 + (((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const u32 lane = threadIdx.x & 63;
    if (r >= nb) return;
    const u64 s0 = start[r];
    const u32 c = count[r];
    const u128 M = (((u128)1) << SB) - 1;
    auto at = [&](u32 j) -> u128 {
        u128 v = a_lo[s0 + j];
        if (a_hi) v |= (u128)a_hi[s0 + j] << 64;
        return v & M;
    };
    u64 b = 0;
    if (lane == 0) {
        if (c == 0) ++b;
        if (strict && ((kind[r] == KIND_TRIE) != (c > VEC_THRESHOLD))) ++b;
    }
    if (kind[r] == KIND_TRIE) {
        for (u32 j = lane + 1; j < c; j += 64) b += at(j - 1) >= at(j) ? 1 : 0;
    } else {
        for (u32 j = lane; j < c; j += 64) {
            const u128 x = at(j);
            for (u32 i = 0; i < j; ++i) b += at(i) == x ? 1 : 0;
        }
    }
    b = wave_reduce_sum(b);
    if (lane == 0 && b) atomicAdd((unsigned long long*)bad, (unsigned long long)b);
}

// grid-stride sum: one atomic per workgroup (launch with sum_grid(n) workgroups of 256)
__global__ __launch_bounds__(256) void k_sum_u32(const u32* __restrict__ v, u64 n, u64* __restrict__ out) {
    __shared__ u64 sm[4];
    u64 s = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) s += v[i];
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 t = sm[0] + sm[1] + sm[2] + sm[3];
        if (t) atomicAdd((unsigned long long*)out, (unsigned long long)t);
    }
}
__global__ __launch_bounds__(256) void k_sum_u8(const u8* __restrict__ v, u64 n, u64* __restrict__ out) {
    __shared__ u64 sm[4];
    u64 s = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) s += v[i];
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 t = sm[0] + sm[1] + sm[2] + sm[3];
        if (t) atomicAdd((unsigned long long*)out, (unsigned long long)t);
    }
}
__global__ void k_fill_u32(u32* p, u64 n, u32 v) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_set_u64(u64* p, u64 v) { *p = v; }


}  // namespace cblx
