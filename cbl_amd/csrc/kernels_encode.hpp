// kernels_encode.hpp — KRN-1: chunk table, validity scan, rolling 2-bit pack + necklace word per k-mer.
//
// Replaces (reference, CPU): CBL::get_seq_chunks / get_seq_words (/root/reference/src/cbl.rs:239-289),
// Base::from_nuc (/root/reference/src/kmer.rs:11-24,209-211), Kmer::from_nucs/append (:61-72,133-135),
// NecklaceQueue (/root/reference/src/necklace/queue.rs) through the normative necklace_pos
// (/root/reference/src/necklace/mod.rs:13-25), merge_necklace_pos (/root/reference/src/cbl.rs:181-184).
//
// Unit of work = a CHUNK (the reference re-seeds its queue per chunk of 2048 k-mers and, in canonical mode,
// emits a chunk's forward-strand words before its reverse-strand words, src/cbl.rs:248-275).
// A workgroup owns a TILE = all chunks whose first byte lies in a 4 KiB window of the base stream: bytes are
// read once with 16-byte coalesced loads, packed to 2 bits/base in LDS, and every lane then builds its k-mer
// from 3 (or 5) LDS dwords. Chunks holding a non-ACGT byte ("dirty", rare) are left to a scalar kernel that
// follows the reference's skip semantics exactly.
#pragma once
#include "common.hpp"
#include "necklace.hpp"

namespace cblx {

struct NoHi {};
template <typename HiT> struct HiTraits { static constexpr bool has = true; };
template <> struct HiTraits<NoHi> { static constexpr bool has = false; };
template <typename HiT> __device__ __forceinline__ u64 ld_hi(const HiT* p, u64 i) {
    if constexpr (HiTraits<HiT>::has) return (u64)p[i];
    else return 0;
}
template <typename HiT> __device__ __forceinline__ void st_hi(HiT* p, u64 i, u64 v) {
    if constexpr (HiTraits<HiT>::has) p[i] = (HiT)v;
}

// Where the bases of a batch live. The caller's format is one ASCII byte per base; a big host batch crosses PCIe as three BIT
// PLANES instead (xfer.hpp: pack_planes, 3 bits per base instead of 8 — the link is the bound of the host-input path): per 16
// bases one dword of code planes — bit i = ASCII bit 1 of base i (code bit 0), bit 16 + i = ASCII bit 2 (code bit 1), the code
// being (b >> 1) & 3 (src/kmer.rs:11-24) — and one 16-bit word of validity (bit i = base i is one of ACGTacgt).
struct BaseView {
    const u8* ascii;   // null: planes
    const u32* codes;
    const u16* valid;
    __host__ __device__ void advance16(u64 bytes /* multiple of 16 */) {
        if (ascii) ascii += bytes; else { codes += bytes >> 4; valid += bytes >> 4; }
    }
};
inline BaseView ascii_view(const u8* p) { return BaseView{p, nullptr, nullptr}; }
__device__ __forceinline__ bool bv_valid(const BaseView& B, u64 i) {
    return B.ascii ? nuc_valid(B.ascii[i]) : ((B.valid[i >> 4] >> (i & 15u)) & 1u) != 0;
}
__device__ __forceinline__ u32 bv_code(const BaseView& B, u64 i) {
    if (B.ascii) return nuc_code(B.ascii[i]);
    const u32 w = B.codes[i >> 4] >> (i & 15u);
    return (w & 1u) | ((w >> 15) & 2u);
}
// one dword of code planes -> the 16 2-bit codes of the tile's packed stream (first base in bits 31:30, like pack16)
__device__ __forceinline__ u32 planes_to_codes(u32 w) {
    auto spread = [](u32 x) {  // bit j -> bit 2 j (x < 2^16)
        x = (x | (x << 8)) & 0x00FF00FFu;
        x = (x | (x << 4)) & 0x0F0F0F0Fu;
        x = (x | (x << 2)) & 0x33333333u;
        return (x | (x << 1)) & 0x55555555u;
    };
    const u32 r0 = __brev(w << 16) & 0xFFFFu, r1 = __brev(w & 0xFFFF0000u);  // base i at bit 15 - i of each plane
    return spread(r0) | (spread(r1) << 1);
}

// ASCII bases -> the bit planes above, on the device (the "replicate" protocol of the multi-GPU build ships a rank's reads as planes:
// 3 bits per base on the wire instead of 8). One thread per group of 16 bases [16 g, 16 g + 16) of `ascii` (16-byte aligned), groups
// [g0, g1); bases at or past `end` leave their bits clear, as the host packer does (xfer.hpp). Plane words are indexed from g0.
__global__ __launch_bounds__(256) void k_pack_planes(const u8* __restrict__ ascii, u64 g0, u64 g1, u64 end, u32* __restrict__ codes, u16* __restrict__ valid) {
    const u64 g = g0 + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= g1) return;
    u32 w[4] = {0, 0, 0, 0};
    if (16 * g + 16 <= end) {
        const uint4 v = *reinterpret_cast<const uint4*>(ascii + 16 * g);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {  // the batch's last, partly filled group: byte by byte (nothing is read past the end)
        for (u64 i = 16 * g; i < end; ++i) w[(i >> 2) & 3] |= (u32)ascii[i] << (8 * (i & 3));
    }
    u32 c0 = 0, c1 = 0, ok = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const u32 b = (w[k >> 2] >> (8 * (k & 3))) & 255u;
        c0 |= ((b >> 1) & 1u) << k;
        c1 |= ((b >> 2) & 1u) << k;
        ok |= (nuc_valid((u8)b) ? 1u : 0u) << k;
    }
    const u64 base = 16 * g;
    const u32 live = base + 16 <= end ? 0xFFFFu : (base >= end ? 0u : (1u << (u32)(end - base)) - 1u);
    ok &= live;
    codes[g - g0] = (c0 & ok) | ((c1 & ok) << 16);
    valid[g - g0] = (u16)ok;
}

#ifndef CBLX_ENC_UNIFORM
#define CBLX_ENC_UNIFORM 1
#endif
#ifndef CBLX_ENC_HIST_VOTE
#define CBLX_ENC_HIST_VOTE 0  // measured (cfg 2): encode 4.69 -> 7.10 ms — the scalar vote loop is a serial dependency chain per wave
#endif
#ifndef CBLX_ENC_HIST_RUNS
#define CBLX_ENC_HIST_RUNS 0  // measured (profiles/r02_variants.md): +0.3 ms — the kernel is VALU-bound, the dozen instructions cost more than the serialised atomics
#endif
static const u32 ENC_TILE_BYTES = 4096;
static const u32 ENC_THREADS = 256;
static const u32 ENC_MAX_CHUNKS = 1024;                                    // >= 4096 / min chunk length (K >= 5)
static const u32 ENC_MAX_BASES = ENC_TILE_BYTES + CHUNK_KMERS + 64 + 32;   // bytes a tile can span (+ align slack)
static const u32 ENC_CODE_WORDS = (ENC_MAX_BASES + 15) / 16 + 8;           // packed dwords (+ read-ahead slack)
static const u32 ENC_MAX_KMERS = ENC_MAX_BASES;                            // k-mers per tile <= bytes spanned
// Optional fused histogram of the first partition pass: counts[window * 256 + digit] where window = output position /
// HIST_WINDOW (= the partition tile size). A tile's outputs are contiguous, so they touch at most ENC_HIST_WINDOWS windows.
static const u32 ENC_HIST_WINDOW = 4096;
static const u32 ENC_HIST_WINDOWS = ENC_MAX_KMERS / ENC_HIST_WINDOW + 2;
struct EncHist {
    u32* counts;                  // null: no fused histogram
    u32 shift, nbits;             // digit = bit field of the word ...
    u32 nd, SB, PB, bounds[15];   // ... or, when nd != 0, the destination rank of its prefix: #{i < nd-1 : bounds[i] <= prefix}
    u32 binRB;                    // ... or, when nd != 0 and binRB != 0, the sender's pass-A BIN (DigitBin, kernels_radix.hpp): (prefix >> binRB) + rank
    const void* cut_tab = nullptr;  // ... or, when set, the bin under a list of cuts (DigitCut, kernels_radix.hpp): {u32 cut, base} per cut_key of the prefix
    u32 cut_ksh = 0xFFFFFFFFu;      // ... or (FINE bins, cuts.hpp) per (prefix >> cut_ksh), and the bin is the count of cuts alone
    u32 reg_x = 0, reg_lmax = 0;    // ... or (FINE bins of one rank) cuts at the multiples of 2^16 up to reg_x and of 2^reg_lmax above: arithmetic, no table
    // u32 words of the cut table (the main kernel stages them in LDS: as a gather from global memory the lookup cost it 1 ms of 7.5)
    __device__ __forceinline__ u32 cut_words() const { return cut_ksh != 0xFFFFFFFFu ? 2048u : 2u * (64u + 26u * 32u); }
    // `lds`: the cut table where the caller staged it in LDS (null: read from global memory). The two sources are read under a uniform
    // branch each — ONE pointer that may hold either becomes a generic pointer, and its loads FLAT loads: the lookup cost KRN-1 1.4 ms of 7.9
    __device__ __forceinline__ u32 digit(u64 lo, u64 hi, const u32* lds = nullptr, bool staged = false) const {
        if (nd == 0) return get_bits(lo, hi, shift, nbits);
        const u32 p = get_bits(lo, hi, SB, PB);
        if (reg_x) {
            const u32 v = p >> binRB, b = ((p < reg_x ? p : reg_x) >> 16) + (p >= reg_x ? (p >> reg_lmax) - (reg_x >> reg_lmax) : 0u);
            return v >= 255u ? 255u : (b < 254u ? b : 254u);
        }
        if (cut_tab) {
            const u32 v = p >> binRB;
            const bool fine = cut_ksh != 0xFFFFFFFFu;
            u32 i0, i1;
            if (fine) {  // FINE bins: low byte = cuts at or below the cell's first prefix, upper bits = offset of the one cut inside the cell
                const u32 k = p >> cut_ksh;
                i0 = i1 = k < 2048u ? k : 2047u;
            } else {
                u32 key = p;
                if (p >= 64u) { const u32 e = 31u - (u32)__builtin_clz(p); key = 64u + ((e - 6u) << 5) + ((p >> (e - 5u)) & 31u); }
                i0 = 2 * key; i1 = 2 * key + 1;  // {the cut inside the cell (or ~0), cuts at or below its first prefix}
            }
            u32 c0, c1;
            // (explicit address spaces: left to itself the compiler folds the two sources into one generic pointer and reads the table with
            // FLAT loads inside the k-mer loop, each of them behind the loop's own global stores)
            typedef const __attribute__((address_space(3))) u32* lds_cptr;
            typedef const __attribute__((address_space(1))) u32* glb_cptr;
            if (staged) { lds_cptr L = (lds_cptr)lds; c0 = L[i0]; c1 = L[i1]; }
            else { glb_cptr g = (glb_cptr)cut_tab; c0 = g[i0]; c1 = g[i1]; }
            const u32 b = fine ? (c0 & 255u) + ((p & ((1u << cut_ksh) - 1u)) >= (c0 >> 8) ? 1u : 0u) : v + c1 + (p >= c0 ? 1u : 0u);
            return v >= 255u ? 255u : (b < 254u ? b : 254u);
        }
        u32 d = 0;
        for (u32 i = 0; i + 1 < nd; ++i) d += bounds[i] <= p ? 1u : 0u;  // nd is uniform: nd - 1 scalar-bound compares
        if (binRB) {
            const u32 v = p >> binRB, b = v + d;
            return v >= 255u ? 255u : (b < 254u ? b : 254u);
        }
        return d;
    }
};

// ---- per-sequence chunk counts; flags sequences shorter than K (src/cbl.rs:329-334) ------------------
__global__ void k_seq_chunk_count(const u64* __restrict__ offsets, u64 nseq, u32 K, u32* __restrict__ nchunks,
                                  u64* __restrict__ err /* [0]=count of short seqs, [1]=first bad len+1 */) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseq) return;
    u64 len = offsets[i + 1] - offsets[i];
    if (len < K) {
        atomicAdd((unsigned long long*)&err[0], 1ull);
        atomicMax((unsigned long long*)&err[1], (unsigned long long)(len + 1));
        nchunks[i] = 0;
        return;
    }
    u64 nk = len - K + 1;
    nchunks[i] = (u32)((nk + CHUNK_KMERS - 1) / CHUNK_KMERS);
}

// one thread per chunk: search the owning sequence in the chunk-base scan
__global__ void k_chunk_fill(const u64* __restrict__ offsets, const u64* __restrict__ chunk_base /* nseq+1 */,
                             u64 nseq, u64 nchunks, u32 K, u64 bias, u64* __restrict__ chunk_start,
                             u32* __restrict__ chunk_len, u32* __restrict__ chunk_nk) {
    u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    // last seq with chunk_base[seq] <= c. Every sequence has at least one chunk, so chunk_base[i] >= i and the answer is at
    // most min(c, nseq - 1) — exactly that when the sequences before it have one chunk each (reads): gallop down from
    // there instead of bisecting [0, nseq) (24 dependent loads per chunk at 10 M reads).
    u64 hi = (c < nseq - 1 ? c : nseq - 1) + 1, lo = hi - 1;  // invariant: chunk_base[hi] > c (or hi is past the range)
    for (u64 step = 1; chunk_base[lo] > c; step <<= 1) {
        hi = lo;
        lo = lo > step ? lo - step : 0;
    }
    while (hi - lo > 1) {
        u64 mid = (lo + hi) >> 1;
        if (chunk_base[mid] <= c) lo = mid; else hi = mid;
    }
    u64 seq = lo;
    u64 j = c - chunk_base[seq];
    u64 s0 = offsets[seq], len = offsets[seq + 1] - s0;
    u64 start = j * CHUNK_KMERS;
    u64 end = start + CHUNK_KMERS + K - 1;
    if (end > len) end = len;
    chunk_start[c] = s0 + start - bias;  // relative to the (16-byte aligned) start of this slice
    chunk_len[c] = (u32)(end - start);
    chunk_nk[c] = (u32)(end - start - K + 1);
}

// ---- validity scan: marks chunks that contain a byte outside ACGTacgt -----------------------------------
__device__ __forceinline__ u32 valid_mask4(u32 w) {  // 0x80 in every byte lane that holds a valid nucleotide
    u32 u = w & 0xDFDFDFDFu, m = 0;
    const u32 L[4] = {0x41414141u, 0x43434343u, 0x47474747u, 0x54545454u};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32 v = u ^ L[i];
        m |= ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu);
    }
    return m;
}
__device__ inline void mark_dirty(u64 b, const u64* chunk_start, const u32* chunk_len, u64 nchunks, u8* dirty, u32* ndirty) {
    u64 lo = 0, hi = nchunks;  // last chunk with start <= b
    while (hi - lo > 1) {
        u64 mid = (lo + hi) >> 1;
        if (chunk_start[mid] <= b) lo = mid; else hi = mid;
    }
    for (int k = 0; k < 2; ++k) {  // a byte lies in at most 2 chunks (K-1 overlap inside a long sequence)
        if (lo < (u64)k) break;
        u64 c = lo - k;
        if (chunk_start[c] <= b && b < chunk_start[c] + chunk_len[c]) {
            dirty[c] = 1;  // benign race: every writer stores 1
            *ndirty = 1u;  // a flag, not a count (one atomicAdd per invalid byte on this word cost 25 ms when every read had an N)
        }
    }
}
__global__ void k_scan_invalid(BaseView B, u64 total, const u64* __restrict__ chunk_start,
                               const u32* __restrict__ chunk_len, u64 nchunks, u8* __restrict__ dirty,
                               u32* __restrict__ ndirty) {
    u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 b0 = g * 16;
    if (b0 >= total) return;
    if (B.ascii) {
        const u8* __restrict__ bases = B.ascii;
        if (b0 + 16 <= total) {
            uint4 v = *reinterpret_cast<const uint4*>(bases + b0);
            u32 m = valid_mask4(v.x) & valid_mask4(v.y) & valid_mask4(v.z) & valid_mask4(v.w);
            if (m == 0x80808080u) return;
        }
        for (u64 b = b0; b < b0 + 16 && b < total; ++b)
            if (!nuc_valid(bases[b])) mark_dirty(b, chunk_start, chunk_len, nchunks, dirty, ndirty);
    } else {
        const u32 n = total - b0 < 16 ? (u32)(total - b0) : 16u;
        u32 bad = ~(u32)B.valid[g] & ((1u << n) - 1u);  // the packer leaves the bits past the end of the batch clear
        while (bad) {
            const u32 k = (u32)__builtin_ctz(bad);
            bad &= bad - 1u;
            mark_dirty(b0 + k, chunk_start, chunk_len, nchunks, dirty, ndirty);
        }
    }
}
// The dirty chunks as a list (any order), so that the kernels below put a whole wave on every one of them: one THREAD per
// dirty chunk walking its 150 .. 2100 bytes alone, among 63 idle lanes, cost 2 ms per 82 000 dirty chunks (1 % of the reads
// with an N) and 130 ms when every read had one.
static const u32 DIRTY_LIST_THREADS = 256;
__global__ __launch_bounds__(DIRTY_LIST_THREADS) void k_dirty_list(const u8* __restrict__ dirty, u64 nchunks, u32* __restrict__ list, u32* __restrict__ list_n) {
    __shared__ u32 s_cnt[DIRTY_LIST_THREADS / 64];
    __shared__ u32 s_base;
    const u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool d = c < nchunks && dirty[c];
    const u64 bal = __ballot(d);
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s_cnt[w] = (u32)__builtin_popcountll(bal);
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (u32 i = 0; i < DIRTY_LIST_THREADS / 64; ++i) { const u32 t = s_cnt[i]; s_cnt[i] = run; run += t; }
        s_base = run ? atomicAdd(list_n, run) : 0u;
    }
    __syncthreads();
    if (d) list[s_base + s_cnt[w] + mbcnt(bal)] = (u32)c;
}
// exact k-mer count of a dirty chunk: 1 + #valid bytes in chunk[K..] (src/cbl.rs:277-287 filter_map), one wave per chunk
__global__ __launch_bounds__(256) void k_dirty_count_wave(BaseView B, const u64* __restrict__ chunk_start, const u32* __restrict__ chunk_len,
                                                          const u32* __restrict__ list, u32 nlist, u32 K, u32* __restrict__ chunk_nk) {
    const u32 li = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (li >= nlist) return;
    const u32 c = list[li];
    const u64 s0 = chunk_start[c];
    const u32 len = chunk_len[c];
    u32 m = 0;
    for (u32 i = K + lane; i < len; i += 64) m += bv_valid(B, s0 + i) ? 1u : 0u;
    m = (u32)wave_reduce_sum((u64)m);
    if (lane == 0) chunk_nk[c] = 1 + m;
}

// first chunk of every 4 KiB tile of the base stream (lower_bound over chunk_start)
__global__ void k_tile_first_chunk(const u64* __restrict__ chunk_start, u64 nchunks, u64 ntiles, u32* __restrict__ tile_first) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > ntiles) return;
    u64 key = t * ENC_TILE_BYTES;
    u64 lo = 0, hi = nchunks;  // first c with chunk_start[c] >= key
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if (chunk_start[mid] < key) lo = mid + 1; else hi = mid;
    }
    tile_first[t] = (u32)lo;
}

#ifndef CBLX_ENC_PROBE
#define CBLX_ENC_PROBE 0
#endif
// ---- word of one k-mer ---------------------------------------------------------------------------------
template <bool WIDE> struct KmerT;
template <> struct KmerT<false> { typedef u64 type; };
template <> struct KmerT<true> { typedef u128 type; };

template <bool WIDE>
__device__ __forceinline__ void kmer_word(typename KmerT<WIDE>::type x, const Consts& P, bool take_rc, u64& lo, u64& hi) {
    typedef typename KmerT<WIDE>::type T;
    if (take_rc) {
        if constexpr (WIDE) x = rev_comp128(x, P.K); else x = rev_comp64(x, P.K);
    }
    T nk;
    unsigned pos;
    necklace_pos_fast<T>(x, P.KB, nk, pos);
    if constexpr (!WIDE) {  // 0 < POS < 64: two plain 64-bit shifts instead of a generic 128-bit one
        lo = ((u64)nk << P.POS) | (u64)pos;
        hi = (u64)nk >> (64 - P.POS);
    } else {
        u128 word = ((u128)nk << P.POS) | (u128)pos;
        lo = (u64)word;
        hi = (u64)(word >> 64);
    }
}
template <bool WIDE> __device__ __forceinline__ bool kmer_is_fwd(typename KmerT<WIDE>::type x) {  // Kmer::is_canonical, src/kmer.rs:94-96
    if constexpr (WIDE) return (popcount128(x) & 1u) == 0; else return (__builtin_popcountll(x) & 1) == 0;
}

// k-mer starting at base index s of the big-endian packed stream `codes` (16 bases per dword, first base on top)
template <bool WIDE> __device__ __forceinline__ typename KmerT<WIDE>::type extract_kmer(const u32* codes, u32 s, u32 K) {
    const u32 w = s >> 4, o = (s & 15u) * 2u;
    if constexpr (!WIDE) {
        u64 a = ((u64)codes[w] << 32) | codes[w + 1];
        u64 v = (a << o) | (((u64)codes[w + 2] << o) >> 32);  // no branch: o = 0 contributes nothing
        return v >> (64 - 2 * K);
    } else {
        u128 a = ((u128)codes[w] << 96) | ((u128)codes[w + 1] << 64) | ((u128)codes[w + 2] << 32) | (u128)codes[w + 3];
        u128 v = (a << o) | (u128)(((u64)codes[w + 4] << o) >> 32);
        return v >> (128 - 2 * K);
    }
}

__device__ __forceinline__ u32 pack16(uint4 v) {  // 16 ASCII bases -> 16 2-bit codes, first base in bits 31:30
    u32 out = 0;
    const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        u32 c = (w[i] >> 1) & 0x03030303u;
        u32 r = ((c & 3u) << 6) | (((c >> 8) & 3u) << 4) | (((c >> 16) & 3u) << 2) | ((c >> 24) & 3u);
        out |= r << (24 - 8 * i);
    }
    return out;
}

// ---- main encode kernel: one workgroup per 4 KiB tile of the base stream --------------------------------
template <bool WIDE, typename HiT>
__global__ __launch_bounds__(ENC_THREADS) void k_encode(BaseView B, u64 total_bases,
                                                        const u64* __restrict__ chunk_start,
                                                        const u32* __restrict__ chunk_len,
                                                        const u64* __restrict__ kmer_off /* nchunks+1 */,
                                                        const u8* __restrict__ dirty /* may be null */,
                                                        const u32* __restrict__ tile_first, Consts P,
                                                        u64* __restrict__ out_lo, HiT* __restrict__ out_hi, u64 out_base,
                                                        EncHist eh, u32 c_lo = 0, u32 c_hi = 0xFFFFFFFFu /* only the chunks in [c_lo, c_hi): a slice of a plan */) {
    typedef typename KmerT<WIDE>::type T;
    __shared__ u32 s_hist[ENC_HIST_WINDOWS * 256];
    __shared__ u32 s_codes[ENC_CODE_WORDS];
    __shared__ u32 s_koff[ENC_MAX_CHUNKS + 1];
    __shared__ u32 s_cstart[ENC_MAX_CHUNKS];  // first base of the chunk in the tile's code stream; bit 31: the chunk is dirty
    __shared__ u64 s_par[(ENC_MAX_KMERS + ENC_THREADS) / 64 + 2];
    __shared__ u32 s_parpre[(ENC_MAX_KMERS + ENC_THREADS) / 64 + 2];
    __shared__ u32 s_cfwd[ENC_MAX_CHUNKS + 1];  // canonical: forward-strand k-mers of the tile in front of each chunk
    extern __shared__ u32 s_cut[];              // the cut table of the fused histogram's bins (launched with its size only when there is one)

    const u32 tid = threadIdx.x;
    u32 c0 = tile_first[blockIdx.x], c1 = tile_first[blockIdx.x + 1];
    c0 = c0 > c_lo ? c0 : c_lo;  // (a tile on the edge of the slice holds chunks of its neighbour: theirs to encode)
    c1 = c1 < c_hi ? c1 : c_hi;
    if (c0 >= c1) return;
    const u32 nc = c1 - c0;  // <= ENC_MAX_CHUNKS by construction (chunk length >= K >= 5)
    const u64 B0 = chunk_start[c0];
    const u64 B1 = chunk_start[c1 - 1] + chunk_len[c1 - 1];
    const u64 A0 = B0 & ~(u64)15;
    const u64 kbase = kmer_off[c0];

    for (u32 i = tid; i <= nc; i += ENC_THREADS) s_koff[i] = (u32)(kmer_off[c0 + i] - kbase);
    for (u32 i = tid; i < nc; i += ENC_THREADS) {
        s_cstart[i] = (u32)(chunk_start[c0 + i] - A0) | ((dirty && dirty[c0 + i]) ? 0x80000000u : 0u);
    }
    const u64 win0 = (out_base + kbase) / ENC_HIST_WINDOW;  // first window this tile's outputs fall into
    if (eh.counts)
        for (u32 i = tid; i < ENC_HIST_WINDOWS * 256; i += ENC_THREADS) s_hist[i] = 0;
    const bool cut_staged = eh.counts && eh.cut_tab;
    if (cut_staged) {  // (the barrier behind the code stream below covers it)
        const u32 nw = eh.cut_words();
        for (u32 i = tid * 4; i < nw; i += ENC_THREADS * 4) *reinterpret_cast<uint4*>(s_cut + i) = *reinterpret_cast<const uint4*>(reinterpret_cast<const u32*>(eh.cut_tab) + i);
    }
    const u32 nwords = (u32)((B1 - A0 + 15) >> 4);
    for (u32 i = tid; i < nwords + 6 && i < ENC_CODE_WORDS; i += ENC_THREADS) {
        u64 b = A0 + (u64)i * 16;
        u32 packed = 0;
        if (i < nwords) {
            if (!B.ascii) {
                if (b < total_bases) packed = planes_to_codes(B.codes[b >> 4]);  // (bits past the end of the batch are clear)
            } else if (b + 16 <= total_bases) {
                packed = pack16(*reinterpret_cast<const uint4*>(B.ascii + b));
            } else {
                for (u32 k = 0; k < 16 && b + k < total_bases; ++k) packed |= nuc_code(B.ascii[b + k]) << (30 - 2 * k);
            }
        }
        s_codes[i] = packed;
    }
    __syncthreads();
    const u32 Q = s_koff[nc];
#if CBLX_ENC_UNIFORM
    // Reads of one length back to back (the common input): every chunk of the tile has nk0 k-mers and starts len0 bases
    // after its predecessor, so chunk and position of k-mer q follow from q by arithmetic and the k-mer loop never reads
    // the chunk tables.
    const u32 nk0 = s_koff[1], cs0 = s_cstart[0], len0 = nc > 1 ? s_cstart[1] - s_cstart[0] : 0u;
    bool uni = nk0 != 0 && !(cs0 >> 31);
    // (a dirty chunk carries bit 31 in s_cstart: chunk 1 must be tested for it explicitly, len0 is defined by its start)
    for (u32 i = tid; i < nc; i += ENC_THREADS) uni = uni && s_koff[i + 1] == (i + 1) * nk0 && s_cstart[i] == cs0 + i * len0 && !(s_cstart[i] >> 31);
    const bool uniform = __syncthreads_and(uni ? 1 : 0) != 0;
#else
    const bool uniform = false;
    const u32 nk0 = 1, cs0 = 0, len0 = 0;
#endif

    auto find_chunk = [&](u32 q) -> u32 {  // last i with s_koff[i] <= q
        u32 lo = 0, hi = nc;
        while (hi - lo > 1) {
            u32 mid = (lo + hi) >> 1;
            if (s_koff[mid] <= q) lo = mid; else hi = mid;
        }
        return lo;
    };

    if (P.canonical) {
        // strand flags of every k-mer of the tile: bit = 1 -> forward strand (even popcount)
        const u32 qpad = (Q + ENC_THREADS - 1) / ENC_THREADS * ENC_THREADS;
        u32 cf = tid < Q ? find_chunk(tid) : 0u;
        for (u32 q = tid; q < qpad; q += ENC_THREADS) {
            bool fwd = false;
            if (q < Q) {
                while (q >= s_koff[cf + 1]) ++cf;
                const u32 cs = s_cstart[cf];
                if (!(cs >> 31)) fwd = kmer_is_fwd<WIDE>(extract_kmer<WIDE>(s_codes, cs + (q - s_koff[cf]), P.K));
            }
            u64 bal = __ballot(fwd);
            if ((tid & 63) == 0) s_par[q >> 6] = bal;
        }
        __syncthreads();
        if (tid == 0) {
            u32 run = 0;
            const u32 nw = qpad >> 6;
            for (u32 i = 0; i < nw; ++i) { s_parpre[i] = run; run += (u32)__builtin_popcountll(s_par[i]); }
            s_parpre[nw] = run;
            s_par[nw] = 0;
        }
        __syncthreads();
    }
    auto cum_fwd = [&](u32 q) -> u32 {
        return s_parpre[q >> 6] + (u32)__builtin_popcountll(s_par[q >> 6] & ((1ull << (q & 63)) - 1ull));
    };
    if (P.canonical) {  // once per chunk instead of twice per k-mer
        for (u32 i = tid; i <= nc; i += ENC_THREADS) s_cfwd[i] = cum_fwd(s_koff[i]);
        __syncthreads();
    }

    // The k-mer loop with K as a compile-time constant (every rotate / mask / shift of the necklace then has constant
    // operands) or read from P.
    auto kmer_loop = [&](auto kc) {
        constexpr u32 KC = decltype(kc)::value;
        Consts PK = P;
        if constexpr (KC != 0) { PK.K = KC; PK.KB = 2 * KC; PK.POS = 32 - __builtin_clz(2 * KC - 1); }
        const u64 obase = out_base + kbase;
        const u32 hbase = (u32)(obase & (ENC_HIST_WINDOW - 1));  // (obase + i) / WINDOW - win0 == (hbase + i) / WINDOW
        auto body = [&](u32 q, u32 ci, u32 cs, u32 j, u32 koff) {
            T x = extract_kmer<WIDE>(s_codes, cs + j, PK.K);
            u32 drel = q;  // output slot relative to obase
            bool rc = false;
            if (P.canonical) {
                const u32 cb = s_cfwd[ci];
                const u32 nfwd = s_cfwd[ci + 1] - cb;
                const u32 rf = cum_fwd(q) - cb;
                rc = !kmer_is_fwd<WIDE>(x);
                drel = koff + (rc ? (nfwd + (j - rf)) : rf);
            }
            u64 lo, hi;
            kmer_word<WIDE>(x, PK, rc, lo, hi);
#if CBLX_ENC_PROBE == 1  // timing probe only (tools/dev_encode_probe.py): one byte per k-mer instead of the word
            reinterpret_cast<u8*>(out_lo)[obase + drel] = (u8)lo ^ (u8)hi;
#else
            out_lo[obase + drel] = lo;
            st_hi<HiT>(out_hi, obase + drel, hi);
#endif
            if (eh.counts) {
                const u32 key = ((hbase + drel) / ENC_HIST_WINDOW) * 256 + eh.digit(lo, hi, s_cut, cut_staged);
#if CBLX_ENC_HIST_VOTE
                // The first-pass digit is the skewed one: a wave's 64 keys are a handful of distinct values (2-3 on
                // average), which per-lane LDS atomics serialise address by address. Instead the wave votes value by
                // value on the scalar unit: broadcast the first unsettled lane's key, ballot who shares it, one lane adds
                // the count. The loop is scalar work plus one compare per round.
                u64 todo = __ballot(true);
                while (todo) {
                    const u32 k0 = (u32)__builtin_amdgcn_readlane((int)key, (int)__builtin_ctzll(todo));
                    const u64 same = __ballot(key == k0);
                    if ((tid & 63u) == (u32)__builtin_ctzll(todo)) atomicAdd(&s_hist[k0], (u32)__builtin_popcountll(same));
                    todo &= ~same;
                }
#elif CBLX_ENC_HIST_RUNS
                // Neighbouring lanes hold consecutive k-mers of a read: their necklaces share the leading bits, and the
                // first-pass digit is the skewed one — a wave's 64 updates hit a handful of counters, which the LDS
                // serialises address by address. One update per RUN of equal keys instead: the first lane of a run adds
                // the run's length (lanes that sit this iteration out break a run).
                const u64 act = __ballot(true);
                const u32 lane = tid & 63u;
                const u32 prev = (u32)__builtin_amdgcn_update_dpp((int)~key, (int)key, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
                const bool lead = lane == 0 || !((act >> (lane - 1)) & 1ull) || prev != key;
                const u64 leaders = __ballot(lead);
                if (lead) {
                    const u64 above = lane == 63 ? 0ull : ((leaders | ~act) >> (lane + 1));  // next leader or inactive lane ends the run
                    const u32 len = above ? (u32)__builtin_ctzll(above) + 1u : 64u - lane;
                    atomicAdd(&s_hist[key], len);
                }
#else
                atomicAdd(&s_hist[key], 1u);
#endif
            }
        };
        if (uniform) {
            u32 ci = tid / nk0, j = tid - ci * nk0;
            for (u32 q = tid; q < Q; q += ENC_THREADS) {
                body(q, ci, cs0 + ci * len0, j, ci * nk0);
                j += ENC_THREADS;
                while (j >= nk0) { j -= nk0; ++ci; }
            }
        } else {
            u32 ci = tid < Q ? find_chunk(tid) : 0u;
            for (u32 q = tid; q < Q; q += ENC_THREADS) {
                while (q >= s_koff[ci + 1]) ++ci;  // q only grows: walk forward from the previous chunk instead of searching again
                const u32 cs = s_cstart[ci];
                if (cs >> 31) continue;
                body(q, ci, cs, q - s_koff[ci], s_koff[ci]);
            }
        }
    };
    // the K of BASELINE.json's configurations (and the other common choices, 21 and 27) get their own copy of the loop; any
    // other K reads it from P
    if constexpr (!WIDE) {
        if (P.K == 31) kmer_loop(std::integral_constant<u32, 31>());
        else if (P.K == 25) kmer_loop(std::integral_constant<u32, 25>());
        else if (P.K == 21) kmer_loop(std::integral_constant<u32, 21>());
        else if (P.K == 27) kmer_loop(std::integral_constant<u32, 27>());
        else kmer_loop(std::integral_constant<u32, 0>());
    } else {
        if (P.K == 59) kmer_loop(std::integral_constant<u32, 59>());
        else kmer_loop(std::integral_constant<u32, 0>());
    }
#if CBLX_ENC_PROBE == 2  // timing probe only (wrong histogram): what the flush of the tile's counts costs
    if (false) {
#else
    if (eh.counts) {  // the first-pass digit is the skewed one: only a few dozen bins per window are non-zero
#endif
        __syncthreads();
        for (u32 i = tid; i < ENC_HIST_WINDOWS * 256; i += ENC_THREADS) {
            const u32 v = s_hist[i];
            if (v) atomicAdd(&eh.counts[(win0 + (i >> 8)) * 256 + (i & 255u)], v);
        }
    }
}

// ---- dirty chunks (a byte outside ACGTacgt): the reference's skip semantics ----------------------------------------------
// words = K-windows of  zeros(K - n0) ++ valid(chunk[0..K]) ++ valid(chunk[K..])   (src/kmer.rs:133-135: the first
// k-mer is folded from the valid bases among the first K BYTES; src/cbl.rs:283 filter_map on the rest).
// One WAVE per dirty chunk (list from k_dirty_list). The chunk's words are the K-windows of a cleaned base string
// — zeros(K - n0) ++ its valid bases — so the wave first compacts the valid bases into LDS (ballot + prefix per 64 bytes),
// packs them 16 to a dword like the main kernel's code stream, and then every lane takes k-mers lane, lane + 64, .. from it
// with the main kernel's own extract / word functions. Canonical order (forward-strand words of the chunk first) from a
// first sweep that only takes the strand flags.
static const u32 DIRTY_MAX_BASES = CHUNK_KMERS + 64 + 64;  // a chunk spans at most CHUNK_KMERS + K - 1 bytes, K <= 59
template <bool WIDE, typename HiT>
__global__ __launch_bounds__(256) void k_encode_dirty_wave(BaseView B, const u64* __restrict__ chunk_start, const u32* __restrict__ chunk_len,
                                                           const u64* __restrict__ kmer_off, const u32* __restrict__ list, u32 nlist, Consts P,
                                                           u64* __restrict__ out_lo, HiT* __restrict__ out_hi, u64 out_base, EncHist eh, u32 c_lo = 0, u32 c_hi = 0xFFFFFFFFu) {
    typedef typename KmerT<WIDE>::type T;
    constexpr u32 NWV = 4, CW = DIRTY_MAX_BASES / 16 + 8, PW = DIRTY_MAX_BASES / 64 + 2;
    __shared__ u8 s_cb_all[NWV][DIRTY_MAX_BASES + 16];
    __shared__ u32 s_codes_all[NWV][CW];
    __shared__ u64 s_par_all[NWV][PW];
    __shared__ u32 s_hist_all[NWV][2 * 256];  // the chunk's outputs are contiguous and at most 2048 + K: two histogram windows
    const u32 wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 li = blockIdx.x * NWV + wv;
    if (li >= nlist) return;  // no workgroup barrier below: waves are on their own
    u8* s_cb = s_cb_all[wv];
    u32* s_codes = s_codes_all[wv];
    u64* s_par = s_par_all[wv];
    u32* s_hist = s_hist_all[wv];
    if (eh.counts)
        for (u32 i = lane; i < 2 * 256; i += 64) s_hist[i] = 0;
    const u32 c = list[li];
    if (c < c_lo || c >= c_hi) return;  // (another slice's chunk)
    const u64 s0 = chunk_start[c];
    const u32 len = chunk_len[c], K = P.K;
    const u64 o0 = out_base + kmer_off[c];
    // cleaned string: zeros(K - n0), then the valid bases in order (src/kmer.rs:133-135, src/cbl.rs:283)
    const u64 first = __ballot(lane < K && lane < len && bv_valid(B, s0 + lane));
    const u32 n0 = (u32)__builtin_popcountll(first), Z = K - n0;
    for (u32 i = lane; i < Z; i += 64) s_cb[i] = 0;
    u32 fill = Z;
    for (u32 i0 = 0; i0 < len; i0 += 64) {
        const u32 i = i0 + lane;
        const bool ok = i < len && bv_valid(B, s0 + i);
        const u64 bal = __ballot(ok);
        if (ok) s_cb[fill + mbcnt(bal)] = (u8)bv_code(B, s0 + i);
        fill += (u32)__builtin_popcountll(bal);
    }
    const u32 nk = fill - K + 1;  // = kmer_off[c + 1] - kmer_off[c] (k_dirty_count_wave); fill >= K always
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const u32 nw = (fill + 15) / 16;
    for (u32 w = lane; w < nw + 6 && w < CW; w += 64) {
        u32 packed = 0;
        if (w < nw)
            for (u32 k = 0; k < 16 && w * 16 + k < fill; ++k) packed |= (u32)s_cb[w * 16 + k] << (30 - 2 * k);
        s_codes[w] = packed;
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    u32 nfwd = 0;
    if (P.canonical) {  // strand flags of every k-mer, 64 per word; nfwd = forward-strand k-mers of the chunk
        for (u32 j0 = 0; j0 < nk; j0 += 64) {
            const u32 j = j0 + lane;
            const bool fwd = j < nk && kmer_is_fwd<WIDE>(extract_kmer<WIDE>(s_codes, j, K));
            const u64 bal = __ballot(fwd);
            if (lane == 0) s_par[j0 >> 6] = bal;
            nfwd += (u32)__builtin_popcountll(bal);
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
    const u64 win0 = o0 / ENC_HIST_WINDOW;
    u32 fwd_before = 0;  // forward-strand k-mers in front of this 64-k-mer step
    for (u32 j0 = 0; j0 < nk; j0 += 64) {
        const u32 j = j0 + lane;
        u64 fbal = 0;
        if (P.canonical) fbal = s_par[j0 >> 6];
        if (j < nk) {
            const T x = extract_kmer<WIDE>(s_codes, j, K);
            bool rc = false;
            u64 dst = o0 + j;
            if (P.canonical) {
                rc = !((fbal >> lane) & 1ull);
                const u32 fb = fwd_before + mbcnt(fbal);  // forward-strand k-mers of the chunk in front of k-mer j
                dst = rc ? o0 + nfwd + (j - fb) : o0 + fb;
            }
            u64 lo, hi;
            kmer_word<WIDE>(x, P, rc, lo, hi);
            out_lo[dst] = lo;
            st_hi<HiT>(out_hi, dst, hi);
            // (one device atomic per k-mer on the count matrix — neighbours share their digit, hence their address — was
            // most of this kernel's time; per wave an LDS histogram, flushed below)
            if (eh.counts) atomicAdd(&s_hist[(u32)(dst / ENC_HIST_WINDOW - win0) * 256 + eh.digit(lo, hi)], 1u);
        }
        fwd_before += (u32)__builtin_popcountll(fbal);
    }
    if (eh.counts) {
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
        for (u32 i = lane; i < 2 * 256; i += 64) {
            const u32 v = s_hist[i];
            if (v) atomicAdd(&eh.counts[(win0 + (i >> 8)) * 256 + (i & 255u)], v);
        }
    }
}

}  // namespace cblx
