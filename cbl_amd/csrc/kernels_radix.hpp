// kernels_radix.hpp — KRN-2: stable LSD radix partition of (lo, hi) word records by their prefix bits,
// plus the device-wide exclusive scan it needs.
//
// Replaces (reference, CPU): WordSet::split_prefix_suffix + the per-group Fenwick rank / tiered-vector
// lookup that routes each word to its bucket (/root/reference/src/wordset/mod.rs:63-71,187-216).
// Stability is what carries the reference's "first occurrence order" inside a Vec bucket
// (/root/reference/src/trievec/mod.rs:81-87): equal prefixes keep stream order through every pass.
//
// Tile ranking (no LDS atomics): a wave owns a contiguous slice of the tile; per 64-element round the lanes
// holding the same 8-bit digit find each other with 8 ballots, rank = per-wave running count + lanes below
// (v_mbcnt); per-wave counts are then scanned across waves and digits. Elements are staged through LDS in
// sorted order so every digit's run leaves the tile as one contiguous global write.
#pragma once
#include "kernels_encode.hpp"

namespace cblx {

// ------------------------------------------------------------------------------------------------
// device-wide exclusive scan of u32 -> u64 (three kernels: block sums, spine, apply)
static const u32 SCAN_THREADS = 256;
static const u32 SCAN_ITEMS = 16;
static const u32 SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const u32* __restrict__ in, u64 n, u64* __restrict__ block_sums) {
    __shared__ u64 sm[SCAN_THREADS / 64];
    u64 base = (u64)blockIdx.x * SCAN_TILE;
    u64 s = 0;
    for (u32 j = 0; j < SCAN_ITEMS; ++j) {
        u64 i = base + (u64)j * SCAN_THREADS + threadIdx.x;
        if (i < n) s += in[i];
    }
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 t = 0;
        for (u32 w = 0; w < SCAN_THREADS / 64; ++w) t += sm[w];
        block_sums[blockIdx.x] = t;
    }
}
// single workgroup: exclusive scan of block_sums[0..nb) in place, total to block_sums[nb]
__global__ __launch_bounds__(1024) void k_scan_spine(u64* __restrict__ block_sums, u64 nb) {
    __shared__ u64 sm[1024 / 64 + 1];
    __shared__ u64 carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (u64 base = 0; base < nb; base += 1024) {
        u64 i = base + threadIdx.x;
        u64 v = i < nb ? block_sums[i] : 0;
        u64 total;
        u64 ex = block_exclusive_scan<1024, u64>(v, sm, &total);
        u64 carry = carry_s;
        if (i < nb) block_sums[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[nb] = carry_s;
}
template <typename OutT>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const u32* __restrict__ in, u64 n, const u64* __restrict__ block_sums,
                                                             OutT* __restrict__ out) {
    __shared__ u64 sm[SCAN_THREADS / 64 + 1];
    // blocked arrangement: thread t owns items [t*ITEMS, (t+1)*ITEMS) of the tile
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
    u32 v[SCAN_ITEMS];
    u64 s = 0;
#pragma unroll
    for (u32 j = 0; j < SCAN_ITEMS; ++j) {
        u64 i = base + j;
        v[j] = i < n ? in[i] : 0u;
        s += v[j];
    }
    u64 ex = block_exclusive_scan<SCAN_THREADS, u64>(s, sm, nullptr) + block_sums[blockIdx.x];
#pragma unroll
    for (u32 j = 0; j < SCAN_ITEMS; ++j) {
        u64 i = base + j;
        if (i < n) out[i] = (OutT)ex;
        ex += v[j];
    }
}

// ------------------------------------------------------------------------------------------------
// Stable ranking of one tile by an 8-bit digit.
//   THREADS = 64 * NW lanes, `rounds` (<= ITEMS, workgroup-uniform) rounds; element e = w * (64*rounds) + j * 64 + lane
//   (wave-contiguous slices, so the order of equal digits is (wave, round, lane) = tile order).
//   Every lane of every round takes part: slots past the end of the data must carry digit 255 and be the LAST
//   slots of the tile; being stable, the ranking then leaves them in the last positions.
//   s_wcnt: LDS [NW][256] u32.  s_dbase: LDS [256] u32 (tile-local start of each digit, valid after return).
//   Returns pos[j] = tile-local sorted position. Contains __syncthreads().
template <int THREADS, int ITEMS>
__device__ __forceinline__ void tile_rank(const u32 (&digit)[ITEMS], u32 (&pos)[ITEMS], u32* s_wcnt, u32* s_dbase,
                                          u32* s_scan /* THREADS/64+1 */, u32 rounds) {
    constexpr int NW = THREADS / 64;
    const u32 tid = threadIdx.x, w = tid >> 6;
    for (u32 i = tid; i < NW * 256; i += THREADS) s_wcnt[i] = 0;
    __syncthreads();
    u32* my = s_wcnt + w * 256;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if ((u32)j < rounds) {
            const u32 d = digit[j];
            u64 m = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const u32 bit = (d >> b) & 1u;
                const u64 bal = __ballot(bit != 0);
                m &= bal ^ (u64)(long long)((int)bit - 1);  // lanes whose bit b equals mine
            }
            const u32 lower = mbcnt(m);
            const u32 tot = (u32)__builtin_popcountll(m);
            const u32 old = my[d];
            __builtin_amdgcn_wave_barrier();
            if (lower == 0) my[d] = old + tot;
            __builtin_amdgcn_wave_barrier();
            pos[j] = old + lower;
        }
    }
    __syncthreads();
    // per digit: exclusive scan across waves; digit totals -> exclusive scan across digits
    u32 dtot = 0;
    if (tid < 256) {
        u32 run = 0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            u32 t = s_wcnt[ww * 256 + tid];
            s_wcnt[ww * 256 + tid] = run;
            run += t;
        }
        dtot = run;
    }
    u32 ex = block_exclusive_scan<THREADS, u32>(dtot, s_scan, nullptr);
    if (tid < 256) s_dbase[tid] = ex;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < ITEMS; ++j)
        if ((u32)j < rounds) pos[j] += s_dbase[digit[j]] + my[digit[j]];
}

static const int RDX_THREADS = 512;
static const int RDX_ITEMS = 8;
static const int RDX_TILE = RDX_THREADS * RDX_ITEMS;  // 4096 records per workgroup

// digit of a record: a bit field of the word (LSD passes) ...
struct DigitBits {
    u32 shift, nbits;
    __device__ __forceinline__ u32 operator()(u64 lo, u64 hi) const { return get_bits(lo, hi, shift, nbits); }
};
// ... or the destination rank of its prefix under quantile range sharding: dest = #{i : bounds[i] <= prefix}
static const u32 MAX_DEST = 16;
struct DigitDest {
    u32 SB, PB, nd;
    u32 bounds[MAX_DEST - 1];
    __device__ __forceinline__ u32 operator()(u64 lo, u64 hi) const {
        const u32 p = get_bits(lo, hi, SB, PB);
        u32 d = 0;
#pragma unroll
        for (u32 i = 0; i < MAX_DEST - 1; ++i) d += (i + 1 < nd && bounds[i] <= p) ? 1u : 0u;
        return d;
    }
};

// per-tile digit histogram -> counts[digit * ntiles + tile]
template <typename HiT, typename DigitFn>
__global__ __launch_bounds__(RDX_THREADS) void k_radix_hist(const u64* __restrict__ lo, const HiT* __restrict__ hi, u64 n,
                                                            DigitFn dfn, u32 ntiles, u32* __restrict__ counts) {
    __shared__ u32 s_wcnt[(RDX_THREADS / 64) * 256];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    for (u32 i = tid; i < (RDX_THREADS / 64) * 256; i += RDX_THREADS) s_wcnt[i] = 0;
    __syncthreads();
    const u64 tbase = (u64)blockIdx.x * RDX_TILE;
    u32* my = s_wcnt + w * 256;
#pragma unroll 4
    for (int j = 0; j < RDX_ITEMS; ++j) {
        const u64 i = tbase + (u64)w * (64 * RDX_ITEMS) + (u64)j * 64 + lane;
        const bool v = i < n;
        u32 d = 0;
        if (v) d = dfn(lo[i], ld_hi<HiT>(hi, i));
        u64 m = __ballot(v);
        const u64 vm = m;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const u64 bal = __ballot(bit && v);
            m &= bit ? bal : ~bal;
        }
        m &= vm;
        if (v && mbcnt(m) == 0) my[d] += (u32)__builtin_popcountll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (tid < 256) {
        u32 t = 0;
#pragma unroll
        for (int ww = 0; ww < RDX_THREADS / 64; ++ww) t += s_wcnt[ww * 256 + tid];
        counts[(u64)tid * ntiles + blockIdx.x] = t;
    }
}

// scatter: offsets[digit * ntiles + tile] = global position of the tile's first element with that digit
template <typename HiT, typename DigitFn>
__global__ __launch_bounds__(RDX_THREADS) void k_radix_scatter(const u64* __restrict__ lo, const HiT* __restrict__ hi, u64 n,
                                                               DigitFn dfn, u32 ntiles,
                                                               const u32* __restrict__ offsets, u64* __restrict__ out_lo,
                                                               HiT* __restrict__ out_hi) {
    __shared__ u64 s_lo[RDX_TILE];
    __shared__ typename std::conditional<HiTraits<HiT>::has, HiT, u8>::type s_hi[HiTraits<HiT>::has ? RDX_TILE : 1];
    __shared__ u32 s_wcnt[(RDX_THREADS / 64) * 256];
    __shared__ u32 s_dbase[256];
    __shared__ u64 s_gbase[256];
    __shared__ u32 s_scan[RDX_THREADS / 64 + 1];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const u64 tbase = (u64)blockIdx.x * RDX_TILE;
    const u32 n_tile = (u32)((n - tbase) < (u64)RDX_TILE ? (n - tbase) : (u64)RDX_TILE);

    u64 klo[RDX_ITEMS];
    typename std::conditional<std::is_same<HiT, u64>::value, u64, u32>::type khi[RDX_ITEMS];
    u32 digit[RDX_ITEMS], pos[RDX_ITEMS];
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) {
        const u32 e = w * (64 * RDX_ITEMS) + j * 64 + lane;
        const bool valid = e < n_tile;
        const u64 idx = valid ? tbase + e : tbase;  // tail slots re-read slot 0 and are never written back
        klo[j] = lo[idx];
        khi[j] = ld_hi<HiT>(hi, idx);
        digit[j] = valid ? dfn(klo[j], (u64)khi[j]) : 255u;
    }
    tile_rank<RDX_THREADS, RDX_ITEMS>(digit, pos, s_wcnt, s_dbase, s_scan, RDX_ITEMS);
    if (tid < 256) s_gbase[tid] = offsets[(u64)tid * ntiles + blockIdx.x] - s_dbase[tid];
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) {
        s_lo[pos[j]] = klo[j];  // pos < RDX_TILE always; tail slots land in [n_tile, RDX_TILE)
        if constexpr (HiTraits<HiT>::has) s_hi[pos[j]] = (HiT)khi[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) {
        const u32 s = j * RDX_THREADS + tid;
        if (s < n_tile) {
            const u64 a = s_lo[s];
            u64 b = 0;
            if constexpr (HiTraits<HiT>::has) b = (u64)s_hi[s];
            const u32 d = dfn(a, b);
            const u64 dst = s_gbase[d] + s;
            out_lo[dst] = a;
            st_hi<HiT>(out_hi, dst, b);
        }
    }
}

}  // namespace cblx
