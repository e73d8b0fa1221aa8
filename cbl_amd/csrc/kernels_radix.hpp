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
#include <type_traits>

#include "cuts.hpp"
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
            // lanes holding my digit: AND over the 8 bits of (ballot(bit) XNOR my bit). t = 0 / ~0 is my bit spread over
            // a dword (one v_bfe_i32), so each half of the mask costs one v_xnor + one v_and per bit.
            u32 m_lo = ~0u, m_hi = ~0u;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const u32 t = (u32)__builtin_amdgcn_sbfe((int)d, (unsigned)b, 1u);
                const u64 bal = __ballot(t != 0);
                m_lo &= ~((u32)bal ^ t);
                m_hi &= ~((u32)(bal >> 32) ^ t);
            }
            const u32 lower = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
            const u32 tot = (u32)__builtin_popcount(m_lo) + (u32)__builtin_popcount(m_hi);
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

// Same ranking with digit and position in ONE register per element: in: dp[j] = digit; out: dp[j] = digit << 16 | position
// (tile <= 65536 records). The scatter kernel is register-bound: 8 registers fewer keep it at 64 VGPRs without spills.
template <int THREADS, int ITEMS>
__device__ __forceinline__ void tile_rank_packed(u32 (&dp)[ITEMS], u32* s_wcnt, u32* s_dbase,
                                          u32* s_scan /* THREADS/64+1 */, u32 rounds) {
    constexpr int NW = THREADS / 64;
    const u32 tid = threadIdx.x, w = tid >> 6;
    for (u32 i = tid; i < NW * 256; i += THREADS) s_wcnt[i] = 0;
    __syncthreads();
    u32* my = s_wcnt + w * 256;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if ((u32)j < rounds) {
            const u32 d = dp[j];
            // lanes holding my digit: AND over the 8 bits of (ballot(bit) XNOR my bit). t = 0 / ~0 is my bit spread over
            // a dword (one v_bfe_i32), so each half of the mask costs one v_xnor + one v_and per bit.
            u32 m_lo = ~0u, m_hi = ~0u;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const u32 t = (u32)__builtin_amdgcn_sbfe((int)d, (unsigned)b, 1u);
                const u64 bal = __ballot(t != 0);
                m_lo &= ~((u32)bal ^ t);
                m_hi &= ~((u32)(bal >> 32) ^ t);
            }
            const u32 lower = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
            const u32 tot = (u32)__builtin_popcount(m_lo) + (u32)__builtin_popcount(m_hi);
            const u32 old = my[d];
            __builtin_amdgcn_wave_barrier();
            if (lower == 0) my[d] = old + tot;
            __builtin_amdgcn_wave_barrier();
            dp[j] = (old + lower) | (d << 16);
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
        if ((u32)j < rounds) { const u32 d = dp[j] >> 16; dp[j] += s_dbase[d] + my[d]; }
}

#ifndef CBLX_RDX_THREADS
#define CBLX_RDX_THREADS 512
#endif
#ifndef CBLX_RDX_ITEMS
#define CBLX_RDX_ITEMS 8
#endif
static const int RDX_THREADS = CBLX_RDX_THREADS;
static const int RDX_ITEMS = CBLX_RDX_ITEMS;
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
        for (u32 i = 0; i + 1 < nd; ++i) d += bounds[i] <= p ? 1u : 0u;  // nd is uniform: nd - 1 scalar-bound compares
        return d;
    }
};

// ... or (sender side of the multi-GPU build, comm.hpp) the BIN of its prefix: the value of the top prefix bits pass A sorts
// by, refined by the destination rank, bin = (prefix >> RB) + #{i : bounds[i] <= prefix}. Monotone in the prefix, so a pass on
// it leaves the records contiguous per destination AND per pass-A segment inside every destination. It fits 8 bits: a
// necklace with its top bit set is all ones (every bit comes to the top under some rotation, and the necklace is the
// smallest rotation), so the top-8 values 128..254 never occur and 0..127 + (nd - 1 <= 15) stays below 255, which is left
// to the all-ones word.
struct DigitBin {
    u32 SB, PB, RB, nd;
    u32 bounds[MAX_DEST - 1];
    __device__ __forceinline__ u32 operator()(u64 lo, u64 hi) const {
        const u32 p = get_bits(lo, hi, SB, PB);
        u32 d = 0;
        for (u32 i = 0; i + 1 < nd; ++i) d += bounds[i] <= p ? 1u : 0u;
        const u32 v = p >> RB, b = v + d;
        return v >= 255u ? 255u : (b < 254u ? b : 254u);
    }
};
// ... or (grouped receiver, comm.hpp) the bin under an arbitrary ascending list of CUTS of the prefix space — the destination
// bounds plus, inside every destination's range, the bounds of the GROUPS the receiver works through while later groups are
// still on the wire: bin = (prefix >> RB) + #{cuts <= prefix}. Up to 126 cuts: counting them by compares would cost two
// VALU instructions each, so the count comes from a table over a floating-point-like key of the prefix (the 6 leading bits:
// 32 cells per octave, exact below 64): entry = {cuts <= the cell's first prefix, the one cut inside the cell or ~0}. The host
// refuses a cut list with two cuts inside one cell (make_cut_table); 768 entries of 8 bytes, read through the vector L1.
// FINE bins (cuts.hpp: make_fine_plan; PREFIX_BITS > 24): bin = #{cuts <= prefix} alone, from a linear table of FINE_CELLS u32 over
// prefix >> ksh (low byte = cuts at or below the cell's first prefix, upper bits = offset of the one cut inside the cell).
// The kernels STAGE the table in LDS (`staged`): as a gather from global memory it cost the first pass 3.7 ms of 9.6 per 1.5 G records and
// KRN-1 1 ms of 7.5 (a wave's 64 lanes touch about ten cells: ten L1 lines per wave instruction next to the streaming records).
struct DigitCut {
    u32 SB, PB, RB;
    const void* tab;        // CutCell[CUT_KEYS] (key = cut_key(prefix)), or u32[FINE_CELLS] for FINE bins
    u32 ksh = 0xFFFFFFFFu;  // FINE bins: key shift of the linear table
    // FINE bins of ONE rank (comm.hpp insert_device_fine): the cuts are the multiples of 2^16 up to reg_x and the multiples of 2^reg_lmax
    // above it — the count is arithmetic, no table (reg_x = 0: not this case)
    u32 reg_x = 0, reg_lmax = 0;
    static constexpr u32 LDS_WORDS = 2048;  // u32 words of LDS either table takes (CUT_KEYS * 2 = 1792 <= FINE_CELLS = 2048)
    __device__ __forceinline__ u32 words() const { return reg_x ? 0u : (ksh != 0xFFFFFFFFu ? FINE_CELLS : CUT_KEYS * 2); }
    // bin of a word from the table at `t` (LDS or global)
    __device__ __forceinline__ u32 at(const u32* t, u64 lo, u64 hi) const {
        const u32 p = get_bits(lo, hi, SB, PB);
        const u32 v = p >> RB;
        u32 b;
        if (reg_x) {
            b = ((p < reg_x ? p : reg_x) >> 16) + (p >= reg_x ? (p >> reg_lmax) - (reg_x >> reg_lmax) : 0u);
        } else if (ksh != 0xFFFFFFFFu) {
            const u32 k = p >> ksh, c = t[k < FINE_CELLS ? k : FINE_CELLS - 1u];
            b = (c & 255u) + ((p & ((1u << ksh) - 1u)) >= (c >> 8) ? 1u : 0u);
        } else {
            const u32 k = cut_key(p);
            b = v + t[2 * k + 1] + (p >= t[2 * k] ? 1u : 0u);  // CutCell{cut, base}
        }
        return v >= 255u ? 255u : (b < 254u ? b : 254u);
    }
    __device__ __forceinline__ u32 operator()(u64 lo, u64 hi) const { return at(reinterpret_cast<const u32*>(tab), lo, hi); }
};
template <typename F> struct DigitUsesTable : std::false_type {};
template <> struct DigitUsesTable<DigitCut> : std::true_type {};
template <typename F> __device__ __forceinline__ u32 digit_at(const F& f, const u32* t, u64 lo, u64 hi) {
    if constexpr (DigitUsesTable<F>::value) return f.at(t, lo, hi); else return f(lo, hi);
}
// Where the records of the sender's OWN destination go: positions [a, b) of the pass's output order leave for other arrays
// (position - a); what lies behind them moves down by b - a, so the send buffer holds the other ranks' records only.
struct OwnWindow {
    u64 a, b;
    u64* lo;
    void* hi;
    u8* next;
};

// XCD-aware workgroup -> tile map. The dispatcher is observed to place workgroup b on XCD b % 8 (speed only, never
// relied on for correctness): XCD x then walks the CONTIGUOUS tile range [x * per, (x + 1) * per) in order, so the runs
// that neighbouring tiles append to the same bin are written through the same L2 close in time and their partial
// cache lines merge there instead of reaching HBM twice. Launch xcd_grid(ntiles) workgroups.
#ifndef CBLX_XCD_MAP
#define CBLX_XCD_MAP 1
#endif
__host__ __device__ inline u32 xcd_grid(u32 ntiles) { return ((ntiles + 7u) / 8u) * 8u; }
__device__ __forceinline__ u32 xcd_tile(u32 b, u32 ntiles) {
#if CBLX_XCD_MAP
    const u32 per = (ntiles + 7u) / 8u;
    return (b & 7u) * per + (b >> 3);
#else
    return b;
#endif
}

// Where the tiles of a pass live. Plain passes cut [0, n) into RDX_TILE-record tiles. After the first (most significant
// digit) pass the array is a sequence of SEGMENTS (one per value of that digit); the remaining passes are stable LSD
// passes INSIDE every segment, so their tiles never straddle a segment: start/count/seg come from a table built on the
// device (k_seg_table / k_tile_table) and the tile count is read from device memory (no host round trip).
struct TileView {
    const u32* start;       // null: tile t = records [t * RDX_TILE, ...)
    const u32* count;
    const u16* seg;         // segment of the tile (null: 0)
    const u32* ntiles_dev;  // null: ntiles below
    u32 ntiles;
    u64 n;
    // tiles anywhere in a resident arena (the long runs of the bucket stage, kernels_bucket.hpp: k_big_*): 64-bit starts and
    // 32-bit segment numbers instead of start / seg
    const u64* start64 = nullptr;
    const u32* seg32 = nullptr;
};
__device__ __forceinline__ bool tile_get(const TileView& tv, u32 b, u32& tile, u64& tbase, u32& n_tile, u32& seg) {
    const u32 nt = tv.ntiles_dev ? *tv.ntiles_dev : tv.ntiles;
    if ((b >> 3) >= (nt + 7u) / 8u) return false;  // the grid may be sized for an upper bound of the tile count
    tile = xcd_tile(b, nt);
    if (tile >= nt) return false;
    if (tv.start64) {
        tbase = tv.start64[tile];
        n_tile = tv.count[tile];
        seg = tv.seg32[tile];
    } else if (tv.start) {
        tbase = tv.start[tile];
        n_tile = tv.count[tile];
        seg = tv.seg[tile];
    } else {
        tbase = (u64)tile * RDX_TILE;
        n_tile = (u32)((tv.n - tbase) < (u64)RDX_TILE ? (tv.n - tbase) : (u64)RDX_TILE);
        seg = 0;
    }
    return true;
}

// per-tile digit histogram -> counts[tile * 256 + digit] (tile-major: one coalesced 1 KiB row per workgroup).
// Counting needs no ranks: per-wave private LDS histograms fed by non-returning ds_add (the ballot matching of
// tile_rank costs ~50 VALU per record and made this kernel VALU-bound).
template <typename HiT, typename DigitFn>
__global__ __launch_bounds__(RDX_THREADS) void k_radix_hist(const u64* __restrict__ lo, const HiT* __restrict__ hi, TileView tv,
                                                            DigitFn dfn, u32* __restrict__ counts) {
    __shared__ u32 s_wcnt[(RDX_THREADS / 64) * 256];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    for (u32 i = tid; i < (RDX_THREADS / 64) * 256; i += RDX_THREADS) s_wcnt[i] = 0;
    __syncthreads();
    u32 tile, n_tile, seg;
    u64 tbase;
    if (!tile_get(tv, blockIdx.x, tile, tbase, n_tile, seg)) return;
    u32* my = s_wcnt + w * 256;
    u64 klo[RDX_ITEMS];
    typename std::conditional<std::is_same<HiT, u64>::value, u64, u32>::type khi[RDX_ITEMS];
    const u64* __restrict__ lo_t = lo + tbase;
    const HiT* __restrict__ hi_t = HiTraits<HiT>::has ? hi + tbase : hi;
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) {
        const u32 e = w * (64 * RDX_ITEMS) + j * 64 + lane;
        const u32 eo = e < n_tile ? e : 0u;
        klo[j] = lo_t[eo];
        khi[j] = ld_hi<HiT>(hi_t, eo);
    }
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) {
        const u32 e = w * (64 * RDX_ITEMS) + j * 64 + lane;
        if (e < n_tile) atomicAdd(&my[dfn(klo[j], (u64)khi[j])], 1u);
    }
    __syncthreads();
    if (tid < 256) {
        u32 t = 0;
#pragma unroll
        for (int ww = 0; ww < RDX_THREADS / 64; ++ww) t += s_wcnt[ww * 256 + tid];
        counts[(u64)tile * 256 + tid] = t;
    }
}

// The same histogram from the DIGIT SIDE CHANNEL the previous pass's scatter left behind (dig[i] = this pass's digit
// of record i): 1 byte per record is read instead of the whole record. A thread takes one aligned 8-byte word of the
// tile's byte range (RDX_THREADS * 8 = RDX_TILE bytes, plus one word for an unaligned start).
static const int HISTB_WAVES = 4;  // tiles per workgroup of k_radix_hist_bytes
__global__ __launch_bounds__(64 * HISTB_WAVES) void k_radix_hist_bytes(const u8* __restrict__ dig, TileView tv, u32* __restrict__ counts) {
    // one WAVE per tile (no workgroup barrier): a lane takes every 64th aligned 8-byte word of the tile's byte range.
    // Two private counter sets per wave (even / odd lanes) halve the same-address serialisation of the LDS atomics
    // (four sets measured the same).
    __shared__ u32 s_cnt[HISTB_WAVES * 512];
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32* my = s_cnt + w * 512;
#pragma unroll
    for (int k = 0; k < 8; ++k) my[k * 64 + lane] = 0;
    u32 tile, n_tile, seg;
    u64 tbase;
    // tile_get maps a WORKGROUP index to a tile (XCD-aware); here the unit is the wave
    if (!tile_get(tv, blockIdx.x * HISTB_WAVES + w, tile, tbase, n_tile, seg)) return;
    __builtin_amdgcn_wave_barrier();
    u32* mine = my + (lane & 1u) * 256;
    const u64 w0 = tbase >> 3, wend = (tbase + n_tile + 7) >> 3;  // aligned words covering [tbase, tbase + n_tile)
    const u64* __restrict__ words = reinterpret_cast<const u64*>(dig);
    // a tile spans at most RDX_TILE / 8 + 1 aligned words = 8 full rounds of the wave + one word: all loads are issued
    // before the first counter update (one load -> 8 LDS atomics -> next load left the wave latency-bound)
    constexpr int ROUNDS = RDX_TILE / 512 + 1;
    u64 v[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const u64 wi = w0 + lane + (u64)r * 64;
        v[r] = wi < wend ? words[wi] : 0ull;
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const u64 wi = w0 + lane + (u64)r * 64;
        if (wi < wend) {
            const u64 b0 = wi << 3;
            if (b0 >= tbase && b0 + 8 <= tbase + n_tile) {  // whole word inside the tile (all but the first and last)
#pragma unroll
                for (int k = 0; k < 8; ++k) atomicAdd(&mine[(u32)(v[r] >> (8 * k)) & 255u], 1u);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const u64 pos = b0 + k;
                    if (pos >= tbase && pos < tbase + n_tile) atomicAdd(&mine[(u32)(v[r] >> (8 * k)) & 255u], 1u);
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
#pragma unroll
    for (int k = 0; k < 4; ++k) counts[(u64)tile * 256 + k * 64 + lane] = my[k * 64 + lane] + my[256 + k * 64 + lane];
}

// scatter. colpre[tile * 256 + d] = records with digit d in earlier tiles (pure column prefix); adj[seg * 256 + d] turns it
// into a global position (k_seg_adjust). OutHiT = NoHi drops the hi part on the way out (first pass of 65..72-bit words:
// the bits it held are implied by the segment from then on).
template <typename HiT, typename OutHiT, typename DigitFn, bool REDIR = false>
__global__ __launch_bounds__(RDX_THREADS, (HiTraits<HiT>::has && HiTraits<OutHiT>::has) ? 4 : 8) void k_radix_scatter(const u64* __restrict__ lo, const HiT* __restrict__ hi, TileView tv,
                                                               DigitFn dfn, const u32* __restrict__ colpre,
                                                               const u32* __restrict__ adj, u64* __restrict__ out_lo,
                                                               OutHiT* __restrict__ out_hi, DigitBits next_dfn = DigitBits{0, 0},
                                                               u8* __restrict__ out_next = nullptr, u32* __restrict__ start_dense = nullptr,
                                                               u32 pfx_shift = 0, u32 pfx_bits = 0, u32 grp_bits = 0, u32* __restrict__ amb = nullptr,
                                                               u32 amb_stride = 0, OwnWindow ow = OwnWindow{0, 0, nullptr, nullptr, nullptr},
                                                               const u64* __restrict__ seg_base = nullptr /* adj is relative to the segment's own start */,
                                                               const u32* __restrict__ seg_prefix = nullptr /* fused directory: first prefix of the segment's block (null: seg << pfx_bits) */) {
    constexpr bool STAGE_HI = HiTraits<HiT>::has && HiTraits<OutHiT>::has;
    constexpr bool TABLE = DigitUsesTable<DigitFn>::value;  // the digit comes from a table staged in LDS (dead once the records are staged)
    constexpr bool KEEP_DIG = !STAGE_HI || TABLE;
    __shared__ u64 s_lo[RDX_TILE];
    __shared__ typename std::conditional<STAGE_HI, HiT, u8>::type s_hi[STAGE_HI ? RDX_TILE : 1];
    __shared__ u8 s_dig[KEEP_DIG ? RDX_TILE : 1];  // digit of the staged record when it cannot be recomputed from lo alone
    // the ranking counters live in the staging area (they are dead before the first record is staged): 39 KB of LDS per
    // workgroup instead of 47 KB = four resident workgroups per CU instead of three
    static_assert((RDX_THREADS / 64) * 256 * 4 <= RDX_TILE * 8, "rank counters must fit the staging area");
    u32* s_wcnt = reinterpret_cast<u32*>(s_lo);
    __shared__ u32 s_dbase[256];
    __shared__ u64 s_gbase[256];
    __shared__ u32 s_scan[RDX_THREADS / 64 + 1];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    u32 tile, n_tile, seg;
    u64 tbase;
    if (!tile_get(tv, blockIdx.x, tile, tbase, n_tile, seg)) return;

    u64 klo[RDX_ITEMS];
    typename std::conditional<std::is_same<HiT, u64>::value, u64, u32>::type khi[RDX_ITEMS];
    u32 digit[RDX_ITEMS];
    const u64* __restrict__ lo_t = lo + tbase;
    const HiT* __restrict__ hi_t = HiTraits<HiT>::has ? hi + tbase : hi;
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) {
        const u32 e = w * (64 * RDX_ITEMS) + j * 64 + lane;
        const bool valid = e < n_tile;
        const u32 eo = valid ? e : 0u;  // tail slots re-read slot 0 and are never written back
        // streaming read (uniform tile base + 32-bit lane offset): the records are used once, and what they would evict
        // from the L2 are the partial lines of the output waiting for their neighbours (-1.5 .. -2.6 % per launch; the
        // same hint on the STORES doubles the time: they are what must stay)
        klo[j] = __builtin_nontemporal_load(&lo_t[eo]);
        if constexpr (HiTraits<HiT>::has) khi[j] = __builtin_nontemporal_load(&hi_t[eo]); else khi[j] = 0;
        if constexpr (!TABLE) digit[j] = valid ? dfn(klo[j], (u64)khi[j]) : 255u;
    }
    if constexpr (TABLE) {
        // the bin table goes into LDS behind the rank counters (the first 8 KB of the staging area) — its loads are issued behind the
        // records' so that the tile waits for memory once, not twice
        static_assert((RDX_THREADS / 64) * 256 * 4 + DigitCut::LDS_WORDS * 4 <= RDX_TILE * 8, "rank counters + digit table must fit the staging area");
        u32* t = reinterpret_cast<u32*>(s_lo) + (RDX_THREADS / 64) * 256;
        const u32 nw = dfn.words();
        if (nw) {
            for (u32 i = tid * 4; i < nw; i += RDX_THREADS * 4) *reinterpret_cast<uint4*>(t + i) = *reinterpret_cast<const uint4*>(reinterpret_cast<const u32*>(dfn.tab) + i);
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < RDX_ITEMS; ++j) {
            const u32 e = w * (64 * RDX_ITEMS) + j * 64 + lane;
            digit[j] = e < n_tile ? dfn.at(t, klo[j], (u64)khi[j]) : 255u;
        }
    }
    tile_rank_packed<RDX_THREADS, RDX_ITEMS>(digit, s_wcnt, s_dbase, s_scan, RDX_ITEMS);  // digit[j] = digit << 16 | position
    if (tid < 256) {
        // seg_base: adj is relative to the segment's own start and may be "negative" (the column prefix counts the digit in
        // the tiles of EARLIER segments too): 32-bit modular arithmetic, sign-extended (a segment is shorter than 2^31)
        if (seg_base) s_gbase[tid] = seg_base[seg] + (u64)(long long)(int)(adj[(u64)seg * 256 + tid] + colpre[(u64)tile * 256 + tid] - s_dbase[tid]);
        else s_gbase[tid] = (u64)adj[(u64)seg * 256 + tid] + colpre[(u64)tile * 256 + tid] - s_dbase[tid];
    }
    __syncthreads();  // every wave is done with the rank counters before records are staged over them
    u32 g_first = 0;
    if (start_dense) {  // fused directory (last pass): group the tile starts in; its row of parked candidates starts empty
        g_first = get_bits(lo_t[0], (u64)ld_hi<HiT>(hi_t, 0), pfx_shift, pfx_bits) & ((1u << grp_bits) - 1u);
        if (tid < amb_stride) amb[(u64)tile * amb_stride + tid] = 0xFFFFFFFFu;
    }
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) {
        const u32 pj = digit[j] & 0xFFFFu;  // < RDX_TILE always; tail slots land in [n_tile, RDX_TILE)
        s_lo[pj] = klo[j];
        if constexpr (STAGE_HI) s_hi[pj] = (HiT)khi[j];
        if constexpr (KEEP_DIG) s_dig[pj] = (u8)(digit[j] >> 16);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) {
        const u32 s = j * RDX_THREADS + tid;
        if (s < n_tile) {
            const u64 a = s_lo[s];
            u64 b = 0;
            u32 d;
            if constexpr (STAGE_HI) b = (u64)s_hi[s];
            if constexpr (KEEP_DIG) d = s_dig[s]; else d = dfn(a, b);
            u64 dst = s_gbase[d] + s;
            if constexpr (REDIR) {  // multi-GPU sender: the records of its own prefix range go straight to the receive arena
                const bool own = dst >= ow.a && dst < ow.b;
                dst -= own ? ow.a : (dst >= ow.b ? ow.b - ow.a : 0ull);
                u64* pl = own ? ow.lo : out_lo;
                OutHiT* ph = own ? (OutHiT*)ow.hi : out_hi;
                u8* pn = own ? ow.next : out_next;
                // (no send buffer — the "replicate" protocol, comm.hpp: every rank transforms every rank's reads and keeps the words of its
                // own prefix range; the others' words are dropped here)
                if (own || out_lo) {
                    pl[dst] = a;
                    st_hi<OutHiT>(ph, dst, b);
                    if (pn) pn[dst] = (u8)next_dfn(a, b);
                }
            } else {
            out_lo[dst] = a;
            st_hi<OutHiT>(out_hi, dst, b);
            // side channel: the next pass's histogram reads 1 byte per record instead of the record (measured: packing four
            // digits of a run into one unaligned dword store is slower than the byte stores — the extra LDS pass costs more)
            if (out_next) out_next[dst] = (u8)next_dfn(a, b);
            }
            // LAST pass, bucket directory on the fly: the tile (sorted by this digit, and by the lower digits before that)
            // is in final order, so a record whose prefix differs from its predecessor's starts a bucket — unless an
            // earlier tile holds the same prefix, which can only be for the GROUP (value of the lower digits) the tile
            // starts in. Those (at most one per digit) are parked in amb[tile][digit] and settled by k_dir_resolve, which
            // looks at the record in front of them in the finished array (atomicMin here cost 7 ms at 2^28 prefixes).
            if (start_dense) {
                const u32 p = get_bits(a, b, pfx_shift, pfx_bits);
                bool st = s == 0;
                if (s > 0) {
                    const u64 a1 = s_lo[s - 1];
                    u64 b1 = 0;
                    if constexpr (STAGE_HI) b1 = (u64)s_hi[s - 1];
                    st = get_bits(a1, b1, pfx_shift, pfx_bits) != p;
                }
                if (st) {
                    if ((p & ((1u << grp_bits) - 1u)) == g_first) amb[(u64)tile * amb_stride + d] = (u32)dst;
                    else start_dense[(seg_prefix ? (u64)seg_prefix[seg] : (u64)seg << pfx_bits) | p] = (u32)dst;
                }
            }
        }
    }
}
// settles the parked candidates of the fused directory: a candidate starts its bucket iff the record in front of it (in
// the finished array) has another prefix or belongs to another segment
template <typename HiT>
__global__ void k_dir_resolve(const u32* __restrict__ ntiles_dev, u32 amb_stride, const u32* __restrict__ amb, const u16* __restrict__ t_seg,
                              const u32* __restrict__ seg_start, const u64* __restrict__ lo, const HiT* __restrict__ hi, u32 pfx_shift,
                              u32 pfx_bits, u32* __restrict__ start_dense, const u32* __restrict__ seg_prefix = nullptr) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (u64)*ntiles_dev * amb_stride) return;
    const u32 dst = amb[i];
    if (dst == 0xFFFFFFFFu) return;
    const u32 seg = t_seg ? t_seg[i / amb_stride] : 0u;
    const u32 p = get_bits(lo[dst], (u64)ld_hi<HiT>(hi, dst), pfx_shift, pfx_bits);
    const bool first = dst == (seg_start ? seg_start[seg] : 0u) || get_bits(lo[dst - 1], (u64)ld_hi<HiT>(hi, dst - 1), pfx_shift, pfx_bits) != p;
    if (first) start_dense[(seg_prefix ? (u64)seg_prefix[seg] : (u64)seg << pfx_bits) | p] = dst;
}

// ------------------------------------------------------------------------------------------------
// Column prefixes of the tile-major count matrix C[tile][256]: P[tile][d] = sum_{t' < tile} C[t'][d], and the column
// totals. Every access is a coalesced 1 KiB row (thread d owns column d); rows are cut into chunks of COLSCAN_ROWS for
// parallelism. k_seg_adjust then supplies, per segment, what must be added to P to get a global position.
static const u32 COLSCAN_ROWS = 1024;
__device__ __forceinline__ u32 dev_ntiles(const u32* ntiles_dev, u32 ntiles) { return ntiles_dev ? *ntiles_dev : ntiles; }
__global__ __launch_bounds__(256) void k_colscan_reduce(const u32* __restrict__ counts, const u32* ntiles_dev, u32 ntiles,
                                                        u32* __restrict__ chunk_sums) {
    const u32 nt = dev_ntiles(ntiles_dev, ntiles);
    const u32 d = threadIdx.x, r0 = blockIdx.x * COLSCAN_ROWS;
    const u32 r1 = r0 + COLSCAN_ROWS < nt ? r0 + COLSCAN_ROWS : nt;
    u32 s = 0;
#pragma unroll 8
    for (u32 r = r0; r < r1; ++r) s += counts[(u64)r * 256 + d];
    chunk_sums[(u64)blockIdx.x * 256 + d] = s;
}
__global__ __launch_bounds__(256) void k_colscan_spine(u32* __restrict__ chunk_sums, u32 nchunks, u32* __restrict__ coltot /* 256 */) {
    const u32 d = threadIdx.x;
    u32 run = 0;
#pragma unroll 8
    for (u32 c = 0; c < nchunks; ++c) {
        const u32 v = chunk_sums[(u64)c * 256 + d];
        chunk_sums[(u64)c * 256 + d] = run;
        run += v;
    }
    coltot[d] = run;
}
__global__ __launch_bounds__(256) void k_colscan_apply(const u32* __restrict__ counts, const u32* ntiles_dev, u32 ntiles,
                                                       const u32* __restrict__ chunk_sums, u32* __restrict__ colpre) {
    const u32 nt = dev_ntiles(ntiles_dev, ntiles);
    const u32 d = threadIdx.x, r0 = blockIdx.x * COLSCAN_ROWS;
    const u32 r1 = r0 + COLSCAN_ROWS < nt ? r0 + COLSCAN_ROWS : nt;
    u32 run = chunk_sums[(u64)blockIdx.x * 256 + d];
#pragma unroll 8
    for (u32 r = r0; r < r1; ++r) {
        const u32 v = counts[(u64)r * 256 + d];
        colpre[(u64)r * 256 + d] = run;
        run += v;
    }
}
// One workgroup per segment s (tiles [first[s], first[s+1]), records from seg_start[s]):
//   adj[s][d] = seg_start[s] + (records of the segment with a smaller digit) - P[first[s]][d]
// so that adj[s][d] + P[tile][d] is where tile's first record with digit d goes. A plain pass is the case of one segment.
__global__ __launch_bounds__(256) void k_seg_adjust(const u32* __restrict__ colpre, const u32* __restrict__ coltot,
                                                    const u32* __restrict__ seg_first /* nseg+1, null: {0, ntiles} */,
                                                    const u32* __restrict__ seg_start /* nseg, null: {0} */, const u32* ntiles_dev,
                                                    u32 ntiles, u32 nseg, u32* __restrict__ adj, u32* __restrict__ grp_start = nullptr) {
    __shared__ u32 sm[256 / 64 + 1];
    const u32 nt = dev_ntiles(ntiles_dev, ntiles);
    const u32 s = blockIdx.x, d = threadIdx.x;
    const u32 f0 = seg_first ? seg_first[s] : 0u, f1 = seg_first ? seg_first[s + 1] : nt;
    const u32 p0 = f0 < nt ? colpre[(u64)f0 * 256 + d] : coltot[d];
    const u32 p1 = f1 < nt ? colpre[(u64)f1 * 256 + d] : coltot[d];
    const u32 ex = block_exclusive_scan<256, u32>(p1 - p0, sm, nullptr);
    adj[(u64)s * 256 + d] = (seg_start ? seg_start[s] : 0u) + ex - p0;
    // where digit d of segment s starts after this pass: the (segment, digit) GROUPS the next pass may cut its tiles at
    if (grp_start) grp_start[(u64)s * 256 + d] = (seg_start ? seg_start[s] : 0u) + ex;
    (void)nseg;
}
// Segments of the remaining passes from the first pass's column totals: seg_start (exclusive scan of the totals),
// seg_first (exclusive scan of the tiles each segment needs), total tile count.
__global__ __launch_bounds__(256) void k_seg_table(const u32* __restrict__ coltot, u32* __restrict__ seg_start /* 257 */,
                                                   u32* __restrict__ seg_first /* 257 */, u32* __restrict__ ntiles_dev) {
    __shared__ u32 sm[256 / 64 + 1];
    const u32 d = threadIdx.x;
    const u32 c = coltot[d];
    u32 tot;
    const u32 st = block_exclusive_scan<256, u32>(c, sm, &tot);
    seg_start[d] = st;
    if (d == 255) seg_start[256] = tot;
    u32 ttot;
    const u32 ft = block_exclusive_scan<256, u32>((c + RDX_TILE - 1) / RDX_TILE, sm, &ttot);
    seg_first[d] = ft;
    if (d == 255) { seg_first[256] = ttot; *ntiles_dev = ttot; }
}
__global__ void k_tile_table(const u32* __restrict__ seg_start, const u32* __restrict__ seg_first, const u32* __restrict__ ntiles_dev,
                             u32* __restrict__ t_start, u32* __restrict__ t_count, u16* __restrict__ t_seg) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= *ntiles_dev) return;
    u32 lo = 0, hi = 256;  // last segment with seg_first[s] <= t (segments without tiles share their successor's first tile)
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (seg_first[mid] <= t) lo = mid; else hi = mid;
    }
    const u32 st = seg_start[lo] + (t - seg_first[lo]) * RDX_TILE;
    const u32 en = seg_start[lo + 1];
    t_start[t] = st;
    t_count[t] = en - st < (u32)RDX_TILE ? en - st : (u32)RDX_TILE;
    t_seg[t] = (u16)lo;
}

// ---- tiles of the LAST LSD pass cut at GROUP boundaries (group = segment x value of the lower digits) ---------------
// After the earlier passes a segment is sorted by the lower digits, so a tile that stays inside one group holds ONE
// value of them: its records with last-pass digit d all belong to the single bucket (segment, d, lower digits). The
// bucket boundaries then follow from the last pass's own tables (k_dir_gather) and the sorted array is never re-read to
// find them. G groups, grp_start[G + 1] (record positions; [G] = n).
// Cutting costs up to one partly filled tile per group, which only pays in segments that are much longer than their
// groups are many: a COLD segment (fewer than GRP_COLD_MAX records) keeps plain tiles and its few boundaries are found
// by a scan of its records (k_boundaries_cold).
static const u32 GRP_COLD_MAX = 64 * RDX_TILE;
__device__ __forceinline__ bool seg_cold(const u32* seg_start, u32 s) { return seg_start[s + 1] - seg_start[s] < GRP_COLD_MAX; }
// records group g contributes to the tiling: a cold segment is tiled as one piece, charged to its first group
__device__ __forceinline__ u32 grp_tiled_size(u32 g, u32 G, u32 low_bits, const u32* grp_start, const u32* seg_start, u32 n_total) {
    const u32 s = g >> low_bits;
    if (seg_cold(seg_start, s)) return (g & ((1u << low_bits) - 1u)) == 0 ? seg_start[s + 1] - seg_start[s] : 0u;
    return (g + 1 < G ? grp_start[g + 1] : n_total) - grp_start[g];
}
__global__ __launch_bounds__(1024) void k_grp_table(u32 G, u32 low_bits, const u32* __restrict__ grp_start, const u32* __restrict__ seg_start, u32 n_total,
                                                    u32* __restrict__ grp_first /* G+1 */, u32* __restrict__ seg_first /* (G >> low_bits) + 1 */,
                                                    u32* __restrict__ ntiles_dev) {
    __shared__ u32 sm[1024 / 64 + 1];
    const u32 tid = threadIdx.x, per = (G + 1023u) / 1024u, g0 = tid * per < G ? tid * per : G, g1 = g0 + per < G ? g0 + per : G;
    u32 mine = 0;
    for (u32 g = g0; g < g1; ++g) mine += (grp_tiled_size(g, G, low_bits, grp_start, seg_start, n_total) + RDX_TILE - 1) / RDX_TILE;
    u32 tot;
    u32 run = block_exclusive_scan<1024, u32>(mine, sm, &tot);
    for (u32 g = g0; g < g1; ++g) {
        grp_first[g] = run;
        if ((g & ((1u << low_bits) - 1u)) == 0) seg_first[g >> low_bits] = run;
        run += (grp_tiled_size(g, G, low_bits, grp_start, seg_start, n_total) + RDX_TILE - 1) / RDX_TILE;
    }
    if (tid == 0) { grp_first[G] = tot; seg_first[G >> low_bits] = tot; *ntiles_dev = tot; }
}
__global__ void k_tile_table_grp(u32 G, u32 low_bits, const u32* __restrict__ grp_start, const u32* __restrict__ seg_start, u32 n_total,
                                 const u32* __restrict__ grp_first, const u32* __restrict__ ntiles_dev, u32* __restrict__ t_start, u32* __restrict__ t_count,
                                 u16* __restrict__ t_seg) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= *ntiles_dev) return;
    u32 lo = 0, hi = G;  // last group with grp_first[g] <= t (groups without tiles share their successor's first tile)
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (grp_first[mid] <= t) lo = mid; else hi = mid;
    }
    const u32 st = grp_start[lo] + (t - grp_first[lo]) * RDX_TILE;
    const u32 en = grp_start[lo] + grp_tiled_size(lo, G, low_bits, grp_start, seg_start, n_total);
    t_start[t] = st;
    t_count[t] = en - st < (u32)RDX_TILE ? en - st : (u32)RDX_TILE;
    t_seg[t] = (u16)(lo >> low_bits);
}
// start_dense[prefix] (EMPTY = 0xFFFFFFFF) for every prefix = (segment, last digit d, lower digits) from the last pass's
// column prefixes: the bucket starts where the group's first tile puts its digit-d records and holds what the group's
// tiles count for d. One workgroup per group, one thread per digit value.
__global__ __launch_bounds__(256) void k_dir_gather(u32 low_bits, u32 last_bits, const u32* __restrict__ grp_first, const u32* __restrict__ seg_start,
                                                    const u32* __restrict__ ntiles_dev, const u32* __restrict__ colpre, const u32* __restrict__ coltot,
                                                    const u32* __restrict__ adj, u32* __restrict__ start_dense, u32 w_lo = 0, u32 w_hi = 0xFFFFFFFFu,
                                                    const u32* __restrict__ seg_prefix = nullptr /* FINE bins: first prefix of the segment's block */) {
    // [w_lo, w_hi): the prefixes this launch owns (a receiver's group works on its window of the prefix space: start_dense is that
    // window's array, passed as `array - w_lo`; prefixes outside it belong to other groups and are not touched)
    const u32 g = blockIdx.x, d = threadIdx.x;
    if (d >= (1u << last_bits)) return;
    const u32 nt = *ntiles_dev, s = g >> low_bits, low = g & ((1u << low_bits) - 1u);
    if (seg_prefix && seg_prefix[s] == 0xFFFFFFFFu) return;  // no such segment (its rows would land on another segment's prefixes)
    const u32 prefix = (seg_prefix ? seg_prefix[s] : s << (low_bits + last_bits)) | (d << low_bits) | low;
    if (prefix < w_lo || prefix >= w_hi) return;
    // (FINE bins: the caller filled the window with EMPTY, and only rows that hold records are written — the blocks of two segments
    // never share a prefix that holds records, an empty row of one must not land on the other's)
    if (low_bits && seg_cold(seg_start, s)) {  // its tiles were not cut at the groups: k_boundaries_cold fills these in
        if (!seg_prefix) start_dense[prefix] = 0xFFFFFFFFu;
        return;
    }
    const u32 f0 = grp_first[g], f1 = grp_first[g + 1];
    const u32 p0 = f0 < nt ? colpre[(u64)f0 * 256 + d] : coltot[d];
    const u32 p1 = f1 < nt ? colpre[(u64)f1 * 256 + d] : coltot[d];
    if (p1 > p0) start_dense[prefix] = adj[s * 256 + d] + p0;
    else if (!seg_prefix) start_dense[prefix] = 0xFFFFFFFFu;
}

// bucket boundaries of the cold segments from their (few) sorted records
template <typename HiT>
__global__ __launch_bounds__(256) void k_boundaries_cold(const u64* __restrict__ lo, const HiT* __restrict__ hi, u32 SB, u32 R, const u32* __restrict__ seg_start,
                                                         u32* __restrict__ start_dense, const u32* __restrict__ seg_prefix = nullptr) {
    const u32 s = blockIdx.x;  // gridDim.y workgroups share a segment
    if (!seg_cold(seg_start, s)) return;
    const u32 a = seg_start[s], b = seg_start[s + 1];
    for (u32 i = a + blockIdx.y * blockDim.x + threadIdx.x; i < b; i += blockDim.x * gridDim.y) {
        const u32 p = get_bits(lo[i], (u64)ld_hi<HiT>(hi, i), SB, R);
        if (i == a || get_bits(lo[i - 1], (u64)ld_hi<HiT>(hi, i - 1), SB, R) != p) start_dense[(seg_prefix ? seg_prefix[s] : s << R) | p] = i;
    }
}

// ---- PREFIX_BITS > 24: the last 1 .. 4 prefix bits without a fourth pass over HBM through the scatter kernel ----------------------
// The partition passes sort 24 prefix bits (A + two LSD passes, bucket starts from the last pass's tables); a run of equal 24-bit
// prefix — a few hundred to a few thousand records, up to 2^xb real buckets interleaved in stream order — is then split by its last
// xb prefix bits by ONE workgroup that stages the run (tile by tile of SPLIT_TILE records) in LDS in its final order and writes it
// out front to back: whole cache lines, where the scatter kernel's 16-record runs leave partial ones (DESIGN_HISTORY.md §3.4: that is what a
// scatter pass costs over a copy). Stable (ballot ranking, as tile_rank): the order inside a bucket stays the stream order. A run of
// several tiles is counted first (its records are read twice; the second read comes from the L2). The kernel also writes the bucket
// starts of the run's 2^xb prefixes (EMPTY32 for the absent ones) — the fused directory of the old last pass, without candidates to
// settle. Reads `in`, writes `out` (the ping-pong partner) at the same run positions.
// Four workgroup sizes by run length — one wave (up to 512 records), two (up to 1024), four (up to 2048), eight (longer: tiles of 4096;
// only a run of more than one tile is read twice) — each launched over the list of its own runs (k_split_classify). (The two-wave class
// came last: a 900-record run in a four-wave workgroup keeps four loads per lane in flight where eight fit; cfg 3 -0.3 ms, same box.)
// Per tile: the lanes' records (wave-contiguous slices) are ranked by digit with xb ballots (stable), the per-wave counts are scanned
// across the waves, the tile is staged in LDS in its final order and leaves as one contiguous piece per digit.
static const int SPLIT_ITEMS = 8;
// runs by length class (one list per workgroup size): a launch over ALL runs whose workgroups leave when the run is not theirs costs
// more than the split itself — 1.7 M workgroups of 512 threads that only read two offsets still hold their wave slots for 2 us each
// (3 of 8 ms at cfg 3, profiles/r04_kernel_stats_cfg3.md)
struct SplitRun { u64 start; u32 len, idx; };  // one 16-byte descriptor per run (idx = its rank among the runs): a split workgroup starts from a single load
static const int SPLIT_CLASSES = 4;  // workgroups of 64 / 128 / 256 / 512 threads: runs of up to 512 / 1024 / 2048 records in one tile, longer ones in tiles of 4096
__global__ __launch_bounds__(1024) void k_split_classify(u64 nruns, const u64* __restrict__ run_start, SplitRun* __restrict__ lists /* [SPLIT_CLASSES][nruns] */,
                                                         u32* __restrict__ list_n /* SPLIT_CLASSES */) {
    __shared__ u32 s_cnt[SPLIT_CLASSES * 16];
    __shared__ u32 s_base[SPLIT_CLASSES];
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int cls = -1;
    u64 st = 0, c = 0;
    if (i < nruns) {
        st = run_start[i];
        c = run_start[i + 1] - st;
        cls = c <= 64 * SPLIT_ITEMS ? 0 : (c <= 128 * SPLIT_ITEMS ? 1 : (c <= 256 * SPLIT_ITEMS ? 2 : 3));
    }
    u32 my_rank = 0;
#pragma unroll
    for (int k = 0; k < SPLIT_CLASSES; ++k) {
        const u64 bal = __ballot(cls == k);
        if (cls == k) my_rank = mbcnt(bal);
        if (lane == 0) s_cnt[k * 16 + w] = (u32)__builtin_popcountll(bal);
    }
    __syncthreads();
    if (threadIdx.x < SPLIT_CLASSES) {
        const u32 k = threadIdx.x;
        u32 run = 0;
        for (int ww = 0; ww < 16; ++ww) { const u32 t = s_cnt[k * 16 + ww]; s_cnt[k * 16 + ww] = run; run += t; }
        s_base[k] = run ? atomicAdd(&list_n[k], run) : 0u;
    }
    __syncthreads();
    if (cls >= 0) lists[(u64)cls * nruns + s_base[cls] + s_cnt[cls * 16 + w] + my_rank] = SplitRun{st, (u32)c, (u32)i};
}
template <typename H, int THREADS>
__global__ __launch_bounds__(THREADS) void k_prefix_split(const SplitRun* __restrict__ run_list, const u64* __restrict__ in_lo,
                                                          const H* __restrict__ in_hi, u64* __restrict__ out_lo, H* __restrict__ out_hi, u32 SB, u32 xb,
                                                          u32* __restrict__ sub_start /* [nruns][16]: first record of the run's every prefix */,
                                                          u16* __restrict__ sub_mask /* [nruns]: which of them hold records */, u32* __restrict__ sub_nz /* [nruns]: how many */) {
    constexpr bool HAS = HiTraits<H>::has;
    constexpr int NW = THREADS / 64, TILE = THREADS * SPLIT_ITEMS;
    __shared__ u32 s_wcnt[NW * 16];  // per wave and digit: records of the tile seen so far, then the wave's offset inside the digit
    __shared__ u32 s_rbase[16];      // run: first position of the digit
    __shared__ u32 s_roff[16];       // run: records of the digit in earlier tiles
    __shared__ u32 s_tbase[16];      // tile: first staged slot of the digit
    // the tile in its final order: what leaves is one contiguous piece per digit (the whole tile front to back for a one-tile run).
    // (Straight from the registers every store instruction touched ~16 lines with 32 bytes each — the partial-line pieces that make the
    // scatter kernel slow, DESIGN_HISTORY.md §3.4 — and the split took as long as the pass it replaced.)
    // 16-byte records (K = 59) go straight from the registers instead: staged they take 64 KB per eight-wave workgroup, two per CU,
    // and the split loses more to occupancy than the pieces cost (cfg 4: 61.7 ms staged, 60.1 not)
    constexpr bool STAGE = !HAS;
    __shared__ u64 s_lo[STAGE ? TILE : 1];
    const SplitRun rd = run_list[blockIdx.x];
    const u64 s0 = rd.start;
    const u32 c = rd.len;
    const u32 NB = 1u << xb, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const u32 ntiles = (c + TILE - 1) / TILE;
    const u64* __restrict__ lo_r = in_lo + s0;
    const H* __restrict__ hi_r = HAS ? in_hi + s0 : in_hi;
    if (tid < 16) { s_rbase[tid] = 0; s_roff[tid] = 0; }
    __syncthreads();
    if (ntiles > 1) {  // a run of several tiles is counted first (the second read of its records comes from the L2)
        u32 cnt[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) cnt[k] = 0;
        for (u32 t = 0; t < ntiles; ++t) {
            const u32 t0 = t * TILE, n_tile = c - t0 < (u32)TILE ? c - t0 : (u32)TILE;
            u32 dg[SPLIT_ITEMS];
#pragma unroll
            for (int j = 0; j < SPLIT_ITEMS; ++j) {
                const u32 e = j * THREADS + tid;  // (order does not matter for counting: coalesced rows)
                const bool valid = e < n_tile;
                const u64 v = lo_r[t0 + (valid ? e : 0u)];
                u64 h = 0;
                if constexpr (HAS) h = (u64)hi_r[t0 + (valid ? e : 0u)];
                dg[j] = valid ? (u32)get_bits(v, h, SB, xb) : 0xFFu;
            }
#pragma unroll
            for (int j = 0; j < SPLIT_ITEMS; ++j)
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if ((u32)k < NB) cnt[k] += (u32)__builtin_popcountll(__ballot(dg[j] == (u32)k));
        }
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) if ((u32)k < NB && cnt[k]) atomicAdd(&s_rbase[k], cnt[k]);
        }
        __syncthreads();
        if (w == 0) {
            const u32 x = lane < 16 ? s_rbase[lane] : 0u;
            u32 inc = x;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { const u32 t = __shfl_up(inc, o, 64); if (lane >= (u32)o) inc += t; }
            if (lane < 16) s_rbase[lane] = inc - x;
        }
        __syncthreads();
    }
    for (u32 t = 0; t < ntiles; ++t) {
        const u32 t0 = t * TILE, n_tile = c - t0 < (u32)TILE ? c - t0 : (u32)TILE;
        const u32 R = (n_tile + THREADS - 1) / THREADS, EPW = 64 * R;  // wave-contiguous slices: (wave, round, lane) is the stream order
        if (tid < NW * 16) s_wcnt[tid] = 0;
        __syncthreads();
        u64 klo[SPLIT_ITEMS];
        u64 khi[HAS ? SPLIT_ITEMS : 1];
        u32 dp[SPLIT_ITEMS];
        u32* my = s_wcnt + w * 16;
#pragma unroll
        for (int j = 0; j < SPLIT_ITEMS; ++j) {
            const u32 e = w * EPW + j * 64 + lane;
            const bool valid = (u32)j < R && e < n_tile;
            klo[j] = lo_r[t0 + (valid ? e : 0u)];  // (non-temporal loads here: +0.5 ms at cfg 3, measured)
            if constexpr (HAS) khi[j] = (u64)hi_r[t0 + (valid ? e : 0u)];
            dp[j] = valid ? (u32)get_bits(klo[j], HAS ? khi[j] : 0ull, SB, xb) : 0xFFu;
        }
#pragma unroll
        for (int j = 0; j < SPLIT_ITEMS; ++j) {
            if ((u32)j < R) {
                const u32 d = dp[j];
                const bool valid = d != 0xFFu;
                u64 m = __ballot(valid);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const u64 bal = __ballot(valid && ((d >> b) & 1u));
                    m &= ((d >> b) & 1u) ? bal : ~bal;
                }
                const u32 lower = mbcnt(m), tot = (u32)__builtin_popcountll(m);
                const u32 old = valid ? my[d] : 0u;
                __builtin_amdgcn_wave_barrier();
                if (valid && lower == 0) my[d] = old + tot;
                __builtin_amdgcn_wave_barrier();
                dp[j] = valid ? (d << 16) | (old + lower) : 0xFFFFFFFFu;
            }
        }
        __syncthreads();
        u32 tcnt = 0;
        if (w == 0) {  // per digit: exclusive scan across the waves, the tile's digit starts; a one-tile run's are the run's
            if (lane < 16) {
                u32 runc = 0;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) { const u32 x = s_wcnt[ww * 16 + lane]; s_wcnt[ww * 16 + lane] = runc; runc += x; }
                tcnt = runc;
            }
            u32 inc = tcnt;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { const u32 x = __shfl_up(inc, o, 64); if (lane >= (u32)o) inc += x; }
            if (lane < 16) { s_tbase[lane] = inc - tcnt; if (ntiles == 1) s_rbase[lane] = inc - tcnt; }
        }
        __syncthreads();
        if constexpr (STAGE) {
#pragma unroll
            for (int j = 0; j < SPLIT_ITEMS; ++j) {
                if ((u32)j < R && dp[j] != 0xFFFFFFFFu) {
                    const u32 d = dp[j] >> 16;
                    s_lo[s_tbase[d] + my[d] + (dp[j] & 0xFFFFu)] = klo[j];
                }
            }
            __syncthreads();
            for (u32 i = tid; i < n_tile; i += THREADS) {
                const u64 v = s_lo[i];
                const u32 d = (u32)get_bits(v, 0ull, SB, xb);
                out_lo[s0 + s_rbase[d] + s_roff[d] + (i - s_tbase[d])] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < SPLIT_ITEMS; ++j) {
                if ((u32)j < R && dp[j] != 0xFFFFFFFFu) {
                    const u32 d = dp[j] >> 16;
                    const u64 dst = s0 + s_rbase[d] + s_roff[d] + my[d] + (dp[j] & 0xFFFFu);
                    out_lo[dst] = klo[j];
                    if constexpr (HAS) out_hi[dst] = (H)khi[j];
                }
            }
        }
        __syncthreads();
        if (w == 0 && lane < 16) s_roff[lane] += tcnt;
    }
    __syncthreads();
    // the run's buckets: digit d starts at s0 + rbase[d] and holds roff[d] records. Three small per-run tables instead of 2^xb entries of
    // a dense array over all 2^PREFIX_BITS prefixes: the bitvector, rank directory and bucket table then cost O(runs), not three passes
    // over a gigabyte (k_split_table)
    if (w == 0) {
        const u64 m = __ballot(lane < NB && s_roff[lane < 16 ? lane : 0] != 0);
        if (lane < 16) sub_start[(u64)rd.idx * 16 + lane] = lane < NB ? (u32)(s0 + s_rbase[lane]) : 0u;
        if (lane == 0) { sub_mask[rd.idx] = (u16)m; sub_nz[rd.idx] = (u32)__builtin_popcountll(m); }
    }
}

// bucket table and bitvector from the per-run tables of k_prefix_split: thread (run, d) of a non-empty sub-bucket writes its row at
// rank_base[run] + (non-empty sub-buckets of the run below d); one thread per run ORs the run's 2^xb bits into the bitvector (they lie
// inside one 64-bit word: 2^xb divides 64 and the run's first prefix is a multiple of 2^xb)
__global__ void k_split_table(u64 nruns, const u32* __restrict__ run_prefix, const u32* __restrict__ sub_start, const u16* __restrict__ sub_mask, const u64* __restrict__ rank_base,
                              u32 xb, u32* __restrict__ bucket_prefix, u64* __restrict__ raw_start, u64* __restrict__ bv) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 run = i >> 4;
    const u32 d = (u32)i & 15u;
    if (run >= nruns) return;
    const u32 m = sub_mask[run];
    const u64 p0 = (u64)run_prefix[run] << xb;
    if (d == 0 && m) atomicOr((unsigned long long*)&bv[p0 >> 6], (unsigned long long)m << (p0 & 63u));
    if (!((m >> d) & 1u)) return;
    const u64 r = rank_base[run] + (u64)__builtin_popcount(m & ((1u << d) - 1u));
    bucket_prefix[r] = (u32)(p0 | d);
    raw_start[r] = sub_start[run * 16 + d];
}

// ---- tiles of the first LSD pass over records that ARRIVE in pieces (receiver of the multi-GPU build) ------------------
// A piece = what one (slice, source rank) sent: its records for this rank, already sorted by the pass-A segment (segment
// = top prefix bits), cnt[piece][256] records per segment, first record at arena position pbase[piece]. Stream order
// inside a segment = piece order, so the tiles of a segment are the tiles of its pieces in piece order; a tile never
// straddles a piece. Everything behind this pass sees one contiguous array again (the pass writes to seg_start-based
// positions).
//   k_piece_tables (one workgroup, thread v = segment): seg_start / seg_first / tile count, and per (segment, piece) the
//   first tile (pfirst) and the arena position (ppos).   k_piece_tile_table: one thread per tile.
__global__ __launch_bounds__(256) void k_piece_tables(const u32* __restrict__ cnt /* [np][256] */, const u32* __restrict__ pbase /* [np] */, u32 np,
                                                      u32* __restrict__ seg_start /* 257 */, u32* __restrict__ seg_first /* 257 */, u32* __restrict__ ntiles_dev,
                                                      u32* __restrict__ pfirst /* [256][np] */, u32* __restrict__ ppos /* [256][np] */,
                                                      u32* __restrict__ seg_tot /* 256: what pass A's column totals would be */) {
    __shared__ u32 sm[256 / 64 + 1];
    const u32 v = threadIdx.x;
    u32 tot = 0, tiles = 0;
    for (u32 p = 0; p < np; ++p) {
        const u32 c = cnt[p * 256 + v];
        const u32 ex = block_exclusive_scan<256, u32>(c, sm, nullptr);  // records of the piece in earlier segments
        ppos[v * np + p] = pbase[p] + ex;
        tot += c;
        tiles += (c + RDX_TILE - 1) / RDX_TILE;
    }
    u32 all, tall;
    const u32 st = block_exclusive_scan<256, u32>(tot, sm, &all);
    const u32 ft = block_exclusive_scan<256, u32>(tiles, sm, &tall);
    seg_start[v] = st;
    seg_first[v] = ft;
    seg_tot[v] = tot;
    if (v == 255) { seg_start[256] = all; seg_first[256] = tall; *ntiles_dev = tall; }
    u32 run = ft;
    for (u32 p = 0; p < np; ++p) {
        pfirst[v * np + p] = run;
        run += (cnt[p * 256 + v] + RDX_TILE - 1) / RDX_TILE;
    }
}
__global__ void k_piece_tile_table(const u32* __restrict__ cnt, u32 np, const u32* __restrict__ seg_first, const u32* __restrict__ ntiles_dev,
                                   const u32* __restrict__ pfirst, const u32* __restrict__ ppos, u32* __restrict__ t_start, u32* __restrict__ t_count,
                                   u16* __restrict__ t_seg) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= *ntiles_dev) return;
    u32 lo = 0, hi = 256;  // last segment with seg_first[s] <= t (segments without tiles share their successor's first tile)
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (seg_first[mid] <= t) lo = mid; else hi = mid;
    }
    const u32* pf = pfirst + lo * np;
    u32 a = 0, b = np;     // last piece of the segment with pfirst <= t
    while (b - a > 1) {
        const u32 mid = (a + b) >> 1;
        if (pf[mid] <= t) a = mid; else b = mid;
    }
    const u32 off = (t - pf[a]) * RDX_TILE, c = cnt[a * 256 + lo];
    t_start[t] = ppos[lo * np + a] + off;
    t_count[t] = c - off < (u32)RDX_TILE ? c - off : (u32)RDX_TILE;
    t_seg[t] = (u16)lo;
}

}  // namespace cblx
