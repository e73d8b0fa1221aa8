// onesweep_experiment.hpp — NOT part of libcblx. The single-read ("onesweep") form of a partition pass, kept for the
// dev harness (tools/dev_radix_bench.cpp): one up-front histogram of all digits + a decoupled look-back across tiles.
// Bit-identical to the two-kernel pass, but not faster on MI355X (DESIGN_HISTORY.md §3.4): an agent-scope look-back hop across
// the non-coherent per-XCD L2s costs 1-3 us and tiles arrive at ~35/us, so the walk is latency-unstable.
#pragma once
#include "kernels_radix.hpp"

namespace cblx {

// ------------------------------------------------------------------------------------------------
// Onesweep form of the same pass: one read + one write of every record per pass. The per-tile histogram pass and the
// device-wide scan are replaced by (a) ONE up-front histogram of every pass's digit (order independent) and (b) a
// decoupled look-back across tiles inside the scatter kernel.
//
// Inter-workgroup protocol (MI355X: per-XCD L2s are not coherent, a CU's L1 is never refreshed by other CUs):
// every shared word is an 8-byte {epoch, flag, value} granule written by ONE relaxed agent-scope store and polled
// with relaxed agent-scope loads - the data is the flag, so no fence is needed (cdna_hip_programming.md §6 G16, R2).
// Tiles take their index from an atomic ticket, so every predecessor of a tile is already resident and will publish
// without waiting on anything later: the look-back cannot deadlock. Spins are bounded; a timeout raises *err.
static const u32 MAX_PASSES = 4;
static const u64 OS_FLAG_AGG = 1ull << 40, OS_FLAG_INCL = 2ull << 40, OS_VALUE_MASK = (1ull << 40) - 1;
__device__ __forceinline__ u64 os_pack(u32 epoch, u64 flag, u64 value) { return ((u64)epoch << 48) | flag | value; }

// histograms of all LSD digits in one read of the records: ghist[pass * 256 + digit]
template <typename HiT>
__global__ __launch_bounds__(512) void k_digit_hist(const u64* __restrict__ lo, const HiT* __restrict__ hi, u64 n, u32 SB, u32 PB,
                                                    u32 npass, unsigned long long* __restrict__ ghist) {
    __shared__ u32 s_h[MAX_PASSES * 256];
    for (u32 i = threadIdx.x; i < MAX_PASSES * 256; i += blockDim.x) s_h[i] = 0;
    __syncthreads();
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 a = lo[i], b = ld_hi<HiT>(hi, i);
        const u32 prefix = get_bits(a, b, SB, PB);
        for (u32 p = 0; p < npass; ++p) atomicAdd(&s_h[p * 256 + ((prefix >> (8 * p)) & 255u)], 1u);
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < npass * 256; i += blockDim.x)
        if (s_h[i]) atomicAdd(&ghist[i], (unsigned long long)s_h[i]);
}

template <typename HiT, typename DigitFn>
__global__ __launch_bounds__(RDX_THREADS) void k_onesweep(const u64* __restrict__ lo, const HiT* __restrict__ hi, u64 n, DigitFn dfn,
                                                          const unsigned long long* __restrict__ ghist /* this pass, 256 */,
                                                          u32* __restrict__ ticket /* 8 counters, 64 B apart */, u32 ntiles,
                                                          u64* __restrict__ status /* [ntiles][256] */,
                                                          u32 epoch, u64* __restrict__ out_lo, HiT* __restrict__ out_hi,
                                                          u32* __restrict__ err, u32 dbg) {
    __shared__ u64 s_lo[RDX_TILE];
    __shared__ typename std::conditional<HiTraits<HiT>::has, HiT, u8>::type s_hi[HiTraits<HiT>::has ? RDX_TILE : 1];
    __shared__ u32 s_wcnt[(RDX_THREADS / 64) * 256];
    __shared__ u32 s_dbase[256];
    __shared__ u64 s_gbase[256];
    __shared__ u32 s_scan[RDX_THREADS / 64 + 1];
    __shared__ u64 s_scan64[RDX_THREADS / 64 + 1];
    __shared__ u32 s_tile;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    // Tile index from a ticket. One counter word serves only ~88 tickets/us on this part, so there are 8 of them,
    // picked by the XCD the workgroup runs on (speed only): counter x hands out tiles x, x+8, x+16, ... Liveness does
    // not depend on the choice of x: the lowest unstarted tile only ever waits for started ones. When a counter's
    // class is exhausted (XCDs drift apart by a few tiles at the end) the workgroup takes from the next class.
    if (tid == 0) {
        u32 x = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;  // HW_REG_XCC_ID[3:0]
        u32 t = 0xFFFFFFFFu;
        for (u32 k = 0; k < 8; ++k) {
            const u32 cls = (x + k) & 7u;
            const u32 n_cls = (ntiles + 7u - cls) >> 3;  // tiles with index = cls (mod 8)
            const u32 got = atomicAdd(ticket + cls * 16, 1u);  // counters 64 B apart
            if (got < n_cls) { t = got * 8u + cls; break; }
        }
        s_tile = t;
    }
    __syncthreads();
    const u32 tile = s_tile;
    if (tile == 0xFFFFFFFFu) return;  // cannot happen: #workgroups == #tiles
    const u64 tbase = (u64)tile * RDX_TILE;
    const u32 n_tile = (u32)((n - tbase) < (u64)RDX_TILE ? (n - tbase) : (u64)RDX_TILE);

    u64 klo[RDX_ITEMS];
    typename std::conditional<std::is_same<HiT, u64>::value, u64, u32>::type khi[RDX_ITEMS];
    u32 digit[RDX_ITEMS], pos[RDX_ITEMS];
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) {
        const u32 e = w * (64 * RDX_ITEMS) + j * 64 + lane;
        const bool valid = e < n_tile;
        const u64 idx = valid ? tbase + e : tbase;
        klo[j] = lo[idx];
        khi[j] = ld_hi<HiT>(hi, idx);
        digit[j] = valid ? dfn(klo[j], (u64)khi[j]) : 255u;
    }
    tile_rank<RDX_THREADS, RDX_ITEMS>(digit, pos, s_wcnt, s_dbase, s_scan, RDX_ITEMS);
    // global start of every digit (exclusive scan of this pass's histogram) + look-back over earlier tiles
    u64 gh = tid < 256 ? (u64)ghist[tid] : 0ull;
    const u64 dbase_g = block_exclusive_scan<RDX_THREADS, u64>(gh, s_scan64, nullptr);
    if (tid < 256) {
        const u32 nxt = tid == 255 ? n_tile : s_dbase[tid + 1];
        const u64 cnt = tid == 255 ? (u64)(n_tile > s_dbase[255] ? n_tile - s_dbase[255] : 0) : (u64)(nxt - s_dbase[tid]);
        u64* mine = status + (u64)tile * 256 + tid;
        u64 excl = 0;
        if (dbg & 1) {
        } else if (tile == 0) {
            __hip_atomic_store(mine, os_pack(epoch, OS_FLAG_INCL, cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(mine, os_pack(epoch, OS_FLAG_AGG, cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // walk back over the predecessors, LB of them per step with independent loads (an agent-scope hop costs
            // ~1-3 us on this part, so a serial walk would dominate the pass)
            constexpr int LB = 8;
            int p = (int)tile - 1;
            u32 spins = 0;
            bool done = (dbg & 2) != 0;
            while (!done && p >= 0) {
                u64 v[LB];
#pragma unroll
                for (int k = 0; k < LB; ++k) {
                    const int q = p - k;
                    v[k] = q >= 0 ? __hip_atomic_load(status + (u64)q * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                  : os_pack(epoch, OS_FLAG_INCL, 0);
                }
                int used = 0;
#pragma unroll
                for (int k = 0; k < LB; ++k) {
                    if (done || used != k) continue;
                    const u64 x = v[k];
                    if ((u32)(x >> 48) != epoch || (x & (OS_FLAG_AGG | OS_FLAG_INCL)) == 0) continue;  // not published yet
                    excl += x & OS_VALUE_MASK;
                    ++used;
                    if (x & OS_FLAG_INCL) done = true;
                }
                p -= used;
                if (!done && used < LB) {
                    if (++spins > (1u << 20)) { atomicExch(err, 1u); break; }
                    if ((spins & 255u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __hip_atomic_store(mine, os_pack(epoch, OS_FLAG_INCL, excl + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_gbase[tid] = dbase_g + excl - s_dbase[tid];
    }
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) {
        s_lo[pos[j]] = klo[j];
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
