// cuts.hpp — host-side logic of the grouped receiver's CUTS (comm.hpp: sharded_insert_grouped): the key of the cut table, the group
// cuts chosen from the sampled prefix histogram, and the bin map. No HIP in here: tests/host/cut_plan_unit.cpp builds it with g++
// and checks it against the definitions (bin = top 8 prefix bits + #{cuts <= prefix}; bins of one rank consecutive; groups by mass).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#if !defined(__HIPCC__) && !defined(__host__)
#define __host__
#define __device__
#endif
#ifndef __forceinline__
#define CBLX_CUTS_INLINE inline
#else
#define CBLX_CUTS_INLINE __forceinline__
#endif

namespace cblx {

typedef uint32_t u32;
typedef uint64_t u64;
static const u32 CUT_MAX_DEST = 16;

struct CutCell { u32 cut, base; };
static const u32 CUT_KEYS = 64 + 26 * 32;
__host__ __device__ CBLX_CUTS_INLINE u32 cut_key(u32 p) {
    if (p < 64u) return p;
    const u32 e = 31u - (u32)__builtin_clz(p);
    return 64u + ((e - 6u) << 5) + ((p >> (e - 5u)) & 31u);
}
__host__ __device__ CBLX_CUTS_INLINE u32 cut_key_first(u32 k) {  // smallest prefix with that key
    if (k < 64u) return k;
    const u32 e = 6u + ((k - 64u) >> 5), m = (k - 64u) & 31u;
    return (32u + m) << (e - 5u);
}

// The cuts of the grouped receiver: inside every rank's range [bounds[d-1], bounds[d]) up to G - 1 more prefix values that split
// the range's sampled mass evenly — multiples of 64 (two groups of one rank never share a bitvector word), strictly inside the
// range, ascending. Fewer than G - 1 where the histogram has no room for them (a cell of it is 2^(PB-16) prefixes wide).
inline std::vector<u32> choose_group_cuts(const std::vector<u64>& hist, const u32* bounds, u32 W, u32 G, u32 PB) {
    std::vector<u32> cuts;
    const size_t nh = hist.size();
    u32 hb = 0;
    while (((size_t)1 << hb) < nh) ++hb;
    const int shift = (int)PB - (int)hb;
    if (shift < 0 || G < 2) return cuts;
    std::vector<double> cum(nh + 1, 0.0);
    for (size_t i = 0; i < nh; ++i) cum[i + 1] = cum[i] + (double)hist[i];
    auto mass_below = [&](u64 prefix) {  // sampled words with a smaller prefix (linear inside a cell)
        const u64 cell = prefix >> shift;
        if (cell >= nh) return cum[nh];
        return cum[cell] + (double)hist[cell] * (double)(prefix - (cell << shift)) / (double)(1ull << shift);
    };
    for (u32 d = 0; d < W; ++d) {
        const u64 lo = d ? bounds[d - 1] : 0, hi = d + 1 < W ? bounds[d] : 1ull << PB;
        if (hi <= lo) continue;
        const double m0 = mass_below(lo), m1 = mass_below(hi);
        u64 last = lo;
        for (u32 j = 1; j < G; ++j) {
            const double want = m0 + (m1 - m0) * j / G;
            u64 cell = (u64)(std::upper_bound(cum.begin(), cum.end(), want) - cum.begin());  // first cell boundary with more mass below it
            cell = std::min<u64>(std::max<u64>(cell, 1), nh);
            u64 v = ((cell << shift) + 63) & ~63ull;
            if (v <= last || v >= hi || v <= lo) continue;
            cuts.push_back((u32)v);
            last = v;
        }
    }
    return cuts;
}


struct CutPlan {
    bool ok = false;
    std::vector<u32> cuts;              // ascending, distinct, non-zero: rank bounds and group cuts
    std::vector<u32> dest_of, grp_of;   // per interval [cuts[i-1], cuts[i]): owner rank, group inside the owner's range
    std::vector<CutCell> tab;           // DigitCut's table
    u32 v_of[256], iv_of[256];          // bin -> pass-A segment, interval (0xFFFFFFFF: no such bin)
    u32 bin_lo[CUT_MAX_DEST + 1];           // first bin of every rank (bins of one rank are consecutive)
    u32 ngroups[CUT_MAX_DEST];              // groups of every rank
};
inline CutPlan make_cut_plan(u32 PB, const u32* bounds, u32 W, const std::vector<u32>& gcuts) {
    CutPlan M;
    const u32 RB = PB - 8;
    for (u32 i = 0; i < 256; ++i) M.v_of[i] = M.iv_of[i] = 0xFFFFFFFFu;
    for (u32 i = 0; i + 1 < W; ++i) {
        if (bounds[i] == 0 || (i && bounds[i] <= bounds[i - 1]) || (u64)bounds[i] > (255ull << RB)) return M;  // an empty range, or the all-ones segment cut
        M.cuts.push_back(bounds[i]);
    }
    for (u32 g : gcuts) M.cuts.push_back(g);
    std::sort(M.cuts.begin(), M.cuts.end());
    for (size_t i = 1; i < M.cuts.size(); ++i) if (M.cuts[i] == M.cuts[i - 1]) return M;
    if (M.cuts.size() > 120 || M.cuts.empty() || (u64)M.cuts.back() > (255ull << RB)) return M;
    const u32 nc = (u32)M.cuts.size();
    // the table: per key the cuts at or below the key's first prefix, and the one cut inside the key's range
    M.tab.assign(CUT_KEYS, CutCell{0xFFFFFFFFu, 0u});
    for (u32 k = 0; k < CUT_KEYS; ++k) {
        const u32 first = cut_key_first(k);
        M.tab[k].base = (u32)(std::upper_bound(M.cuts.begin(), M.cuts.end(), first) - M.cuts.begin());
    }
    for (u32 cv : M.cuts) {
        const u32 k = cut_key(cv);
        if (cv == cut_key_first(k)) continue;  // counted in the key's base
        if (M.tab[k].cut != 0xFFFFFFFFu) return M;  // two cuts inside one cell of the table
        M.tab[k].cut = cv;
    }
    auto cnt = [&](u64 p) { return (u32)(std::upper_bound(M.cuts.begin(), M.cuts.end(), (u32)p) - M.cuts.begin()); };
    M.dest_of.resize(nc + 1);
    M.grp_of.resize(nc + 1);
    for (u32 i = 0; i <= nc; ++i) {
        const u32 first = i ? M.cuts[i - 1] : 0u;
        u32 d = 0;
        for (u32 j = 0; j + 1 < W; ++j) d += bounds[j] <= first ? 1u : 0u;
        M.dest_of[i] = d;
        M.grp_of[i] = i && M.dest_of[i - 1] == d ? M.grp_of[i - 1] + 1 : 0u;
    }
    for (u32 d = 0; d < W; ++d) M.ngroups[d] = 0;
    for (u32 i = 0; i <= nc; ++i) M.ngroups[M.dest_of[i]] = std::max(M.ngroups[M.dest_of[i]], M.grp_of[i] + 1);
    for (u32 v = 0; v < 128; ++v)
        for (u32 k = cnt((u64)v << RB); k <= cnt((((u64)v + 1) << RB) - 1); ++k) {
            if (v + k >= 254) return M;
            M.v_of[v + k] = v;
            M.iv_of[v + k] = k;
        }
    M.v_of[255] = 255; M.iv_of[255] = nc;
    // bins of one rank are consecutive: first bin per rank (a rank without any bin gets an empty range)
    for (u32 d = 0; d <= W; ++d) M.bin_lo[d] = 256;
    for (int b = 255; b >= 0; --b) if (M.iv_of[b] != 0xFFFFFFFFu) M.bin_lo[M.dest_of[M.iv_of[b]]] = (u32)b;
    M.bin_lo[W] = 256;
    for (int d = (int)W - 1; d >= 0; --d) if (M.bin_lo[d] == 256) M.bin_lo[d] = M.bin_lo[d + 1];
    M.ok = true;
    return M;
}
}  // namespace cblx
