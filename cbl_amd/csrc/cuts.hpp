// cuts.hpp — host-side logic of the grouped receiver's CUTS (comm.hpp: sharded_insert_grouped): the key of the cut table, the group
// cuts chosen from the sampled prefix histogram, and the bin map. No HIP in here: tests/host/cut_plan_unit.cpp builds it with g++
// and checks it against the definitions (bin = top 8 prefix bits + #{cuts <= prefix}; bins of one rank consecutive; groups by mass).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
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

// ---- FINE bins (PREFIX_BITS > 24): the senders' first pass does more of the receiver's sorting --------------------------------------
// With bin = (top 8 prefix bits) + #{cuts <= prefix} a receiver is left with PREFIX_BITS - 8 bits to sort: three LSD passes at
// PREFIX_BITS = 28. But a rank of a many-GPU job owns a NARROW prefix range (necklace prefixes are dense at the bottom of the prefix space:
// the lowest eighth of the mass spans 2^20 of the 2^28 prefixes), and the bins below 128 + cuts are mostly unused. Here the bins are the
// intervals between up to 253 cuts, bin(p) = #{cuts <= p}: the rank bounds and group cuts as before, plus STRUCTURAL cuts at multiples of
// 2^lmax everywhere (so that every bin lies inside an aligned block of 2^lmax prefixes: the bin implies the prefix bits from lmax up,
// which is what lets 65..72-bit words travel without their hi byte) and at multiples of 2^16 inside the groups that are narrow enough
// for the budget — narrowest first: groups hold about equal mass, so those are the cheapest per word. A group all of whose bins lie
// inside aligned blocks of 2^16 prefixes needs 16 bits sorted behind the first pass — two LSD passes instead of three; the others sort
// 24 bits in three passes of 8. A bin's LEVEL = the smallest L with (first >> L) == (last >> L); `sort_bits` of a group = 16 or 24.
// The lookup table is linear: key = prefix >> ksh (8192 cells), entry = {cuts at or below the cell's first prefix, the one cut inside
// the cell or ~0}; structural cuts are multiples of 2^16 >= the cell width, so only rank bounds and group cuts can fall inside a cell,
// and a list with two of them in one cell is refused (a cell is 1/8192 of the prefix space, a group 1/(W G) of the MASS).
static const u32 FINE_LEVEL = 16, FINE_MAX_CUTS = 253;
// the table the kernels stage in LDS: one u32 per cell of 2^(PB - 12) prefixes over [0, 2^(PB-1)) — low byte = cuts at or below the
// cell's first prefix, upper 24 bits = offset of the one cut inside the cell (FINE_NO_CUT: none). 8 KB.
static const u32 FINE_CELLS = 2048, FINE_NO_CUT = 0xFFFFFFu;
struct FinePlan : CutPlan {
    u32 ksh = 0;                         // key of `tab32` = prefix >> ksh
    std::vector<u32> tab32;              // FINE_CELLS entries (see above); `tab` stays empty
    std::vector<u32> first;              // per interval (= bin, except the all-ones bin 255): its first prefix
    std::vector<u32> level;              // per interval
    u32 sort_bits[CUT_MAX_DEST][16];     // per (rank, group): prefix bits the receiver sorts behind the first pass (16 or 24)
    u32 seg_of[256];                     // bin -> segment number inside its group (what the receiver's tables are indexed by)
};
inline u32 fine_bin(const FinePlan& M, u32 PB, u32 p) {  // (host restatement of DigitCut in its linear-key mode)
    if ((p >> (PB - 8)) >= 255u) return 255u;
    const u32 k = p >> M.ksh, c = M.tab32[k < FINE_CELLS ? k : FINE_CELLS - 1];
    const u32 b = (c & 255u) + ((p & ((1u << M.ksh) - 1u)) >= (c >> 8) ? 1u : 0u);
    return b < 254u ? b : 254u;
}
// `lmax`: every bin must lie inside an aligned block of 2^lmax prefixes (<= 24: three passes of 8 bits sort that much)
// `bottom_first`: the windows are filled from the lowest prefixes up instead of narrowest first (one rank, whose "groups" are not
// equal shares of the mass: comm.hpp insert_device_fine)
inline FinePlan make_fine_plan(u32 PB, u32 lmax, const u32* bounds, u32 W, const std::vector<u32>& gcuts, bool bottom_first = false) {
    FinePlan M;
    for (u32 i = 0; i < 256; ++i) M.v_of[i] = M.iv_of[i] = M.seg_of[i] = 0xFFFFFFFFu;
    for (u32 d = 0; d < CUT_MAX_DEST; ++d) for (u32 g = 0; g < 16; ++g) M.sort_bits[d][g] = 24;
    if (PB <= 24 || PB > 28 || lmax <= FINE_LEVEL || lmax > 24 || W > CUT_MAX_DEST) return M;  // (PB <= 28: a cell of the table is at most 2^16 prefixes wide)
    const u32 RB8 = PB - 8;
    const u64 half = 1ull << (PB - 1);  // necklace prefixes lie below it, except the all-ones word (bin 255)
    // forced cuts: rank bounds (kind 1), group cuts (kind 2)
    std::vector<std::pair<u32, u32>> cuts;  // (value, kind); kind 0 = structural
    for (u32 i = 0; i + 1 < W; ++i) {
        if (bounds[i] == 0 || (i && bounds[i] <= bounds[i - 1]) || (u64)bounds[i] > (255ull << RB8)) return M;
        cuts.push_back({bounds[i], 1u});
    }
    for (u32 g : gcuts) cuts.push_back({g, 2u});
    std::sort(cuts.begin(), cuts.end());
    for (size_t i = 1; i < cuts.size(); ++i) if (cuts[i].first == cuts[i - 1].first) return M;
    if (cuts.empty() || cuts.front().first == 0) return M;
    const size_t nforced = cuts.size();
    auto has = [&](u64 v) { return std::binary_search(cuts.begin(), cuts.end(), std::make_pair((u32)v, 0u), [](const std::pair<u32, u32>& a, const std::pair<u32, u32>& b) { return a.first < b.first; }); };
    // structural cuts at the multiples of 2^lmax up to 2^(PB-1): everything above is one never-used bin in front of the all-ones one
    {
        std::vector<std::pair<u32, u32>> add;
        for (u64 v = 1ull << lmax; v <= half; v += 1ull << lmax) if (!has(v)) add.push_back({(u32)v, 0u});
        cuts.insert(cuts.end(), add.begin(), add.end());
        std::sort(cuts.begin(), cuts.end());
    }
    if (cuts.size() > FINE_MAX_CUTS) return M;
    // the groups' windows (between consecutive forced cuts), narrowest first: multiples of 2^16 inside them while the budget lasts
    {
        std::vector<u64> edge{0};
        for (const auto& c : cuts) if (c.second) edge.push_back(c.first);
        edge.push_back(half);
        struct Win { u64 need, lo, hi; };
        std::vector<Win> wins;
        for (size_t i = 0; i + 1 < edge.size(); ++i) {
            const u64 lo = edge[i], hi = std::min<u64>(edge[i + 1], half);
            if (hi <= lo) continue;
            u64 need = 0;
            for (u64 v = ((lo >> FINE_LEVEL) + 1) << FINE_LEVEL; v < hi; v += 1ull << FINE_LEVEL) if (!has(v)) ++need;
            wins.push_back(Win{need, lo, hi});
        }
        if (!bottom_first) std::sort(wins.begin(), wins.end(), [](const Win& a, const Win& b) { return a.need != b.need ? a.need < b.need : a.lo < b.lo; });
        size_t total = cuts.size();
        std::vector<std::pair<u32, u32>> add;
        for (const Win& w : wins) {
            if (total + w.need > FINE_MAX_CUTS) { if (bottom_first) continue; break; }
            for (u64 v = ((w.lo >> FINE_LEVEL) + 1) << FINE_LEVEL; v < w.hi; v += 1ull << FINE_LEVEL) if (!has(v)) add.push_back({(u32)v, 0u});
            total += w.need;
        }
        cuts.insert(cuts.end(), add.begin(), add.end());
        std::sort(cuts.begin(), cuts.end());
    }
    (void)nforced;
    const u32 nc = (u32)cuts.size();
    for (const auto& c : cuts) M.cuts.push_back(c.first);
    // intervals: owner, group, first prefix, level
    M.dest_of.resize(nc + 1); M.grp_of.resize(nc + 1); M.first.resize(nc + 1); M.level.resize(nc + 1);
    for (u32 d = 0; d < W; ++d) M.ngroups[d] = 0;
    for (u32 i = 0; i <= nc; ++i) {
        const u32 first = i ? M.cuts[i - 1] : 0u;
        const u64 end = i < nc ? M.cuts[i] : 1ull << PB;
        u32 d = 0;
        for (u32 j = 0; j + 1 < W; ++j) d += bounds[j] <= first ? 1u : 0u;
        M.dest_of[i] = d;
        M.grp_of[i] = !i || M.dest_of[i - 1] != d ? 0u : M.grp_of[i - 1] + (cuts[i - 1].second == 2u ? 1u : 0u);
        if (M.grp_of[i] >= 16) return M;
        M.ngroups[d] = std::max(M.ngroups[d], M.grp_of[i] + 1);
        M.first[i] = first;
        u32 L = 0;
        if (first < half) { const u64 last = std::min<u64>(end, half) - 1; while ((first >> L) != (last >> L)) ++L; }  // (the bin above 2^(PB-1) holds nothing)
        M.level[i] = L;
        if (L > lmax) return M;  // (cannot happen: the multiples of 2^lmax are all cuts)
    }
    for (u32 d = 0; d < W; ++d) for (u32 g = 0; g < 16; ++g) M.sort_bits[d][g] = FINE_LEVEL;
    for (u32 i = 0; i <= nc; ++i) if (M.level[i] > FINE_LEVEL) M.sort_bits[M.dest_of[i]][M.grp_of[i]] = 24;
    // the all-ones word rides in bin 255 of the last interval's (rank, group), whatever that group's other bins look like
    // bins: interval i is bin i; 255 = the all-ones segment
    if (nc + 1 > 254) return M;
    for (u32 i = 0; i <= nc; ++i) { M.v_of[i] = i; M.iv_of[i] = i; }
    M.v_of[255] = 255; M.iv_of[255] = nc;
    for (u32 d = 0; d <= W; ++d) M.bin_lo[d] = 256;
    for (int b = 255; b >= 0; --b) if (M.iv_of[b] != 0xFFFFFFFFu) M.bin_lo[M.dest_of[M.iv_of[b]]] = (u32)b;
    M.bin_lo[W] = 256;
    for (int d = (int)W - 1; d >= 0; --d) if (M.bin_lo[d] == 256) M.bin_lo[d] = M.bin_lo[d + 1];
    // segment numbers inside a group: 0, 1, ... in bin order; the all-ones bin is segment 255 of its group
    {
        u32 run = 0;
        for (u32 i = 0; i <= nc; ++i) {
            if (i && (M.dest_of[i] != M.dest_of[i - 1] || M.grp_of[i] != M.grp_of[i - 1])) run = 0;
            M.seg_of[i] = run++;
            if (run > 254) return M;
        }
        M.seg_of[255] = 255;
    }
    // the table
    M.ksh = PB - 12;  // FINE_CELLS cells below 2^(PB-1)
    M.tab32.assign(FINE_CELLS, FINE_NO_CUT << 8);
    for (u32 k = 0; k < FINE_CELLS; ++k) M.tab32[k] |= (u32)(std::upper_bound(M.cuts.begin(), M.cuts.end(), (u32)((u64)k << M.ksh)) - M.cuts.begin());
    for (u32 cv : M.cuts) {
        if ((cv & ((1u << M.ksh) - 1u)) == 0 || cv >= half) continue;  // on a cell's first prefix: counted in its base (above 2^(PB-1): no necklace prefix)
        u32& c = M.tab32[cv >> M.ksh];
        if ((c >> 8) != FINE_NO_CUT) return M;  // two cuts inside one cell
        c = (c & 255u) | ((cv & ((1u << M.ksh) - 1u)) << 8);
    }
    M.ok = true;
    return M;
}
// PREFIX_BITS > 24 (FINE bins, cuts.hpp): the budget of 253 cuts covers about the lowest 214 blocks of 2^16 prefixes at 8 ranks x 4 groups — the
// ranges below sort 16 bits behind the first pass (two passes), the sparse tail above 24 (three passes, and a directory over a wide window).
// Rehearsed at cfg 3 (profiles/r05_wire_emulated.md): with equal words per rank, ranks 0 - 6 take 43 - 44 ms without a wire (47 - 50 at 55 GB/s
// per link) and rank 7 — the tail — 51 (52.5). The quantiles (rank bounds and group cuts) are therefore taken over the histogram with the
// tail's cells weighted, so that the ranks take equal TIME rather than equal words: 1.20 levels rank 0 and rank 7 within a millisecond of
// each other at 55 GB/s per link and without a wire (47.6 / 46.5 and 45.4 / 46.2 ms with 8-byte records on the wire; 1.12: 46.4 / 48.3,
// 1.28: 47.6 / 45.6 — box-to-box noise is a millisecond). (A heuristic: the factor is cfg 3's; any bounds are correct, only the balance
// depends on it. CBLX_FINE_TAIL_WEIGHT overrides it in percent, 100 = off.)
inline u64 fine_tail_weight_pct() {  // (job-wide like every switch that shapes the plan: comm.hpp checks that the ranks agree)
    const char* we = std::getenv("CBLX_FINE_TAIL_WEIGHT");
    return we ? std::strtoull(we, nullptr, 10) : 120;
}
inline void weigh_tail_for_fine_bins(std::vector<u64>& hist, u32 PB, u32 W, u32 G, u32 hb) {
    const char* fe = std::getenv("CBLX_FINE_BINS");
    if (PB <= 24 || PB > 28 || W < 2 || (fe && fe[0] == '0') || hb + FINE_LEVEL < PB) return;  // (cells must not be wider than 2^16 prefixes)
    const u64 pct = fine_tail_weight_pct();
    if (pct == 100 || pct == 0) return;
    const u64 forced = (u64)W * G - 1, fixed = 8;  // rank bounds + group cuts, the multiples of 2^lmax
    if (forced + fixed + 16 >= FINE_MAX_CUTS) return;
    const u64 x = (FINE_MAX_CUTS - forced - fixed) << FINE_LEVEL;   // where the blocks of 2^16 prefixes end, about
    const size_t first = (size_t)(x >> (PB - hb));
    for (size_t i = 0; i < hist.size(); ++i) hist[i] = hist[i] * (i >= first ? pct : 100);
}
}  // namespace cblx
