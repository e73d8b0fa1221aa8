// Host-side unit check of cbl_amd/csrc/cuts.hpp (no HIP): the key of the cut table, the group cuts, the bin map of the grouped
// receiver — against the definitions: bin(p) = (p >> (PB - 8)) + #{cuts <= p} computed by plain counting; the table lookup DigitCut
// does on the device (base of the key + one compare) gives the same count for every prefix tried; bins of one rank are consecutive,
// ranks and groups ascend with the prefix, a segment value occurs in at most one bin of a group; group cuts are multiples of 64,
// strictly inside their rank's range, and split its sampled mass about evenly.
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <random>

#include "../../cbl_amd/csrc/cuts.hpp"

using namespace cblx;

static long bad = 0;
#define CHECK(c, ...) do { if (!(c)) { ++bad; if (bad < 20) { printf(__VA_ARGS__); printf("\n"); } } } while (0)

int main() {
    std::mt19937_64 rnd(12345);
    // the key: monotone, cut_key_first is its smallest member
    u32 prev = 0;
    for (u64 p = 0; p < (1ull << 28); p += (p < 70000 ? 1 : 1 + rnd() % 997)) {
        const u32 k = cut_key((u32)p);
        CHECK(k >= prev && k < CUT_KEYS, "key not monotone at %llu", (unsigned long long)p);
        CHECK(cut_key_first(k) <= p && cut_key(cut_key_first(k)) == k, "cut_key_first(%u) wrong at %llu", k, (unsigned long long)p);
        if (cut_key_first(k) > 0) CHECK(cut_key(cut_key_first(k) - 1) == k - 1, "key %u does not start at its first prefix", k);
        prev = k;
    }
    long plans = 0, refused = 0;
    for (int trial = 0; trial < 400; ++trial) {
        const u32 PB = 9 + rnd() % 20, W = 2 + rnd() % 7, G = 2 + rnd() % 13, RB = PB - 8;
        // a histogram shaped like necklace prefixes: mass concentrated at small values, over 2^min(16, PB) cells
        const u32 hb = PB < 16 ? PB : 16;
        std::vector<u64> hist((size_t)1 << hb, 0);
        for (int i = 0; i < 200000; ++i) {
            const double u = (double)(rnd() >> 11) / 9007199254740992.0;
            const double x = (0.002 + u * u * u) * 0.4;  // skewed towards small values (no atom in the first cell)
            hist[(size_t)(x * hist.size())]++;
        }
        // rank bounds = quantiles (as choose_bounds does)
        std::vector<u32> bounds;
        {
            u64 tot = 0, run = 0;
            for (u64 h : hist) tot += h;
            size_t cell = 0;
            for (u32 d = 1; d < W; ++d) {
                while (cell < hist.size() && run + hist[cell] < tot * d / W) run += hist[cell++];
                bounds.push_back((u32)std::min<u64>(((u64)cell + 1) << (PB - hb), (1ull << PB) - 1));
            }
            for (size_t i = 1; i < bounds.size(); ++i) if (bounds[i] < bounds[i - 1]) bounds[i] = bounds[i - 1];
        }
        const std::vector<u32> gc = choose_group_cuts(hist, bounds.data(), W, G, PB);
        for (size_t i = 0; i < gc.size(); ++i) {
            CHECK((gc[i] & 63u) == 0, "group cut %u is not a multiple of 64", gc[i]);
            if (i) CHECK(gc[i] > gc[i - 1], "group cuts not ascending");
            bool is_bound = false;
            for (u32 b : bounds) is_bound |= b == gc[i];
            CHECK(!is_bound, "group cut %u equals a rank bound", gc[i]);
        }
        const CutPlan M = make_cut_plan(PB, bounds.data(), W, gc);
        if (!M.ok) { ++refused; continue; }  // coinciding bounds, two cuts in one cell of the table, ...: the caller takes the ungrouped path
        ++plans;
        auto cnt = [&](u32 p) { u32 n = 0; for (u32 cv : M.cuts) n += cv <= p ? 1u : 0u; return n; };
        auto dest = [&](u32 p) { u32 d = 0; for (u32 b : bounds) d += b <= p ? 1u : 0u; return d; };
        u32 last_bin = 0, last_d = 0, last_g = 0;
        bool first = true;
        for (int i = 0; i < 30000; ++i) {
            u32 p;
            const int kind = i % 4;
            if (kind == 0) p = (u32)(rnd() % (1ull << PB));
            else if (kind == 1) p = (u32)((rnd() % (1ull << PB)) >> (rnd() % PB));  // small values, where the mass and the cuts are
            else { const u32 cv = M.cuts[rnd() % M.cuts.size()]; p = cv - 1 + (u32)(rnd() % 3); if (p >= (1u << PB)) p = cv; }
            if ((p >> RB) >= 128) continue;  // (necklace prefixes never have their top bit set, except the all-ones word)
            const CutCell cc = M.tab[cut_key(p)];
            const u32 fast = cc.base + (p >= cc.cut ? 1u : 0u);
            CHECK(fast == cnt(p), "PB %u: table count %u != %u at prefix %u", PB, fast, cnt(p), p);
            const u32 bin = (p >> RB) + fast;
            CHECK(bin < 254 && M.v_of[bin] == (p >> RB) && M.iv_of[bin] == fast, "bin map wrong at prefix %u (bin %u)", p, bin);
            const u32 d = M.dest_of[fast], g = M.grp_of[fast];
            CHECK(d == dest(p), "owner of prefix %u: %u != %u", p, d, dest(p));
            CHECK(bin >= M.bin_lo[d] && bin < M.bin_lo[d + 1], "bin %u of rank %u outside [%u, %u)", bin, d, M.bin_lo[d], M.bin_lo[d + 1]);
            CHECK(g < M.ngroups[d], "group %u of rank %u (has %u)", g, d, M.ngroups[d]);
            (void)first; (void)last_bin; (void)last_d; (void)last_g;
        }
        // a segment value occurs in at most one bin of a (rank, group); ranks / groups ascend with the bin
        for (u32 d = 0; d < W; ++d)
            for (u32 b1 = M.bin_lo[d]; b1 < M.bin_lo[d + 1]; ++b1)
                for (u32 b2 = b1 + 1; b2 < M.bin_lo[d + 1]; ++b2) {
                    if (M.iv_of[b1] == 0xFFFFFFFFu || M.iv_of[b2] == 0xFFFFFFFFu) continue;
                    CHECK(M.dest_of[M.iv_of[b1]] == d && M.dest_of[M.iv_of[b2]] == d, "bins of rank %u hold another rank's interval", d);
                    CHECK(M.grp_of[M.iv_of[b1]] <= M.grp_of[M.iv_of[b2]], "groups do not ascend with the bin");
                    if (M.grp_of[M.iv_of[b1]] == M.grp_of[M.iv_of[b2]]) CHECK(M.v_of[b1] != M.v_of[b2], "segment %u twice in one group", M.v_of[b1]);
                }
        // groups split the rank's sampled mass about evenly where the histogram has room (within a factor of 3 of the mean, coarse cells aside)
        if (PB >= 20) {
            std::vector<double> cum(hist.size() + 1, 0.0);
            for (size_t i = 0; i < hist.size(); ++i) cum[i + 1] = cum[i] + (double)hist[i];
            auto mass_below = [&](u64 pfx) { const u64 cell = pfx >> (PB - hb); return cell >= hist.size() ? cum.back() : cum[cell]; };
            for (u32 d = 0; d < W; ++d) {
                const u64 lo = d ? bounds[d - 1] : 0, hi = d + 1 < W ? bounds[d] : 1ull << PB;
                std::vector<u64> edges{lo};
                for (u32 cv : gc) if (cv > lo && cv < hi) edges.push_back(cv);
                edges.push_back(hi);
                const double total = mass_below(hi) - mass_below(lo);
                if (edges.size() < 4 || total < 5000) continue;
                for (size_t i = 0; i + 1 < edges.size(); ++i) {
                    const double m = mass_below(edges[i + 1]) - mass_below(edges[i]);
                    CHECK(m <= 3.0 * total / (edges.size() - 1) + 2000, "PB %u rank %u: group %zu holds %.0f of %.0f sampled words in %zu groups", PB, d, i, m, total, edges.size() - 1);
                }
            }
        }
    }
    // ---- FINE bins (PREFIX_BITS > 24): bin(p) = #{cuts <= p} from the linear table; every bin inside an aligned block of 2^level
    // prefixes with level <= lmax; a group's sort_bits covers the levels of its bins; segments number a group's bins 0, 1, ...; the
    // narrow groups of a necklace-shaped histogram get 16 bits
    long fplans = 0, frefused = 0, fine16 = 0, fgroups = 0;
    for (int trial = 0; trial < 300; ++trial) {
        const u32 PB = 25 + rnd() % 4, W = 2 + rnd() % 7, G = 2 + rnd() % 5;
        const u32 lmax = (trial & 1) ? 24u : std::min(24u, PB - 4);  // (K = 31: 65..72-bit words travel as their low 64 bits)
        const u32 hb = 16;
        std::vector<u64> hist((size_t)1 << hb, 0);
        for (int i = 0; i < 200000; ++i) {  // density ~ n (1 - x)^(n - 1), n = 40: the minimum of n uniform rotations
            const double u = (double)(rnd() >> 11) / 9007199254740992.0;
            const double x = 1.0 - std::pow(1.0 - u, 1.0 / 40.0);
            hist[(size_t)(x * 0.5 * hist.size())]++;
        }
        std::vector<u32> bounds;
        {
            u64 tot = 0, run = 0;
            for (u64 h : hist) tot += h;
            size_t cell = 0;
            for (u32 d = 1; d < W; ++d) {
                while (cell < hist.size() && run + hist[cell] < tot * d / W) run += hist[cell++];
                bounds.push_back((u32)std::min<u64>(((u64)cell + 1) << (PB - hb), (1ull << PB) - 1));
            }
            for (size_t i = 1; i < bounds.size(); ++i) if (bounds[i] < bounds[i - 1]) bounds[i] = bounds[i - 1];
        }
        const std::vector<u32> gc = choose_group_cuts(hist, bounds.data(), W, G, PB);
        const FinePlan M = make_fine_plan(PB, lmax, bounds.data(), W, gc);
        if (!M.ok) { ++frefused; continue; }
        ++fplans;
        CHECK(M.cuts.size() <= FINE_MAX_CUTS, "fine plan: %zu cuts", M.cuts.size());
        auto cnt = [&](u32 p) { return (u32)(std::upper_bound(M.cuts.begin(), M.cuts.end(), p) - M.cuts.begin()); };
        auto dest = [&](u32 p) { u32 d = 0; for (u32 b : bounds) d += b <= p ? 1u : 0u; return d; };
        auto grp = [&](u32 p) { const u32 d = dest(p); const u32 lo = d ? bounds[d - 1] : 0u; u32 g = 0; for (u32 cv : gc) g += (cv > lo && cv <= p) ? 1u : 0u; return g; };
        for (int i = 0; i < 30000; ++i) {
            u32 p;
            const int kind = i % 4;
            if (kind == 0) p = (u32)(rnd() % (1ull << (PB - 1)));
            else if (kind == 1) p = (u32)((rnd() % (1ull << (PB - 1))) >> (rnd() % PB));
            else { const u32 cv = M.cuts[rnd() % M.cuts.size()]; p = cv - 1 + (u32)(rnd() % 3); if (p >= (1u << (PB - 1))) p = cv - 1; }
            const u32 bin = fine_bin(M, PB, p);
            CHECK(bin == cnt(p) && bin < 254, "PB %u: fine bin %u != %u at prefix %u", PB, bin, cnt(p), p);
            if (bin >= 254) continue;
            const u32 iv = M.iv_of[bin];
            CHECK(iv == bin, "fine plan: bin %u is interval %u", bin, iv);
            CHECK(M.dest_of[iv] == dest(p) && M.grp_of[iv] == grp(p), "fine plan: prefix %u owner %u/%u group %u/%u", p, M.dest_of[iv], dest(p), M.grp_of[iv], grp(p));
            const u32 L = M.level[iv];
            CHECK(L <= lmax && (p >> L) == (M.first[iv] >> L), "fine plan: prefix %u not in the level-%u block of its bin (first %u)", p, L, M.first[iv]);
            CHECK(L <= M.sort_bits[M.dest_of[iv]][M.grp_of[iv]], "fine plan: level %u above the group's %u sorted bits", L, M.sort_bits[M.dest_of[iv]][M.grp_of[iv]]);
            CHECK(bin >= M.bin_lo[M.dest_of[iv]] && bin < M.bin_lo[M.dest_of[iv] + 1], "fine plan: bin %u outside its rank's bins", bin);
        }
        CHECK(fine_bin(M, PB, (u32)((1ull << PB) - 1)) == 255u && M.iv_of[255] == (u32)M.cuts.size(), "fine plan: the all-ones word");
        // segments: 0, 1, 2, ... inside every (rank, group)
        for (u32 b = 0; b + 1 <= (u32)M.cuts.size(); ++b) {
            const bool same = b && M.dest_of[b] == M.dest_of[b - 1] && M.grp_of[b] == M.grp_of[b - 1];
            CHECK(M.seg_of[b] == (same ? M.seg_of[b - 1] + 1 : 0u), "fine plan: segment numbers at bin %u", b);
        }
        for (u32 d = 0; d < W; ++d) for (u32 g = 0; g < M.ngroups[d]; ++g) { ++fgroups; fine16 += M.sort_bits[d][g] == 16; }
        // the lowest rank's groups are the narrowest: with 8 ranks or more of a necklace-shaped histogram they sort 16 bits
        if (W >= 6 && G <= 4 && PB == 28) CHECK(M.sort_bits[0][0] == 16, "fine plan: PB %u W %u G %u: the first group sorts %u bits", PB, W, G, M.sort_bits[0][0]);
    }
    // one rank (insert_device_fine): one cut at 245 * 2^16, filled from the bottom — the low group sorts 16 bits at every PREFIX_BITS > 24
    for (u32 PB = 25; PB <= 28; ++PB)
        for (u32 lmax : {24u, PB - 4 < 24u ? PB - 4 : 24u}) {
            const std::vector<u32> gc{std::min<u32>(245u << 16, (1u << (PB - 1)) - (1u << 16))};
            const FinePlan M = make_fine_plan(PB, lmax, nullptr, 1, gc, true);
            CHECK(M.ok && M.ngroups[0] == 2 && M.sort_bits[0][0] == 16, "one-rank fine plan at PB %u lmax %u: ok %d groups %u bits %u", PB, lmax, (int)M.ok, M.ngroups[0], M.sort_bits[0][0]);
            if (!M.ok) continue;
            for (u32 p = 0; p < (1u << (PB - 1)); p += 4099) {
                const u32 bin = fine_bin(M, PB, p);
                CHECK(bin == (u32)(std::upper_bound(M.cuts.begin(), M.cuts.end(), p) - M.cuts.begin()) && (p >> M.level[bin]) == (M.first[bin] >> M.level[bin]) && M.level[bin] <= lmax,
                      "one-rank fine plan at PB %u: prefix %u bin %u", PB, p, bin);
                CHECK(M.grp_of[bin] == (p >= gc[0] ? 1u : 0u), "one-rank fine plan at PB %u: group of prefix %u", PB, p);
            }
        }
    // the tail weight (cost-weighted quantiles): only cells at or above the end of the fine region are scaled, by the same factor; PREFIX_BITS <= 24,
    // one rank or a weight of 100 leave the histogram alone; with the weighted histogram the tail rank of 8 gets FEWER sampled words than an eighth
    {
        std::vector<u64> h((size_t)1 << 16, 0);
        for (int i = 0; i < 400000; ++i) {
            const double u = (double)(rnd() >> 11) / 9007199254740992.0;
            h[(size_t)((1.0 - std::pow(1.0 - u, 1.0 / 40.0)) * 0.5 * h.size())]++;
        }
        std::vector<u64> w = h;
        weigh_tail_for_fine_bins(w, 28, 8, 4, 16);
        const size_t first = (size_t)(((u64)(FINE_MAX_CUTS - 31 - 8) << FINE_LEVEL) >> 12);
        bool ok = true;
        for (size_t i = 0; i < h.size(); ++i) ok = ok && w[i] == h[i] * (i >= first ? 120u : 100u);
        CHECK(ok, "tail weight: cells below %zu x 100, from there on x 120", first);
        std::vector<u64> same = h;
        weigh_tail_for_fine_bins(same, 24, 8, 4, 16);
        CHECK(same == h, "tail weight touches PREFIX_BITS = 24");
        weigh_tail_for_fine_bins(same, 28, 1, 4, 16);
        CHECK(same == h, "tail weight touches a one-rank job");
        auto tail_share = [&](const std::vector<u64>& hist) {  // sampled words (unweighted) of the last of 8 quantile ranges of `hist`
            u64 tot = 0, run = 0, words = 0, all = 0;
            for (u64 x : hist) tot += x;
            size_t cell = 0;
            while (cell < hist.size() && run + hist[cell] < tot * 7 / 8) run += hist[cell++];
            for (size_t i = 0; i < h.size(); ++i) { all += h[i]; if (i > cell) words += h[i]; }
            return (double)words / (double)all;
        };
        CHECK(tail_share(w) < tail_share(h) && tail_share(w) > 0.08 && tail_share(h) < 0.13, "tail weight: the tail rank's share %.4f -> %.4f", tail_share(h), tail_share(w));
    }
    printf("cut plan unit: %ld plans checked, %ld refused, %ld bad; fine: %ld plans, %ld refused, %ld of %ld groups sort 16 bits\n", plans, refused, bad, fplans, frefused, fine16, fgroups);
    return bad || plans < 100 || fplans < 100 ? 1 : 0;
}
