// pipeline.hpp — one batch of words into the resident index: KRN-1 front end (chunk plan + encode), KRN-2 stable partition,
// KRN-4 directory, KRN-3 bucket kernels, the incremental (non-empty index) path and `self |= other`. Included by cblx.cpp only.
#pragma once
#include "ctx.hpp"
#include "kernels_kmer.hpp"

#ifndef CBLX_CLAIM_FIRST
#define CBLX_CLAIM_FIRST 0  // measured: the claim table as FIRST kernel of the runs <= 1024 words costs cfg 2 (no repeats) +0.25 ms and saves the 30x-coverage workload 2.4 ms
#endif

namespace {

// ---- the sort + directory + per-bucket pipeline over N records (lo/hi), resident records first -------------
struct Records {
    Buf<u64> lo, lo2;
    Buf<u8> hi, hi2;  // raw bytes; element size = hi_elem_size
    const u64* ext_lo = nullptr;  // optional caller-owned source of the FIRST pass (no resident words in front)
    const void* ext_hi = nullptr;
};

// Records that arrive with the first partition pass already done and in PIECES (receiver of the multi-GPU build, comm.hpp):
// rec.lo (and rec.hi for words that keep their hi part) is the receive arena; piece p = what one (slice, source rank) sent,
// sorted by the pass-A segment, cnt[p][256] records per segment from arena position pbase[p]; stream order = piece order.
struct PieceInput {
    u32 np = 0;
    const u32* cnt = nullptr;    // host [np][256]
    const u32* pbase = nullptr;  // host [np]
    Buf<u8>* dig = nullptr;      // the first LSD pass's digit of every record (same positions), owned by the caller; reused as the side channel
    // ... or, for ONE GROUP of a receive log that other groups still need (comm.hpp, grouped receiver): the digits are read where they
    // are and the later passes' side channel goes to a buffer of the group's own (at least N + 64 bytes)
    const u8* dig_in = nullptr;
    u8* dig_out = nullptr;
    // FINE bins (cuts.hpp: make_fine_plan): the "segments" are bins of the senders' first pass that lie inside aligned blocks of
    // 2^sort_bits prefixes — the passes sort that many prefix bits (16: two passes, 24: three) and a record's prefix is
    // seg_prefix[segment] | its low sort_bits bits (host array of 256; 0 / null: segment = top 8 prefix bits, PREFIX_BITS - 8 bits sorted)
    u32 sort_bits = 0;
    const u32* seg_prefix = nullptr;
};

// A group of the grouped receiver works on its WINDOW [w_lo, w_hi) of the prefix space (both multiples of 64; its records hold no
// other prefix): start_dense, the popcounts and the bucket table cover the window only, the bitvector words are written straight
// into the final bitvector `bv` (2^PB bits, zeroed by the caller; windows of different groups share no word), and the directory
// that comes back is the window's: nr.nb buckets, nr.prefix (absolute values), nr.start (positions relative to the group's records),
// nr.rank_dir relative to the window. nr.bv stays empty.
struct DirWindow {
    u32 w_lo = 0, w_hi = 0;
    u64* bv = nullptr;
};

// the LSD passes behind pass A: digit widths and shifts (relative to SUFFIX_BITS) of the remaining prefix bits
#ifndef CBLX_PREFIX_SPLIT
#define CBLX_PREFIX_SPLIT 1  // PREFIX_BITS > 24: the last PB - 24 bits by k_prefix_split instead of a third LSD pass (0: three LSD passes of 7 + 7 + 6 bits)
#endif
struct LsdPlan {
    u32 npass = 0, wid[4] = {0, 0, 0, 0}, sh[5] = {0, 0, 0, 0, 0};
    u32 xb = 0;  // lowest prefix bits left to k_prefix_split (the passes sort the PB - xb bits above them)
    u32 total() const { return npass + (xb ? 1u : 0u); }  // times the records change buffers behind pass A
};
// `split_ok` = false: records that arrive in pieces (the receiver of the multi-GPU build, and its senders, whose side channel carries
// the first pass's digit): at the bucket depth of a many-GPU job a run of equal 24-bit prefix holds tens of thousands of records, which
// k_prefix_split has to read twice (12 ms per 1.5 G records against 6.2 for a scatter pass, profiles/r04_wire_emulated.md): three passes.
inline LsdPlan lsd_plan(const Consts& P, bool split_ok = true) {
    // Up to two passes: 8 bits, then the rest (the group-cut tiles of the last pass are built for that shape). PREFIX_BITS > 24:
    // 8 + 8 bits by two passes with the directory of 2^24 "super-prefixes" from the second one's tables, and the last 1 .. 4 bits
    // by k_prefix_split (a run of equal 24-bit prefix staged in LDS, written back in order: copy speed, DESIGN_HISTORY.md §3.11). Three passes
    // of 7 + 7 + 6 bits before that (CBLX_PREFIX_SPLIT=0): a pass costs nearly the same whatever its width.
    LsdPlan L;
    const u32 RB = P.PB - std::min(8u, P.PB);
    if (CBLX_PREFIX_SPLIT && split_ok && P.PB > 24) {
        L.xb = P.PB - 24;
        L.npass = 2;
        L.wid[0] = L.wid[1] = 8;
        L.sh[0] = L.xb; L.sh[1] = L.xb + 8; L.sh[2] = L.xb + 16;
        return L;
    }
    L.npass = (RB + 7) / 8;
    for (u32 i = 0; i < L.npass; ++i) {
        L.wid[i] = L.npass <= 2 ? std::min(8u, RB - 8 * i) : RB / L.npass + (i < RB % L.npass ? 1u : 0u);
        L.sh[i + 1] = L.sh[i] + L.wid[i];
    }
    return L;
}
// FINE bins: `bits` (16 or 24) prefix bits in passes of 8 from the bottom — the first digit is the same for every group, so the senders'
// side channel does not depend on who receives a record
inline LsdPlan lsd_plan_bits(u32 bits) {
    LsdPlan L;
    L.npass = (bits + 7) / 8;
    for (u32 i = 0; i < L.npass; ++i) { L.wid[i] = std::min(8u, bits - 8 * i); L.sh[i + 1] = L.sh[i] + L.wid[i]; }
    return L;
}

// KRN-2 + KRN-4 over N records: stable partition by prefix, then the directory (bitvector, rank directory, bucket table
// with the RAW run of every prefix; nr.cnt / nr.kind are allocated, not filled). The sorted records end up in rec.lo/hi.
// `countsA`: histogram of the first pass already accumulated by KRN-1 (empty Buf = compute it here)
template <typename C> void partition_and_directory(cblx_ctx* c, Records& rec, u64 N, Buf<u32> countsA, Resident& nr, const PieceInput* pin = nullptr, const DirWindow* win = nullptr) {
    typedef typename C::HiT HiT;
    const Consts& P = c->P;
    if (N >= 0xFFFFFFF0ull) throw Error(CBLX_ERANGE, "a single batch takes fewer than 2^32-16 words (callers cut larger inserts into sub-batches)");
    // ping-pong: A = rec.lo/hi, B = rec.lo2/hi2. With an external source pass 0 reads it and writes A.
    const u64* lo = rec.ext_lo ? rec.ext_lo : rec.lo.get();
    const HiT* hi = rec.ext_lo ? (const HiT*)rec.ext_hi : (const HiT*)rec.hi.get();
    u64* lo2 = rec.ext_lo ? rec.lo.get() : rec.lo2.get();
    HiT* hi2 = rec.ext_lo ? (HiT*)rec.hi.get() : (HiT*)rec.hi2.get();
    u64* lo_other = rec.ext_lo ? rec.lo2.get() : rec.lo.get();   // the buffer that becomes the destination after pass 0
    HiT* hi_other = rec.ext_lo ? (HiT*)rec.hi2.get() : (HiT*)rec.hi.get();
    auto advance = [&]() { const u64* nl = lo2; const HiT* nh = hi2; lo2 = lo_other; hi2 = hi_other; lo_other = const_cast<u64*>(nl); hi_other = const_cast<HiT*>(nh); lo = nl; hi = nh; };
    // -- KRN-2: stable radix partition on the PREFIX_BITS above SUFFIX_BITS.
    //    Pass A sorts by the MOST significant 8 prefix bits (the skewed digit: long output runs) and cuts the array into
    //    <= 256 segments; the remaining bits are sorted by stable LSD passes INSIDE every segment (tiles never straddle a
    //    segment). For 65..72-bit words (K = 31) the bits the hi byte held are implied by the segment after pass A, so
    //    it is dropped there: every later pass, the boundary scan and KRN-3 move 8-byte records only.
    //    Per pass: tile histogram, column scan, per-segment adjust, LDS-staged scatter.
    constexpr bool DROP_HI = std::is_same<HiT, u8>::value;
    Buf<u32> seg_start(c->pool, 257);
    // PREFIX_BITS > 24: the passes sort the top PBs = 24 prefix bits ("super-prefixes": the xb bits below them count as suffix here,
    // SBs), their directory comes from the last pass's tables, and k_prefix_split finishes the job run by run (below)
    const bool fine = pin && pin->sort_bits != 0;
    const LsdPlan LP = fine ? lsd_plan_bits(pin->sort_bits) : lsd_plan(P, pin == nullptr);
    const u32 xb = LP.xb, PBs = P.PB - xb, SBs = P.SB + xb;
    const u32 nA = std::min(8u, PBs), RB = fine ? pin->sort_bits : PBs - nA;  // bits of pass A, bits left for the LSD passes
    Buf<u32> d_segp;  // FINE bins: first prefix of every segment's aligned block
    if (fine) { d_segp = Buf<u32>(c->pool, 256); h2d(c, d_segp.get(), pin->seg_prefix, 256); }
    const u32* segp = d_segp.get();
    const u32 w_lo = win ? win->w_lo : 0u;
    const u64 nprefix = win ? (u64)win->w_hi - win->w_lo : 1ull << P.PB, nwords = std::max<u64>(1, nprefix / 64);
    if (win && ((win->w_lo | win->w_hi) & 63u)) throw Error(CBLX_EINVAL, "a directory window must be cut at multiples of 64 (internal error)");
    // the directory the PASSES produce: of the prefixes (xb = 0), or of the super-prefixes of the window
    const u32 sw_lo = w_lo >> xb;
    const u64 sw_hi = win ? ((u64)win->w_hi + ((1u << xb) - 1u)) >> xb : 1ull << PBs, nsuper = sw_hi - sw_lo;
    Buf<u32> start_dense(c->pool, std::max<u64>(nsuper, 1));
    u32* sd = start_dense.get() - sw_lo;  // indexed by the absolute (super-)prefix
    bool have_dense = false;
    {
        // pieces: every (segment, piece) may end in a partly filled tile
        const u32 ntiles = (u32)ceil_div(N, RDX_TILE), nt_max = ntiles + 256 * (pin ? std::max(1u, pin->np) : 1u);
        const u32 npassL = LP.npass, nseg = 1u << nA;
        const u32 (&wid)[4] = LP.wid;
        const u32 (&sh)[5] = LP.sh;
        if (pin && (npassL == 0 || nA != 8)) throw Error(CBLX_EINVAL, "records in pieces need PREFIX_BITS >= 9 (internal error)");
        // The last LSD pass cuts its tiles at (segment x lower digits) groups when there are few enough of them; the
        // bucket directory then comes from that pass's tables (k_dir_gather) instead of a scan of the sorted records.
        const u32 low_bits = npassL ? sh[npassL - 1] - xb : 0, last_bits = RB - low_bits;
        const bool tbl_dir = npassL >= 1 && nA + low_bits <= 16;
        const bool grp_tiles = tbl_dir && low_bits > 0;  // low_bits = 0: the groups are the segments (existing tile table)
        const u32 G = nseg << low_bits, nt_maxC = grp_tiles ? ntiles + G + 256 : nt_max;
        const bool haveA = countsA.get() != nullptr;
        Buf<u32> counts = haveA ? std::move(countsA) : Buf<u32>(c->pool, (size_t)256 * nt_max);
        Buf<u32> colpre(c->pool, (size_t)256 * nt_maxC), scratch, coltot(c->pool, 256),
            adj(c->pool, 256 * 256), seg_first(c->pool, 257), nt_dev(c->pool, 1), t_start(c->pool, nt_max), t_count(c->pool, nt_max);
        Buf<u16> t_seg(c->pool, nt_max);
        Buf<u32> grp_start, grp_first, seg_firstC, nt_devC, t_startC, t_countC;
        Buf<u16> t_segC;
        if (grp_tiles) {
            grp_start = Buf<u32>(c->pool, G + 1);
            grp_first = Buf<u32>(c->pool, G + 1);
            seg_firstC = Buf<u32>(c->pool, 257);
            nt_devC = Buf<u32>(c->pool, 1);
            t_startC = Buf<u32>(c->pool, nt_maxC);
            t_countC = Buf<u32>(c->pool, nt_maxC);
            t_segC = Buf<u16>(c->pool, nt_maxC);
        }
        // digit side channel: a scatter also writes the NEXT pass's digit of every record (1 byte, same order). (When the
        // hi byte is dropped by pass A the remaining digits all lie in the lo word: the word has <= 72 bits.)
        Buf<u8> dig;
        const u8* dig_rd = nullptr;  // what this pass's histogram reads ...
        u8* dig_wr = nullptr;        // ... and where its scatter leaves the next pass's digits (the same array unless the caller split them)
        Buf<u32> seg_tot;  // pieces: records per pass-A segment
        bool have_dig = false;
        auto next_digit = [&](u32 next_pass) -> DigitBits {
            if (next_pass >= npassL) return DigitBits{0, 0};
            return DigitBits{P.SB + sh[next_pass], wid[next_pass]};
        };
        if (pin && pin->dig_in) { dig_rd = pin->dig_in; dig_wr = pin->dig_out; have_dig = true; }
        else if (pin && pin->dig) { dig = std::move(*pin->dig); have_dig = dig.get() != nullptr; dig_rd = dig_wr = dig.get(); }
        else if (next_digit(0).nbits) { dig = Buf<u8>(c->pool, N + 64); dig_rd = dig_wr = dig.get(); }
        if (pin) {
            // pass A ran on the senders: segment and tile tables of the first LSD pass from the piece table
            StageTimer t(c, ST_SCAN);
            const u32 np = pin->np;
            seg_tot = Buf<u32>(c->pool, 256);
            Buf<u32> d_cnt(c->pool, (size_t)np * 256), d_pbase(c->pool, np), pfirst(c->pool, (size_t)np * 256), ppos(c->pool, (size_t)np * 256);
            h2d(c, d_cnt.get(), pin->cnt, (size_t)np * 256);
            h2d(c, d_pbase.get(), pin->pbase, np);
            hipLaunchKernelGGL(k_piece_tables, dim3(1), dim3(256), 0, c->stream, d_cnt.get(), d_pbase.get(), np, seg_start.get(), seg_first.get(), nt_dev.get(), pfirst.get(), ppos.get(),
                               seg_tot.get());
            hipLaunchKernelGGL(k_piece_tile_table, grid1(nt_max, 256), dim3(256), 0, c->stream, d_cnt.get(), np, seg_first.get(), nt_dev.get(), pfirst.get(), ppos.get(), t_start.get(),
                               t_count.get(), t_seg.get());
            CBLX_HIP(hipGetLastError());
            // the piece tables are released here. (A group of the grouped receiver skips the host wait: everything that could reuse the
            // blocks is queued on this stream behind the kernels that read them, and eight groups pay every idle gap eight times)
            if (!win) CBLX_HIP(hipStreamSynchronize(c->stream));
        } else {   // pass A
            const TileView tv{nullptr, nullptr, nullptr, nullptr, ntiles, N};
            const DigitBits dfn{SBs + RB, nA};
            const DigitBits nd = next_digit(0);
            u8* ndp = nd.nbits ? dig_wr : nullptr;
            if (!haveA) { StageTimer t(c, ST_HIST);
              hipLaunchKernelGGL((k_radix_hist<HiT, DigitBits>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, c->stream, lo, hi, tv, dfn, counts.get()); }
            { StageTimer t(c, ST_SCAN);
              colscan(c, counts.get(), nullptr, ntiles, colpre.get(), coltot.get(), scratch);
              hipLaunchKernelGGL(k_seg_adjust, dim3(1), dim3(256), 0, c->stream, colpre.get(), coltot.get(), (const u32*)nullptr, (const u32*)nullptr,
                                 (const u32*)nullptr, ntiles, 1u, adj.get(), (u32*)nullptr);
              hipLaunchKernelGGL(k_seg_table, dim3(1), dim3(256), 0, c->stream, coltot.get(), seg_start.get(), seg_first.get(), nt_dev.get());
              hipLaunchKernelGGL(k_tile_table, grid1(nt_max, 256), dim3(256), 0, c->stream, seg_start.get(), seg_first.get(), nt_dev.get(), t_start.get(),
                                 t_count.get(), t_seg.get()); }
            { StageTimer t(c, ST_SCATTER);
              c->stages[ST_SCATTER].units += N;  // (records through a partition pass: cblx_stage_units — the passes a record takes vary with the route)
              if constexpr (DROP_HI)
                  hipLaunchKernelGGL((k_radix_scatter<HiT, NoHi, DigitBits>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, c->stream, lo, hi, tv, dfn, colpre.get(),
                                     adj.get(), lo2, (NoHi*)nullptr, nd, ndp);
              else
                  hipLaunchKernelGGL((k_radix_scatter<HiT, HiT, DigitBits>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, c->stream, lo, hi, tv, dfn, colpre.get(),
                                     adj.get(), lo2, hi2, nd, ndp); }
            have_dig = ndp != nullptr;
            advance();
        }
        if (counts.n < (size_t)256 * nt_maxC) counts = Buf<u32>(c->pool, (size_t)256 * nt_maxC);
        const TileView tvL{t_start.get(), t_count.get(), t_seg.get(), nt_dev.get(), nt_max, N};
        const TileView tvC{t_startC.get(), t_countC.get(), t_segC.get(), nt_devC.get(), nt_maxC, N};
        for (u32 pass = 0; pass < npassL; ++pass) {
            const DigitBits dfn{P.SB + sh[pass], wid[pass]};
            const bool last = pass + 1 == npassL;
            const bool cut = last && grp_tiles;  // this pass runs on the group-cut tiles
            const TileView& tv = cut ? tvC : tvL;
            const u32 ntm = cut ? nt_maxC : nt_max;
            const u32 *ntd = cut ? nt_devC.get() : nt_dev.get(), *sf = cut ? seg_firstC.get() : seg_first.get();
            auto run = [&](auto hi_tag) {
                typedef decltype(hi_tag) H;  // record layout of the LSD passes: no hi once it was dropped
                const H* hin = (const H*)hi;
                H* hout = (H*)hi2;
                const DigitBits nd = next_digit(pass + 1);
                u8* ndp = nd.nbits && dig_wr ? dig_wr : nullptr;
                { StageTimer t(c, ST_HIST);
                  if (have_dig)
                      hipLaunchKernelGGL(k_radix_hist_bytes, dim3((xcd_grid(ntm) + HISTB_WAVES - 1) / HISTB_WAVES + 8), dim3(64 * HISTB_WAVES), 0, c->stream, dig_rd, tv, counts.get());
                  else
                      hipLaunchKernelGGL((k_radix_hist<H, DigitBits>), dim3(xcd_grid(ntm)), dim3(RDX_THREADS), 0, c->stream, lo, hin, tv, dfn, counts.get()); }
                { StageTimer t(c, ST_SCAN);
                  colscan(c, counts.get(), ntd, ntm, colpre.get(), coltot.get(), scratch);
                  hipLaunchKernelGGL(k_seg_adjust, dim3(nseg), dim3(256), 0, c->stream, colpre.get(), coltot.get(), sf, seg_start.get(), ntd, ntm, nseg, adj.get(),
                                     (grp_tiles && pass + 2 == npassL) ? grp_start.get() : (u32*)nullptr);
                  if (grp_tiles && pass + 2 == npassL) {  // the next pass is the last one: cut its tiles at the groups this pass creates
                      hipLaunchKernelGGL(k_grp_table, dim3(1), dim3(1024), 0, c->stream, G, low_bits, grp_start.get(), seg_start.get(), (u32)N, grp_first.get(), seg_firstC.get(),
                                         nt_devC.get());
                      hipLaunchKernelGGL(k_tile_table_grp, grid1(nt_maxC, 256), dim3(256), 0, c->stream, G, low_bits, grp_start.get(), seg_start.get(), (u32)N, grp_first.get(), nt_devC.get(),
                                         t_startC.get(), t_countC.get(), t_segC.get());
                  } }
                // more groups than the table method takes: the last pass finds the bucket starts itself (fused directory)
                const bool fused_dir = last && !tbl_dir;
                const u32 amb_stride = 1u << dfn.nbits;
                Buf<u32> amb;
                if (fused_dir) {
                    amb = Buf<u32>(c->pool, (size_t)ntm * amb_stride);
                    CBLX_HIP(hipMemsetAsync(start_dense.get(), 0xFF, nsuper * 4, c->stream));
                }
                { StageTimer t(c, ST_SCATTER);
                  c->stages[ST_SCATTER].units += N;
                  hipLaunchKernelGGL((k_radix_scatter<H, H, DigitBits>), dim3(xcd_grid(ntm)), dim3(RDX_THREADS), 0, c->stream, lo, hin, tv, dfn, colpre.get(),
                                     adj.get(), lo2, hout, nd, ndp, fused_dir ? sd : (u32*)nullptr, SBs, RB, low_bits, amb.get(), amb_stride,
                                     OwnWindow{0, 0, nullptr, nullptr, nullptr}, (const u64*)nullptr, segp); }
                if (fused_dir) {
                    StageTimer t(c, ST_DIR);
                    hipLaunchKernelGGL(k_dir_resolve<H>, grid1((u64)ntm * amb_stride, 256), dim3(256), 0, c->stream, ntd, amb_stride, (const u32*)amb.get(), tv.seg,
                                       (const u32*)seg_start.get(), (const u64*)lo2, (const H*)hout, SBs, RB, sd, segp);
                    CBLX_HIP(hipStreamSynchronize(c->stream));  // amb is released at the end of this scope
                    have_dense = true;
                }
                have_dig = ndp != nullptr;
                if (have_dig) dig_rd = dig_wr;  // the next pass reads what this one wrote
                if (pin && pass == 0 && !last) {
                    // the piece tiles described the arena; from here on the segments are contiguous: the plain tile table
                    // (the column totals of the pass that made the segments were kept aside: this pass's scan overwrote coltot)
                    StageTimer t(c, ST_SCAN);
                    hipLaunchKernelGGL(k_seg_table, dim3(1), dim3(256), 0, c->stream, (const u32*)seg_tot.get(), seg_start.get(), seg_first.get(), nt_dev.get());
                    hipLaunchKernelGGL(k_tile_table, grid1(nt_max, 256), dim3(256), 0, c->stream, seg_start.get(), seg_first.get(), nt_dev.get(), t_start.get(), t_count.get(), t_seg.get());
                }
                if (last && tbl_dir) {
                    StageTimer t(c, ST_DIR);
                    // FINE bins: the segments' blocks need not cover the window (the last group's reaches up to 2^PREFIX_BITS, its bins do not)
                    if (fine) CBLX_HIP(hipMemsetAsync(start_dense.get(), 0xFF, nsuper * 4, c->stream));
                    hipLaunchKernelGGL(k_dir_gather, dim3(G), dim3(256), 0, c->stream, low_bits, last_bits, grp_tiles ? grp_first.get() : seg_first.get(), seg_start.get(), ntd,
                                       colpre.get(), coltot.get(), adj.get(), sd, sw_lo, win ? (u32)std::min<u64>(sw_hi, 0xFFFFFFFFull) : 0xFFFFFFFFu, segp);
                    if (grp_tiles)  // cold segments kept plain tiles: their boundaries come from their (few) records, now in lo2
                        hipLaunchKernelGGL(k_boundaries_cold<H>, dim3(nseg, 32), dim3(256), 0, c->stream, (const u64*)lo2, (const H*)hout, SBs, RB, seg_start.get(), sd, segp);
                    have_dense = true;
                }
            };
            if constexpr (DROP_HI) run(NoHi()); else run(HiT());
            advance();
        }
        CBLX_HIP(hipGetLastError());
        if (!win) CBLX_HIP(hipStreamSynchronize(c->stream));  // the pass tables are released here (a group: see above)
    }
    bool dir_done = false;
    if (xb) {
        // -- the last xb prefix bits: every run of equal super-prefix split in LDS, written to the other buffer in final order; the
        //    directory comes from the runs' own small tables (k_prefix_split, k_split_table) — nothing walks the 2^PREFIX_BITS prefixes
        if (!have_dense) throw Error(CBLX_EDEVICE, "prefix split without a super-prefix directory (internal error)");
        if (win) throw Error(CBLX_EINVAL, "prefix split inside a directory window (internal error)");
        const u64 swords = std::max<u64>(1, (nsuper + 63) / 64);
        Buf<u32> spopc(c->pool, swords), sprefix;
        Buf<u64> sbv(c->pool, swords), srank(c->pool, swords + 1), sstart;
        u64 nruns = 0;
        { StageTimer t(c, ST_DIR);
          CBLX_HIP(hipMemsetAsync(spopc.get(), 0, swords * 4, c->stream));
          CBLX_HIP(hipMemsetAsync(sbv.get(), 0, swords * 8, c->stream));
          hipLaunchKernelGGL(k_bitvector, grid1(std::max<u64>(nsuper, 64), 256), dim3(256), 0, c->stream, start_dense.get(), nsuper, sbv.get(), spopc.get());
          nruns = exclusive_scan<u64>(c, spopc.get(), swords, srank.get());
          sprefix = Buf<u32>(c->pool, nruns + 1);
          sstart = Buf<u64>(c->pool, nruns + 1);
          hipLaunchKernelGGL(k_bucket_table, grid1(std::max<u64>(nsuper, 1), 256), dim3(256), 0, c->stream, start_dense.get(), nsuper, sbv.get(), srank.get(), sprefix.get(), sstart.get(), sw_lo);
          hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, sstart.get() + nruns, N); }
        if (nruns >= 0xFFFFFFF0ull / 16) throw Error(CBLX_ERANGE, "too many runs for the prefix split");
        Buf<u32> sub_start(c->pool, 16 * nruns + 16), sub_nz(c->pool, nruns + 1);
        Buf<u16> sub_mask(c->pool, nruns + 1);
        Buf<u64> rank_base(c->pool, nruns + 1);
        if (nruns) {
            StageTimer t(c, ST_SCATTER);
            c->stages[ST_SCATTER].units += N;
            Buf<SplitRun> lists(c->pool, SPLIT_CLASSES * nruns);
            Buf<u32> list_n(c->pool, SPLIT_CLASSES);
            CBLX_HIP(hipMemsetAsync(list_n.get(), 0, SPLIT_CLASSES * 4, c->stream));
            hipLaunchKernelGGL(k_split_classify, grid1(nruns, 1024), dim3(1024), 0, c->stream, nruns, sstart.get(), lists.get(), list_n.get());
            const std::vector<u32> ln = d2h_vec<u32>(c, list_n.get(), SPLIT_CLASSES);
            auto launch = [&](auto h_tag, const auto* hin, auto* hout) {
                typedef decltype(h_tag) H;
                if (ln[0]) hipLaunchKernelGGL((k_prefix_split<H, 64>), dim3(ln[0]), dim3(64), 0, c->stream, lists.get(), lo, hin, lo2, hout, P.SB, xb, sub_start.get(), sub_mask.get(), sub_nz.get());
                if (ln[1]) hipLaunchKernelGGL((k_prefix_split<H, 128>), dim3(ln[1]), dim3(128), 0, c->stream, lists.get() + nruns, lo, hin, lo2, hout, P.SB, xb, sub_start.get(), sub_mask.get(), sub_nz.get());
                if (ln[2]) hipLaunchKernelGGL((k_prefix_split<H, 256>), dim3(ln[2]), dim3(256), 0, c->stream, lists.get() + 2 * nruns, lo, hin, lo2, hout, P.SB, xb, sub_start.get(), sub_mask.get(), sub_nz.get());
                if (ln[3]) hipLaunchKernelGGL((k_prefix_split<H, 512>), dim3(ln[3]), dim3(512), 0, c->stream, lists.get() + 3 * nruns, lo, hin, lo2, hout, P.SB, xb, sub_start.get(), sub_mask.get(), sub_nz.get());
            };
            if constexpr (DROP_HI || !HiTraits<HiT>::has) launch(NoHi(), (const NoHi*)nullptr, (NoHi*)nullptr);
            else launch(HiT(), hi, hi2);
            CBLX_HIP(hipGetLastError());
            advance();
        }
        {   // KRN-4 from the runs' tables: bucket table rows by rank, bitvector bits run by run, rank directory by one popcount scan
            StageTimer t(c, ST_DIR);
            nr.nb = nruns ? exclusive_scan<u64>(c, sub_nz.get(), nruns, rank_base.get()) : 0;
            nr.bv = Buf<u64>(c->pool, nwords);
            nr.rank_dir = Buf<u64>(c->pool, nwords + 1);
            nr.prefix = Buf<u32>(c->pool, nr.nb + 1);
            nr.start = Buf<u64>(c->pool, nr.nb + 1);
            nr.cnt = Buf<u32>(c->pool, nr.nb + 1);
            nr.kind = Buf<u8>(c->pool, nr.nb + 1);
            Buf<u32> popc(c->pool, nwords);
            CBLX_HIP(hipMemsetAsync(nr.bv.get(), 0, nwords * 8, c->stream));
            if (nruns)
                hipLaunchKernelGGL(k_split_table, grid1(nruns * 16, 256), dim3(256), 0, c->stream, nruns, (const u32*)sprefix.get(), (const u32*)sub_start.get(), (const u16*)sub_mask.get(),
                                   (const u64*)rank_base.get(), xb, nr.prefix.get(), nr.start.get(), nr.bv.get());
            hipLaunchKernelGGL(k_popc_words, grid1(nwords, 256), dim3(256), 0, c->stream, nwords, (const u64*)nr.bv.get(), popc.get());
            const u64 nb2 = exclusive_scan<u64>(c, popc.get(), nwords, nr.rank_dir.get());
            if (nb2 != nr.nb) throw Error(CBLX_EDEVICE, "prefix split: the runs hold " + std::to_string(nr.nb) + " buckets, the bitvector " + std::to_string(nb2) + " (internal error)");
            hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, nr.start.get() + nr.nb, N);
            CBLX_HIP(hipGetLastError());
            CBLX_HIP(hipStreamSynchronize(c->stream));  // the run tables are released here
        }
        dir_done = true;
    }
    if (lo == rec.lo2.get()) { std::swap(rec.lo, rec.lo2); std::swap(rec.hi, rec.hi2); }  // final data -> rec.lo/hi
    rec.lo2.reset();
    rec.hi2.reset();
    // -- KRN-4: bitvector, rank directory, bucket table (of the whole prefix space, or of the caller's window of it)
    if (!dir_done) {
        StageTimer t(c, ST_DIR);
        Buf<u32> popc(c->pool, nwords);
        u64* bvp;
        if (win) bvp = win->bv + (w_lo >> 6);
        else { nr.bv = Buf<u64>(c->pool, nwords); bvp = nr.bv.get(); CBLX_HIP(hipMemsetAsync(bvp, 0, nwords * 8, c->stream)); }
        nr.rank_dir = Buf<u64>(c->pool, nwords + 1);
        CBLX_HIP(hipMemsetAsync(popc.get(), 0, nwords * 4, c->stream));
        if (!have_dense) {  // boundaries from a scan of the sorted records (more groups than the table method takes)
            if (win) throw Error(CBLX_EINVAL, "a directory window needs PREFIX_BITS >= 9 (internal error)");
            CBLX_HIP(hipMemsetAsync(start_dense.get(), 0xFF, nprefix * 4, c->stream));
            if constexpr (DROP_HI)
                hipLaunchKernelGGL(k_boundaries_seg, grid1(ceil_div(N, 4), 256), dim3(256), 0, c->stream, lo, N, P.SB, RB, seg_start.get(), start_dense.get());
            else
                hipLaunchKernelGGL(k_boundaries<HiT>, grid1(ceil_div(N, 4), 256), dim3(256), 0, c->stream, lo, hi, N, P.SB, P.PB, start_dense.get());
        }
        if (nprefix) hipLaunchKernelGGL(k_bitvector, grid1(std::max<u64>(nprefix, 64), 256), dim3(256), 0, c->stream, start_dense.get(), nprefix, bvp, popc.get());
        nr.nb = nprefix ? exclusive_scan<u64>(c, popc.get(), nwords, nr.rank_dir.get()) : 0;
        nr.prefix = Buf<u32>(c->pool, nr.nb + 1);
        nr.start = Buf<u64>(c->pool, nr.nb + 1);
        nr.cnt = Buf<u32>(c->pool, nr.nb + 1);
        nr.kind = Buf<u8>(c->pool, nr.nb + 1);
        if (nprefix) hipLaunchKernelGGL(k_bucket_table, grid1(nprefix, 256), dim3(256), 0, c->stream, start_dense.get(), nprefix, bvp, nr.rank_dir.get(), nr.prefix.get(), nr.start.get(), w_lo);
        hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, nr.start.get() + nr.nb, N);
        CBLX_HIP(hipGetLastError());
    }
}

// general kernel for runs of any length (and the fallback of the big path): one workgroup per run, LSD radix in global scratch
template <typename C> void huge_stage(cblx_ctx* c, const BDesc* d_list, const u32* d_list_n, u32 nh, u64* a_lo, typename C::HiT* a_hi, Resident& nr, const MergeArgs& ma) {
    typedef typename C::HiT HiT;
    if (nh == 0) return;
    StageTimer t(c, ST_BHUGE);
    std::vector<BDesc> hl = d2h_vec<BDesc>(c, d_list, nh);
    std::vector<u64> so(nh);
    u64 tot = 0;
    for (u32 i = 0; i < nh; ++i) { so[i] = tot; tot += hl[i].c & ~BDESC_TRIE; }
    Buf<u64> d_so(c->pool, nh), s_alo(c->pool, tot), s_blo(c->pool, tot), s_ahi(c->pool, C::WS ? tot : 1), s_bhi(c->pool, C::WS ? tot : 1);
    Buf<u32> s_aidx(c->pool, tot), s_bidx(c->pool, tot);
    h2d(c, d_so.get(), so.data(), nh);
    hipLaunchKernelGGL((k_bucket_huge<C::WS, HiT>), dim3(nh), dim3(256), 0, c->stream, d_list, d_list_n, d_so.get(), a_lo, a_hi, c->P.SB, s_alo.get(), s_ahi.get(),
                       s_aidx.get(), s_blo.get(), s_bhi.get(), s_bidx.get(), nr.cnt.get(), nr.kind.get(), ma);
    CBLX_HIP(hipGetLastError());
    CBLX_HIP(hipStreamSynchronize(c->stream));
}
// lanes per bucket of the per-bucket copy kernels, from the average run length (words / buckets)
template <typename F> void with_lpb(u64 words, u64 buckets, F&& f) {
    const u64 avg = buckets ? words / buckets : 0;
    if (avg >= 48) f(std::integral_constant<int, 64>());
    else if (avg >= 12) f(std::integral_constant<int, 16>());
    else f(std::integral_constant<int, 4>());
}
inline dim3 lpb_grid(u64 buckets, int lpb) { return dim3((unsigned)std::max<u64>(1, ceil_div(buckets * (u64)lpb, 256))); }

__global__ void k_sum_bdesc_len(const BDesc* __restrict__ list, u32 n, u64* __restrict__ out) {
    u64 s = 0;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += list[i].c & ~BDESC_TRIE;
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd((unsigned long long*)out, (unsigned long long)s);
}
// (profiling) sum of the final counts of the buckets of a list
__global__ void k_sum_list_counts(const BDesc* __restrict__ list, u32 n, const u32* __restrict__ cnt, u64* __restrict__ out) {
    u64 s = 0;
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += cnt[list[i].r];
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd((unsigned long long*)out, (unsigned long long)s);
}
// runs of 4097 .. BIG_MAX words (kernels_bucket.hpp: k_big_*): one more partition pass on the top suffix bits with the runs as
// segments, out of the arena into a twin buffer at the same positions; sub-ranges sorted + deduplicated in place there by
// k_bucket_msd; what cannot be finished that way takes the general kernel on the (untouched) arena run. The finished runs stay
// in the twin: finish_twin decides which buffer becomes the arena.
// CBLX_SORTED_KERNEL=0: the runs that end up sorted take k_bucket_msd as they did until round 5 instead of k_bucket_sorted (tests compare both
// routes); read per call
inline bool sorted_kernel() {
    const char* e = std::getenv("CBLX_SORTED_KERNEL");
    return !(e && e[0] == '0');
}
struct Twin {
    Buf<u64> lo, hi;   // same positions as the arena; hi only for suffixes wider than 64 bits
    Buf<u8> in_twin;   // per bucket rank: 1 = its final words are in the twin
    u64 arrivals = 0;  // words of the runs that went through the twin
    u64 runs = 0;      // ... and their number
    bool used() const { return lo.get() != nullptr; }
};
__global__ void k_set_u32(u32* p, u32 v) { *p = v; }
template <typename C> void big_stage(cblx_ctx* c, const BDesc* d_list, const u32* d_list_n, u32 nbig, Resident& nr, const MergeArgs& ma, Twin& tw) {
    typedef typename C::HiT HiT;
    constexpr bool WS = C::WS;
    typedef typename std::conditional<WS, u64, NoHi>::type AH;  // hi part of an arena element as the partition kernels see it
    if (nbig == 0) return;
    const Consts& P = c->P;
    u64* a_lo = nr.a_lo.get();
    HiT* a_hi = WS ? (HiT*)nr.a_hi.get() : (HiT*)nullptr;
    if (!tw.used()) {
        const u64 T = d2h<u64>(c, nr.start.get() + nr.nb);  // every run position lies below the total arrival count
        try {
            tw.lo = Buf<u64>(c->pool, T + 2);
            if (WS) tw.hi = Buf<u64>(c->pool, T + 2);
            tw.in_twin = Buf<u8>(c->pool, nr.nb + 1);
        } catch (const Error& e) {
            // no room for a second arena (an index that fills most of the HBM): the long runs take the general kernel, whose
            // scratch is only as large as the runs themselves
            if (e.code != CBLX_ENOMEM) throw;
            tw = Twin();
            huge_stage<C>(c, d_list, d_list_n, nbig, a_lo, a_hi, nr, ma);
            return;
        }
        CBLX_HIP(hipMemsetAsync(tw.in_twin.get(), 0, nr.nb + 1, c->stream));
    }
    Buf<BDesc> fb(c->pool, nbig);
    Buf<u32> fb_n(c->pool, 1);
    {
        StageTimer t(c, ST_BBIG);
        Buf<u32> ntile(c->pool, nbig), nv(c->pool, nbig), tile_first(c->pool, nbig + 1);
        Buf<u64> vb(c->pool, nbig + 1), run_start(c->pool, nbig);
        hipLaunchKernelGGL(k_big_plan, grid1(nbig, 256), dim3(256), 0, c->stream, d_list, nbig, P.SB, ntile.get(), nv.get());
        const u64 nt64 = exclusive_scan<u32>(c, ntile.get(), nbig, tile_first.get());
        const u64 vtot = exclusive_scan<u64>(c, nv.get(), nbig, vb.get());
        if (nt64 >= 0xFFFFFF00ull / 256) throw Error(CBLX_ERANGE, "too many tiles in the long runs of one batch");
        const u32 nt = (u32)nt64;
        hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, c->stream, tile_first.get() + nbig, nt);
        Buf<u64> t_start(c->pool, nt);
        Buf<u32> t_count(c->pool, nt), t_seg(c->pool, nt), counts(c->pool, (size_t)256 * nt), colpre(c->pool, (size_t)256 * nt), scratch, coltot(c->pool, 256),
            adj(c->pool, (size_t)256 * nbig), rel(c->pool, (size_t)256 * nbig);
        hipLaunchKernelGGL(k_big_tile_table, grid1(std::max<u64>(nt, nbig), 256), dim3(256), 0, c->stream, d_list, nbig, tile_first.get(), t_start.get(), t_count.get(), t_seg.get(),
                           run_start.get());
        TileView tv{nullptr, t_count.get(), nullptr, nullptr, nt, 0};
        tv.start64 = t_start.get();
        tv.seg32 = t_seg.get();
        const u32 DB = big_digit_bits(P.SB);
        const DigitBits dfn{P.SB - DB, DB};
        const AH* in_hi = (const AH*)a_hi;
        AH* out_hi = (AH*)tw.hi.get();
        hipLaunchKernelGGL((k_radix_hist<AH, DigitBits>), dim3(xcd_grid(nt)), dim3(RDX_THREADS), 0, c->stream, (const u64*)a_lo, in_hi, tv, dfn, counts.get());
        colscan(c, counts.get(), nullptr, nt, colpre.get(), coltot.get(), scratch);
        hipLaunchKernelGGL(k_seg_adjust, dim3(nbig), dim3(256), 0, c->stream, colpre.get(), coltot.get(), (const u32*)tile_first.get(), (const u32*)nullptr, (const u32*)nullptr, nt, nbig,
                           adj.get(), rel.get());
        hipLaunchKernelGGL((k_radix_scatter<AH, AH, DigitBits>), dim3(xcd_grid(nt)), dim3(RDX_THREADS), 0, c->stream, (const u64*)a_lo, in_hi, tv, dfn, (const u32*)colpre.get(),
                           (const u32*)adj.get(), tw.lo.get(), out_hi, DigitBits{0, 0}, (u8*)nullptr, (u32*)nullptr, 0u, 0u, 0u, (u32*)nullptr, 0u,
                           OwnWindow{0, 0, nullptr, nullptr, nullptr}, (const u64*)run_start.get());
        Buf<BDesc> vlist(c->pool, vtot), cls_lists(c->pool, 2 * vtot);
        Buf<u32> v_count(c->pool, vtot + 1), cls_n(c->pool, 2);
        Buf<u8> v_kind(c->pool, vtot + 1);
        Buf<BDesc> retry(c->pool, vtot);  // sub-ranges the sort gives up on (crowded sub-bucket): their runs fall back as a whole
        Buf<u32> retry_n(c->pool, 1);
        CBLX_HIP(hipMemsetAsync(retry_n.get(), 0, 4, c->stream));
        CBLX_HIP(hipMemsetAsync(cls_n.get(), 0, 8, c->stream));
        CBLX_HIP(hipMemsetAsync(fb_n.get(), 0, 4, c->stream));
        hipLaunchKernelGGL(k_big_vlist, dim3((nbig + BIG_VLIST_RUNS - 1) / BIG_VLIST_RUNS), dim3(256), 0, c->stream, d_list, nbig, vb.get(), rel.get(), P.SB, vlist.get(), v_count.get(), cls_lists.get(), cls_n.get(), vtot);
        const std::vector<u32> cn = d2h_vec<u32>(c, cls_n.get(), 2);
        HiT* th = WS ? (HiT*)tw.hi.get() : (HiT*)nullptr;
        // sub-ranges of up to 1024 words take the 128-thread workgroup — its `self |= other` instantiation without merge
        // arguments: that one keeps the top-bit sub-bucket limit (the build's 1024-word class assumes hashed sub-buckets)
        auto sort = [&](auto pk) {
            constexpr bool PK = decltype(pk)::value;
            if constexpr (PK || WS) {  // sub-ranges are always asked for the sorted list: the walk kernel (round 6)
                if (sorted_kernel()) {
                    if (cn[0])
                        hipLaunchKernelGGL((k_bucket_sorted<128, 1024, WS, HiT>), dim3(cn[0]), dim3(128), 0, c->stream, cls_lists.get(), cls_n.get() + 0, tw.lo.get(), th, P.SB, v_count.get(),
                                           v_kind.get(), retry.get(), retry_n.get(), (u8*)nullptr, (u32*)nullptr);
                    if (cn[1])
                        hipLaunchKernelGGL((k_bucket_sorted<256, BIG_VCAP, WS, HiT>), dim3(cn[1]), dim3(256), 0, c->stream, cls_lists.get() + vtot, cls_n.get() + 1, tw.lo.get(), th, P.SB,
                                           v_count.get(), v_kind.get(), retry.get(), retry_n.get(), (u8*)nullptr, (u32*)nullptr);
                    return;
                }
            }
            if (cn[0])
                hipLaunchKernelGGL((k_bucket_msd<128, 1024, PK, WS, HiT, true>), dim3(cn[0]), dim3(128), 0, c->stream, cls_lists.get(), cls_n.get() + 0, tw.lo.get(), th, P.SB, v_count.get(),
                                   v_kind.get(), retry.get(), retry_n.get(), MergeArgs{});
            if (cn[1])
                hipLaunchKernelGGL((k_bucket_msd<256, BIG_VCAP, PK, WS, HiT>), dim3(cn[1]), dim3(256), 0, c->stream, cls_lists.get() + vtot, cls_n.get() + 1, tw.lo.get(), th, P.SB, v_count.get(),
                                   v_kind.get(), retry.get(), retry_n.get(), MergeArgs{});
        };
        if constexpr (!WS) { if (P.SB + PK_BITS <= 64) sort(std::true_type()); else sort(std::false_type()); }
        else sort(std::false_type());
        hipLaunchKernelGGL((k_big_finish<WS>), dim3(nbig), dim3(256), 0, c->stream, d_list, d_list_n, vb.get(), vlist.get(), v_count.get(), P.SB, tw.lo.get(), tw.hi.get(), nr.cnt.get(),
                           nr.kind.get(), tw.in_twin.get(), fb.get(), fb_n.get());
        CBLX_HIP(hipGetLastError());
        CBLX_HIP(hipStreamSynchronize(c->stream));  // the pass tables die here
    }
    {   // arrivals of the runs that went through the twin = their lengths (the fallback runs are few)
        Buf<u64> tot(c->pool, 1);
        CBLX_HIP(hipMemsetAsync(tot.get(), 0, 8, c->stream));
        hipLaunchKernelGGL(k_sum_bdesc_len, dim3((unsigned)std::min<u64>(1024, ceil_div(nbig, 256))), dim3(256), 0, c->stream, d_list, nbig, tot.get());
        tw.arrivals += d2h<u64>(c, tot.get());
    }
    const u32 nfb = d2h<u32>(c, fb_n.get());
    tw.runs += nbig - nfb;
    huge_stage<C>(c, fb.get(), fb_n.get(), nfb, a_lo, a_hi, nr, ma);
}
// After every bucket kernel of the stage: the finished long runs sit in the twin, everything else in the arena. The buffer that
// holds more words becomes the arena; the other side's buckets are copied over (count[r] words at the same positions).
// `force`: -1 = the rule above; 0 / 1 = the result must end up in the arena / in the twin (a group of the grouped receiver, whose
// two buffers are a slot of the final arena and a scratch area: pipeline_group)
template <typename C> void finish_twin(cblx_ctx* c, Resident& nr, Twin& tw, int force = -1) {
    constexpr bool WS = C::WS;
    if (!tw.used()) return;
    if (force == 0 && tw.runs == 0) { tw = Twin(); return; }  // a preset twin nothing went through
    StageTimer t(c, ST_BBIG);
    const u64 T = d2h<u64>(c, nr.start.get() + nr.nb);
    const bool to_twin = force < 0 ? tw.arrivals * 2 > T : force == 1;
    const u64* src_lo = to_twin ? nr.a_lo.get() : tw.lo.get();
    const u64* src_hi = to_twin ? nr.a_hi.get() : tw.hi.get();
    u64* dst_lo = to_twin ? tw.lo.get() : nr.a_lo.get();
    u64* dst_hi = to_twin ? tw.hi.get() : nr.a_hi.get();
    // lanes per bucket: the long runs take a wave each; the others follow the average bucket length
    auto copy = [&](auto lpb) {
        constexpr int LPB = decltype(lpb)::value;
        hipLaunchKernelGGL((k_copy_buckets<WS, LPB>), lpb_grid(nr.nb, LPB), dim3(256), 0, c->stream, nr.nb, nr.start.get(), nr.cnt.get(), tw.in_twin.get(), to_twin ? 0u : 1u, src_lo, src_hi,
                           dst_lo, dst_hi);
    };
    if (to_twin) with_lpb(T - tw.arrivals, nr.nb - std::min<u64>(tw.runs, nr.nb), copy); else copy(std::integral_constant<int, 64>());
    CBLX_HIP(hipGetLastError());
    CBLX_HIP(hipStreamSynchronize(c->stream));
    if (to_twin) {
        nr.a_lo = std::move(tw.lo);
        if (WS) nr.a_hi = std::move(tw.hi);
    }
    tw = Twin();
}

// KRN-3 over the runs of `nr` (run of a prefix = [its resident suffixes as stored][the new words in stream order]) in the
// arena a_lo / a_hi: per-bucket dedup / sort by size class; fills nr.cnt, nr.kind, nr.count. `old` = the resident index
// the runs were built against (tells which buckets are untouched and which are Tries already).
// CBLX_SPANS=1: the clean-span pre-filter in front of the classification (kernels_bucket.hpp: k_bucket_span). OFF by default — measured (round 6, one
// MI355X): cfg 3 38.71 against 38.80 ms, cfg 4 50.9 against 51.6, cfg 2 +0.4 ms, 30 x coverage +2.4 ms: a span's life is its chain of dependent HBM
// round trips (head -> first start -> flags, starts and words) just as a bucket's is, and eight 1 500-word spans per CU keep no more bytes in flight
// per microsecond than thirty-two 77-word waves (DESIGN_HISTORY.md §3.13). Tests run both routes; read per call
inline bool spans_enabled() {
    const char* e = std::getenv("CBLX_SPANS");
    return e && e[0] == '1';
}
// CBLX_REPEAT_PREPASS=0 switches the pre-pass of the long runs off (tests compare both routes); read per call
bool repeat_prepass() {
    const char* e = std::getenv("CBLX_REPEAT_PREPASS");
    return !(e && e[0] == '0');
}
// `preset` (optional): the twin buffer of the long-run path, supplied by the caller (lo / hi set, same positions as the arena) instead of
// allocated here; `force`: see finish_twin
template <typename C> void bucket_stage(cblx_ctx* c, Resident& nr, const DirView& old, Twin* preset = nullptr, int force = -1) {
    typedef typename C::HiT HiT;
    const Consts& P = c->P;
    u64* a_lo = nr.a_lo.get();  // the arena (the long runs may move it to a twin buffer at the end: finish_twin)
    HiT* a_hi = C::WS ? (HiT*)nr.a_hi.get() : (HiT*)nullptr;
    Twin tw;
    if (preset) {
        tw = std::move(*preset);
        tw.in_twin = Buf<u8>(c->pool, nr.nb + 1);
        CBLX_HIP(hipMemsetAsync(tw.in_twin.get(), 0, nr.nb + 1, c->stream));
    }
    {
    const u64 nb = nr.nb;
    Buf<BDesc> lists(c->pool, (size_t)CLS_N * std::max<u64>(nb, 1));
    Buf<u32> list_n(c->pool, CLS_N);
    Buf<u32> res_count(c->pool, nb + 1);
    Buf<u8> res_kind(c->pool, nb + 1);
    CBLX_HIP(hipMemsetAsync(list_n.get(), 0, CLS_N * 4, c->stream));
    // (suffixes wider than 64 bits: an element takes 18 bytes of LDS, the 4096-word workgroup 82 KB = one per CU, and the Trie
    // classes cost 60 ps per word on 2000-word buckets against 10 on short ones. Sending runs over 2048 / 1024 words down the
    // long-run path instead was measured — CBLX_LDS_MAX_WS — and is slower: 134 -> 143 / 165 ms per 1.2 G words at K = 59, the
    // cost is the ranking inside clusters of up to K mates, whatever the workgroup)
    static const u32 lds_max_ws = [] { const char* e = std::getenv("CBLX_LDS_MAX_WS"); const u32 v = e ? (u32)std::strtoul(e, nullptr, 10) : 0; return v ? v : 4096u; }();
    // clean spans (kernels_bucket.hpp: k_bucket_span; a measured switch, off by default): on an empty index, stretches of consecutive short runs
    // without a single repeat are settled by one workgroup each before anything is classified
    Buf<u8> span_done;
    if ((old.bv == nullptr || old.nb == 0) && nb >= SPAN_MIN_RUNS && spans_enabled()) {
        StageTimer t(c, ST_BSMALL);
        span_done = Buf<u8>(c->pool, nb + 1);
        Buf<u32> heads(c->pool, nb + 1), nheads(c->pool, 1);
        Buf<u8> cont(c->pool, nb + 1);
        CBLX_HIP(hipMemsetAsync(span_done.get(), 0, nb + 1, c->stream));
        CBLX_HIP(hipMemsetAsync(nheads.get(), 0, 4, c->stream));
        hipLaunchKernelGGL(k_span_heads, grid1(nb, CLASSIFY_THREADS), dim3(CLASSIFY_THREADS), 0, c->stream, nb, (const u64*)nr.start.get(), heads.get(), nheads.get(), cont.get());
        const u32 nh = d2h<u32>(c, nheads.get());
        if (nh)
            hipLaunchKernelGGL((k_bucket_span<C::WS, HiT>), dim3(nh), dim3(SPAN_THREADS), 0, c->stream, (const u32*)heads.get(), (const u32*)nheads.get(), nb, (const u64*)nr.start.get(),
                               (const u8*)cont.get(), (const u64*)a_lo, (const HiT*)a_hi, P.SB, nr.cnt.get(), nr.kind.get(), span_done.get());
        CBLX_HIP(hipGetLastError());
        CBLX_HIP(hipStreamSynchronize(c->stream));  // heads die here
    }
    hipLaunchKernelGGL(k_classify, grid1(nb, CLASSIFY_THREADS), dim3(CLASSIFY_THREADS), 0, c->stream, nb, C::WS ? lds_max_ws : 4096u, nr.prefix.get(), nr.start.get(), old,
                       res_count.get(), res_kind.get(), nr.cnt.get(), nr.kind.get(), lists.get(), list_n.get(), (const u8*)span_done.get());
    std::vector<u32> ln = d2h_vec<u32>(c, list_n.get(), CLS_N);
    if (ln[CLS_S32] | ln[CLS_S16]) {
        StageTimer t(c, ST_BSMALL);
        if (ln[CLS_S16])
            hipLaunchKernelGGL((k_bucket_small<16, C::WS, HiT>), grid1((u64)ln[CLS_S16] * 16, 256), dim3(256), 0, c->stream,
                               lists.get() + (size_t)CLS_S16 * nb, list_n.get() + CLS_S16, a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get());
        if (ln[CLS_S32])
            hipLaunchKernelGGL((k_bucket_small<32, C::WS, HiT>), grid1((u64)ln[CLS_S32] * 32, 256), dim3(256), 0, c->stream,
                               lists.get() + (size_t)CLS_S32 * nb, list_n.get() + CLS_S32, a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get());
    }
    // whether the batch looks like one full of repeats: some shorter run gave up in the counting sort (unknown = yes when
    // there are no shorter runs at all). Long runs of distinct words — the receiving side of a many-GPU build — then skip
    // the pre-pass below altogether.
    bool saw_repeats = true;
    {
        StageTimer t(c, ST_BMED);
        // fast path (counting sort on the top suffix bits + in-sub-bucket ranking); a run with a crowded sub-bucket marks its
        // list entry and takes the claim-table kernel of its length class (runs full of repeats); what needs the sorted layout
        // after that goes to the LDS radix sort
        constexpr int NM = 6;
        static const int MCLS[NM] = {CLS_M16, CLS_M32, CLS_M64, CLS_M128, CLS_M256, CLS_M512};
        u64 roff[NM + 1] = {0};
        for (int k = 0; k < NM; ++k) roff[k + 1] = roff[k] + ln[MCLS[k]];
        Buf<u8> bail(c->pool, roff[NM] + 8);
        Buf<u32> bail_any(c->pool, NM + 1);  // one word per class, then the radix kernel's list counter
        Buf<BDesc> retry2(c->pool, std::max<u64>(roff[NM], 1));
        CBLX_HIP(hipMemsetAsync(bail.get(), 0, roff[NM] + 8, c->stream));
        CBLX_HIP(hipMemsetAsync(bail_any.get(), 0, (NM + 1) * 4, c->stream));
        u32* r2n = bail_any.get() + NM;
        // The classes up to 1024 words (hashed sub-buckets) never give up without repeats in the batch: once one of them did
        // — the shortest is asked first — the classes after it start with the claim table (`repeat_mode`) instead of a
        // counting sort that is going to give up.
        bool repeat_mode = false;
        std::vector<u32> any(NM, 0u);
        bool used_msd[NM] = {false};
        auto any_of = [&](const std::vector<u32>& v) { u32 a = 0; for (u32 x : v) a |= x; return a != 0; };
        // suffixes too wide for the counting-sort kernel's 16-byte elements (SUFFIX_BITS > 116: K >= 57 with a handful of prefix
        // bits): every class takes the LDS radix kernel
        const bool radix_only = !msd_takes<C::WS>(P.SB);
        if (radix_only) {
            for (int k = 0; k < NM; ++k)
                if (ln[MCLS[k]])
                    hipLaunchKernelGGL((k_bucket_medium<512, C::WS, HiT>), dim3(ln[MCLS[k]]), dim3(512), 0, c->stream, lists.get() + (size_t)MCLS[k] * nb, list_n.get() + MCLS[k], a_lo, a_hi,
                                       P.SB, nr.cnt.get(), nr.kind.get(), MergeArgs{});
            CBLX_HIP(hipGetLastError());
        }
        const bool use_sorted = sorted_kernel();
        auto stage = [&](auto packed_tag) {
            constexpr bool PK = decltype(packed_tag)::value;
            if (radix_only) return;
            auto go = [&](auto thr, auto cap, int k) {
                constexpr int T = decltype(thr)::value, CAPV = decltype(cap)::value;
                const int cls = MCLS[k];
                if (!ln[cls]) return;
                if ((CBLX_CLAIM_FIRST && CAPV <= (int)VEC_THRESHOLD) || repeat_mode) {
                    hipLaunchKernelGGL((k_bucket_claim<T, CAPV, C::WS, HiT>), dim3(ln[cls]), dim3(T), 0, c->stream, lists.get() + (size_t)cls * nb, list_n.get() + cls, a_lo, a_hi, P.SB,
                                       nr.cnt.get(), nr.kind.get(), retry2.get(), r2n, (const u8*)nullptr);
                    return;
                }
                used_msd[k] = true;
                if constexpr ((PK || C::WS) && CAPV > (int)VEC_THRESHOLD) {  // runs of more than 1024 words end up sorted (or give up: repeats): the walk kernel (round 6)
                    if (use_sorted) {
                        hipLaunchKernelGGL((k_bucket_sorted<T, CAPV, C::WS, HiT>), dim3(ln[cls]), dim3(T), 0, c->stream, lists.get() + (size_t)cls * nb, list_n.get() + cls, a_lo, a_hi, P.SB,
                                           nr.cnt.get(), nr.kind.get(), (BDesc*)nullptr, (u32*)nullptr, bail.get() + roff[k], bail_any.get() + k);
                        return;
                    }
                }
                hipLaunchKernelGGL((k_bucket_msd<T, CAPV, PK, C::WS, HiT>), dim3(ln[cls]), dim3(T), 0, c->stream, lists.get() + (size_t)cls * nb, list_n.get() + cls, a_lo, a_hi, P.SB,
                                   nr.cnt.get(), nr.kind.get(), (BDesc*)nullptr, (u32*)nullptr, MergeArgs{}, bail.get() + roff[k], bail_any.get() + k);
            };
            go(std::integral_constant<int, 64>(), std::integral_constant<int, 128>(), 0);
            if (used_msd[0] && (ln[CLS_M32] | ln[CLS_M64] | ln[CLS_M128] | ln[CLS_M256] | ln[CLS_M512]))  // the shortest class is the cheapest witness
                repeat_mode = d2h<u32>(c, bail_any.get() + 0) != 0;
            go(std::integral_constant<int, 64>(), std::integral_constant<int, 256>(), 1);
            go(std::integral_constant<int, 64>(), std::integral_constant<int, 512>(), 2);
            go(std::integral_constant<int, 128>(), std::integral_constant<int, 1024>(), 3);
            if (!repeat_mode && (ln[CLS_M256] | ln[CLS_M512]) && (used_msd[1] | used_msd[2] | used_msd[3]))
                repeat_mode = any_of(d2h_vec<u32>(c, bail_any.get(), 4));
            go(std::integral_constant<int, 256>(), std::integral_constant<int, 2048>(), 4);
            go(std::integral_constant<int, 512>(), std::integral_constant<int, 4096>(), 5);
            if (!roff[NM]) return;
            any = d2h_vec<u32>(c, bail_any.get(), NM);
            auto claim = [&](auto thr, auto cap, int k) {
                constexpr int T = decltype(thr)::value, CAPV = decltype(cap)::value;
                const int cls = MCLS[k];
                if (!used_msd[k] || !any[k]) return;
                hipLaunchKernelGGL((k_bucket_claim<T, CAPV, C::WS, HiT>), dim3((ln[cls] + 63) / 64), dim3(T), 0, c->stream, lists.get() + (size_t)cls * nb, list_n.get() + cls, a_lo, a_hi,
                                   P.SB, nr.cnt.get(), nr.kind.get(), retry2.get(), r2n, (const u8*)(bail.get() + roff[k]));
            };
            claim(std::integral_constant<int, 64>(), std::integral_constant<int, 128>(), 0);
            claim(std::integral_constant<int, 64>(), std::integral_constant<int, 256>(), 1);
            claim(std::integral_constant<int, 64>(), std::integral_constant<int, 512>(), 2);
            claim(std::integral_constant<int, 128>(), std::integral_constant<int, 1024>(), 3);
            claim(std::integral_constant<int, 256>(), std::integral_constant<int, 2048>(), 4);
            claim(std::integral_constant<int, 512>(), std::integral_constant<int, 4096>(), 5);
            if (CBLX_CLAIM_FIRST || repeat_mode || any_of(any)) {
                // what the claim tables left for the sorted layout: distinct words now, at most 4096 per run
                const u32 n2 = d2h<u32>(c, r2n);
                if (n2) {
                    Buf<BDesc> retry3(c->pool, n2);
                    Buf<u32> r3n(c->pool, 1);
                    CBLX_HIP(hipMemsetAsync(r3n.get(), 0, 4, c->stream));
                    bool done2 = false;
                    if constexpr (PK || C::WS) {
                        if (use_sorted) {
                            hipLaunchKernelGGL((k_bucket_sorted<512, 4096, C::WS, HiT>), dim3(n2), dim3(512), 0, c->stream, retry2.get(), r2n, a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), retry3.get(),
                                               r3n.get(), (u8*)nullptr, (u32*)nullptr);
                            done2 = true;
                        }
                    }
                    if (!done2)
                    hipLaunchKernelGGL((k_bucket_msd<512, 4096, PK, C::WS, HiT>), dim3(n2), dim3(512), 0, c->stream, retry2.get(), r2n, a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(),
                                       retry3.get(), r3n.get(), MergeArgs{});
                    const u32 n3 = d2h<u32>(c, r3n.get());
                    if (n3)
                        hipLaunchKernelGGL((k_bucket_medium<512, C::WS, HiT>), dim3(n3), dim3(512), 0, c->stream, retry3.get(), r3n.get(), a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), MergeArgs{});
                    CBLX_HIP(hipStreamSynchronize(c->stream));  // retry3 dies here
                }
            }
        };
        if constexpr (!C::WS) {
            if (P.SB + PK_BITS <= 64) stage(std::true_type()); else stage(std::false_type());
        } else {
            stage(std::false_type());
        }
        saw_repeats = roff[NM] ? repeat_mode : true;  // no shorter runs at all: unknown = yes
        if (roff[NM] && !repeat_mode && (ln[CLS_BIG] | ln[CLS_HUGE]) && any_of(any)) {
            // no shorter hashed runs to tell: a sizeable share of the runs that went through the counting sort must have given up
            // (a few do without a single repeat — necklace clusters)
            Buf<u64> nbail(c->pool, 1);
            CBLX_HIP(hipMemsetAsync(nbail.get(), 0, 8, c->stream));
            hipLaunchKernelGGL(k_sum_u8, dim3((unsigned)std::min<u64>(1024, ceil_div(roff[NM], 256))), dim3(256), 0, c->stream, bail.get(), roff[NM], nbail.get());
            saw_repeats = d2h<u64>(c, nbail.get()) * 8 >= roff[NM];
        }
        CBLX_HIP(hipStreamSynchronize(c->stream));  // retry buffers die here
    }
    bool long_done = false;
    if constexpr (!C::WS) {
        if (P.SB < 64 && ln[CLS_BIG] + ln[CLS_HUGE] > 0 && saw_repeats && repeat_prepass()) {
            // runs too long for one workgroup's sort: first the pre-pass that shrinks the ones full of repeats to their first
            // occurrences (k_big_claim); what it passes on takes the paths below, what it shrank to more than 1024 distinct
            // words is sorted by one workgroup
            const u32 nbig = ln[CLS_BIG], nhuge = ln[CLS_HUGE];
            Buf<BDesc> next_big(c->pool, std::max<u32>(nbig, 1)), next_huge(c->pool, std::max<u32>(nhuge, 1)), srt(c->pool, (size_t)nbig + nhuge), retry(c->pool, (size_t)nbig + nhuge);
            Buf<u32> cnts(c->pool, 4);  // next_big, next_huge, sorted, retry
            std::vector<u32> cn;
            {
            StageTimer t(c, ST_BBIG);
            CBLX_HIP(hipMemsetAsync(cnts.get(), 0, 16, c->stream));
            if (nbig)
                hipLaunchKernelGGL((k_big_claim<HiT>), dim3(nbig), dim3(BCL_THREADS), 0, c->stream, lists.get() + (size_t)CLS_BIG * nb, list_n.get() + CLS_BIG, a_lo, P.SB, nr.cnt.get(),
                                   nr.kind.get(), next_big.get(), cnts.get() + 0, srt.get(), cnts.get() + 2);
            if (nhuge)
                hipLaunchKernelGGL((k_big_claim<HiT>), dim3(nhuge), dim3(BCL_THREADS), 0, c->stream, lists.get() + (size_t)CLS_HUGE * nb, list_n.get() + CLS_HUGE, a_lo, P.SB, nr.cnt.get(),
                                   nr.kind.get(), next_huge.get(), cnts.get() + 1, srt.get(), cnts.get() + 2);
            cn = d2h_vec<u32>(c, cnts.get(), 3);
            if (cn[2]) {
                auto sort = [&](auto pk) {
                    if constexpr (decltype(pk)::value) {
                        if (sorted_kernel()) {
                            hipLaunchKernelGGL((k_bucket_sorted<256, 2048, false, HiT>), dim3(cn[2]), dim3(256), 0, c->stream, srt.get(), cnts.get() + 2, a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(),
                                               retry.get(), cnts.get() + 3, (u8*)nullptr, (u32*)nullptr);
                            return;
                        }
                    }
                    hipLaunchKernelGGL((k_bucket_msd<256, 2048, decltype(pk)::value, false, HiT>), dim3(cn[2]), dim3(256), 0, c->stream, srt.get(), cnts.get() + 2, a_lo, a_hi, P.SB,
                                       nr.cnt.get(), nr.kind.get(), retry.get(), cnts.get() + 3, MergeArgs{});
                };
                if (P.SB + PK_BITS <= 64) sort(std::true_type()); else sort(std::false_type());
                const u32 nre = d2h<u32>(c, cnts.get() + 3);
                if (nre)
                    hipLaunchKernelGGL((k_bucket_medium<512, false, HiT>), dim3(nre), dim3(512), 0, c->stream, retry.get(), cnts.get() + 3, a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), MergeArgs{});
            }
            CBLX_HIP(hipGetLastError());
            }
            big_stage<C>(c, next_big.get(), cnts.get() + 0, cn[0], nr, MergeArgs{}, tw);
            huge_stage<C>(c, next_huge.get(), cnts.get() + 1, cn[1], a_lo, a_hi, nr, MergeArgs{});
            CBLX_HIP(hipStreamSynchronize(c->stream));  // the lists die here
            long_done = true;
        }
    }
    if (!long_done) {
        if (msd_takes<C::WS>(P.SB)) big_stage<C>(c, lists.get() + (size_t)CLS_BIG * nb, list_n.get() + CLS_BIG, ln[CLS_BIG], nr, MergeArgs{}, tw);
        else huge_stage<C>(c, lists.get() + (size_t)CLS_BIG * nb, list_n.get() + CLS_BIG, ln[CLS_BIG], a_lo, a_hi, nr, MergeArgs{});
        huge_stage<C>(c, lists.get() + (size_t)CLS_HUGE * nb, list_n.get() + CLS_HUGE, ln[CLS_HUGE], a_lo, a_hi, nr, MergeArgs{});
    }
    }
    CBLX_HIP(hipGetLastError());
    finish_twin<C>(c, nr, tw, force);
    {
        Buf<u64> total(c->pool, 1);
        CBLX_HIP(hipMemsetAsync(total.get(), 0, 8, c->stream));
        hipLaunchKernelGGL(k_sum_u32, dim3((unsigned)std::min<u64>(2048, std::max<u64>(1, ceil_div(nr.nb, 256)))), dim3(256), 0, c->stream, nr.cnt.get(), nr.nb, total.get());
        nr.count = d2h<u64>(c, total.get());
    }
}

// rows of k_merge_table's `other` side for a freshly partitioned batch: every run is a Vec of its raw length
__global__ void k_run_lengths(u64 nb, const u64* __restrict__ start, u32* __restrict__ cnt, u8* __restrict__ kind) {
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nb) return;
    cnt[r] = (u32)(start[r + 1] - start[r]);
    kind[r] = KIND_VEC;
}

// One batch of N new words (rec, first n_pre slots unused = 0) into the index.
//   Empty index: partition + buckets, the sorted record array becomes the arena.
//   Non-empty index: only the NEW words are partitioned; the resident buckets are already grouped by prefix, so the merged
//   directory is the OR of the two bitvectors and every merged run = [resident suffixes as stored][new words of the
//   prefix] is gathered straight from the two arrays (the resident words are never expanded and re-partitioned).
template <typename C> void pipeline(cblx_ctx* c, Records& rec, u64 N, Buf<u32> countsA = Buf<u32>(), const PieceInput* pin = nullptr) {
    typedef typename C::HiT HiT;
    constexpr bool WS = C::WS;
    const Consts& P = c->P;
    Resident nb_;  // directory of the batch
    partition_and_directory<C>(c, rec, N, std::move(countsA), nb_, pin);
    auto adopt_arena = [&](Resident& nr) {
        nr.a_lo = std::move(rec.lo);
        if (WS) {  // arena hi lives in the records' hi buffer (u64 elements in this configuration)
            nr.a_hi.pool = rec.hi.pool; nr.a_hi.p = (u64*)rec.hi.p; nr.a_hi.n = rec.hi.n / 8;
            rec.hi.p = nullptr; rec.hi.n = 0;
        } else {
            rec.hi.reset();
        }
    };
    if (c->res.count == 0) {
        adopt_arena(nb_);
        bucket_stage<C>(c, nb_, c->res.view());
        c->res = std::move(nb_);
        return;
    }
    const Resident& s = c->res;
    const u64 nprefix = 1ull << P.PB, nwords = std::max<u64>(1, nprefix / 64);
    hipLaunchKernelGGL(k_run_lengths, grid1(nb_.nb, 256), dim3(256), 0, c->stream, nb_.nb, nb_.start.get(), nb_.cnt.get(), nb_.kind.get());
    adopt_arena(nb_);  // the sorted batch plays `other` in the gather below
    Resident nr;
    Buf<u32> raw, m_cs;
    Buf<u64> m_sstart, m_ostart;
    Buf<u8> m_skind, m_okind;
    u64 T = 0;
    {
        StageTimer t(c, ST_DIR);
        Buf<u32> popc(c->pool, nwords);
        nr.bv = Buf<u64>(c->pool, nwords);
        nr.rank_dir = Buf<u64>(c->pool, nwords + 1);
        hipLaunchKernelGGL(k_bv_or, grid1(nwords, 256), dim3(256), 0, c->stream, nwords, s.bv.get(), nb_.bv.get(), nr.bv.get(), popc.get());
        nr.nb = exclusive_scan<u64>(c, popc.get(), nwords, nr.rank_dir.get());
        const u64 nb = nr.nb;
        nr.prefix = Buf<u32>(c->pool, nb + 1);
        nr.start = Buf<u64>(c->pool, nb + 1);
        nr.cnt = Buf<u32>(c->pool, nb + 1);
        nr.kind = Buf<u8>(c->pool, nb + 1);
        raw = Buf<u32>(c->pool, nb + 1);
        m_cs = Buf<u32>(c->pool, nb + 1);
        m_sstart = Buf<u64>(c->pool, nb + 1);
        m_ostart = Buf<u64>(c->pool, nb + 1);
        m_skind = Buf<u8>(c->pool, nb + 1);
        m_okind = Buf<u8>(c->pool, nb + 1);
        hipLaunchKernelGGL(k_merge_table, grid1(nprefix, 256), dim3(256), 0, c->stream, nprefix, nr.bv.get(), nr.rank_dir.get(), s.view(), nb_.view(), nr.prefix.get(),
                           raw.get(), m_cs.get(), m_sstart.get(), m_ostart.get(), m_skind.get(), m_okind.get());
        T = exclusive_scan<u64>(c, raw.get(), nb, nr.start.get());
        hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, nr.start.get() + nb, T);
        CBLX_HIP(hipGetLastError());
    }
    if (T != s.count + N) throw Error(CBLX_EDEVICE, "insert: run lengths do not match the index and the batch (internal error)");
    nr.a_lo = Buf<u64>(c->pool, T + 2);
    if (WS) nr.a_hi = Buf<u64>(c->pool, T + 2);
    {
        StageTimer t(c, ST_EXPAND);
        with_lpb(T, nr.nb, [&](auto lpb) {
            constexpr int LPB = decltype(lpb)::value;
            hipLaunchKernelGGL((k_merge_gather<WS, LPB>), lpb_grid(nr.nb, LPB), dim3(256), 0, c->stream, nr.nb, nr.start.get(), m_cs.get(), m_sstart.get(), m_ostart.get(),
                               s.a_lo.get(), s.a_hi.get(), nb_.a_lo.get(), nb_.a_hi.get(), nr.a_lo.get(), nr.a_hi.get());
        });
        CBLX_HIP(hipGetLastError());
    }
    bucket_stage<C>(c, nr, s.view());
    CBLX_HIP(hipStreamSynchronize(c->stream));  // the batch and the table buffers are released at scope exit
    c->res = std::move(nr);
}

// ---- one GROUP of the grouped receiver (comm.hpp: sharded_insert_grouped) ----------------------------------------------------------
// The records of the group's prefix window lie in pieces of the receive log (pass A done by the senders; `pin` describes the group's
// share of every piece). They go through the LSD passes, the window's directory and the bucket kernels while later groups are still
// on the wire, and end up in the group's SLOT of the final arena (`fin`, the slot's first element); `scr` is a scratch area of at
// least N + 2 elements that plays the other ping-pong buffer (and the twin of the long-run path). Positions in `out.start` are
// relative to the slot. A non-owning Buf (pool = nullptr) stands for a region of somebody else's allocation.
template <typename T> Buf<T> alias_buf(T* p, size_t n) { Buf<T> b; b.pool = nullptr; b.p = p; b.n = n; return b; }
struct GroupRegions {
    const u64* log_lo = nullptr; const void* log_hi = nullptr;   // receive log (hi: words that keep their hi part behind pass A)
    u64* fin_lo = nullptr; u64* fin_hi = nullptr;                // the group's slot in the final arena (hi: 8-byte elements)
    u64* scr_lo = nullptr; u64* scr_hi = nullptr;                // scratch, >= N + 2 elements
};
template <typename C> void pipeline_group(cblx_ctx* c, const GroupRegions& R, const PieceInput& pin, u64 N, const DirWindow& win, Resident& out) {
    typedef typename C::HiT HiT;
    constexpr bool WS = C::WS;
    constexpr bool DROP_HI = std::is_same<HiT, u8>::value;
    constexpr bool KEEP_HI = HiTraits<HiT>::has && !DROP_HI;  // 16-byte records through the LSD passes
    const Consts& P = c->P;
    const u32 npass = pin.sort_bits ? lsd_plan_bits(pin.sort_bits).total() : lsd_plan(P, false).total();  // buffer changes behind pass A (records in pieces: LSD passes only, no prefix split)
    // DEEP group (thousands of words per possible prefix: the dense low ranges of a many-GPU job at PREFIX_BITS <= 24): nearly every run
    // takes the long-run path, whose output is the twin — so the LSD passes end in the scratch area and the twin IS the slot.
    // (CBLX_GROUP_DEEP = 0 / 1 forces the choice: tests run small groups through both layouts)
    const char* deep_env = std::getenv("CBLX_GROUP_DEEP");
    const bool deep = msd_takes<WS>(P.SB) && (deep_env ? deep_env[0] == '1' : N / std::max<u64>(1, (u64)win.w_hi - win.w_lo) >= 2048);
    u64* last_lo = deep ? R.scr_lo : R.fin_lo;   // where the last LSD pass writes
    u64* last_hi = deep ? R.scr_hi : R.fin_hi;
    u64* oth_lo = deep ? R.fin_lo : R.scr_lo;
    u64* oth_hi = deep ? R.fin_hi : R.scr_hi;
    Records rec;
    rec.ext_lo = R.log_lo;
    rec.ext_hi = R.log_hi;
    // with an external source the passes write rec.lo, rec.lo2, rec.lo, ...: the last of `npass` lands in rec.lo iff npass is odd
    const bool last_is_first = (npass & 1u) != 0;
    rec.lo = alias_buf<u64>(last_is_first ? last_lo : oth_lo, N + 2);
    rec.lo2 = alias_buf<u64>(last_is_first ? oth_lo : last_lo, N + 2);
    if (KEEP_HI) {
        rec.hi = alias_buf<u8>((u8*)(last_is_first ? last_hi : oth_hi), (N + 2) * 8);
        rec.hi2 = alias_buf<u8>((u8*)(last_is_first ? oth_hi : last_hi), (N + 2) * 8);
    }
    partition_and_directory<C>(c, rec, N, Buf<u32>(), out, &pin, &win);
    if (rec.lo.get() != last_lo) throw Error(CBLX_EDEVICE, "grouped receiver: the last pass landed in the wrong buffer (internal error)");
    out.a_lo = alias_buf<u64>(last_lo, N + 2);
    if (WS) out.a_hi = alias_buf<u64>(last_hi, N + 2);
    Twin tw;
    tw.lo = alias_buf<u64>(oth_lo, N + 2);
    if (WS) tw.hi = alias_buf<u64>(oth_hi, N + 2);
    bucket_stage<C>(c, out, DirView{nullptr, nullptr, nullptr, nullptr, nullptr, 0}, &tw, deep ? 1 : 0);
    if (out.a_lo.get() != R.fin_lo) throw Error(CBLX_EDEVICE, "grouped receiver: a group did not end in its slot (internal error)");
    out.a_lo = Buf<u64>();  // the slot belongs to the final arena
    out.a_hi = Buf<u64>();
}

// record buffers (ping-pong) for a batch of n_new words
template <typename C> void begin_records(cblx_ctx* c, Records& rec, u64 n_new) {
    const size_t hs = hi_elem_size(c->P);
    rec.lo = Buf<u64>(c->pool, n_new + 2);
    rec.lo2 = Buf<u64>(c->pool, n_new + 2);
    rec.hi = Buf<u8>(c->pool, hs ? (n_new + 2) * hs : 8);
    rec.hi2 = Buf<u8>(c->pool, hs ? (n_new + 2) * hs : 8);
}

// KRN-1 front end: chunk table + validity + encode. Returns the number of new words written at rec[out_base..).
struct ChunkPlan {
    u64 nchunks = 0, n_kmers = 0, total_bases = 0;
    u32 ndirty = 0;
    u64 bias = 0;  // bytes skipped in front of the slice (multiple of 16)
    Buf<u64> chunk_start, kmer_off;
    Buf<u32> chunk_len, tile_first;
    Buf<u8> dirty;
    Buf<u32> dirty_list;  // the dirty chunks, any order (ndirty of them)
};
// what a SLICE of a plan needs (sequences [seq_a, seq_b) of the planned ones): its chunks, k-mers and tiles
struct PlanSlice { u64 c_lo = 0, c_hi = 0, k_lo = 0, k_hi = 0, t_lo = 0, t_hi = 0; };
__global__ void k_plan_marks(const u64* __restrict__ chunk_base, const u64* __restrict__ marks, u32 n, u64 nchunks, const u64* __restrict__ chunk_start, const u64* __restrict__ kmer_off,
                             u64* __restrict__ out /* [n][3]: chunk, its first k-mer, its first base */) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 ch = chunk_base[marks[i]];
    out[3 * i] = ch;
    out[3 * i + 1] = kmer_off[ch];                     // (kmer_off[nchunks] = all k-mers)
    out[3 * i + 2] = ch < nchunks ? chunk_start[ch] : ~0ull;
}
// `ends`: offsets[0] and offsets[nseq] when the caller has read them already (each read is a host round trip)
// `B`: where the bases live (ASCII bytes, or the bit planes a big host batch crossed PCIe as); advanced to the slice's aligned start
// `seq_marks` / `slices`: sequence indices m[0] <= m[1] <= ... (relative to d_offsets) -> the slices [m[i], m[i+1]) of the plan, so that a caller
// that works slice by slice plans ONCE (a plan costs half a dozen host round trips)
void plan_chunks(cblx_ctx* c, BaseView& d_bases, const u64* d_offsets, u64 nseq, ChunkPlan& pl, const u64* ends = nullptr, const std::vector<u64>* seq_marks = nullptr,
                 std::vector<PlanSlice>* slices = nullptr) {
    StageTimer t(c, ST_CHUNKS);
    const Consts& P = c->P;
    // offsets may start anywhere in the buffer (a slice of a larger batch): work relative to the 16-byte aligned
    // position below offsets[0] so that the tile grid and the validity scan cover only this slice
    const u64 first = ends ? ends[0] : d2h<u64>(c, d_offsets);
    pl.bias = first & ~(u64)15;
    const u64 last = ends ? ends[1] : d2h<u64>(c, d_offsets + nseq);
    if (last < first) throw Error(CBLX_EINVAL, "offsets must be non-decreasing");
    pl.total_bases = last - pl.bias;
    d_bases.advance16(pl.bias);
    Buf<u32> nch(c->pool, nseq + 1);
    Buf<u64> err(c->pool, 2), chunk_base(c->pool, nseq + 1);
    CBLX_HIP(hipMemsetAsync(err.get(), 0, 16, c->stream));
    hipLaunchKernelGGL(k_seq_chunk_count, grid1(nseq, 256), dim3(256), 0, c->stream, d_offsets, nseq, P.K, nch.get(), err.get());
    pl.nchunks = exclusive_scan<u64>(c, nch.get(), nseq, chunk_base.get());
    std::vector<u64> e = d2h_vec<u64>(c, err.get(), 2);
    if (e[0]) throw Error(CBLX_ESHORT, "Sequence size (" + std::to_string(e[1] - 1) + ") is smaller than K (" + std::to_string(P.K) + ")");
    if (pl.nchunks >= 0xFFFFFFF0ull) throw Error(CBLX_ERANGE, "too many chunks in one batch");
    hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, chunk_base.get() + nseq, pl.nchunks);
    pl.chunk_start = Buf<u64>(c->pool, pl.nchunks + 1);
    pl.chunk_len = Buf<u32>(c->pool, pl.nchunks + 1);
    Buf<u32> chunk_nk(c->pool, pl.nchunks + 1);
    pl.dirty = Buf<u8>(c->pool, pl.nchunks + 8);
    Buf<u32> ndirty(c->pool, 1);
    hipLaunchKernelGGL(k_chunk_fill, grid1(pl.nchunks, 256), dim3(256), 0, c->stream, d_offsets, chunk_base.get(), nseq, pl.nchunks, P.K, pl.bias,
                       pl.chunk_start.get(), pl.chunk_len.get(), chunk_nk.get());
    CBLX_HIP(hipMemsetAsync(pl.dirty.get(), 0, pl.nchunks + 8, c->stream));
    CBLX_HIP(hipMemsetAsync(ndirty.get(), 0, 4, c->stream));
    hipLaunchKernelGGL(k_scan_invalid, grid1(ceil_div(pl.total_bases, 16), 256), dim3(256), 0, c->stream, d_bases, pl.total_bases,
                       pl.chunk_start.get(), pl.chunk_len.get(), pl.nchunks, pl.dirty.get(), ndirty.get());
    pl.ndirty = d2h<u32>(c, ndirty.get());  // a flag so far
    if (pl.ndirty) {
        pl.dirty_list = Buf<u32>(c->pool, pl.nchunks + 1);
        CBLX_HIP(hipMemsetAsync(ndirty.get(), 0, 4, c->stream));
        hipLaunchKernelGGL(k_dirty_list, grid1(pl.nchunks, DIRTY_LIST_THREADS), dim3(DIRTY_LIST_THREADS), 0, c->stream, pl.dirty.get(), pl.nchunks, pl.dirty_list.get(), ndirty.get());
        pl.ndirty = d2h<u32>(c, ndirty.get());  // the number of dirty chunks
        hipLaunchKernelGGL(k_dirty_count_wave, dim3((pl.ndirty + 3) / 4), dim3(256), 0, c->stream, d_bases, pl.chunk_start.get(), pl.chunk_len.get(), pl.dirty_list.get(), pl.ndirty, P.K,
                           chunk_nk.get());
    }
    pl.kmer_off = Buf<u64>(c->pool, pl.nchunks + 1);
    pl.n_kmers = exclusive_scan<u64>(c, chunk_nk.get(), pl.nchunks, pl.kmer_off.get());
    hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, pl.kmer_off.get() + pl.nchunks, pl.n_kmers);
    const u64 ntiles = ceil_div(pl.total_bases, ENC_TILE_BYTES);
    pl.tile_first = Buf<u32>(c->pool, ntiles + 2);
    hipLaunchKernelGGL(k_tile_first_chunk, grid1(ntiles + 1, 256), dim3(256), 0, c->stream, pl.chunk_start.get(), pl.nchunks, ntiles, pl.tile_first.get());
    CBLX_HIP(hipGetLastError());
    if (seq_marks && slices) {
        const u32 nm = (u32)seq_marks->size();
        Buf<u64> d_marks(c->pool, nm), d_out(c->pool, 3 * (size_t)nm);
        h2d(c, d_marks.get(), seq_marks->data(), nm);
        hipLaunchKernelGGL(k_plan_marks, grid1(nm, 64), dim3(64), 0, c->stream, (const u64*)chunk_base.get(), (const u64*)d_marks.get(), nm, pl.nchunks, (const u64*)pl.chunk_start.get(),
                           (const u64*)pl.kmer_off.get(), d_out.get());
        const std::vector<u64> o = d2h_vec<u64>(c, d_out.get(), 3 * (size_t)nm);
        slices->assign(nm ? nm - 1 : 0, PlanSlice());
        for (u32 i = 0; i + 1 < nm; ++i) {
            PlanSlice& S = (*slices)[i];
            S.c_lo = o[3 * i]; S.c_hi = o[3 * (i + 1)]; S.k_lo = o[3 * i + 1]; S.k_hi = o[3 * (i + 1) + 1];
            if (S.c_hi > S.c_lo) {  // tiles of the base stream (4 KiB each) that hold the slice's chunk starts
                const std::vector<u64> last = d2h_vec<u64>(c, pl.chunk_start.get() + (S.c_hi - 1), 1);
                S.t_lo = o[3 * i + 2] / ENC_TILE_BYTES;
                S.t_hi = last[0] / ENC_TILE_BYTES + 1;
            }
        }
    }
    CBLX_HIP(hipStreamSynchronize(c->stream));  // temporaries (nch, err, chunk_base, chunk_nk, ndirty) die here
}
void plan_chunks(cblx_ctx* c, const u8*& d_bases, const u64* d_offsets, u64 nseq, ChunkPlan& pl, const u64* ends = nullptr) {
    BaseView B = ascii_view(d_bases);
    plan_chunks(c, B, d_offsets, nseq, pl, ends);
    d_bases = B.ascii;
}
// `sl`: only that slice of the plan (its outputs start at out_base: k-mer k of the plan goes to out_base + k - sl->k_lo)
template <typename C> void encode(cblx_ctx* c, const BaseView& d_bases, const ChunkPlan& pl, u64* out_lo, typename C::HiT* out_hi, u64 out_base,
                                  EncHist eh = EncHist{}, const PlanSlice* sl = nullptr) {
    typedef typename C::HiT HiT;
    StageTimer t(c, ST_ENCODE);
    if (sl) {
        if (sl->c_hi <= sl->c_lo) return;
        const u64 nt = sl->t_hi - sl->t_lo, ob = out_base - sl->k_lo;  // (modulo 2^64: the kernels add a chunk's k-mer offset back)
        hipLaunchKernelGGL((k_encode<C::WIDE, HiT>), dim3((unsigned)nt), dim3(ENC_THREADS), (eh.counts && eh.cut_tab) ? 8192 : 0, c->stream, d_bases, pl.total_bases, pl.chunk_start.get(),
                           pl.chunk_len.get(), pl.kmer_off.get(), pl.ndirty ? pl.dirty.get() : (const u8*)nullptr, pl.tile_first.get() + sl->t_lo, c->P, out_lo, out_hi, ob, eh, (u32)sl->c_lo,
                           (u32)sl->c_hi);
        if (pl.ndirty)
            hipLaunchKernelGGL((k_encode_dirty_wave<C::WIDE, HiT>), dim3((pl.ndirty + 3) / 4), dim3(256), 0, c->stream, d_bases, pl.chunk_start.get(), pl.chunk_len.get(),
                               pl.kmer_off.get(), pl.dirty_list.get(), pl.ndirty, c->P, out_lo, out_hi, ob, eh, (u32)sl->c_lo, (u32)sl->c_hi);
        CBLX_HIP(hipGetLastError());
        return;
    }
    const u64 ntiles = ceil_div(pl.total_bases, ENC_TILE_BYTES);
    if (ntiles)  // (dynamic LDS: the cut table of the fused histogram's bins, when there is one — 8 KB more per workgroup cost the kernel nothing, measured)
        hipLaunchKernelGGL((k_encode<C::WIDE, HiT>), dim3((unsigned)ntiles), dim3(ENC_THREADS), (eh.counts && eh.cut_tab) ? 8192 : 0, c->stream, d_bases, pl.total_bases, pl.chunk_start.get(),
                           pl.chunk_len.get(), pl.kmer_off.get(), pl.ndirty ? pl.dirty.get() : (const u8*)nullptr, pl.tile_first.get(), c->P, out_lo, out_hi, out_base, eh);
    if (pl.ndirty)
        hipLaunchKernelGGL((k_encode_dirty_wave<C::WIDE, HiT>), dim3((pl.ndirty + 3) / 4), dim3(256), 0, c->stream, d_bases, pl.chunk_start.get(), pl.chunk_len.get(),
                           pl.kmer_off.get(), pl.dirty_list.get(), pl.ndirty, c->P, out_lo, out_hi, out_base, eh);
    CBLX_HIP(hipGetLastError());
}
template <typename C> void encode(cblx_ctx* c, const u8* d_bases, const ChunkPlan& pl, u64* out_lo, typename C::HiT* out_hi, u64 out_base,
                                  EncHist eh = EncHist{}) {
    encode<C>(c, ascii_view(d_bases), pl, out_lo, out_hi, out_base, eh);
}

void check_aligned16(const void* p, const char* what) {
    if (((uintptr_t)p) & 15) throw Error(CBLX_EINVAL, std::string(what) + " must be 16-byte aligned");
}

// One call of the kernels takes fewer than 2^32 words (32-bit positions inside a batch: tile tables, start_dense). A larger
// insert is cut at sequence boundaries into sub-batches that go in one after the other: the result is the same (batch
// boundaries do not change what an insert-only history builds), the resident index itself has
// 64-bit positions throughout. CBLX_BATCH_MAX_BASES overrides the cut (tests use a tiny value).
u64 batch_max_bases() {
    const char* e = std::getenv("CBLX_BATCH_MAX_BASES");
    const u64 x = e ? std::strtoull(e, nullptr, 10) : 0;
    return x ? x : (1ull << 31);
}
void insert_device_one(cblx_ctx* c, const u8* d_bases, const u64* d_offsets, u64 nseq, const u64* ends = nullptr);
// PREFIX_BITS > 24 on an empty index: the build on FINE bins (comm.hpp; false = not taken, nothing was touched)
bool insert_device_fine(cblx_ctx* c, const u8* d_bases, const u64* d_offsets, u64 nseq);
void insert_device(cblx_ctx* c, const u8* d_bases, const u64* d_offsets, u64 nseq) {
    if (nseq == 0) return;
    check_aligned16(d_bases, "d_bases");
    const u64 cap = batch_max_bases();
    const u64 first = d2h<u64>(c, d_offsets), last = d2h<u64>(c, d_offsets + nseq);
    if (last < first) throw Error(CBLX_EINVAL, "offsets must be non-decreasing");
    if (last - first <= cap) {
        const u64 ends[2] = {first, last};
        insert_device_one(c, d_bases, d_offsets, nseq, ends);
        return;
    }
    u64 a = 0, oa = first;
    while (a < nseq) {
        u64 lo = a + 1, hi = nseq;  // largest b in [a + 1, nseq] with offsets[b] - oa <= cap (a + 1 if even one sequence is longer)
        if (d2h<u64>(c, d_offsets + hi) - oa <= cap) lo = hi;
        else {
            while (hi - lo > 1) {  // offsets[hi] - oa > cap
                const u64 mid = lo + (hi - lo) / 2;
                if (d2h<u64>(c, d_offsets + mid) - oa <= cap) lo = mid; else hi = mid;
            }
        }
        insert_device_one(c, d_bases, d_offsets + a, lo - a);
        a = lo;
        oa = d2h<u64>(c, d_offsets + a);
    }
}
void insert_device_one(cblx_ctx* c, const u8* d_bases, const u64* d_offsets, u64 nseq, const u64* ends) {
    if (c->res.count == 0 && c->P.PB > 24 && insert_device_fine(c, d_bases, d_offsets, nseq)) { collect_events(c); return; }
    dispatch(c->P, [&](auto cfg) {
        typedef decltype(cfg) C;
        ChunkPlan pl;
        plan_chunks(c, d_bases, d_offsets, nseq, pl, ends);
        if (pl.n_kmers == 0) return;
        if (pl.n_kmers >= 0xFFFFFFF0ull) throw Error(CBLX_ERANGE, "one sequence of 2^32-16 k-mers or more is not supported");
        Records rec;
        begin_records<C>(c, rec, pl.n_kmers);
        const u64 base = 0;
        Buf<u32> countsA;
        EncHist eh{};
        {   // KRN-1 also accumulates the first partition pass's tile histogram
            static_assert(ENC_HIST_WINDOW == RDX_TILE, "fused histogram windows must be the partition tiles");
            const size_t ntmax = (size_t)ceil_div(pl.n_kmers, RDX_TILE) + 256;
            countsA = Buf<u32>(c->pool, 256 * ntmax);
            CBLX_HIP(hipMemsetAsync(countsA.get(), 0, 256 * ntmax * 4, c->stream));
            const u32 nA = std::min(8u, c->P.PB);
            eh.counts = countsA.get();
            eh.shift = c->P.SB + (c->P.PB - nA);
            eh.nbits = nA;
        }
        encode<C>(c, d_bases, pl, rec.lo.get(), (typename C::HiT*)rec.hi.get(), base, eh);
        pipeline<C>(c, rec, base + pl.n_kmers, std::move(countsA));
        c->kmers_inserted += pl.n_kmers;
    });
    collect_events(c);
}

// membership flags of n device words against the resident index (WordSet::contains_batch)
template <typename C> void contains_words(cblx_ctx* c, const u64* w_lo, const typename C::HiT* w_hi, u64 n, u8* d_out) {
    typedef typename C::HiT HiT;
    const u64 step = 1ull << 28;  // QL lanes per query: 2^31 work items per launch
    for (u64 a = 0; a < n; a += step) {
        const u64 m = std::min(step, n - a);
        hipLaunchKernelGGL(k_contains<HiT>, grid1(m * QL, 256), dim3(256), 0, c->stream, w_lo + a, HiTraits<HiT>::has ? w_hi + a : w_hi, m, c->P.SB, c->P.PB, c->res.view(),
                           c->res.a_lo.get(), c->P.wide_suffix() ? c->res.a_hi.get() : (const u64*)nullptr, d_out + a);
    }
    CBLX_HIP(hipGetLastError());
}
// Tallies of a big query batch by join (k_query_join): KRN-1 + the stable partition of the query words + one workgroup per
// prefix run. CBLX_QUERY_JOIN_MIN overrides the batch size (k-mers) from which it replaces the per-query kernel.
u64 query_join_min() {  // read per call: tests switch it
    const char* e = std::getenv("CBLX_QUERY_JOIN_MIN");
    return e ? std::strtoull(e, nullptr, 10) : (u64)(4u << 20);
}
template <typename C> u64 query_join(cblx_ctx* c, const u8* d_bases /* as left by plan_chunks */, const ChunkPlan& pl, u8* d_flags /* nk bytes or null */) {
    typedef typename C::HiT HiT;
    const Consts& P = c->P;
    const u64 nk = pl.n_kmers;
    if (d_flags && nk) CBLX_HIP(hipMemsetAsync(d_flags, 0, nk, c->stream));
    if (nk == 0 || c->res.count == 0) return 0;
    Buf<u32> countsA;
    EncHist eh{};
    {   // as insert_device: KRN-1 accumulates the first partition pass's tile histogram
        const size_t ntmax = (size_t)ceil_div(nk, RDX_TILE) + 256;
        countsA = Buf<u32>(c->pool, 256 * ntmax);
        CBLX_HIP(hipMemsetAsync(countsA.get(), 0, 256 * ntmax * 4, c->stream));
        const u32 nA = std::min(8u, P.PB);
        eh.counts = countsA.get();
        eh.shift = P.SB + (P.PB - nA);
        eh.nbits = nA;
    }
    Records rec;
    Resident qd;  // directory of the query batch: prefixes and run starts
    Buf<u64> pos(c->pool, 1);
    CBLX_HIP(hipMemsetAsync(pos.get(), 0, 8, c->stream));
    const u64 step = 1ull << 23;  // workgroups per launch (x 256 threads < 2^32 work items)
    auto join = [&](auto hi_tag, u8* flags) {
        typedef decltype(hi_tag) H;
        Buf<u32> per_run(c->pool, qd.nb + 1);
        CBLX_HIP(hipMemsetAsync(per_run.get(), 0, (qd.nb + 1) * 4, c->stream));
        for (u64 b0 = 0; b0 < qd.nb; b0 += step)
            hipLaunchKernelGGL((k_query_join<C::WS, H>), dim3((unsigned)std::min(step, qd.nb - b0)), dim3(JOIN_THREADS), 0, c->stream, qd.nb, b0, qd.prefix.get(),
                               qd.start.get(), rec.lo.get(), (const H*)rec.hi.get(), P.SB, c->res.view(), c->res.a_lo.get(),
                               P.wide_suffix() ? c->res.a_hi.get() : (const u64*)nullptr, per_run.get(), flags);
        hipLaunchKernelGGL(k_sum_u32, dim3((unsigned)std::min<u64>(2048, std::max<u64>(1, ceil_div(qd.nb, 256)))), dim3(256), 0, c->stream, per_run.get(), qd.nb, pos.get());
        CBLX_HIP(hipGetLastError());
        CBLX_HIP(hipStreamSynchronize(c->stream));  // per_run dies here
    };
    if constexpr (!C::WS) {
        if (d_flags) {
            // per-query flags: 16-byte records (lo, ordinal << 32 | hi bits) through the partition of the 64-bit-hi layout
            typedef Cfg<C::WIDE, u64, false> QC;
            rec.lo = Buf<u64>(c->pool, nk + 2);
            rec.lo2 = Buf<u64>(c->pool, nk + 2);
            rec.hi = Buf<u8>(c->pool, (nk + 2) * 8);
            rec.hi2 = Buf<u8>(c->pool, (nk + 2) * 8);
            {
                Buf<u8> hi_tmp(c->pool, (nk + 2) * std::max<size_t>(1, hi_elem_size(P)));
                encode<C>(c, d_bases, pl, rec.lo.get(), (HiT*)hi_tmp.get(), 0, eh);
                hipLaunchKernelGGL(k_query_tag<HiT>, grid1(nk, 256), dim3(256), 0, c->stream, nk, (const HiT*)hi_tmp.get(), (u64*)rec.hi.get());
                CBLX_HIP(hipGetLastError());
                CBLX_HIP(hipStreamSynchronize(c->stream));  // hi_tmp dies here
            }
            partition_and_directory<QC>(c, rec, nk, std::move(countsA), qd);
            join(u64(), d_flags);
            return d2h<u64>(c, pos.get());
        }
    }
    begin_records<C>(c, rec, nk);
    encode<C>(c, d_bases, pl, rec.lo.get(), (HiT*)rec.hi.get(), 0, eh);
    partition_and_directory<C>(c, rec, nk, std::move(countsA), qd);
    join(HiT(), (u8*)nullptr);
    return d2h<u64>(c, pos.get());  // also: the temporaries may go back to the pool
}
// words whose hi part leaves 32 bits for the query's ordinal, and whose suffix fits 64 bits: flags can come from the join
inline bool query_flags_by_join(const Consts& P) { return !P.wide_suffix() && P.WB <= 96; }
// CBL::contains_seq over a batch of device-resident sequences: KRN-1, then one membership flag per k-mer (sequence after
// sequence, each in get_seq_words order) into d_out[cap] when given; *total / *positive count the flags.
void query_device(cblx_ctx* c, const u8* d_bases, const u64* d_offsets, u64 nseq, u8* d_out, u64 cap, u64* total, u64* positive) {
    if (total) *total = 0;
    if (positive) *positive = 0;
    if (nseq == 0) return;
    check_aligned16(d_bases, "d_bases");
    dispatch(c->P, [&](auto cfg) {
        typedef decltype(cfg) C;
        typedef typename C::HiT HiT;
        ChunkPlan pl;
        plan_chunks(c, d_bases, d_offsets, nseq, pl);
        const u64 nk = pl.n_kmers;
        if (total) *total = nk;
        if (nk == 0) return;
        if (nk >= 0xFFFFFFF0ull) throw Error(CBLX_ERANGE, "too many k-mers in one query batch");
        if (nk >= query_join_min() && (!d_out || query_flags_by_join(c->P))) {  // big batch: join instead of one bucket read per query
            if (d_out && nk > cap) throw Error(CBLX_ERANGE, "output capacity too small: " + std::to_string(nk) + " k-mers");
            const u64 p = query_join<C>(c, d_bases, pl, d_out);
            if (positive) *positive = p;
            return;
        }
        if (d_out && nk > cap) throw Error(CBLX_ERANGE, "output capacity too small: " + std::to_string(nk) + " k-mers");
        Buf<u64> w_lo(c->pool, nk + 2);
        Buf<u8> w_hi(c->pool, (nk + 2) * std::max<size_t>(1, hi_elem_size(c->P)));
        Buf<u8> flags;
        if (!d_out) { flags = Buf<u8>(c->pool, nk + 8); d_out = flags.get(); }
        Buf<u32> zeros(c->pool, 1);
        CBLX_HIP(hipMemsetAsync(zeros.get(), 0, 4, c->stream));
        encode<C>(c, d_bases, pl, w_lo.get(), (HiT*)w_hi.get(), 0);
        contains_words<C>(c, w_lo.get(), (const HiT*)w_hi.get(), nk, d_out);
        hipLaunchKernelGGL(k_count_zero_u8, dim3((unsigned)std::min<u64>(4096, ceil_div(nk, 256))), dim3(256), 0, c->stream, (const u8*)d_out, nk, zeros.get());
        CBLX_HIP(hipGetLastError());
        const u64 z = d2h<u32>(c, zeros.get());  // also: the temporaries may go back to the pool
        if (positive) *positive = nk - z;
    });
    collect_events(c);
}

// ---- sorted batches (multi-GPU build, include/cblx.h) -------------------------------------------------------------
// sender: KRN-1 + the full partition of the words of nseq sequences; the batch stays in c->batch
template <typename C> void sorted_batch_begin(cblx_ctx* c, const u8* d_bases, const u64* d_offsets, u64 nseq, const u32* bounds, u32 nd, u64* bucket_split,
                                              u64* word_split) {
    c->batch = SortedBatch();
    for (u32 d = 0; d <= nd; ++d) bucket_split[d] = word_split[d] = 0;
    if (nseq == 0) return;
    ChunkPlan pl;
    plan_chunks(c, d_bases, d_offsets, nseq, pl);
    if (pl.n_kmers == 0) return;
    const u64 N = pl.n_kmers;
    Records rec;
    begin_records<C>(c, rec, N);
    const size_t ntmax = (size_t)ceil_div(N, RDX_TILE) + 256;
    Buf<u32> countsA(c->pool, 256 * ntmax);
    CBLX_HIP(hipMemsetAsync(countsA.get(), 0, 256 * ntmax * 4, c->stream));
    const u32 nA = std::min(8u, c->P.PB);
    EncHist eh{};
    eh.counts = countsA.get();
    eh.shift = c->P.SB + (c->P.PB - nA);
    eh.nbits = nA;
    encode<C>(c, d_bases, pl, rec.lo.get(), (typename C::HiT*)rec.hi.get(), 0, eh);
    Resident dir;
    partition_and_directory<C>(c, rec, N, std::move(countsA), dir);
    SortedBatch& b = c->batch;
    b.n = N;
    b.nb = dir.nb;
    b.prefix = std::move(dir.prefix);
    b.start = std::move(dir.start);
    b.lo = std::move(rec.lo);
    b.hi = std::move(rec.hi);
    bucket_split[nd] = b.nb;
    word_split[nd] = N;
    if (nd > 1) {
        Buf<u32> d_bounds(c->pool, nd);
        Buf<u64> d_bs(c->pool, nd), d_ws(c->pool, nd);
        h2d(c, d_bounds.get(), bounds, nd - 1);
        hipLaunchKernelGGL(k_batch_split, grid1(nd - 1, 64), dim3(64), 0, c->stream, b.nb, b.prefix.get(), b.start.get(), nd - 1, d_bounds.get(), d_bs.get(), d_ws.get());
        CBLX_HIP(hipGetLastError());
        std::vector<u64> bs = d2h_vec<u64>(c, d_bs.get(), nd - 1), ws = d2h_vec<u64>(c, d_ws.get(), nd - 1);
        for (u32 d = 1; d < nd; ++d) { bucket_split[d] = bs[d - 1]; word_split[d] = ws[d - 1]; }
    }
    CBLX_HIP(hipStreamSynchronize(c->stream));
}
template <typename C> void sorted_batch_export(cblx_ctx* c, u32* d_prefix, u32* d_count, u8* d_suffix) {
    typedef typename C::HiT HiT;
    SortedBatch& b = c->batch;
    if (b.nb) {
        CBLX_HIP(hipMemcpyAsync(d_prefix, b.prefix.get(), b.nb * 4, hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(k_batch_counts, grid1(b.nb, 256), dim3(256), 0, c->stream, b.nb, b.start.get(), d_count);
        // the hi part matters only for suffixes wider than 64 bits (then it was carried through every pass)
        hipLaunchKernelGGL((k_batch_pack<C::WS, HiT>), dim3((unsigned)ceil_div(b.n, PACK_TILE)), dim3(PACK_THREADS), (size_t)PACK_TILE * c->P.BYTES, c->stream, b.n, b.lo.get(),
                           C::WS ? (const HiT*)b.hi.get() : (const HiT*)nullptr, c->P.SB, c->P.BYTES, d_suffix);
        CBLX_HIP(hipGetLastError());
        CBLX_HIP(hipStreamSynchronize(c->stream));
    }
    c->batch = SortedBatch();
}
// receiver: WordSet::insert_batch over sorted batches, stream order = batch order; the resident index (if any) comes first
template <typename C> void insert_sorted_batches(cblx_ctx* c, const cblx_batch_view* bt, u32 nbt) {
    typedef typename C::HiT HiT;
    constexpr bool WS = C::WS;
    const Consts& P = c->P;
    const Resident& s = c->res;
    u64 add = 0;
    for (u32 b = 0; b < nbt; ++b) {
        if (bt[b].n_buckets && (!bt[b].d_prefix || !bt[b].d_count)) throw Error(CBLX_EINVAL, "null batch arrays");
        if (bt[b].n_words && !bt[b].d_suffix) throw Error(CBLX_EINVAL, "null batch suffixes");
        add += bt[b].n_words;
    }
    if (add == 0) return;
    const u64 nprefix = 1ull << P.PB, nwords = std::max<u64>(1, nprefix / 64);
    Resident nr;
    Buf<u32> raw, m_cs, bad(c->pool, 1);
    Buf<u64> m_sstart, m_ostart;
    Buf<u8> m_skind, m_okind;
    std::vector<Buf<u32>> off_in_run(nbt);
    std::vector<Buf<u64>> src_off(nbt);
    u64 T = 0;
    {
        StageTimer t(c, ST_DIR);
        Buf<u32> popc(c->pool, nwords);
        nr.bv = Buf<u64>(c->pool, nwords);
        nr.rank_dir = Buf<u64>(c->pool, nwords + 1);
        if (s.count) CBLX_HIP(hipMemcpyAsync(nr.bv.get(), s.bv.get(), nwords * 8, hipMemcpyDeviceToDevice, c->stream));
        else CBLX_HIP(hipMemsetAsync(nr.bv.get(), 0, nwords * 8, c->stream));
        CBLX_HIP(hipMemsetAsync(bad.get(), 0, 4, c->stream));
        for (u32 b = 0; b < nbt; ++b)
            if (bt[b].n_buckets)
                hipLaunchKernelGGL(k_batch_bits, grid1(bt[b].n_buckets, 256), dim3(256), 0, c->stream, bt[b].n_buckets, bt[b].d_prefix, bt[b].d_count, nprefix, nr.bv.get(), bad.get());
        hipLaunchKernelGGL(k_popc_words, grid1(nwords, 256), dim3(256), 0, c->stream, nwords, nr.bv.get(), popc.get());
        nr.nb = exclusive_scan<u64>(c, popc.get(), nwords, nr.rank_dir.get());
        if (d2h<u32>(c, bad.get())) throw Error(CBLX_EINVAL, "sorted batch: prefixes must be strictly ascending and below 2^PREFIX_BITS, counts non-zero");
        const u64 nb = nr.nb;
        nr.prefix = Buf<u32>(c->pool, nb + 1);
        nr.start = Buf<u64>(c->pool, nb + 1);
        nr.cnt = Buf<u32>(c->pool, nb + 1);
        nr.kind = Buf<u8>(c->pool, nb + 1);
        raw = Buf<u32>(c->pool, nb + 1);
        m_cs = Buf<u32>(c->pool, nb + 1);
        m_sstart = Buf<u64>(c->pool, nb + 1);
        m_ostart = Buf<u64>(c->pool, nb + 1);
        m_skind = Buf<u8>(c->pool, nb + 1);
        m_okind = Buf<u8>(c->pool, nb + 1);
        // the resident part of every merged run (raw = its length), then batch after batch on top of it
        hipLaunchKernelGGL(k_merge_table, grid1(nprefix, 256), dim3(256), 0, c->stream, nprefix, nr.bv.get(), nr.rank_dir.get(), s.view(), DirView{}, nr.prefix.get(),
                           raw.get(), m_cs.get(), m_sstart.get(), m_ostart.get(), m_skind.get(), m_okind.get());
        for (u32 b = 0; b < nbt; ++b) {
            if (!bt[b].n_buckets) continue;
            off_in_run[b] = Buf<u32>(c->pool, bt[b].n_buckets);
            src_off[b] = Buf<u64>(c->pool, bt[b].n_buckets + 1);
            hipLaunchKernelGGL(k_batch_offsets, grid1(bt[b].n_buckets, 256), dim3(256), 0, c->stream, bt[b].n_buckets, bt[b].d_prefix, bt[b].d_count, nr.bv.get(),
                               nr.rank_dir.get(), raw.get(), off_in_run[b].get());
            const u64 nw = exclusive_scan<u64>(c, bt[b].d_count, bt[b].n_buckets, src_off[b].get());
            if (nw != bt[b].n_words) throw Error(CBLX_EINVAL, "sorted batch: n_words does not match the counts");
        }
        T = exclusive_scan<u64>(c, raw.get(), nb, nr.start.get());
        hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, nr.start.get() + nb, T);
        CBLX_HIP(hipGetLastError());
    }
    if (T != s.count + add) throw Error(CBLX_EDEVICE, "sorted batches: run lengths do not add up (internal error)");
    nr.a_lo = Buf<u64>(c->pool, T + 2);
    if (WS) nr.a_hi = Buf<u64>(c->pool, T + 2);
    {
        StageTimer t(c, ST_EXPAND);
        if (s.count)
            with_lpb(s.count, s.nb, [&](auto lpb) {
                constexpr int LPB = decltype(lpb)::value;
                hipLaunchKernelGGL((k_gather_resident<WS, LPB>), lpb_grid(nr.nb, LPB), dim3(256), 0, c->stream, nr.nb, nr.start.get(), m_cs.get(), m_sstart.get(),
                                   s.a_lo.get(), s.a_hi.get(), nr.a_lo.get(), nr.a_hi.get());
            });
        std::vector<Buf<u64>> dst(nbt);  // released after the bucket stage below has synchronised
        for (u32 b = 0; b < nbt; ++b) {
            if (!bt[b].n_buckets) continue;
            dst[b] = Buf<u64>(c->pool, bt[b].n_buckets);
            hipLaunchKernelGGL(k_batch_dst, grid1(bt[b].n_buckets, 256), dim3(256), 0, c->stream, bt[b].n_buckets, bt[b].d_prefix, off_in_run[b].get(), nr.bv.get(),
                               nr.rank_dir.get(), nr.start.get(), dst[b].get());
            with_lpb(bt[b].n_words, bt[b].n_buckets, [&](auto lpb) {
                constexpr int LPB = decltype(lpb)::value;
                hipLaunchKernelGGL((k_gather_packed<WS, LPB>), lpb_grid(bt[b].n_buckets, LPB), dim3(256), 0, c->stream, bt[b].n_buckets, bt[b].d_count, src_off[b].get(),
                                   dst[b].get(), bt[b].d_suffix, bt[b].n_words * P.BYTES, P.BYTES, nr.a_lo.get(), nr.a_hi.get());
            });
        }
        CBLX_HIP(hipGetLastError());
        CBLX_HIP(hipStreamSynchronize(c->stream));
        CBLX_HIP(hipGetLastError());
    }
    bucket_stage<C>(c, nr, s.view());
    CBLX_HIP(hipStreamSynchronize(c->stream));
    c->res = std::move(nr);
    c->kmers_inserted += add;
}

// ---- `self |= other`, both resident on this device (src/cbl.rs:433-449 -> src/wordset/set_ops.rs:123-157) ---------
// `s`: self's index (c->res for `c |= o`; another context's for cblx_merge_from, which leaves it untouched); the result becomes c->res
template <typename C> void merge_direct(cblx_ctx* c, const Resident& s, const Resident& o) {
    typedef typename C::HiT HiT;
    constexpr bool WS = C::WS;
    const Consts& P = c->P;
    const u64 nprefix = 1ull << P.PB, nwords = std::max<u64>(1, nprefix / 64);
    Resident nr;
    Buf<u32> raw, m_cs;
    Buf<u64> m_sstart, m_ostart;
    Buf<u8> m_skind, m_okind;
    u64 N = 0;
    {
        StageTimer t(c, ST_DIR);
        Buf<u32> popc(c->pool, nwords);
        nr.bv = Buf<u64>(c->pool, nwords);
        nr.rank_dir = Buf<u64>(c->pool, nwords + 1);
        hipLaunchKernelGGL(k_bv_or, grid1(nwords, 256), dim3(256), 0, c->stream, nwords, s.bv.get(), o.bv.get(), nr.bv.get(), popc.get());
        nr.nb = exclusive_scan<u64>(c, popc.get(), nwords, nr.rank_dir.get());
        const u64 nb = nr.nb;
        nr.prefix = Buf<u32>(c->pool, nb + 1);
        nr.start = Buf<u64>(c->pool, nb + 1);
        nr.cnt = Buf<u32>(c->pool, nb + 1);
        nr.kind = Buf<u8>(c->pool, nb + 1);
        raw = Buf<u32>(c->pool, nb + 1);
        m_cs = Buf<u32>(c->pool, nb + 1);
        m_sstart = Buf<u64>(c->pool, nb + 1);
        m_ostart = Buf<u64>(c->pool, nb + 1);
        m_skind = Buf<u8>(c->pool, nb + 1);
        m_okind = Buf<u8>(c->pool, nb + 1);
        hipLaunchKernelGGL(k_merge_table, grid1(nprefix, 256), dim3(256), 0, c->stream, nprefix, nr.bv.get(), nr.rank_dir.get(), s.view(), o.view(), nr.prefix.get(),
                           raw.get(), m_cs.get(), m_sstart.get(), m_ostart.get(), m_skind.get(), m_okind.get());
        N = exclusive_scan<u64>(c, raw.get(), nb, nr.start.get());
        hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, nr.start.get() + nb, N);
        CBLX_HIP(hipGetLastError());
    }
    if (N != s.count + o.count) throw Error(CBLX_EDEVICE, "merge: run lengths do not match the two indexes (internal error)");
    const u64 nb = nr.nb;
    nr.a_lo = Buf<u64>(c->pool, N + 2);
    if (WS) nr.a_hi = Buf<u64>(c->pool, N + 2);
    // Trie |= Trie (both lists ascending): merged by k_bucket_union straight from the two arenas — not gathered, not sorted again
    // (wide suffixes too since round 5: two-word elements). CBLX_MERGE_UNION=0 keeps the counting-sort route (tests compare the two)
    const char* union_env = std::getenv("CBLX_MERGE_UNION");  // (read per call: tests switch it)
    const bool union_path = !(union_env && union_env[0] == '0');
    // both-sided buckets of up to 4096 words (the counting-sort classes) are read where they are stored: their kernel loads self's part from
    // self's arena and other's from other's, and only the result is written — the gather moved 16 bytes per word for nothing.
    // CBLX_MERGE_DIRECT=0 gathers them as before
    const char* direct_env = std::getenv("CBLX_MERGE_DIRECT");
    const bool direct = msd_takes<WS>(P.SB) && !(direct_env && direct_env[0] == '0');
    const u32 direct_upto = direct ? 512u * MED_ITEMS : 0u;
    {
        StageTimer t(c, ST_EXPAND);
        with_lpb(N, nb, [&](auto lpb) {
            constexpr int LPB = decltype(lpb)::value;
            hipLaunchKernelGGL((k_merge_gather<WS, LPB>), lpb_grid(nb, LPB), dim3(256), 0, c->stream, nb, nr.start.get(), m_cs.get(), m_sstart.get(), m_ostart.get(),
                               s.a_lo.get(), s.a_hi.get(), o.a_lo.get(), o.a_hi.get(), nr.a_lo.get(), nr.a_hi.get(), union_path ? m_skind.get() : (const u8*)nullptr,
                               union_path ? m_okind.get() : (const u8*)nullptr, direct_upto);
        });
    }
    Buf<BDesc> lists(c->pool, (size_t)CLS_N * std::max<u64>(nb, 1));
    Buf<u32> list_n(c->pool, CLS_N);
    CBLX_HIP(hipMemsetAsync(list_n.get(), 0, CLS_N * 4, c->stream));
    const bool prof = (c->flags & CBLX_FLAG_PROFILE) != 0;
    Buf<unsigned long long> cls_words;
    if (prof) { cls_words = Buf<unsigned long long>(c->pool, CLS_N + 2); CBLX_HIP(hipMemsetAsync(cls_words.get(), 0, (CLS_N + 2) * 8, c->stream)); }
    hipLaunchKernelGGL(k_classify_merge, grid1(nb, CLASSIFY_THREADS), dim3(CLASSIFY_THREADS), 0, c->stream, nb, WS ? 512u : 1024u, nr.start.get(), m_cs.get(), m_skind.get(), m_okind.get(),
                       nr.cnt.get(), nr.kind.get(), lists.get(), list_n.get(), union_path, cls_words.get());
    CBLX_HIP(hipGetLastError());
    std::vector<u32> ln = d2h_vec<u32>(c, list_n.get(), CLS_N);
    std::vector<unsigned long long> cw;
    if (prof) cw = d2h_vec<unsigned long long>(c, cls_words.get(), CLS_N + 1);
    const MergeArgs ma{m_cs.get(), m_ostart.get(), m_okind.get(), o.a_lo.get(), o.a_hi.get()};  // (kernels that work on the gathered run)
    MergeArgs ma_msd = ma;                                                                       // (the counting-sort classes: in place)
    if (direct) { ma_msd.s_lo = s.a_lo.get(); ma_msd.s_hi = s.a_hi.get(); ma_msd.sstart = m_sstart.get(); }
    u64* a_lo = nr.a_lo.get();
    HiT* a_hi = WS ? (HiT*)nr.a_hi.get() : (HiT*)nullptr;
    // (Round 5 measured the unions on a second stream beside the counting-sort classes — launched first they take every wave slot and the two run one
    //  after the other, launched second they share the chip and the pair takes exactly the sum of the two: 10.10 against 10.06 ms. One stream it stays.)
    if (ln[CLS_UNION]) {
        StageTimer t(c, ST_BBIG);
        hipLaunchKernelGGL((k_bucket_union<WS>), dim3(ln[CLS_UNION]), dim3(UNI_THREADS), 0, c->stream, lists.get() + (size_t)CLS_UNION * nb, list_n.get() + CLS_UNION, m_cs.get(), m_sstart.get(),
                           m_ostart.get(), (const u64*)s.a_lo.get(), (const u64*)s.a_hi.get(), (const u64*)o.a_lo.get(), (const u64*)o.a_hi.get(), a_lo, (u64*)nr.a_hi.get(), P.SB,
                           nr.cnt.get(), nr.kind.get());
        CBLX_HIP(hipGetLastError());
    }
    {
        StageTimer t(c, ST_BMED);
        // both-sided buckets: counting sort on the top suffix bits + ranking inside the sub-buckets (k_bucket_msd in its
        // merge mode); a bucket with a crowded sub-bucket comes back through `retry` and takes the LDS radix sort
        Buf<BDesc> retry(c->pool, std::max<u64>(nb, 1));
        Buf<u32> retry_n(c->pool, 1);
        CBLX_HIP(hipMemsetAsync(retry_n.get(), 0, 4, c->stream));
        auto msd = [&](auto packed_tag) {
            constexpr bool PK = decltype(packed_tag)::value;
            if constexpr (PK || WS) {  // packed elements: the walk kernel in its merge mode (round 6), class by class as below
                if (sorted_kernel()) {
                    auto go = [&](auto thr, auto cap, int cls) {
                        constexpr int T = decltype(thr)::value, CAPV = decltype(cap)::value;
                        if (ln[cls])
                            hipLaunchKernelGGL((k_bucket_sorted<T, CAPV, WS, HiT, true>), dim3(ln[cls]), dim3(T), 0, c->stream, lists.get() + (size_t)cls * nb, list_n.get() + cls, a_lo, a_hi, P.SB,
                                               nr.cnt.get(), nr.kind.get(), retry.get(), retry_n.get(), (u8*)nullptr, (u32*)nullptr, ma_msd);
                    };
                    go(std::integral_constant<int, 64>(), std::integral_constant<int, 128>(), CLS_M16);
                    go(std::integral_constant<int, 64>(), std::integral_constant<int, 512>(), CLS_M64);
                    go(std::integral_constant<int, 128>(), std::integral_constant<int, 1024>(), CLS_M128);
                    go(std::integral_constant<int, 256>(), std::integral_constant<int, 2048>(), CLS_M256);
                    go(std::integral_constant<int, 512>(), std::integral_constant<int, 4096>(), CLS_M512);
                    return;
                }
            }
            if (ln[CLS_M16])
                hipLaunchKernelGGL((k_bucket_msd<64, 128, PK, WS, HiT, true>), dim3(ln[CLS_M16]), dim3(64), 0, c->stream, lists.get() + (size_t)CLS_M16 * nb, list_n.get() + CLS_M16,
                                   a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), retry.get(), retry_n.get(), ma_msd);
            if (ln[CLS_M64])
                hipLaunchKernelGGL((k_bucket_msd<64, 512, PK, WS, HiT, true>), dim3(ln[CLS_M64]), dim3(64), 0, c->stream, lists.get() + (size_t)CLS_M64 * nb, list_n.get() + CLS_M64,
                                   a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), retry.get(), retry_n.get(), ma_msd);
            if (ln[CLS_M128])
                hipLaunchKernelGGL((k_bucket_msd<128, 1024, PK, WS, HiT, true>), dim3(ln[CLS_M128]), dim3(128), 0, c->stream, lists.get() + (size_t)CLS_M128 * nb, list_n.get() + CLS_M128,
                                   a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), retry.get(), retry_n.get(), ma_msd);
            if (ln[CLS_M256])
                hipLaunchKernelGGL((k_bucket_msd<256, 2048, PK, WS, HiT, true>), dim3(ln[CLS_M256]), dim3(256), 0, c->stream, lists.get() + (size_t)CLS_M256 * nb, list_n.get() + CLS_M256,
                                   a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), retry.get(), retry_n.get(), ma_msd);
            if (ln[CLS_M512])
                hipLaunchKernelGGL((k_bucket_msd<512, 4096, PK, WS, HiT, true>), dim3(ln[CLS_M512]), dim3(512), 0, c->stream, lists.get() + (size_t)CLS_M512 * nb, list_n.get() + CLS_M512,
                                   a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), retry.get(), retry_n.get(), ma_msd);
        };
        if constexpr (!WS) {
            if (P.SB + PK_BITS <= 64) msd(std::true_type()); else msd(std::false_type());
        } else {
            if (msd_takes<WS>(P.SB)) msd(std::false_type());
            else  // (see bucket_stage) every class to the LDS radix kernel
                for (int cls : {CLS_M16, CLS_M64, CLS_M128, CLS_M256, CLS_M512})
                    if (ln[cls])
                        hipLaunchKernelGGL((k_bucket_medium<512, WS, HiT>), dim3(ln[cls]), dim3(512), 0, c->stream, lists.get() + (size_t)cls * nb, list_n.get() + cls, a_lo, a_hi, P.SB,
                                           nr.cnt.get(), nr.kind.get(), ma);
        }
        const u32 nretry = (ln[CLS_M16] || ln[CLS_M64] || ln[CLS_M128] || ln[CLS_M256] || ln[CLS_M512]) ? d2h<u32>(c, retry_n.get()) : 0u;
        if (nretry) {
            if (direct)  // (these buckets were not gathered: the radix kernel works on the run)
                hipLaunchKernelGGL((k_merge_gather_list<WS>), dim3(nretry), dim3(256), 0, c->stream, retry.get(), retry_n.get(), m_cs.get(), m_sstart.get(), m_ostart.get(), s.a_lo.get(),
                                   s.a_hi.get(), (const u64*)o.a_lo.get(), (const u64*)o.a_hi.get(), nr.a_lo.get(), nr.a_hi.get());
            hipLaunchKernelGGL((k_bucket_medium<512, WS, HiT>), dim3(nretry), dim3(512), 0, c->stream, retry.get(), retry_n.get(), a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), ma);
        }
        if constexpr (!WS) if (ln[CLS_M1024])
            hipLaunchKernelGGL((k_bucket_medium<1024, WS, HiT>), dim3(ln[CLS_M1024]), dim3(1024), 0, c->stream, lists.get() + (size_t)CLS_M1024 * nb,
                               list_n.get() + CLS_M1024, a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), ma);
        CBLX_HIP(hipGetLastError());
        CBLX_HIP(hipStreamSynchronize(c->stream));  // retry buffers die here
    }
    {
        Twin tw;
        if (msd_takes<WS>(P.SB)) big_stage<C>(c, lists.get() + (size_t)CLS_BIG * nb, list_n.get() + CLS_BIG, ln[CLS_BIG], nr, ma, tw);
        else huge_stage<C>(c, lists.get() + (size_t)CLS_BIG * nb, list_n.get() + CLS_BIG, ln[CLS_BIG], a_lo, a_hi, nr, ma);
        huge_stage<C>(c, lists.get() + (size_t)CLS_HUGE * nb, list_n.get() + CLS_HUGE, ln[CLS_HUGE], a_lo, a_hi, nr, ma);
        finish_twin<C>(c, nr, tw);
    }
    {
        Buf<u64> total(c->pool, 1);
        CBLX_HIP(hipMemsetAsync(total.get(), 0, 8, c->stream));
        hipLaunchKernelGGL(k_sum_u32, dim3((unsigned)std::min<u64>(2048, std::max<u64>(1, ceil_div(nb, 256)))), dim3(256), 0, c->stream, nr.cnt.get(), nb, total.get());
        nr.count = d2h<u64>(c, total.get());
    }
    if (prof) {
        // words every stage's kernels were given (cblx_stage_units): the gather copies the one-sided buckets and the both-sided ones their
        // kernel does not read in place; the unions are priced on what they WRITE (2 BYTES read + BYTES written per output)
        u64 msd = 0, gathered = cw[CLS_N];
        for (int cls : {CLS_M16, CLS_M64, CLS_M128, CLS_M256, CLS_M512}) msd += cw[cls];
        if (!direct) gathered += msd;
        gathered += cw[CLS_M1024] + cw[CLS_HUGE] + cw[CLS_BIG];
        c->stages[ST_EXPAND].units += gathered;
        c->stages[ST_BMED].units += msd + cw[CLS_M1024];
        c->stages[ST_BHUGE].units += cw[CLS_HUGE];
        u64 uni_out = 0;
        if (ln[CLS_UNION]) {
            Buf<u64> tot(c->pool, 1);
            CBLX_HIP(hipMemsetAsync(tot.get(), 0, 8, c->stream));
            hipLaunchKernelGGL(k_sum_list_counts, dim3((unsigned)std::min<u64>(1024, ceil_div(ln[CLS_UNION], 256))), dim3(256), 0, c->stream, lists.get() + (size_t)CLS_UNION * nb, ln[CLS_UNION],
                               (const u32*)nr.cnt.get(), tot.get());
            uni_out = d2h<u64>(c, tot.get());
        }
        c->stages[ST_BBIG].units += uni_out + cw[CLS_BIG];
    }
    c->res = std::move(nr);
}


}  // namespace
