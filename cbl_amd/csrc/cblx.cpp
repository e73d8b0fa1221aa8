// cblx.cpp — host side of libcblx: the C ABI of include/cblx.h over the HIP kernels (gfx950).
//
// Mirrors the reference's `CBL<K, T, PREFIX_BITS>` surface for the bulk-insert path
// (/root/reference/src/cbl.rs:71-79,127-177,328-339,433-449) with the WordSet state
// (/root/reference/src/wordset/mod.rs:18-26: prefix bitvector + rank->bucket directory + suffix containers) held in
// HBM: bitvector words, popcount-scan rank directory, bucket table indexed by rank, one suffix arena.
// There is no CPU fallback: every data-path step below is a kernel launch.
#include "../../include/cblx.h"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <thread>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "kernels_bucket.hpp"
#include "kernels_serde.hpp"
#include "xfer.hpp"

using namespace cblx;

namespace {

thread_local std::string g_global_err;

inline u32 ilog2_npo2(u32 v) { u32 l = 0; while ((1u << l) < v) ++l; return l; }
inline u64 ceil_div(u64 a, u64 b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------------
// cached device allocations (hipMalloc is kept out of the hot path between flushes)
struct Pool {
    struct Blk { void* p; size_t sz; bool used; };
    std::vector<Blk> blks;
    void* alloc(size_t sz) {
        if (sz == 0) sz = 256;
        sz = (sz + 255) & ~(size_t)255;
        int best = -1;
        for (size_t i = 0; i < blks.size(); ++i)
            if (!blks[i].used && blks[i].sz >= sz && blks[i].sz <= sz + sz / 2 + 4096 && (best < 0 || blks[i].sz < blks[best].sz)) best = (int)i;
        if (best >= 0) { blks[best].used = true; return blks[best].p; }
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, sz);
        if (e != hipSuccess) {
            trim();
            e = hipMalloc(&p, sz);
            if (e != hipSuccess) throw Error(CBLX_ENOMEM, "hipMalloc(" + std::to_string(sz) + ") failed: " + hipGetErrorString(e));
        }
        blks.push_back({p, sz, true});
        return p;
    }
    void release(void* p) {
        if (!p) return;
        for (auto& b : blks) if (b.p == p) { b.used = false; return; }
    }
    void trim() {
        std::vector<Blk> keep;
        for (auto& b : blks) { if (b.used) keep.push_back(b); else (void)hipFree(b.p); }
        blks.swap(keep);
    }
    ~Pool() { for (auto& b : blks) (void)hipFree(b.p); }
};

template <typename T> struct Buf {  // RAII view on a pool allocation
    Pool* pool = nullptr;
    T* p = nullptr;
    size_t n = 0;
    Buf() {}
    Buf(Pool& pl, size_t count) : pool(&pl), p((T*)pl.alloc(count * sizeof(T))), n(count) {}
    Buf(const Buf&) = delete;
    Buf& operator=(const Buf&) = delete;
    Buf(Buf&& o) noexcept : pool(o.pool), p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    Buf& operator=(Buf&& o) noexcept { if (this != &o) { reset(); pool = o.pool; p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    void reset() { if (p && pool) pool->release(p); p = nullptr; n = 0; }
    ~Buf() { reset(); }
    T* get() const { return p; }
};

// resident index (WordSet state) in HBM
struct Resident {
    u64 nb = 0;          // non-empty prefixes (tiered.len())
    u64 count = 0;       // k-mers
    Buf<u64> bv;         // 2^PB bits
    Buf<u64> rank_dir;   // per bv word, exclusive
    Buf<u32> prefix;     // per rank
    Buf<u64> start;      // per rank (+1): first arena slot
    Buf<u32> cnt;        // per rank
    Buf<u8> kind;        // per rank
    Buf<u64> a_lo, a_hi; // suffix arena (slack layout); a_hi only when SUFFIX_BITS > 64
    bool empty() const { return nb == 0; }
    DirView view() const { return DirView{bv.get(), rank_dir.get(), cnt.get(), kind.get(), start.get(), nb}; }
};

struct Stage { const char* name; double ms = 0; u64 launches = 0; };
enum { ST_CHUNKS, ST_ENCODE, ST_HIST, ST_SCAN, ST_SCATTER, ST_DIR, ST_BSMALL, ST_BMED, ST_BHUGE, ST_EXPAND, ST_N };
const char* kStageNames[ST_N] = {"chunks", "encode", "radix_hist", "radix_scan", "radix_scatter", "directory",
                                 "bucket_small", "bucket_medium", "bucket_huge", "merge_gather"};

// Sequences enqueued by cblx_insert_seq / cblx_insert_seqs / the FASTA reader. They are staged straight into HBM
// while the caller keeps enqueueing: small appends fill pinned write blocks that are DMA'd as they fill up, bulk
// appends go through the Xfer lanes. flush() only has to wait for the last DMA.
struct Ingest {
    static constexpr size_t BASES_BLK = 4u << 20, OFF_BLK = 512u << 10;
    struct Writer {
        u8* blk[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        bool busy[2] = {false, false};
        int cur = 0;
        size_t cap = 0, fill = 0;
        u64 issued = 0;  // bytes of the logical stream already handed to the DMA engine
    };
    Buf<u8> d_bases;   // capacity >= nbytes + 64
    Buf<u64> d_off;    // capacity >= nseq + 1; d_off[0] = 0
    u64 nbytes = 0, nseq = 0;
    u64 last_end = 0;  // nbytes at the end of the last complete sequence
    Writer wb, wo;
    hipStream_t s = nullptr;
    std::unique_ptr<Xfer> xfer;
};

}  // namespace

struct cblx_ctx {
    Consts P;
    int device = 0;
    u32 flags = 0;
    hipStream_t stream = nullptr;
    Pool pool;
    Resident res;
    Ingest ing;
    std::string err;
    u64 kmers_inserted = 0;
    Stage stages[ST_N];
    struct Ev { int st; hipEvent_t a, b; };
    std::vector<Ev> evs;
    std::vector<hipEvent_t> ev_free;

    cblx_ctx() { for (int i = 0; i < ST_N; ++i) stages[i].name = kStageNames[i]; }
};

namespace {

struct StageTimer {  // brackets a group of launches with HIP events when profiling is on
    cblx_ctx* c;
    int idx = -1;
    StageTimer(cblx_ctx* ctx, int st) : c(ctx) {
        if (!(c->flags & CBLX_FLAG_PROFILE)) return;
        auto get = [&]() { hipEvent_t e; if (!c->ev_free.empty()) { e = c->ev_free.back(); c->ev_free.pop_back(); } else CBLX_HIP(hipEventCreate(&e)); return e; };
        cblx_ctx::Ev ev{st, get(), get()};
        CBLX_HIP(hipEventRecord(ev.a, c->stream));
        c->evs.push_back(ev);
        idx = (int)c->evs.size() - 1;
    }
    ~StageTimer() { if (idx >= 0) (void)hipEventRecord(c->evs[idx].b, c->stream); }
};
void collect_events(cblx_ctx* c) {
    if (c->evs.empty()) return;
    CBLX_HIP(hipStreamSynchronize(c->stream));
    for (auto& e : c->evs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { c->stages[e.st].ms += ms; c->stages[e.st].launches++; }
        c->ev_free.push_back(e.a);
        c->ev_free.push_back(e.b);
    }
    c->evs.clear();
}

inline dim3 grid1(u64 n, u32 threads) { return dim3((unsigned)std::max<u64>(1, ceil_div(n, threads))); }

template <typename T> T d2h(cblx_ctx* c, const T* dptr) {
    T v;
    CBLX_HIP(hipMemcpyAsync(&v, dptr, sizeof(T), hipMemcpyDeviceToHost, c->stream));
    CBLX_HIP(hipStreamSynchronize(c->stream));
    return v;
}
template <typename T> std::vector<T> d2h_vec(cblx_ctx* c, const T* dptr, size_t n) {
    std::vector<T> v(n);
    if (n) {
        CBLX_HIP(hipMemcpyAsync(v.data(), dptr, n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
        CBLX_HIP(hipStreamSynchronize(c->stream));
    }
    return v;
}
template <typename T> void h2d(cblx_ctx* c, T* dptr, const T* h, size_t n) {
    if (n) CBLX_HIP(hipMemcpyAsync(dptr, h, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
}

// device-wide exclusive scan of u32 -> OutT; returns the total
template <typename OutT> u64 exclusive_scan(cblx_ctx* c, const u32* in, u64 n, OutT* out) {
    if (n == 0) return 0;
    const u64 nb = ceil_div(n, SCAN_TILE);
    Buf<u64> sums(c->pool, nb + 1);
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, c->stream, in, n, sums.get());
    hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, c->stream, sums.get(), nb);
    hipLaunchKernelGGL(k_scan_apply<OutT>, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, c->stream, in, n, sums.get(), out);
    CBLX_HIP(hipGetLastError());
    return d2h<u64>(c, sums.get() + nb);
}

// column prefixes + column totals of counts[tile][256] (see k_colscan_*); nt_dev (device) overrides nt_upper when set
void colscan(cblx_ctx* c, const u32* counts, const u32* nt_dev, u32 nt_upper, u32* colpre, u32* coltot, Buf<u32>& scratch) {
    const u32 nchunks = (u32)std::max<u64>(1, ceil_div(nt_upper, COLSCAN_ROWS));
    if (scratch.n < (size_t)nchunks * 256) scratch = Buf<u32>(c->pool, (size_t)nchunks * 256);
    hipLaunchKernelGGL(k_colscan_reduce, dim3(nchunks), dim3(256), 0, c->stream, counts, nt_dev, nt_upper, scratch.get());
    hipLaunchKernelGGL(k_colscan_spine, dim3(1), dim3(256), 0, c->stream, scratch.get(), nchunks, coltot);
    hipLaunchKernelGGL(k_colscan_apply, dim3(nchunks), dim3(256), 0, c->stream, counts, nt_dev, nt_upper, scratch.get(), colpre);
}

// ---- template configuration ------------------------------------------------------------------------------
template <bool WIDE_, typename HiT_, bool WS_> struct Cfg {
    static constexpr bool WIDE = WIDE_;
    typedef HiT_ HiT;
    static constexpr bool WS = WS_;
};
template <typename F> void dispatch(const Consts& P, F&& f) {
    if (!P.has_hi()) f(Cfg<false, NoHi, false>());
    else if (!P.wide_kmer()) f(Cfg<false, u8, false>());
    else if (!P.wide_suffix()) f(Cfg<true, u64, false>());
    else f(Cfg<true, u64, true>());
}
inline size_t hi_elem_size(const Consts& P) { return !P.has_hi() ? 0 : (!P.wide_kmer() ? 1 : 8); }

// ---- the sort + directory + per-bucket pipeline over N records (lo/hi), resident records first -------------
struct Records {
    Buf<u64> lo, lo2;
    Buf<u8> hi, hi2;  // raw bytes; element size = hi_elem_size
    const u64* ext_lo = nullptr;  // optional caller-owned source of the FIRST pass (no resident words in front)
    const void* ext_hi = nullptr;
};

// KRN-2 + KRN-4 over N records: stable partition by prefix, then the directory (bitvector, rank directory, bucket table
// with the RAW run of every prefix; nr.cnt / nr.kind are allocated, not filled). The sorted records end up in rec.lo/hi.
// `countsA`: histogram of the first pass already accumulated by KRN-1 (empty Buf = compute it here)
template <typename C> void partition_and_directory(cblx_ctx* c, Records& rec, u64 N, Buf<u32> countsA, Resident& nr) {
    typedef typename C::HiT HiT;
    const Consts& P = c->P;
    if (N >= 0xFFFFFFF0ull) throw Error(CBLX_ERANGE, "more than 2^32-16 words in one index are not supported yet");
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
    const u32 nA = std::min(8u, P.PB), RB = P.PB - nA;  // bits of pass A, bits left for the LSD passes
    const u64 nprefix = 1ull << P.PB, nwords = std::max<u64>(1, nprefix / 64);
    Buf<u32> start_dense(c->pool, nprefix);
    bool have_dense = false;
    {
        const u32 ntiles = (u32)ceil_div(N, RDX_TILE), nt_max = ntiles + 256;
        const u32 npassL = (RB + 7) / 8, nseg = 1u << nA;
        // The last LSD pass cuts its tiles at (segment x lower digits) groups when there are few enough of them; the
        // bucket directory then comes from that pass's tables (k_dir_gather) instead of a scan of the sorted records.
        const u32 low_bits = npassL ? 8 * (npassL - 1) : 0, last_bits = RB - low_bits;
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
        bool have_dig = false;
        auto next_digit = [&](u32 next_pass) -> DigitBits {
            if (next_pass >= npassL) return DigitBits{0, 0};
            return DigitBits{P.SB + 8 * next_pass, std::min(8u, RB - 8 * next_pass)};
        };
        if (next_digit(0).nbits) dig = Buf<u8>(c->pool, N + 64);
        {   // pass A
            const TileView tv{nullptr, nullptr, nullptr, nullptr, ntiles, N};
            const DigitBits dfn{P.SB + RB, nA};
            const DigitBits nd = next_digit(0);
            u8* ndp = nd.nbits ? dig.get() : nullptr;
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
            const DigitBits dfn{P.SB + 8 * pass, std::min(8u, RB - 8 * pass)};
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
                u8* ndp = nd.nbits && dig.get() ? dig.get() : nullptr;
                { StageTimer t(c, ST_HIST);
                  if (have_dig)
                      hipLaunchKernelGGL(k_radix_hist_bytes, dim3((xcd_grid(ntm) + HISTB_WAVES - 1) / HISTB_WAVES + 8), dim3(64 * HISTB_WAVES), 0, c->stream, (const u8*)dig.get(), tv, counts.get());
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
                    CBLX_HIP(hipMemsetAsync(start_dense.get(), 0xFF, nprefix * 4, c->stream));
                }
                { StageTimer t(c, ST_SCATTER);
                  hipLaunchKernelGGL((k_radix_scatter<H, H, DigitBits>), dim3(xcd_grid(ntm)), dim3(RDX_THREADS), 0, c->stream, lo, hin, tv, dfn, colpre.get(),
                                     adj.get(), lo2, hout, nd, ndp, fused_dir ? start_dense.get() : (u32*)nullptr, P.SB, RB, low_bits, amb.get(), amb_stride); }
                if (fused_dir) {
                    StageTimer t(c, ST_DIR);
                    hipLaunchKernelGGL(k_dir_resolve<H>, grid1((u64)ntm * amb_stride, 256), dim3(256), 0, c->stream, ntd, amb_stride, (const u32*)amb.get(), tv.seg,
                                       (const u32*)seg_start.get(), (const u64*)lo2, (const H*)hout, P.SB, RB, start_dense.get());
                    CBLX_HIP(hipStreamSynchronize(c->stream));  // amb is released at the end of this scope
                    have_dense = true;
                }
                have_dig = ndp != nullptr;
                if (last && tbl_dir) {
                    StageTimer t(c, ST_DIR);
                    hipLaunchKernelGGL(k_dir_gather, dim3(G), dim3(256), 0, c->stream, low_bits, last_bits, grp_tiles ? grp_first.get() : seg_first.get(), seg_start.get(), ntd,
                                       colpre.get(), coltot.get(), adj.get(), start_dense.get());
                    if (grp_tiles)  // cold segments kept plain tiles: their boundaries come from their (few) records, now in lo2
                        hipLaunchKernelGGL(k_boundaries_cold<H>, dim3(nseg, 32), dim3(256), 0, c->stream, (const u64*)lo2, (const H*)hout, P.SB, RB, seg_start.get(), start_dense.get());
                    have_dense = true;
                }
            };
            if constexpr (DROP_HI) run(NoHi()); else run(HiT());
            advance();
        }
        CBLX_HIP(hipGetLastError());
        if (lo == rec.lo2.get()) { std::swap(rec.lo, rec.lo2); std::swap(rec.hi, rec.hi2); }  // final data -> rec.lo/hi
        CBLX_HIP(hipStreamSynchronize(c->stream));  // the pass tables are released here
    }
    rec.lo2.reset();
    rec.hi2.reset();
    // -- KRN-4: bitvector, rank directory, bucket table
    {
        StageTimer t(c, ST_DIR);
        Buf<u32> popc(c->pool, nwords);
        nr.bv = Buf<u64>(c->pool, nwords);
        nr.rank_dir = Buf<u64>(c->pool, nwords + 1);
        CBLX_HIP(hipMemsetAsync(nr.bv.get(), 0, nwords * 8, c->stream));
        CBLX_HIP(hipMemsetAsync(popc.get(), 0, nwords * 4, c->stream));
        if (!have_dense) {  // boundaries from a scan of the sorted records (more groups than the table method takes)
            CBLX_HIP(hipMemsetAsync(start_dense.get(), 0xFF, nprefix * 4, c->stream));
            if constexpr (DROP_HI)
                hipLaunchKernelGGL(k_boundaries_seg, grid1(ceil_div(N, 4), 256), dim3(256), 0, c->stream, lo, N, P.SB, RB, seg_start.get(), start_dense.get());
            else
                hipLaunchKernelGGL(k_boundaries<HiT>, grid1(ceil_div(N, 4), 256), dim3(256), 0, c->stream, lo, hi, N, P.SB, P.PB, start_dense.get());
        }
        hipLaunchKernelGGL(k_bitvector, grid1(std::max<u64>(nprefix, 64), 256), dim3(256), 0, c->stream, start_dense.get(), nprefix, nr.bv.get(), popc.get());
        nr.nb = exclusive_scan<u64>(c, popc.get(), nwords, nr.rank_dir.get());
        nr.prefix = Buf<u32>(c->pool, nr.nb + 1);
        nr.start = Buf<u64>(c->pool, nr.nb + 1);
        nr.cnt = Buf<u32>(c->pool, nr.nb + 1);
        nr.kind = Buf<u8>(c->pool, nr.nb + 1);
        hipLaunchKernelGGL(k_bucket_table, grid1(nprefix, 256), dim3(256), 0, c->stream, start_dense.get(), nprefix, nr.bv.get(), nr.rank_dir.get(), nr.prefix.get(), nr.start.get());
        hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, nr.start.get() + nr.nb, N);
        CBLX_HIP(hipGetLastError());
    }
}

// KRN-3 over the runs of `nr` (run of a prefix = [its resident suffixes as stored][the new words in stream order]) in the
// arena a_lo / a_hi: per-bucket dedup / sort by size class; fills nr.cnt, nr.kind, nr.count. `old` = the resident index
// the runs were built against (tells which buckets are untouched and which are Tries already).
template <typename C> void bucket_stage(cblx_ctx* c, Resident& nr, u64* a_lo, typename C::HiT* a_hi, const DirView& old) {
    typedef typename C::HiT HiT;
    const Consts& P = c->P;
    {
    const u64 nb = nr.nb;
    Buf<BDesc> lists(c->pool, (size_t)CLS_N * std::max<u64>(nb, 1));
    Buf<u32> list_n(c->pool, CLS_N);
    Buf<u32> res_count(c->pool, nb + 1);
    Buf<u8> res_kind(c->pool, nb + 1);
    CBLX_HIP(hipMemsetAsync(list_n.get(), 0, CLS_N * 4, c->stream));
    hipLaunchKernelGGL(k_classify, grid1(nb, CLASSIFY_THREADS), dim3(CLASSIFY_THREADS), 0, c->stream, nb, C::WS ? 512u : 1024u, nr.prefix.get(), nr.start.get(), old,
                       res_count.get(), res_kind.get(), nr.cnt.get(), nr.kind.get(), lists.get(), list_n.get());
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
    {
        StageTimer t(c, ST_BMED);
        // fast path (counting sort on the top suffix bits + in-sub-bucket ranking); skewed buckets come back via `retry`
        Buf<BDesc> retry(c->pool, std::max<u64>(nb, 1));
        Buf<u32> retry_n(c->pool, 1);
        CBLX_HIP(hipMemsetAsync(retry_n.get(), 0, 4, c->stream));
        auto msd = [&](auto packed_tag) {
            constexpr bool PK = decltype(packed_tag)::value;
            if (ln[CLS_M16])
                hipLaunchKernelGGL((k_bucket_msd<64, 128, PK, C::WS, HiT>), dim3(ln[CLS_M16]), dim3(64), 0, c->stream, lists.get() + (size_t)CLS_M16 * nb,
                                   list_n.get() + CLS_M16, a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), retry.get(), retry_n.get());
            if (ln[CLS_M64])
                hipLaunchKernelGGL((k_bucket_msd<64, 512, PK, C::WS, HiT>), dim3(ln[CLS_M64]), dim3(64), 0, c->stream, lists.get() + (size_t)CLS_M64 * nb,
                                   list_n.get() + CLS_M64, a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), retry.get(), retry_n.get());
            if (ln[CLS_M128])
                hipLaunchKernelGGL((k_bucket_msd<128, 1024, PK, C::WS, HiT>), dim3(ln[CLS_M128]), dim3(128), 0, c->stream, lists.get() + (size_t)CLS_M128 * nb,
                                   list_n.get() + CLS_M128, a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), retry.get(), retry_n.get());
            if (ln[CLS_M256])
                hipLaunchKernelGGL((k_bucket_msd<256, 2048, PK, C::WS, HiT>), dim3(ln[CLS_M256]), dim3(256), 0, c->stream, lists.get() + (size_t)CLS_M256 * nb,
                                   list_n.get() + CLS_M256, a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), retry.get(), retry_n.get());
            if (ln[CLS_M512])
                hipLaunchKernelGGL((k_bucket_msd<512, 4096, PK, C::WS, HiT>), dim3(ln[CLS_M512]), dim3(512), 0, c->stream, lists.get() + (size_t)CLS_M512 * nb,
                                   list_n.get() + CLS_M512, a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), retry.get(), retry_n.get());
        };
        if constexpr (!C::WS) {
            if (P.SB + PK_BITS <= 64) msd(std::true_type()); else msd(std::false_type());
        } else {
            msd(std::false_type());
        }
        const u32 nretry = (ln[CLS_M64] || ln[CLS_M128] || ln[CLS_M256] || ln[CLS_M512]) ? d2h<u32>(c, retry_n.get()) : 0u;
        if (nretry)
            hipLaunchKernelGGL((k_bucket_medium<512, C::WS, HiT>), dim3(nretry), dim3(512), 0, c->stream, retry.get(), retry_n.get(), a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), MergeArgs{});
        if constexpr (!C::WS) if (ln[CLS_M1024])  // 128-bit suffixes: 8192 keys + indices exceed the 160 KiB LDS, such runs go to the huge path
            hipLaunchKernelGGL((k_bucket_medium<1024, C::WS, HiT>), dim3(ln[CLS_M1024]), dim3(1024), 0, c->stream, lists.get() + (size_t)CLS_M1024 * nb,
                               list_n.get() + CLS_M1024, a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), MergeArgs{});
        CBLX_HIP(hipStreamSynchronize(c->stream));  // retry buffers die here
    }
    if (ln[CLS_HUGE]) {
        StageTimer t(c, ST_BHUGE);
        const u32 nh = ln[CLS_HUGE];
        std::vector<BDesc> hl = d2h_vec<BDesc>(c, lists.get() + (size_t)CLS_HUGE * nb, nh);
        std::vector<u64> so(nh);
        u64 tot = 0;
        for (u32 i = 0; i < nh; ++i) { so[i] = tot; tot += hl[i].c & ~BDESC_TRIE; }
        Buf<u64> d_so(c->pool, nh), s_alo(c->pool, tot), s_blo(c->pool, tot), s_ahi(c->pool, C::WS ? tot : 1), s_bhi(c->pool, C::WS ? tot : 1);
        Buf<u32> s_aidx(c->pool, tot), s_bidx(c->pool, tot);
        h2d(c, d_so.get(), so.data(), nh);
        hipLaunchKernelGGL((k_bucket_huge<C::WS, HiT>), dim3(nh), dim3(256), 0, c->stream, lists.get() + (size_t)CLS_HUGE * nb, list_n.get() + CLS_HUGE,
                           d_so.get(), a_lo, a_hi, P.SB, s_alo.get(), s_ahi.get(), s_aidx.get(), s_blo.get(), s_bhi.get(),
                           s_bidx.get(), nr.cnt.get(), nr.kind.get(), MergeArgs{});
        CBLX_HIP(hipStreamSynchronize(c->stream));
    }
    }
    CBLX_HIP(hipGetLastError());
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
template <typename C> void pipeline(cblx_ctx* c, Records& rec, u64 N, Buf<u32> countsA = Buf<u32>()) {
    typedef typename C::HiT HiT;
    constexpr bool WS = C::WS;
    const Consts& P = c->P;
    Resident nb_;  // directory of the batch
    partition_and_directory<C>(c, rec, N, std::move(countsA), nb_);
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
        bucket_stage<C>(c, nb_, rec.lo.get(), (HiT*)rec.hi.get(), c->res.view());
        adopt_arena(nb_);
        c->res = std::move(nb_);
        return;
    }
    const Resident& s = c->res;
    if (s.count + N >= 0xFFFFFFF0ull) throw Error(CBLX_ERANGE, "more than 2^32-16 words in one index are not supported yet");
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
        hipLaunchKernelGGL((k_merge_gather<WS>), dim3((unsigned)ceil_div(nr.nb, 4)), dim3(256), 0, c->stream, nr.nb, nr.start.get(), m_cs.get(), m_sstart.get(), m_ostart.get(),
                           s.a_lo.get(), s.a_hi.get(), nb_.a_lo.get(), nb_.a_hi.get(), nr.a_lo.get(), nr.a_hi.get());
        CBLX_HIP(hipGetLastError());
    }
    bucket_stage<C>(c, nr, nr.a_lo.get(), WS ? (HiT*)nr.a_hi.get() : (HiT*)nullptr, s.view());
    CBLX_HIP(hipStreamSynchronize(c->stream));  // the batch and the table buffers are released at scope exit
    c->res = std::move(nr);
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
};
void plan_chunks(cblx_ctx* c, const u8*& d_bases, const u64* d_offsets, u64 nseq, ChunkPlan& pl) {
    StageTimer t(c, ST_CHUNKS);
    const Consts& P = c->P;
    // offsets may start anywhere in the buffer (a slice of a larger batch): work relative to the 16-byte aligned
    // position below offsets[0] so that the tile grid and the validity scan cover only this slice
    const u64 first = d2h<u64>(c, d_offsets);
    pl.bias = first & ~(u64)15;
    const u64 last = d2h<u64>(c, d_offsets + nseq);
    if (last < first) throw Error(CBLX_EINVAL, "offsets must be non-decreasing");
    pl.total_bases = last - pl.bias;
    d_bases += pl.bias;
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
    pl.ndirty = d2h<u32>(c, ndirty.get());
    if (pl.ndirty)
        hipLaunchKernelGGL(k_dirty_count, grid1(pl.nchunks, 256), dim3(256), 0, c->stream, d_bases, pl.chunk_start.get(), pl.chunk_len.get(),
                           pl.dirty.get(), pl.nchunks, P.K, chunk_nk.get());
    pl.kmer_off = Buf<u64>(c->pool, pl.nchunks + 1);
    pl.n_kmers = exclusive_scan<u64>(c, chunk_nk.get(), pl.nchunks, pl.kmer_off.get());
    hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, pl.kmer_off.get() + pl.nchunks, pl.n_kmers);
    const u64 ntiles = ceil_div(pl.total_bases, ENC_TILE_BYTES);
    pl.tile_first = Buf<u32>(c->pool, ntiles + 2);
    hipLaunchKernelGGL(k_tile_first_chunk, grid1(ntiles + 1, 256), dim3(256), 0, c->stream, pl.chunk_start.get(), pl.nchunks, ntiles, pl.tile_first.get());
    CBLX_HIP(hipGetLastError());
    CBLX_HIP(hipStreamSynchronize(c->stream));  // temporaries (nch, err, chunk_base, chunk_nk, ndirty) die here
}
template <typename C> void encode(cblx_ctx* c, const u8* d_bases, const ChunkPlan& pl, u64* out_lo, typename C::HiT* out_hi, u64 out_base,
                                  EncHist eh = EncHist{}) {
    typedef typename C::HiT HiT;
    StageTimer t(c, ST_ENCODE);
    const u64 ntiles = ceil_div(pl.total_bases, ENC_TILE_BYTES);
    if (ntiles)
        hipLaunchKernelGGL((k_encode<C::WIDE, HiT>), dim3((unsigned)ntiles), dim3(ENC_THREADS), 0, c->stream, d_bases, pl.total_bases, pl.chunk_start.get(),
                           pl.chunk_len.get(), pl.kmer_off.get(), pl.ndirty ? pl.dirty.get() : (const u8*)nullptr, pl.tile_first.get(), c->P, out_lo, out_hi, out_base, eh);
    if (pl.ndirty)
        hipLaunchKernelGGL((k_encode_dirty<C::WIDE, HiT>), grid1(pl.nchunks, 64), dim3(64), 0, c->stream, d_bases, pl.chunk_start.get(), pl.chunk_len.get(),
                           pl.kmer_off.get(), pl.dirty.get(), pl.nchunks, c->P, out_lo, out_hi, out_base, eh);
    CBLX_HIP(hipGetLastError());
}

void check_aligned16(const void* p, const char* what) {
    if (((uintptr_t)p) & 15) throw Error(CBLX_EINVAL, std::string(what) + " must be 16-byte aligned");
}

void insert_device(cblx_ctx* c, const u8* d_bases, const u64* d_offsets, u64 nseq) {
    if (nseq == 0) return;
    check_aligned16(d_bases, "d_bases");
    dispatch(c->P, [&](auto cfg) {
        typedef decltype(cfg) C;
        ChunkPlan pl;
        plan_chunks(c, d_bases, d_offsets, nseq, pl);
        if (pl.n_kmers == 0) return;
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

// ---- `self |= other`, both resident on this device (src/cbl.rs:433-449 -> src/wordset/set_ops.rs:123-157) ---------
template <typename C> void merge_direct(cblx_ctx* c, const Resident& o) {
    typedef typename C::HiT HiT;
    constexpr bool WS = C::WS;
    const Consts& P = c->P;
    const Resident& s = c->res;
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
    {
        StageTimer t(c, ST_EXPAND);
        hipLaunchKernelGGL((k_merge_gather<WS>), dim3((unsigned)ceil_div(nb, 4)), dim3(256), 0, c->stream, nb, nr.start.get(), m_cs.get(), m_sstart.get(), m_ostart.get(),
                           s.a_lo.get(), s.a_hi.get(), o.a_lo.get(), o.a_hi.get(), nr.a_lo.get(), nr.a_hi.get());
    }
    Buf<BDesc> lists(c->pool, (size_t)CLS_N * std::max<u64>(nb, 1));
    Buf<u32> list_n(c->pool, CLS_N);
    CBLX_HIP(hipMemsetAsync(list_n.get(), 0, CLS_N * 4, c->stream));
    hipLaunchKernelGGL(k_classify_merge, grid1(nb, CLASSIFY_THREADS), dim3(CLASSIFY_THREADS), 0, c->stream, nb, WS ? 512u : 1024u, nr.start.get(), m_cs.get(), m_skind.get(), m_okind.get(),
                       nr.cnt.get(), nr.kind.get(), lists.get(), list_n.get());
    CBLX_HIP(hipGetLastError());
    std::vector<u32> ln = d2h_vec<u32>(c, list_n.get(), CLS_N);
    const MergeArgs ma{m_cs.get(), m_ostart.get(), m_okind.get(), o.a_lo.get(), o.a_hi.get()};
    u64* a_lo = nr.a_lo.get();
    HiT* a_hi = WS ? (HiT*)nr.a_hi.get() : (HiT*)nullptr;
    {
        StageTimer t(c, ST_BMED);
        if (ln[CLS_M256])
            hipLaunchKernelGGL((k_bucket_medium<256, WS, HiT>), dim3(ln[CLS_M256]), dim3(256), 0, c->stream, lists.get() + (size_t)CLS_M256 * nb, list_n.get() + CLS_M256,
                               a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), ma);
        if (ln[CLS_M512])
            hipLaunchKernelGGL((k_bucket_medium<512, WS, HiT>), dim3(ln[CLS_M512]), dim3(512), 0, c->stream, lists.get() + (size_t)CLS_M512 * nb, list_n.get() + CLS_M512,
                               a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), ma);
        if constexpr (!WS) if (ln[CLS_M1024])
            hipLaunchKernelGGL((k_bucket_medium<1024, WS, HiT>), dim3(ln[CLS_M1024]), dim3(1024), 0, c->stream, lists.get() + (size_t)CLS_M1024 * nb,
                               list_n.get() + CLS_M1024, a_lo, a_hi, P.SB, nr.cnt.get(), nr.kind.get(), ma);
        CBLX_HIP(hipGetLastError());
    }
    if (ln[CLS_HUGE]) {
        StageTimer t(c, ST_BHUGE);
        const u32 nh = ln[CLS_HUGE];
        std::vector<BDesc> hl = d2h_vec<BDesc>(c, lists.get() + (size_t)CLS_HUGE * nb, nh);
        std::vector<u64> so(nh);
        u64 tot = 0;
        for (u32 i = 0; i < nh; ++i) { so[i] = tot; tot += hl[i].c & ~BDESC_TRIE; }
        Buf<u64> d_so(c->pool, nh), s_alo(c->pool, tot), s_blo(c->pool, tot), s_ahi(c->pool, WS ? tot : 1), s_bhi(c->pool, WS ? tot : 1);
        Buf<u32> s_aidx(c->pool, tot), s_bidx(c->pool, tot);
        h2d(c, d_so.get(), so.data(), nh);
        hipLaunchKernelGGL((k_bucket_huge<WS, HiT>), dim3(nh), dim3(256), 0, c->stream, lists.get() + (size_t)CLS_HUGE * nb, list_n.get() + CLS_HUGE, d_so.get(), a_lo, a_hi,
                           P.SB, s_alo.get(), s_ahi.get(), s_aidx.get(), s_blo.get(), s_bhi.get(), s_bidx.get(), nr.cnt.get(), nr.kind.get(), ma);
        CBLX_HIP(hipGetLastError());
        CBLX_HIP(hipStreamSynchronize(c->stream));
    }
    {
        Buf<u64> total(c->pool, 1);
        CBLX_HIP(hipMemsetAsync(total.get(), 0, 8, c->stream));
        hipLaunchKernelGGL(k_sum_u32, dim3((unsigned)std::min<u64>(2048, std::max<u64>(1, ceil_div(nb, 256)))), dim3(256), 0, c->stream, nr.cnt.get(), nb, total.get());
        nr.count = d2h<u64>(c, total.get());
    }
    c->res = std::move(nr);
}

// ---- ingest: host sequences -> pending buffers in HBM -----------------------------------------------------------
Xfer& xfer(cblx_ctx* c) {
    if (!c->ing.xfer) c->ing.xfer.reset(new Xfer(c->device));
    return *c->ing.xfer;
}
void ingest_wait(cblx_ctx* c) {  // every DMA issued so far has landed
    Ingest& g = c->ing;
    if (g.s) CBLX_HIP(hipStreamSynchronize(g.s));
    if (g.xfer) g.xfer->sync();
}
void writer_issue(cblx_ctx* c, Ingest::Writer& w, u8* d_dst) {  // hand the current block to the DMA engine
    if (w.fill == 0) return;
    Ingest& g = c->ing;
    CBLX_HIP(hipMemcpyAsync(d_dst + w.issued, w.blk[w.cur], w.fill, hipMemcpyHostToDevice, g.s));
    CBLX_HIP(hipEventRecord(w.ev[w.cur], g.s));
    w.busy[w.cur] = true;
    w.issued += w.fill;
    w.fill = 0;
    w.cur ^= 1;
    if (w.busy[w.cur]) { CBLX_HIP(hipEventSynchronize(w.ev[w.cur])); w.busy[w.cur] = false; }
}
void writer_put(cblx_ctx* c, Ingest::Writer& w, size_t blk_bytes, u8* d_dst, const u8* src, size_t n) {
    Ingest& g = c->ing;
    if (!w.blk[0]) {
        if (!g.s) CBLX_HIP(hipStreamCreateWithFlags(&g.s, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            CBLX_HIP(hipHostMalloc((void**)&w.blk[k], blk_bytes, hipHostMallocDefault));
            CBLX_HIP(hipEventCreateWithFlags(&w.ev[k], hipEventDisableTiming));
        }
        w.cap = blk_bytes;
    }
    while (n) {
        const size_t m = std::min(n, w.cap - w.fill);
        std::memcpy(w.blk[w.cur] + w.fill, src, m);
        w.fill += m; src += m; n -= m;
        if (w.fill == w.cap) writer_issue(c, w, d_dst);
    }
}
// room for `add_bytes` more bases and `add_seqs` more sequences
void ingest_reserve(cblx_ctx* c, u64 add_bytes, u64 add_seqs) {
    Ingest& g = c->ing;
    const u64 need_b = g.nbytes + add_bytes + 64, need_o = g.nseq + add_seqs + 1;
    if (g.d_bases.n < need_b) {
        Buf<u8> nb(c->pool, std::max<u64>({need_b, 2 * (u64)g.d_bases.n, 1u << 20}));
        if (g.nbytes) {
            ingest_wait(c);
            CBLX_HIP(hipMemcpyAsync(nb.get(), g.d_bases.get(), g.wb.issued, hipMemcpyDeviceToDevice, c->stream));
            CBLX_HIP(hipStreamSynchronize(c->stream));
        }
        g.d_bases = std::move(nb);
    }
    if (g.d_off.n < need_o) {
        Buf<u64> no(c->pool, std::max<u64>({need_o, 2 * (u64)g.d_off.n, 1u << 14}));
        ingest_wait(c);
        CBLX_HIP(hipMemsetAsync(no.get(), 0, 8, c->stream));
        if (g.nseq) CBLX_HIP(hipMemcpyAsync(no.get() + 1, g.d_off.get() + 1, g.wo.issued, hipMemcpyDeviceToDevice, c->stream));
        CBLX_HIP(hipStreamSynchronize(c->stream));
        g.d_off = std::move(no);
    }
}
void flush(cblx_ctx* c);
// one sequence (the cblx_insert_seq / FASTA-record granularity)
// a piece of the sequence being enqueued (a FASTA record arrives line by line), then its end
void ingest_bases(cblx_ctx* c, const u8* p, u64 len) {
    Ingest& g = c->ing;
    ingest_reserve(c, len, 1);
    writer_put(c, g.wb, Ingest::BASES_BLK, g.d_bases.get(), p, len);
    g.nbytes += len;
}
void ingest_abort_seq(cblx_ctx* c) {  // drop the bases of an unfinished sequence
    Ingest& g = c->ing;
    const u64 begin = g.nseq ? g.last_end : 0;
    g.nbytes = begin;
    if (begin >= g.wb.issued) g.wb.fill = (size_t)(begin - g.wb.issued);
    else { g.wb.issued = begin; g.wb.fill = 0; }
}
// the queue is bounded: past this many pending bases the batch is inserted (same result: batches go in in order).
// CBLX_INGEST_FLUSH_BYTES overrides it (tests use a tiny value to exercise the incremental path).
u64 ingest_flush_bytes() {
    static const u64 v = [] {
        const char* e = std::getenv("CBLX_INGEST_FLUSH_BYTES");
        const u64 x = e ? std::strtoull(e, nullptr, 10) : 0;
        return x ? x : (2ull << 30);
    }();
    return v;
}
void ingest_end_seq(cblx_ctx* c, u64 flush_at = ingest_flush_bytes()) {
    Ingest& g = c->ing;
    const u64 begin = g.nseq ? g.last_end : 0, len = g.nbytes - begin;
    if (len < c->P.K) {  // src/cbl.rs:329-334; the record is dropped from the queue
        ingest_abort_seq(c);
        throw Error(CBLX_ESHORT, "Sequence size (" + std::to_string(len) + ") is smaller than K (" + std::to_string(c->P.K) + ")");
    }
    const u64 end = g.nbytes;
    writer_put(c, g.wo, Ingest::OFF_BLK, (u8*)(g.d_off.get() + 1), (const u8*)&end, 8);
    g.nseq += 1;
    g.last_end = end;
    if (g.nbytes >= flush_at) flush(c);  // bound the queue (same result: batches are inserted in order)
}
void ingest_seq(cblx_ctx* c, const u8* seq, u64 len) {
    ingest_bases(c, seq, len);
    ingest_end_seq(c);
}
// n sequences at once (offsets already validated)
void ingest_seqs(cblx_ctx* c, const u8* bases, const u64* offsets, u64 n) {
    Ingest& g = c->ing;
    const u64 len = offsets[n] - offsets[0];
    if (len < (1u << 20)) {
        for (u64 i = 0; i < n; ++i) ingest_seq(c, bases + offsets[i], offsets[i + 1] - offsets[i]);
        return;
    }
    ingest_reserve(c, len, n);
    if (!g.s) CBLX_HIP(hipStreamCreateWithFlags(&g.s, hipStreamNonBlocking));
    if (g.wb.blk[0]) writer_issue(c, g.wb, g.d_bases.get());
    if (g.wo.blk[0]) writer_issue(c, g.wo, (u8*)(g.d_off.get() + 1));
    Xfer& x = xfer(c);
    x.h2d_copy(g.d_bases.get() + g.nbytes, bases + offsets[0], len);
    const u64 base = g.nbytes, o0 = offsets[0];
    x.h2d(g.d_off.get() + 1 + g.nseq, n * 8, [&](u8* dst, size_t off, size_t nb) {
        u64* d = (u64*)dst;
        const u64* src = offsets + off / 8 + 1;
        for (size_t j = 0; j < nb / 8; ++j) d[j] = base + (src[j] - o0);
    });
    g.nbytes += len;
    g.nseq += n;
    g.last_end = g.nbytes;
    g.wb.issued = g.nbytes;
    g.wo.issued = g.nseq * 8;
    if (g.nbytes >= ingest_flush_bytes()) flush(c);
}
void ingest_drop(cblx_ctx* c) {  // forget everything enqueued (clear / load)
    Ingest& g = c->ing;
    ingest_wait(c);
    for (Ingest::Writer* w : {&g.wb, &g.wo}) { w->fill = 0; w->issued = 0; w->busy[0] = w->busy[1] = false; }
    g.nbytes = g.nseq = g.last_end = 0;
}
void ingest_destroy(cblx_ctx* c) {
    Ingest& g = c->ing;
    if (g.s) (void)hipStreamSynchronize(g.s);
    g.xfer.reset();
    for (Ingest::Writer* w : {&g.wb, &g.wo})
        for (int k = 0; k < 2; ++k) {
            if (w->ev[k]) (void)hipEventDestroy(w->ev[k]);
            if (w->blk[k]) (void)hipHostFree(w->blk[k]);
            w->ev[k] = nullptr; w->blk[k] = nullptr;
        }
    if (g.s) (void)hipStreamDestroy(g.s);
    g.s = nullptr;
    g.d_bases.reset();
    g.d_off.reset();
}

void flush(cblx_ctx* c) {
    Ingest& g = c->ing;
    const u64 nseq = g.nseq;
    if (nseq == 0) return;
    CBLX_HIP(hipSetDevice(c->device));
    if (g.wb.blk[0]) writer_issue(c, g.wb, g.d_bases.get());
    if (g.wo.blk[0]) writer_issue(c, g.wo, (u8*)(g.d_off.get() + 1));
    ingest_wait(c);
    // the pending queue is consumed even if the insert fails (the reference would have panicked)
    for (Ingest::Writer* w : {&g.wb, &g.wo}) { w->issued = 0; w->busy[0] = w->busy[1] = false; }
    g.nbytes = g.nseq = g.last_end = 0;
    insert_device(c, g.d_bases.get(), g.d_off.get(), nseq);
    CBLX_HIP(hipStreamSynchronize(c->stream));
}

// ---- host-side views of the resident index (export / serialize / merge) ---------------------------------------
struct HostIndex {
    std::vector<u32> prefix, cnt;
    std::vector<u8> kind;
    std::vector<u64> off;  // nb+1 into lo/hi
    std::vector<u64> lo, hi;
};
void download(cblx_ctx* c, HostIndex& h) {
    const Resident& r = c->res;
    h.prefix = d2h_vec<u32>(c, r.prefix.get(), r.nb);
    h.cnt = d2h_vec<u32>(c, r.cnt.get(), r.nb);
    h.kind = d2h_vec<u8>(c, r.kind.get(), r.nb);
    h.off.assign(r.nb + 1, 0);
    for (u64 i = 0; i < r.nb; ++i) h.off[i + 1] = h.off[i] + h.cnt[i];
    const u64 n = h.off[r.nb];
    if (n == 0) { h.lo.clear(); h.hi.clear(); return; }
    Buf<u64> d_off(c->pool, r.nb + 1), d_lo(c->pool, n), d_hi(c->pool, c->P.wide_suffix() ? n : 1);
    h2d(c, d_off.get(), h.off.data(), r.nb + 1);
    hipLaunchKernelGGL(k_gather_dense, grid1(n, 256), dim3(256), 0, c->stream, n, r.nb, d_off.get(), r.start.get(), r.a_lo.get(),
                       c->P.wide_suffix() ? r.a_hi.get() : (const u64*)nullptr, c->P.SB, d_lo.get(), c->P.wide_suffix() ? d_hi.get() : (u64*)nullptr);
    CBLX_HIP(hipGetLastError());
    CBLX_HIP(hipStreamSynchronize(c->stream));
    h.lo.resize(n);
    xfer(c).d2h_copy(h.lo.data(), d_lo.get(), n * 8);  // pinned lanes (a pageable hipMemcpy runs at a few GB/s)
    if (c->P.wide_suffix()) { h.hi.resize(n); xfer(c).d2h_copy(h.hi.data(), d_hi.get(), n * 8); } else h.hi.clear();
}
// replace the resident index by a host-built one (load / merge): dense arena, directory built on the host
void upload(cblx_ctx* c, const HostIndex& h) {
    const Consts& P = c->P;
    Resident nr;
    nr.nb = h.prefix.size();
    const u64 n = h.off.empty() ? 0 : h.off.back();
    nr.count = n;
    const u64 nprefix = 1ull << P.PB, nwords = std::max<u64>(1, nprefix / 64);
    std::vector<u64> bv(nwords, 0), rd(nwords + 1, 0);
    for (u64 i = 0; i < nr.nb; ++i) {
        if (h.prefix[i] >= nprefix) throw Error(CBLX_EFORMAT, "prefix out of range for PREFIX_BITS");
        if (i && h.prefix[i] <= h.prefix[i - 1]) throw Error(CBLX_EFORMAT, "prefixes are not strictly ascending");
        bv[h.prefix[i] >> 6] |= 1ull << (h.prefix[i] & 63);
    }
    for (u64 w = 0; w < nwords; ++w) rd[w + 1] = rd[w] + (u64)__builtin_popcountll(bv[w]);
    nr.bv = Buf<u64>(c->pool, nwords);
    nr.rank_dir = Buf<u64>(c->pool, nwords + 1);
    nr.prefix = Buf<u32>(c->pool, nr.nb + 1);
    nr.start = Buf<u64>(c->pool, nr.nb + 1);
    nr.cnt = Buf<u32>(c->pool, nr.nb + 1);
    nr.kind = Buf<u8>(c->pool, nr.nb + 1);
    nr.a_lo = Buf<u64>(c->pool, n + 2);
    if (P.wide_suffix()) nr.a_hi = Buf<u64>(c->pool, n + 2);
    h2d(c, nr.bv.get(), bv.data(), nwords);
    h2d(c, nr.rank_dir.get(), rd.data(), nwords + 1);
    h2d(c, nr.prefix.get(), h.prefix.data(), nr.nb);
    h2d(c, nr.start.get(), h.off.data(), nr.nb + 1);
    h2d(c, nr.cnt.get(), h.cnt.data(), nr.nb);
    h2d(c, nr.kind.get(), h.kind.data(), nr.nb);
    xfer(c).h2d_copy(nr.a_lo.get(), h.lo.data(), n * 8);
    if (P.wide_suffix()) xfer(c).h2d_copy(nr.a_hi.get(), h.hi.data(), n * 8);
    xfer(c).sync();
    CBLX_HIP(hipStreamSynchronize(c->stream));
    c->res = std::move(nr);
}

// ---- bincode 1.3 DefaultOptions (varint, little endian): src/cbl.rs:132-135 -----------------------------------
struct Sink {
    u8* buf;
    u64 cap, pos = 0;
    Sink(u8* b, u64 c) : buf(b), cap(c) {}
    inline void u8_(u8 v) { if (buf && pos < cap) buf[pos] = v; ++pos; }
    inline void raw(const u8* p, u64 n) { if (buf && pos + n <= cap) memcpy(buf + pos, p, n); pos += n; }
    inline void varint(u64 v) {
        if (v <= 250) { u8_((u8)v); return; }
        int nb = v < (1ull << 16) ? 2 : v < (1ull << 32) ? 4 : 8;
        u8_(nb == 2 ? 0xFB : nb == 4 ? 0xFC : 0xFD);
        for (int i = 0; i < nb; ++i) u8_((u8)(v >> (8 * i)));
    }
};
struct SfxView {
    const u64* lo;
    const u64* hi;
    inline u128 at(u64 i) const { return hi ? (((u128)hi[i] << 64) | lo[i]) : (u128)lo[i]; }
};
// Trie node over sorted suffixes [a, b) that agree on their top `depth` bytes (src/trie.rs:53-57 derive,
// src/bitvector/tiny/mod.rs:97-105): varint(c) | c byte values | varint(#children) | children...
void emit_trie(Sink& s, const SfxView& v, u64 a, u64 b, u32 depth, u32 BYTES) {
    const u32 shift = 8 * (BYTES - 1 - depth);
    u8 vals[256];
    u64 starts[257];
    u32 c = 0;
    u64 i = a;
    while (i < b) {
        const u8 by = (u8)(v.at(i) >> shift);
        vals[c] = by;
        starts[c++] = i;
        // gallop to the end of this byte's run
        u64 lo = i + 1, hi = b;
        while (lo < hi) {
            u64 mid = (lo + hi) >> 1;
            if ((u8)(v.at(mid) >> shift) == by) lo = mid + 1; else hi = mid;
        }
        i = lo;
    }
    starts[c] = b;
    s.varint(c);
    s.raw(vals, c);
    if (depth + 1 == BYTES) { s.varint(0); return; }
    s.varint(c);
    for (u32 k = 0; k < c; ++k) emit_trie(s, v, starts[k], starts[k + 1], depth + 1, BYTES);
}
void serialize_bucket(const Consts& P, const HostIndex& h, const SfxView& v, u64 r, Sink& s) {
    s.varint(h.prefix[r]);
    const u64 a = h.off[r], b = h.off[r + 1];
    if (h.kind[r] == KIND_VEC) {             // TrieOrVec::Vec  src/trievec/mod.rs:10-11
        s.varint(0);
        s.varint(b - a);
        for (u64 i = a; i < b; ++i) {
            s.varint(P.BYTES);               // SlicedInt::serialize -> serialize_bytes  src/sliced_int.rs:110-114
            u128 x = v.at(i);
            u8 tmp[16];
            for (u32 k = 0; k < P.BYTES; ++k) tmp[k] = (u8)(x >> (8 * k));
            s.raw(tmp, P.BYTES);
        }
    } else {                                 // TrieOrVec::Trie(trie, len)  src/trievec/mod.rs:12
        s.varint(1);
        emit_trie(s, v, a, b, 0, P.BYTES);
        s.varint(b - a);
    }
}
// Serialized form, emitted by a pool of host threads: bucket entries are independent byte ranges, so sizes are
// computed in parallel, prefix-summed, and every bucket is then written at its own offset. (The reference writes
// sequentially through a BufWriter, examples/cbl.rs:132-142; the bytes are the same.)
void serialize_host(const Consts& P, const HostIndex& h, Sink& s) {
    s.u8_(P.canonical ? 1 : 0);                  // CBL.canonical (src/cbl.rs:48)
    s.varint(h.prefix.size());                   // serialize_map(Some(tiered.len()))  src/wordset/mod.rs:388
    const u64 nb = h.prefix.size();
    if (nb == 0) return;
    SfxView v{h.lo.data(), h.hi.empty() ? nullptr : h.hi.data()};
    unsigned nt = std::thread::hardware_concurrency();
    nt = std::max(1u, std::min(nt ? nt : 1u, 64u));
    if (nb < 4096 || h.lo.size() < (1u << 18)) nt = 1;
    // split the buckets into ranges of roughly equal element counts
    std::vector<u64> cut(nt + 1, nb);
    cut[0] = 0;
    const u64 total = h.off[nb];
    for (unsigned t = 1; t < nt; ++t) {
        const u64 target = total / nt * t;
        cut[t] = (u64)(std::lower_bound(h.off.begin(), h.off.begin() + nb, target) - h.off.begin());
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    auto run = [&](auto&& fn) {
        if (nt == 1) { fn(0u); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back(fn, t);
        for (auto& x : th) x.join();
    };
    std::vector<u64> part(nt + 1, 0);
    run([&](unsigned t) {
        Sink cnt(nullptr, 0);
        for (u64 r = cut[t]; r < cut[t + 1]; ++r) serialize_bucket(P, h, v, r, cnt);
        part[t + 1] = cnt.pos;
    });
    for (unsigned t = 0; t < nt; ++t) part[t + 1] += part[t];
    const u64 base = s.pos;
    if (s.buf && base + part[nt] <= s.cap) {
        run([&](unsigned t) {
            Sink out(s.buf + base + part[t], part[t + 1] - part[t]);
            for (u64 r = cut[t]; r < cut[t + 1]; ++r) serialize_bucket(P, h, v, r, out);
        });
    }
    s.pos = base + part[nt];
}
// ---- the same bytes, produced in HBM (kernels_serde.hpp): size pass -> exclusive scan -> emit pass ---------------
struct DevBlob { Buf<u8> bytes; u64 n = 0; };
// false: a bucket is longer than the device emitters handle (SER_CAP1024) -> the caller takes the host path
template <typename C> bool serialize_device(cblx_ctx* c, bool emit, DevBlob& blob) {
    constexpr bool WS = C::WS;
    const Resident& r = c->res;
    const Consts& P = c->P;
    const u64 nb = r.nb;
    u8 hdr[16];
    Sink hs(hdr, sizeof hdr);
    hs.u8_(P.canonical ? 1 : 0);  // CBL.canonical (src/cbl.rs:48)
    hs.varint(nb);                // serialize_map(Some(tiered.len()))  src/wordset/mod.rs:388
    u64 total = 0;
    Buf<u32> size, lists, list_n;
    Buf<u64> off;
    std::vector<u32> ln(SER_NCLS, 0);
    const u64 *a_lo = r.a_lo.get(), *a_hi = WS ? r.a_hi.get() : (const u64*)nullptr;
    auto buckets = [&](auto em, u8* body) {
        constexpr bool EM = decltype(em)::value;
        if (ln[SER_C64])
            hipLaunchKernelGGL((k_serde_bucket<64, 16, WS, EM>), dim3(ln[SER_C64]), dim3(64), 0, c->stream, lists.get() + (size_t)SER_C64 * nb, list_n.get() + SER_C64,
                               r.prefix.get(), r.start.get(), r.cnt.get(), r.kind.get(), a_lo, a_hi, P.SB, P.BYTES, size.get(), off.get(), body);
        if (ln[SER_C256])
            hipLaunchKernelGGL((k_serde_bucket<256, 16, WS, EM>), dim3(ln[SER_C256]), dim3(256), 0, c->stream, lists.get() + (size_t)SER_C256 * nb, list_n.get() + SER_C256,
                               r.prefix.get(), r.start.get(), r.cnt.get(), r.kind.get(), a_lo, a_hi, P.SB, P.BYTES, size.get(), off.get(), body);
        if (ln[SER_C1024])
            hipLaunchKernelGGL((k_serde_bucket<1024, 8, WS, EM>), dim3(ln[SER_C1024]), dim3(1024), 0, c->stream, lists.get() + (size_t)SER_C1024 * nb, list_n.get() + SER_C1024,
                               r.prefix.get(), r.start.get(), r.cnt.get(), r.kind.get(), a_lo, a_hi, P.SB, P.BYTES, size.get(), off.get(), body);
        CBLX_HIP(hipGetLastError());
    };
    if (nb) {
        size = Buf<u32>(c->pool, nb);
        lists = Buf<u32>(c->pool, (size_t)SER_NCLS * nb);
        list_n = Buf<u32>(c->pool, SER_NCLS);
        off = Buf<u64>(c->pool, nb + 1);
        CBLX_HIP(hipMemsetAsync(list_n.get(), 0, SER_NCLS * 4, c->stream));
        hipLaunchKernelGGL((k_serde_tiny<WS, false>), grid1(nb, CLASSIFY_THREADS), dim3(CLASSIFY_THREADS), 0, c->stream, nb, r.prefix.get(), r.start.get(), r.cnt.get(), r.kind.get(), a_lo, a_hi,
                           P.SB, P.BYTES, size.get(), (const u64*)nullptr, (u8*)nullptr, lists.get(), list_n.get());
        CBLX_HIP(hipGetLastError());
        ln = d2h_vec<u32>(c, list_n.get(), SER_NCLS);
        if (ln[SER_HOST]) return false;
        buckets(std::false_type(), nullptr);
        total = exclusive_scan<u64>(c, size.get(), nb, off.get());
    }
    blob.n = hs.pos + total;
    if (!emit) return true;
    blob.bytes = Buf<u8>(c->pool, blob.n + 16);
    CBLX_HIP(hipMemcpyAsync(blob.bytes.get(), hdr, hs.pos, hipMemcpyHostToDevice, c->stream));
    if (nb) {
        u8* body = blob.bytes.get() + hs.pos;
        hipLaunchKernelGGL((k_serde_tiny<WS, true>), grid1(nb, CLASSIFY_THREADS), dim3(CLASSIFY_THREADS), 0, c->stream, nb, r.prefix.get(), r.start.get(), r.cnt.get(), r.kind.get(), a_lo, a_hi,
                           P.SB, P.BYTES, size.get(), off.get(), body, (u32*)nullptr, (u32*)nullptr);
        buckets(std::true_type(), body);
    }
    CBLX_HIP(hipStreamSynchronize(c->stream));
    return true;
}
bool serialize_device(cblx_ctx* c, bool emit, DevBlob& blob) {
    if (const char* e = std::getenv("CBLX_HOST_SERDE")) if (e[0] == '1') return false;  // test hook: force the host emitter
    bool ok = false;
    dispatch(c->P, [&](auto cfg) { ok = serialize_device<decltype(cfg)>(c, emit, blob); });
    return ok;
}

struct Src {
    const u8* p;
    const u8* end;
    u8 u8_() { if (p >= end) throw Error(CBLX_EFORMAT, "index: unexpected end of data"); return *p++; }
    u64 varint() {
        u8 t = u8_();
        if (t <= 250) return t;
        int nb = t == 0xFB ? 2 : t == 0xFC ? 4 : t == 0xFD ? 8 : 0;
        if (!nb) throw Error(CBLX_EFORMAT, "index: bad varint tag");
        u64 v = 0;
        for (int i = 0; i < nb; ++i) v |= (u64)u8_() << (8 * i);
        return v;
    }
};
void parse_trie(Src& s, u32 depth, u32 BYTES, u128 acc, std::vector<u128>& out) {
    u64 c = s.varint();
    if (c > 256) throw Error(CBLX_EFORMAT, "index: trie node with more than 256 entries");
    u8 vals[256];
    for (u64 i = 0; i < c; ++i) vals[i] = s.u8_();
    u64 nc = s.varint();
    const u32 shift = 8 * (BYTES - 1 - depth);
    if (depth + 1 == BYTES) {
        if (nc != 0) throw Error(CBLX_EFORMAT, "index: leaf trie node with children");
        for (u64 i = 0; i < c; ++i) out.push_back(acc | ((u128)vals[i] << shift));
        return;
    }
    if (nc != c) throw Error(CBLX_EFORMAT, "index: trie node children count mismatch");
    for (u64 i = 0; i < c; ++i) parse_trie(s, depth + 1, BYTES, acc | ((u128)vals[i] << shift), out);
}
void parse_index(const Consts& P, const u8* data, u64 len, HostIndex& h, bool& canonical) {
    Src s{data, data + len};
    canonical = s.u8_() != 0;
    const u64 nb = s.varint();
    h.off.assign(1, 0);
    const bool wide = P.wide_suffix();
    std::vector<u128> tmp;
    for (u64 r = 0; r < nb; ++r) {
        h.prefix.push_back((u32)s.varint());
        const u64 tag = s.varint();
        if (tag == 0) {
            const u64 n = s.varint();
            for (u64 i = 0; i < n; ++i) {
                const u64 nbts = s.varint();
                u128 x = 0;
                for (u64 k = 0; k < nbts; ++k) { u8 b = s.u8_(); if (k < P.BYTES) x |= (u128)b << (8 * k); }
                h.lo.push_back((u64)x);
                if (wide) h.hi.push_back((u64)(x >> 64));
            }
            h.kind.push_back(KIND_VEC);
            h.cnt.push_back((u32)n);
        } else if (tag == 1) {
            tmp.clear();
            parse_trie(s, 0, P.BYTES, 0, tmp);
            const u64 n = s.varint();
            if (n != tmp.size()) throw Error(CBLX_EFORMAT, "index: trie length field does not match its contents");
            for (u128 x : tmp) { h.lo.push_back((u64)x); if (wide) h.hi.push_back((u64)(x >> 64)); }
            h.kind.push_back(KIND_TRIE);
            h.cnt.push_back((u32)n);
        } else throw Error(CBLX_EFORMAT, "index: bad TrieOrVec tag");
        h.off.push_back(h.lo.size());
    }
    if (s.p != s.end) throw Error(CBLX_EFORMAT, "index: trailing bytes");  // reject_trailing_bytes
}

// ---- index bytes -> resident index, streamed (cblx_load): one pass over the bytes, elements go to HBM as they are
// decoded. The format is a sequential pre-order walk (no lengths to skip by), so the walk itself stays on one host
// thread; everything around it (pinned double buffering, DMA, directory upload) overlaps with it.
struct StreamUp {  // single producer -> device array of u64
    static constexpr size_t CAP = 1u << 20;  // elements per pinned block
    cblx_ctx* c;
    hipStream_t s = nullptr;
    u64* blk[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int cur = 0;
    size_t fill = 0;
    u64 issued = 0;
    Buf<u64> dev;
    StreamUp(cblx_ctx* ctx, u64 guess) : c(ctx) {
        CBLX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            CBLX_HIP(hipHostMalloc((void**)&blk[k], CAP * 8, hipHostMallocDefault));
            CBLX_HIP(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        }
        dev = Buf<u64>(c->pool, guess + 2);
    }
    StreamUp(const StreamUp&) = delete;
    ~StreamUp() {
        if (s) (void)hipStreamSynchronize(s);
        for (int k = 0; k < 2; ++k) { if (ev[k]) (void)hipEventDestroy(ev[k]); if (blk[k]) (void)hipHostFree(blk[k]); }
        if (s) (void)hipStreamDestroy(s);
    }
    inline u64* room(size_t need) { if (fill + need > CAP) issue(); return blk[cur] + fill; }  // need <= CAP
    inline void commit(size_t k) { fill += k; }
    void issue() {
        if (fill == 0) return;
        if (dev.n < issued + fill + 2) {
            Buf<u64> nd(c->pool, std::max<u64>(2 * (u64)dev.n, issued + fill + 2));
            CBLX_HIP(hipStreamSynchronize(s));
            if (issued) CBLX_HIP(hipMemcpyAsync(nd.get(), dev.get(), issued * 8, hipMemcpyDeviceToDevice, s));
            CBLX_HIP(hipStreamSynchronize(s));
            dev = std::move(nd);
        }
        CBLX_HIP(hipMemcpyAsync(dev.get() + issued, blk[cur], fill * 8, hipMemcpyHostToDevice, s));
        CBLX_HIP(hipEventRecord(ev[cur], s));
        busy[cur] = true;
        issued += fill;
        fill = 0;
        cur ^= 1;
        if (busy[cur]) { CBLX_HIP(hipEventSynchronize(ev[cur])); busy[cur] = false; }
    }
    Buf<u64> finish() { issue(); CBLX_HIP(hipStreamSynchronize(s)); return std::move(dev); }
};
inline u64 load_le64(const u8* p) { u64 v; std::memcpy(&v, p, 8); return v; }

// directory of a loaded / host-merged index: bitvector + rank directory + per-rank tables, arena supplied by the caller
void install_index(cblx_ctx* c, const std::vector<u32>& prefix, const std::vector<u32>& cnt, const std::vector<u8>& kind, Buf<u64>&& a_lo, Buf<u64>&& a_hi) {
    const Consts& P = c->P;
    Resident nr;
    nr.nb = prefix.size();
    const u64 nprefix = 1ull << P.PB, nwords = std::max<u64>(1, nprefix / 64);
    std::vector<u64> bv(nwords, 0), rd(nwords + 1, 0), start(nr.nb + 1, 0);
    for (u64 i = 0; i < nr.nb; ++i) {
        if (prefix[i] >= nprefix) throw Error(CBLX_EFORMAT, "prefix out of range for PREFIX_BITS");
        if (i && prefix[i] <= prefix[i - 1]) throw Error(CBLX_EFORMAT, "prefixes are not strictly ascending");
        bv[prefix[i] >> 6] |= 1ull << (prefix[i] & 63);
        start[i + 1] = start[i] + cnt[i];
    }
    for (u64 w = 0; w < nwords; ++w) rd[w + 1] = rd[w] + (u64)__builtin_popcountll(bv[w]);
    nr.count = start[nr.nb];
    nr.bv = Buf<u64>(c->pool, nwords);
    nr.rank_dir = Buf<u64>(c->pool, nwords + 1);
    nr.prefix = Buf<u32>(c->pool, nr.nb + 1);
    nr.start = Buf<u64>(c->pool, nr.nb + 1);
    nr.cnt = Buf<u32>(c->pool, nr.nb + 1);
    nr.kind = Buf<u8>(c->pool, nr.nb + 1);
    Xfer& x = xfer(c);
    x.h2d_copy(nr.bv.get(), bv.data(), nwords * 8);
    x.h2d_copy(nr.rank_dir.get(), rd.data(), (nwords + 1) * 8);
    x.h2d_copy(nr.prefix.get(), prefix.data(), nr.nb * 4);
    x.h2d_copy(nr.start.get(), start.data(), (nr.nb + 1) * 8);
    x.h2d_copy(nr.cnt.get(), cnt.data(), nr.nb * 4);
    x.h2d_copy(nr.kind.get(), kind.data(), nr.nb);
    x.sync();
    nr.a_lo = std::move(a_lo);
    if (P.wide_suffix()) nr.a_hi = std::move(a_hi);
    c->res = std::move(nr);
}

template <bool WS> void load_stream(cblx_ctx* c, const u8* data, u64 len, bool& canonical) {
    const Consts& P = c->P;
    const u32 BYTES = P.BYTES;
    Src s{data, data + len};
    canonical = s.u8_() != 0;
    const u64 nb = s.varint();
    const u64 nprefix = 1ull << P.PB;
    if (nb > nprefix) throw Error(CBLX_EFORMAT, "index: more buckets than prefixes (wrong PREFIX_BITS?)");
    std::vector<u32> prefix(nb), cnt(nb);
    std::vector<u8> kind(nb);
    StreamUp lo(c, len / 6 + 1024);
    std::unique_ptr<StreamUp> hi;
    if (WS) hi.reset(new StreamUp(c, len / 6 + 1024));
    const u64 lo_mask = BYTES >= 8 ? ~0ull : ((1ull << (8 * BYTES)) - 1ull);
    const u64 hi_mask = WS ? ((BYTES >= 16) ? ~0ull : ((1ull << (8 * (BYTES - 8))) - 1ull)) : 0ull;
    u64 total = 0;
    for (u64 r = 0; r < nb; ++r) {
        const u64 p = s.varint();
        if (p >= nprefix) throw Error(CBLX_EFORMAT, "prefix out of range for PREFIX_BITS");
        prefix[r] = (u32)p;
        const u64 tag = s.varint();
        u64 n = 0;
        if (tag == 0) {  // Vec: varint(n) then n x (varint(BYTES) | BYTES little-endian bytes), stored order
            n = s.varint();
            if (n > 0xFFFFFFF0ull) throw Error(CBLX_EFORMAT, "index: bucket too long");
            u64 left = n;
            while (left) {
                const size_t k = (size_t)std::min<u64>(left, StreamUp::CAP);
                u64* ol = lo.room(k);
                u64* oh = WS ? hi->room(k) : nullptr;
                // fast path: every element has the expected length byte and 16 readable bytes follow the chunk
                if ((u64)(s.end - s.p) >= (u64)k * (1 + BYTES) + 16) {
                    const u8* q = s.p;
                    bool regular = true;
                    for (size_t i = 0; i < k; ++i, q += 1 + BYTES) {
                        regular &= q[0] == BYTES;
                        ol[i] = load_le64(q + 1) & lo_mask;
                        if (WS) oh[i] = load_le64(q + 9) & hi_mask;
                    }
                    if (regular) { s.p = q; lo.commit(k); if (WS) hi->commit(k); left -= k; continue; }
                }
                for (size_t i = 0; i < k; ++i) {  // general path (length byte != BYTES, or the tail of the input)
                    const u64 nbts = s.varint();
                    u128 x = 0;
                    for (u64 b = 0; b < nbts; ++b) { const u8 v = s.u8_(); if (b < BYTES) x |= (u128)v << (8 * b); }
                    ol[i] = (u64)x;
                    if (WS) oh[i] = (u64)(x >> 64);
                }
                lo.commit(k);
                if (WS) hi->commit(k);
                left -= k;
            }
            kind[r] = KIND_VEC;
        } else if (tag == 1) {  // Trie: pre-order nodes (explicit stack), then varint(len)
            struct Fr { const u8* vals; u32 c, i; };
            Fr st[16];
            u32 d = 0;
            u64 alo = 0, ahi = 0;
            auto set_byte = [&](u32 depth, u8 b) {
                u32 sh = 8 * (BYTES - 1 - depth);
                if (sh < 64) alo = (alo & ~(0xFFull << sh)) | ((u64)b << sh);
                else { sh -= 64; ahi = (ahi & ~(0xFFull << sh)) | ((u64)b << sh); }
            };
            for (;;) {
                const u64 cc = s.varint();
                if (cc > 256) throw Error(CBLX_EFORMAT, "index: trie node with more than 256 entries");
                if ((u64)(s.end - s.p) < cc) throw Error(CBLX_EFORMAT, "index: unexpected end of data");
                const u8* vals = s.p;
                s.p += cc;
                const u64 nc = s.varint();
                bool descend = false;
                if (d + 1 == BYTES) {
                    if (nc != 0) throw Error(CBLX_EFORMAT, "index: leaf trie node with children");
                    u64* ol = lo.room((size_t)cc);
                    for (u64 i = 0; i < cc; ++i) ol[i] = alo | vals[i];
                    lo.commit((size_t)cc);
                    if (WS) { u64* oh = hi->room((size_t)cc); for (u64 i = 0; i < cc; ++i) oh[i] = ahi; hi->commit((size_t)cc); }
                    n += cc;
                } else {
                    if (nc != cc) throw Error(CBLX_EFORMAT, "index: trie node children count mismatch");
                    if (cc) { st[d] = Fr{vals, (u32)cc, 0}; set_byte(d, vals[0]); ++d; descend = true; }
                }
                if (descend) continue;
                bool done = false;
                for (;;) {  // back up to the next sibling
                    if (d == 0) { done = true; break; }
                    Fr& f = st[d - 1];
                    if (++f.i < f.c) { set_byte(d - 1, f.vals[f.i]); break; }
                    --d;
                }
                if (done) break;
            }
            const u64 nlen = s.varint();
            if (nlen != n) throw Error(CBLX_EFORMAT, "index: trie length field does not match its contents");
            if (n > 0xFFFFFFF0ull) throw Error(CBLX_EFORMAT, "index: bucket too long");
            kind[r] = KIND_TRIE;
        } else {
            throw Error(CBLX_EFORMAT, "index: bad TrieOrVec tag");
        }
        cnt[r] = (u32)n;
        total += n;
    }
    if (s.p != s.end) throw Error(CBLX_EFORMAT, "index: trailing bytes");  // reject_trailing_bytes
    if (total >= 0xFFFFFFF0ull) throw Error(CBLX_ERANGE, "index: more than 2^32 - 16 words (per-GPU limit of this build)");
    Buf<u64> a_lo = lo.finish(), a_hi;
    if (WS) a_hi = hi->finish();
    install_index(c, prefix, cnt, kind, std::move(a_lo), std::move(a_hi));
}

// `self |= other` on host copies (v1): src/wordset/set_ops.rs:123-157 + src/trievec/set_ops.rs:43-71
// returns true when `b` changed: the reference's |= walks other's bucket with iter_sorted, which sorts a Vec in place
bool merge_host(const Consts& P, const HostIndex& a, HostIndex& b, HostIndex& o) {
    bool b_changed = false;
    const bool wide = P.wide_suffix();
    auto get = [&](const HostIndex& h, u64 i) -> u128 { return wide ? (((u128)h.hi[i] << 64) | h.lo[i]) : (u128)h.lo[i]; };
    auto put = [&](u128 x) { o.lo.push_back((u64)x); if (wide) o.hi.push_back((u64)(x >> 64)); };
    o.off.assign(1, 0);
    u64 i = 0, j = 0;
    const u64 na = a.prefix.size(), nb = b.prefix.size();
    std::vector<u128> sa, sb;
    while (i < na || j < nb) {
        if (j >= nb || (i < na && a.prefix[i] < b.prefix[j])) {  // self only: untouched
            o.prefix.push_back(a.prefix[i]); o.kind.push_back(a.kind[i]); o.cnt.push_back(a.cnt[i]);
            for (u64 t = a.off[i]; t < a.off[i + 1]; ++t) put(get(a, t));
            ++i;
        } else if (i >= na || b.prefix[j] < a.prefix[i]) {       // other only: cloned as stored
            o.prefix.push_back(b.prefix[j]); o.kind.push_back(b.kind[j]); o.cnt.push_back(b.cnt[j]);
            for (u64 t = b.off[j]; t < b.off[j + 1]; ++t) put(get(b, t));
            ++j;
        } else {                                                  // both: sorted(self) ++ sorted(other \ self) or trie union
            sa.clear(); sb.clear();
            for (u64 t = a.off[i]; t < a.off[i + 1]; ++t) sa.push_back(get(a, t));
            for (u64 t = b.off[j]; t < b.off[j + 1]; ++t) sb.push_back(get(b, t));
            std::sort(sa.begin(), sa.end());
            if (!std::is_sorted(sb.begin(), sb.end())) {
                std::sort(sb.begin(), sb.end());
                for (u64 t = b.off[j], q = 0; t < b.off[j + 1]; ++t, ++q) { b.lo[t] = (u64)sb[q]; if (wide) b.hi[t] = (u64)(sb[q] >> 64); }
                b_changed = true;
            }
            std::vector<u128> ins;
            std::set_difference(sb.begin(), sb.end(), sa.begin(), sa.end(), std::back_inserter(ins));
            u64 n = 0;
            if (a.kind[i] == KIND_VEC) {
                for (u128 x : sa) { put(x); ++n; }
                for (u128 x : ins) { put(x); ++n; }   // pushed at the end; no threshold check (Vec may exceed 1024)
            } else {
                std::vector<u128> u;
                std::merge(sa.begin(), sa.end(), ins.begin(), ins.end(), std::back_inserter(u));
                for (u128 x : u) { put(x); ++n; }
            }
            o.prefix.push_back(a.prefix[i]); o.kind.push_back(a.kind[i]); o.cnt.push_back((u32)n);
            ++i; ++j;
        }
        o.off.push_back(o.lo.size());
    }
    return b_changed;
}

template <typename F> int guard(cblx_ctx* c, F&& f) {
    try {
        if (c) CBLX_HIP(hipSetDevice(c->device));
        f();
        return CBLX_OK;
    } catch (const Error& e) {
        (c ? c->err : g_global_err) = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        (c ? c->err : g_global_err) = "out of host memory";
        return CBLX_ENOMEM;
    } catch (const std::exception& e) {
        (c ? c->err : g_global_err) = e.what();
        return CBLX_EINVAL;
    }
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

uint32_t cblx_abi_version(void) { return CBLX_ABI_VERSION; }
const char* cblx_last_global_error(void) { return g_global_err.c_str(); }
const char* cblx_last_error(const cblx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_global_err.c_str(); }

int cblx_create(const cblx_params* p, cblx_ctx** out) {
    return guard(nullptr, [&] {
        if (!p || !out) throw Error(CBLX_EINVAL, "null argument");
        *out = nullptr;
        if (p->k < 5 || p->k > 59 || (p->k & 1) == 0) throw Error(CBLX_EINVAL, "K must be odd and in [5, 59]");
        Consts P;
        P.K = p->k; P.PB = p->prefix_bits; P.KB = 2 * p->k; P.POS = ilog2_npo2(P.KB); P.WB = P.KB + P.POS;
        P.canonical = p->canonical ? 1 : 0;
        if (P.PB < 1 || P.PB > 32) throw Error(CBLX_EINVAL, "PREFIX_BITS=" + std::to_string(P.PB) + " but it should be in [1, 32]");
        if (P.PB > 28) throw Error(CBLX_EINVAL, "PREFIX_BITS > 28 is not supported (README.md:130 caps it at 28)");
        if (P.WB <= P.PB) throw Error(CBLX_EINVAL, "SUFFIX_BITS should be != 0");
        if (P.WB > 128) throw Error(CBLX_EINVAL, "Cannot fit a K-mer and its length in a 128-bit integer");
        P.SB = P.WB - P.PB;
        P.BYTES = (P.SB + 7) / 8;
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw Error(CBLX_EDEVICE, "no HIP device available (libcblx has no CPU fallback)");
        int dev = p->device;
        if (dev < 0) CBLX_HIP(hipGetDevice(&dev));
        if (dev >= ndev) throw Error(CBLX_EINVAL, "device ordinal out of range");
        CBLX_HIP(hipSetDevice(dev));
        std::unique_ptr<cblx_ctx> c(new cblx_ctx());
        c->P = P;
        c->device = dev;
        c->flags = p->flags;
        CBLX_HIP(hipStreamCreate(&c->stream));
        *out = c.release();
    });
}
void cblx_destroy(cblx_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& e : ctx->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto& e : ctx->ev_free) (void)hipEventDestroy(e);
    ctx->res = Resident();
    ingest_destroy(ctx);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int cblx_insert_seq(cblx_ctx* c, const uint8_t* seq, uint64_t len) {
    return guard(c, [&] {
        if (!seq && len) throw Error(CBLX_EINVAL, "null sequence");
        if (len < c->P.K) throw Error(CBLX_ESHORT, "Sequence size (" + std::to_string(len) + ") is smaller than K (" + std::to_string(c->P.K) + ")");
        ingest_seq(c, seq, len);
    });
}
int cblx_insert_seqs(cblx_ctx* c, const uint8_t* bases, const uint64_t* offsets, uint64_t n) {
    return guard(c, [&] {
        if (n == 0) return;
        if (!bases || !offsets) throw Error(CBLX_EINVAL, "null argument");
        bool mono = true;
        u64 minlen = ~0ull;
        for (u64 i = 0; i < n; ++i) {  // branch-free scan; the offender is looked up only on failure
            mono &= offsets[i + 1] >= offsets[i];
            minlen = std::min(minlen, offsets[i + 1] - offsets[i]);
        }
        if (!mono) throw Error(CBLX_EINVAL, "offsets must be non-decreasing");
        if (minlen < c->P.K) throw Error(CBLX_ESHORT, "Sequence size (" + std::to_string(minlen) + ") is smaller than K (" + std::to_string(c->P.K) + ")");
        ingest_seqs(c, bases, offsets, n);
    });
}
int cblx_insert_seqs_device(cblx_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n) {
    return guard(c, [&] {
        if (n && (!d_bases || !d_offsets)) throw Error(CBLX_EINVAL, "null argument");
        flush(c);  // keep stream order with anything enqueued earlier
        insert_device(c, d_bases, d_offsets, n);
        CBLX_HIP(hipStreamSynchronize(c->stream));
    });
}
int cblx_flush(cblx_ctx* c) { return guard(c, [&] { flush(c); }); }

// The `read_fasta` + `while let Some(record) = reader.next() { cbl.insert_seq(&seqrec.seq()) }` loop of
// examples/cbl.rs:112-115,154-163 (needletail stand-in): FASTA (multi-line, CRLF tolerated) or 4-line FASTQ, plain or
// gzip (zlib is looked up at run time; without it a .gz input is an error). The file is read in 16 MiB blocks and
// scanned line by line with memchr; every line of bases goes straight into the pinned ingest blocks, so parsing,
// PCIe and the GPU insert of the previous batch (every ~1 GiB of bases) overlap.
struct ByteSource {
    int fd = -1;
    void* gz = nullptr;
    void* zlib = nullptr;
    int (*gzread_)(void*, void*, unsigned) = nullptr;
    int (*gzclose_)(void*) = nullptr;
    ~ByteSource() {
        if (gz && gzclose_) gzclose_(gz);
        if (fd >= 0) ::close(fd);
        if (zlib) dlclose(zlib);
    }
    void open(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) throw Error(CBLX_EINVAL, std::string("Failed to open ") + path);
        u8 magic[2] = {0, 0};
        const ssize_t got = ::pread(fd, magic, 2, 0);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            zlib = dlopen("libz.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!zlib) throw Error(CBLX_EFORMAT, std::string(path) + " is gzip-compressed and zlib (libz.so.1) is not available");
            auto gzdopen_ = (void* (*)(int, const char*))dlsym(zlib, "gzdopen");
            auto gzbuffer_ = (int (*)(void*, unsigned))dlsym(zlib, "gzbuffer");
            gzread_ = (int (*)(void*, void*, unsigned))dlsym(zlib, "gzread");
            gzclose_ = (int (*)(void*))dlsym(zlib, "gzclose");
            if (!gzdopen_ || !gzread_ || !gzclose_) throw Error(CBLX_EFORMAT, "zlib: missing gz* symbols");
            gz = gzdopen_(fd, "rb");
            if (!gz) throw Error(CBLX_EINVAL, std::string("Failed to open ") + path);
            fd = -1;  // owned by the gz handle now
            if (gzbuffer_) gzbuffer_(gz, 1u << 20);
        }
    }
    size_t read(u8* dst, size_t cap) {
        if (gz) {
            const int r = gzread_(gz, dst, (unsigned)std::min<size_t>(cap, 1u << 30));
            if (r < 0) throw Error(CBLX_EFORMAT, "gzip: read error");
            return (size_t)r;
        }
        const ssize_t r = ::read(fd, dst, cap);
        if (r < 0) throw Error(CBLX_EINVAL, "read error");
        return (size_t)r;
    }
};
struct LineReader {
    ByteSource& src;
    std::vector<u8> buf;
    size_t beg = 0, end = 0;
    bool eof = false;
    explicit LineReader(ByteSource& s) : src(s), buf(16u << 20) {}
    // next line without its terminator ('\n' or '\r\n'); false at the end of the input
    bool next(const u8*& p, size_t& n) {
        for (;;) {
            if (beg < end) {
                const u8* nl = (const u8*)std::memchr(buf.data() + beg, '\n', end - beg);
                if (nl || eof) {
                    const size_t stop = nl ? (size_t)(nl - buf.data()) : end;
                    p = buf.data() + beg;
                    n = stop - beg;
                    beg = nl ? stop + 1 : end;
                    if (n && p[n - 1] == '\r') --n;
                    return true;
                }
            } else if (eof) {
                return false;
            }
            // no complete line buffered: keep the partial one at the front and read more
            if (beg) { std::memmove(buf.data(), buf.data() + beg, end - beg); end -= beg; beg = 0; }
            if (end == buf.size()) buf.resize(buf.size() * 2);
            const size_t got = src.read(buf.data() + end, buf.size() - end);
            if (got == 0) eof = true;
            end += got;
        }
    }
};
int cblx_insert_fastx_file(cblx_ctx* c, const char* path, uint64_t* n_records) {
    return guard(c, [&] {
        if (n_records) *n_records = 0;
        if (!path) throw Error(CBLX_EINVAL, "null argument");
        ByteSource src;
        src.open(path);
        LineReader lr(src);
        u64 nrec = 0;
        const u8* p;
        size_t n;
        // skip leading blank lines, then the first byte decides the format
        bool have_line = false;
        while ((have_line = lr.next(p, n)) && n == 0) {}
        if (!have_line) return;
        if (p[0] != '>' && p[0] != '@') throw Error(CBLX_EFORMAT, "not a FASTA/FASTQ file (first record does not start with '>' or '@')");
        const u64 flush_at = std::min<u64>(1ull << 30, ingest_flush_bytes());
        try {
        if (p[0] == '>') {
            bool open_rec = true;  // the header line has been consumed
            while (lr.next(p, n)) {
                if (n && p[0] == '>') { ingest_end_seq(c, flush_at); ++nrec; continue; }
                if (n) ingest_bases(c, p, n);
            }
            if (open_rec) { ingest_end_seq(c, flush_at); ++nrec; }
        } else {
            for (;;) {
                if (n == 0) { if (!lr.next(p, n)) break; continue; }  // blank line between records
                if (p[0] != '@') throw Error(CBLX_EFORMAT, "FASTQ: expected '@' header");
                const u8 *sq, *pl, *ql;
                size_t ns, npl, nq;
                if (!lr.next(sq, ns)) throw Error(CBLX_EFORMAT, "FASTQ: truncated record");
                if (ns) ingest_bases(c, sq, ns);  // the buffer may move on the next call: consume the line first
                if (!lr.next(pl, npl)) throw Error(CBLX_EFORMAT, "FASTQ: truncated record");
                if (npl == 0 || pl[0] != '+') throw Error(CBLX_EFORMAT, "FASTQ: expected '+' separator");
                if (!lr.next(ql, nq)) throw Error(CBLX_EFORMAT, "FASTQ: truncated record");
                ingest_end_seq(c, flush_at);
                ++nrec;
                if (!lr.next(p, n)) break;
            }
        }
        } catch (...) { ingest_abort_seq(c); if (n_records) *n_records = nrec; throw; }
        if (n_records) *n_records = nrec;
    });
}

int cblx_insert_words_device(cblx_ctx* c, const uint64_t* d_lo, const void* d_hi, uint64_t n) {
    return guard(c, [&] {
        flush(c);
        if (n == 0) return;
        if (!d_lo || (c->P.has_hi() && !d_hi)) throw Error(CBLX_EINVAL, "null argument");
        dispatch(c->P, [&](auto cfg) {
            typedef decltype(cfg) C;
            Records rec;
            begin_records<C>(c, rec, n);
            rec.ext_lo = d_lo;  // the first partition pass reads the caller's arrays in place
            rec.ext_hi = d_hi;
            pipeline<C>(c, rec, n);
            c->kmers_inserted += n;
        });
        collect_events(c);
        CBLX_HIP(hipStreamSynchronize(c->stream));
    });
}
int cblx_seq_words_device(cblx_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n, uint64_t* d_lo, void* d_hi,
                          uint64_t cap, uint64_t* n_words) {
    return guard(c, [&] {
        if (n_words) *n_words = 0;
        if (n == 0) return;
        if (!d_bases || !d_offsets || !d_lo || (c->P.has_hi() && !d_hi)) throw Error(CBLX_EINVAL, "null argument");
        check_aligned16(d_bases, "d_bases");
        dispatch(c->P, [&](auto cfg) {
            typedef decltype(cfg) C;
            typedef typename C::HiT HiT;
            ChunkPlan pl;
            plan_chunks(c, d_bases, d_offsets, n, pl);
            if (n_words) *n_words = pl.n_kmers;
            if (pl.n_kmers > cap) throw Error(CBLX_ERANGE, "output capacity too small");
            if (pl.n_kmers == 0) return;
            encode<C>(c, d_bases, pl, d_lo, (HiT*)d_hi, 0);
            CBLX_HIP(hipStreamSynchronize(c->stream));
        });
        collect_events(c);
    });
}

// KRN-1 + the exchange partition in one call: words of the sequences, already grouped by destination prefix range
// (stable). The destination histogram is accumulated by KRN-1 itself, so the words are read once (scatter) instead of
// twice (histogram + scatter), and the unpartitioned words never leave the library's workspace.
int cblx_seq_words_partitioned_device(cblx_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n, const uint32_t* bounds,
                                      uint32_t nd, uint64_t* d_out_lo, void* d_out_hi, uint64_t cap, uint64_t* counts, uint64_t* n_words) {
    return guard(c, [&] {
        if (n_words) *n_words = 0;
        if (nd < 1 || nd > MAX_DEST) throw Error(CBLX_EINVAL, "number of destinations must be in [1, 16]");
        if (!counts || (nd > 1 && !bounds)) throw Error(CBLX_EINVAL, "null argument");
        for (u32 d = 0; d < nd; ++d) counts[d] = 0;
        if (n == 0) return;
        if (!d_bases || !d_offsets || !d_out_lo || (c->P.has_hi() && !d_out_hi)) throw Error(CBLX_EINVAL, "null argument");
        check_aligned16(d_bases, "d_bases");
        for (u32 i = 1; i + 1 < nd; ++i) if (bounds[i] < bounds[i - 1]) throw Error(CBLX_EINVAL, "bounds must be ascending");
        dispatch(c->P, [&](auto cfg) {
            typedef decltype(cfg) C;
            typedef typename C::HiT H;
            ChunkPlan pl;
            plan_chunks(c, d_bases, d_offsets, n, pl);
            const u64 nw = pl.n_kmers;
            if (n_words) *n_words = nw;
            if (nw > cap) throw Error(CBLX_ERANGE, "output capacity too small");
            if (nw == 0) return;
            if (nw >= 0xFFFFFFF0ull) throw Error(CBLX_ERANGE, "too many words in one partition call");
            const size_t hs = hi_elem_size(c->P);
            Buf<u64> t_lo(c->pool, nw + 2);
            Buf<u8> t_hi(c->pool, hs ? (nw + 2) * hs : 8);
            const u32 ntiles = (u32)ceil_div(nw, RDX_TILE);
            Buf<u32> cnt(c->pool, (size_t)256 * (ntiles + 2)), colpre(c->pool, (size_t)256 * ntiles), scratch, coltot(c->pool, 256), adj(c->pool, 256);
            CBLX_HIP(hipMemsetAsync(cnt.get(), 0, (size_t)256 * (ntiles + 2) * 4, c->stream));
            DigitDest fn;
            fn.SB = c->P.SB; fn.PB = c->P.PB; fn.nd = nd;
            EncHist eh{};
            eh.counts = cnt.get();
            eh.nd = nd; eh.SB = c->P.SB; eh.PB = c->P.PB;
            for (u32 i = 0; i < MAX_DEST - 1; ++i) { fn.bounds[i] = i + 1 < nd ? bounds[i] : 0xFFFFFFFFu; eh.bounds[i] = fn.bounds[i]; }
            encode<C>(c, d_bases, pl, t_lo.get(), (H*)t_hi.get(), 0, eh);
            const TileView tv{nullptr, nullptr, nullptr, nullptr, ntiles, nw};
            { StageTimer t(c, ST_SCAN);
              colscan(c, cnt.get(), nullptr, ntiles, colpre.get(), coltot.get(), scratch);
              hipLaunchKernelGGL(k_seg_adjust, dim3(1), dim3(256), 0, c->stream, colpre.get(), coltot.get(), (const u32*)nullptr, (const u32*)nullptr,
                                 (const u32*)nullptr, ntiles, 1u, adj.get()); }
            { StageTimer t(c, ST_SCATTER);
              hipLaunchKernelGGL((k_radix_scatter<H, H, DigitDest>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, c->stream, t_lo.get(), (const H*)t_hi.get(), tv, fn,
                                 colpre.get(), adj.get(), d_out_lo, (H*)d_out_hi); }
            CBLX_HIP(hipGetLastError());
            std::vector<u32> tot = d2h_vec<u32>(c, coltot.get(), 256);
            for (u32 d = 0; d < nd; ++d) counts[d] = tot[d];
        });
        collect_events(c);
    });
}

// Stable partition of words by destination prefix range (multi-GPU exchange step).
int cblx_partition_words_device(cblx_ctx* c, const uint64_t* d_lo, const void* d_hi, uint64_t n, const uint32_t* bounds, uint32_t nd,
                                uint64_t* d_out_lo, void* d_out_hi, uint64_t* counts) {
    return guard(c, [&] {
        if (nd < 1 || nd > MAX_DEST) throw Error(CBLX_EINVAL, "number of destinations must be in [1, 16]");
        if (!counts || (nd > 1 && !bounds)) throw Error(CBLX_EINVAL, "null argument");
        for (u32 d = 0; d < nd; ++d) counts[d] = 0;
        if (n == 0) return;
        if (!d_lo || !d_out_lo || (c->P.has_hi() && (!d_hi || !d_out_hi))) throw Error(CBLX_EINVAL, "null argument");
        if (n >= 0xFFFFFFF0ull) throw Error(CBLX_ERANGE, "too many words in one partition call");
        DigitDest fn;
        fn.SB = c->P.SB; fn.PB = c->P.PB; fn.nd = nd;
        for (u32 i = 0; i < MAX_DEST - 1; ++i) fn.bounds[i] = i + 1 < nd ? bounds[i] : 0xFFFFFFFFu;
        for (u32 i = 1; i + 1 < nd; ++i) if (bounds[i] < bounds[i - 1]) throw Error(CBLX_EINVAL, "bounds must be ascending");
        const u32 ntiles = (u32)ceil_div(n, RDX_TILE);
        Buf<u32> cnt(c->pool, (size_t)256 * ntiles), colpre(c->pool, (size_t)256 * ntiles), scratch, coltot(c->pool, 256), adj(c->pool, 256);
        const TileView tv{nullptr, nullptr, nullptr, nullptr, ntiles, n};
        dispatch(c->P, [&](auto cfg) {
            typedef typename decltype(cfg)::HiT H;
            const H* hi = (const H*)d_hi;
            H* ohi = (H*)d_out_hi;
            { StageTimer t(c, ST_HIST);
              hipLaunchKernelGGL((k_radix_hist<H, DigitDest>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, c->stream, d_lo, hi, tv, fn, cnt.get()); }
            { StageTimer t(c, ST_SCAN);
              colscan(c, cnt.get(), nullptr, ntiles, colpre.get(), coltot.get(), scratch);
              hipLaunchKernelGGL(k_seg_adjust, dim3(1), dim3(256), 0, c->stream, colpre.get(), coltot.get(), (const u32*)nullptr, (const u32*)nullptr,
                                 (const u32*)nullptr, ntiles, 1u, adj.get()); }
            { StageTimer t(c, ST_SCATTER);
              hipLaunchKernelGGL((k_radix_scatter<H, H, DigitDest>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, c->stream, d_lo, hi, tv, fn, colpre.get(),
                                 adj.get(), d_out_lo, ohi); }
            CBLX_HIP(hipGetLastError());
        });
        // run length of every destination = its column total
        std::vector<u64> starts(nd + 1, n);
        {
            std::vector<u32> tot = d2h_vec<u32>(c, coltot.get(), 256);
            u64 run = 0;
            for (u32 d = 0; d < nd; ++d) { starts[d] = run; run += tot[d]; }
        }
        for (u32 d = 0; d < nd; ++d) counts[d] = starts[d + 1] - starts[d];
        collect_events(c);
    });
}

int cblx_count(cblx_ctx* c, uint64_t* out) { return guard(c, [&] { flush(c); *out = c->res.count; }); }
int cblx_num_buckets(cblx_ctx* c, uint64_t* out) { return guard(c, [&] { flush(c); *out = c->res.nb; }); }
int cblx_is_empty(cblx_ctx* c, int* out) { return guard(c, [&] { flush(c); *out = c->res.nb == 0; }); }
int cblx_is_canonical(const cblx_ctx* c, int* out) { if (!c || !out) return CBLX_EINVAL; *out = (int)c->P.canonical; return CBLX_OK; }

int cblx_serialized_size(cblx_ctx* c, uint64_t* nbytes) {
    return guard(c, [&] {
        flush(c);
        DevBlob blob;
        if (serialize_device(c, false, blob)) { *nbytes = blob.n; return; }
        HostIndex h;
        download(c, h);
        Sink s(nullptr, 0);
        serialize_host(c->P, h, s);
        *nbytes = s.pos;
    });
}
int cblx_serialize(cblx_ctx* c, uint8_t* buf, uint64_t cap, uint64_t* written) {
    return guard(c, [&] {
        flush(c);
        DevBlob blob;
        if (serialize_device(c, false, blob)) {
            if (written) *written = blob.n;
            if (blob.n > cap) throw Error(CBLX_ERANGE, "buffer too small: need " + std::to_string(blob.n) + " bytes");
            serialize_device(c, true, blob);
            xfer(c).d2h_copy(buf, blob.bytes.get(), blob.n);
            return;
        }
        HostIndex h;
        download(c, h);
        Sink s(buf, cap);
        serialize_host(c->P, h, s);
        if (written) *written = s.pos;
        if (s.pos > cap) throw Error(CBLX_ERANGE, "buffer too small: need " + std::to_string(s.pos) + " bytes");
    });
}
int cblx_save_to_file(cblx_ctx* c, const char* path) {
    return guard(c, [&] {
        flush(c);
        DevBlob blob;
        if (serialize_device(c, true, blob)) {  // the lanes write their chunks straight from pinned memory
            const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (fd < 0) throw Error(CBLX_EINVAL, std::string("Failed to create ") + path);
            std::atomic<bool> bad{false};
            try {
                xfer(c).d2h(blob.bytes.get(), blob.n, [&](const u8* src, size_t off, size_t n) {
                    while (n) {
                        const ssize_t w = ::pwrite(fd, src, n, (off_t)off);
                        if (w <= 0) { bad = true; return; }
                        src += w; off += (size_t)w; n -= (size_t)w;
                    }
                });
            } catch (...) { ::close(fd); throw; }
            if (::close(fd) != 0 || bad) throw Error(CBLX_EINVAL, std::string("Failed to write index to ") + path);
            return;
        }
        HostIndex h;
        download(c, h);
        Sink cnt(nullptr, 0);
        serialize_host(c->P, h, cnt);
        std::vector<u8> out(cnt.pos);
        Sink s(out.data(), out.size());
        serialize_host(c->P, h, s);
        std::ofstream f(path, std::ios::binary);
        if (!f) throw Error(CBLX_EINVAL, std::string("Failed to create ") + path);
        f.write((const char*)out.data(), (std::streamsize)out.size());
        if (!f) throw Error(CBLX_EINVAL, std::string("Failed to write index to ") + path);
    });
}
int cblx_load(cblx_ctx* c, const uint8_t* data, uint64_t len) {
    return guard(c, [&] {
        if (!data) throw Error(CBLX_EINVAL, "null argument");
        bool canon = false;
        CBLX_HIP(hipStreamSynchronize(c->stream));
        ingest_drop(c);
        c->res = Resident();  // the old index is gone even if the bytes turn out to be malformed (it is being replaced)
        if (c->P.wide_suffix()) load_stream<true>(c, data, len, canon);
        else load_stream<false>(c, data, len, canon);
        c->P.canonical = canon ? 1 : 0;
    });
}
int cblx_load_from_file(cblx_ctx* c, const char* path) {
    if (!c) return CBLX_EINVAL;
    if (!path) { c->err = "null argument"; return CBLX_EINVAL; }
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { c->err = std::string("Failed to open ") + path; return CBLX_EINVAL; }
    struct stat st;
    if (::fstat(fd, &st) != 0) { ::close(fd); c->err = std::string("Failed to stat ") + path; return CBLX_EINVAL; }
    const size_t len = (size_t)st.st_size;
    if (len == 0) { ::close(fd); c->err = "index: unexpected end of data"; return CBLX_EFORMAT; }
    void* m = ::mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m == MAP_FAILED) { c->err = std::string("Failed to map ") + path; return CBLX_EINVAL; }
    (void)::madvise(m, len, MADV_SEQUENTIAL);
    const int rc = cblx_load(c, (const u8*)m, len);
    ::munmap(m, len);
    return rc;
}
int cblx_merge_assign(cblx_ctx* self, cblx_ctx* other) {
    return guard(self, [&] {
        if (!other) throw Error(CBLX_EINVAL, "null argument");
        if (self->P.K != other->P.K || self->P.PB != other->P.PB) throw Error(CBLX_EINVAL, "merge: K / PREFIX_BITS mismatch");
        if (self->P.canonical != other->P.canonical) throw Error(CBLX_EINVAL, "One of the index is canonical while the other isn't");
        if (self == other) return;  // x |= x
        flush(self);
        CBLX_HIP(hipSetDevice(other->device));
        flush(other);
        CBLX_HIP(hipSetDevice(self->device));
        if (other->res.count == 0) return;
        bool on_device = false;
        if (self->device == other->device && self->res.count != 0 && self->res.count + other->res.count < 0xFFFFFFF0ull) {
            CBLX_HIP(hipStreamSynchronize(other->stream));
            dispatch(self->P, [&](auto cfg) { merge_direct<decltype(cfg)>(self, other->res); });
            on_device = true;
            collect_events(self);
        }
        if (!on_device) {  // indexes on different devices (or self empty): through the host
            HostIndex a, b, o;
            CBLX_HIP(hipSetDevice(other->device));
            download(other, b);
            CBLX_HIP(hipSetDevice(self->device));
            download(self, a);
            const bool other_changed = merge_host(self->P, a, b, o);
            upload(self, o);
            if (other_changed) {
                CBLX_HIP(hipSetDevice(other->device));
                upload(other, b);
                CBLX_HIP(hipSetDevice(self->device));
            }
        }
    });
}
int cblx_export_buckets(cblx_ctx* c, cblx_bucket_cb cb, void* user) {
    return guard(c, [&] {
        if (!cb) throw Error(CBLX_EINVAL, "null callback");
        flush(c);
        HostIndex h;
        download(c, h);
        for (u64 r = 0; r < h.prefix.size(); ++r) {
            const u64 a = h.off[r];
            if (cb(user, h.prefix[r], h.kind[r], h.cnt[r], h.lo.data() + a, h.hi.empty() ? nullptr : h.hi.data() + a)) break;
        }
    });
}
int cblx_contains_seq(cblx_ctx* c, const uint8_t* seq, uint64_t len, uint8_t* out, uint64_t cap, uint64_t* n) {
    return guard(c, [&] {
        flush(c);
        if (len < c->P.K) throw Error(CBLX_ESHORT, "Sequence size (" + std::to_string(len) + ") is smaller than K (" + std::to_string(c->P.K) + ")");
        dispatch(c->P, [&](auto cfg) {
            typedef decltype(cfg) C;
            typedef typename C::HiT HiT;
            Buf<u8> d_b(c->pool, len + 64);
            Buf<u64> d_o(c->pool, 2);
            u64 offs[2] = {0, len};
            xfer(c).h2d_copy(d_b.get(), seq, len);  // pinned lanes: the caller's buffer is pageable
            xfer(c).sync();
            h2d(c, d_o.get(), offs, 2);
            ChunkPlan pl;
            const u8* pb = d_b.get();
            plan_chunks(c, pb, d_o.get(), 1, pl);
            if (n) *n = pl.n_kmers;
            if (pl.n_kmers > cap) throw Error(CBLX_ERANGE, "output capacity too small");
            Buf<u64> w_lo(c->pool, pl.n_kmers + 2);
            Buf<u8> w_hi(c->pool, (pl.n_kmers + 2) * std::max<size_t>(1, hi_elem_size(c->P)));
            Buf<u8> d_out(c->pool, pl.n_kmers + 8);
            encode<C>(c, pb, pl, w_lo.get(), (HiT*)w_hi.get(), 0);
            hipLaunchKernelGGL(k_contains<HiT>, grid1(pl.n_kmers, 256), dim3(256), 0, c->stream, w_lo.get(), (const HiT*)w_hi.get(), pl.n_kmers, c->P.SB,
                               c->P.PB, c->res.view(), c->res.a_lo.get(), c->P.wide_suffix() ? c->res.a_hi.get() : (const u64*)nullptr, d_out.get());
            CBLX_HIP(hipGetLastError());
            CBLX_HIP(hipStreamSynchronize(c->stream));
            xfer(c).d2h_copy(out, d_out.get(), pl.n_kmers);
        });
        collect_events(c);
    });
}
int cblx_checksum(cblx_ctx* c, uint64_t* sum) {
    return guard(c, [&] {
        flush(c);
        *sum = 0;
        const Resident& r = c->res;
        if (r.count == 0) return;
        Buf<u64> res_off(c->pool, r.nb + 1), out(c->pool, 1);
        u64 tot = exclusive_scan<u64>(c, r.cnt.get(), r.nb, res_off.get());
        hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, res_off.get() + r.nb, tot);
        CBLX_HIP(hipMemsetAsync(out.get(), 0, 8, c->stream));
        hipLaunchKernelGGL(k_checksum_index, dim3(256 * 16), dim3(256), 0, c->stream, tot, r.nb, res_off.get(), r.prefix.get(), r.start.get(), r.a_lo.get(),
                           c->P.wide_suffix() ? r.a_hi.get() : (const u64*)nullptr, c->P.SB, out.get());
        CBLX_HIP(hipGetLastError());
        *sum = d2h<u64>(c, out.get());
    });
}
int cblx_checksum_words_device(cblx_ctx* c, const uint64_t* d_lo, const void* d_hi, uint64_t n, uint64_t* sum) {
    return guard(c, [&] {
        *sum = 0;
        if (n == 0) return;
        Buf<u64> out(c->pool, 1);
        CBLX_HIP(hipMemsetAsync(out.get(), 0, 8, c->stream));
        dispatch(c->P, [&](auto cfg) {
            typedef typename decltype(cfg)::HiT H;
            hipLaunchKernelGGL(k_checksum_words<H>, dim3(256 * 16), dim3(256), 0, c->stream, d_lo, (const H*)d_hi, n, out.get());
        });
        CBLX_HIP(hipGetLastError());
        *sum = d2h<u64>(c, out.get());
    });
}
int cblx_validate(cblx_ctx* c, int strict, uint64_t* violations) {
    return guard(c, [&] {
        flush(c);
        *violations = 0;
        const Resident& r = c->res;
        if (r.nb == 0) return;
        Buf<u64> bad(c->pool, 1);
        CBLX_HIP(hipMemsetAsync(bad.get(), 0, 8, c->stream));
        hipLaunchKernelGGL(k_validate, grid1(r.nb * 64, 256), dim3(256), 0, c->stream, r.nb, r.start.get(), r.cnt.get(), r.kind.get(), r.a_lo.get(),
                           c->P.wide_suffix() ? r.a_hi.get() : (const u64*)nullptr, c->P.SB, (u32)(strict != 0), bad.get());
        CBLX_HIP(hipGetLastError());
        *violations = d2h<u64>(c, bad.get());
    });
}
int cblx_get_consts(const cblx_ctx* c, cblx_consts* o) {
    if (!c || !o) return CBLX_EINVAL;
    o->kmer_bits = c->P.KB; o->pos_bits = c->P.POS; o->word_bits = c->P.WB; o->suffix_bits = c->P.SB; o->bytes = c->P.BYTES;
    o->chunk_size = CHUNK_KMERS; o->threshold = VEC_THRESHOLD; o->hi_bytes = (uint32_t)hi_elem_size(c->P);
    return CBLX_OK;
}
int cblx_stage_times(cblx_ctx* c, const char** names, double* ms, uint64_t* launches, uint32_t cap, uint32_t* n) {
    return guard(c, [&] {
        collect_events(c);
        u32 k = 0;
        for (; k < ST_N && k < cap; ++k) {
            if (names) names[k] = c->stages[k].name;
            if (ms) ms[k] = c->stages[k].ms;
            if (launches) launches[k] = c->stages[k].launches;
        }
        if (n) *n = k;
    });
}
int cblx_stage_times_reset(cblx_ctx* c) {
    return guard(c, [&] { collect_events(c); for (auto& s : c->stages) { s.ms = 0; s.launches = 0; } });
}
int cblx_kmers_inserted(cblx_ctx* c, uint64_t* out) { if (!c || !out) return CBLX_EINVAL; *out = c->kmers_inserted; return CBLX_OK; }
int cblx_trim(cblx_ctx* c) {
    return guard(c, [&] {
        if (c->ing.nseq == 0) { ingest_wait(c); c->ing.d_bases.reset(); c->ing.d_off.reset(); }
        c->pool.trim();
    });
}
int cblx_clear(cblx_ctx* c) {
    return guard(c, [&] {
        CBLX_HIP(hipStreamSynchronize(c->stream));
        c->res = Resident();
        ingest_drop(c);
    });
}

}  // extern "C"
