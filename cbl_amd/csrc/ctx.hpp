// ctx.hpp — the context behind the opaque cblx_ctx handle: device memory pool, resident index (WordSet state in HBM),
// ingest queue, stage timers, small device<->host helpers, the word-layout dispatch. Included by cblx.cpp only.
#pragma once
#include "../../include/cblx.h"

#include <algorithm>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels_bucket.hpp"
#include "xfer.hpp"

using namespace cblx;

namespace {

thread_local std::string g_global_err;

inline u32 ilog2_npo2(u32 v) { u32 l = 0; while ((1u << l) < v) ++l; return l; }
inline u64 ceil_div(u64 a, u64 b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------------------------------------
// cached device allocations (hipMalloc is kept out of the hot path between flushes). One pool per ctx; when the device
// runs out of memory every pool of the process gives its idle blocks back before the allocation is retried (an index
// that was built earlier keeps tens of GB of idle workspace cached).
struct Pool {
    struct Blk { void* p; size_t sz; bool used; };
    std::vector<Blk> blks;
    std::mutex mu;  // a ctx is single-owner, but another ctx's thread may trim this pool under memory pressure
    static std::mutex& reg_mu() { static std::mutex m; return m; }
    static std::vector<Pool*>& registry() { static std::vector<Pool*> r; return r; }
    Pool() { std::lock_guard<std::mutex> g(reg_mu()); registry().push_back(this); }
    Pool(const Pool&) = delete;
    Pool& operator=(const Pool&) = delete;
    void* alloc(size_t sz) {
        if (sz == 0) sz = 256;
        sz = (sz + 255) & ~(size_t)255;
        {
            std::lock_guard<std::mutex> g(mu);
            int best = -1;
            for (size_t i = 0; i < blks.size(); ++i)
                if (!blks[i].used && blks[i].sz >= sz && blks[i].sz <= sz + sz / 2 + 4096 && (best < 0 || blks[i].sz < blks[best].sz)) best = (int)i;
            if (best >= 0) { blks[best].used = true; return blks[best].p; }
        }
        void* p = nullptr;
        if (sz > (64ull << 30) && std::getenv("CBLX_TRACE_SHARDED")) fprintf(stderr, "[cblx pool] allocation of %zu bytes\n", sz);  // trace only: legitimate on a 288 GB part
        hipError_t e = hipMalloc(&p, sz);
        if (e != hipSuccess) {
            (void)hipGetLastError();  // the failed call must not surface at the next hipGetLastError() after a launch
            trim_all();
            e = hipMalloc(&p, sz);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                throw Error(CBLX_ENOMEM, "hipMalloc(" + std::to_string(sz) + ") failed: " + hipGetErrorString(e));
            }
        }
        std::lock_guard<std::mutex> g(mu);
        blks.push_back({p, sz, true});
        return p;
    }
    void release(void* p) {
        if (!p) return;
        std::lock_guard<std::mutex> g(mu);
        for (auto& b : blks) if (b.p == p) { b.used = false; return; }
    }
    void trim() {
        std::lock_guard<std::mutex> g(mu);
        std::vector<Blk> keep;
        for (auto& b : blks) { if (b.used) keep.push_back(b); else (void)hipFree(b.p); }
        blks.swap(keep);
    }
    static void trim_all() {
        std::lock_guard<std::mutex> g(reg_mu());
        for (Pool* p : registry()) p->trim();
    }
    ~Pool() {
        { std::lock_guard<std::mutex> g(reg_mu()); auto& r = registry(); r.erase(std::remove(r.begin(), r.end(), this), r.end()); }
        for (auto& b : blks) (void)hipFree(b.p);
    }
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

struct Stage { const char* name; double ms = 0; u64 launches = 0; u64 units = 0; /* words the stage's kernels were given, where the pipeline counts them (`|=`: per bucket class) */ };
enum { ST_CHUNKS, ST_ENCODE, ST_HIST, ST_SCAN, ST_SCATTER, ST_DIR, ST_BSMALL, ST_BMED, ST_BHUGE, ST_EXPAND, ST_BBIG, ST_N };
const char* kStageNames[ST_N] = {"chunks", "encode", "radix_hist", "radix_scan", "radix_scatter", "directory",
                                 "bucket_small", "bucket_medium", "bucket_huge", "merge_gather", "bucket_big"};

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
    // a big host batch that crosses PCIe as bit planes (ingest_seqs_planes): device planes and the pinned staging they are packed into
    Buf<u32> d_codes;
    Buf<u16> d_valid;
    u8* pin = nullptr;
    size_t pin_cap = 0;
    u64 nbytes = 0, nseq = 0;
    u64 last_end = 0;  // nbytes at the end of the last complete sequence
    Writer wb, wo;
    hipStream_t s = nullptr;
    std::unique_ptr<Xfer> xfer;
    // consumer of the queue: the insert pipeline, or (cblx_query_fastx_file) the membership query with these tallies, or
    // (cblx_stage_fastx_blocks) the caller, who borrows the staged buffers until cblx_stage_release
    // one big batch from pinned host memory, on the wire in slices that land front to back (ingest_seqs): flush() then runs
    // KRN-1 and the first partition pass of slice k while the later slices are still crossing PCIe
    struct Streamed {
        std::vector<u64> seq_cuts;                        // sequence index where every slice starts (+ the end)
        std::vector<std::vector<hipEvent_t>> ready;       // per slice: one event per transfer lane
        std::vector<hipEvent_t> offsets_ready;            // the offsets array has landed
        bool active() const { return !seq_cuts.empty(); }
        void clear() {
            for (auto& v : ready) for (hipEvent_t e : v) (void)hipEventDestroy(e);
            for (hipEvent_t e : offsets_ready) (void)hipEventDestroy(e);
            ready.clear(); offsets_ready.clear(); seq_cuts.clear();
        }
    } streamed;
    bool staged = false;
    bool query = false;
    u64 q_total = 0, q_positive = 0;
};

// a fully partitioned batch of words waiting to be exported (multi-GPU build, sender side)
struct SortedBatch {
    u64 n = 0, nb = 0;   // words, non-empty prefixes
    Buf<u32> prefix;     // [nb]
    Buf<u64> start;      // [nb + 1] first record of every prefix
    Buf<u64> lo;         // sorted records
    Buf<u8> hi;          // their hi parts (raw bytes, element size = hi_elem_size; empty when dropped)
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
    SortedBatch batch;
    std::string err;
    u64 kmers_inserted = 0;
    u64 fine_builds = 0;       // batches that took the FINE-bins build (PREFIX_BITS > 24 on an empty index: comm.hpp insert_device_fine)
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

// 1-D grid for n work items. A HIP launch addresses at most 2^32 - 1 work items in x: a larger grid does not fail, it
// wraps (measured: 9.6 G threads ran as 9.6 G mod 2^32), so it is refused here and big launches are cut by the caller.
inline dim3 grid1(u64 n, u32 threads) {
    const u64 blocks = std::max<u64>(1, ceil_div(n, threads));
    if (blocks * threads >= (1ull << 32)) throw Error(CBLX_ERANGE, "launch of " + std::to_string(n) + " work items exceeds the 2^32 limit of one grid");
    return dim3((unsigned)blocks);
}

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

// A large copy inside one device (the clone of an index: `|=` into an empty one, src/trievec/set_ops.rs:43-71 clones every bucket as stored):
// 16 bytes per lane, eight loads in flight per thread before the first store. hipMemcpyAsync's blit moved the 6 GB arena of cfg 5's operand
// at 4.3 TB/s of read + written bytes; small or odd-sized buffers still take it.
__global__ __launch_bounds__(512) void k_copy16(const uint4* __restrict__ a, uint4* __restrict__ b, u64 n) {
    const u64 base = (u64)blockIdx.x * (512 * 8);
    uint4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const u64 i = base + (u64)j * 512 + threadIdx.x; v[j] = i < n ? a[i] : uint4{0, 0, 0, 0}; }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const u64 i = base + (u64)j * 512 + threadIdx.x; if (i < n) b[i] = v[j]; }
}
inline void device_copy(hipStream_t st, void* dst, const void* src, size_t bytes) {
    const size_t n16 = bytes / 16;
    if (bytes < (1u << 20) || (((uintptr_t)dst | (uintptr_t)src) & 15u) || ceil_div((u64)n16, 512 * 8) >= (1ull << 31)) {
        if (bytes) CBLX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
        return;
    }
    hipLaunchKernelGGL(k_copy16, dim3((unsigned)ceil_div((u64)n16, 512 * 8)), dim3(512), 0, st, (const uint4*)src, (uint4*)dst, (u64)n16);
    CBLX_HIP(hipGetLastError());
    if (bytes & 15u) CBLX_HIP(hipMemcpyAsync((char*)dst + n16 * 16, (const char*)src + n16 * 16, bytes & 15u, hipMemcpyDeviceToDevice, st));
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
    else if (!P.wide_kmer() && !P.wide_suffix()) f(Cfg<false, u8, false>());  // 65..72-bit word, hi bits inside the first pass's digit
    else if (!P.wide_suffix()) f(Cfg<true, u64, false>());
    else f(Cfg<true, u64, true>());
}
inline size_t hi_elem_size(const Consts& P) { return !P.has_hi() ? 0 : ((!P.wide_kmer() && !P.wide_suffix()) ? 1 : 8); }


}  // namespace
