// cblx.cpp — host side of libcblx: the C ABI of include/cblx.h over the HIP kernels (gfx950).
//
// Mirrors the reference's `CBL<K, T, PREFIX_BITS>` surface for the bulk-insert path
// (/root/reference/src/cbl.rs:71-79,127-177,328-339,433-449) with the WordSet state
// (/root/reference/src/wordset/mod.rs:18-26: prefix bitvector + rank->bucket directory + suffix containers) held in
// HBM: bitvector words, popcount-scan rank directory, bucket table indexed by rank, one suffix arena.
// There is no CPU fallback: every data-path step below is a kernel launch.
#include "comm.hpp"

namespace {

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

// CBL::new's asserts (src/cbl.rs:87-91, src/wordset/mod.rs:37-41) and the derived constants (src/cbl.rs:16-32,65-67)
Consts make_consts(const cblx_params& p) {
    if (p.k < 5 || p.k > 59 || (p.k & 1) == 0) throw Error(CBLX_EINVAL, "K must be odd and in [5, 59]");
    Consts P;
    P.K = p.k; P.PB = p.prefix_bits; P.KB = 2 * p.k; P.POS = ilog2_npo2(P.KB); P.WB = P.KB + P.POS;
    P.canonical = p.canonical ? 1 : 0;
    if (P.PB < 1 || P.PB > 32) throw Error(CBLX_EINVAL, "PREFIX_BITS=" + std::to_string(P.PB) + " but it should be in [1, 32]");
    if (P.PB > 28) throw Error(CBLX_EINVAL, "PREFIX_BITS > 28 is not supported (README.md:130 caps it at 28)");
    if (P.WB <= P.PB) throw Error(CBLX_EINVAL, "SUFFIX_BITS should be != 0");
    if (P.WB > 128) throw Error(CBLX_EINVAL, "Cannot fit a K-mer and its length in a 128-bit integer");
    P.SB = P.WB - P.PB;
    P.BYTES = (P.SB + 7) / 8;
    return P;
}

// words of the k-mers of one host sequence, on the device (KRN-1 over a one-sequence batch)
template <typename C> u64 seq_words_of_host_seq(cblx_ctx* c, const uint8_t* seq, uint64_t len, Buf<u64>& w_lo, Buf<u8>& w_hi) {
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
    w_lo = Buf<u64>(c->pool, pl.n_kmers + 2);
    w_hi = Buf<u8>(c->pool, (pl.n_kmers + 2) * std::max<size_t>(1, hi_elem_size(c->P)));
    encode<C>(c, pb, pl, w_lo.get(), (HiT*)w_hi.get(), 0);
    CBLX_HIP(hipStreamSynchronize(c->stream));  // d_b / d_o go back to the pool on return
    return pl.n_kmers;
}
// words (get_word, src/cbl.rs:199-206) of n packed host k-mers, on the device
template <typename C> void words_of_host_kmers(cblx_ctx* c, const uint64_t* lo, const uint64_t* hi, u64 n, Buf<u64>& w_lo, Buf<u8>& w_hi) {
    typedef typename C::HiT HiT;
    if (!lo || (c->P.wide_kmer() && !hi)) throw Error(CBLX_EINVAL, c->P.wide_kmer() ? "null argument (K >= 33 needs the hi halves of the k-mers)" : "null argument");
    if (n >= 0xFFFFFFF0ull) throw Error(CBLX_ERANGE, "too many k-mers in one call");
    Buf<u64> k_lo(c->pool, n + 1), k_hi(c->pool, hi ? n + 1 : 1);
    Buf<u32> bad(c->pool, 1);
    xfer(c).h2d_copy(k_lo.get(), lo, n * 8);
    if (hi) xfer(c).h2d_copy(k_hi.get(), hi, n * 8);
    xfer(c).sync();
    CBLX_HIP(hipMemsetAsync(bad.get(), 0, 4, c->stream));
    w_lo = Buf<u64>(c->pool, n + 2);
    w_hi = Buf<u8>(c->pool, (n + 2) * std::max<size_t>(1, hi_elem_size(c->P)));
    hipLaunchKernelGGL((k_kmers_to_words<C::WIDE, HiT>), grid1(n, 256), dim3(256), 0, c->stream, k_lo.get(), hi ? k_hi.get() : (const u64*)nullptr, n, c->P,
                       w_lo.get(), (HiT*)w_hi.get(), bad.get());
    CBLX_HIP(hipGetLastError());
    if (d2h<u32>(c, bad.get())) throw Error(CBLX_EINVAL, "k-mer has bits set above 2K (not an IntKmer<K>)");
}

// which records of a file a reader keeps: all (block = 0), or the ones block-cyclic dealing gives `rank` of `world` — record i
// (file order, 0-based) belongs to rank (i / block) % world. count_only: keep none, just count.
struct RecordFilter {
    u64 block = 0;
    u32 rank = 0, world = 1;
    bool count_only = false;
    bool mine(u64 i) const { return !count_only && (block == 0 || (i / block) % world == rank); }
};
void read_fastx_into_queue(cblx_ctx* c, const char* path, uint64_t* n_records, const RecordFilter* filt = nullptr) {
    {
        if (n_records) *n_records = 0;
        if (!path) throw Error(CBLX_EINVAL, "null argument");
        if (!filt) {   // large plain files: parallel readers; anything they do not take is read sequentially below
            u64 npar = 0;
            if (fastx_parallel_planes(c, path, &npar)) { if (n_records) *n_records = npar; return; }  // (comm.hpp) bit planes, the insert behind the parse
            if (fastx_parallel(c, path, &npar)) { if (n_records) *n_records = npar; return; }
        }
        auto mine = [&](u64 i) { return !filt || filt->mine(i); };
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
        const u64 flush_at = filt ? ~0ull : std::min<u64>(1ull << 30, ingest_flush_bytes());  // staged records are never inserted here
        try {
        if (p[0] == '>') {
            bool open_rec = true;  // the header line has been consumed
            while (lr.next(p, n)) {
                if (n && p[0] == '>') { if (mine(nrec)) ingest_end_seq(c, flush_at); ++nrec; continue; }
                if (n && mine(nrec)) ingest_bases(c, p, n);
            }
            if (open_rec) { if (mine(nrec)) ingest_end_seq(c, flush_at); ++nrec; }
        } else {
            for (;;) {
                if (n == 0) { if (!lr.next(p, n)) break; continue; }  // blank line between records
                if (p[0] != '@') throw Error(CBLX_EFORMAT, "FASTQ: expected '@' header");
                const u8 *sq, *pl, *ql;
                size_t ns, npl, nq;
                if (!lr.next(sq, ns)) throw Error(CBLX_EFORMAT, "FASTQ: truncated record");
                if (ns && mine(nrec)) ingest_bases(c, sq, ns);  // the buffer may move on the next call: consume the line first
                if (!lr.next(pl, npl)) throw Error(CBLX_EFORMAT, "FASTQ: truncated record");
                if (npl == 0 || pl[0] != '+') throw Error(CBLX_EFORMAT, "FASTQ: expected '+' separator");
                if (!lr.next(ql, nq)) throw Error(CBLX_EFORMAT, "FASTQ: truncated record");
                if (mine(nrec)) ingest_end_seq(c, flush_at);
                ++nrec;
                if (!lr.next(p, n)) break;
            }
        }
        } catch (...) { ingest_abort_seq(c); if (n_records) *n_records = nrec; throw; }
        if (n_records) *n_records = nrec;
    }
}

}  // namespace

// ================================================================================================ C ABI
namespace {
// a deep copy of `src`'s resident index into dst's pool (same device: a copy kernel; another device: over the fabric)
Resident clone_resident(cblx_ctx* dst, const cblx_ctx* src) {
    const Resident& o = src->res;
    Resident copy;
    copy.nb = o.nb;
    copy.count = o.count;
    auto dup = [&](auto& d, const auto& s) {
        typedef typename std::remove_reference<decltype(*s.get())>::type T;
        if (!s.get()) return;
        d = Buf<T>(dst->pool, s.n);
        if (dst->device == src->device) device_copy(dst->stream, d.get(), s.get(), s.n * sizeof(T));
        else CBLX_HIP(hipMemcpyPeerAsync(d.get(), dst->device, s.get(), src->device, s.n * sizeof(T), dst->stream));
    };
    dup(copy.bv, o.bv); dup(copy.rank_dir, o.rank_dir); dup(copy.prefix, o.prefix); dup(copy.start, o.start);
    dup(copy.cnt, o.cnt); dup(copy.kind, o.kind); dup(copy.a_lo, o.a_lo); dup(copy.a_hi, o.a_hi);
    CBLX_HIP(hipStreamSynchronize(dst->stream));
    return copy;
}
}  // namespace

extern "C" {

uint32_t cblx_abi_version(void) { return CBLX_ABI_VERSION; }
const char* cblx_last_global_error(void) { return g_global_err.c_str(); }
const char* cblx_last_error(const cblx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_global_err.c_str(); }

int cblx_create(const cblx_params* p, cblx_ctx** out) {
    return guard(nullptr, [&] {
        if (!p || !out) throw Error(CBLX_EINVAL, "null argument");
        *out = nullptr;
        const Consts P = make_consts(*p);
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
    // The reference's call pattern is one insert_seq per record (/root/reference/examples/cbl.rs:160-163): ten million calls for
    // cfg 2. The common call — the record fits the current pinned blocks of the queue and the device room reserved behind them —
    // is two copies and a few compares: no HIP call (hipSetDevice alone cost most of the 85 ns a call took), no exception frame.
    // Anything else (first call, a full block to hand to the DMA engine, growth, the flush threshold, errors) takes the path below.
    if (c && seq) {
        Ingest& g = c->ing;
        Ingest::Writer &wb = g.wb, &wo = g.wo;
        if (len >= c->P.K && wb.blk[0] && wo.blk[0] && !g.staged && !g.streamed.active() && g.nbytes == (g.nseq ? g.last_end : 0) &&
            wb.fill + len < wb.cap && wo.fill + 8 < wo.cap && g.d_bases.n >= g.nbytes + len + 64 && g.d_off.n >= g.nseq + 2 &&
            g.nbytes + len < ingest_flush_bytes()) {
            std::memcpy(wb.blk[wb.cur] + wb.fill, seq, len);
            wb.fill += len;
            g.nbytes += len;
            const u64 end = g.nbytes;
            std::memcpy(wo.blk[wo.cur] + wo.fill, &end, 8);
            wo.fill += 8;
            g.nseq += 1;
            g.last_end = end;
            return CBLX_OK;
        }
    }
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
        // branch-free scan of sequences [i0, i1) (on several threads when there are millions of them): monotone? shortest length
        auto scan = [&](u64 i0, u64 i1, u32 max_threads, bool& mono_out, u64& minlen_out) {
            const unsigned hc = std::thread::hardware_concurrency();
            const u64 cnt = i1 - i0;
            const u64 T = cnt >= (1u << 20) ? std::max(1u, std::min(max_threads, hc / 2)) : 1;
            std::vector<u8> mono(T, 1);
            std::vector<u64> minlen(T, ~0ull);
            auto part = [&](u64 t) {
                bool m = true;
                u64 ml = ~0ull;
                for (u64 i = i0 + cnt * t / T, e = i0 + cnt * (t + 1) / T; i < e; ++i) {
                    m &= offsets[i + 1] >= offsets[i];
                    ml = std::min(ml, offsets[i + 1] - offsets[i]);
                }
                mono[t] = m;
                minlen[t] = ml;
            };
            std::vector<std::thread> th;
            for (u64 t = 1; t < T; ++t) th.emplace_back(part, t);
            part(0);
            for (auto& x : th) x.join();
            mono_out = true;
            for (u64 t = 0; t < T; ++t) mono_out = mono_out && mono[t];
            minlen_out = *std::min_element(minlen.begin(), minlen.end());
        };
        auto validate = [&] {  // the whole batch; the offender is looked up only on failure
            bool mono;
            u64 ml;
            scan(0, n, 16u, mono, ml);
            if (!mono) throw Error(CBLX_EINVAL, "offsets must be non-decreasing");
            if (ml < c->P.K) throw Error(CBLX_ESHORT, "Sequence size (" + std::to_string(ml) + ") is smaller than K (" + std::to_string(c->P.K) + ")");
        };
        // a slice of a streamed batch, checked while the slices in front of it run on the device: anything wrong sends the whole
        // batch through `validate`, which then fails the way a check in front of everything would have
        auto check_slice = [&](u64 i0, u64 i1) {
            bool mono;
            u64 ml;
            scan(i0, i1, 4u, mono, ml);
            if (!mono || ml < c->P.K) { validate(); throw Error(CBLX_EINVAL, "offsets must be non-decreasing"); }
        };
        // a big batch into an empty queue crosses PCIe as bit planes and is inserted right behind the transfer (comm.hpp)
        if (ingest_seqs_planes(c, bases, offsets, n, check_slice)) return;
        ingest_seqs(c, bases, offsets, n, validate);
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

int cblx_insert_fastx_file(cblx_ctx* c, const char* path, uint64_t* n_records) {
    return guard(c, [&] { read_fastx_into_queue(c, path, n_records); });
}
int cblx_query_fastx_file(cblx_ctx* c, const char* path, uint64_t* n_records, uint64_t* total, uint64_t* positive) {
    return guard(c, [&] {
        if (total) *total = 0;
        if (positive) *positive = 0;
        flush(c);  // pending inserts first: the queue changes consumer
        Ingest& g = c->ing;
        g.query = true;
        g.q_total = g.q_positive = 0;
        try {
            read_fastx_into_queue(c, path, n_records);
            flush(c);
        } catch (...) {
            ingest_drop(c);
            g.query = false;
            throw;
        }
        g.query = false;
        if (total) *total = g.q_total;
        if (positive) *positive = g.q_positive;
    });
}

int cblx_stage_fastx_blocks(cblx_ctx* c, const char* path, uint64_t block, uint32_t rank, uint32_t world, const uint8_t** d_bases,
                            const uint64_t** d_offsets, uint64_t* n_staged, uint64_t* n_in_file) {
    return guard(c, [&] {
        if (!path || !n_in_file) throw Error(CBLX_EINVAL, "null argument");
        if (block && (world == 0 || rank >= world)) throw Error(CBLX_EINVAL, "rank must be below world");
        if (block && (!d_bases || !d_offsets || !n_staged)) throw Error(CBLX_EINVAL, "null argument");
        flush(c);  // whatever was enqueued for this index goes in first: the queue changes hands
        Ingest& g = c->ing;
        if (g.staged) throw Error(CBLX_EINVAL, "records are staged in this context already: call cblx_stage_release first");
        RecordFilter f;
        f.block = block; f.rank = rank; f.world = world; f.count_only = block == 0;
        try {
            read_fastx_into_queue(c, path, n_in_file, &f);
        } catch (...) { ingest_drop(c); throw; }
        if (block == 0) return;
        if (g.wb.blk[0]) writer_issue(c, g.wb, g.d_bases.get());
        if (g.wo.blk[0]) writer_issue(c, g.wo, (u8*)(g.d_off.get() + 1));
        ingest_wait(c);
        if (g.nseq == 0) ingest_reserve(c, 0, 0);  // a rank without records still gets valid (empty) arrays
        g.staged = true;
        *d_bases = g.d_bases.get();
        *d_offsets = g.d_off.get();
        *n_staged = g.nseq;
    });
}
int cblx_stage_release(cblx_ctx* c) {
    return guard(c, [&] { if (c->ing.staged) ingest_drop(c); });
}
int cblx_insert_words_device(cblx_ctx* c, const uint64_t* d_lo, const void* d_hi, uint64_t n) {
    return guard(c, [&] {
        flush(c);
        if (n == 0) return;
        if (!d_lo || (c->P.has_hi() && !d_hi)) throw Error(CBLX_EINVAL, "null argument");
        const size_t hs = hi_elem_size(c->P);
        const u64 step = 1ull << 31;  // sub-batches (32-bit positions inside one batch; same result, see insert_device)
        for (u64 a = 0; a < n; a += step) {
            const u64 m = std::min(step, n - a);
            dispatch(c->P, [&](auto cfg) {
                typedef decltype(cfg) C;
                Records rec;
                begin_records<C>(c, rec, m);
                rec.ext_lo = d_lo + a;  // the first partition pass reads the caller's arrays in place
                rec.ext_hi = hs ? (const void*)((const u8*)d_hi + a * hs) : d_hi;
                pipeline<C>(c, rec, m);
                c->kmers_inserted += m;
            });
        }
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

int cblx_sorted_batch_begin(cblx_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n, const uint32_t* bounds, uint32_t nd,
                            uint64_t* bucket_split, uint64_t* word_split) {
    return guard(c, [&] {
        if (nd == 0 || nd > MAX_DEST) throw Error(CBLX_EINVAL, "nd must be in 1.." + std::to_string(MAX_DEST));
        if (!bucket_split || !word_split || (nd > 1 && !bounds) || (n && (!d_bases || !d_offsets))) throw Error(CBLX_EINVAL, "null argument");
        for (u32 d = 0; d + 2 < nd; ++d) if (bounds[d + 1] < bounds[d]) throw Error(CBLX_EINVAL, "bounds must be ascending");
        if (n) check_aligned16(d_bases, "d_bases");
        dispatch(c->P, [&](auto cfg) { sorted_batch_begin<decltype(cfg)>(c, d_bases, d_offsets, n, bounds, nd, bucket_split, word_split); });
        collect_events(c);
    });
}
int cblx_sorted_batch_export(cblx_ctx* c, uint32_t* d_prefix, uint32_t* d_count, uint8_t* d_suffix) {
    return guard(c, [&] {
        if (c->batch.nb && (!d_prefix || !d_count || !d_suffix)) throw Error(CBLX_EINVAL, "null argument");
        dispatch(c->P, [&](auto cfg) { sorted_batch_export<decltype(cfg)>(c, d_prefix, d_count, d_suffix); });
    });
}
int cblx_insert_sorted_batches_device(cblx_ctx* c, const cblx_batch_view* batches, uint32_t n_batches) {
    return guard(c, [&] {
        flush(c);
        if (n_batches == 0) return;
        if (!batches) throw Error(CBLX_EINVAL, "null argument");
        dispatch(c->P, [&](auto cfg) { insert_sorted_batches<decltype(cfg)>(c, batches, n_batches); });
        collect_events(c);
    });
}

int cblx_comm_unique_id(uint8_t* id) {
    return guard(nullptr, [&] {
        if (!id) throw Error(CBLX_EINVAL, "null argument");
        Id128 uid;
        CBLX_RCCL(rccl().GetUniqueId(&uid));
        std::memcpy(id, uid.internal, sizeof uid.internal);
    });
}
int cblx_comm_init_rccl(cblx_comm** out, const uint8_t* id, uint32_t rank, uint32_t world, int32_t device) {
    return guard(nullptr, [&] {
        if (!out || !id) throw Error(CBLX_EINVAL, "null argument");
        *out = nullptr;
        if (world == 0 || rank >= world || world > MAX_DEST) throw Error(CBLX_EINVAL, "rank must be below world, world at most " + std::to_string(MAX_DEST));
        int dev = device;
        if (dev < 0) CBLX_HIP(hipGetDevice(&dev));
        std::unique_ptr<cblx_comm> cm(new cblx_comm());
        cm->device = dev;
        cm->t.reset(new RcclTransport(id, rank, world, dev));
        *out = cm.release();
    });
}
int cblx_comm_init_transport(cblx_comm** out, const cblx_transport* t, uint32_t rank, uint32_t world, int32_t device) {
    return guard(nullptr, [&] {
        if (!out || !t || !t->all_reduce_sum_u64 || !t->all_to_all_u64 || !t->exchange) throw Error(CBLX_EINVAL, "null argument");
        *out = nullptr;
        if (world == 0 || rank >= world || world > MAX_DEST) throw Error(CBLX_EINVAL, "rank must be below world, world at most " + std::to_string(MAX_DEST));
        int dev = device;
        if (dev < 0) CBLX_HIP(hipGetDevice(&dev));
        std::unique_ptr<cblx_comm> cm(new cblx_comm());
        cm->device = dev;
        cm->t.reset(new CallbackTransport(*t, rank, world));
        *out = cm.release();
    });
}
int cblx_comm_init_sim(cblx_comm** out, uint32_t rank, uint32_t world, int32_t device, uint64_t store_id, double link_gbps) {
    return guard(nullptr, [&] {
        if (!out) throw Error(CBLX_EINVAL, "null argument");
        *out = nullptr;
        if (world < 2 || rank >= world || world > MAX_DEST || link_gbps < 0) throw Error(CBLX_EINVAL, "rehearsal: 2 <= world <= " + std::to_string(MAX_DEST) + ", rank below world");
        int dev = device;
        if (dev < 0) CBLX_HIP(hipGetDevice(&dev));
        CBLX_HIP(hipSetDevice(dev));
        std::unique_ptr<cblx_comm> cm(new cblx_comm());
        cm->device = dev;
        cm->t.reset(new SimTransport(SimStore::get(store_id, world), rank, world, link_gbps));
        *out = cm.release();
    });
}
int cblx_sim_store_free(uint64_t store_id) {
    return guard(nullptr, [&] {
        std::lock_guard<std::mutex> g(SimStore::mu());
        SimStore::all().erase(store_id);
    });
}
void cblx_comm_destroy(cblx_comm* cm) { delete cm; }
const char* cblx_comm_last_error(const cblx_comm* cm) { return cm ? cm->err.c_str() : g_global_err.c_str(); }
int cblx_comm_stats(cblx_comm* cm, cblx_exchange_stats* out, int reset) {
    if (!cm || !out) return CBLX_EINVAL;
    out->sent_bytes = cm->t->sent_bytes; out->recv_bytes = cm->t->recv_bytes; out->messages = cm->t->messages;
    if (reset) cm->t->sent_bytes = cm->t->recv_bytes = cm->t->messages = 0;
    return CBLX_OK;
}
int cblx_comm_set_protocol(cblx_comm* cm, uint32_t protocol) {
    if (!cm || (protocol != CBLX_PROTO_SORTED && protocol != CBLX_PROTO_BINS && protocol != CBLX_PROTO_AUTO && protocol != CBLX_PROTO_REPLICATE)) return CBLX_EINVAL;
    cm->protocol = protocol;
    return CBLX_OK;
}
int cblx_comm_protocol_used(const cblx_comm* cm, uint32_t* out) {
    if (!cm || !out) return CBLX_EINVAL;
    *out = cm->protocol_used;
    return CBLX_OK;
}
int cblx_comm_set_recv_groups(cblx_comm* cm, uint32_t groups) {
    if (!cm || groups > 14) return CBLX_EINVAL;
    cm->recv_groups = groups;
    cm->g_bounds.clear();  // the cuts are chosen again, for this number of groups
    cm->g_cuts.clear();
    return CBLX_OK;
}
int cblx_comm_groups_used(const cblx_comm* cm, uint32_t* out) {
    if (!cm || !out) return CBLX_EINVAL;
    *out = cm->groups_used;
    return CBLX_OK;
}
int cblx_comm_groups_fine(const cblx_comm* cm, uint32_t* out) {
    if (!cm || !out) return CBLX_EINVAL;
    *out = cm->groups_fine;
    return CBLX_OK;
}
int cblx_stage_fastx_blocks_comm(cblx_ctx* c, cblx_comm* cm, const char* path, uint64_t* block, uint32_t slices, const uint8_t** d_bases, const uint64_t** d_offsets,
                                 uint64_t* n_staged, uint64_t* n_in_file) {
    return guard(c, [&] {
        if (!cm || !path || !block || !d_bases || !d_offsets || !n_staged || !n_in_file) throw Error(CBLX_EINVAL, "null argument");
        flush(c);  // whatever was enqueued for this index goes in first: the queue changes hands
        Ingest& g = c->ing;
        if (g.staged) throw Error(CBLX_EINVAL, "records are staged in this context already: call cblx_stage_release first");
        Transport& T = *cm->t;
        bool shared = false;
        try {
            shared = fastx_stage_distributed(c, path, *block, slices, T.rank, T.world, [&](u64* v, size_t n) { T.all_reduce_sum_u64(v, n); }, n_staged, n_in_file);
        } catch (...) { ingest_drop(c); throw; }
        if (!shared) {  // every rank parses the file (the sequential reader's cases), keeping its own blocks
            RecordFilter f;
            if (*block == 0) {
                f.count_only = true;
                try { read_fastx_into_queue(c, path, n_in_file, &f); } catch (...) { ingest_drop(c); throw; }
                *block = std::max<u64>(1, ceil_div(*n_in_file, (u64)T.world * std::max(1u, slices)));
            }
            f.block = *block; f.rank = T.rank; f.world = T.world; f.count_only = false;
            try { read_fastx_into_queue(c, path, n_in_file, &f); } catch (...) { ingest_drop(c); throw; }
            if (g.wb.blk[0]) writer_issue(c, g.wb, g.d_bases.get());
            if (g.wo.blk[0]) writer_issue(c, g.wo, (u8*)(g.d_off.get() + 1));
        }
        ingest_wait(c);
        if (g.nseq == 0) ingest_reserve(c, 0, 0);  // a rank without records still gets valid (empty) arrays
        g.staged = true;
        *d_bases = g.d_bases.get();
        *d_offsets = g.d_off.get();
        *n_staged = g.nseq;
    });
}
int cblx_sharded_insert_seqs_device(cblx_ctx* c, cblx_comm* cm, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n, const uint64_t* slice_cuts,
                                    uint32_t n_slices, uint32_t* bounds, int* bounds_valid) {
    return guard(c, [&] {
        if (!cm || !slice_cuts || !bounds_valid || (cm->t->world > 1 && !bounds) || (n && (!d_bases || !d_offsets))) throw Error(CBLX_EINVAL, "null argument");
        if (n_slices == 0) throw Error(CBLX_EINVAL, "at least one slice (an empty one still takes part in the exchange)");
        if (c->device != cm->device) throw Error(CBLX_EINVAL, "ctx and communicator live on different devices");
        if (n) check_aligned16(d_bases, "d_bases");
        flush(c);  // keep stream order with anything enqueued earlier
        u32 dummy = 0;
        SimTransport* sim = dynamic_cast<SimTransport*>(cm->t.get());  // rehearsal: every rank uses the cuts the first one chose
        if (sim && sim->st->have_cuts) { cm->g_bounds = sim->st->g_bounds; cm->g_cuts = sim->st->g_cuts; }
        try {
            dispatch(c->P, [&](auto cfg) { sharded_insert<decltype(cfg)>(c, cm, d_bases, d_offsets, n, slice_cuts, n_slices, bounds ? bounds : &dummy, bounds_valid); });
        } catch (const Error& e) { cm->err = e.what(); throw; }
        if (sim && !sim->st->have_cuts) { sim->st->g_bounds = cm->g_bounds; sim->st->g_cuts = cm->g_cuts; sim->st->have_cuts = true; }
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
        // (one sizing pass; the download of a chunk of buckets runs while the next chunk is being emitted)
        if (serialize_device(c, true, blob, cap, [&](u64 lo, u64 hi) { xfer(c).d2h_copy(buf + lo, blob.bytes.get() + lo, hi - lo); })) {
            if (written) *written = blob.n;
            if (blob.over_cap) throw Error(CBLX_ERANGE, "buffer too small: need " + std::to_string(blob.n) + " bytes");
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
        // The device emitter's bytes go to the file chunk by chunk (the download of a chunk of buckets runs while the next chunk is
        // being emitted); the lanes write straight from their pinned slots. The file is opened by the first chunk: when the
        // device emitter does not take the index (an entry of 4 GiB or more) nothing has been created yet.
        DevBlob blob;
        int fd = -1;
        void* map = MAP_FAILED;
        std::atomic<bool> bad{false};
        struct Closer {
            int& fd; void*& map; DevBlob& blob;
            ~Closer() { if (map != MAP_FAILED) ::munmap(map, blob.n); if (fd >= 0) ::close(fd); }
        } closer{fd, map, blob};
        auto sink = [&](u64 lo, u64 hi) {
            if (fd < 0) {
                fd = ::open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
                if (fd < 0) throw Error(CBLX_EINVAL, std::string("Failed to create ") + path);
                // The lanes fill a shared mapping of the file when the file system gives one: concurrent pwrite()s to ONE file
                // serialise on its inode lock (3 GB/s on tmpfs with eight lanes), page faults of a mapping do not.
                if (blob.n >= (64u << 20) && ::ftruncate(fd, (off_t)blob.n) == 0) {
                    // the file's pages in ONE call first: page by page from the lanes' faults a 9.4 GB file on tmpfs took 2.6 s
                    // (the page allocation of one file does not scale over threads); allocated up front (0.55 s) the lanes only
                    // map and fill them (tools/dev_tmpfs_write.cpp). A file system that cannot preallocate just says so.
                    (void)::posix_fallocate(fd, 0, (off_t)blob.n);
                    map = ::mmap(nullptr, blob.n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                }
            }
            if (map != MAP_FAILED) {
                xfer(c).d2h(blob.bytes.get() + lo, hi - lo, [&](const u8* src, size_t off, size_t n) { std::memcpy((u8*)map + lo + off, src, n); });
                return;
            }
            xfer(c).d2h(blob.bytes.get() + lo, hi - lo, [&](const u8* src, size_t off, size_t n) {
                off += lo;
                while (n) {
                    const ssize_t w = ::pwrite(fd, src, n, (off_t)off);
                    if (w <= 0) { bad = true; return; }
                    src += w; off += (size_t)w; n -= (size_t)w;
                }
            });
        };
        if (serialize_device(c, true, blob, ~0ull, sink)) {
            bool ok = !bad;
            if (map != MAP_FAILED) { ok = ::munmap(map, blob.n) == 0 && ok; map = MAP_FAILED; }
            if (fd >= 0) { ok = ::close(fd) == 0 && ok; fd = -1; }
            if (!ok) throw Error(CBLX_EINVAL, std::string("Failed to write index to ") + path);
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
    // tearing down the page tables of a multi-GB mapping takes a tenth of a second (9.4 GB: 0.12 s of a 1.0 s load): a helper
    // thread does it while the caller goes on (as for a mapped FASTA file, fastx_parse.hpp)
    if (len >= (256u << 20)) {
        try { std::thread([m, len] { ::munmap(m, len); }).detach(); return rc; } catch (...) {}
    }
    ::munmap(m, len);
    return rc;
}
int cblx_load_shard_from_file(cblx_ctx* c, const char* path, uint32_t rank, uint32_t world, const uint32_t* bounds, int sequential, uint32_t* bounds_out,
                              cblx_shard_info* info) {
    return guard(c, [&] {
        if (!path || !info) throw Error(CBLX_EINVAL, "null argument");
        if (world == 0 || rank >= world) throw Error(CBLX_EINVAL, "rank must be below world");
        for (u32 d = 0; bounds && d + 2 < world; ++d) if (bounds[d + 1] < bounds[d]) throw Error(CBLX_EINVAL, "bounds must be ascending");
        std::memset(info, 0, sizeof *info);
        CBLX_HIP(hipStreamSynchronize(c->stream));
        ingest_drop(c);
        c->res = Resident();
        MappedFile f(path);
        Src s{f.d, f.d + f.n};
        const bool canon = s.u8_() != 0;
        const u64 nb = s.varint();
        if (nb > (1ull << c->P.PB)) throw Error(CBLX_EFORMAT, "index: more buckets than prefixes (wrong PREFIX_BITS?)");
        info->header_entries = nb;
        info->canonical = canon ? 1 : 0;
        c->P.canonical = canon ? 1 : 0;
        const u8 *body = s.p, *end = f.d + f.n;
        auto run = [&](auto ws) {
            constexpr bool WS = decltype(ws)::value;
            const ShardCuts cuts = shard_cuts<WS>(body, end, c->P, world, bounds, sequential != 0);
            info->begin_off = (u64)(cuts.start[rank] - f.d);
            info->end_off = (u64)(cuts.start[rank + 1] - f.d);
            for (u32 r = 1; bounds_out && r < world; ++r) bounds_out[r - 1] = bounds ? bounds[r - 1] : cuts.first[r];
            if (!cuts.ok) { info->exact = 0; return; }
            bool exact = true;
            load_range<WS>(c, cuts.start[rank], cuts.start[rank + 1], info->local_entries, exact, info->first_prefix, info->last_prefix);
            // the loaded prefixes must lie in this rank's range (a speculative start that parsed anyway would not)
            if (exact && info->local_entries) {
                const u32 lo_b = rank ? (bounds ? bounds[rank - 1] : cuts.first[rank]) : 0u;
                const u64 hi_b = rank + 1 < world ? (u64)(bounds ? bounds[rank] : cuts.first[rank + 1]) : (1ull << c->P.PB);
                if (info->first_prefix < lo_b || (u64)info->last_prefix >= hi_b) exact = false;
            }
            if (!exact) { c->res = Resident(); info->local_entries = 0; }
            info->exact = exact ? 1 : 0;
        };
        if (c->P.wide_suffix()) run(std::true_type()); else run(std::false_type());
    });
}
int cblx_index_shard_cuts(const cblx_params* p, const char* path, uint32_t world, const uint32_t* bounds, int sequential, uint64_t* offs, uint32_t* first,
                          int* ok) {
    return guard(nullptr, [&] {
        if (!p || !path || !offs || !first || !ok || world == 0) throw Error(CBLX_EINVAL, "null argument");
        const Consts P = make_consts(*p);
        MappedFile f(path);
        Src s{f.d, f.d + f.n};
        (void)s.u8_();
        (void)s.varint();
        ShardCuts cuts;
        if (P.wide_suffix()) cuts = shard_cuts<true>(s.p, f.d + f.n, P, world, bounds, sequential != 0);
        else cuts = shard_cuts<false>(s.p, f.d + f.n, P, world, bounds, sequential != 0);
        for (u32 r = 0; r <= world; ++r) { offs[r] = (u64)(cuts.start[r] - f.d); first[r] = cuts.first[r]; }
        *ok = cuts.ok ? 1 : 0;
    });
}
int cblx_resident_split(cblx_ctx* c, const uint32_t* bounds, uint32_t nd, uint64_t* bucket_split, uint64_t* word_split) {
    return guard(c, [&] {
        if (nd == 0 || !bucket_split || !word_split || (nd > 1 && !bounds)) throw Error(CBLX_EINVAL, "null argument");
        for (u32 d = 0; d + 2 < nd; ++d) if (bounds[d + 1] < bounds[d]) throw Error(CBLX_EINVAL, "bounds must be ascending");
        flush(c);
        resident_split(c, bounds, nd, bucket_split, word_split);
    });
}
int cblx_resident_export(cblx_ctx* c, uint32_t* d_prefix, uint32_t* d_count, uint8_t* d_kind, uint8_t* d_suffix) {
    return guard(c, [&] {
        flush(c);
        if (c->res.nb && (!d_prefix || !d_count || !d_kind || !d_suffix)) throw Error(CBLX_EINVAL, "null argument");
        dispatch(c->P, [&](auto cfg) { resident_export<decltype(cfg)>(c, d_prefix, d_count, d_kind, d_suffix); });
    });
}
int cblx_install_buckets_device(cblx_ctx* c, const cblx_bucket_view* parts, uint32_t n_parts) {
    return guard(c, [&] {
        if (n_parts && !parts) throw Error(CBLX_EINVAL, "null argument");
        dispatch(c->P, [&](auto cfg) { install_buckets<decltype(cfg)>(c, parts, n_parts); });
    });
}
int cblx_serialized_body_size(cblx_ctx* c, uint64_t* n_entries, uint64_t* nbytes) {
    return guard(c, [&] {
        if (!n_entries || !nbytes) throw Error(CBLX_EINVAL, "null argument");
        flush(c);
        BodyImage im;
        body_image(c, false, im);
        *n_entries = c->res.nb;
        *nbytes = im.total - im.hdr;
    });
}
int cblx_write_body_at(cblx_ctx* c, const char* path, uint64_t file_off) {
    return guard(c, [&] {
        if (!path) throw Error(CBLX_EINVAL, "null argument");
        flush(c);
        write_body_at(c, path, file_off);
    });
}
int cblx_merge_from(cblx_ctx* dst, cblx_ctx* self, cblx_ctx* other) {
    return guard(dst, [&] {
        if (!self || !other) throw Error(CBLX_EINVAL, "null argument");
        if (dst == self || dst == other || self == other) throw Error(CBLX_EINVAL, "merge_from: dst, self and other must be three different contexts");
        for (cblx_ctx* x : {self, other}) {
            if (dst->P.K != x->P.K || dst->P.PB != x->P.PB) throw Error(CBLX_EINVAL, "merge: K / PREFIX_BITS mismatch");
            if (dst->P.canonical != x->P.canonical) throw Error(CBLX_EINVAL, "One of the index is canonical while the other isn't");
        }
        if (dst->device != self->device) throw Error(CBLX_EINVAL, "merge_from: dst and self must live on the same device");
        CBLX_HIP(hipStreamSynchronize(dst->stream));
        dst->res = Resident();
        dst->batch = SortedBatch();
        ingest_drop(dst);
        for (cblx_ctx* x : {self, other}) {
            CBLX_HIP(hipSetDevice(x->device));
            flush(x);
            CBLX_HIP(hipStreamSynchronize(x->stream));
        }
        CBLX_HIP(hipSetDevice(dst->device));
        if (self->res.count == 0 && other->res.count == 0) return;
        if (other->res.count == 0) { dst->res = clone_resident(dst, self); return; }
        if (self->res.count == 0) { dst->res = clone_resident(dst, other); return; }  // every bucket other-only: cloned as stored
        const char* fp = std::getenv("CBLX_FORCE_PEER_COPY");
        const bool peer = dst->device != other->device || (fp && fp[0] == '1');
        Resident copy;
        if (peer) copy = clone_resident(dst, other);
        dispatch(dst->P, [&](auto cfg) { merge_direct<decltype(cfg)>(dst, self->res, peer ? copy : other->res); });
        collect_events(dst);
        CBLX_HIP(hipStreamSynchronize(dst->stream));
        if (peer) {  // (as in cblx_merge_assign: the merge sorted other's Vec buckets in the copy)
            Resident& o = other->res;
            CBLX_HIP(hipMemcpyPeerAsync(o.a_lo.get(), other->device, copy.a_lo.get(), dst->device, o.a_lo.n * sizeof(u64), dst->stream));
            if (o.a_hi.get()) CBLX_HIP(hipMemcpyPeerAsync(o.a_hi.get(), other->device, copy.a_hi.get(), dst->device, o.a_hi.n * sizeof(u64), dst->stream));
            CBLX_HIP(hipStreamSynchronize(dst->stream));
        }
    });
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
        CBLX_HIP(hipStreamSynchronize(other->stream));
        // `other` on another GPU (or CBLX_FORCE_PEER_COPY=1: tests): its resident index is copied into self's HBM over the
        // fabric (hipMemcpyPeer) and the merge runs here on the device like any other; there is no host merge.
        const char* fp = std::getenv("CBLX_FORCE_PEER_COPY");
        const bool peer = self->device != other->device || (fp && fp[0] == '1');
        Resident copy;
        if (peer || self->res.count == 0) {
            const Resident& o = other->res;
            copy.nb = o.nb;
            copy.count = o.count;
            auto dup = [&](auto& dst, const auto& src) {
                typedef typename std::remove_reference<decltype(*src.get())>::type T;
                if (!src.get()) return;
                dst = Buf<T>(self->pool, src.n);
                if (self->device == other->device) device_copy(self->stream, dst.get(), src.get(), src.n * sizeof(T));
                else CBLX_HIP(hipMemcpyPeerAsync(dst.get(), self->device, src.get(), other->device, src.n * sizeof(T), self->stream));
            };
            dup(copy.bv, o.bv); dup(copy.rank_dir, o.rank_dir); dup(copy.prefix, o.prefix); dup(copy.start, o.start);
            dup(copy.cnt, o.cnt); dup(copy.kind, o.kind); dup(copy.a_lo, o.a_lo); dup(copy.a_hi, o.a_hi);
            CBLX_HIP(hipStreamSynchronize(self->stream));
        }
        if (self->res.count == 0) {
            // every bucket is other-only: cloned as stored, kind and order kept (src/trievec/set_ops.rs:43-71) = a deep copy
            self->res = std::move(copy);
            return;
        }
        dispatch(self->P, [&](auto cfg) { merge_direct<decltype(cfg)>(self, self->res, peer ? copy : other->res); });
        collect_events(self);
        CBLX_HIP(hipStreamSynchronize(self->stream));
        if (peer) {
            // the reference's |= sorts other's Vec buckets that met a bucket of self (iter_sorted, src/trievec/mod.rs:209-220):
            // the merge did that to the copy, so the arena goes back to where `other` lives
            Resident& o = other->res;
            CBLX_HIP(hipMemcpyPeerAsync(o.a_lo.get(), other->device, copy.a_lo.get(), self->device, o.a_lo.n * sizeof(u64), self->stream));
            if (o.a_hi.get()) CBLX_HIP(hipMemcpyPeerAsync(o.a_hi.get(), other->device, copy.a_hi.get(), self->device, o.a_hi.n * sizeof(u64), self->stream));
            CBLX_HIP(hipStreamSynchronize(self->stream));
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
            Buf<u64> w_lo;
            Buf<u8> w_hi;
            const u64 nk = seq_words_of_host_seq<C>(c, seq, len, w_lo, w_hi);
            if (n) *n = nk;
            if (nk > cap) throw Error(CBLX_ERANGE, "output capacity too small");
            Buf<u8> d_out(c->pool, nk + 8);
            contains_words<C>(c, w_lo.get(), (const HiT*)w_hi.get(), nk, d_out.get());
            CBLX_HIP(hipStreamSynchronize(c->stream));
            xfer(c).d2h_copy(out, d_out.get(), nk);
        });
        collect_events(c);
    });
}
int cblx_contains_seqs(cblx_ctx* c, const uint8_t* bases, const uint64_t* offsets, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* n_out,
                       uint64_t* positive) {
    return guard(c, [&] {
        if (n_out) *n_out = 0;
        if (positive) *positive = 0;
        flush(c);
        if (n == 0) return;
        if (!bases || !offsets) throw Error(CBLX_EINVAL, "null argument");
        u64 minlen = ~0ull;
        bool mono = true;
        for (u64 i = 0; i < n; ++i) { mono &= offsets[i + 1] >= offsets[i]; minlen = std::min(minlen, offsets[i + 1] - offsets[i]); }
        if (!mono) throw Error(CBLX_EINVAL, "offsets must be non-decreasing");
        if (minlen < c->P.K) throw Error(CBLX_ESHORT, "Sequence size (" + std::to_string(minlen) + ") is smaller than K (" + std::to_string(c->P.K) + ")");
        const u64 b0 = offsets[0], nb = offsets[n] - b0;
        Buf<u8> d_b(c->pool, nb + 64), d_out(c->pool, out ? nb + 8 : 8);
        Buf<u64> d_o(c->pool, n + 1);
        std::vector<u64> rel(n + 1);
        for (u64 i = 0; i <= n; ++i) rel[i] = offsets[i] - b0;
        xfer(c).h2d_copy(d_b.get(), bases + b0, nb);  // pinned lanes: the caller's buffers are pageable
        xfer(c).h2d_copy(d_o.get(), rel.data(), (n + 1) * 8);
        xfer(c).sync();
        u64 tot = 0;
        query_device(c, d_b.get(), d_o.get(), n, out ? d_out.get() : nullptr, nb, &tot, positive);
        if (n_out) *n_out = tot;
        if (out) {
            if (tot > cap) throw Error(CBLX_ERANGE, "output capacity too small: " + std::to_string(tot) + " k-mers");
            xfer(c).d2h_copy(out, d_out.get(), tot);
        }
    });
}
int cblx_contains_seqs_device(cblx_ctx* c, const uint8_t* d_bases, const uint64_t* d_offsets, uint64_t n, uint8_t* d_out, uint64_t cap, uint64_t* n_out,
                              uint64_t* positive) {
    return guard(c, [&] {
        if (n_out) *n_out = 0;
        if (positive) *positive = 0;
        flush(c);
        if (n == 0) return;
        if (!d_bases || !d_offsets) throw Error(CBLX_EINVAL, "null argument");
        query_device(c, d_bases, d_offsets, n, d_out, cap, n_out, positive);
        CBLX_HIP(hipStreamSynchronize(c->stream));
    });
}
int cblx_contains_all(cblx_ctx* c, const uint8_t* seq, uint64_t len, int* out) {
    return guard(c, [&] {
        if (!out) throw Error(CBLX_EINVAL, "null argument");
        flush(c);
        if (len < c->P.K) throw Error(CBLX_ESHORT, "Sequence size (" + std::to_string(len) + ") is smaller than K (" + std::to_string(c->P.K) + ")");
        dispatch(c->P, [&](auto cfg) {
            typedef decltype(cfg) C;
            typedef typename C::HiT HiT;
            Buf<u64> w_lo;
            Buf<u8> w_hi;
            const u64 nk = seq_words_of_host_seq<C>(c, seq, len, w_lo, w_hi);
            Buf<u8> d_out(c->pool, nk + 8);
            Buf<u32> zeros(c->pool, 1);
            CBLX_HIP(hipMemsetAsync(zeros.get(), 0, 4, c->stream));
            contains_words<C>(c, w_lo.get(), (const HiT*)w_hi.get(), nk, d_out.get());
            hipLaunchKernelGGL(k_count_zero_u8, dim3((unsigned)std::min<u64>(1024, std::max<u64>(1, ceil_div(nk, 256)))), dim3(256), 0, c->stream, d_out.get(), nk, zeros.get());
            CBLX_HIP(hipGetLastError());
            *out = d2h<u32>(c, zeros.get()) == 0;
        });
        collect_events(c);
    });
}
int cblx_contains_kmers(cblx_ctx* c, const uint64_t* lo, const uint64_t* hi, uint64_t n, uint8_t* out) {
    return guard(c, [&] {
        flush(c);
        if (n == 0) return;
        if (!out) throw Error(CBLX_EINVAL, "null argument");
        dispatch(c->P, [&](auto cfg) {
            typedef decltype(cfg) C;
            typedef typename C::HiT HiT;
            Buf<u64> w_lo;
            Buf<u8> w_hi;
            words_of_host_kmers<C>(c, lo, hi, n, w_lo, w_hi);
            Buf<u8> d_out(c->pool, n + 8);
            contains_words<C>(c, w_lo.get(), (const HiT*)w_hi.get(), n, d_out.get());
            CBLX_HIP(hipStreamSynchronize(c->stream));
            xfer(c).d2h_copy(out, d_out.get(), n);
        });
    });
}
int cblx_insert_kmers(cblx_ctx* c, const uint64_t* lo, const uint64_t* hi, uint64_t n, uint8_t* was_absent) {
    return guard(c, [&] {
        flush(c);  // keep stream order with anything enqueued earlier
        if (n == 0) return;
        dispatch(c->P, [&](auto cfg) {
            typedef decltype(cfg) C;
            typedef typename C::HiT HiT;
            Buf<u64> w_lo;
            Buf<u8> w_hi;
            words_of_host_kmers<C>(c, lo, hi, n, w_lo, w_hi);
            if (was_absent) {  // the return values of n successive CBL::insert calls, before the index changes
                Buf<u8> flag(c->pool, n + 8);
                contains_words<C>(c, w_lo.get(), (const HiT*)w_hi.get(), n, flag.get());
                u64 slots = 64;
                while (slots < 2 * n) slots <<= 1;
                Buf<u32> table(c->pool, slots);
                CBLX_HIP(hipMemsetAsync(table.get(), 0, slots * 4, c->stream));
                hipLaunchKernelGGL(k_first_claim<HiT>, grid1(n, 256), dim3(256), 0, c->stream, w_lo.get(), (const HiT*)w_hi.get(), n, table.get(), (u32)(slots - 1));
                hipLaunchKernelGGL(k_first_flag<HiT>, grid1(n, 256), dim3(256), 0, c->stream, w_lo.get(), (const HiT*)w_hi.get(), n, table.get(), (u32)(slots - 1),
                                   flag.get());
                CBLX_HIP(hipGetLastError());
                CBLX_HIP(hipStreamSynchronize(c->stream));
                xfer(c).d2h_copy(was_absent, flag.get(), n);
            }
            Records rec;
            begin_records<C>(c, rec, n);
            rec.ext_lo = w_lo.get();  // the first partition pass reads the words in place
            rec.ext_hi = w_hi.get();
            pipeline<C>(c, rec, n);
            c->kmers_inserted += n;
            CBLX_HIP(hipStreamSynchronize(c->stream));  // w_lo / w_hi go back to the pool on return
        });
        collect_events(c);
    });
}
int cblx_export_kmers(cblx_ctx* c, uint64_t* lo, uint64_t* hi, uint64_t cap, uint64_t* n) {
    return guard(c, [&] {
        flush(c);
        const Resident& r = c->res;
        if (n) *n = r.count;
        if (r.count == 0) return;
        if (r.count > cap) throw Error(CBLX_ERANGE, "output capacity too small: the index holds " + std::to_string(r.count) + " k-mers");
        if (!lo || (c->P.wide_kmer() && !hi)) throw Error(CBLX_EINVAL, c->P.wide_kmer() ? "null argument (K >= 33 needs a hi array)" : "null argument");
        Buf<u64> res_off(c->pool, r.nb + 1), d_lo(c->pool, r.count), d_hi(c->pool, hi ? r.count : 1);
        const u64 tot = exclusive_scan<u64>(c, r.cnt.get(), r.nb, res_off.get());
        hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, res_off.get() + r.nb, tot);
        for (u64 e0 = 0; e0 < tot; e0 += 1ull << 31)  // one launch addresses fewer than 2^32 work items
            hipLaunchKernelGGL(k_export_kmers, grid1(std::min<u64>(1ull << 31, tot - e0), 256), dim3(256), 0, c->stream, e0, tot, r.nb, res_off.get(), r.prefix.get(), r.start.get(),
                               r.a_lo.get(), c->P.wide_suffix() ? r.a_hi.get() : (const u64*)nullptr, c->P, d_lo.get(), hi ? d_hi.get() : (u64*)nullptr);
        CBLX_HIP(hipGetLastError());
        CBLX_HIP(hipStreamSynchronize(c->stream));
        xfer(c).d2h_copy(lo, d_lo.get(), tot * 8);
        if (hi) xfer(c).d2h_copy(hi, d_hi.get(), tot * 8);
    });
}
int cblx_bucket_sizes(cblx_ctx* c, uint32_t* prefix, uint32_t* len, uint8_t* kind, uint64_t cap, uint64_t* n) {
    return guard(c, [&] {
        flush(c);
        const Resident& r = c->res;
        if (n) *n = r.nb;
        if (r.nb == 0) return;
        if (r.nb > cap) throw Error(CBLX_ERANGE, "output capacity too small: the index holds " + std::to_string(r.nb) + " buckets");
        if (prefix) CBLX_HIP(hipMemcpyAsync(prefix, r.prefix.get(), r.nb * 4, hipMemcpyDeviceToHost, c->stream));
        if (len) CBLX_HIP(hipMemcpyAsync(len, r.cnt.get(), r.nb * 4, hipMemcpyDeviceToHost, c->stream));
        if (kind) CBLX_HIP(hipMemcpyAsync(kind, r.kind.get(), r.nb, hipMemcpyDeviceToHost, c->stream));
        CBLX_HIP(hipStreamSynchronize(c->stream));
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
        for (u64 r0 = 0; r0 < r.nb; r0 += 1ull << 25)  // one wave per bucket, fewer than 2^32 work items per launch
            hipLaunchKernelGGL(k_validate, grid1(std::min<u64>(1ull << 25, r.nb - r0) * 64, 256), dim3(256), 0, c->stream, r0, r.nb, r.start.get(), r.cnt.get(), r.kind.get(),
                               r.a_lo.get(), c->P.wide_suffix() ? r.a_hi.get() : (const u64*)nullptr, c->P.SB, (u32)(strict != 0), bad.get());
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
    return guard(c, [&] { collect_events(c); for (auto& s : c->stages) { s.ms = 0; s.launches = 0; s.units = 0; } });
}
int cblx_stage_units(cblx_ctx* c, uint64_t* units, uint32_t cap, uint32_t* n) {
    return guard(c, [&] {
        u32 k = 0;
        for (; k < ST_N && k < cap; ++k) if (units) units[k] = c->stages[k].units;
        if (n) *n = k;
    });
}
int cblx_kmers_inserted(cblx_ctx* c, uint64_t* out) { if (!c || !out) return CBLX_EINVAL; *out = c->kmers_inserted; return CBLX_OK; }
int cblx_fine_builds(cblx_ctx* c, uint64_t* out) { if (!c || !out) return CBLX_EINVAL; *out = c->fine_builds; return CBLX_OK; }
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
        c->batch = SortedBatch();
        ingest_drop(c);
    });
}

}  // extern "C"
