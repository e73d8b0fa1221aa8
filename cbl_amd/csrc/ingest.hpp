// ingest.hpp — host sequences on their way to HBM: the pinned ingest queue behind cblx_insert_seq / cblx_insert_seqs, flush,
// and the FASTA / FASTQ reader behind cblx_insert_fastx_file. Included by cblx.cpp only.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>

#include <cstring>

#include <chrono>
#include <cstdio>

#include "fastx_parse.hpp"
#include "pipeline.hpp"

namespace {

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
// (comm.hpp) insert of device-resident sequences in slices, slice k as soon as ready[k] has fired
void insert_device_streamed(cblx_ctx* c, const u8* d_bases, const u64* d_offsets, u64 nseq, const Ingest::Streamed& plan);
// slices of a batch that is streamed from pinned host memory (CBLX_H2D_SLICES overrides; 1 = no streaming)
inline u32 h2d_slices() {
    const char* e = std::getenv("CBLX_H2D_SLICES");
    const u32 v = e ? (u32)std::strtoul(e, nullptr, 10) : 0;
    return v ? std::min(v, 64u) : 16u;
}
// one sequence (the cblx_insert_seq / FASTA-record granularity)
// a piece of the sequence being enqueued (a FASTA record arrives line by line), then its end
void ingest_bases(cblx_ctx* c, const u8* p, u64 len) {
    Ingest& g = c->ing;
    if (g.staged) throw Error(CBLX_EINVAL, "records are staged in this context: call cblx_stage_release first");
    g.streamed.clear();  // more than the one streamed batch in the queue: flush() waits for all of it
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
// n sequences at once. `validate` (offsets ascending, no sequence shorter than K: throws otherwise) runs while the bases are
// on their way over PCIe; nothing is enqueued when it throws.
template <typename V> void ingest_seqs(cblx_ctx* c, const u8* bases, const u64* offsets, u64 n, V&& validate) {
    Ingest& g = c->ing;
    const u64 len = offsets[n] >= offsets[0] ? offsets[n] - offsets[0] : 0;
    if (len < (1u << 20)) {
        validate();
        for (u64 i = 0; i < n; ++i) ingest_seq(c, bases + offsets[i], offsets[i + 1] - offsets[i]);
        return;
    }
    if (g.staged) throw Error(CBLX_EINVAL, "records are staged in this context: call cblx_stage_release first");
    ingest_reserve(c, len, n);
    if (!g.s) CBLX_HIP(hipStreamCreateWithFlags(&g.s, hipStreamNonBlocking));
    if (g.wb.blk[0]) writer_issue(c, g.wb, g.d_bases.get());
    if (g.wo.blk[0]) writer_issue(c, g.wo, (u8*)(g.d_off.get() + 1));
    Xfer& x = xfer(c);
    const u64 base = g.nbytes, o0 = offsets[0];
    auto send_offsets = [&] {
        x.h2d(g.d_off.get() + 1 + g.nseq, n * 8, [&](u8* dst, size_t off, size_t nb) {
            u64* d = (u64*)dst;
            const u64* src = offsets + off / 8 + 1;
            for (size_t j = 0; j < nb / 8; ++j) d[j] = base + (src[j] - o0);
        });
    };
    // A batch that is the whole queue, comes from pinned memory and goes into the insert pipeline is sent in slices that
    // land front to back (offsets first): flush() works on slice k while the later ones are on the wire.
    const u32 S = h2d_slices();
    const bool stream = S > 1 && g.nseq == 0 && g.nbytes == 0 && !g.query && len >= (64u << 20) && n >= 4 * (u64)S && len < ingest_flush_bytes() &&
                        Xfer::is_pinned(bases + o0) && Xfer::is_pinned(bases + o0 + len - 1);
    if (stream) {
        g.streamed.clear();
        // offsets first (slice 0 cannot be planned without them): straight from the caller's array when it is pinned and
        // starts at 0 (the copy IS the transform then), else through the lanes' slots; then the bases, slice after slice;
        // the checks of the offsets run on the host while all of that is on the wire
        if (o0 == 0 && Xfer::is_pinned(offsets) && Xfer::is_pinned(offsets + n)) {
            hipEvent_t e;
            x.h2d_pinned_lane0(g.d_off.get() + 1, offsets + 1, n * 8, e);
            g.streamed.offsets_ready.push_back(e);
        } else {
            send_offsets();
            x.mark(g.streamed.offsets_ready);
        }
        std::vector<size_t> bcuts(1, 0);
        g.streamed.seq_cuts.assign(1, 0);
        bool mono = true;
        for (u32 k = 1; k <= S && mono; ++k) {  // slice k ends at the first sequence boundary at or behind k / S of the bytes
            const u64 want = o0 + len * k / S;
            const u64 i = k == S ? n : (u64)(std::lower_bound(offsets, offsets + n + 1, want) - offsets);
            if (i <= g.streamed.seq_cuts.back()) continue;
            mono = offsets[i] >= o0 + bcuts.back() && offsets[i] - o0 <= len;  // (unsorted offsets: validate() reports them below)
            g.streamed.seq_cuts.push_back(i);
            bcuts.push_back((size_t)(offsets[i] - o0));
        }
        if (mono) x.h2d_pinned_sliced(g.d_bases.get(), bases + o0, bcuts, g.streamed.ready);
        try {
            validate();
            if (!mono) throw Error(CBLX_EINVAL, "offsets must be non-decreasing");
        } catch (...) { x.sync(); g.streamed.clear(); throw; }  // the queue's counters were not advanced: what was copied is ignored
    } else {
        x.h2d_copy(g.d_bases.get() + g.nbytes, bases + offsets[0], len);
        try { validate(); } catch (...) { x.sync(); throw; }  // the queue's counters were not advanced: the bytes just copied are ignored
        send_offsets();
    }
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
    g.staged = false;
    g.streamed.clear();
}
void ingest_destroy(cblx_ctx* c) {
    Ingest& g = c->ing;
    if (g.s) (void)hipStreamSynchronize(g.s);
    g.streamed.clear();
    if (g.pin) { (void)hipHostFree(g.pin); g.pin = nullptr; g.pin_cap = 0; }
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
    if (nseq == 0 || g.staged) return;  // staged records belong to the caller (cblx_stage_fastx_blocks), not to this index
    CBLX_HIP(hipSetDevice(c->device));
    if (g.wb.blk[0]) writer_issue(c, g.wb, g.d_bases.get());
    if (g.wo.blk[0]) writer_issue(c, g.wo, (u8*)(g.d_off.get() + 1));
    const bool streamed = g.streamed.active() && !g.query && g.streamed.seq_cuts.back() == nseq;
    if (!streamed) { ingest_wait(c); g.streamed.clear(); }
    // the pending queue is consumed even if the insert fails (the reference would have panicked)
    for (Ingest::Writer* w : {&g.wb, &g.wo}) { w->issued = 0; w->busy[0] = w->busy[1] = false; }
    g.nbytes = g.nseq = g.last_end = 0;
    if (streamed) {
        struct Done { cblx_ctx* c; ~Done() { try { ingest_wait(c); } catch (...) {} c->ing.streamed.clear(); } } done{c};
        insert_device_streamed(c, g.d_bases.get(), g.d_off.get(), nseq, g.streamed);
    } else if (g.query) {  // examples/cbl.rs:205-228: contains_seq per record, tallies only
        u64 tot = 0, pos = 0;
        query_device(c, g.d_bases.get(), g.d_off.get(), nseq, nullptr, 0, &tot, &pos);
        g.q_total += tot;
        g.q_positive += pos;
    } else {
        insert_device(c, g.d_bases.get(), g.d_off.get(), nseq);
    }
    CBLX_HIP(hipStreamSynchronize(c->stream));
}


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

// ---- the same reader, parallel (plain files): regions of the mapped file are cut at record starts; pass 1 counts the
// records and bases of every region (and finds anything irregular), prefix sums give every region its place in the
// pending queue, pass 2 copies the sequence lines through one Xfer lane per thread straight to that place. Anything
// irregular (a record shorter than K, a malformed FASTQ record, gzip, a small file) leaves it to the sequential reader
// above, which then reproduces the reference's behaviour record by record.
// threads of the counting pass when the caller has no better figure
inline unsigned fx_count_threads() {
    const unsigned hc = std::thread::hardware_concurrency();
    return std::max(1u, std::min(2u * (unsigned)Xfer::max_parallel(), hc ? hc / 2 : 2u));
}
// pass 2 over counted regions, in windows of about `window` bytes of bases: the sequence lines go through one transfer lane per
// thread straight to their place in the pending queue. flush_windows: insert every full window (the build from a file);
// else everything stays in the queue (staging). Returns the records copied.
inline u64 fx_copy_regions(cblx_ctx* c, const FastxMap& m, std::vector<FastxRegion>& regs, u64 window, bool flush_windows) {
    const int T = (int)std::min<size_t>((size_t)Xfer::max_parallel(), regs.size());
    const u32 K = c->P.K;
    const u8* d = m.d;
    const char fmt = m.fmt;
    Ingest& g = c->ing;
    u64 total_rec = 0;
    for (size_t w0 = 0; w0 < regs.size();) {
        size_t w1 = w0;
        u64 wb = 0, wr = 0;
        while (w1 < regs.size() && (w1 == w0 || wb + regs[w1].nbases <= window)) { wb += regs[w1].nbases; wr += regs[w1].nrec; ++w1; }
        if (wr) {
            ingest_reserve(c, wb, wr);
            if (!g.s) CBLX_HIP(hipStreamCreateWithFlags(&g.s, hipStreamNonBlocking));
            if (g.wb.blk[0]) writer_issue(c, g.wb, g.d_bases.get());
            if (g.wo.blk[0]) writer_issue(c, g.wo, (u8*)(g.d_off.get() + 1));
            std::vector<u64> base(w1 - w0 + 1, g.nbytes), rec0(w1 - w0 + 1, 0);
            for (size_t i = w0; i < w1; ++i) { base[i - w0 + 1] = base[i - w0] + regs[i].nbases; rec0[i - w0 + 1] = rec0[i - w0] + regs[i].nrec; }
            // record ends as absolute positions in the pending queue, written by the parsing threads straight to their place
            std::unique_ptr<u64[]> ends(new u64[wr]);
            Xfer& x = xfer(c);
            std::atomic<size_t> next{w0};
            std::atomic<bool> failed{false};
            const int TW = (int)std::min<size_t>((size_t)T, w1 - w0);
            x.with_lanes(TW, [&](int t) {
                for (size_t i; (i = next.fetch_add(1)) < w1;) {
                    FastxRegion& r = regs[i];
                    Xfer::LaneWriter lw(x, t, g.d_bases.get() + base[i - w0]);
                    struct Copy {
                        Xfer::LaneWriter& lw;
                        u64* out;       // this region's slots of `ends`
                        u64 cap, n, b0;
                        void seq(const u8* p, size_t len) { lw.put(p, len); }
                        void rec_end() { if (n < cap) out[n] = b0 + lw.written(); ++n; }
                    } cp{lw, ends.get() + rec0[i - w0], r.nrec, 0, base[i - w0]};
                    if (!fx_walk(d, r, fmt, K, cp) || lw.written() != r.nbases || cp.n != r.nrec) failed = true;
                    lw.finish();
                }
            });
            x.sync();
            if (failed) throw Error(CBLX_EDEVICE, "fastx: the file changed while it was being read");
            x.h2d_copy(g.d_off.get() + 1 + g.nseq, ends.get(), (size_t)wr * 8);
            x.sync();
            g.nbytes += wb;
            g.nseq += wr;
            g.last_end = g.nbytes;
            g.wb.issued = g.nbytes;
            g.wo.issued = g.nseq * 8;
            total_rec += wr;
            if (flush_windows && g.nbytes >= window) flush(c);
        }
        w0 = w1;
    }
    return total_rec;
}
bool fastx_parallel(cblx_ctx* c, const char* path, u64* nrec_out) {
    const size_t MIN_BYTES = fastx_env_bytes("CBLX_FASTX_PARALLEL_MIN", 32u << 20), REGION = fastx_env_bytes("CBLX_FASTX_REGION_BYTES", 16u << 20);
    FastxMap m;
    if (!m.open(path, MIN_BYTES)) return false;
    std::vector<FastxRegion> regs;
    fx_make_regions(m, m.first, m.size, REGION, regs);
    if (!fx_count_regions(m, regs, c->P.K, fx_count_threads())) return false;
    // pass 2, in windows of about 1 GiB of bases (the sequential reader's flush cadence)
    const u64 total = fx_copy_regions(c, m, regs, std::min<u64>(1ull << 30, ingest_flush_bytes()), true);
    if (nrec_out) *nrec_out = total;
    return true;
}

// ---- one file, W ranks, every rank touches 1 / W of it (the sharded build from a file, cblx_stage_fastx_blocks_comm) -------
// The blocks of the file are dealt to the ranks cyclically (block j -> rank j % W), so a rank needs the byte range of ITS
// blocks — which only a walk over the records can tell. That walk is shared: rank r counts the records of bytes
// [r, r + 1) * size / W (cut at record starts); the counts are summed over the ranks; every rank then knows the record number its
// range starts with, finds the byte offset of every block start INSIDE its range (a second, partial walk) and the offsets are
// summed again (each is known to exactly one rank). Pass 2 reads only the rank's own blocks through the pinned lanes.
// `reduce(vals, n)`: in-place sum over the ranks (the communicator's all-reduce). Returns false when the file is not one
// these readers take — decided from the file alone or from the summed flags, so every rank returns the same.
template <typename Reduce>
bool fastx_stage_distributed(cblx_ctx* c, const char* path, u64& block, u32 target_slices, u32 rank, u32 world, Reduce&& reduce, u64* n_staged, u64* n_in_file) {
    const size_t REGION = fastx_env_bytes("CBLX_FASTX_REGION_BYTES", 16u << 20);
    const bool trace = std::getenv("CBLX_INGEST_TRACE") != nullptr;  // phase times on stderr (tuning)
    auto t_phase = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto t = std::chrono::steady_clock::now();
        if (trace) std::fprintf(stderr, "[fastx rank %u/%u] %-22s %8.2f ms\n", rank, world, what, std::chrono::duration<double, std::milli>(t - t_phase).count());
        t_phase = t;
    };
    FastxMap m;
    u64 flag = m.open(path, 0) ? 0 : 1;
    const u32 K = c->P.K;
    // pass 1: the records of my byte range
    std::vector<FastxRegion> regs;
    size_t cut0 = 0, cut1 = 0;
    if (!flag) {
        auto cut = [&](u32 r) { return r == 0 ? m.first : (r >= world ? m.size : fx_next_record(m.d, m.size, std::max<size_t>(m.first, (size_t)((unsigned __int128)m.size * r / world)), m.fmt)); };
        cut0 = cut(rank);
        cut1 = cut(rank + 1);
        fx_make_regions(m, cut0, cut1, REGION, regs);
        if (!fx_count_regions(m, regs, K, fx_count_threads())) flag = 1;
    }
    std::vector<u64> v(world + 1, 0);
    u64 mine = 0;
    for (auto& r : regs) mine += r.nrec;
    v[rank] = flag ? 0 : mine;
    v[world] = flag;
    lap("map + count my range");
    reduce(v.data(), v.size());
    lap("sum of the counts");
    if (v[world]) return false;  // some rank met something irregular (a record shorter than K, a malformed FASTQ record, gzip)
    u64 n_file = 0, g0 = 0;
    for (u32 r = 0; r < world; ++r) { if (r < rank) g0 += v[r]; n_file += v[r]; }
    *n_in_file = n_file;
    if (block == 0) block = std::max<u64>(1, ceil_div(n_file, (u64)world * std::max(1u, target_slices)));
    const u64 nblocks = ceil_div(n_file, block);
    // byte offset of every block start inside my range
    std::vector<u64> boff(nblocks + 1, 0);
    {
        size_t ri = 0;
        u64 rfirst = g0;  // global number of region ri's first record
        for (u64 j = ceil_div(g0, block); j < nblocks && j * block < g0 + mine; ++j) {
            const u64 want = j * block;
            while (ri < regs.size() && rfirst + regs[ri].nrec <= want) { rfirst += regs[ri].nrec; ++ri; }
            size_t pos = regs[ri].beg;
            for (u64 k = rfirst; k < want; ++k) pos = fx_next_record(m.d, regs[ri].end, pos + 1, m.fmt);  // skip want - rfirst records
            boff[j] = pos;
        }
    }
    lap("block starts in my range");
    if (nblocks) reduce(boff.data(), nblocks);  // (the last entry is the same everywhere)
    lap("sum of the offsets");
    boff[nblocks] = m.size;
    // pass 2: my blocks j = rank, rank + W, ... in order
    std::vector<FastxRegion> mine_regs;
    u64 expect = 0;
    for (u64 j = rank; j < nblocks; j += world) {
        fx_make_regions(m, (size_t)boff[j], (size_t)boff[j + 1], REGION, mine_regs);
        expect += std::min<u64>(block, n_file - j * block);
    }
    if (!fx_count_regions(m, mine_regs, K, fx_count_threads())) throw Error(CBLX_EDEVICE, "fastx: the file changed while it was being read");
    lap("count my blocks");
    const u64 got = fx_copy_regions(c, m, mine_regs, ~0ull >> 1, false);
    lap("copy my blocks to HBM");
    if (got != expect) throw Error(CBLX_EDEVICE, "fastx: the file changed while it was being read");
    *n_staged = got;
    return true;
}

}  // namespace
