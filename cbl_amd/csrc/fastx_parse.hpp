// fastx_parse.hpp — the host-only part of the parallel FASTA / FASTQ readers (ingest.hpp, comm.hpp): a mapped file cut into regions
// at record starts, one walk over a region (what needletail's `seq()` yields for every record: the sequence lines with their line
// ends stripped, FASTQ qualities dropped; /root/reference/examples/cbl.rs:112-115,154-163), the counting pass, and PlaneSink, which
// appends a region's sequence lines to a batch's bit planes (kernels_encode.hpp BaseView). No HIP in here: tests/host/fastx_planes_unit.cpp
// builds it with g++ under AddressSanitizer / UBSan and checks it against a byte-by-byte definition.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define CBLX_FX_SSE2 1
#else
#define CBLX_FX_SSE2 0
#endif

namespace cblx {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

struct FastxRegion {
    size_t beg = 0, end = 0;
    u64 nrec = 0, nbases = 0;
    bool bad = false;
};
inline const u8* fx_line_end(const u8* p, const u8* end) {
    const u8* nl = (const u8*)std::memchr(p, '\n', (size_t)(end - p));
    return nl ? nl : end;
}
// next position >= pos where a record starts (or size)
inline size_t fx_next_record(const u8* d, size_t size, size_t pos, char fmt) {
    const u8* end = d + size;
    const u8* p = d + pos;
    if (pos != 0) {  // move to the start of the next line
        p = fx_line_end(p - 1, end);
        if (p < end) ++p;
    }
    while (p < end) {
        if (*p == (u8)fmt) {
            if (fmt == '>') return (size_t)(p - d);
            // FASTQ: a header is a '@' line whose second next line starts with '+' (a quality line starting with '@' is
            // followed by a header and then a sequence line, which never starts with '+')
            const u8* l1 = fx_line_end(p, end);
            const u8* l2 = l1 < end ? fx_line_end(l1 + 1, end) : end;
            if (l2 < end && l2 + 1 < end && l2[1] == '+') return (size_t)(p - d);
        }
        p = fx_line_end(p, end);
        if (p < end) ++p;
    }
    return size;
}
// one walk over a region; Sink: seq(ptr, n) for every piece of sequence, rec_end() after every record
template <typename Sink> bool fx_walk(const u8* d, const FastxRegion& r, char fmt, u32 K, Sink&& sink) {
    const u8* p = d + r.beg;
    const u8* end = d + r.end;
    auto line = [&](const u8*& b, size_t& n) -> bool {
        if (p >= end) return false;
        const u8* e = fx_line_end(p, end);
        b = p;
        n = (size_t)(e - p);
        if (n && b[n - 1] == '\r') --n;
        p = e < end ? e + 1 : end;
        return true;
    };
    const u8* b;
    size_t n;
    if (fmt == '>') {
        bool open_rec = false;
        u64 len = 0;
        while (line(b, n)) {
            if (n && b[0] == '>') {
                if (open_rec) { if (len < K) return false; sink.rec_end(); }
                open_rec = true;
                len = 0;
            } else if (n) {
                if (!open_rec) return false;  // sequence before the first header
                sink.seq(b, n);
                len += n;
            }
        }
        if (open_rec) { if (len < K) return false; sink.rec_end(); }
        return true;
    }
    while (line(b, n)) {
        if (n == 0) continue;  // blank line between records
        if (b[0] != '@') return false;
        const u8 *sq, *pl, *ql;
        size_t ns, npl, nq;
        if (!line(sq, ns) || !line(pl, npl) || !line(ql, nq)) return false;
        if (npl == 0 || pl[0] != '+' || ns < K) return false;
        sink.seq(sq, ns);
        sink.rec_end();
    }
    return true;
}
// a plain FASTA / FASTQ file mapped for the parallel readers
struct FastxMap {
    const u8* d = nullptr;
    size_t size = 0, first = 0;
    char fmt = 0;
    ~FastxMap() {
        // tearing down the page tables of a multi-GB mapping takes tens of milliseconds (30 ms for 1.7 GB): a helper thread
        // does it while the caller goes on
        if (!d) return;
        const u8* dd = d; const size_t nn = size;
        try { std::thread([dd, nn] { ::munmap((void*)dd, nn); }).detach(); } catch (...) { ::munmap((void*)dd, nn); }
    }
    // false: not a file these readers take (small, gzip, unreadable, no record at the start) — the sequential reader's case
    bool open(const char* path, size_t min_bytes) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (::fstat(fd, &st) != 0 || (size_t)st.st_size < std::max<size_t>(min_bytes, 2)) { ::close(fd); return false; }
        size = (size_t)st.st_size;
        const u8* m = (const u8*)::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == (const u8*)MAP_FAILED) return false;
        d = m;
        (void)::madvise((void*)d, size, MADV_SEQUENTIAL);
        if (d[0] == 0x1f && d[1] == 0x8b) return false;  // gzip
        while (first < size && (d[first] == '\n' || d[first] == '\r')) ++first;
        if (first == size || (d[first] != '>' && d[first] != '@')) return false;
        fmt = (char)d[first];
        return true;
    }
};
inline size_t fastx_env_bytes(const char* name, size_t dflt) {  // test hooks: small files through the parallel paths
    const char* e = std::getenv(name);
    const unsigned long long v = e ? std::strtoull(e, nullptr, 10) : 0;
    return v ? (size_t)v : dflt;
}
// [beg, end) (beg at a record start) cut into regions of about `region` bytes at record starts
inline void fx_make_regions(const FastxMap& m, size_t beg, size_t end, size_t region, std::vector<FastxRegion>& regs) {
    for (size_t pos = beg; pos < end;) {
        size_t nxt = pos + region < end ? fx_next_record(m.d, m.size, pos + region, m.fmt) : end;
        if (nxt > end) nxt = end;
        FastxRegion r;
        r.beg = pos;
        r.end = nxt;
        regs.push_back(std::move(r));
        pos = nxt;
    }
}
// pass 1: records and bases of every region (threads; counting needs no transfer lanes). false: something irregular
inline bool fx_count_regions(const FastxMap& m, std::vector<FastxRegion>& regs, u32 K, unsigned threads) {
    if (regs.empty()) return true;
    const int TC = (int)std::min<size_t>(std::max(1u, threads), regs.size());
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    struct Count { u64 nrec = 0, nbases = 0; void seq(const u8*, size_t n) { nbases += n; } void rec_end() { ++nrec; } };
    auto body = [&] {
        for (size_t i; (i = next.fetch_add(1)) < regs.size();) {
            Count cnt;
            regs[i].bad = !fx_walk(m.d, regs[i], m.fmt, K, cnt);
            regs[i].nrec = cnt.nrec;
            regs[i].nbases = cnt.nbases;
        }
    };
    for (int t = 1; t < TC; ++t) th.emplace_back(body);
    body();
    for (auto& x : th) x.join();
    for (auto& r : regs) if (r.bad) return false;
    return true;
}

// A region's sequence lines appended to the bit planes of a batch IN PLACE: 16 bytes -> three 16-bit words by SSE2 movemask,
// shifted to the region's bit offset. Only the first and the last word a region touches can be shared with its neighbours: those
// take an atomic OR (the caller zeroes them before the threads start), every other word is a plain store.
struct PlaneSink {
    u32* codes; u16* valid;   // pinned staging of the window, indexed by group of 16 bases
    u64 pos;                  // bases written so far + the region's first base
    u64 a0 = 0, a1 = 0, av = 0;
    u32 nb;                   // bits waiting in the accumulators (the low `nb` bits)
    u64 gcur;                 // group the accumulators' low bits belong to
    bool shared = true;       // the next word to leave is the region's first: shared with the previous region
    u64* ends; u64 nrec = 0, cap;
    u64 end;                  // first base of the NEXT region: nothing is written at or past it (the counting pass fixed the layout;
                              // a file that changed between the two walks must not run into its neighbours' words or past the staging buffer)
    bool overflow = false;
    PlaneSink(u32* c, u16* v, u64 first_base, u64* e, u64 ecap, u64 end_base = ~0ull)
        : codes(c), valid(v), pos(first_base), nb((u32)(first_base & 15)), gcur(first_base >> 4), ends(e), cap(ecap), end(end_base) {}
    void word_out(bool last) {
        const u32 w = (u32)(a0 & 0xFFFFu) | ((u32)(a1 & 0xFFFFu) << 16);
        const u16 vw = (u16)(av & 0xFFFFu);
        if (shared || last) {  // a word another region also writes into (zeroed before the threads started)
            __atomic_fetch_or(&codes[gcur], w, __ATOMIC_RELAXED);
            __atomic_fetch_or(&valid[gcur], vw, __ATOMIC_RELAXED);
            shared = false;
        } else {
            codes[gcur] = w;
            valid[gcur] = vw;
        }
        a0 >>= 16; a1 >>= 16; av >>= 16;
        ++gcur;
    }
    void put(u32 p0, u32 p1, u32 v, u32 k) {  // k <= 16 bases as plane bits
        a0 |= (u64)p0 << nb; a1 |= (u64)p1 << nb; av |= (u64)v << nb;
        nb += k;
        if (nb >= 16) { word_out(false); nb -= 16; }
    }
    void seq(const u8* p, size_t n) {
        if (overflow || n > end - pos) { overflow = true; pos += n; return; }  // counted, never written: the caller sees pos != its region's end
        size_t i = 0;
#if CBLX_FX_SSE2
        const __m128i up = _mm_set1_epi8((char)0xDF), cA = _mm_set1_epi8('A'), cC = _mm_set1_epi8('C'), cG = _mm_set1_epi8('G'), cT = _mm_set1_epi8('T');
        auto planes16 = [&](const __m128i v, u32& p0, u32& p1, u32& ok) {
            const __m128i u = _mm_and_si128(v, up);
            const __m128i m = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(u, cA), _mm_cmpeq_epi8(u, cC)), _mm_or_si128(_mm_cmpeq_epi8(u, cG), _mm_cmpeq_epi8(u, cT)));
            p0 = (u32)_mm_movemask_epi8(_mm_slli_epi16(v, 6));
            p1 = (u32)_mm_movemask_epi8(_mm_slli_epi16(v, 5));
            ok = (u32)_mm_movemask_epi8(m);
        };
        for (; i + 16 <= n; i += 16) {
            u32 p0, p1, ok;
            planes16(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p + i)), p0, p1, ok);
            put(p0, p1, ok, 16);
        }
        if (i < n) {  // the line's tail through a local copy (a 16-byte load could run past the end of the mapping)
            alignas(16) u8 tmp[16] = {0};
            std::memcpy(tmp, p + i, n - i);
            u32 p0, p1, ok;
            planes16(_mm_load_si128(reinterpret_cast<const __m128i*>(tmp)), p0, p1, ok);
            const u32 k = (u32)(n - i), mk = (1u << k) - 1u;
            put(p0 & mk, p1 & mk, ok & mk, k);
        }
#else
        for (; i < n; ++i) {
            const u8 b = p[i], uc = b & 0xDF;
            put((b >> 1) & 1u, (b >> 2) & 1u, (u32)(uc == 'A' || uc == 'C' || uc == 'G' || uc == 'T'), 1);
        }
#endif
        pos += n;
    }
    void rec_end() { if (nrec < cap && !overflow) ends[nrec] = pos; ++nrec; }
    void finish() { if (nb) { word_out(true); nb = 0; } }
};

// the words two regions may share start from zero (PlaneSink ORs into them): base[i] = first base of region i, base[nr] = all bases
inline void fx_planes_prezero(const std::vector<u64>& base, u64 ngroups_cap, u32* codes, u16* valid) {
    for (const u64 b : base)
        for (const u64 gq : {b >> 4, (b ? b - 1 : 0) >> 4})
            if (gq < ngroups_cap) { codes[gq] = 0; valid[gq] = 0; }
}

}  // namespace cblx
