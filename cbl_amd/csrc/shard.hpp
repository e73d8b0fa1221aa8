// shard.hpp — one rank's share of a prefix-range sharded index: load of a prefix range of an index file, the resident
// index as a bucket batch (export / install: re-sharding to other bounds), and the rank-ordered save of the entries.
// Included by cblx.cpp only.
//
// Reference path this carries to N GPUs (BASELINE.json cfg 5): `cbl merge a b -o out` = read_index x 2
// (/root/reference/examples/cbl.rs:117-130), `cbl |= cbl2` (:275 -> src/cbl.rs:433-449 -> src/wordset/set_ops.rs:123-157),
// write_index (:132-142). Buckets are independent by prefix, so every step runs per prefix range; only a re-shard (two
// operands cut at different bounds) moves data between ranks.
#pragma once
#include "host_index.hpp"

namespace {

struct MappedFile {
    const u8* d = nullptr;
    size_t n = 0;
    explicit MappedFile(const char* path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) throw Error(CBLX_EINVAL, std::string("Failed to open ") + path);
        struct stat st;
        if (::fstat(fd, &st) != 0) { ::close(fd); throw Error(CBLX_EINVAL, std::string("Failed to stat ") + path); }
        n = (size_t)st.st_size;
        if (n == 0) { ::close(fd); throw Error(CBLX_EFORMAT, "index: unexpected end of data"); }
        void* m = ::mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) throw Error(CBLX_EINVAL, std::string("Failed to map ") + path);
        d = (const u8*)m;
    }
    MappedFile(const MappedFile&) = delete;
    ~MappedFile() { if (d) ::munmap((void*)d, n); }
};

// ---- entry starts ------------------------------------------------------------------------------------------------
// The format has no lengths to skip by (a pre-order walk). A rank that wants the entries of ITS
// prefix range only must still find where they start. Two ways:
//   speculative  an entry start can be RECOGNISED (find_entry: a few consecutive entries parse under the strict
//                checks with ascending prefixes); a bisection over byte offsets then finds the first entry of a prefix
//                range in O(log) probes. A wrong guess is caught by the neighbour: the rank in front must stop EXACTLY at
//                this rank's start, and the entries of all ranks must add up to the header's count (checked by the caller
//                over all ranks; cblx_shard_info.exact).
//   sequential   walk every entry from the first one. Always right; every rank reads the file up to the end of its range.
struct ShardCuts {
    std::vector<const u8*> start;  // world + 1 entry starts; [world] = end of the data
    std::vector<u32> first;        // first prefix of every range (nprefix when the range is empty and nothing follows)
    bool ok = true;                // false: the speculation failed, the caller must go sequential
};

// first entry start >= body whose prefix is >= bound: bisection over byte offsets on TRUSTED (second-entry) boundaries, then
// a short strict walk
template <bool WS> const u8* seek_prefix(const u8* body, const u8* end, const Consts& P, u64 nprefix, u32 bound, NullOut& dry, u32& first_prefix, bool& ok) {
    const u8 *lo = body, *hi = end;  // lo: trusted entry start, every entry in front of it has a prefix < bound
    const u64 WALK = 1ull << 20;
    while ((u64)(hi - lo) > WALK) {
        const u8* mid = lo + (hi - lo) / 2;
        Found f;
        const int st = find_entry<WS>(mid, end, P, nprefix, dry, f);
        if (st == FIND_GAVE_UP) { ok = false; return end; }
        if (st == FIND_NONE || !f.at2 || f.prefix2 >= bound) hi = mid;
        else lo = f.at2;  // the entry at at2 (and, prefixes ascending, everything before it) is below the bound
    }
    return walk_to<WS>(lo, end, P, nprefix, dry, [&](const u8*, u32 pf) { return pf >= bound; }, first_prefix, ok);
}

template <bool WS> ShardCuts shard_cuts(const u8* body, const u8* end, const Consts& P, u32 world, const u32* bounds, bool sequential) {
    const u64 nprefix = 1ull << P.PB, len = (u64)(end - body);
    const u32 none = (u32)std::min<u64>(nprefix, 0xFFFFFFFFull);
    ShardCuts c;
    c.start.assign(world + 1, end);
    c.first.assign(world + 1, none);
    c.start[0] = body;
    NullOut dry;
    if (sequential) {
        // one walk over all entries: range r starts at the first entry whose offset reaches cut r (byte-balanced) or whose
        // prefix reaches bounds[r - 1]
        Src s{body, end};
        u32 r = 1;
        bool first_entry = true;
        while (s.p < end) {
            const u8* here = s.p;
            u32 pf, cn;
            u8 kd;
            parse_entry<WS, true>(s, P, nprefix, dry, pf, cn, kd);
            if (first_entry) { c.first[0] = pf; first_entry = false; }
            while (r < world && (bounds ? pf >= bounds[r - 1] : (u64)(here - body) >= len / world * r)) {
                c.start[r] = here;
                c.first[r] = pf;
                ++r;
            }
        }
        return c;
    }
    if (body < end) {  // first prefix of the file (range 0 starts at the first entry by definition)
        try {
            Src s{body, end};
            u32 pf, cn;
            u8 kd;
            parse_entry<WS, true>(s, P, nprefix, dry, pf, cn, kd);
            c.first[0] = pf;
        } catch (const Error&) { c.ok = false; return c; }
    }
    for (u32 r = 1; r < world; ++r) {
        if (bounds) {
            c.start[r] = seek_prefix<WS>(body, end, P, nprefix, bounds[r - 1], dry, c.first[r], c.ok);
        } else {
            c.start[r] = seek_offset<WS>(body, end, P, nprefix, body + len / world * r, dry, c.first[r], c.ok);
        }
        if (!c.ok) return c;
        if (c.start[r] < c.start[r - 1]) { c.ok = false; return c; }
    }
    return c;
}

// entries of the byte range [begin, end) -> resident index. exact = the walk ended at `end` (and, for a speculative start,
// nothing looked wrong on the way)
template <bool WS> void load_range(cblx_ctx* c, const u8* begin, const u8* end, u64& n_entries, bool& exact, u32& first, u32& last) {
    const Consts& P = c->P;
    const u64 nprefix = 1ull << P.PB;
    n_entries = 0;
    exact = true;
    c->res = Resident();
    if (begin >= end) return;
    auto first_last = [&]() {
        if (c->res.nb == 0) return;
        first = d2h<u32>(c, c->res.prefix.get());
        last = d2h<u32>(c, c->res.prefix.get() + (c->res.nb - 1));
    };
    try {
        if (load_parallel<WS>(c, P, begin, end, NB_UNKNOWN, nprefix, &n_entries)) { first_last(); return; }
        const u64 len = (u64)(end - begin);
        std::vector<u32> prefix, cnt;
        std::vector<u8> kind;
        StreamUp lo(c, len / 6 + 1024);
        std::unique_ptr<StreamUp> hi;
        if (WS) hi.reset(new StreamUp(c, len / 6 + 1024));
        StreamOut out{&lo, hi.get()};
        Src s{begin, end};
        u64 total = 0;
        while (s.p < end) {
            u32 pf, cn;
            u8 kd;
            parse_entry<WS, true>(s, P, nprefix, out, pf, cn, kd);
            if (!prefix.empty() && pf <= prefix.back()) throw Error(CBLX_EFORMAT, "prefixes are not strictly ascending");
            prefix.push_back(pf); cnt.push_back(cn); kind.push_back(kd);
            total += cn;
        }
        Buf<u64> a_lo = lo.finish(), a_hi;
        if (WS) a_hi = hi->finish();
        n_entries = prefix.size();
        install_index(c, prefix, cnt, kind, std::move(a_lo), std::move(a_hi));
        first_last();
    } catch (const Error& e) {
        if (e.code != CBLX_EFORMAT) throw;
        c->res = Resident();
        n_entries = 0;
        exact = false;
    }
}

// ---- resident index <-> bucket batch --------------------------------------------------------------------------------
// one wave per bucket: stored suffixes (arena, slack layout) -> BYTES little-endian bytes each at the bucket's dense offset
template <bool WS>
__global__ __launch_bounds__(256) void k_resident_pack(u64 nb, const u64* __restrict__ start, const u32* __restrict__ cnt, const u64* __restrict__ dense_off,
                                                       const u64* __restrict__ a_lo, const u64* __restrict__ a_hi, u32 SB, u32 BYTES, u8* __restrict__ out) {
    const u64 r = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nb) return;
    const u32 lane = threadIdx.x & 63, c = cnt[r];
    const u64 s0 = start[r], d0 = dense_off[r];
    for (u32 j = lane; j < c; j += 64) {
        const Sfx<WS> s = arena_sfx<WS>(a_lo, a_hi, s0 + j, SB);
        u8* o = out + (d0 + j) * BYTES;
        for (u32 k = 0; k < BYTES; ++k) o[k] = (u8)sfx_byte_le<WS>(s, k);
    }
}
// packed suffixes -> arena words (dense layout), one thread per word
template <bool WS>
__global__ void k_unpack_suffix(u64 n, const u8* __restrict__ packed, u32 BYTES, u64* __restrict__ out_lo, u64* __restrict__ out_hi) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u8* q = packed + i * BYTES;
    u64 lo = 0, hi = 0;
    for (u32 k = 0; k < BYTES; ++k) {
        const u64 b = q[k];
        if (k < 8) lo |= b << (8 * k); else hi |= b << (8 * (k - 8));
    }
    out_lo[i] = lo;
    if constexpr (WS) out_hi[i] = hi;
}
__global__ void k_check_kinds(u64 nb, const u8* __restrict__ kind, const u32* __restrict__ cnt, u32* __restrict__ bad) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nb && (kind[i] > KIND_TRIE || cnt[i] == 0)) atomicAdd(bad, 1u);
}

// dense word offset of every resident bucket (+ total), for split / export
u64 resident_dense_offsets(cblx_ctx* c, Buf<u64>& off) {
    const Resident& r = c->res;
    off = Buf<u64>(c->pool, r.nb + 1);
    const u64 tot = exclusive_scan<u64>(c, r.cnt.get(), r.nb, off.get());
    hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, off.get() + r.nb, tot);
    CBLX_HIP(hipGetLastError());
    return tot;
}

void resident_split(cblx_ctx* c, const u32* bounds, u32 nd, u64* bucket_split, u64* word_split) {
    const Resident& r = c->res;
    for (u32 d = 0; d <= nd; ++d) bucket_split[d] = word_split[d] = 0;
    bucket_split[nd] = r.nb;
    word_split[nd] = r.count;
    if (r.nb == 0 || nd < 2) return;
    Buf<u64> off;
    resident_dense_offsets(c, off);
    Buf<u32> d_bounds(c->pool, nd);
    Buf<u64> d_bs(c->pool, nd), d_ws(c->pool, nd);
    h2d(c, d_bounds.get(), bounds, nd - 1);
    hipLaunchKernelGGL(k_batch_split, grid1(nd - 1, 64), dim3(64), 0, c->stream, r.nb, r.prefix.get(), off.get(), nd - 1, d_bounds.get(), d_bs.get(), d_ws.get());
    CBLX_HIP(hipGetLastError());
    std::vector<u64> bs = d2h_vec<u64>(c, d_bs.get(), nd - 1), ws = d2h_vec<u64>(c, d_ws.get(), nd - 1);
    for (u32 d = 1; d < nd; ++d) { bucket_split[d] = bs[d - 1]; word_split[d] = ws[d - 1]; }
}

template <typename C> void resident_export(cblx_ctx* c, u32* d_prefix, u32* d_count, u8* d_kind, u8* d_suffix) {
    constexpr bool WS = C::WS;
    const Resident& r = c->res;
    if (r.nb == 0) return;
    Buf<u64> off;
    resident_dense_offsets(c, off);
    CBLX_HIP(hipMemcpyAsync(d_prefix, r.prefix.get(), r.nb * 4, hipMemcpyDeviceToDevice, c->stream));
    CBLX_HIP(hipMemcpyAsync(d_count, r.cnt.get(), r.nb * 4, hipMemcpyDeviceToDevice, c->stream));
    CBLX_HIP(hipMemcpyAsync(d_kind, r.kind.get(), r.nb, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL((k_resident_pack<WS>), dim3((unsigned)ceil_div(r.nb, 4)), dim3(256), 0, c->stream, r.nb, r.start.get(), r.cnt.get(), off.get(), r.a_lo.get(),
                       WS ? r.a_hi.get() : (const u64*)nullptr, c->P.SB, c->P.BYTES, d_suffix);
    CBLX_HIP(hipGetLastError());
    CBLX_HIP(hipStreamSynchronize(c->stream));
}

template <typename C> void install_buckets(cblx_ctx* c, const cblx_bucket_view* parts, u32 nparts) {
    constexpr bool WS = C::WS;
    const Consts& P = c->P;
    const u64 nprefix = 1ull << P.PB, nwords = std::max<u64>(1, nprefix / 64);
    u64 nb = 0, nw = 0;
    for (u32 i = 0; i < nparts; ++i) {
        if (parts[i].n_buckets && (!parts[i].d_prefix || !parts[i].d_count || !parts[i].d_kind)) throw Error(CBLX_EINVAL, "null bucket arrays");
        if (parts[i].n_words && !parts[i].d_suffix) throw Error(CBLX_EINVAL, "null bucket suffixes");
        nb += parts[i].n_buckets;
        nw += parts[i].n_words;
    }
    CBLX_HIP(hipStreamSynchronize(c->stream));
    ingest_drop(c);
    c->res = Resident();
    if (nb == 0) {
        if (nw) throw Error(CBLX_EINVAL, "bucket batch: words without buckets");
        return;
    }
    Resident nr;
    nr.bv = Buf<u64>(c->pool, nwords);
    nr.rank_dir = Buf<u64>(c->pool, nwords + 1);
    nr.prefix = Buf<u32>(c->pool, nb + 1);
    nr.start = Buf<u64>(c->pool, nb + 1);
    nr.cnt = Buf<u32>(c->pool, nb + 1);
    nr.kind = Buf<u8>(c->pool, nb + 1);
    nr.a_lo = Buf<u64>(c->pool, nw + 2);
    if (WS) nr.a_hi = Buf<u64>(c->pool, nw + 2);
    Buf<u32> popc(c->pool, nwords), bad(c->pool, 1);
    CBLX_HIP(hipMemsetAsync(nr.bv.get(), 0, nwords * 8, c->stream));
    CBLX_HIP(hipMemsetAsync(bad.get(), 0, 4, c->stream));
    u64 b0 = 0, w0 = 0;
    for (u32 i = 0; i < nparts; ++i) {
        const cblx_bucket_view& v = parts[i];
        if (v.n_buckets) {
            CBLX_HIP(hipMemcpyAsync(nr.prefix.get() + b0, v.d_prefix, v.n_buckets * 4, hipMemcpyDeviceToDevice, c->stream));
            CBLX_HIP(hipMemcpyAsync(nr.cnt.get() + b0, v.d_count, v.n_buckets * 4, hipMemcpyDeviceToDevice, c->stream));
            CBLX_HIP(hipMemcpyAsync(nr.kind.get() + b0, v.d_kind, v.n_buckets, hipMemcpyDeviceToDevice, c->stream));
        }
        for (u64 e0 = 0; e0 < v.n_words; e0 += 1ull << 31) {  // one launch addresses fewer than 2^32 work items
            const u64 m = std::min<u64>(1ull << 31, v.n_words - e0);
            hipLaunchKernelGGL((k_unpack_suffix<WS>), grid1(m, 256), dim3(256), 0, c->stream, m, v.d_suffix + e0 * P.BYTES, P.BYTES, nr.a_lo.get() + w0 + e0,
                               WS ? nr.a_hi.get() + w0 + e0 : (u64*)nullptr);
        }
        b0 += v.n_buckets;
        w0 += v.n_words;
    }
    // one pass over the concatenation: strictly ascending prefixes below 2^PREFIX_BITS, no empty bucket, valid kinds
    hipLaunchKernelGGL(k_batch_bits, grid1(nb, 256), dim3(256), 0, c->stream, nb, nr.prefix.get(), nr.cnt.get(), nprefix, nr.bv.get(), bad.get());
    hipLaunchKernelGGL(k_check_kinds, grid1(nb, 256), dim3(256), 0, c->stream, nb, nr.kind.get(), nr.cnt.get(), bad.get());
    hipLaunchKernelGGL(k_popc_words, grid1(nwords, 256), dim3(256), 0, c->stream, nwords, nr.bv.get(), popc.get());
    CBLX_HIP(hipGetLastError());
    nr.nb = exclusive_scan<u64>(c, popc.get(), nwords, nr.rank_dir.get());
    if (d2h<u32>(c, bad.get()) || nr.nb != nb)
        throw Error(CBLX_EINVAL, "bucket batch: prefixes must be strictly ascending over all parts and below 2^PREFIX_BITS, counts non-zero, kinds 0 / 1");
    const u64 tot = exclusive_scan<u64>(c, nr.cnt.get(), nb, nr.start.get());
    hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, c->stream, nr.start.get() + nb, tot);
    CBLX_HIP(hipGetLastError());
    if (tot != nw) throw Error(CBLX_EINVAL, "bucket batch: n_words does not match the counts");
    nr.count = tot;
    CBLX_HIP(hipStreamSynchronize(c->stream));
    c->res = std::move(nr);
}

// ---- the entries alone (no header), for a file that several ranks write ---------------------------------------------
struct BodyImage {
    DevBlob blob;            // device image incl. header (when the device emitter took it)
    std::vector<u8> host;    // host image incl. header (all-host emitter)
    u64 hdr = 0, total = 0;  // header bytes, image bytes
    bool on_device = false;
};
void body_image(cblx_ctx* c, bool emit, BodyImage& im) {
    u8 hdr[16];
    Sink hs(hdr, sizeof hdr);
    hs.u8_(c->P.canonical ? 1 : 0);
    hs.varint(c->res.nb);
    im.hdr = hs.pos;
    if (serialize_device(c, emit, im.blob)) { im.on_device = true; im.total = im.blob.n; return; }
    HostIndex h;
    download(c, h);
    Sink cnt(nullptr, 0);
    serialize_host(c->P, h, cnt);
    im.total = cnt.pos;
    if (!emit) return;
    im.host.resize(cnt.pos);
    Sink s(im.host.data(), im.host.size());
    serialize_host(c->P, h, s);
}
void write_body_at(cblx_ctx* c, const char* path, u64 file_off) {
    BodyImage im;
    body_image(c, true, im);
    const u64 n = im.total - im.hdr;
    if (n == 0) return;
    const int fd = ::open(path, O_WRONLY);
    if (fd < 0) throw Error(CBLX_EINVAL, std::string("Failed to open ") + path + " for writing");
    std::atomic<bool> bad{false};
    auto put = [&](const u8* src, size_t off, size_t len) {
        while (len) {
            const ssize_t w = ::pwrite(fd, src, len, (off_t)(file_off + off));
            if (w <= 0) { bad = true; return; }
            src += w; off += (size_t)w; len -= (size_t)w;
        }
    };
    try {
        if (im.on_device) xfer(c).d2h(im.blob.bytes.get() + im.hdr, n, put);
        else put(im.host.data() + im.hdr, 0, n);
    } catch (...) { ::close(fd); throw; }
    if (::close(fd) != 0 || bad) throw Error(CBLX_EINVAL, std::string("Failed to write index to ") + path);
}

}  // namespace
