// host_index.hpp — the index as bytes: host-side copy of the resident index (export), the bincode emitter on the host and
// its device counterpart's driver (kernels_serde.hpp), the streaming / parallel loader. Included by cblx.cpp only.
#pragma once
#include <functional>
#include <sys/mman.h>
#include <sys/stat.h>

#include <fstream>
#include <mutex>

#include "ingest.hpp"
#include "kernels_serde.hpp"

namespace {

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
    for (u64 e0 = 0; e0 < n; e0 += 1ull << 31)  // one launch addresses fewer than 2^32 work items
        hipLaunchKernelGGL(k_gather_dense, grid1(std::min<u64>(1ull << 31, n - e0), 256), dim3(256), 0, c->stream, e0, n, r.nb, d_off.get(), r.start.get(), r.a_lo.get(),
                           c->P.wide_suffix() ? r.a_hi.get() : (const u64*)nullptr, c->P.SB, d_lo.get(), c->P.wide_suffix() ? d_hi.get() : (u64*)nullptr);
    CBLX_HIP(hipGetLastError());
    CBLX_HIP(hipStreamSynchronize(c->stream));
    h.lo.resize(n);
    xfer(c).d2h_copy(h.lo.data(), d_lo.get(), n * 8);  // pinned lanes (a pageable hipMemcpy runs at a few GB/s)
    if (c->P.wide_suffix()) { h.hi.resize(n); xfer(c).d2h_copy(h.hi.data(), d_hi.get(), n * 8); } else h.hi.clear();
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
struct DevBlob { Buf<u8> bytes; u64 n = 0; bool over_cap = false; };
// consumer of the emitted bytes, chunk by chunk: called with [lo, hi) once those bytes of blob.bytes are final, while the emitter
// goes on with the buckets behind them (the download of a chunk hides the emission of the next)
typedef std::function<void(u64, u64)> BlobChunkFn;
// false: only when an entry would not fit the 32-bit size table (>= 4 GiB) -> the caller takes the all-host path.
// emit with blob.n > cap: nothing is emitted, blob.over_cap is set.
template <typename C> bool serialize_device(cblx_ctx* c, bool emit, DevBlob& blob, u64 cap, const BlobChunkFn& on_chunk) {
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
    std::vector<u32> ln(SER_NCLS * SER_CHUNKS, 0), host_r;  // [class][chunk]
    std::vector<std::vector<u8>> host_bytes;
    const u64 *a_lo = r.a_lo.get(), *a_hi = WS ? r.a_hi.get() : (const u64*)nullptr;
    u32 nsplit = 0;
    Buf<u64> vstart, voff;
    Buf<u32> vcnt, vsize, vlists, vlist_n, split_bad;
    std::vector<u32> vln(3, 0);
    auto sub_buckets = [&](auto em, u8* outp) {  // the sub-ranges of split Tries: sizes (em = false) or bytes at body + voff
        constexpr bool EM = decltype(em)::value;
        const u64 nv = (u64)nsplit * 256;
        auto go = [&](auto th, auto it, u32 cls) {
            constexpr int TH = decltype(th)::value, IT = decltype(it)::value;
            hipLaunchKernelGGL((k_serde_bucket<TH, IT, WS, EM, true>), dim3(vln[cls]), dim3(TH), 0, c->stream, vlists.get() + cls * nv, vlist_n.get() + cls, (const u32*)nullptr, vstart.get(),
                               vcnt.get(), (const u8*)nullptr, a_lo, a_hi, P.SB, P.BYTES, vsize.get(), voff.get(), outp);
        };
        using std::integral_constant;
        if (vln[0]) {  // (4 elements per thread when emitting: see `buckets` below)
            if constexpr (EM) go(integral_constant<int, 256>(), integral_constant<int, 4>(), 0u);
            else go(integral_constant<int, 64>(), integral_constant<int, 16>(), 0u);
        }
        if (vln[1]) {
            if constexpr (EM) go(integral_constant<int, 1024>(), integral_constant<int, 4>(), 1u);
            else go(integral_constant<int, 256>(), integral_constant<int, 16>(), 1u);
        }
        if (vln[2])
            hipLaunchKernelGGL((k_serde_bucket<1024, 8, WS, EM, true>), dim3(vln[2]), dim3(1024), 0, c->stream, vlists.get() + 2 * nv, vlist_n.get() + 2, (const u32*)nullptr, vstart.get(),
                               vcnt.get(), (const u8*)nullptr, a_lo, a_hi, P.SB, P.BYTES, vsize.get(), voff.get(), outp);
        CBLX_HIP(hipGetLastError());
    };
    const u64 per = std::max<u64>(1, ceil_div(nb, (u64)SER_CHUNKS));  // buckets per chunk of the class lists
    auto LN = [&](u32 cls, u32 k = 0) -> u32& { return ln[cls * SER_CHUNKS + k]; };
    // the workgroup entries of chunks [k0, k1): sizes (em = false) or bytes
    auto buckets = [&](auto em, u8* body, u32 k0 = 0u, u32 k1 = SER_CHUNKS) {
        constexpr bool EM = decltype(em)::value;
        // The emitter runs as 4 elements per thread (the sizing pass as 16): its second walk keeps the first walk's offsets next to
        // the ranks, and at 16 elements per thread that is 240 VGPRs — one wave per SIMD (cfg 2's 9.4 GB: 92 ms; 47 ms like this).
        using std::integral_constant;
        for (u32 k = k0; k < k1; ++k) {
            auto go = [&](auto th, auto it, u32 cls) {
                constexpr int TH = decltype(th)::value, IT = decltype(it)::value;
                hipLaunchKernelGGL((k_serde_bucket<TH, IT, WS, EM>), dim3(LN(cls, k)), dim3(TH), 0, c->stream, lists.get() + (size_t)cls * nb + (size_t)k * per,
                                   list_n.get() + cls * SER_CHUNKS + k, r.prefix.get(), r.start.get(), r.cnt.get(), r.kind.get(), a_lo, a_hi, P.SB, P.BYTES, size.get(), off.get(), body);
            };
            if (LN(SER_C64, k)) {
                if constexpr (EM) go(integral_constant<int, 256>(), integral_constant<int, 4>(), SER_C64);
                else go(integral_constant<int, 64>(), integral_constant<int, 16>(), SER_C64);
            }
            if (LN(SER_C256, k)) {
                if constexpr (EM) go(integral_constant<int, 1024>(), integral_constant<int, 4>(), SER_C256);
                else go(integral_constant<int, 256>(), integral_constant<int, 16>(), SER_C256);
            }
            if (LN(SER_C1024, k)) go(integral_constant<int, 1024>(), integral_constant<int, 8>(), SER_C1024);
        }
        CBLX_HIP(hipGetLastError());
    };
    if (nb) {
        size = Buf<u32>(c->pool, nb);
        lists = Buf<u32>(c->pool, (size_t)SER_NCLS * nb);
        list_n = Buf<u32>(c->pool, SER_NCLS * SER_CHUNKS);
        off = Buf<u64>(c->pool, nb + 1);
        CBLX_HIP(hipMemsetAsync(list_n.get(), 0, SER_NCLS * SER_CHUNKS * 4, c->stream));
        hipLaunchKernelGGL((k_serde_tiny<WS, false>), grid1(nb, CLASSIFY_THREADS), dim3(CLASSIFY_THREADS), 0, c->stream, nb, r.prefix.get(), r.start.get(), r.cnt.get(), r.kind.get(), a_lo, a_hi,
                           P.SB, P.BYTES, size.get(), (const u64*)nullptr, (u8*)nullptr, lists.get(), list_n.get(), per);
        CBLX_HIP(hipGetLastError());
        ln = d2h_vec<u32>(c, list_n.get(), SER_NCLS * SER_CHUNKS);
        if (LN(SER_SPLIT)) {
            // long Tries, cut at the root: sub-ranges by top byte ("virtual buckets"), sized here by the workgroup kernels
            nsplit = LN(SER_SPLIT);
            const u64 nv = (u64)nsplit * 256;
            vstart = Buf<u64>(c->pool, nv);
            voff = Buf<u64>(c->pool, nv);
            vcnt = Buf<u32>(c->pool, nv);
            vsize = Buf<u32>(c->pool, nv);
            vlists = Buf<u32>(c->pool, 3 * nv);
            vlist_n = Buf<u32>(c->pool, 3);
            split_bad = Buf<u32>(c->pool, nsplit);
            CBLX_HIP(hipMemsetAsync(vlist_n.get(), 0, 12, c->stream));
            hipLaunchKernelGGL((k_serde_split_plan<WS>), dim3(nsplit), dim3(256), 0, c->stream, lists.get() + (size_t)SER_SPLIT * nb, list_n.get() + SER_SPLIT * SER_CHUNKS, r.start.get(),
                               r.cnt.get(), a_lo, a_hi, P.SB, P.BYTES, nb, vstart.get(), vcnt.get(), vsize.get(), vlists.get(), nv, vlist_n.get(), split_bad.get(),
                               lists.get() + (size_t)SER_HOST * nb, list_n.get() + SER_HOST * SER_CHUNKS);
            CBLX_HIP(hipGetLastError());
            ln = d2h_vec<u32>(c, list_n.get(), SER_NCLS * SER_CHUNKS);  // the plan may have handed buckets to the host emitter
            vln = d2h_vec<u32>(c, vlist_n.get(), 3);
            sub_buckets(std::false_type(), nullptr);
            hipLaunchKernelGGL(k_serde_split_size, dim3(nsplit), dim3(256), 0, c->stream, lists.get() + (size_t)SER_SPLIT * nb, list_n.get() + SER_SPLIT * SER_CHUNKS, r.prefix.get(), r.cnt.get(),
                               vcnt.get(), vsize.get(), split_bad.get(), size.get());
            CBLX_HIP(hipGetLastError());
        }
        if (LN(SER_HOST)) {
            // Buckets longer than one workgroup's emitter takes (low-complexity k-mers, tiny PREFIX_BITS): their entries
            // are emitted by host threads from a download of just those buckets and patched into the device-emitted
            // body at their offsets; every other bucket stays on the device path.
            host_r = d2h_vec<u32>(c, lists.get() + (size_t)SER_HOST * nb, LN(SER_HOST));
            const size_t nh = host_r.size();
            host_bytes.assign(nh, std::vector<u8>());
            std::vector<HostIndex> hb(nh);
            const u64 lo_mask = P.SB >= 64 ? ~0ull : ((1ull << P.SB) - 1ull);
            const u64 hi_mask = WS ? (P.SB >= 128 ? ~0ull : ((1ull << (P.SB - 64)) - 1ull)) : 0ull;
            // many such buckets (a deep index: one rank of a multi-GPU job at PREFIX_BITS = 24): one download of the tables
            // instead of four synchronous 4-byte reads per bucket
            std::vector<u32> all_prefix, all_cnt;
            std::vector<u8> all_kind;
            std::vector<u64> all_start;
            const bool bulk = nh > 64;
            if (bulk) {
                all_prefix = d2h_vec<u32>(c, r.prefix.get(), nb);
                all_cnt = d2h_vec<u32>(c, r.cnt.get(), nb);
                all_kind = d2h_vec<u8>(c, r.kind.get(), nb);
                all_start = d2h_vec<u64>(c, r.start.get(), nb);
            }
            for (size_t i = 0; i < nh; ++i) {
                const u32 rr = host_r[i];
                HostIndex& h = hb[i];
                h.prefix = {bulk ? all_prefix[rr] : d2h<u32>(c, r.prefix.get() + rr)};
                h.cnt = {bulk ? all_cnt[rr] : d2h<u32>(c, r.cnt.get() + rr)};
                h.kind = {bulk ? all_kind[rr] : d2h<u8>(c, r.kind.get() + rr)};
                h.off = {0, h.cnt[0]};
                const u64 st = bulk ? all_start[rr] : d2h<u64>(c, r.start.get() + rr);
                h.lo.resize(h.cnt[0]);
                xfer(c).d2h_copy(h.lo.data(), a_lo + st, (size_t)h.cnt[0] * 8);
                for (u64& x : h.lo) x &= lo_mask;
                if (WS) {
                    h.hi.resize(h.cnt[0]);
                    xfer(c).d2h_copy(h.hi.data(), a_hi + st, (size_t)h.cnt[0] * 8);
                    for (u64& x : h.hi) x &= hi_mask;
                }
            }
            std::atomic<size_t> next{0};
            std::atomic<bool> too_big{false};
            auto work = [&]() {
                for (size_t i; (i = next.fetch_add(1)) < nh;) {
                    const HostIndex& h = hb[i];
                    SfxView v{h.lo.data(), h.hi.empty() ? nullptr : h.hi.data()};
                    Sink cnt_(nullptr, 0);
                    serialize_bucket(P, h, v, 0, cnt_);
                    if (cnt_.pos >= (1ull << 32)) { too_big = true; continue; }
                    host_bytes[i].resize(cnt_.pos);
                    Sink out(host_bytes[i].data(), cnt_.pos);
                    serialize_bucket(P, h, v, 0, out);
                }
            };
            {
                unsigned nt = std::max(1u, std::min<unsigned>({std::thread::hardware_concurrency(), 16u, (unsigned)nh}));
                std::vector<std::thread> th;
                for (unsigned t = 1; t < nt; ++t) th.emplace_back(work);
                work();
                for (auto& x : th) x.join();
            }
            if (too_big) return false;  // an entry of 4 GiB or more: the all-host emitter (64-bit sizes) takes the call
            if (bulk) {  // the size table goes down, gets its host-emitted entries, and comes back: two copies instead of nh
                std::vector<u32> sz = d2h_vec<u32>(c, size.get(), nb);
                for (size_t i = 0; i < nh; ++i) sz[host_r[i]] = (u32)host_bytes[i].size();
                h2d(c, size.get(), sz.data(), nb);
                CBLX_HIP(hipStreamSynchronize(c->stream));
            } else {
                for (size_t i = 0; i < nh; ++i) {
                    const u32 sz = (u32)host_bytes[i].size();
                    CBLX_HIP(hipMemcpyAsync(size.get() + host_r[i], &sz, 4, hipMemcpyHostToDevice, c->stream));
                    CBLX_HIP(hipStreamSynchronize(c->stream));  // `sz` is a stack variable
                }
            }
        }
        buckets(std::false_type(), nullptr);
        total = exclusive_scan<u64>(c, size.get(), nb, off.get());
    }
    blob.n = hs.pos + total;
    if (!emit) return true;
    if (blob.n > cap) { blob.over_cap = true; return true; }
    bool handed = false;  // the chunks went to on_chunk one by one
    blob.bytes = Buf<u8>(c->pool, blob.n + 16);
    CBLX_HIP(hipMemcpyAsync(blob.bytes.get(), hdr, hs.pos, hipMemcpyHostToDevice, c->stream));
    if (nb) {
        u8* body = blob.bytes.get() + hs.pos;
        hipLaunchKernelGGL((k_serde_tiny<WS, true>), grid1(nb, CLASSIFY_THREADS), dim3(CLASSIFY_THREADS), 0, c->stream, nb, r.prefix.get(), r.start.get(), r.cnt.get(), r.kind.get(), a_lo, a_hi,
                           P.SB, P.BYTES, size.get(), off.get(), body, (u32*)nullptr, (u32*)nullptr, per);
        // (CBLX_SERDE_CHUNK_MIN: test hook — small indexes through the chunked hand-over)
        static const u64 chunk_min = [] { const char* e = std::getenv("CBLX_SERDE_CHUNK_MIN"); const unsigned long long v = e ? std::strtoull(e, nullptr, 10) : 0; return v ? (u64)v : (u64)(256u << 20); }();
        const bool chunked = (bool)on_chunk && nb >= 4 * SER_CHUNKS && blob.n >= chunk_min;
        if (!chunked) buckets(std::true_type(), body);
        if (nsplit) {  // split Tries: header / root / length by one kernel, then every sub-trie at its absolute offset (voff is relative to the blob)
            hipLaunchKernelGGL(k_serde_split_emit, dim3(nsplit), dim3(256), 0, c->stream, lists.get() + (size_t)SER_SPLIT * nb, list_n.get() + SER_SPLIT * SER_CHUNKS, r.prefix.get(), r.cnt.get(),
                               vcnt.get(), vsize.get(), split_bad.get(), off.get(), voff.get(), body);
            CBLX_HIP(hipGetLastError());
            sub_buckets(std::true_type(), body);
        }
        std::vector<u64> all_off;
        if (host_r.size() > 64) all_off = d2h_vec<u64>(c, off.get(), nb);
        for (size_t i = 0; i < host_r.size(); ++i) {
            const u64 o = all_off.empty() ? d2h<u64>(c, off.get() + host_r[i]) : all_off[host_r[i]];
            xfer(c).h2d_copy(body + o, host_bytes[i].data(), host_bytes[i].size());
        }
        if (!host_r.empty()) xfer(c).sync();
        if (chunked) {
            // the workgroup entries in chunks of consecutive buckets (everything else of the body is in place by now): chunk k's
            // bytes [cut[k], cut[k + 1]) are final when its launches are, and are handed over while chunk k + 1 is being emitted
            const u32 NCH = SER_CHUNKS;
            std::vector<u64> cut(NCH + 1, 0);
            for (u32 k = 1; k < NCH; ++k) cut[k] = (u64)k * per < nb ? hs.pos + d2h<u64>(c, off.get() + (u64)k * per) : blob.n;
            cut[NCH] = blob.n;
            // groups of chunks by bytes: 1/16 of the blob first (the download starts early), the rest in three parts (the low
            // prefixes hold the long buckets: chunks of equal bucket counts are far from equal bytes)
            std::vector<u32> grp(1, 0);
            {
                const double share[4] = {1.0 / 16, 6.0 / 16, 11.0 / 16, 1.0};
                for (int g = 0; g < 4; ++g) {
                    u32 k = grp.back();
                    while (k < NCH && (double)cut[k + 1] <= share[g] * (double)blob.n) ++k;
                    if (g == 3) k = NCH;
                    if (k == grp.back() && k < NCH) ++k;  // at least one chunk per group
                    if (k > grp.back()) grp.push_back(k);
                }
            }
            const u32 NG = (u32)grp.size() - 1;
            struct Events {
                std::vector<hipEvent_t> e;
                ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
            } ev;
            ev.e.assign(NG, nullptr);
            for (u32 g = 0; g < NG; ++g) {
                buckets(std::true_type(), body, grp[g], grp[g + 1]);
                CBLX_HIP(hipEventCreateWithFlags(&ev.e[g], hipEventDisableTiming));
                CBLX_HIP(hipEventRecord(ev.e[g], c->stream));
            }
            for (u32 g = 0; g < NG; ++g) {
                CBLX_HIP(hipEventSynchronize(ev.e[g]));
                if (cut[grp[g + 1]] > cut[grp[g]]) on_chunk(cut[grp[g]], cut[grp[g + 1]]);
            }
            handed = true;
        }
    }
    CBLX_HIP(hipStreamSynchronize(c->stream));
    if (on_chunk && !handed && blob.n) on_chunk(0, blob.n);
    return true;
}
bool serialize_device(cblx_ctx* c, bool emit, DevBlob& blob, u64 cap = ~0ull, const BlobChunkFn& on_chunk = BlobChunkFn()) {
    if (const char* e = std::getenv("CBLX_HOST_SERDE")) if (e[0] == '1') return false;  // test hook: force the host emitter
    bool ok = false;
    dispatch(c->P, [&](auto cfg) { ok = serialize_device<decltype(cfg)>(c, emit, blob, cap, on_chunk); });
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
// ---- index bytes -> resident index, streamed (cblx_load): one pass over the bytes, elements go to HBM as they are
// decoded. The format is a sequential pre-order walk (no lengths to skip by): small files are walked by one host thread
// with everything around it (pinned double buffering, DMA, directory upload) overlapped; big files are cut speculatively
// at recognised entry starts and walked by several (load_parallel).
struct StreamUp {  // single producer -> device array of u64
    static constexpr size_t CAP = 1u << 20;  // elements per pinned block
    static std::mutex& pool_mu() { static std::mutex m; return m; }
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
            std::lock_guard<std::mutex> lk(pool_mu());  // the pool is single-threaded; the parallel loader grows buffers from its threads
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

// One `prefix, TrieOrVec` entry -> its elements through `out` (room(k) / commit(k) for the
// lo and hi arrays). STRICT adds the checks the speculative splitter needs to tell an entry from bytes that merely look
// like one (ascending node values, element length bytes, non-empty nodes).
struct NullOut {  // validation only
    std::vector<u64> scratch = std::vector<u64>(StreamUp::CAP);
    u64* room_lo(size_t) { return scratch.data(); }
    u64* room_hi(size_t) { return scratch.data(); }
    void commit_lo(size_t) {}
    void commit_hi(size_t) {}
};
struct StreamOut {
    StreamUp* lo;
    StreamUp* hi;
    u64* room_lo(size_t k) { return lo->room(k); }
    u64* room_hi(size_t k) { return hi->room(k); }
    void commit_lo(size_t k) { lo->commit(k); }
    void commit_hi(size_t k) { hi->commit(k); }
};
template <bool WS, bool STRICT, typename Out>
void parse_entry(Src& s, const Consts& P, u64 nprefix, Out& out, u32& prefix_out, u32& cnt_out, u8& kind_out) {
    const u32 BYTES = P.BYTES;
    const u64 lo_mask = BYTES >= 8 ? ~0ull : ((1ull << (8 * BYTES)) - 1ull);
    const u64 hi_mask = WS ? ((BYTES >= 16) ? ~0ull : ((1ull << (8 * (BYTES - 8))) - 1ull)) : 0ull;
    const u64 p = s.varint();
    if (p >= nprefix) throw Error(CBLX_EFORMAT, "prefix out of range for PREFIX_BITS");
    prefix_out = (u32)p;
    const u64 tag = s.varint();
    u64 n = 0;
    if (tag == 0) {  // Vec: varint(n) then n x (varint(BYTES) | BYTES little-endian bytes), stored order
        n = s.varint();
        if (n > 0xFFFFFFF0ull) throw Error(CBLX_EFORMAT, "index: bucket too long");
        if (STRICT && (n == 0 || n > (u64)(s.end - s.p) / (1 + BYTES))) throw Error(CBLX_EFORMAT, "index: implausible Vec length");
        u64 left = n;
        while (left) {
            const size_t k = (size_t)std::min<u64>(left, StreamUp::CAP);
            u64* ol = out.room_lo(k);
            u64* oh = WS ? out.room_hi(k) : nullptr;
            // fast path: every element has the expected length byte and 16 readable bytes follow the chunk
            if ((u64)(s.end - s.p) >= (u64)k * (1 + BYTES) + 16) {
                const u8* q = s.p;
                bool regular = true;
                if (STRICT && q[0] != BYTES) throw Error(CBLX_EFORMAT, "index: element length byte");  // recogniser probes fail here, not after k elements
                for (size_t i = 0; i < k; ++i, q += 1 + BYTES) {
                    regular &= q[0] == BYTES;
                    ol[i] = load_le64(q + 1) & lo_mask;
                    if (WS) oh[i] = load_le64(q + 9) & hi_mask;
                    if (STRICT && !regular) break;
                }
                if (regular) { s.p = q; out.commit_lo(k); if (WS) out.commit_hi(k); left -= k; continue; }
                if (STRICT) throw Error(CBLX_EFORMAT, "index: element length byte");
            }
            for (size_t i = 0; i < k; ++i) {  // general path (length byte != BYTES, or the tail of the input)
                const u64 nbts = s.varint();
                if (STRICT && nbts != BYTES) throw Error(CBLX_EFORMAT, "index: element length byte");
                u128 x = 0;
                for (u64 b = 0; b < nbts; ++b) { const u8 v = s.u8_(); if (b < BYTES) x |= (u128)v << (8 * b); }
                ol[i] = (u64)x;
                if (WS) oh[i] = (u64)(x >> 64);
            }
            out.commit_lo(k);
            if (WS) out.commit_hi(k);
            left -= k;
        }
        kind_out = KIND_VEC;
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
            // The sparse levels of a Trie are chains of nodes with ONE entry — `01 value 01` three bytes each, about four of them
            // per word — and this walk is what bounds the loader (0.41 GB/s per thread on Trie data): such a node is taken here
            // without the general node's varints and checks (which it passes: one value, as many children). Same frames, same bytes.
            while (d + 1 < BYTES && (size_t)(s.end - s.p) >= 3 && s.p[0] == 1 && s.p[2] == 1) {
                st[d] = Fr{s.p + 1, 1u, 0u};
                set_byte(d, s.p[1]);
                ++d;
                s.p += 3;
            }
            bool descend = false;
            if (d + 1 == BYTES && (size_t)(s.end - s.p) >= 3 && s.p[0] == 1 && s.p[2] == 0) {  // ... and the leaf under it: `01 value 00`
                u64* ol = out.room_lo(1);
                ol[0] = alo | s.p[1];
                out.commit_lo(1);
                if (WS) { u64* oh = out.room_hi(1); oh[0] = ahi; out.commit_hi(1); }
                n += 1;
                s.p += 3;
            } else {
            const u64 cc = s.varint();
            if (cc > 256) throw Error(CBLX_EFORMAT, "index: trie node with more than 256 entries");
            if ((u64)(s.end - s.p) < cc) throw Error(CBLX_EFORMAT, "index: unexpected end of data");
            const u8* vals = s.p;
            s.p += cc;
            if (STRICT) {
                if (cc == 0) throw Error(CBLX_EFORMAT, "index: empty trie node");
                for (u64 i = 1; i < cc; ++i) if (vals[i] <= vals[i - 1]) throw Error(CBLX_EFORMAT, "index: trie node values not ascending");
            }
            const u64 nc = s.varint();
            if (d + 1 == BYTES) {
                if (nc != 0) throw Error(CBLX_EFORMAT, "index: leaf trie node with children");
                u64* ol = out.room_lo((size_t)cc);
                for (u64 i = 0; i < cc; ++i) ol[i] = alo | vals[i];
                out.commit_lo((size_t)cc);
                if (WS) { u64* oh = out.room_hi((size_t)cc); for (u64 i = 0; i < cc; ++i) oh[i] = ahi; out.commit_hi((size_t)cc); }
                n += cc;
            } else {
                if (nc != cc) throw Error(CBLX_EFORMAT, "index: trie node children count mismatch");
                if (cc) { st[d] = Fr{vals, (u32)cc, 0}; set_byte(d, vals[0]); ++d; descend = true; }
            }
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
        kind_out = KIND_TRIE;
    } else {
        throw Error(CBLX_EFORMAT, "index: bad TrieOrVec tag");
    }
    cnt_out = (u32)n;
}

// Speculative split for big files: the format has nothing to skip by, but an entry start can be RECOGNISED — from a byte
// offset, scan forward for a position from which a few consecutive entries parse under the strict checks with ascending
// prefixes. Every region [start_t, start_t+1) is then parsed by its own thread into its own pinned blocks / stream /
// device buffer. A wrong guess cannot survive: thread t must stop EXACTLY at start_t+1, the entries must add up to the
// header's count and the prefixes must ascend across regions — anything else sends the whole file down the sequential
// path (which also owns the error messages).
// A recognised start may still be a FAKE one: the tail of a Vec entry whose elements end in `00 00 01` reads as the entry
// `prefix 0, Vec, 1 element` followed by the true entries. Such a fake entry ends exactly where a true one starts (the
// entries after it parse), so the SECOND entry of a recognised run is a true one: callers only trust boundaries and
// prefixes from the second entry on, and reach the boundary they want by walking forward from there.
enum { FIND_OK = 0, FIND_NONE = 1, FIND_GAVE_UP = 2 };
static const u64 FIND_LIMIT = 64ull << 20;
struct Found {
    const u8* at = nullptr;   // recognised start (possibly fake)
    const u8* at2 = nullptr;  // start of the entry after it (trusted); null when the run has one entry only
    u32 prefix2 = 0;
};
template <bool WS> int find_entry(const u8* from, const u8* end, const Consts& P, u64 nprefix, NullOut& dry, Found& f) {
    if (from >= end) return FIND_NONE;
    const u8* stop = std::min(end, from + FIND_LIMIT);
    for (const u8* p = from; p < stop; ++p) {
        {   // cheap rejection without an exception: a prefix is a varint of at most 32 bits below 2^PREFIX_BITS, the tag is 0 or 1
            u64 pv = *p;
            const u8* q = p + 1;
            if (pv > 250) {
                const int nbts = pv == 0xFB ? 2 : pv == 0xFC ? 4 : 0;
                if (!nbts || (u64)(end - q) < (u64)nbts) continue;
                pv = 0;
                for (int i = 0; i < nbts; ++i) pv |= (u64)q[i] << (8 * i);
                q += nbts;
            }
            if (pv >= nprefix || q >= end || *q > 1) continue;
        }
        try {
            Src s{p, end};
            u32 last = 0, pf = 0, cn;
            u8 kd;
            int e = 0;
            Found g;
            g.at = p;
            for (; e < 4 && s.p < end; ++e) {
                const u8* here = s.p;
                parse_entry<WS, true>(s, P, nprefix, dry, pf, cn, kd);
                if (e && pf <= last) throw Error(CBLX_EFORMAT, "prefix order");
                if (e == 1) { g.at2 = here; g.prefix2 = pf; }
                last = pf;
            }
            if (e >= 2 || s.p == end) { f = g; return FIND_OK; }
        } catch (const Error&) {
        }
    }
    return stop == end ? FIND_NONE : FIND_GAVE_UP;
}

// strict walk over the entries from a trusted start `from`: first entry start that satisfies pred(offset, prefix), or `end`
template <bool WS, typename Pred>
const u8* walk_to(const u8* from, const u8* end, const Consts& P, u64 nprefix, NullOut& dry, Pred&& pred, u32& first_prefix, bool& ok) {
    try {
        Src s{from, end};
        while (s.p < end) {
            const u8* here = s.p;
            u32 pf, cn;
            u8 kd;
            parse_entry<WS, true>(s, P, nprefix, dry, pf, cn, kd);
            if (pred(here, pf)) { first_prefix = pf; return here; }
        }
    } catch (const Error&) {
        ok = false;
    }
    first_prefix = (u32)std::min<u64>(nprefix, 0xFFFFFFFFull);
    return end;
}

// first entry start at or after byte `cut`: synchronise on a run recognised well BEFORE the cut, walk forward from its
// second entry
template <bool WS> const u8* seek_offset(const u8* body, const u8* end, const Consts& P, u64 nprefix, const u8* cut, NullOut& dry, u32& first_prefix, bool& ok) {
    auto pred = [&](const u8* here, u32) { return here >= cut; };
    for (u64 back = 64u << 10;; back *= 8) {
        if ((u64)(cut - body) <= back) return walk_to<WS>(body, end, P, nprefix, dry, pred, first_prefix, ok);
        Found f;
        const int st = find_entry<WS>(cut - back, end, P, nprefix, dry, f);
        if (st == FIND_GAVE_UP) { ok = false; return end; }
        if (st == FIND_NONE) { first_prefix = (u32)std::min<u64>(nprefix, 0xFFFFFFFFull); return end; }  // `cut - back` lies in the last entry
        if (f.at2 && f.at2 <= cut) return walk_to<WS>(f.at2, end, P, nprefix, dry, pred, first_prefix, ok);
        if (!f.at2 && f.at < cut) {  // a single (last) entry recognised in front of the cut: nothing starts after the cut
            first_prefix = (u32)std::min<u64>(nprefix, 0xFFFFFFFFull);
            return end;
        }
        // the trusted boundary lies beyond the cut: look further back
    }
}

// nb = NB_UNKNOWN: a byte range of a file whose entry count is not known (sharded load); *n_entries receives the count
static const u64 NB_UNKNOWN = ~0ull;
template <bool WS> bool load_parallel(cblx_ctx* c, const Consts& P, const u8* body, const u8* end, u64 nb, u64 nprefix, u64* n_entries = nullptr) {
    const u64 len = (u64)(end - body);
    unsigned hc = std::thread::hardware_concurrency();
    // (measured on the 9.4 GB index of cfg 2, 256 cores: 16 / 24 / 32 / 48 / 64 threads 1.35 / 1.15 / 1.16 / 1.57 / 1.78 s)
    unsigned T = (unsigned)std::min<u64>({24ull, hc ? hc / 2 : 1ull, len / (48ull << 20)});
    if (const char* e = std::getenv("CBLX_LOAD_THREADS")) T = (unsigned)std::strtoul(e, nullptr, 10);
    if (T < 2 || (nb != NB_UNKNOWN && nb < 8 * (u64)T) || len < 64 * (u64)T) return false;
    std::vector<const u8*> start(T + 1, nullptr);
    start[0] = body;
    start[T] = end;
    {   // entry starts near the even cuts, found in parallel
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; ++t)
            th.emplace_back([&, t] {
                NullOut dry;
                u32 pf;
                bool ok = true;
                const u8* at = seek_offset<WS>(body, end, P, nprefix, body + len / T * t, dry, pf, ok);
                start[t] = ok ? at : nullptr;
            });
        for (auto& x : th) x.join();
        for (unsigned t = 1; t < T; ++t) if (!start[t] || start[t] < start[t - 1]) return false;
    }
    struct Part {
        std::vector<u32> prefix, cnt;
        std::vector<u8> kind;
        std::unique_ptr<StreamUp> lo, hi;
        Buf<u64> d_lo, d_hi;
        u64 total = 0;
        bool ok = false;
    };
    std::vector<Part> part(T);
    for (unsigned t = 0; t < T; ++t) {  // streams, pinned blocks and first device buffers from the calling thread
        const u64 guess = (u64)(start[t + 1] - start[t]) / 6 + 1024;
        part[t].lo.reset(new StreamUp(c, guess));
        if (WS) part[t].hi.reset(new StreamUp(c, guess));
    }
    // The arena the parts are concatenated into, allocated by a helper thread WHILE the regions are parsed: a fresh multi-GB
    // hipMalloc costs 0.25 - 0.8 s in a process that has not held that memory before (tools/dev_alloc_cost.cpp) — every CLI command
    // is such a process. Sized by a Vec's 7 bytes per word (a Trie takes more per word; a dense one less: then the exact size is
    // allocated afterwards, as before).
    Buf<u64> pre_lo, pre_hi;
    std::thread pre_alloc([&] {
        try {
            CBLX_HIP(hipSetDevice(c->device));
            // (no StreamUp::pool_mu here: Pool::alloc locks its own bookkeeping and calls hipMalloc outside that lock, so the parser
            // threads that grow their buffers meanwhile do not wait 0.25 - 0.8 s behind this allocation)
            pre_lo = Buf<u64>(c->pool, len / 7 + len / 64 + 4096);
            if (WS) pre_hi = Buf<u64>(c->pool, len / 7 + len / 64 + 4096);
        } catch (...) {
        }
    });
    struct JoinPre { std::thread& t; ~JoinPre() { if (t.joinable()) t.join(); } } join_pre{pre_alloc};
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; ++t)
            th.emplace_back([&, t] {
                try {
                    CBLX_HIP(hipSetDevice(c->device));
                    Part& pt = part[t];
                    StreamOut out{pt.lo.get(), pt.hi.get()};
                    Src s{start[t], end};
                    u32 pf, cn;
                    u8 kd;
                    while (s.p < start[t + 1]) {
                        parse_entry<WS, true>(s, P, nprefix, out, pf, cn, kd);
                        pt.prefix.push_back(pf); pt.cnt.push_back(cn); pt.kind.push_back(kd);
                        pt.total += cn;
                    }
                    if (s.p != start[t + 1]) return;  // walked over the next region's guessed start: the guess was wrong
                    pt.d_lo = pt.lo->finish();
                    if (WS) pt.d_hi = pt.hi->finish();
                    pt.ok = true;
                } catch (...) {
                }
            });
        for (auto& x : th) x.join();
    }
    u64 total = 0, entries = 0;
    bool have_last = false;
    u32 last_prefix = 0;
    for (unsigned t = 0; t < T; ++t) {
        if (!part[t].ok) return false;
        if (!part[t].prefix.empty()) {
            if (have_last && part[t].prefix.front() <= last_prefix) return false;
            last_prefix = part[t].prefix.back();
            have_last = true;
        }
        total += part[t].total;
        entries += part[t].prefix.size();
    }
    if (nb != NB_UNKNOWN && entries != nb) return false;
    if (n_entries) *n_entries = entries;
    std::vector<u32> prefix, cnt;
    std::vector<u8> kind;
    prefix.reserve(entries); cnt.reserve(entries); kind.reserve(entries);
    if (pre_alloc.joinable()) pre_alloc.join();
    Buf<u64> a_lo, a_hi;
    // adopted only when it is close to the exact size: a Trie-heavy file (12 bytes per word) would otherwise keep an arena of up to
    // 1.8x the words resident for the life of the index
    const u64 fit = total + 2 + (total + 2) / 8;
    if (pre_lo.n >= total + 2 && pre_lo.n <= fit && (!WS || (pre_hi.n >= total + 2 && pre_hi.n <= fit))) { a_lo = std::move(pre_lo); a_hi = std::move(pre_hi); }
    else {
        pre_lo.reset(); pre_hi.reset();
        a_lo = Buf<u64>(c->pool, total + 2);
        if (WS) a_hi = Buf<u64>(c->pool, total + 2);
    }
    u64 at = 0;
    for (unsigned t = 0; t < T; ++t) {
        Part& pt = part[t];
        prefix.insert(prefix.end(), pt.prefix.begin(), pt.prefix.end());
        cnt.insert(cnt.end(), pt.cnt.begin(), pt.cnt.end());
        kind.insert(kind.end(), pt.kind.begin(), pt.kind.end());
        if (pt.total) {
            CBLX_HIP(hipMemcpyAsync(a_lo.get() + at, pt.d_lo.get(), pt.total * 8, hipMemcpyDeviceToDevice, c->stream));
            if (WS) CBLX_HIP(hipMemcpyAsync(a_hi.get() + at, pt.d_hi.get(), pt.total * 8, hipMemcpyDeviceToDevice, c->stream));
        }
        at += pt.total;
    }
    CBLX_HIP(hipStreamSynchronize(c->stream));
    part.clear();
    install_index(c, prefix, cnt, kind, std::move(a_lo), std::move(a_hi));
    return true;
}

template <bool WS> void load_stream(cblx_ctx* c, const u8* data, u64 len, bool& canonical) {
    const Consts& P = c->P;
    Src s{data, data + len};
    canonical = s.u8_() != 0;
    const u64 nb = s.varint();
    const u64 nprefix = 1ull << P.PB;
    if (nb > nprefix) throw Error(CBLX_EFORMAT, "index: more buckets than prefixes (wrong PREFIX_BITS?)");
    if (load_parallel<WS>(c, P, s.p, s.end, nb, nprefix)) return;
    std::vector<u32> prefix(nb), cnt(nb);
    std::vector<u8> kind(nb);
    StreamUp lo(c, len / 6 + 1024);
    std::unique_ptr<StreamUp> hi;
    if (WS) hi.reset(new StreamUp(c, len / 6 + 1024));
    StreamOut out{&lo, hi.get()};
    u64 total = 0;
    for (u64 r = 0; r < nb; ++r) {
        parse_entry<WS, true>(s, P, nprefix, out, prefix[r], cnt[r], kind[r]);
        total += cnt[r];
    }
    if (s.p != s.end) throw Error(CBLX_EFORMAT, "index: trailing bytes");  // reject_trailing_bytes
    Buf<u64> a_lo = lo.finish(), a_hi;
    if (WS) a_hi = hi->finish();
    install_index(c, prefix, cnt, kind, std::move(a_lo), std::move(a_hi));
}


}  // namespace
