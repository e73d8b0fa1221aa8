// xfer.hpp — host <-> HBM transfers for caller-owned (pageable) buffers.
//
// The ABI borrows plain host pointers (include/cblx.h: cblx_insert_seq / cblx_insert_seqs / cblx_load / cblx_serialize).
// A hipMemcpy from pageable memory runs at a few GB/s on this platform; PCIe Gen5 x16 moves ~55 GB/s from pinned
// memory. So every bulk transfer goes through a small set of LANES: each lane is a host thread with its own HIP stream
// and two pinned slots; chunk i of a transfer belongs to lane i % L, which copies (or transforms) it into a slot and
// issues the DMA while the other slot of the lane is being filled. Chunks carry their destination offset, so lanes
// never need to agree on an order.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <thread>
#include <vector>

#include "common.hpp"

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define CBLX_HAVE_SSE2 1
#else
#define CBLX_HAVE_SSE2 0
#endif

#ifndef CBLX_PARSE_THREADS_DEFAULT
#define CBLX_PARSE_THREADS_DEFAULT 16u
#endif

namespace cblx {

// n ASCII bases (n a multiple of 16, or the tail of a batch) -> the bit planes of kernels_encode.hpp's BaseView: per 16 bases one
// dword of code planes (bit i = ASCII bit 1 of base i, bit 16 + i = ASCII bit 2: the nucleotide code is (b >> 1) & 3) and one
// 16-bit validity word (bit i = base i is one of ACGTacgt). 3 bits cross PCIe per base instead of 8; bits past n stay clear.
#if CBLX_HAVE_SSE2
// 32 bases per step where the host has AVX2 (checked once at run time: the library is built on another machine than it runs on)
__attribute__((target("avx2"))) inline size_t pack_planes_avx2(const u8* src, size_t ng /* groups of 16 */, u32* codes, u16* valid, u32& all /* AND of the validity masks */) {
    const __m256i up = _mm256_set1_epi8((char)0xDF), cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
    size_t g = 0;
    for (; g + 2 <= ng; g += 2) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 16 * g));
        const __m256i u = _mm256_and_si256(v, up);
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)), _mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT)));
        const u32 p0 = (u32)_mm256_movemask_epi8(_mm256_slli_epi16(v, 6));
        const u32 p1 = (u32)_mm256_movemask_epi8(_mm256_slli_epi16(v, 5));
        const u32 vm = (u32)_mm256_movemask_epi8(ok);
        codes[g] = (p0 & 0xFFFFu) | (p1 << 16);
        codes[g + 1] = (p0 >> 16) | (p1 & 0xFFFF0000u);
        valid[g] = (u16)vm;
        valid[g + 1] = (u16)(vm >> 16);
        all &= vm;
    }
    return g;
}
#endif
// Returns true when every one of the n bases is valid (the validity plane of such a stretch need not cross the link at all).
inline bool pack_planes(const u8* src, size_t n, u32* codes, u16* valid) {
    size_t g = 0;
    const size_t ng = n / 16;
    u32 all = 0xFFFFFFFFu;
#if CBLX_HAVE_SSE2
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) g = pack_planes_avx2(src, ng, codes, valid, all);
    all &= all >> 16;  // (both halves of the 32-base masks)
    all |= 0xFFFF0000u;
    const __m128i up = _mm_set1_epi8((char)0xDF), cA = _mm_set1_epi8('A'), cC = _mm_set1_epi8('C'), cG = _mm_set1_epi8('G'), cT = _mm_set1_epi8('T');
    for (; g < ng; ++g) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + 16 * g));
        const __m128i u = _mm_and_si128(v, up);
        const __m128i ok = _mm_or_si128(_mm_or_si128(_mm_cmpeq_epi8(u, cA), _mm_cmpeq_epi8(u, cC)), _mm_or_si128(_mm_cmpeq_epi8(u, cG), _mm_cmpeq_epi8(u, cT)));
        const u32 p0 = (u32)_mm_movemask_epi8(_mm_slli_epi16(v, 6));  // ASCII bit 1 of every byte -> its bit 7
        const u32 p1 = (u32)_mm_movemask_epi8(_mm_slli_epi16(v, 5));  // ASCII bit 2
        const u32 vm = (u32)_mm_movemask_epi8(ok);
        codes[g] = p0 | (p1 << 16);
        valid[g] = (u16)vm;
        all &= vm | 0xFFFF0000u;
    }
#endif
    for (; g * 16 < n; ++g) {
        u32 c = 0, ok = 0;
        for (size_t i = 0; i < 16 && g * 16 + i < n; ++i) {
            const u8 b = src[g * 16 + i], uc = b & 0xDF;
            c |= (u32)((b >> 1) & 1u) << i | (u32)((b >> 2) & 1u) << (16 + i);
            ok |= (u32)(uc == 'A' || uc == 'C' || uc == 'G' || uc == 'T') << i;
        }
        codes[g] = c;
        valid[g] = (u16)ok;
        const size_t cnt = std::min<size_t>(16, n - g * 16);
        if (ok != (cnt == 16 ? 0xFFFFu : (1u << cnt) - 1u)) all = 0;
    }
    return (all & 0xFFFFu) == 0xFFFFu;
}

class Xfer {
    struct Lane {
        hipStream_t s = nullptr;
        u8* slot[2] = {nullptr, nullptr};
        hipEvent_t ev[2] = {nullptr, nullptr};
        bool busy[2] = {false, false};
    };

public:
    static constexpr size_t SLOT = 8u << 20;       // bytes per pinned slot
    static constexpr size_t PARALLEL_MIN = 4u << 20;  // below this a transfer stays on the calling thread (lane 0)

    explicit Xfer(int device) : device_(device) {}
    Xfer(const Xfer&) = delete;
    Xfer& operator=(const Xfer&) = delete;
    ~Xfer() {
        for (auto& l : lanes_) {
            if (l.s) (void)hipStreamSynchronize(l.s);
            for (int k = 0; k < 2; ++k) {
                if (l.ev[k]) (void)hipEventDestroy(l.ev[k]);
                if (l.slot[k]) (void)hipHostFree(l.slot[k]);
            }
            if (l.s) (void)hipStreamDestroy(l.s);
        }
    }

    // host -> device: fill(dst_pinned, off, n) must produce bytes [off, off+n) of the logical source.
    // Returns when every chunk has been handed to the DMA engines; call sync() before the device reads `d_dst`.
    template <typename F> void h2d(void* d_dst, size_t bytes, F&& fill) {
        run(bytes, [&](Lane& l, int k, size_t off, size_t n) {
            fill(l.slot[k], off, n);
            CBLX_HIP(hipMemcpyAsync((u8*)d_dst + off, l.slot[k], n, hipMemcpyHostToDevice, l.s));
        });
    }
    // true when the runtime knows the host range as pinned (hipHostMalloc / hipHostRegister, e.g. a torch pin_memory tensor)
    static bool is_pinned(const void* p) {
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }  // pageable memory is simply unknown
        return a.type == hipMemoryTypeHost;
    }
    void h2d_copy(void* d_dst, const void* h_src, size_t bytes) {
        if (bytes >= PARALLEL_MIN && is_pinned(h_src) && is_pinned((const u8*)h_src + bytes - 1)) {
            // the caller's buffer is pinned already: DMA straight from it (no trip through the lanes' slots), split over the
            // lanes' streams so that several copy engines work; complete after sync() like every h2d
            const int L = lanes_for(bytes);
            ensure(L);
            const size_t per = ((bytes + (size_t)L - 1) / (size_t)L + 4095) & ~(size_t)4095;
            for (int i = 0; i < L; ++i) {
                const size_t off = (size_t)i * per;
                if (off >= bytes) break;
                CBLX_HIP(hipMemcpyAsync((u8*)d_dst + off, (const u8*)h_src + off, std::min(per, bytes - off), hipMemcpyHostToDevice, lanes_[(size_t)i].s));
            }
            return;
        }
        h2d(d_dst, bytes, [&](u8* dst, size_t off, size_t n) { std::memcpy(dst, (const u8*)h_src + off, n); });
    }
    // A pinned source in SLICES that land front to back: slice k = bytes [cuts[k], cuts[k + 1]) is split over the lanes'
    // streams, which take their pieces in slice order; ready[k] receives one event per lane, recorded behind the slice's
    // pieces — a consumer stream that waits on them may read the slice (and everything in front of it) while the
    // later slices are still on the wire. The caller destroys the events.
    void h2d_pinned_sliced(void* d_dst, const void* h_src, const std::vector<size_t>& cuts, std::vector<std::vector<hipEvent_t>>& ready) {
        const size_t bytes = cuts.empty() ? 0 : cuts.back();
        // few streams: the runtime maps streams onto a handful of hardware queues, and a consumer stream that shares its
        // queue with a copy stream runs BEHIND the copies queued there (measured: 8 copy streams made the sliced insert
        // slower than the one-shot copy); CBLX_H2D_LANES overrides
        static const int want = [] { const char* e = std::getenv("CBLX_H2D_LANES"); const int v = e ? std::atoi(e) : 0; return v > 0 ? std::min(v, 8) : 2; }();
        const int L = std::min(want, lanes_for(bytes));
        ensure(L);
        ready.assign(cuts.size() ? cuts.size() - 1 : 0, std::vector<hipEvent_t>());
        for (size_t k = 0; k + 1 < cuts.size(); ++k) {
            const size_t a = cuts[k], n = cuts[k + 1] - a;
            const size_t per = ((n + (size_t)L - 1) / (size_t)L + 4095) & ~(size_t)4095;
            for (int i = 0; i < L; ++i) {
                const size_t off = (size_t)i * per;
                if (off < n) CBLX_HIP(hipMemcpyAsync((u8*)d_dst + a + off, (const u8*)h_src + a + off, std::min(per, n - off), hipMemcpyHostToDevice, lanes_[(size_t)i].s));
                hipEvent_t e;
                CBLX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                ready[k].push_back(e);
                CBLX_HIP(hipEventRecord(e, lanes_[(size_t)i].s));
            }
        }
    }
    // a pinned source, whole, at the front of lane 0's queue; `done` fires when it has landed (the caller destroys it)
    void h2d_pinned_lane0(void* d_dst, const void* h_src, size_t bytes, hipEvent_t& done) {
        ensure(1);
        if (bytes) CBLX_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, lanes_[0].s));
        CBLX_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        CBLX_HIP(hipEventRecord(done, lanes_[0].s));
    }
    hipStream_t lane_stream(int i) { ensure(i + 1); return lanes_[(size_t)i].s; }
    // one event per active lane, recorded behind everything issued so far (the caller destroys them)
    void mark(std::vector<hipEvent_t>& evs) {
        for (auto& l : lanes_) {
            if (!l.s) continue;
            hipEvent_t e;
            CBLX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            evs.push_back(e);
            CBLX_HIP(hipEventRecord(e, l.s));
        }
    }
    // device -> host: drain(src_pinned, off, n) consumes bytes [off, off+n). Complete on return.
    template <typename F> void d2h(const void* d_src, size_t bytes, F&& drain) {
        // software pipeline per lane: issue chunk j, then drain chunk j-1 while j is in flight
        const size_t nchunk = (bytes + SLOT - 1) / SLOT;
        const int L = lanes_for(bytes);
        ensure(L);
        auto work = [&](int li) {
            CBLX_HIP(hipSetDevice(device_));
            Lane& l = lanes_[li];
            for (int b = 0; b < 2; ++b)
                if (l.busy[b]) { CBLX_HIP(hipEventSynchronize(l.ev[b])); l.busy[b] = false; }  // slots still feeding an h2d
            size_t prev_off = 0, prev_n = 0;
            int k = 0;
            bool have_prev = false;
            for (size_t ci = (size_t)li; ci < nchunk; ci += (size_t)L) {
                const size_t off = ci * SLOT, n = std::min(SLOT, bytes - off);
                CBLX_HIP(hipMemcpyAsync(l.slot[k], (const u8*)d_src + off, n, hipMemcpyDeviceToHost, l.s));
                CBLX_HIP(hipEventRecord(l.ev[k], l.s));
                if (have_prev) {
                    CBLX_HIP(hipEventSynchronize(l.ev[k ^ 1]));
                    drain(l.slot[k ^ 1], prev_off, prev_n);
                }
                prev_off = off; prev_n = n; have_prev = true;
                k ^= 1;
            }
            if (have_prev) {
                CBLX_HIP(hipEventSynchronize(l.ev[k ^ 1]));
                drain(l.slot[k ^ 1], prev_off, prev_n);
            }
        };
        fan_out(L, work);
    }
    void d2h_copy(void* h_dst, const void* d_src, size_t bytes) {
        if (bytes >= PARALLEL_MIN && is_pinned(h_dst) && is_pinned((const u8*)h_dst + bytes - 1)) {
            // the caller's buffer is pinned: DMA straight into it, split over a few of the lanes' streams (complete on return)
            const int L = std::min(4, lanes_for(bytes));
            ensure(L);
            const size_t per = ((bytes + (size_t)L - 1) / (size_t)L + 4095) & ~(size_t)4095;
            for (int i = 0; i < L; ++i) {
                const size_t off = (size_t)i * per;
                if (off >= bytes) break;
                CBLX_HIP(hipMemcpyAsync((u8*)h_dst + off, (const u8*)d_src + off, std::min(per, bytes - off), hipMemcpyDeviceToHost, lanes_[(size_t)i].s));
            }
            for (int i = 0; i < L; ++i) CBLX_HIP(hipStreamSynchronize(lanes_[(size_t)i].s));
            return;
        }
        d2h(d_src, bytes, [&](const u8* src, size_t off, size_t n) { std::memcpy((u8*)h_dst + off, src, n); });
    }
    // Lanes lent to producers that generate bytes at their own pace (file parsers): with_lanes(T, work) runs work(t) on T
    // threads; thread t may open LaneWriters on lane t only. A LaneWriter is an append-only stream into device memory:
    // put() copies into the lane's pinned slots and DMAs every full slot to d_dst + (bytes put so far).
    static int max_parallel() {  // producers are CPU-bound (parsing): more lanes than a plain copy needs to fill the link
        static const int v = [] {
            const char* e = std::getenv("CBLX_PARSE_THREADS");  // override (tests, tuning)
            const int x = e ? std::atoi(e) : 0;
            if (x > 0) return std::min(x, 64);
            const unsigned hc = std::thread::hardware_concurrency();
            return (int)std::max(1u, std::min(CBLX_PARSE_THREADS_DEFAULT, hc ? hc / 2 : 2u));
        }();
        return v;
    }
    template <typename W> void with_lanes(int T, W&& work) {
        ensure(T);
        fan_out(T, [&](int t) { CBLX_HIP(hipSetDevice(device_)); work(t); });
    }
    class LaneWriter {
    public:
        LaneWriter(Xfer& x, int lane, u8* d_dst) : l_(x.lanes_[(size_t)lane]), d_dst_(d_dst) {
            for (int b = 0; b < 2; ++b)
                if (l_.busy[b]) { CBLX_HIP(hipEventSynchronize(l_.ev[b])); l_.busy[b] = false; }
        }
        void put(const void* src, size_t n) {
            const u8* p = (const u8*)src;
            while (n) {
                const size_t m = std::min(n, SLOT - fill_);
                std::memcpy(l_.slot[k_] + fill_, p, m);
                fill_ += m; p += m; n -= m;
                if (fill_ == SLOT) issue();
            }
        }
        void finish() { issue(); }
        u64 written() const { return done_ + fill_; }
    private:
        void issue() {
            if (fill_ == 0) return;
            CBLX_HIP(hipMemcpyAsync(d_dst_ + done_, l_.slot[k_], fill_, hipMemcpyHostToDevice, l_.s));
            CBLX_HIP(hipEventRecord(l_.ev[k_], l_.s));
            l_.busy[k_] = true;
            done_ += fill_;
            fill_ = 0;
            k_ ^= 1;
            if (l_.busy[k_]) { CBLX_HIP(hipEventSynchronize(l_.ev[k_])); l_.busy[k_] = false; }
        }
        Lane& l_;
        u8* d_dst_;
        int k_ = 0;
        size_t fill_ = 0;
        u64 done_ = 0;
    };

    // all DMA issued by h2d() has landed
    void sync() {
        for (auto& l : lanes_) if (l.s) CBLX_HIP(hipStreamSynchronize(l.s));
    }

private:
    int device_;
    std::vector<Lane> lanes_;

    static int max_lanes() {
        unsigned hc = std::thread::hardware_concurrency();
        return (int)std::max(1u, std::min(8u, hc ? hc / 2 : 2u));
    }
    int lanes_for(size_t bytes) const {
        if (bytes < PARALLEL_MIN) return 1;
        return (int)std::min<size_t>((size_t)max_lanes(), (bytes + SLOT - 1) / SLOT);
    }
    void ensure(int L) {
        if ((int)lanes_.size() < L) lanes_.resize(L);
        for (int i = 0; i < L; ++i) {
            Lane& l = lanes_[i];
            if (l.s) continue;
            CBLX_HIP(hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking));
            for (int k = 0; k < 2; ++k) {
                CBLX_HIP(hipHostMalloc((void**)&l.slot[k], SLOT, hipHostMallocDefault));
                CBLX_HIP(hipEventCreateWithFlags(&l.ev[k], hipEventDisableTiming));
            }
        }
    }
    template <typename W> void fan_out(int L, W&& work) {
        if (L == 1) { work(0); return; }
        std::vector<std::thread> th;
        std::vector<std::exception_ptr> errs((size_t)L);
        for (int i = 1; i < L; ++i)
            th.emplace_back([&, i] { try { work(i); } catch (...) { errs[(size_t)i] = std::current_exception(); } });
        try { work(0); } catch (...) { errs[0] = std::current_exception(); }
        for (auto& t : th) t.join();
        for (auto& e : errs) if (e) std::rethrow_exception(e);
    }
    template <typename Issue> void run(size_t bytes, Issue&& issue) {
        if (bytes == 0) return;
        const size_t nchunk = (bytes + SLOT - 1) / SLOT;
        const int L = lanes_for(bytes);
        ensure(L);
        auto work = [&](int li) {
            CBLX_HIP(hipSetDevice(device_));
            Lane& l = lanes_[li];
            int k = 0;
            for (size_t ci = (size_t)li; ci < nchunk; ci += (size_t)L) {
                const size_t off = ci * SLOT, n = std::min(SLOT, bytes - off);
                if (l.busy[k]) { CBLX_HIP(hipEventSynchronize(l.ev[k])); l.busy[k] = false; }  // slot's previous DMA done
                issue(l, k, off, n);
                CBLX_HIP(hipEventRecord(l.ev[k], l.s));
                l.busy[k] = true;
                k ^= 1;
            }
        };
        fan_out(L, work);
    }
};

}  // namespace cblx
