// Dev harness (not shipped): k_bucket_msd alone, every length class, on the REAL buckets of a K = 31 build (necklace clusters and all):
// words from libcblx's KRN-1 (cblx_seq_words_device), a stable sort by prefix (rocPRIM, setup only), run lengths classified as k_classify
// does, then the instantiation of every class timed over pristine copies of the arena. The kernels come from THIS translation unit
// (kernels_bucket.hpp compiled with the -D switches of the variant under test), libcblx.so only supplies the words. Prints per class:
// buckets, words, ms, and a checksum of (counts, kinds, arena) that must not change between variants.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cbl_amd/csrc -I include tools/dev_msd_bench.cpp -o tools/dev_msd_bench.bin -L cbl_amd -lcblx -Wl,-rpath,'$ORIGIN/../cbl_amd'
//   tools/dev_msd_bench.bin [reads = 10000000] [prefix_bits = 24] [reps = 5]
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "cblx.h"
#include "kernels_bucket.hpp"
#ifdef MSD_BENCH_HASHED
#include "bucket_hashed_experiment.hpp"
#endif
using namespace cblx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define CB(x) do { int r_ = (x); if (r_) { printf("cblx error %d at %s:%d\n", r_, __FILE__, __LINE__); exit(1); } } while (0)
template <typename T> T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T))); return p; }

__global__ void k_gen_bases(u8* b, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 z = (i >> 5) * 0x9E3779B97F4A7C15ull + 42;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    b[i] = "ACGT"[(z >> (2 * (i & 31))) & 3];
}
__global__ void k_gen_offsets(u64* o, u64 n, u64 L) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) o[i] = i * L;
}
__global__ void k_split_words(const u64* lo, const u8* hi, u64 n, u32 SB, u32 PB, u32* prefix) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    prefix[i] = get_bits(lo[i], hi ? (u64)hi[i] : 0ull, SB, PB);
}
// run heads of the sorted prefixes -> one descriptor per run in the list of its length class (k_classify's classes)
__global__ void k_runs(const u32* prefix, u64 n, u32* run_start, u32* nruns) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == 0 || prefix[i] != prefix[i - 1]) run_start[atomicAdd(nruns, 1u)] = (u32)i;
}
__global__ void k_lists(const u32* run_start, u32 nruns, u64 n, BDesc* lists, u32* list_n, u32 cap) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nruns) return;
    const u32 s = run_start[r], c = (r + 1 < nruns ? run_start[r + 1] : (u32)n) - s;
    int cls = -1;
    if (c <= SMALL_MAX) cls = -1;
    else if (c <= 16 * MED_ITEMS) cls = 0;
    else if (c <= 64 * MED_ITEMS) cls = 1;
    else if (c <= 128 * MED_ITEMS) cls = 2;
    else if (c <= 256 * MED_ITEMS) cls = 3;
    else if (c <= 512 * MED_ITEMS) cls = 4;
    if (cls < 0) return;
    lists[(u64)cls * cap + atomicAdd(&list_n[cls], 1u)] = BDesc{s, c, r};
#ifdef MSD_BENCH_SPLIT  // half classes: 5 = (32, 64], 6 = (64, 128], 7 = (128, 256], 8 = (256, 512]
    const int h = c <= 64 ? 5 : c <= 128 ? 6 : c <= 256 ? 7 : c <= 512 ? 8 : -1;
    if (h >= 0) lists[(u64)h * cap + atomicAdd(&list_n[h], 1u)] = BDesc{s, c, r};
#endif
}
__global__ void k_sum64(const u64* v, u64 n, unsigned long long* out) {
    u64 s = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) s += v[i] * (2 * i + 1);
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)s);
}
__global__ void k_sum_res(const BDesc* list, u32 n, const u32* cnt, const u8* kind, const u64* lo, unsigned long long* out) {
    // the distinct words of every listed bucket, position-weighted inside the bucket, + count and kind
    const u32 b = blockIdx.x;
    if (b >= n) return;
    const BDesc d = list[b];
    const u32 c = cnt[d.r];
    u64 s = 0;
    for (u32 i = threadIdx.x; i < c; i += blockDim.x) s += lo[d.start + i] * (2ull * i + 1);
    if (threadIdx.x == 0) s += (u64)c * 0x9E3779B97F4A7C15ull + kind[d.r];
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, (unsigned long long)s);
}

int main(int argc, char** argv) {
    const u64 reads = argc > 1 ? strtoull(argv[1], 0, 10) : 10000000ull;
    const u32 PB = argc > 2 ? atoi(argv[2]) : 24;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    const u32 K = 31, L = 150;
    cblx_params prm{K, PB, 0, -1, 0, 0};
    cblx_ctx* ctx;
    CB(cblx_create(&prm, &ctx));
    cblx_consts cc;
    CB(cblx_get_consts(ctx, &cc));
    const u32 SB = cc.suffix_bits;
    const u64 nb = reads * L, n = reads * (L - K + 1);
    u8* bases = dalloc<u8>(nb + 64);
    u64* offs = dalloc<u64>(reads + 1);
    hipLaunchKernelGGL(k_gen_bases, dim3((nb + 255) / 256), dim3(256), 0, 0, bases, nb);
    hipLaunchKernelGGL(k_gen_offsets, dim3((reads + 256) / 256), dim3(256), 0, 0, offs, reads, (u64)L);
    u64 *w_lo = dalloc<u64>(n + 8), *s_lo = dalloc<u64>(n + 8);
    u8* w_hi = dalloc<u8>(n + 8);
    u64 nw = 0;
    CB(cblx_seq_words_device(ctx, bases, offs, reads, w_lo, w_hi, n, &nw));
    if (nw != n) { printf("words %llu != %llu\n", (unsigned long long)nw, (unsigned long long)n); return 1; }
    CK(hipFree(bases));
    u32 *pfx = dalloc<u32>(n + 8), *pfx_s = dalloc<u32>(n + 8);
    hipLaunchKernelGGL(k_split_words, dim3((n + 255) / 256), dim3(256), 0, 0, w_lo, w_hi, n, SB, PB, pfx);
    CK(hipFree(w_hi));
    {   // stable sort by prefix: values = the low words (what the arena holds behind the partition passes)
        size_t tb = 0;
        CK(rocprim::radix_sort_pairs(nullptr, tb, pfx, pfx_s, w_lo, s_lo, n, 0, PB));
        void* tmp;
        CK(hipMalloc(&tmp, tb));
        CK(rocprim::radix_sort_pairs(tmp, tb, pfx, pfx_s, w_lo, s_lo, n, 0, PB));
        CK(hipDeviceSynchronize());
        CK(hipFree(tmp));
    }
    CK(hipFree(pfx));
    u32* run_start = dalloc<u32>(n / 8 + (1u << 22));
    u32* d_nruns = dalloc<u32>(1);
    CK(hipMemset(d_nruns, 0, 4));
    hipLaunchKernelGGL(k_runs, dim3((n + 255) / 256), dim3(256), 0, 0, pfx_s, n, run_start, d_nruns);
    u32 nruns;
    CK(hipMemcpy(&nruns, d_nruns, 4, hipMemcpyDeviceToHost));
    {   // the atomics appended the run heads in no order: sort them
        u32* rs2 = dalloc<u32>(nruns);
        size_t tb = 0;
        CK(rocprim::radix_sort_keys(nullptr, tb, run_start, rs2, nruns));
        void* tmp;
        CK(hipMalloc(&tmp, tb));
        CK(rocprim::radix_sort_keys(tmp, tb, run_start, rs2, nruns));
        CK(hipMemcpy(run_start, rs2, (size_t)nruns * 4, hipMemcpyDeviceToDevice));
        CK(hipFree(tmp)); CK(hipFree(rs2));
    }
    CK(hipFree(pfx_s));
    constexpr int NL = 9;
    BDesc* lists = dalloc<BDesc>((size_t)NL * nruns);
    u32* list_n = dalloc<u32>(16);
    CK(hipMemset(list_n, 0, 64));
    hipLaunchKernelGGL(k_lists, dim3((nruns + 255) / 256), dim3(256), 0, 0, run_start, nruns, n, lists, list_n, nruns);
    u32 ln[NL];
    CK(hipMemcpy(ln, list_n, 4 * NL, hipMemcpyDeviceToHost));
    {   // bucket order inside every class list (k_classify appends workgroup by workgroup: nearly ascending)
        std::vector<BDesc> h((size_t)NL * nruns);
        CK(hipMemcpy(h.data(), lists, h.size() * sizeof(BDesc), hipMemcpyDeviceToHost));
        for (int k = 0; k < NL; ++k) std::sort(h.begin() + (size_t)k * nruns, h.begin() + (size_t)k * nruns + ln[k], [](const BDesc& a, const BDesc& b) { return a.start < b.start; });
        CK(hipMemcpy(lists, h.data(), h.size() * sizeof(BDesc), hipMemcpyHostToDevice));
    }
    u64* arena = dalloc<u64>(n + 8);
    u32* cnt = dalloc<u32>(nruns + 1);
    u8* kind = dalloc<u8>(nruns + 1);
    u8* bail = dalloc<u8>(nruns + 8);
    u32* bail_any = dalloc<u32>(8);
    unsigned long long* d_sum = dalloc<unsigned long long>(1);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("K=%u PB=%u SB=%u reads=%llu words=%llu runs=%u  packed=%d\n", K, PB, SB, (unsigned long long)reads, (unsigned long long)n, nruns, (int)(SB + PK_BITS <= 64));
    double total = 0;
    auto run_class = [&](int k, auto thr, auto cap, auto pk) {
        constexpr int T = decltype(thr)::value, CAPV = decltype(cap)::value;
        constexpr bool PKD = decltype(pk)::value;
        if (!ln[k]) return;
        float best = 1e30f, sum = 0;
        unsigned long long words = 0, chk = 0;
        for (int rep = 0; rep < reps + 1; ++rep) {
            CK(hipMemcpy(arena, s_lo, n * 8, hipMemcpyDeviceToDevice));
            CK(hipMemset(bail, 0, nruns + 8));
            CK(hipMemset(bail_any, 0, 32));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_bucket_msd<T, CAPV, PKD, false, u8>), dim3(ln[k]), dim3(T), 0, 0, lists + (size_t)k * nruns, list_n + k, arena, (u8*)nullptr, SB, cnt, kind,
                               (BDesc*)nullptr, (u32*)nullptr, MergeArgs{}, bail, bail_any);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) { best = std::min(best, ms); sum += ms; }
        }
        CK(hipMemset(d_sum, 0, 8));
        hipLaunchKernelGGL(k_sum_res, dim3(ln[k]), dim3(256), 0, 0, lists + (size_t)k * nruns, ln[k], cnt, kind, arena, d_sum);
        CK(hipMemcpy(&chk, d_sum, 8, hipMemcpyDeviceToHost));
        u32 nbail = 0;
        {
            std::vector<u8> hb(ln[k]);
            CK(hipMemcpy(hb.data(), bail, ln[k], hipMemcpyDeviceToHost));
            for (u8 x : hb) nbail += x;
            std::vector<BDesc> h(ln[k]);
            CK(hipMemcpy(h.data(), lists + (size_t)k * nruns, (size_t)ln[k] * sizeof(BDesc), hipMemcpyDeviceToHost));
            for (const BDesc& d : h) words += d.c;
        }
        printf("class %d <%3d,%4d>: %8u buckets %11llu words  best %7.3f ms  avg %7.3f ms  %6.1f ps/word  gave up %u  chk %016llx\n", k, T, CAPV, ln[k], words, best, sum / reps,
               best * 1e9 / (double)words, nbail, chk);
        total += best;
    };
    auto all = [&](auto pk) {
        run_class(0, std::integral_constant<int, 64>(), std::integral_constant<int, 128>(), pk);
        run_class(1, std::integral_constant<int, 64>(), std::integral_constant<int, 512>(), pk);
        run_class(2, std::integral_constant<int, 128>(), std::integral_constant<int, 1024>(), pk);
        run_class(3, std::integral_constant<int, 256>(), std::integral_constant<int, 2048>(), pk);
        run_class(4, std::integral_constant<int, 512>(), std::integral_constant<int, 4096>(), pk);
    };
    if (SB + PK_BITS <= 64) all(std::true_type()); else all(std::false_type());
    printf("total best %.3f ms\n", total);
#ifdef MSD_BENCH_SPLIT
    if (SB + PK_BITS <= 64) {
        run_class(5, std::integral_constant<int, 64>(), std::integral_constant<int, 64>(), std::true_type());
        run_class(6, std::integral_constant<int, 64>(), std::integral_constant<int, 128>(), std::true_type());
        run_class(7, std::integral_constant<int, 64>(), std::integral_constant<int, 256>(), std::true_type());
        run_class(8, std::integral_constant<int, 64>(), std::integral_constant<int, 512>(), std::true_type());
        run_class(5, std::integral_constant<int, 64>(), std::integral_constant<int, 128>(), std::true_type());
        run_class(7, std::integral_constant<int, 64>(), std::integral_constant<int, 512>(), std::true_type());
        run_class(7, std::integral_constant<int, 128>(), std::integral_constant<int, 256>(), std::true_type());
    }
#endif
    // round 6: the sorted classes through k_bucket_sorted (a lane walks the span of its eight slots once)
    auto run_sorted = [&](int k, auto thr, auto cap) {
        constexpr int T = decltype(thr)::value, CAPV = decltype(cap)::value;
        if (!ln[k] || SB + PK_BITS > 64) return;
        float best = 1e30f, sum = 0;
        unsigned long long chk = 0;
        for (int rep = 0; rep < reps + 1; ++rep) {
            CK(hipMemcpy(arena, s_lo, n * 8, hipMemcpyDeviceToDevice));
            CK(hipMemset(bail, 0, nruns + 8));
            CK(hipMemset(bail_any, 0, 32));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
#ifdef CBLX_SORTED_STATS
            static unsigned long long* d_st = dalloc<unsigned long long>(8);
            CK(hipMemset(d_st, 0, 64));
            hipLaunchKernelGGL((k_bucket_sorted<T, CAPV, false, u8>), dim3(ln[k]), dim3(T), 0, 0, lists + (size_t)k * nruns, list_n + k, arena, (u8*)nullptr, SB, cnt, kind, (BDesc*)nullptr, (u32*)d_st, bail, bail_any);
            if (rep == reps) { unsigned long long h[4]; CK(hipMemcpy(h, d_st, 32, hipMemcpyDeviceToHost));
                printf("  span: mean per owning lane %.1f, mean of the wave maxima %.1f (%llu waves)\n", (double)h[0] / h[2], (double)h[1] / h[3], h[3]); }
#else
            hipLaunchKernelGGL((k_bucket_sorted<T, CAPV, false, u8>), dim3(ln[k]), dim3(T), 0, 0, lists + (size_t)k * nruns, list_n + k, arena, (u8*)nullptr, SB, cnt, kind, (BDesc*)nullptr, (u32*)nullptr, bail, bail_any);
#endif
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) { best = std::min(best, ms); sum += ms; }
        }
        CK(hipMemset(d_sum, 0, 8));
        hipLaunchKernelGGL(k_sum_res, dim3(ln[k]), dim3(256), 0, 0, lists + (size_t)k * nruns, ln[k], cnt, kind, arena, d_sum);
        CK(hipMemcpy(&chk, d_sum, 8, hipMemcpyDeviceToHost));
        u32 nbail = 0;
        std::vector<u8> hb(ln[k]);
        CK(hipMemcpy(hb.data(), bail, ln[k], hipMemcpyDeviceToHost));
        for (u8 x : hb) nbail += x;
        printf("sorted %d <%3d,%4d>: %8u buckets  best %7.3f ms  avg %7.3f ms  gave up %u  chk %016llx\n", k, T, CAPV, ln[k], best, sum / reps, nbail, chk);
    };
    run_sorted(3, std::integral_constant<int, 256>(), std::integral_constant<int, 2048>());
    run_sorted(4, std::integral_constant<int, 512>(), std::integral_constant<int, 4096>());
#ifdef MSD_BENCH_HASHED
    // round 6: the short classes through k_bucket_hashed (a persistent grid, the next bucket's words in flight): tools/bucket_hashed_experiment.hpp
    auto run_hashed = [&](int k, auto thr, auto cap, u32 wgs_per_cu) {
        constexpr int T = decltype(thr)::value, CAPV = decltype(cap)::value;
        if (!ln[k] || SB + PK_BITS > 64) return;
        const u32 grid = std::min<u32>(ln[k], 256u * wgs_per_cu);
        float best = 1e30f, sum = 0;
        unsigned long long chk = 0;
        for (int rep = 0; rep < reps + 1; ++rep) {
            CK(hipMemcpy(arena, s_lo, n * 8, hipMemcpyDeviceToDevice));
            CK(hipMemset(bail, 0, nruns + 8));
            CK(hipMemset(bail_any, 0, 32));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_bucket_hashed<T, CAPV, u8>), dim3(grid), dim3(T), 0, 0, lists + (size_t)k * nruns, list_n + k, arena, SB, cnt, kind, bail, bail_any);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) { best = std::min(best, ms); sum += ms; }
        }
        CK(hipMemset(d_sum, 0, 8));
        hipLaunchKernelGGL(k_sum_res, dim3(ln[k]), dim3(256), 0, 0, lists + (size_t)k * nruns, ln[k], cnt, kind, arena, d_sum);
        CK(hipMemcpy(&chk, d_sum, 8, hipMemcpyDeviceToHost));
        u32 nbail = 0;
        std::vector<u8> hb(ln[k]);
        CK(hipMemcpy(hb.data(), bail, ln[k], hipMemcpyDeviceToHost));
        for (u8 x : hb) nbail += x;
        printf("hashed %d <%3d,%4d> %3u wg/cu: %8u buckets  best %7.3f ms  avg %7.3f ms  gave up %u  chk %016llx\n", k, T, CAPV, wgs_per_cu, ln[k], best, sum / reps, nbail, chk);
    };
    for (u32 f : {16u, 32u, 64u}) run_hashed(0, std::integral_constant<int, 64>(), std::integral_constant<int, 128>(), f);
    for (u32 f : {16u, 32u, 64u}) run_hashed(1, std::integral_constant<int, 64>(), std::integral_constant<int, 512>(), f);
    for (u32 f : {8u, 16u, 32u}) run_hashed(2, std::integral_constant<int, 128>(), std::integral_constant<int, 1024>(), f);
#endif
#ifdef MSD_BENCH_SHAPES
    run_sorted(3, std::integral_constant<int, 512>(), std::integral_constant<int, 2048>());
    run_sorted(3, std::integral_constant<int, 128>(), std::integral_constant<int, 2048>());
    run_sorted(4, std::integral_constant<int, 1024>(), std::integral_constant<int, 4096>());
    run_sorted(4, std::integral_constant<int, 256>(), std::integral_constant<int, 4096>());
    run_sorted(2, std::integral_constant<int, 128>(), std::integral_constant<int, 1024>());
    run_sorted(2, std::integral_constant<int, 256>(), std::integral_constant<int, 1024>());
#endif
    cblx_destroy(ctx);
    return 0;
}
