// Dev harness (not shipped): times the radix partition kernels in isolation on synthetic records.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cbl_amd/csrc tools/dev_radix_bench.cpp -o /tmp/dev_radix_bench
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels_bucket.hpp"
#include "onesweep_experiment.hpp"
using namespace cblx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void k_gen(u64* lo, u8* hi, u64 n, u32 skew) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 z = i * 0x9E3779B97F4A7C15ull + 12345; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    u64 y = (i + 77) * 0x9E3779B97F4A7C15ull; y = (y ^ (y >> 30)) * 0xBF58476D1CE4E5B9ull; y ^= y >> 29;
    // 68-bit word: prefix 24 bits skewed toward small values: min of `skew` uniforms approximated by shifting
    u32 p = (u32)(z >> 40);
    for (u32 k = 1; k < skew; ++k) { y = y * 6364136223846793005ull + 1442695040888963407ull; u32 q = (u32)(y >> 40); p = q < p ? q : p; }
    u64 sfx = z & ((1ull << 44) - 1);
    lo[i] = ((u64)(p & 0xFFFFF) << 44) | sfx;
    hi[i] = (u8)(p >> 20);
}
template <typename T> T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); return p; }

int main(int argc, char** argv) {
    u64 n = argc > 1 ? strtoull(argv[1], 0, 10) : 240000000ull;
    u32 skew = argc > 2 ? atoi(argv[2]) : 62;
    u32 dbgmask = argc > 3 ? atoi(argv[3]) : 0;
    const u32 SB = 44, PB = 24;
    u64 *lo = dalloc<u64>(n + 8), *lo2 = dalloc<u64>(n + 8), *lo3 = dalloc<u64>(n + 8);
    u8 *hi = dalloc<u8>(n + 8), *hi2 = dalloc<u8>(n + 8), *hi3 = dalloc<u8>(n + 8);
    hipLaunchKernelGGL(k_gen, dim3((n + 255) / 256), dim3(256), 0, 0, lo, hi, n, skew);
    CK(hipDeviceSynchronize());
    const u32 ntiles = (u32)((n + RDX_TILE - 1) / RDX_TILE);
    u32* counts = dalloc<u32>((size_t)256 * ntiles); u32* offsets = dalloc<u32>((size_t)256 * ntiles);
    u64* sums = dalloc<u64>(((size_t)ntiles / COLSCAN_ROWS + 2) * 256 + 1024);
    unsigned long long* ghist = dalloc<unsigned long long>(MAX_PASSES * 256);
    u32* ctl = dalloc<u32>(MAX_PASSES * 128 + 16);
    u64* status = dalloc<u64>((size_t)ntiles * 256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto fn) { CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("%-28s %8.3f ms  (%.1f GB/s at 18 B/rec)\n", name, ms, n * 18.0 / ms / 1e6); CK(hipGetLastError()); };
    for (int pass = 0; pass < 3; ++pass) {
        DigitBits d{SB + 8 * pass, 8};
        printf("--- digit bits %u..%u\n", SB + 8 * pass, SB + 8 * pass + 7);
        const TileView tv{nullptr, nullptr, nullptr, nullptr, ntiles, n};
        timeit("hist", [&] { hipLaunchKernelGGL((k_radix_hist<u8, DigitBits>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, 0, lo, hi, tv, d, counts); });
                const u32 nch = (ntiles + COLSCAN_ROWS - 1) / COLSCAN_ROWS;
        u32* chunk = (u32*)sums; u32* coltot = chunk + (size_t)nch * 256; u32* adj = coltot + 256;
        timeit("colscan + adjust", [&] {
            hipLaunchKernelGGL(k_colscan_reduce, dim3(nch), dim3(256), 0, 0, counts, (const u32*)nullptr, ntiles, chunk);
            hipLaunchKernelGGL(k_colscan_spine, dim3(1), dim3(256), 0, 0, chunk, nch, coltot);
            hipLaunchKernelGGL(k_colscan_apply, dim3(nch), dim3(256), 0, 0, counts, (const u32*)nullptr, ntiles, chunk, offsets);
            hipLaunchKernelGGL(k_seg_adjust, dim3(1), dim3(256), 0, 0, offsets, coltot, (const u32*)nullptr, (const u32*)nullptr, (const u32*)nullptr, ntiles, 1u, adj); });
        timeit("scatter (2-kernel form)", [&] { hipLaunchKernelGGL((k_radix_scatter<u8, u8, DigitBits>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, 0, lo, hi, tv, d, offsets, adj, lo2, hi2); });
        timeit("scatter u8 in, no hi out", [&] { hipLaunchKernelGGL((k_radix_scatter<u8, NoHi, DigitBits>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, 0, lo, hi, tv, d, offsets, adj, lo3, (NoHi*)nullptr); });
        if (SB + 8 * pass + 8 <= 64) timeit("scatter NoHi (8 B records)", [&] { hipLaunchKernelGGL((k_radix_scatter<NoHi, NoHi, DigitBits>), dim3(xcd_grid(ntiles)), dim3(RDX_THREADS), 0, 0, lo, (const NoHi*)nullptr, tv, d, offsets, adj, lo3, (NoHi*)nullptr); });
        CK(hipMemset(ghist, 0, MAX_PASSES * 256 * 8)); CK(hipMemset(ctl, 0, (MAX_PASSES * 128 + 16) * 4)); CK(hipMemset(status, 0, (size_t)ntiles * 256 * 8));
        timeit("digit_hist (all passes)", [&] { hipLaunchKernelGGL(k_digit_hist<u8>, dim3(2048), dim3(512), 0, 0, lo, hi, n, SB, PB, 3u, ghist); });
        for (u32 dbg : {0u, 1u, 2u, 4u}) {
            if (dbg && !(dbgmask & dbg)) continue;
            CK(hipMemset(ctl, 0, (MAX_PASSES * 128 + 16) * 4)); CK(hipMemset(status, 0, (size_t)ntiles * 256 * 8));
            char nm[64]; snprintf(nm, 64, "onesweep dbg=%u", dbg);
            timeit(nm, [&] { hipLaunchKernelGGL((k_onesweep<u8, DigitBits>), dim3(ntiles), dim3(RDX_THREADS), 0, 0, lo, hi, n, d, ghist + pass * 256, ctl + pass * 128, ntiles, status, (u32)(pass + 1), lo3, hi3, ctl + MAX_PASSES * 128, dbg); });
            if (dbg == 0) {
                std::vector<u64> a(1 << 20), b(1 << 20); u64 off = n > (1 << 20) ? n / 2 : 0, m = n > (1 << 20) ? (1 << 20) : n;
                CK(hipMemcpy(a.data(), lo2 + off, m * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), lo3 + off, m * 8, hipMemcpyDeviceToHost));
                u64 bad = 0; for (u64 i = 0; i < m; ++i) bad += a[i] != b[i];
                u32 err; CK(hipMemcpy(&err, ctl + MAX_PASSES * 128, 4, hipMemcpyDeviceToHost));
                printf("   onesweep vs 2-kernel: %llu mismatches in %llu sampled, err flag %u\n", (unsigned long long)bad, (unsigned long long)m, err);
            }
        }
        std::swap(lo, lo2); std::swap(hi, hi2);
    }
    return 0;
}
