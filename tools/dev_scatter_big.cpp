// Dev harness (not shipped): a scatter pass over 8192-record tiles (512 threads x 16 records, ranked once, staged and
// written in two halves through the same 32 KB of LDS): runs of 32 records instead of 16 = half as many partial lines.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cbl_amd/csrc tools/dev_scatter_big.cpp -o tools/dev_scatter_big.bin
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels_bucket.hpp"
using namespace cblx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
__global__ void k_gen(u64* lo, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 z = i * 0x9E3779B97F4A7C15ull + 12345; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    lo[i] = z;
}
#ifndef BIG_OCC
#define BIG_OCC 6
#endif
static const int BIG_ITEMS = 16, BIG_TILE = RDX_THREADS * BIG_ITEMS, BIG_HALF = BIG_TILE / 2;
template <typename DigitFn>
__global__ __launch_bounds__(RDX_THREADS, BIG_OCC) void k_scatter_big(const u64* __restrict__ lo, u32 nbig, u64 n, DigitFn dfn, const u32* __restrict__ colpre,
                                                                      const u32* __restrict__ adj, u64* __restrict__ out_lo, DigitBits next_dfn, u8* __restrict__ out_next) {
    __shared__ u64 s_lo[BIG_HALF];
    __shared__ u8 s_dig[BIG_HALF];
    u32* s_wcnt = reinterpret_cast<u32*>(s_lo);
    __shared__ u32 s_dbase[256];
    __shared__ u64 s_gbase[256];
    __shared__ u32 s_scan[RDX_THREADS / 64 + 1];
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const u32 tile = xcd_tile(blockIdx.x, nbig);
    if (tile >= nbig) return;
    const u64 tbase = (u64)tile * BIG_TILE;
    const u32 n_tile = (u32)((n - tbase) < (u64)BIG_TILE ? (n - tbase) : (u64)BIG_TILE);
    u64 klo[BIG_ITEMS];
    u32 digit[BIG_ITEMS];
    const u64* __restrict__ lo_t = lo + tbase;
#pragma unroll
    for (int j = 0; j < BIG_ITEMS; ++j) {
        const u32 e = w * (64 * BIG_ITEMS) + j * 64 + lane;
        const bool valid = e < n_tile;
        klo[j] = lo_t[valid ? e : 0u];
        digit[j] = valid ? dfn(klo[j], 0) : 255u;
    }
    tile_rank_packed<RDX_THREADS, BIG_ITEMS>(digit, s_wcnt, s_dbase, s_scan, BIG_ITEMS);
    if (tid < 256) s_gbase[tid] = (u64)adj[tid] + colpre[(u64)(2 * tile) * 256 + tid] - s_dbase[tid];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int j = 0; j < BIG_ITEMS; ++j) {
            const u32 pj = digit[j] & 0xFFFFu;
            if ((pj >> 12) == (u32)h) { s_lo[pj & (BIG_HALF - 1)] = klo[j]; s_dig[pj & (BIG_HALF - 1)] = (u8)(digit[j] >> 16); }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < BIG_ITEMS / 2; ++j) {
            const u32 s = j * RDX_THREADS + tid, gs = h * BIG_HALF + s;
            if (gs < n_tile) {
                const u64 a = s_lo[s];
                const u32 d = s_dig[s];
                const u64 dst = s_gbase[d] + gs;
                out_lo[dst] = a;
                if (out_next) out_next[dst] = (u8)next_dfn(a, 0);
            }
        }
        __syncthreads();
    }
}
template <typename T> T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); return p; }
int main(int argc, char** argv) {
    u64 n = argc > 1 ? strtoull(argv[1], 0, 10) : 1200000000ull;
    u64 *lo = dalloc<u64>(n + 8), *ref = dalloc<u64>(n + 8), *out = dalloc<u64>(n + 8);
    u8 *dig = dalloc<u8>(n + 64), *dig2 = dalloc<u8>(n + 64);
    hipLaunchKernelGGL(k_gen, dim3((n + 255) / 256), dim3(256), 0, 0, lo, n);
    const u32 nt = (u32)((n + RDX_TILE - 1) / RDX_TILE), nbig = (u32)((n + BIG_TILE - 1) / BIG_TILE);
    u32 *counts = dalloc<u32>((size_t)256 * (nt + 2)), *colpre = dalloc<u32>((size_t)256 * (nt + 2));
    const u32 nch = (nt + COLSCAN_ROWS - 1) / COLSCAN_ROWS;
    u32 *chunk = dalloc<u32>((size_t)nch * 256), *coltot = dalloc<u32>(256), *adj = dalloc<u32>(256);
    const DigitBits d{44, 8}, nd{52, 8};
    const TileView tv{nullptr, nullptr, nullptr, nullptr, nt, n};
    hipLaunchKernelGGL((k_radix_hist<NoHi, DigitBits>), dim3(xcd_grid(nt)), dim3(RDX_THREADS), 0, 0, lo, (const NoHi*)nullptr, tv, d, counts);
    hipLaunchKernelGGL(k_colscan_reduce, dim3(nch), dim3(256), 0, 0, counts, (const u32*)nullptr, nt, chunk);
    hipLaunchKernelGGL(k_colscan_spine, dim3(1), dim3(256), 0, 0, chunk, nch, coltot);
    hipLaunchKernelGGL(k_colscan_apply, dim3(nch), dim3(256), 0, 0, counts, (const u32*)nullptr, nt, chunk, colpre);
    hipLaunchKernelGGL(k_seg_adjust, dim3(1), dim3(256), 0, 0, colpre, coltot, (const u32*)nullptr, (const u32*)nullptr, (const u32*)nullptr, nt, 1u, adj, (u32*)nullptr);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double bytes, auto fn) {
        float best = 1e9;
        for (int r = 0; r < 4; ++r) { CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; }
        CK(hipGetLastError());
        printf("%-46s %7.3f ms  %.0f GB/s\n", name, best, n * bytes / best / 1e6);
    };
    for (int side = 0; side < 2; ++side) {
        timeit(side ? "4096-record tiles (shipping kernel), side" : "4096-record tiles (shipping kernel)", 16.0 + side, [&] {
            hipLaunchKernelGGL((k_radix_scatter<NoHi, NoHi, DigitBits>), dim3(xcd_grid(nt)), dim3(RDX_THREADS), 0, 0, lo, (const NoHi*)nullptr, tv, d, colpre, adj, ref, (NoHi*)nullptr,
                               side ? nd : DigitBits{0, 0}, side ? dig : (u8*)nullptr); });
        timeit(side ? "8192-record tiles, two staging halves, side" : "8192-record tiles, two staging halves", 16.0 + side, [&] {
            hipLaunchKernelGGL((k_scatter_big<DigitBits>), dim3(xcd_grid(nbig)), dim3(RDX_THREADS), 0, 0, lo, nbig, n, d, colpre, adj, out, side ? nd : DigitBits{0, 0},
                               side ? dig2 : (u8*)nullptr); });
    }
    std::vector<u64> a(1 << 20), b(1 << 20);
    u64 bad = 0;
    for (u64 off : {(u64)0, n / 3, n - (1 << 20)}) {
        CK(hipMemcpy(a.data(), ref + off, a.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), out + off, b.size() * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
    }
    printf("output differs from the shipping kernel in %llu of 3 M sampled records\n", (unsigned long long)bad);
    return 0;
}
