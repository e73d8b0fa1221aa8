// Dev harness (not shipped): does confining a scatter pass's destination window pay? Same kernels, same bytes; the array
// is cut into "segments" of S tiles and every tile scatters by an 8-bit digit INSIDE its segment. S large = what the LSD
// passes do today (window = a pass-A segment, hundreds of MB); S small = an MSD last pass (window = one (segment, digit)
// group, ~1 MB, L2-resident).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cbl_amd/csrc tools/dev_msd_window.cpp -o tools/dev_msd_window.bin
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
__global__ void k_tables(u32 nt, u32 S, u64 n, u32* t_start, u32* t_count, u16* t_seg, u32* seg_first, u32* seg_start, u32 nseg) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nt) {
        t_start[t] = t * RDX_TILE;
        u64 rem = n - (u64)t * RDX_TILE;
        t_count[t] = rem < RDX_TILE ? (u32)rem : RDX_TILE;
        t_seg[t] = (u16)(t / S);
    }
    if (t <= nseg) {
        u32 f = t * S < nt ? t * S : nt;
        seg_first[t] = f;
        seg_start[t] = (u64)f * RDX_TILE < n ? f * RDX_TILE : (u32)n;
    }
}
template <typename T> T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); return p; }

int main(int argc, char** argv) {
    u64 n = argc > 1 ? strtoull(argv[1], 0, 10) : 1200000000ull;
    u64 *lo = dalloc<u64>(n + 8), *lo2 = dalloc<u64>(n + 8);
    u8* dig = dalloc<u8>(n + 64);
    hipLaunchKernelGGL(k_gen, dim3((n + 255) / 256), dim3(256), 0, 0, lo, n);
    CK(hipDeviceSynchronize());
    const u32 nt = (u32)((n + RDX_TILE - 1) / RDX_TILE);
    u32 *counts = dalloc<u32>((size_t)256 * nt), *colpre = dalloc<u32>((size_t)256 * nt);
    const u32 nch = (nt + COLSCAN_ROWS - 1) / COLSCAN_ROWS;
    u32 *chunk = dalloc<u32>((size_t)nch * 256), *coltot = dalloc<u32>(256), *adj = dalloc<u32>((size_t)65536 * 256);
    u32 *t_start = dalloc<u32>(nt), *t_count = dalloc<u32>(nt), *seg_first = dalloc<u32>(65538), *seg_start = dalloc<u32>(65538);
    u16* t_seg = dalloc<u16>(nt);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // argv[2] = digit width (default 8): narrower digits = longer runs per (tile, bin) = fewer partial lines
    const u32 nb = argc > 2 ? (u32)atoi(argv[2]) : 8u;
    const DigitBits d{44, nb}, nd{52, 8};
    printf("digit width %u bits (%u bins, runs of %u records)\n", nb, 1u << nb, RDX_TILE >> nb);
    for (u32 S : {8192u, 36u}) {
        const u32 nseg = (nt + S - 1) / S;
        if (nseg > 65535) { printf("S=%u: too many segments\n", S); continue; }
        hipLaunchKernelGGL(k_tables, dim3((std::max(nt, nseg + 1) + 255) / 256), dim3(256), 0, 0, nt, S, n, t_start, t_count, t_seg, seg_first, seg_start, nseg);
        const TileView tv{t_start, t_count, t_seg, nullptr, nt, n};
        hipLaunchKernelGGL((k_radix_hist<NoHi, DigitBits>), dim3(xcd_grid(nt)), dim3(RDX_THREADS), 0, 0, lo, (const NoHi*)nullptr, tv, d, counts);
        hipLaunchKernelGGL(k_colscan_reduce, dim3(nch), dim3(256), 0, 0, counts, (const u32*)nullptr, nt, chunk);
        hipLaunchKernelGGL(k_colscan_spine, dim3(1), dim3(256), 0, 0, chunk, nch, coltot);
        hipLaunchKernelGGL(k_colscan_apply, dim3(nch), dim3(256), 0, 0, counts, (const u32*)nullptr, nt, chunk, colpre);
        hipLaunchKernelGGL(k_seg_adjust, dim3(nseg), dim3(256), 0, 0, colpre, coltot, seg_first, seg_start, (const u32*)nullptr, nt, nseg, adj, (u32*)nullptr);
        CK(hipDeviceSynchronize());
        for (int side = 0; side < 2; ++side) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL((k_radix_scatter<NoHi, NoHi, DigitBits>), dim3(xcd_grid(nt)), dim3(RDX_THREADS), 0, 0, lo, (const NoHi*)nullptr, tv, d, colpre, adj, lo2,
                                   (NoHi*)nullptr, side ? nd : DigitBits{0, 0}, side ? dig : (u8*)nullptr);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            CK(hipGetLastError());
            printf("S=%6u tiles/segment (window %8.1f MB, %5u segments)  side=%d  scatter %7.3f ms  %.0f GB/s\n", S, S * RDX_TILE * 8.0 / 1e6, nseg, side, best,
                   n * (16.0 + side) / best / 1e6);
        }
    }
    return 0;
}
