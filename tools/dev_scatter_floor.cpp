// Dev harness (not shipped): what bounds k_radix_scatter? Same grid / tile shape, parts removed.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cbl_amd/csrc tools/dev_scatter_floor.cpp -o tools/dev_scatter_floor.bin
#include <cstdio>
#include <cstdlib>
#include "kernels_bucket.hpp"
using namespace cblx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
__global__ void k_gen(u64* lo, u64 n) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 z = i * 0x9E3779B97F4A7C15ull + 12345; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    lo[i] = z;
}
// V2: plain copy, same tile map, 8 records per thread
__global__ __launch_bounds__(RDX_THREADS, 8) void k_copy(const u64* __restrict__ lo, u64* __restrict__ out, u32 nt, u64 n) {
    const u32 tile = xcd_tile(blockIdx.x, nt);
    if (tile >= nt) return;
    const u64 tbase = (u64)tile * RDX_TILE;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    u64 k[RDX_ITEMS];
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) { const u64 e = tbase + w * (64 * RDX_ITEMS) + j * 64 + lane; k[j] = lo[e < n ? e : 0]; }
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) { const u64 e = tbase + w * (64 * RDX_ITEMS) + j * 64 + lane; if (e < n) out[e] = k[j]; }
}
// V1: load -> LDS (permuted inside the tile by a cheap function) -> barrier -> linear store; LDS as in the real kernel
template <int WG_PER_CU_LDS>
__global__ __launch_bounds__(RDX_THREADS, 8) void k_stage(const u64* __restrict__ lo, u64* __restrict__ out, u32 nt, u64 n) {
    __shared__ u64 s_lo[RDX_TILE];
    __shared__ u8 s_pad[WG_PER_CU_LDS];
    const u32 tile = xcd_tile(blockIdx.x, nt);
    if (tile >= nt) return;
    const u64 tbase = (u64)tile * RDX_TILE;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    u64 k[RDX_ITEMS];
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) { const u64 e = tbase + w * (64 * RDX_ITEMS) + j * 64 + lane; k[j] = lo[e < n ? e : 0]; }
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) { const u32 e = w * (64 * RDX_ITEMS) + j * 64 + lane; s_lo[(e * 17u) & (RDX_TILE - 1)] = k[j]; }
    if (tid == 0) s_pad[0] = 1;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) { const u32 s = j * RDX_THREADS + tid; if (tbase + s < n) out[tbase + s] = s_lo[s] + s_pad[0]; }
}
// V3: the real ranking, linear store
__global__ __launch_bounds__(RDX_THREADS, 8) void k_rank_linear(const u64* __restrict__ lo, u64* __restrict__ out, u32 nt, u64 n) {
    __shared__ u64 s_lo[RDX_TILE];
    __shared__ u8 s_dig[RDX_TILE];
    u32* s_wcnt = reinterpret_cast<u32*>(s_lo);
    __shared__ u32 s_dbase[256];
    __shared__ u32 s_scan[RDX_THREADS / 64 + 1];
    const u32 tile = xcd_tile(blockIdx.x, nt);
    if (tile >= nt) return;
    const u64 tbase = (u64)tile * RDX_TILE;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    u64 k[RDX_ITEMS];
    u32 digit[RDX_ITEMS];
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) { const u64 e = tbase + w * (64 * RDX_ITEMS) + j * 64 + lane; k[j] = lo[e < n ? e : 0]; digit[j] = (u32)(k[j] >> 44) & 255u; }
    tile_rank_packed<RDX_THREADS, RDX_ITEMS>(digit, s_wcnt, s_dbase, s_scan, RDX_ITEMS);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) { const u32 pj = digit[j] & 0xFFFFu; s_lo[pj] = k[j]; s_dig[pj] = (u8)(digit[j] >> 16); }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) { const u32 s = j * RDX_THREADS + tid; if (tbase + s < n) out[tbase + s] = s_lo[s] + s_dig[s]; }
}
// V4: no ranking; 256 runs of 16 records per tile scattered to 256 regions (run r of tile t -> region r, slot t)
template <u32 run, u32 MIS>
__global__ __launch_bounds__(RDX_THREADS, 8) void k_norank_scatter(const u64* __restrict__ lo, u64* __restrict__ out, u32 nt, u64 n) {
    __shared__ u64 s_lo[RDX_TILE];
    const u32 tile = xcd_tile(blockIdx.x, nt);
    if (tile >= nt) return;
    const u64 tbase = (u64)tile * RDX_TILE;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    u64 k[RDX_ITEMS];
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) { const u64 e = tbase + w * (64 * RDX_ITEMS) + j * 64 + lane; k[j] = lo[e < n ? e : 0]; }
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) { const u32 e = w * (64 * RDX_ITEMS) + j * 64 + lane; s_lo[(e * 17u) & (RDX_TILE - 1)] = k[j]; }
    __syncthreads();
    const u32 nruns = RDX_TILE / run;
    const u64 region = (u64)nt * run;  // records per region
#pragma unroll
    for (int j = 0; j < RDX_ITEMS; ++j) {
        const u32 s = j * RDX_THREADS + tid;
        const u32 r = s / run, o = s % run;
        const u64 dst = (u64)r * region + (u64)tile * run + o + MIS;  // MIS: runs not line-aligned, like real ones
        if (dst < n) out[dst] = s_lo[s];
    }
    (void)nruns;
}
int main(int argc, char** argv) {
    u64 n = argc > 1 ? strtoull(argv[1], 0, 10) : 1200000000ull;
    u64 *lo, *lo2; CK(hipMalloc(&lo, (n + 8) * 8)); CK(hipMalloc(&lo2, (n + 8) * 8));
    hipLaunchKernelGGL(k_gen, dim3((n + 255) / 256), dim3(256), 0, 0, lo, n);
    CK(hipDeviceSynchronize());
    const u32 nt = (u32)((n + RDX_TILE - 1) / RDX_TILE);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto fn) {
        float best = 1e9;
        for (int r = 0; r < 4; ++r) { CK(hipEventRecord(e0)); fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; }
        CK(hipGetLastError());
        printf("%-40s %7.3f ms  %.0f GB/s\n", name, best, n * 16.0 / best / 1e6);
    };
    timeit("copy (tile map, 8/thread)", [&] { hipLaunchKernelGGL(k_copy, dim3(xcd_grid(nt)), dim3(RDX_THREADS), 0, 0, lo, lo2, nt, n); });
    timeit("stage through LDS, 33 KB (4 WG/CU)", [&] { hipLaunchKernelGGL(k_stage<1024>, dim3(xcd_grid(nt)), dim3(RDX_THREADS), 0, 0, lo, lo2, nt, n); });
    timeit("stage through LDS, 39 KB (4 WG/CU)", [&] { hipLaunchKernelGGL(k_stage<7168>, dim3(xcd_grid(nt)), dim3(RDX_THREADS), 0, 0, lo, lo2, nt, n); });
    timeit("stage through LDS, 48 KB (3 WG/CU)", [&] { hipLaunchKernelGGL(k_stage<16384>, dim3(xcd_grid(nt)), dim3(RDX_THREADS), 0, 0, lo, lo2, nt, n); });
    timeit("real ranking, linear store", [&] { hipLaunchKernelGGL(k_rank_linear, dim3(xcd_grid(nt)), dim3(RDX_THREADS), 0, 0, lo, lo2, nt, n); });
#define RUNV(R, M) timeit("no ranking, runs of " #R " records, misaligned by " #M, [&] { hipLaunchKernelGGL((k_norank_scatter<R, M>), dim3(xcd_grid(nt)), dim3(RDX_THREADS), 0, 0, lo, lo2, nt, n); });
    RUNV(8, 0) RUNV(8, 3) RUNV(16, 0) RUNV(16, 3) RUNV(16, 8) RUNV(32, 0) RUNV(32, 3) RUNV(64, 0) RUNV(64, 3) RUNV(128, 5) RUNV(512, 5)
    return 0;
}
