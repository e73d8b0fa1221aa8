// dev: issue cost of the 64-bit integer VALU instructions the necklace code leans on (v_lshlrev_b64, v_cmp_*_u64) against
// 32-bit ones (v_alignbit_b32, v_lshlrev_b32, v_and_b32) on gfx950. Every wave runs N iterations of 8 independent chains;
// the table prints cycles per wave-instruction at full occupancy (8 waves per SIMD).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o tools/dev_valu_rate.bin tools/dev_valu_rate.cpp && tools/dev_valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(_e), #e); return 1; } } while (0)
constexpr int N = 4096, CH = 8;
template <int MODE> __global__ __launch_bounds__(256) void k(uint64_t* out, uint64_t seed) {
    uint64_t a[CH];
    uint32_t b[CH], c[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { a[j] = seed + threadIdx.x * 77 + j; b[j] = (uint32_t)a[j]; c[j] = (uint32_t)(a[j] >> 7) | 1; }
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (MODE == 0) { asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[j])); }
            if (MODE == 1) { asm volatile("v_lshrrev_b64 %0, 5, %0" : "+v"(a[j])); }
            if (MODE == 2) { asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(b[j]) : "v"(c[j])); }
            if (MODE == 3) { asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(b[j])); }
            if (MODE == 4) { asm volatile("v_and_b32 %0, %0, %1" : "+v"(b[j]) : "v"(c[j])); }
            if (MODE == 5) { asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a[j]), "v"(a[(j + 1) % CH]), "v"(b[j]), "v"(c[j]) : "vcc"); }
            if (MODE == 6) { asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(b[j]), "v"(c[j]), "v"(b[j]), "v"(c[j]) : "vcc"); }
            if (MODE == 7) { asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(a[j]) : "v"(a[(j + 1) % CH])); }
            if (MODE == 8) { asm volatile("v_ffbh_u32 %0, %1" : "=v"(b[j]) : "v"(c[j])); }
            if (MODE == 9) { asm volatile("v_bfe_u32 %0, %1, 3, 8" : "=v"(b[j]) : "v"(c[j])); }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < CH; ++j) s += a[j] + b[j] + c[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> int run(const char* name, int per_iter) {
    uint64_t* d;
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    CHECK(hipMalloc(&d, (size_t)blocks * 256 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1ull);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 2ull);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per SIMD: 8 waves x N x CH x per_iter; clock from the device
    int clk = 0;
    CHECK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    const double cyc = ms * 1e-3 * clk * 1e3 / (8.0 * N * CH * per_iter);
    printf("%-34s %8.3f ms  %6.2f cycles per wave-instruction (SIMD clock %d MHz)\n", name, ms, cyc, clk / 1000);
    CHECK(hipFree(d));
    return 0;
}
int main() {
    run<0>("v_lshlrev_b64", 1); run<1>("v_lshrrev_b64", 1); run<2>("v_alignbit_b32", 1); run<3>("v_lshlrev_b32", 1); run<4>("v_and_b32", 1);
    run<5>("v_cmp_lt_u64 + v_cndmask_b32", 2); run<6>("v_cmp_lt_u32 + v_cndmask_b32", 2); run<7>("v_lshl_add_u64", 1); run<8>("v_ffbh_u32", 1); run<9>("v_bfe_u32", 1);
    return 0;
}
