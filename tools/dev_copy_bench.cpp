// Dev harness: streaming copy rates by access width (how far 8-byte accesses are from the 16-byte rate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <typename T, int ITEMS> __global__ __launch_bounds__(512) void k_copy(const T* __restrict__ a, T* __restrict__ b, size_t n) {
    size_t base = (size_t)blockIdx.x * 512 * ITEMS;
    T v[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { size_t i = base + j * 512 + threadIdx.x; v[j] = i < n ? a[i] : T{}; }
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { size_t i = base + j * 512 + threadIdx.x; if (i < n) b[i] = v[j]; }
}
template <typename T, int ITEMS> __global__ __launch_bounds__(512) void k_read(const T* __restrict__ a, uint64_t* out, size_t n) {
    size_t base = (size_t)blockIdx.x * 512 * ITEMS; uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { size_t i = base + j * 512 + threadIdx.x; if (i < n) { T v = a[i]; s += *(uint64_t*)&v; } }
    if (s == 0x1234567) out[0] = s;
}
template <typename T, int ITEMS> __global__ __launch_bounds__(512) void k_write(T* __restrict__ b, size_t n) {
    size_t base = (size_t)blockIdx.x * 512 * ITEMS;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) { size_t i = base + j * 512 + threadIdx.x; if (i < n) { T v{}; *(uint64_t*)&v = i; b[i] = v; } }
}
int main() {
    size_t bytes = 8ull << 30;
    void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    uint64_t* out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUN(name, T, IT, KIND) { size_t n = bytes / sizeof(T); dim3 g((n + 512 * IT - 1) / (512 * IT)); for (int r = 0; r < 2; ++r) { CK(hipEventRecord(e0)); \
      if (KIND == 0) hipLaunchKernelGGL((k_copy<T, IT>), g, dim3(512), 0, 0, (const T*)a, (T*)b, n); \
      else if (KIND == 1) hipLaunchKernelGGL((k_read<T, IT>), g, dim3(512), 0, 0, (const T*)a, out, n); \
      else hipLaunchKernelGGL((k_write<T, IT>), g, dim3(512), 0, 0, (T*)b, n); \
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); } float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
      printf("%-34s %7.3f ms  %7.1f GB/s\n", name, ms, (KIND == 0 ? 2.0 : 1.0) * bytes / ms / 1e6); }
    RUN("copy  8 B/lane x8", uint64_t, 8, 0) RUN("copy 16 B/lane x8", uint4, 8, 0) RUN("copy 16 B/lane x4", uint4, 4, 0) RUN("copy  4 B/lane x8", uint32_t, 8, 0) RUN("copy 1 B/lane x8", uint8_t, 8, 0)
    RUN("read  8 B/lane x8", uint64_t, 8, 1) RUN("read 16 B/lane x8", uint4, 8, 1) RUN("read 1 B/lane x8", uint8_t, 8, 1)
    RUN("write 8 B/lane x8", uint64_t, 8, 2) RUN("write 16 B/lane x8", uint4, 8, 2)
    return 0;
}
