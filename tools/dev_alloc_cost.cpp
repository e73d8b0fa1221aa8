// Dev probe: what the index loader's allocations cost in a FRESH process (every CLI command is one): 24 device arrays of 0.52 GB and 48
// pinned blocks of 8 MiB, one by one or as one allocation each. hipcc -O2 -o /tmp/alloc_cost tools/dev_alloc_cost.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const bool one = argc > 1;
    hipSetDevice(0);
    hipFree(nullptr);
    const double t0 = now();
    std::vector<void*> d, h;
    if (one) {
        void* p; hipMalloc(&p, 24ull * (520ull << 20)); d.push_back(p);
        hipHostMalloc(&p, 48ull * (8ull << 20), hipHostMallocDefault); h.push_back(p);
    } else {
        for (int i = 0; i < 24; ++i) { void* p; hipMalloc(&p, 520ull << 20); d.push_back(p); }
        const double t1 = now();
        printf("24 hipMalloc of 520 MB: %.1f ms\n", t1 - t0);
        for (int i = 0; i < 48; ++i) { void* p; hipHostMalloc(&p, 8ull << 20, hipHostMallocDefault); h.push_back(p); }
        printf("48 hipHostMalloc of 8 MiB: %.1f ms\n", now() - t1);
    }
    const double t2 = now();
    void* big; hipMalloc(&big, 9600ull << 20);
    printf("%s: %.1f ms; then hipMalloc of 9.6 GB: %.1f ms\n", one ? "one device + one pinned allocation" : "all of them", t2 - t0, now() - t2);
    return 0;
}
