#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t N = (size_t)atof(argv[1]) * (1ull << 30);
    const int T = atoi(argv[2]);
    const char* path = argc > 3 ? argv[3] : "/dev/shm/wtest.bin";
    std::vector<char> src(64 << 20, 'x');
    for (int mode = 0; mode < 5; ++mode) {
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        if (ftruncate(fd, N)) return 1;
        std::vector<std::thread> th;
        if (mode == 0 || mode >= 3) { if (posix_fallocate(fd, 0, N)) return 2; }
        else if (mode == 1) {
            const size_t per = (N / T + 4095) & ~(size_t)4095;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] { size_t a = t * per; if (a < N) fallocate(fd, 0, a, std::min(per, N - a)); });
            for (auto& x : th) x.join();
            th.clear();
        }
        double t1 = now();
        char* m = (char*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        double tp = now();
        if (mode == 3 || mode == 4) {  // pre-fault the mapping over T threads (3) or 4 threads (4)
            const int TP = mode == 3 ? T : 4;
            const size_t per = (N / TP + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
            std::vector<std::thread> pf;
            for (int t = 0; t < TP; ++t) pf.emplace_back([&, t] { size_t a = t * per; if (a < N) madvise(m + a, std::min(per, N - a), 23 /* MADV_POPULATE_WRITE */); });
            for (auto& x : pf) x.join();
            printf("  populate %.2f s; ", now() - tp);
        }
        std::thread fa;
        if (mode == 2) fa = std::thread([&] { for (size_t a = 0; a < N; a += (256u << 20)) fallocate(fd, 0, a, std::min<size_t>(256u << 20, N - a)); });
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
            for (size_t off = (size_t)t * (8 << 20); off < N; off += (size_t)T * (8 << 20)) memcpy(m + off, src.data(), std::min<size_t>(8 << 20, N - off));
        });
        for (auto& x : th) x.join();
        if (fa.joinable()) fa.join();
        munmap(m, N);
        close(fd);
        double t2 = now();
        printf("mode %d (%s): prep %.2f s, total %.2f s = %.2f GB/s\n", mode, mode == 0 ? "fallocate then mmap copy" : mode == 1 ? "parallel fallocate then mmap copy" : mode == 2 ? "fallocate thread alongside the copy" : "fallocate, parallel MADV_POPULATE_WRITE, mmap copy", t1 - t0, t2 - t0, N / (t2 - t0) / 1e9);
    }
    unlink(path);
}
