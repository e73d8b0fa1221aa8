// Dev probe: the reference's own call pattern — one cblx_insert_seq per record (/root/reference/examples/cbl.rs:160-163), then
// cblx_flush — from C++ (no Python in the loop), cfg 2's reads (10 M x 150 bp, iid ACGT), against ONE cblx_insert_seqs call.
// Build: g++ -O2 -std=c++17 -I include -o tools/dev_insert_seq_rate.bin tools/dev_insert_seq_rate.cpp -L cbl_amd -lcblx -Wl,-rpath,$PWD/cbl_amd
// (plain C++ against include/cblx.h — what any host program sees of the library). Usage: dev_insert_seq_rate.bin [reads] [json]:
// with `json` only the per-record leg runs and the last line is a JSON object (bench.py's `per_record` leg).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cblx.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { int _r = (x); if (_r) { fprintf(stderr, "%s -> %d\n", #x, _r); return 1; } } while (0)

int main(int argc, char** argv) {
    const uint64_t NR = argc > 1 ? strtoull(argv[1], nullptr, 10) : 10000000ull, L = 150;
    std::vector<uint8_t> bases(NR * L);
    uint64_t s = 42;
    for (uint64_t i = 0; i < NR * L; i += 32) {  // splitmix64, 2 bits per base
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        for (int k = 0; k < 32 && i + k < NR * L; ++k) bases[i + k] = "ACGT"[(z >> (2 * k)) & 3];
    }
    std::vector<uint64_t> off(NR + 1);
    for (uint64_t i = 0; i <= NR; ++i) off[i] = i * L;
    cblx_params p;
    memset(&p, 0, sizeof p);
    p.k = 31; p.prefix_bits = 24; p.canonical = 0; p.device = 0;
    cblx_ctx* c = nullptr;
    CK(cblx_create(&p, &c));
    const bool json = argc > 2 && !strcmp(argv[2], "json");
    double best = 1e9, best_calls = 0, best_flush = 0;
    uint64_t cnt = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(cblx_clear(c));
        const double t0 = now();
        for (uint64_t i = 0; i < NR; ++i) if (cblx_insert_seq(c, bases.data() + i * L, L)) return 2;
        const double t1 = now();
        CK(cblx_flush(c));
        const double t2 = now();
        CK(cblx_count(c, &cnt));
        printf("per record: %llu calls %.1f ms (%.1f ns per call), flush %.1f ms, total %.1f ms = %.2f G k-mers/s, count %llu\n", (unsigned long long)NR, (t1 - t0) * 1e3,
               (t1 - t0) / NR * 1e9, (t2 - t1) * 1e3, (t2 - t0) * 1e3, NR * (L - 30) / (t2 - t0) / 1e9, (unsigned long long)cnt);
        if (rep && t2 - t0 < best) { best = t2 - t0; best_calls = t1 - t0; best_flush = t2 - t1; }
    }
    if (json) {
        printf("{\"reads\": %llu, \"read_len\": %llu, \"ms_total\": %.3f, \"ms_calls\": %.3f, \"ms_flush\": %.3f, \"ns_per_call\": %.2f, \"value\": %.1f, \"distinct_kmers_in_index\": %llu}\n",
               (unsigned long long)NR, (unsigned long long)L, best * 1e3, best_calls * 1e3, best_flush * 1e3, best_calls / NR * 1e9, NR * (L - 30) / best, (unsigned long long)cnt);
        cblx_destroy(c);
        return 0;
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(cblx_clear(c));
        const double t0 = now();
        CK(cblx_insert_seqs(c, bases.data(), off.data(), NR));
        CK(cblx_flush(c));
        const double t2 = now();
        printf("one batch (pageable): total %.1f ms = %.2f G k-mers/s\n", (t2 - t0) * 1e3, NR * (L - 30) / (t2 - t0) / 1e9);
    }
    cblx_destroy(c);
    return 0;
}
