// Host-side unit check of cbl_amd/csrc/necklace.hpp (the same source the HIP kernels compile):
// fast longest-zero-run method == the definition, on random, sparse, periodic and degenerate words;
// rev_comp == group reversal + complement. Built and run by tests/test_host_units.py.
#include "../../cbl_amd/csrc/necklace.hpp"
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
using namespace cblx;
static uint64_t s = 0x1234567;
static uint64_t rnd() { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
template <class T> static T rnd_t() { return sizeof(T) == 8 ? (T)rnd() : (T)(((nk_u128)rnd() << 64) | rnd()); }
template <class T> static long run(unsigned BITS, long n) {
    const T MASK = (((T)1) << BITS) - 1;
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        T x;
        switch (i % 8) {
            case 0: x = rnd_t<T>(); break;
            case 1: x = rnd_t<T>() & rnd_t<T>() & rnd_t<T>(); break;             // sparse ones
            case 2: x = ~(rnd_t<T>() & rnd_t<T>() & rnd_t<T>()); break;          // sparse zeros
            case 3: { unsigned h = BITS / 2; T w = rnd_t<T>() & ((((T)1) << h) - 1); x = (w << h) | w; break; }  // w||w
            case 4: { T w = rnd_t<T>() & 0xFF; x = 0; for (unsigned b = 0; b < BITS; b += 8) x |= w << b; break; }  // period 8
            case 5: x = ((T)1) << (rnd() % BITS); break;                          // single one
            case 6: x = ~(((T)1) << (rnd() % BITS)); break;                       // single zero
            default: x = (i & 8) ? 0 : MASK; break;
        }
        x &= MASK;
        T a, b; unsigned pa, pb;
        necklace_pos_fast<T>(x, BITS, a, pa);
        necklace_pos_naive<T>(x, BITS, b, pb);
        if (a != b || pa != pb) { if (bad < 5) fprintf(stderr, "mismatch BITS=%u i=%ld pos %u vs %u\n", BITS, i, pa, pb); ++bad; }
    }
    return bad;
}
int main() {
    long bad = 0;
    for (unsigned bits : {10u, 14u, 18u, 50u, 58u, 62u}) bad += run<uint64_t>(bits, 200000);
    for (unsigned bits : {62u, 66u, 90u, 118u}) bad += run<nk_u128>(bits, 100000);
    // rev_comp: against group-wise definition
    for (unsigned K : {5u, 7u, 25u, 31u, 32u}) for (int i = 0; i < 20000; ++i) {
        uint64_t x = K == 32 ? rnd() : rnd() & ((1ull << (2 * K)) - 1), r = 0, y = x;
        for (unsigned j = 0; j < K; ++j) { r = (r << 2) | ((y & 3) ^ 2); y >>= 2; }
        if (rev_comp64(x, K) != r) { ++bad; if (bad < 5) fprintf(stderr, "rc64 mismatch K=%u\n", K); }
    }
    for (unsigned K : {33u, 45u, 59u, 64u}) for (int i = 0; i < 20000; ++i) {
        nk_u128 x = rnd_t<nk_u128>(); if (K < 64) x &= ((((nk_u128)1) << (2 * K)) - 1);
        nk_u128 r = 0, y = x;
        for (unsigned j = 0; j < K; ++j) { r = (r << 2) | ((y & 3) ^ 2); y >>= 2; }
        if (rev_comp128(x, K) != r) { ++bad; if (bad < 5) fprintf(stderr, "rc128 mismatch K=%u\n", K); }
    }
    for (int b = 0; b < 256; ++b) {
        bool v = b=='A'||b=='C'||b=='G'||b=='T'||b=='a'||b=='c'||b=='g'||b=='t';
        if (nuc_valid((uint8_t)b) != v) ++bad;
    }
    const char* L = "ACTGactg"; for (int i = 0; i < 8; ++i) if (nuc_code((uint8_t)L[i]) != (unsigned)(i & 3)) ++bad;
    printf("bad=%ld\n", bad);
    return bad ? 1 : 0;
}
