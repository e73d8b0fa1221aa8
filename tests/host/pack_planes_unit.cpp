// Host-side unit check of cbl_amd/csrc/xfer.hpp: pack_planes (the SSE2 / AVX2 packer of the bit planes a big host batch crosses PCIe
// as) against the scalar definition — code = (b >> 1) & 3 for the valid bytes ACGTacgt (/root/reference/src/kmer.rs:11-24), validity
// for every byte value — on random text over a dirty alphabet, at every alignment and for tails that are not multiples of 16.
// Built with hipcc (the header pulls in the HIP runtime types) and run on the host by tests/test_abi_and_host_units.py.
#include "../../cbl_amd/csrc/xfer.hpp"
#include <cstdio>
#include <vector>
using namespace cblx;
int main() {
    uint64_t s = 12345;
    auto rnd = [&] { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
    long bad = 0;
    const char* al = "ACGTacgtNnRYKMxX-*\n\r\t \xe9\xff\x00@[`{";
    std::vector<u8> src(1 << 16);
    for (auto& b : src) b = (rnd() % 4 == 0) ? (u8)al[rnd() % 33] : (u8)"ACGT"[rnd() & 3];
    for (int b = 0; b < 256; ++b) src[1000 + b] = (u8)b;  // every byte value once
    for (int rep = 0; rep < 200; ++rep) {
        const size_t off = rnd() % 4000, n = 1 + rnd() % 5000;
        std::vector<u32> codes((n + 15) / 16 + 2, 0xDEADBEEFu);
        std::vector<u16> valid((n + 15) / 16 + 2, 0xBEEF);
        const bool clean = pack_planes(src.data() + off, n, codes.data(), valid.data());
        bool want_clean = true;
        for (size_t i = 0; i < n; ++i) { const u8 uc = src[off + i] & 0xDF; want_clean = want_clean && (uc == 'A' || uc == 'C' || uc == 'G' || uc == 'T'); }
        if (clean != want_clean) ++bad;  // the return value: every base valid
        for (size_t i = 0; i < ((n + 15) / 16) * 16; ++i) {
            const u32 w = codes[i >> 4] >> (i & 15);
            const u32 got = (w & 1u) | ((w >> 15) & 2u);
            const bool gok = (valid[i >> 4] >> (i & 15)) & 1;
            if (i >= n) { if (gok || got) ++bad; continue; }  // bits past the end stay clear
            const u8 c = src[off + i], uc = c & 0xDF;
            const bool ok = uc == 'A' || uc == 'C' || uc == 'G' || uc == 'T';
            if (gok != ok || (ok && got != ((c >> 1) & 3u))) ++bad;
        }
        if (codes[(n + 15) / 16] != 0xDEADBEEFu || valid[(n + 15) / 16] != 0xBEEF) ++bad;  // nothing written past the last group
    }
    // the return value on clean text with at most one dirty byte somewhere (also in the last, partly filled group)
    for (int rep = 0; rep < 400; ++rep) {
        const size_t n = 1 + rnd() % 3000;
        std::vector<u8> t(n + 64);
        for (auto& b : t) b = (u8)"ACGTacgt"[rnd() & 7];
        const bool inject = rep % 2 == 1;
        if (inject) t[rep % 8 == 1 ? n - 1 : rnd() % n] = (u8)"N-x\n"[rnd() & 3];
        std::vector<u32> codes((n + 15) / 16 + 1);
        std::vector<u16> valid((n + 15) / 16 + 1);
        if (pack_planes(t.data(), n, codes.data(), valid.data()) != !inject) ++bad;
    }
    printf("bad=%ld\n", bad);
    return bad ? 1 : 0;
}
