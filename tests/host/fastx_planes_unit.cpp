// Host-side unit check of cbl_amd/csrc/fastx_parse.hpp (no HIP): regions cut at record starts, the counting pass and PlaneSink —
// the parser threads' in-place packing of sequence lines into the bit planes a FASTA / FASTQ file crosses PCIe as — against a
// byte-by-byte definition written here: what the reference's reader yields per record (needletail `seq()`: sequence lines, line
// ends stripped, qualities dropped; /root/reference/examples/cbl.rs:112-115) and code = (b >> 1) & 3, valid = ACGTacgt
// (/root/reference/src/kmer.rs:11-24). Built with g++ -fsanitize=address,undefined by tests/test_abi_and_host_units.py: the
// file lives in an exactly sized heap block, so a 16-byte load past its end is reported.
#include "../../cbl_amd/csrc/fastx_parse.hpp"

#include <cstdio>
#include <string>
using namespace cblx;

static uint64_t g_s = 99;
static uint32_t rnd() { g_s = g_s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(g_s >> 33); }

struct Naive { std::vector<u8> bases; std::vector<u64> ends; };

// kind 0: single-line FASTA; 1: multi-line FASTA with CRLF and blank lines; 2: FASTQ (qualities may start with '@' or '+')
static std::string make_file(int kind, int nrec, u32 K, Naive& nv) {
    static const char* al = "ACGTacgtNnRYKMxX-*";
    std::string f;
    if (kind == 1) f += "\n\r\n";
    for (int r = 0; r < nrec; ++r) {
        const size_t len = K + rnd() % 400;
        std::string seq(len, 'A');
        for (auto& c : seq) c = (rnd() % 50 == 0) ? al[rnd() % 18] : "ACGT"[rnd() & 3];
        nv.bases.insert(nv.bases.end(), seq.begin(), seq.end());
        nv.ends.push_back(nv.bases.size());
        if (kind == 0) {
            f += ">r" + std::to_string(r) + " some text\n" + seq + "\n";
        } else if (kind == 1) {
            f += ">r" + std::to_string(r) + "\r\n";
            for (size_t i = 0; i < len;) {
                const size_t w = 1 + rnd() % 90, n = std::min(w, len - i);
                f += seq.substr(i, n) + "\r\n";
                if (rnd() % 7 == 0) f += "\r\n";
                i += n;
            }
        } else {
            std::string q(len, 'I');
            if (rnd() % 3 == 0) q[0] = '@';
            if (rnd() % 5 == 0) q[0] = '+';
            f += "@r" + std::to_string(r) + "\n" + seq + "\n+\n" + q + "\n";
        }
    }
    if (kind == 0 && rnd() % 2) f.pop_back();  // no line end after the last record
    return f;
}

int main() {
    long bad = 0, checked = 0;
    const u32 K = 31;
    for (int rep = 0; rep < 36; ++rep) {
        const int kind = rep % 3;
        Naive nv;
        const std::string text = make_file(kind, 40 + rnd() % 300, K, nv);
        u8* file = (u8*)std::malloc(text.size());  // exactly sized: reads past the end are errors under ASan
        std::memcpy(file, text.data(), text.size());
        FastxMap m;
        m.d = file; m.size = text.size(); m.first = 0;
        while (m.first < m.size && (file[m.first] == '\n' || file[m.first] == '\r')) ++m.first;
        m.fmt = (char)file[m.first];
        std::vector<FastxRegion> regs;
        fx_make_regions(m, m.first, m.size, 500 + rnd() % 6000, regs);
        if (!fx_count_regions(m, regs, K, 4)) { ++bad; printf("count pass rejected a regular file (kind %d)\n", kind); }
        const size_t nr = regs.size();
        std::vector<u64> base(nr + 1, 0), rec0(nr + 1, 0);
        for (size_t i = 0; i < nr; ++i) { base[i + 1] = base[i] + regs[i].nbases; rec0[i + 1] = rec0[i] + regs[i].nrec; }
        if (base[nr] != nv.bases.size() || rec0[nr] != nv.ends.size()) { ++bad; printf("totals differ (kind %d): %llu bases %llu records\n", kind, (unsigned long long)base[nr], (unsigned long long)rec0[nr]); m.d = nullptr; std::free(file); continue; }
        const u64 nb = base[nr], ng = (nb + 15) / 16;
        std::vector<u32> codes(ng + 8, 0xDEADBEEFu);
        std::vector<u16> valid(ng + 8, 0xBEEF);
        std::vector<u64> ends(rec0[nr] + 1, ~0ull);
        fx_planes_prezero(base, ng + 8, codes.data(), valid.data());
        std::atomic<size_t> next{0};
        std::atomic<long> fails{0};
        auto worker = [&] {
            for (size_t i; (i = next.fetch_add(1)) < nr;) {
                PlaneSink sink(codes.data(), valid.data(), base[i], ends.data() + rec0[i], regs[i].nrec, base[i + 1]);
                if (!fx_walk(m.d, regs[i], m.fmt, K, sink) || sink.overflow || sink.pos != base[i + 1] || sink.nrec != regs[i].nrec) ++fails;
                sink.finish();
            }
        };
        std::vector<std::thread> th;
        for (int t = 0; t < 5; ++t) th.emplace_back(worker);
        for (auto& t : th) t.join();
        bad += fails.load();
        for (u64 i = 0; i < ng * 16; ++i) {
            const u32 w = codes[i >> 4] >> (i & 15);
            const u32 got = (w & 1u) | ((w >> 15) & 2u);
            const bool gok = (valid[i >> 4] >> (i & 15)) & 1;
            if (i >= nb) { if (gok) ++bad; continue; }  // positions past the end are not valid bases
            const u8 c = nv.bases[i], uc = c & 0xDF;
            const bool ok = uc == 'A' || uc == 'C' || uc == 'G' || uc == 'T';
            if (gok != ok || (ok && got != ((c >> 1) & 3u))) ++bad;
            ++checked;
        }
        for (size_t r = 0; r < nv.ends.size(); ++r) if (ends[r] != nv.ends[r]) ++bad;
        if (ends[nv.ends.size()] != ~0ull) ++bad;  // nothing written past the records
        // an irregular file: a record shorter than K in the middle is found by the counting pass
        m.d = nullptr;  // (not a mapping: nothing to unmap)
        std::free(file);
    }
    {   // a file that GREW between the counting pass and the second walk: the sink stops at its region's end, flags it, and touches
        // neither its neighbour's words nor anything past an exactly sized staging buffer (the sanitizer watches the heap blocks)
        std::string t = ">a\n" + std::string(70, 'C') + "\n>b\n" + std::string(45, 'G') + "\n";
        FastxMap m;
        m.d = (const u8*)t.data(); m.size = t.size(); m.first = 0; m.fmt = '>';
        std::vector<FastxRegion> regs;
        fx_make_regions(m, 0, m.size, 1 << 20, regs);
        const u64 counted = 64;  // what an earlier counting pass saw: fewer bases than the walk now meets
        u32* codes = (u32*)std::malloc((counted / 16) * 4);
        u16* valid = (u16*)std::malloc((counted / 16) * 2);
        u64* ends = (u64*)std::malloc(8);
        std::memset(codes, 0, (counted / 16) * 4);
        std::memset(valid, 0, (counted / 16) * 2);
        PlaneSink sink(codes, valid, 0, ends, 1, counted);
        const bool ok = fx_walk(m.d, regs[0], m.fmt, K, sink);
        sink.finish();
        if (!ok || !sink.overflow || sink.pos == counted) { ++bad; printf("a sink past its region's end was not flagged\n"); }
        m.d = nullptr;
        std::free(codes); std::free(valid); std::free(ends);
    }
    {   // record shorter than K: the counting pass says no
        std::string t = ">a\n" + std::string(40, 'C') + "\n>b\nACGT\n>c\n" + std::string(50, 'G') + "\n";
        u8* file = (u8*)std::malloc(t.size());
        std::memcpy(file, t.data(), t.size());
        FastxMap m;
        m.d = file; m.size = t.size(); m.first = 0; m.fmt = '>';
        std::vector<FastxRegion> regs;
        fx_make_regions(m, 0, m.size, 1 << 20, regs);
        if (fx_count_regions(m, regs, K, 2)) { ++bad; printf("short record accepted\n"); }
        m.d = nullptr;
        std::free(file);
    }
    printf("fastx planes unit: %ld bases checked, %ld bad\n", checked, bad);
    return bad ? 1 : 0;
}
