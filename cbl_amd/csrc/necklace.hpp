// necklace.hpp — k-mer -> (necklace, pos) word transform, usable from device and host code.
//
// Definition (normative, /root/reference/src/necklace/mod.rs:13-25): necklace = min over p in [0, BITS) of
// rotl_BITS(x, p), pos = the SMALLEST p attaining it; word = necklace << POS_BITS | pos
// (/root/reference/src/cbl.rs:181-184). The reference's streaming NecklaceQueue
// (src/necklace/queue.rs) returns the same pair; a GPU lane per k-mer has no use for the queue.
//
// Method (ours): the minimal rotation must start at the top of a LONGEST cyclic run of zero bits. Runs are
// found with r <- r & rotl(r, 1) (r starts as ~x): after L-1 steps bit s of r says "bits s, s-1, .., s-L+1 of x
// are all zero". The last non-zero r marks the tops of the longest runs; only those candidates (usually 1-2)
// are compared, highest s (= smallest p) first with a strict '<' so ties keep the smallest p.
// ~100-200 integer ops per k-mer instead of BITS full-width rotate+compare steps.
#pragma once
#include <cstdint>

#if !defined(__HIPCC__) && !defined(__host__)
#define __host__
#define __device__
#endif

namespace cblx {

typedef unsigned __int128 nk_u128;

template <typename T> struct NkBits;
template <> struct NkBits<uint64_t> {
    static __host__ __device__ inline int clz(uint64_t v) { return __builtin_clzll(v); }
};
template <> struct NkBits<nk_u128> {
    static __host__ __device__ inline int clz(nk_u128 v) {
        uint64_t hi = (uint64_t)(v >> 64);
        return hi ? __builtin_clzll(hi) : 64 + __builtin_clzll((uint64_t)v);
    }
};

// The same method for a ring of 2H bits (a k-mer of K = H bases wider than 64 bits) kept as TWO 64-bit halves of H bits each: a
// rotation by s < H is hi' = (hi << s | lo >> (H - s)) & M, lo' = (lo << s | hi >> (H - s)) & M — four 64-bit shifts and four
// logic operations — and one by s >= H swaps the halves first; on a 128-bit integer the compiler needs six shifts and more glue per
// rotation (every shift costs 1.7 logic operations on this part, tools/dev_valu_rate.cpp). Steps and tie rule are those of
// necklace_pos_fast below, which hands wide words with an even number of bits over to this function.
__host__ __device__ inline void necklace_pos_halves(nk_u128 x, unsigned BITS, nk_u128& necklace, unsigned& pos) {
    const unsigned H = BITS >> 1;  // < 64
    const uint64_t M = (1ull << H) - 1;
    const uint64_t xh = (uint64_t)(x >> H) & M, xl = (uint64_t)x & M;
    if ((xh | xl) == 0 || (xh & xl) == M) {  // single-symbol words: every rotation equal, smallest p = 0
        necklace = x;
        pos = 0;
        return;
    }
    // rotl by s on the ring, 0 <= s <= BITS (0 and BITS are the identity: b >> H and a << H vanish under the mask)
    auto rot = [&](uint64_t h, uint64_t l, unsigned s, uint64_t& oh, uint64_t& ol) {
        const bool sw = s >= H;
        const unsigned t = sw ? s - H : s;
        const uint64_t a = sw ? l : h, b = sw ? h : l;
        oh = ((a << t) | (b >> (H - t))) & M;
        ol = ((b << t) | (a >> (H - t))) & M;
    };
    uint64_t rh = ~xh & M, rl = ~xl & M, th, tl;
    unsigned L = 1;
    {
        rot(rh, rl, 1, th, tl);
        const uint64_t r2h = rh & th, r2l = rl & tl;      // runs >= 2
        rot(r2h, r2l, 2, th, tl);
        const uint64_t r4h = r2h & th, r4l = r2l & tl;    // runs >= 4
        if (r4h | r4l) {
            rh = r4h; rl = r4l; L = 4;
            rot(rh, rl, 4, th, tl); th &= rh; tl &= rl; if (th | tl) { rh = th; rl = tl; L += 4; }
            rot(rh, rl, 2, th, tl); th &= rh; tl &= rl; if (th | tl) { rh = th; rl = tl; L += 2; }
        } else if (r2h | r2l) {
            rh = r2h; rl = r2l; L = 2;
        }
    }
    {
        rot(rh, rl, 1, th, tl); th &= rh; tl &= rl;
        if (th | tl) { rh = th; rl = tl; ++L; }
        if (L == 11) {
            for (;;) {
                rot(rh, rl, 1, th, tl); th &= rh; tl &= rl;
                if ((th | tl) == 0) break;
                rh = th; rl = tl;
                ++L;
            }
        }
    }
    {
        rot(~xh & M, ~xl & M, L + 1, th, tl); th &= rh; tl &= rl;
        if (th | tl) { rh = th; rl = tl; }
    }
    uint64_t bh = M, bl = M;
    unsigned bestp = 0;
    bool first = true;
    while (rh | rl) {
        unsigned s;  // highest remaining candidate
        if (rh) { const unsigned b = 63u - (unsigned)__builtin_clzll(rh); rh &= ~(1ull << b); s = H + b; }
        else { const unsigned b = 63u - (unsigned)__builtin_clzll(rl); rl &= ~(1ull << b); s = b; }
        const unsigned p = BITS - 1 - s;
        rot(xh, xl, p, th, tl);
        if (first || th < bh || (th == bh && tl < bl)) {
            bh = th; bl = tl;
            bestp = p;
            first = false;
        }
    }
    necklace = ((nk_u128)bh << H) | bl;
    pos = bestp;
}

// x must already be masked to BITS bits, BITS < 8 * sizeof(T). T = uint64_t (BITS <= 62) or nk_u128 (BITS <= 118).
template <typename T> __host__ __device__ inline void necklace_pos_fast(T x, unsigned BITS, T& necklace, unsigned& pos) {
#ifndef CBLX_NK_HALVES
#define CBLX_NK_HALVES 1
#endif
    if constexpr (CBLX_NK_HALVES && sizeof(T) == 16) {
        if ((BITS & 1u) == 0 && BITS < 128) {
            nk_u128 nk;
            necklace_pos_halves((nk_u128)x, BITS, nk, pos);
            necklace = (T)nk;
            return;
        }
    }
    const T MASK = (BITS >= sizeof(T) * 8) ? ~(T)0 : ((((T)1) << BITS) - 1);
    if (x == 0 || x == MASK) {  // single-symbol words: every rotation equal, smallest p = 0
        necklace = x;
        pos = 0;
        return;
    }
    // rotl by s on the BITS-bit ring (0 < s < BITS)
    auto rotl_ring = [&](T v, unsigned s) -> T { return ((v << s) & MASK) | (v >> (BITS - s)); };
    T r = ~x & MASK;  // bit s: x has a zero at s                       (runs >= 1)
    unsigned L = 1;   // length of the runs r marks
    {   // Doubling, then a greedy binary refinement: with R_L = "a run of >= L zeros ends (downwards) at bit s",
        // R_{L+d} = R_L & rotl(R_L, d) for d <= L. Every lane of a wave executes the same few steps (a lane-dependent
        // linear search cost every lane the length of the longest run in the wave).
        const T r2 = r & rotl_ring(r, 1);    // runs >= 2
        const T r4 = r2 & rotl_ring(r2, 2);  // runs >= 4
        if (r4) {
            r = r4;                                            // L = 4
            L = 4;
            T t = r & rotl_ring(r, 4); if (t) { r = t; L += 4; }  // L in {4, 8}
            t = r & rotl_ring(r, 2); if (t) { r = t; L += 2; }    // L in {4, 6, 8, 10}
        } else if (r2) {
            r = r2;                                            // L in {2, 3}
            L = 2;
        }
    }
    {   // One +1 step settles every case but one: the doubling above left L in {1, 2, 4, 6, 8, 10} with the longest run known
        // to be < L + 2 unless L = 10 (L = 1: no run of 2 at all, the step is a no-op), so a second step can only succeed
        // after 10 -> 11. (A plain loop paid one extra rotate-and-test per k-mer to find that out.)
        T t = r & rotl_ring(r, 1);
        if (t != 0) { r = t; ++L; }
        if (L == 11) {
            for (;;) {
                t = r & rotl_ring(r, 1);
                if (t == 0) break;
                r = t;
                ++L;
            }
        }
    }
    {   // Every candidate reads 0^L 1 from its top; the next bit decides next: keep only the candidates that continue with
        // a 0, if there are any (a cheap mask operation that saves a round of the rotate-and-compare loop below for the
        // slowest lane of most waves).
        const T z = ~x & MASK;
        const T t = r & rotl_ring(z, L + 1);  // L + 1 <= BITS; a rotation by BITS is the identity here
        r = t ? t : r;
    }
    constexpr int TB = (int)sizeof(T) * 8;
    T best = MASK;
    unsigned bestp = 0;
    bool first = true;
    while (r != 0) {
        int s = TB - 1 - NkBits<T>::clz(r);  // highest remaining candidate
        r &= ~(((T)1) << s);
        unsigned p = BITS - 1 - (unsigned)s;
        // BITS < bit width of T (K is odd): p = 0 shifts right by BITS, which leaves 0 of the masked x
        T rot = ((x << p) & MASK) | (x >> (BITS - p));
        if (first || rot < best) {
            best = rot;
            bestp = p;
            first = false;
        }
    }
    necklace = best;
    pos = bestp;
}

// The definition, verbatim in spirit (used by tests and as a debug variant).
template <typename T> __host__ __device__ inline void necklace_pos_naive(T x, unsigned BITS, T& necklace, unsigned& pos) {
    T nk = x, rot = x;
    unsigned p = 0;
    for (int i = (int)BITS - 1; i >= 0; --i) {
        rot = ((rot & 1) << (BITS - 1)) | (rot >> 1);
        if (rot <= nk) { nk = rot; p = (unsigned)i; }
    }
    necklace = nk;
    pos = p;
}

// Reverse complement of a K-base packed k-mer (/root/reference/src/kmer.rs:293-348; code A=0 C=1 T=2 G=3, so
// complement = XOR 0b10): reverse the 2-bit groups, complement each, drop the 2*(W/2-K) pad bits.
__host__ __device__ inline uint64_t rev_comp64(uint64_t x, unsigned K) {
    uint64_t r = x;
    r = ((r >> 2) & 0x3333333333333333ull) | ((r & 0x3333333333333333ull) << 2);
    r = ((r >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((r & 0x0F0F0F0F0F0F0F0Full) << 4);
    r = __builtin_bswap64(r);
    r ^= 0xAAAAAAAAAAAAAAAAull;
    return r >> (2 * (32 - K));
}
__host__ __device__ inline nk_u128 rev_comp128(nk_u128 x, unsigned K) {
    uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
    // reverse all 64 groups of the 128-bit value: swap halves, reverse groups inside each
    uint64_t rlo = rev_comp64(hi, 32), rhi = rev_comp64(lo, 32);
    nk_u128 r = ((nk_u128)rhi << 64) | rlo;
    return r >> (2 * (64 - K));
}

__host__ __device__ inline unsigned popcount128(nk_u128 x) {
    return (unsigned)__builtin_popcountll((uint64_t)x) + (unsigned)__builtin_popcountll((uint64_t)(x >> 64));
}

// nucleotide code (/root/reference/src/kmer.rs:11-24): A/a=0 C/c=1 T/t=2 G/g=3, anything else invalid (skipped).
// For the 8 valid bytes the code is (b >> 1) & 3.
__host__ __device__ inline bool nuc_valid(uint8_t b) {
    uint8_t u = b & 0xDF;  // upper-case
    return u == 'A' || u == 'C' || u == 'G' || u == 'T';
}
__host__ __device__ inline unsigned nuc_code(uint8_t b) { return (b >> 1) & 3u; }

}  // namespace cblx
