// REJECTED EXPERIMENT (round 2), kept as the record DESIGN_HISTORY.md §3.4 cites — not part of the library. It was
// cbl_amd/csrc/kernels_fused.hpp; tools/fused_first_pass_experiment.patch holds the plumbing (pipeline.hpp, kernels_encode.hpp).
// Bit-identical (full GPU suite + 300 fuzz cases green with it on by default), but slower: see the numbers in DESIGN_HISTORY.md §3.4.
//
// kernels_fused.hpp — KRN-1b: the word transform fused into the FIRST partition pass.
//
// The plain path writes every word once (KRN-1) only to read it straight back and move it (pass A of KRN-2): 2 x 9 bytes
// per k-mer of HBM traffic whose only purpose is to wait for pass A's histogram. Finding `pos` (the longest-zero-run search,
// necklace.hpp) is what costs KRN-1 its time; GIVEN pos, the word is one rotation of the k-mer away. So for a batch whose
// bases stay in HBM for the whole call (cblx_insert_seqs_device):
//   k_encode<.., FUSED>   reads the bases, stores `pos` (1 byte per k-mer) and the first pass's histogram rows;
//   k_encode_scatter      reads the bases and `pos` again, rebuilds each word in registers, ranks the tile by the first
//                         pass's digit and writes the records where pass A would have put them.
// Same result as k_encode + k_radix_scatter (same stable order: the tile's k-mers are ranked in output-slot order), one
// full write and one full read of the record array less. Words of chunks with non-ACGT bytes are not windows of the base
// stream: k_encode_dirty writes them to the idle ping-pong buffer as before and they are picked up from there.
// Reference semantics are those of kernels_encode.hpp (get_seq_words, src/cbl.rs:239-289) and kernels_radix.hpp.
#pragma once
#include "kernels_radix.hpp"

namespace cblx {

#ifndef CBLX_FUSED_EXP
#define CBLX_FUSED_EXP 0
#endif
static const u32 FUS_PAR_WORDS = (ENC_MAX_KMERS + RDX_THREADS) / 64 + 2;

// Everything a workgroup needs to know about its encode tile, in one record: the tile's loads (bases, chunk tables, `pos`,
// its rows of the column prefixes) then all leave in ONE round trip after this one — the kernel is bound by the length of
// its chain of dependent memory accesses (three workgroups per CU), not by bytes or instructions.
struct FusedTile {
    u64 kbase;  // first k-mer (= output slot) of the tile
    u64 row0;   // its first row of the count matrix
    u64 a0;     // first byte of its window of the base stream (16-byte aligned)
    u32 c0, nc; // its chunks
    u32 span;   // bytes from a0 to the end of its last chunk
    u32 q;      // its k-mers
};
__global__ void k_fused_tiles(const u32* __restrict__ tile_first, const u64* __restrict__ chunk_start, const u32* __restrict__ chunk_len,
                              const u64* __restrict__ kmer_off, const u64* __restrict__ tile_row, u64 ntiles, FusedTile* __restrict__ out) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    const u32 c0 = tile_first[t], c1 = tile_first[t + 1];
    FusedTile f{};
    f.c0 = c0;
    f.nc = c1 > c0 ? c1 - c0 : 0u;
    f.row0 = tile_row[t];
    if (f.nc) {
        f.kbase = kmer_off[c0];
        f.q = (u32)(kmer_off[c1] - f.kbase);
        f.a0 = chunk_start[c0] & ~(u64)15;
        f.span = (u32)(chunk_start[c1 - 1] + chunk_len[c1 - 1] - f.a0);
    }
    out[t] = f;
}

// HiT: layout of the hi part of the DIRTY chunks' words (u8: 65..72-bit words, NoHi: <= 64 bits). The records leave without
// a hi part (what it held is implied by the segment after this pass, as in k_radix_scatter<HiT, NoHi>).
template <typename HiT>
__global__ __launch_bounds__(RDX_THREADS, CBLX_FUSED_EXP ? 8 : 6) void k_encode_scatter(const u8* __restrict__ bases, u64 total_bases, const u64* __restrict__ chunk_start,
                                                                const u32* __restrict__ chunk_len, const u64* __restrict__ kmer_off,
                                                                const u8* __restrict__ dirty /* may be null */, const FusedTile* __restrict__ tiles,
                                                                u32 ntiles, Consts P, const u8* __restrict__ pos,
                                                                const u64* __restrict__ d_lo, const HiT* __restrict__ d_hi, DigitBits dfn,
                                                                const u32* __restrict__ colpre, const u32* __restrict__ adj, u64* __restrict__ out_lo,
                                                                DigitBits next_dfn, u8* __restrict__ out_next) {
    static_assert(!std::is_same<HiT, u64>::value, "k-mers of up to 32 bases only");
    static_assert(ENC_MAX_KMERS < 65536 && ENC_MAX_BASES < 32768, "16-bit tile tables");
    __shared__ u64 s_lo[RDX_TILE];  // canonical: the round's words by output slot; then rank counters; then the staged records
    __shared__ u8 s_dig[RDX_TILE];  // canonical: first-pass digit by output slot; then the digit of the staged record
    u32* s_wcnt = reinterpret_cast<u32*>(s_lo);
    __shared__ u32 s_dbase[256];
    __shared__ u64 s_gbase[256];
    __shared__ u32 s_scan[RDX_THREADS / 64 + 1];
#if CBLX_FUSED_EXP  // occupancy experiment (non-canonical, single-round tiles only): everything the word computation reads lives in the staging area
    u8* s_pos = reinterpret_cast<u8*>(s_lo) + 16384;
    u32* s_codes = reinterpret_cast<u32*>(reinterpret_cast<u8*>(s_lo) + 20480);
    u16* s_koff = reinterpret_cast<u16*>(reinterpret_cast<u8*>(s_lo) + 24576);
    u16* s_cstart = reinterpret_cast<u16*>(reinterpret_cast<u8*>(s_lo) + 28672);
    __shared__ u64 s_par[2];
    __shared__ u16 s_parpre[2];
    __shared__ u16 s_cfwd[2];
#else
    __shared__ u8 s_pos[RDX_TILE];  // `pos` of the round's output slots
    __shared__ u32 s_codes[ENC_CODE_WORDS];
    __shared__ u16 s_koff[ENC_MAX_CHUNKS + 2];
    __shared__ u16 s_cstart[ENC_MAX_CHUNKS];  // first base of the chunk in the tile's code stream; bit 15: the chunk is dirty
    __shared__ u64 s_par[FUS_PAR_WORDS];
    __shared__ u16 s_parpre[FUS_PAR_WORDS];
    __shared__ u16 s_cfwd[ENC_MAX_CHUNKS + 2];
#endif

    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
#if CBLX_FUSED_EXP
    constexpr bool CANON = false;
#else
    const bool CANON = P.canonical;
#endif
    if ((blockIdx.x >> 3) >= (ntiles + 7u) / 8u) return;
    const u32 t = xcd_tile(blockIdx.x, ntiles);  // neighbouring tiles append to the same few bins: keep them on one XCD's L2
    if (t >= ntiles) return;
    const FusedTile ft = tiles[t];
    const u32 c0 = ft.c0, nc = ft.nc;
    if (nc == 0) return;
    const u64 A0 = ft.a0, kbase = ft.kbase, row0 = ft.row0;
    const u32 Q = ft.q;
    auto load_pos = [&](u32 r0) {  // the round's `pos` bytes, 16 per lane
        const u32 n = Q - r0 < (u32)RDX_TILE ? Q - r0 : (u32)RDX_TILE;
        const u8* src = pos + kbase + r0;
        if (tid * 16 < n) {
            if ((((uintptr_t)src) & 15) == 0 && tid * 16 + 16 <= n) {
                *reinterpret_cast<uint4*>(&s_pos[tid * 16]) = *reinterpret_cast<const uint4*>(src + tid * 16);
            } else {
                for (u32 k = 0; k < 16 && tid * 16 + k < n; ++k) s_pos[tid * 16 + k] = src[tid * 16 + k];
            }
        }
    };
    static_assert(RDX_THREADS * 16 >= RDX_TILE, "one 16-byte piece of `pos` per lane");
    load_pos(0);
    u64 gb = 0;  // where this tile's first record of digit `tid` goes (prefetched: nothing else depends on it)
    if (tid < 256) gb = (u64)adj[tid] + colpre[row0 * 256 + tid];

    bool dl = false;
    for (u32 i = tid; i <= nc; i += RDX_THREADS) s_koff[i] = (u16)(kmer_off[c0 + i] - kbase);
    for (u32 i = tid; i < nc; i += RDX_THREADS) {
        const bool d = dirty && dirty[c0 + i];
        dl |= d;
        s_cstart[i] = (u16)((u32)(chunk_start[c0 + i] - A0) | (d ? 0x8000u : 0u));
    }
    const u32 nwords = (ft.span + 15) >> 4;
    for (u32 i = tid; i < nwords + 6 && i < ENC_CODE_WORDS; i += RDX_THREADS) {
        const u64 b = A0 + (u64)i * 16;
        u32 packed = 0;
        if (i < nwords) {
            if (b + 16 <= total_bases) {
                packed = pack16(*reinterpret_cast<const uint4*>(bases + b));
            } else {
                for (u32 k = 0; k < 16 && b + k < total_bases; ++k) packed |= nuc_code(bases[b + k]) << (30 - 2 * k);
            }
        }
        s_codes[i] = packed;
    }
    // (values every lane agrees on are moved to scalar registers by hand: the compiler cannot know, and the kernel is short of vector registers)
    const bool any_dirty = __builtin_amdgcn_readfirstlane(__syncthreads_or(dl ? 1 : 0)) != 0;
    // reads of one length back to back: chunk and position of a slot by arithmetic (as k_encode)
    const u32 nk0 = __builtin_amdgcn_readfirstlane((u32)s_koff[1]), cs0 = __builtin_amdgcn_readfirstlane((u32)s_cstart[0]),
              len0 = nc > 1 ? __builtin_amdgcn_readfirstlane((u32)s_cstart[1]) - cs0 : 0u;
    bool uni = nk0 != 0 && !any_dirty;
    for (u32 i = tid; i < nc; i += RDX_THREADS) uni = uni && s_koff[i + 1] == (i + 1) * nk0 && s_cstart[i] == cs0 + i * len0;
    const bool uniform = __builtin_amdgcn_readfirstlane(__syncthreads_and(uni ? 1 : 0)) != 0;
    const u32 nk0_magic = uniform ? (u32)(0x100000000ull / nk0) + 1u : 0u;  // umulhi(n, magic) = n / nk0 for n < 2^19 (nk0 < 2^13)

    auto find_chunk = [&](u32 q) -> u32 {  // last i with s_koff[i] <= q
        u32 lo = 0, hi = nc;
        while (hi - lo > 1) {
            const u32 mid = (lo + hi) >> 1;
            if (s_koff[mid] <= q) lo = mid; else hi = mid;
        }
        return lo;
    };
    const u64 MASK = (1ull << P.KB) - 1ull;
    auto word_of = [&](u64 x, u32 p, u64& lo, u64& hi) {  // the k-mer (strand already chosen) rotated to its necklace
        const u64 rot = ((x << p) & MASK) | (x >> (P.KB - p));  // p = 0: the second term shifts the masked k-mer out
        lo = (rot << P.POS) | (u64)p;
        hi = rot >> (64 - P.POS);
    };

    if (CANON) {  // strand flags and their prefix counts, as k_encode
        const u32 qpad = (Q + RDX_THREADS - 1) / RDX_THREADS * RDX_THREADS;
        u32 cf = tid < Q ? find_chunk(tid) : 0u;
        for (u32 q = tid; q < qpad; q += RDX_THREADS) {
            bool fwd = false;
            if (q < Q) {
                while (q >= s_koff[cf + 1]) ++cf;
                const u32 cs = s_cstart[cf];
                if (!(cs >> 15)) fwd = kmer_is_fwd<false>(extract_kmer<false>(s_codes, cs + (q - s_koff[cf]), P.K));
            }
            const u64 bal = __ballot(fwd);
            if (lane == 0) s_par[q >> 6] = bal;
        }
        __syncthreads();
        if (tid == 0) {
            u32 run = 0;
            const u32 nw = qpad >> 6;
            for (u32 i = 0; i < nw; ++i) { s_parpre[i] = (u16)run; run += (u32)__builtin_popcountll(s_par[i]); }
            s_parpre[nw] = (u16)run;
            s_par[nw] = 0;
        }
        __syncthreads();
    }
    auto cum_fwd = [&](u32 q) -> u32 { return (u32)s_parpre[q >> 6] + (u32)__builtin_popcountll(s_par[q >> 6] & ((1ull << (q & 63)) - 1ull)); };
    if (CANON) {
        for (u32 i = tid; i <= nc; i += RDX_THREADS) s_cfwd[i] = (u16)cum_fwd(s_koff[i]);
        __syncthreads();
    }

    for (u32 r0 = 0; r0 < Q; r0 += RDX_TILE) {  // one round per row of the count matrix (a tile of short reads: one)
        const u32 n_round = Q - r0 < (u32)RDX_TILE ? Q - r0 : (u32)RDX_TILE;
        if (CANON) {
            // the words of the clean chunks that touch the round, computed in k-mer order and parked at their output slot
            // (forward-strand k-mers of a chunk first, src/cbl.rs:262-281)
            const u32 ca = find_chunk(r0), cb = find_chunk(r0 + n_round - 1);
            const u32 q0 = s_koff[ca], q1 = s_koff[cb + 1];
            u32 ci = ca;
            for (u32 q = q0 + tid; q < q1; q += RDX_THREADS) {
                while (q >= s_koff[ci + 1]) ++ci;
                const u32 cs = s_cstart[ci];
                if (cs >> 15) continue;
                const u32 koff = s_koff[ci], j = q - koff;
                u64 x = extract_kmer<false>(s_codes, cs + j, P.K);
                const u32 cbf = s_cfwd[ci], nfwd = (u32)s_cfwd[ci + 1] - cbf, rf = cum_fwd(q) - cbf;
                const bool rc = !kmer_is_fwd<false>(x);
                const u32 drel = koff + (rc ? nfwd + (j - rf) : rf);
                if (drel < r0 || drel >= r0 + n_round) continue;
                if (rc) x = rev_comp64(x, P.K);
                u64 lo, hi;
                word_of(x, s_pos[drel - r0], lo, hi);
                s_lo[drel - r0] = lo;
                s_dig[drel - r0] = (u8)dfn(lo, hi);
            }
            __syncthreads();
        }
        u64 klo[RDX_ITEMS];
        u32 digit[RDX_ITEMS];
        {
            const u32 e0 = w * (64 * RDX_ITEMS) + lane;  // slot of item 0; item j: + 64 j
            // one item: straight-line code, nothing carried from item to item (carried chunk cursors and their loops cost
            // the unrolled body two dozen spilled registers)
            auto item = [&](int j, auto mode) {
                constexpr int MODE = decltype(mode)::value;  // 0: parked words (canonical, clean tile), 1: uniform reads, 2: general
                const u32 e = e0 + j * 64;
                klo[j] = 0;
                digit[j] = 255u;
                if (e < n_round) {
                    if constexpr (MODE == 0) {
                        klo[j] = s_lo[e];
                        digit[j] = s_dig[e];
                    } else {
                        const u32 slot = r0 + e;
                        u32 ci, cs, jj;
                        if constexpr (MODE == 1) {
                            ci = __umulhi(slot, nk0_magic);  // slot / nk0 (slot < 2^13)
                            jj = slot - ci * nk0;
                            cs = cs0 + ci * len0;
                        } else {
                            ci = find_chunk(slot);
                            jj = slot - s_koff[ci];
                            cs = s_cstart[ci];
                        }
                        if (cs >> 15) {  // dirty chunk: its words were written by k_encode_dirty
                            klo[j] = d_lo[kbase + slot];
                            digit[j] = dfn(klo[j], (u64)ld_hi<HiT>(d_hi, kbase + slot));
                        } else if (CANON) {
                            klo[j] = s_lo[e];
                            digit[j] = s_dig[e];
                        } else {
                            u64 hi;
                            word_of(extract_kmer<false>(s_codes, cs + jj, P.K), s_pos[e], klo[j], hi);
                            digit[j] = dfn(klo[j], hi);
                        }
                    }
                }
            };
            if (CANON && !any_dirty) {
#pragma unroll
                for (int j = 0; j < RDX_ITEMS; ++j) item(j, std::integral_constant<int, 0>());
            } else if (uniform) {
#pragma unroll
                for (int j = 0; j < RDX_ITEMS; ++j) item(j, std::integral_constant<int, 1>());
            } else {
#pragma unroll
                for (int j = 0; j < RDX_ITEMS; ++j) item(j, std::integral_constant<int, 2>());
            }
        }
#if CBLX_FUSED_EXP
        __syncthreads();
#else
        if (CANON) __syncthreads();  // the parked words are in registers before the rank counters overwrite them
#endif
        tile_rank_packed<RDX_THREADS, RDX_ITEMS>(digit, s_wcnt, s_dbase, s_scan, RDX_ITEMS);  // digit[j] = digit << 16 | position
        if (tid < 256) s_gbase[tid] = gb - s_dbase[tid];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RDX_ITEMS; ++j) {
            const u32 p = digit[j] & 0xFFFFu;  // tail slots land in [n_round, RDX_TILE)
            s_lo[p] = klo[j];
            s_dig[p] = (u8)(digit[j] >> 16);
        }
        __syncthreads();
        // (the lane id is laundered per round: otherwise the eight 64-bit slot numbers below are hoisted out of the round loop
        // as loop invariants, spilled, and every reload in this loop waits for the store in front of it to complete)
        u32 tl = tid;
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int j = 0; j < RDX_ITEMS; ++j) {
            const u32 s = j * RDX_THREADS + tl;
            if (s < n_round) {
                const u64 a = s_lo[s];
                const u64 dst = s_gbase[s_dig[s]] + s;
                out_lo[dst] = a;
                if (out_next) out_next[dst] = (u8)next_dfn(a, 0);
            }
        }
        if (r0 + RDX_TILE < Q) {  // a tile of long sequences can hold more k-mers than one round takes: the next row
            load_pos(r0 + RDX_TILE);
            if (tid < 256) gb = (u64)adj[tid] + colpre[(row0 + r0 / RDX_TILE + 1) * 256 + tid];
        }
        __syncthreads();  // the staging area is reused by the next round
    }
}

}  // namespace cblx
