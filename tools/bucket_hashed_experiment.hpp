// Rejected experiment (round 6), kept as a record for DESIGN_HISTORY.md §3.13: "two buckets in flight" for the SHORT classes — a persistent grid whose
// workgroups walk the class list and keep the next bucket's words in flight behind the current bucket's LDS phases. Bit-identical (tools/dev_msd_bench.cpp
// compares the checksums of counts, kinds and words with k_bucket_msd's), and not faster: at PREFIX_BITS = 28 (9.8 M buckets of <= 128 words, 4.0 M of <= 512)
// the 128-slot class takes 3.84 - 4.10 ms against k_bucket_msd's 4.16 - 4.22, the 512-slot class 3.79 - 4.07 against 3.47 - 3.60, the 1024-slot class of cfg 2
// 1.01 against 0.78 — one bucket per wave and 32 waves per CU already overlap every bucket's HBM round trip with the other buckets' phases; what bounds
// the short classes is the instructions a wave issues per bucket (about 600 for 77 words), which the distinctness pre-check of k_bucket_msd's 128-slot
// class removes for runs without repeats. Included by tools/dev_msd_bench.cpp with -DMSD_BENCH_HASHED.
#pragma once
namespace cblx {
// A workgroup barrier that orders LDS accesses only: __syncthreads() also drains the vector-memory counter (s_waitcnt vmcnt(0)), which would make
// every barrier of a bucket's phases wait for the NEXT bucket's words — the loads this kernel wants in flight across them.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int THREADS> __device__ __forceinline__ u32 block_exclusive_scan_lds(u32 v, u32* smem) {
    constexpr int NW = THREADS / 64;
    const u32 lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const u32 inc = wave_inclusive_scan(v);
    if constexpr (NW == 1) return inc - v;
    if (lane == 63) smem[w] = inc;
    lds_barrier();
    u32 run = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) if ((u32)i < w) run += smem[i];
    lds_barrier();
    return run + inc - v;
}
// ---- KRN-3, SHORT runs that can only end up a Vec (<= 1024 words, not a Trie yet), narrow packed elements: buckets in flight (round 6) ------
// At PREFIX_BITS = 28 a bucket holds 90 words on average: cfg 3 on one GPU spends 7.4 ms in k_bucket_msd's 128- and 512-slot classes, one
// workgroup of ONE wave per bucket, 16 M of them — at 2.0 TB/s of the words they read. Such a wave's life is a chain: descriptor -> the
// run's words (an HBM round trip) -> a dozen LDS phases, and the CU's 32 wave slots are what bounds the buckets in flight. Here a workgroup
// WALKS the class list (buckets b, b + G, b + 2G, ...: a persistent grid of a few workgroups per wave slot) and keeps the next bucket's
// words IN FLIGHT while it works on the current one: the loads of bucket b + G are issued before the LDS phases of bucket b (their descriptor
// was fetched one step earlier still), so the HBM latency of every bucket but a workgroup's first hides behind its predecessor's phases, and
// the 16 M workgroup launches become a loop. The phases are k_bucket_msd's for hashed sub-buckets (counting sort on a hash of the suffix,
// then every element looks through its sub-bucket for an equal suffix with a smaller stream index); nothing is written when the run holds
// no repeat. A run that is a Trie already, or has a crowded sub-bucket (repeats), marks its list entry and takes the claim-table kernel.
#ifndef CBLX_HASHED_WAVES
#define CBLX_HASHED_WAVES 5  // the eight-slot classes keep a second bucket's words in registers: 16 more than k_bucket_msd's
#endif
template <int THREADS, int CAP, typename HiT>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(CAP <= 128 ? 8 : CBLX_HASHED_WAVES, 8))) void k_bucket_hashed(
    const BDesc* __restrict__ list, const u32* __restrict__ list_n, u64* __restrict__ lo, u32 SB, u32* __restrict__ out_count, u8* __restrict__ out_kind,
    u8* __restrict__ bail_flag, u32* __restrict__ bail_any) {
    static_assert(CAP <= (int)VEC_THRESHOLD && CAP <= (1 << PK_BITS), "short runs only");
    constexpr int ITEMS = CAP / THREADS, NW = THREADS / 64;
    __shared__ u64 s_klo[CAP + 4];
    __shared__ u32 s_off32[CAP / 2 + 2];
    u16* s_off = reinterpret_cast<u16*>(s_off32);
    __shared__ u32 s_scan[NW + 1];
    __shared__ u32 s_wtot[NW + 1];
    __shared__ u32 s_max;
    const u32 n = *list_n, G = gridDim.x;
    const u32 tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const u64 mask = SB >= 64 ? ~0ull : ((1ull << SB) - 1ull);
    u32 b = blockIdx.x;
    if (b >= n) return;
    auto fetch = [&](const BDesc& d, u64 (&k)[ITEMS]) {  // the run's words in wave-contiguous slices; slots past the run re-read its first word
        const u32 c = d.c & BDESC_LEN_MASK, R = (c + THREADS - 1) / THREADS;
        const u64* __restrict__ run = lo + d.start;
        const u32 first = w * 64 * R + lane;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const u32 e = first + j * 64;
            k[j] = run[((u32)j < R && e < c) ? e : 0u] & mask;
        }
    };
    BDesc dsc = list[b];
    BDesc dnext = b + G < n ? list[b + G] : dsc;
    u64 key[ITEMS];
    fetch(dsc, key);
    for (;;) {
        const bool more = b + G < n;  // uniform
        BDesc dnn = dnext;
        u64 knext[ITEMS];
        if (more) {
            if (b + 2 * G < n) dnn = list[b + 2 * G];  // the descriptor after next: its words are fetched one step from now
            fetch(dnext, knext);                       // in flight during everything below
        }
        {
            const u32 r = dsc.r, c = dsc.c & BDESC_LEN_MASK;
            const bool res_trie = (dsc.c & BDESC_TRIE) != 0;
            const u64 s0 = dsc.start;
            const u32 R = (c + THREADS - 1) / THREADS, EPW = 64 * R, slot_first = w * EPW + lane;
            u32 nbits = 32 - __builtin_clz(c - 1 > 0 ? c - 1 : 1);
            if (nbits > SB) nbits = SB;
            const u32 NB = 1u << nbits;
            bool gave_up = res_trie || c == 0;  // (a Trie needs the sorted layout: claim table -> the sorted kernels; c = 0 never occurs in a class list)
            if (!gave_up) {
                for (u32 i = tid; i < NB / 2 + 1; i += THREADS) s_off32[i] = 0;
                if (tid == 0) s_max = 0;
                lds_barrier();
                u32 sub[ITEMS], arr[ITEMS];
                bool valid[ITEMS];
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    const u32 e = slot_first + j * 64;
                    valid[j] = (u32)j < R && e < c;
                    Sfx<false> kk; kk.lo = key[j];
                    sub[j] = sfx_hash_bits<false>(kk, nbits);
                    arr[j] = 0;
                    if (valid[j]) arr[j] = atomicAdd(&s_off32[sub[j] >> 1], 1u << ((sub[j] & 1u) * 16u));
                }
                bool crowded = false;
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) {
                    arr[j] = (arr[j] >> ((sub[j] & 1u) * 16u)) & 0xFFFFu;
                    crowded |= arr[j] >= MSD_LIMIT_HASHED;
                }
                if (crowded) s_max = MSD_LIMIT_HASHED + 1u;
                lds_barrier();
                gave_up = s_max > MSD_LIMIT_HASHED;  // repeats: every copy of a value lands in its sub-bucket
                if (!gave_up) {
                    {   // exclusive scan of the NB counts
                        const u32 per = (NB + THREADS - 1) / THREADS;
                        const u32 b0 = tid * per;
                        u32 sum = 0, cnt[ITEMS];
#pragma unroll
                        for (int k = 0; k < ITEMS; ++k) {
                            cnt[k] = ((u32)k < per && b0 + k < NB) ? s_off[b0 + k] : 0u;
                            sum += cnt[k];
                        }
                        u32 ex = block_exclusive_scan_lds<THREADS>(sum, s_scan);
#pragma unroll
                        for (int k = 0; k < ITEMS; ++k) {
                            if ((u32)k < per && b0 + k < NB) { s_off[b0 + k] = (u16)ex; ex += cnt[k]; }
                        }
                    }
                    lds_barrier();
                    if (tid == 0) s_off[NB] = (u16)c;
                    u32 sbase[ITEMS];
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j) sbase[j] = s_off[sub[j]];
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j)
                        if (valid[j]) s_klo[sbase[j] + arr[j]] = (key[j] << PK_BITS) | (slot_first + j * 64);
                    lds_barrier();
                    // every element looks through its sub-bucket: a head has no equal suffix with a smaller stream index
                    bool head[ITEMS];
                    u32 wave_heads = 0;
                    u32 sb[ITEMS];
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j) sb[j] = s_off[sub[j] + 1];
#pragma unroll
                    for (int j = 0; j < ITEMS; ++j) {
                        bool dup = false;
                        if (valid[j]) {
                            const u32 e = slot_first + j * 64;
                            const u32 bb = sb[j], a = (bb - sbase[j] > 1u) ? sbase[j] : bb;  // alone in its sub-bucket: nothing to read
                            for (u32 q = a; q < bb; q += MSD_TRIP) {
                                u64 o[MSD_TRIP];
#pragma unroll
                                for (int k = 0; k < MSD_TRIP; ++k) o[k] = s_klo[q + k];  // (4 slack entries; what lies past bb is masked)
#pragma unroll
                                for (int k = 0; k < MSD_TRIP; ++k) dup |= (q + k < bb) && (o[k] >> PK_BITS) == key[j] && ((u32)o[k] & ((1u << PK_BITS) - 1u)) < e;
                            }
                        }
                        head[j] = valid[j] && !dup;
                        wave_heads += (u32)__builtin_popcountll(__ballot(head[j]));
                    }
                    u32 run = 0, d = wave_heads;
                    if constexpr (NW > 1) {
                        if (lane == 0) s_wtot[w] = wave_heads;
                        lds_barrier();
                        d = 0;
#pragma unroll
                        for (int ww = 0; ww < NW; ++ww) { const u32 t = s_wtot[ww]; if ((u32)ww < w) run += t; d += t; }
                    }
                    if (d != c) {  // repeats: the first occurrences in stream order, straight from the registers
                        u64* __restrict__ out = lo + s0;
#pragma unroll
                        for (int j = 0; j < ITEMS; ++j) {
                            const u64 bal = __ballot(head[j]);
                            if (head[j]) out[run + mbcnt(bal)] = key[j];
                            run += (u32)__builtin_popcountll(bal);
                        }
                    }
                    if (tid == 0) { out_count[r] = d; out_kind[r] = KIND_VEC; }
                }
            }
            if (gave_up && tid == 0) { bail_flag[b] = 1; *bail_any = 1u; }
        }
        if (!more) break;
        b += G;
        dsc = dnext;
        dnext = dnn;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) key[j] = knext[j];
        lds_barrier();  // the LDS arrays are the next bucket's from here on
    }
}

}  // namespace cblx
