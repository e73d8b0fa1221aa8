// common.hpp — shared host/device definitions for libcblx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace cblx {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef unsigned __int128 u128;

// Derived constants (reference: src/cbl.rs:16-32,65-67; build.rs:26-52)
struct Consts {
    u32 K, PB, KB, POS, WB, SB, BYTES;
    u32 canonical;
    __host__ __device__ bool wide_kmer() const { return KB > 64; }   // k-mer needs 128 bits (K >= 33)
    __host__ __device__ bool has_hi() const { return WB > 64; }      // word needs a hi part
    __host__ __device__ bool wide_suffix() const { return SB > 64; } // suffix needs 128 bits
};

static const u32 CHUNK_KMERS = 2048;   // src/cbl.rs:67
static const u32 VEC_THRESHOLD = 1024; // src/wordset/mod.rs:34

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define CBLX_HIP(expr)                                                                                  \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            throw ::cblx::Error(4, std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// bit helpers on a (lo, hi) 128-bit word
__host__ __device__ inline u32 get_bits(u64 lo, u64 hi, u32 shift, u32 nbits) {
    u64 v;
    if (shift >= 64) v = hi >> (shift - 64);
    else if (shift == 0) v = lo;
    else v = (lo >> shift) | (hi << (64 - shift));
    return (u32)v & ((1u << nbits) - 1u);
}

// ------------------------------------------------------------------------------------------------
// wave / block primitives (wave = 64 lanes)
__device__ __forceinline__ u32 lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ u32 mbcnt(u64 mask) {  // popcount(mask & lanes below me)
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

template <typename T> __device__ __forceinline__ T wave_inclusive_scan(T v) {
    const u32 lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        T t = __shfl_up(v, o, 64);
        if (lane >= (u32)o) v += t;
    }
    return v;
}
template <typename T> __device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Exclusive scan of one value per thread across a block of THREADS (multiple of 64). `smem` needs THREADS/64+1
// entries of T. Returns the exclusive prefix; *total (if non-null) gets the block total on every thread.
template <int THREADS, typename T> __device__ __forceinline__ T block_exclusive_scan(T v, T* smem, T* total) {
    constexpr int NW = THREADS / 64;
    const u32 lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T inc = wave_inclusive_scan(v);
    if (lane == 63) smem[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        T run = 0;
        for (int i = 0; i < NW; ++i) { T t = smem[i]; smem[i] = run; run += t; }
        smem[NW] = run;
    }
    __syncthreads();
    T res = smem[w] + inc - v;
    if (total) *total = smem[NW];
    __syncthreads();
    return res;
}

}  // namespace cblx
