// prims.h -- wave64 / workgroup device helpers for gfx950 (CDNA4).
// Wavefronts are 64 lanes: every ballot is a 64-bit mask, every cross-lane
// idiom below is written for that width (no 32-lane assumptions anywhere).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pss {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// popcount of the bits of `mask` below this lane (v_mbcnt_lo/hi pair).
__device__ __forceinline__ u32 mbcnt(u64 mask)
{
    return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}

// Lanes of the wave holding the same 8-bit digit as this lane ("match-any").
// Built as four 4-way (2-bit) bucket refinements: each step ballots two digit
// bits and keeps the lanes that fall in the same one of the 4 sub-buckets.
__device__ __forceinline__ u64 match_digit8(u32 d, u64 valid_mask)
{
    u64 peers = valid_mask;
#pragma unroll
    for (int s = 0; s < 8; s += 2) {
        const bool b0 = (d >> s) & 1u;
        const bool b1 = (d >> (s + 1)) & 1u;
        const u64 m0 = __ballot(b0);
        const u64 m1 = __ballot(b1);
        peers &= (b0 ? m0 : ~m0) & (b1 ? m1 : ~m1);
    }
    return peers;
}

// Inclusive wave scans (6 shuffle steps).
__device__ __forceinline__ u32 wave_incl_sum(u32 v)
{
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        u32 t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ u64 wave_incl_sum64(u64 v)
{
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        u64 t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ u32 wave_incl_max(u32 v)
{
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
        u32 t = __shfl_up(v, o);
        if (lane >= o) v = max(v, t);
    }
    return v;
}

// Workgroup-wide exclusive sum over one value per thread.  `scratch` needs one
// u32 per wave (+1).  Returns the exclusive prefix; *total gets the block sum.
// Contains two barriers; every thread of the block must call it.
template <int WAVES>
__device__ __forceinline__ u32 block_excl_sum(u32 v, u32 *scratch, u32 *total)
{
    const u32 incl = wave_incl_sum(v);
    if (lane_id() == kWave - 1) scratch[wave_id()] = incl;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const u32 s = scratch[w];
        if (w < wave_id()) base += s;
        tot += s;
    }
    __syncthreads();
    if (total) *total = tot;
    return base + incl - v;
}

// XCD-aware work assignment.  The dispatcher is observed to place workgroup b
// on XCD (b % 8); handing XCD x a CONTIGUOUS eighth of the ranges keeps
// neighbouring ranges (whose scatter destinations are adjacent inside every
// digit bucket) behind the same L2, so partially written lines merge there.
// Only a speed hint: any placement gives the same result.
__device__ __forceinline__ u32 xcd_range_of_block(u32 b, u32 num_blocks)
{
    const u32 per = num_blocks >> 3;   // num_blocks is a multiple of 8
    return (b & 7u) * per + (b >> 3);
}

// Unaligned little-endian 8-byte load from byte address p (two aligned dwords
// pairs + funnel shift; never reads past p+8 rounded up to the next dword).
__device__ __forceinline__ u64 load_u64_unaligned(const u8 *p)
{
    const uintptr_t a = (uintptr_t)p;
    const u32 *q = (const u32 *)(a & ~(uintptr_t)3);
    const u32 sh = (u32)(a & 3) * 8;
    const u32 w0 = q[0], w1 = q[1];
    if (sh == 0) return (u64)w0 | ((u64)w1 << 32);
    const u32 w2 = q[2];
    const u32 lo = (w0 >> sh) | (w1 << (32 - sh));
    const u32 hi = (w1 >> sh) | (w2 << (32 - sh));
    return (u64)lo | ((u64)hi << 32);
}

}  // namespace pss
