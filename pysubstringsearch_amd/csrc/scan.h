// scan.h -- device-wide exclusive sum over n values produced by a functor
// (u64 accumulation): per-block partial sums, scan of <= 1024 partials, apply.
// Header-only templates shared by search.hip and sa_build.hip.
#pragma once
#include "common.h"
#include "prims.h"

namespace pss {

constexpr u32 SC_BLOCK = 256;
constexpr u32 SC_MAX_BLOCKS = 1024;

__device__ __forceinline__ u64 block_excl_sum64(u64 v, u64 *scr, u64 *total)
{
    const u64 incl = wave_incl_sum64(v);
    if (lane_id() == kWave - 1) scr[wave_id()] = incl;
    __syncthreads();
    u64 base = 0, tot = 0;
    for (u32 w = 0; w < SC_BLOCK / kWave; ++w) {
        const u64 s = scr[w];
        if (w < (u32)wave_id()) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

struct InU32 {
    const u32 *p;
    __device__ u64 operator()(u64 i) const { return p[i]; }
};
template <typename In>
__global__ __launch_bounds__(SC_BLOCK) void scan_reduce_kernel(In in, u64 n, u64 per_block, u64 *partial)
{
    __shared__ u64 scr[SC_BLOCK / kWave];
    const u64 b0 = (u64)blockIdx.x * per_block, b1 = (b0 + per_block < n) ? b0 + per_block : n;
    u64 acc = 0;
    for (u64 i = b0 + threadIdx.x; i < b1; i += SC_BLOCK) acc += in(i);
    u64 tot;
    (void)block_excl_sum64(acc, scr, &tot);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

template <int DUMMY>
__global__ __launch_bounds__(1024) void scan_partials_kernel(u64 *partial, u32 nb, u64 *total)
{
    __shared__ u64 s[16];
    const u32 t = threadIdx.x, lane = lane_id(), w = wave_id();
    const u64 v = t < nb ? partial[t] : 0;
    const u64 incl = wave_incl_sum64(v);
    if (lane == 63) s[w] = incl;
    __syncthreads();
    u64 base = 0, tot = 0;
    for (u32 k = 0; k < 16; ++k) {
        if (k < w) base += s[k];
        tot += s[k];
    }
    if (t < nb) partial[t] = base + incl - v;
    if (t == 0) *total = tot;
}

template <typename In>
__global__ __launch_bounds__(SC_BLOCK) void scan_apply_kernel(In in, u64 n, u64 per_block, const u64 *partial,
                                                                u64 *out, const u64 *total)
{
    __shared__ u64 scr[SC_BLOCK / kWave];
    const u64 b0 = (u64)blockIdx.x * per_block, b1 = (b0 + per_block < n) ? b0 + per_block : n;
    u64 carry = partial[blockIdx.x];
    for (u64 base = b0; base < b1; base += SC_BLOCK) {
        const u64 i = base + threadIdx.x;
        const u64 v = i < b1 ? in(i) : 0;
        u64 tot;
        const u64 ex = block_excl_sum64(v, scr, &tot);
        if (i < b1) out[i] = carry + ex;
        carry += tot;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;   // out has n+1 slots
}

template <typename In>
inline int device_excl_scan(DeviceCtx *ctx, In in, u64 n, u64 *partial, u64 *d_total, u64 *out)
{
    u64 per_block = (n + SC_MAX_BLOCKS - 1) / SC_MAX_BLOCKS;
    per_block = round_up(per_block ? per_block : 1, SC_BLOCK);
    const u32 nb = (u32)((n + per_block - 1) / per_block);
    const u32 nbl = nb ? nb : 1;
    hipLaunchKernelGGL(scan_reduce_kernel<In>, dim3(nbl), dim3(SC_BLOCK), 0, ctx->stream, in, n, per_block, partial);
    hipLaunchKernelGGL(scan_partials_kernel<0>, dim3(1), dim3(1024), 0, ctx->stream, partial, nbl, d_total);
    hipLaunchKernelGGL(scan_apply_kernel<In>, dim3(nbl), dim3(SC_BLOCK), 0, ctx->stream, in, n, per_block, partial, out,
                       d_total);
    PSS_HIP(hipGetLastError());
    return PSS_OK;
}

}  // namespace pss
