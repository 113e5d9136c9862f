// Micro-benchmark behind DESIGN 4.2 (round-5 review, item 1: "single-pass partition primitive, decoupled look-back").
//
// What it models: the SECOND partition pass of the `lines` build in LSD order -- the input is already grouped by the
// second 10-bit digit d (1024 "d-regions"), the pass partitions by the FIRST digit b, and tiles must arrive in every
// b-bucket in tile order (a tile lies inside one d-region, so tile order IS d order: the joint buckets (b, d) come
// out contiguous without the order inside a tile mattering).  The offset of (tile, bin) is
//      start[b] + sum over the tiles before it of their count of b
// -- today that sum is a table made by a histogram pass that reads all 8 n bytes once more (msd_hist_kernel<false>,
// 0.85 ms at n = 2^29).  Here the tile counts its own 16384 elements (it must anyway, to rank them), publishes the
// 1024 counts, and adds up what its predecessors have published: decoupled look-back.
//
// The variable that decides whether it works is the DEPTH of the look-back: a tile retires every ~80 ns (32768 tiles
// in ~2.6 ms) and a device-scope round trip takes a microsecond or two, so on ONE chain over all tiles a newcomer finds
// dozens of predecessors that have counted but not yet finished their own look-back, and reads 4 KiB of counts from
// each.  The array is therefore cut into C independent CHAINS (C consecutive groups of d-regions; the counts of b per
// chain come from the text histogram pass for nothing: C x 1024 counters) and the ticket order interleaves them:
// neighbours on a chain are C tickets = C x 80 ns apart.
//
// Modes timed, all over the same n = 2^29 elements of 8 bytes, 16384-element tiles, 1024 threads:
//   table   today's structure: histogram pass + scan + scatter from a table of per-tile offsets
//   lb<C>   one pass, look-back on C chains (C = 1, 4, 16, 64), window of 4 predecessors per trip
// and per mode: time, mean / max look-back depth (rows of predecessors read per tile), result check.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lb tests/tools/lookback_micro.hip && /tmp/lb
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint16_t u16;

constexpr u32 BINS = 1024, BLOCK = 1024, IPT = 16, TILE = BLOCK * IPT, PIECE = 8192;
constexpr int SHIFT = 51;          // digit b = bits 51..60 of the element
constexpr u32 ST_A = 1u << 30, ST_P = 2u << 30, ST_MASK = 3u << 30, VAL_MASK = ST_A - 1u;

__device__ __forceinline__ u32 mix(u64 x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (u32)x;
}
// ~400 of the 1024 values of b occur (as the first two symbols of `lines` do): b = 2.5 * (hash % 400)
__global__ __launch_bounds__(256) void fill(u64 *e, u64 n)
{
    for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const u32 h = mix(i);
        const u32 b = ((h % 400u) * 5u) >> 1;
        e[i] = ((u64)b << SHIFT) | ((u64)((h >> 10) & 0x3fffffu) << 29) | (i & ((1ull << 29) - 1ull));
    }
}

// ---- today: histogram pass, scan, scatter from the table -------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void hist_kernel(const u64 *in, u32 n, u32 *T)
{
    __shared__ u32 hist[BINS];
    const u32 tid = threadIdx.x, t = blockIdx.x, base = t * TILE;
    hist[tid] = 0;
    __syncthreads();
#pragma unroll
    for (u32 k = 0; k < IPT; ++k) {
        const u32 p = k * BLOCK + tid;
        if (base + p < n) atomicAdd(&hist[(u32)(in[base + p] >> SHIFT) & (BINS - 1u)], 1u);
    }
    __syncthreads();
    T[(size_t)t * BINS + tid] = hist[tid];
}
// thread = bin: exclusive running sum over the tiles of [t0, t1) starting at base[bin]
__global__ __launch_bounds__(BINS) void scan_kernel(u32 *T, u32 t0, u32 t1, const u32 *base, u32 *total)
{
    const u32 b = threadIdx.x;
    u32 run = base ? base[b] : 0u;
    for (u32 t = t0; t < t1; ++t) {
        const u32 c = T[(size_t)t * BINS + b];
        T[(size_t)t * BINS + b] = run;
        run += c;
    }
    if (total) total[b] = run;
}
__global__ __launch_bounds__(BINS) void starts_kernel(const u32 *total, u32 *start)
{
    __shared__ u32 s[BINS];
    const u32 b = threadIdx.x;
    s[b] = total[b];
    __syncthreads();
    if (b == 0) {
        u32 run = 0;
        for (u32 i = 0; i < BINS; ++i) { const u32 c = s[i]; s[i] = run; run += c; }
    }
    __syncthreads();
    start[b] = s[b];
}
__global__ __launch_bounds__(BINS) void add_starts_kernel(u32 *T, u32 nt, const u32 *start)
{
    const u32 b = threadIdx.x;
    for (u32 t = blockIdx.x; t < nt; t += gridDim.x) T[(size_t)t * BINS + b] += start[b];
}

__device__ __forceinline__ u32 block_excl_sum(u32 v, u32 *scr)
{
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u32 incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = __shfl_up(incl, o);
        if (lane >= (u32)o) incl += t;
    }
    if (lane == 63) scr[wave] = incl;
    __syncthreads();
    u32 base = 0;
#pragma unroll
    for (u32 w = 0; w < BLOCK / 64; ++w)
        if (w < wave) base += scr[w];
    __syncthreads();
    return base + incl - v;
}

struct LbArgs {
    const u64 *in;
    u64 *out;
    u32 n, nt;
    const u32 *T;          // table mode: absolute offset of (tile, bin)
    // look-back mode
    u32 chains, tiles_per_chain;
    const u32 *base;       // [chains][BINS] absolute start of (chain, bin)
    u32 *status;           // [nt][BINS]
    u32 *ticket;
    u32 *depth;            // [nt] rows read by the tile's look-back
    u32 *J;                // [BINS][BINS] joint table: the last tile of every d-region leaves its inclusive prefixes
    u32 tiles_per_region;
};

// One tile: load, rank with returning LDS atomics, offsets (table or look-back), stage through LDS in two pieces, write.
// W = predecessors read per look-back trip.  ORDER: where the look-back sits in the tile's schedule --
//   0  publish, look back (all trips), then scan the bins and stage
//   1  publish, first trip's loads issued, scan the bins, then the look-back consumes them (and goes on if it must)
//   2  ... and the first piece is staged into LDS before the look-back is consumed (only the OUTPUT needs the offsets)
// The ticket is taken right before the tile's loads (taking the next ticket early -- to prefetch its elements -- makes
// the time between a ticket and its counts depend on how long the previous tile waited, and the chain behind it waits
// for the slowest: measured, 2.77 vs 2.48 ms).
template <bool LOOKBACK, int W, int ORDER>
__global__ __launch_bounds__(BLOCK) void scatter_kernel(LbArgs a)
{
    __shared__ __attribute__((aligned(16))) u64 exch[PIECE];
    __shared__ u32 hist[BINS], s_delta[BINS];
    __shared__ u16 s_start[BINS];
    __shared__ u32 scr[BLOCK / 64 + 1];
    __shared__ u32 s_ticket;
    const u32 tid = threadIdx.x;
    hist[tid] = 0;
    __syncthreads();
    for (;;) {
        if (tid == 0) s_ticket = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const u32 k = s_ticket;
        if (k >= a.nt) break;
        u32 t = k, tc = k;
        if (LOOKBACK) {
            const u32 c = k % a.chains;
            tc = k / a.chains;
            t = c * a.tiles_per_chain + tc;
        }
        const u32 base = t * TILE;
        const u32 valid = min(TILE, a.n - base);
        u64 elem[IPT];
        u32 lp[IPT];
#pragma unroll
        for (u32 j = 0; j < IPT; ++j) {
            const u32 p = j * BLOCK + tid;
            elem[j] = p < valid ? a.in[base + p] : 0ull;
        }
#pragma unroll
        for (u32 j = 0; j < IPT; ++j) {
            const u32 p = j * BLOCK + tid;
            const u32 d = (u32)(elem[j] >> SHIFT) & (BINS - 1u);
            lp[j] = p < valid ? (atomicAdd(&hist[d], 1u) | (d << 22)) : 0xffffffffu;
        }
        __syncthreads();                                    // (A) counts complete; previous tile fully written out
        const u32 c = hist[tid];
        u32 *row = a.status + (size_t)t * BINS;
        u32 off = 0, rows = 0;
        u32 v[W];
        u32 p = tc;                                         // predecessors of the chain not yet added: the next one is p - 1
        auto ask = [&]() {
#pragma unroll
            for (u32 w = 0; w < (u32)W; ++w) {
                const u32 q = p > w ? p - 1 - w : 0u;
                v[w] = __hip_atomic_load(&a.status[(size_t)(t - tc + q) * BINS + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        auto lookback = [&](bool asked) {
            if (tc == 0) {
                off = a.base[(size_t)(t / a.tiles_per_chain) * BINS + tid];
            } else {
                u32 sum = 0;
                for (;;) {
                    if (!asked) ask();
                    asked = false;
                    const u32 p0 = p;
                    ++rows;
                    bool stop = false, done = false;
#pragma unroll
                    for (u32 w = 0; w < (u32)W; ++w) {
                        if (!stop && w < p0) {
                            const u32 st = v[w] & ST_MASK;
                            if (st == 0) {
                                stop = true;              // not published yet: ask again from here
                            } else {
                                sum += v[w] & VAL_MASK;
                                --p;
                                if (st == ST_P) done = stop = true;
                            }
                        }
                    }
                    if (done || p == 0) break;
                }
                off = sum;
            }
            __hip_atomic_store(&row[tid], ST_P | (off + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.depth) {
                u32 m = rows;                             // the slowest lane of the tile
#pragma unroll
                for (int o = 32; o; o >>= 1) m = max(m, (u32)__shfl_xor((int)m, o));
                if ((tid & 63u) == 0) atomicMax(&a.depth[t], m);
            }
            // the last tile of a d-region: its inclusive prefixes are the ends of the joint buckets (b, d)
            if ((t + 1) % a.tiles_per_region == 0) a.J[(size_t)tid * BINS + t / a.tiles_per_region] = off + c;
        };
        if (LOOKBACK) {
            if (tc != 0) __hip_atomic_store(&row[tid], ST_A | c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ORDER == 0) lookback(false);
            else if (tc != 0) ask();
        } else {
            off = a.T[(size_t)t * BINS + tid];
        }
        const u32 ex = block_excl_sum(c, scr);
        s_start[tid] = (u16)ex;
        hist[tid] = 0;
        if (LOOKBACK && ORDER == 1) lookback(true);
        if (!LOOKBACK || ORDER != 2) s_delta[tid] = off - ex;
        __syncthreads();                                    // (B) bin starts published
#pragma unroll
        for (u32 j = 0; j < IPT; ++j)
            if (lp[j] != 0xffffffffu) lp[j] = (lp[j] & 0xffffu) + (u32)s_start[lp[j] >> 22];
#pragma unroll
        for (u32 h = 0; h < TILE / PIECE; ++h) {
            if (h * PIECE >= valid) break;
            if (h) __syncthreads();
#pragma unroll
            for (u32 j = 0; j < IPT; ++j) {
                const u32 q = lp[j] - h * PIECE;
                if (q < PIECE) exch[q] = elem[j];
            }
            if (LOOKBACK && ORDER == 2 && h == 0) {
                lookback(true);
                s_delta[tid] = off - ex;
            }
            __syncthreads();                                // (C) the piece in bin order
#pragma unroll
            for (u32 j = 0; j < PIECE / BLOCK; ++j) {
                const u32 q = j * BLOCK + tid, pp = h * PIECE + q;
                if (pp < valid) {
                    const u64 e = exch[q];
                    const u32 d = (u32)(e >> SHIFT) & (BINS - 1u);
                    a.out[s_delta[d] + pp] = e;
                }
            }
        }
    }
}

// The occupancy question: the 1024-thread kernel above runs ONE workgroup per CU (~100 VGPRs), so nothing overlaps its
// barrier phases.  The same tile by 512 threads x 32 elements leaves room for TWO workgroups per CU (128 VGPRs each, 76
// KiB of LDS each).  Table mode only: does the pass get faster?
template <int LPACK>
__global__ __launch_bounds__(512, 4) void scatter512_kernel(LbArgs a)
{
    constexpr u32 BLK = 512, IPT5 = TILE / BLK;       // 32
    __shared__ __attribute__((aligned(16))) u64 exch[PIECE];
    __shared__ u32 hist[BINS], s_delta[BINS];
    __shared__ u16 s_start[BINS];
    __shared__ u32 scr[BLK / 64 + 1];
    __shared__ u32 s_ticket;
    const u32 tid = threadIdx.x;
    hist[2 * tid] = hist[2 * tid + 1] = 0;
    __syncthreads();
    for (;;) {
        if (tid == 0) s_ticket = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const u32 t = s_ticket;
        if (t >= a.nt) break;
        const u32 base = t * TILE;
        const u32 valid = min(TILE, a.n - base);
        u64 elem[IPT5];
        u16 lp[IPT5];
#pragma unroll
        for (u32 j = 0; j < IPT5; ++j) {
            const u32 p = j * BLK + tid;
            elem[j] = p < valid ? a.in[base + p] : 0ull;
        }
#pragma unroll
        for (u32 j = 0; j < IPT5; ++j) {
            const u32 p = j * BLK + tid;
            const u32 d = (u32)(elem[j] >> SHIFT) & (BINS - 1u);
            lp[j] = p < valid ? (u16)atomicAdd(&hist[d], 1u) : (u16)0xffffu;
        }
        __syncthreads();                                    // (A)
        {
            const u32 c0 = hist[2 * tid], c1 = hist[2 * tid + 1];
            // exclusive sum over pairs of bins
            const u32 lane = tid & 63u, wave = tid >> 6;
            u32 incl = c0 + c1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const u32 x = __shfl_up(incl, o);
                if (lane >= (u32)o) incl += x;
            }
            if (lane == 63) scr[wave] = incl;
            __syncthreads();
            u32 bs = 0;
#pragma unroll
            for (u32 w = 0; w < BLK / 64; ++w)
                if (w < wave) bs += scr[w];
            const u32 ex = bs + incl - (c0 + c1);
            s_start[2 * tid] = (u16)ex;
            s_start[2 * tid + 1] = (u16)(ex + c0);
            s_delta[2 * tid] = a.T[(size_t)t * BINS + 2 * tid] - ex;
            s_delta[2 * tid + 1] = a.T[(size_t)t * BINS + 2 * tid + 1] - (ex + c0);
            hist[2 * tid] = hist[2 * tid + 1] = 0;
        }
        __syncthreads();                                    // (B)
#pragma unroll
        for (u32 j = 0; j < IPT5; ++j) {
            const u32 d = (u32)(elem[j] >> SHIFT) & (BINS - 1u);
            if (lp[j] != 0xffffu) lp[j] = (u16)(lp[j] + s_start[d]);
        }
#pragma unroll
        for (u32 h = 0; h < TILE / PIECE; ++h) {
            if (h * PIECE >= valid) break;
            if (h) __syncthreads();
#pragma unroll
            for (u32 j = 0; j < IPT5; ++j) {
                const u32 q = (u32)lp[j] - h * PIECE;
                if (lp[j] != 0xffffu && q < PIECE) exch[q] = elem[j];
            }
            __syncthreads();                                // (C)
#pragma unroll
            for (u32 j = 0; j < PIECE / BLK; ++j) {
                const u32 q = j * BLK + tid, pp = h * PIECE + q;
                if (pp < valid) {
                    const u64 e = exch[q];
                    const u32 d = (u32)(e >> SHIFT) & (BINS - 1u);
                    a.out[s_delta[d] + pp] = e;
                }
            }
        }
    }
}

// ---- check: every element in the region of its digit, regions in d order, nothing lost -------------------------------
__global__ __launch_bounds__(256) void check_kernel(const u64 *out, u32 n, const u32 *start, u32 region_elems, u32 *bad, u64 *sum)
{
    u64 acc = 0;
    for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        const u64 e = out[i];
        const u32 b = (u32)(e >> SHIFT) & (BINS - 1u);
        const u32 lo = start[b], hi = b + 1 < BINS ? start[b + 1] : n;
        if (i < lo || i >= hi) atomicAdd(&bad[0], 1u);
        if (i > lo) {
            const u64 p = out[i - 1];
            const u32 dp = (u32)(p & ((1ull << 29) - 1ull)) / region_elems, dc = (u32)(e & ((1ull << 29) - 1ull)) / region_elems;
            if (dp > dc) atomicAdd(&bad[1], 1u);
        }
        acc += (u64)mix(e);
    }
    atomicAdd((unsigned long long *)sum, (unsigned long long)acc);
}

int main(int argc, char **argv)
{
    const u32 n = argc > 1 ? (u32)strtoul(argv[1], nullptr, 0) : (1u << 29);
    const u32 nt = n / TILE;
    const u32 region_elems = n / BINS;                     // d-regions of equal size: region = position / region_elems
    const u32 tiles_per_region = region_elems / TILE;
    if (n % (BINS * TILE)) { printf("n must be a multiple of %u\n", BINS * TILE); return 1; }
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    printf("%s, %d CUs; n = %u, %u tiles of %u, %u threads per workgroup\n", prop.gcnArchName, prop.multiProcessorCount, n, nt, TILE, BLOCK);
    u64 *in, *out;
    u32 *T, *status, *ticket, *depth, *J, *total, *start, *base, *bad;
    u64 *sum;
    CK(hipMalloc(&in, (size_t)n * 8));
    CK(hipMalloc(&out, (size_t)n * 8));
    CK(hipMalloc(&T, (size_t)nt * BINS * 4));
    CK(hipMalloc(&status, (size_t)nt * BINS * 4));
    CK(hipMalloc(&ticket, 64));
    CK(hipMalloc(&depth, (size_t)nt * 4));
    CK(hipMalloc(&J, (size_t)BINS * BINS * 4));
    CK(hipMalloc(&total, BINS * 4));
    CK(hipMalloc(&start, BINS * 4));
    CK(hipMalloc(&base, 64 * BINS * 4));
    CK(hipMalloc(&bad, 64));
    CK(hipMalloc(&sum, 8));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, in, (u64)n);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto check = [&](const char *what, u64 *ref_sum) -> int {
        CK(hipMemset(bad, 0, 64));
        CK(hipMemset(sum, 0, 8));
        hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, 0, (const u64 *)out, n, (const u32 *)start, region_elems, bad, sum);
        u32 hb[2];
        u64 hs;
        CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&hs, sum, 8, hipMemcpyDeviceToHost));
        if (*ref_sum == 0) *ref_sum = hs;
        printf("    %s: misplaced %u, out of d order %u, checksum %s\n", what, hb[0], hb[1], hs == *ref_sum ? "ok" : "DIFFERS");
        return 0;
    };
    u64 ref_sum = 0;
    const int reps = 5;
    typedef void (*Kern)(LbArgs);
    struct Variant { const char *name; Kern table, lb; u32 wgs_per_cu; };
    const Variant variants[] = {
        {"window 4, look back then scan", scatter_kernel<false, 4, 0>, scatter_kernel<true, 4, 0>, 1},
        {"window 4, asked before the scan, consumed after the first piece is staged", scatter_kernel<false, 4, 0>, scatter_kernel<true, 4, 2>, 1},
    };
    float ms_h = 0;
    for (int r = 0; r < reps; ++r) {
        float ms;
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(hist_kernel, dim3(nt), dim3(BLOCK), 0, 0, (const u64 *)in, n, T);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms_h = r ? std::min(ms_h, ms) : ms;
    }
    printf("histogram pass (what the look-back deletes): %.3f ms\n", ms_h);
    {
        hipLaunchKernelGGL(hist_kernel, dim3(nt), dim3(BLOCK), 0, 0, (const u64 *)in, n, T);
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(BINS), 0, 0, T, 0u, nt, (const u32 *)nullptr, total);
        hipLaunchKernelGGL(starts_kernel, dim3(1), dim3(BINS), 0, 0, (const u32 *)total, start);
        hipLaunchKernelGGL(add_starts_kernel, dim3(1024), dim3(BINS), 0, 0, T, nt, (const u32 *)start);
        for (u32 wgs = 1; wgs <= 2; ++wgs) {
            float ms_s = 0;
            for (int r = 0; r < reps; ++r) {
                float ms;
                CK(hipMemset(ticket, 0, 4));
                LbArgs a{};
                a.in = in; a.out = out; a.n = n; a.nt = nt; a.T = T; a.ticket = ticket;
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(scatter512_kernel<0>, dim3(wgs * (u32)prop.multiProcessorCount), dim3(512), 0, 0, a);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                ms_s = r ? std::min(ms_s, ms) : ms;
            }
            printf("512 threads x 32 elements, %u workgroup(s) per CU: scatter from the table %.3f ms\n", wgs, ms_s);
            if (check("512", &ref_sum)) return 1;
        }
    }
    for (const Variant &var : variants) {
    const u32 grid = var.wgs_per_cu * (u32)prop.multiProcessorCount;
    printf("---- %s: %u persistent workgroups ----\n", var.name, grid);
    // ---- table mode ----
    {
        float ms_s = 0;
        hipLaunchKernelGGL(hist_kernel, dim3(nt), dim3(BLOCK), 0, 0, (const u64 *)in, n, T);
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(BINS), 0, 0, T, 0u, nt, (const u32 *)nullptr, total);
        hipLaunchKernelGGL(starts_kernel, dim3(1), dim3(BINS), 0, 0, (const u32 *)total, start);
        hipLaunchKernelGGL(add_starts_kernel, dim3(1024), dim3(BINS), 0, 0, T, nt, (const u32 *)start);
        for (int r = 0; r < reps; ++r) {
            float ms;
            CK(hipMemset(ticket, 0, 4));
            LbArgs a{};
            a.in = in; a.out = out; a.n = n; a.nt = nt; a.T = T; a.ticket = ticket;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(var.table, dim3(grid), dim3(BLOCK), 0, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms_s = r ? std::min(ms_s, ms) : ms;
        }
        printf("table       scatter from the table %.3f ms  => histogram + scatter %.3f ms\n", ms_s, ms_h + ms_s);
        if (check("table", &ref_sum)) return 1;
    }
    // ---- look-back modes ----
    const u32 chain_counts[] = {1, 8};
    for (u32 C : chain_counts) {
        const u32 tiles_per_chain = nt / C;
        if (tiles_per_chain % tiles_per_region) { printf("lb%u: chains do not end at region borders, skipped\n", C); continue; }
        // base[c][b] = start[b] + elements of digit b in the chains before c (from per-tile counts)
        hipLaunchKernelGGL(hist_kernel, dim3(nt), dim3(BLOCK), 0, 0, (const u64 *)in, n, T);
        CK(hipMemcpy(base, start, BINS * 4, hipMemcpyDeviceToDevice));
        for (u32 c = 0; c < C; ++c)
            hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(BINS), 0, 0, T, c * tiles_per_chain, (c + 1) * tiles_per_chain,
                               (const u32 *)(base + (size_t)c * BINS), c + 1 < C ? base + (size_t)(c + 1) * BINS : total);
        CK(hipDeviceSynchronize());
        float best = 0, best_nodepth = 0;
        std::vector<u32> hd(nt);
        double mean = 0;
        u32 mx = 0;
        for (int r = 0; r < reps * 2; ++r) {
            const bool with_depth = r < reps;
            CK(hipMemset(ticket, 0, 4));
            CK(hipMemset(depth, 0, (size_t)nt * 4));
            if (r == 0) CK(hipMemset(out, 0xff, (size_t)n * 8));
            LbArgs a{};
            a.in = in; a.out = out; a.n = n; a.nt = nt; a.ticket = ticket;
            a.chains = C; a.tiles_per_chain = tiles_per_chain; a.base = base; a.status = status; a.depth = with_depth ? depth : nullptr;
            a.J = J; a.tiles_per_region = tiles_per_region;
            float ms;
            CK(hipMemsetAsync(status, 0, (size_t)nt * BINS * 4, 0));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(var.lb, dim3(grid), dim3(BLOCK), 0, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (with_depth) {
                best = r ? std::min(best, ms) : ms;
                CK(hipMemcpy(hd.data(), depth, (size_t)nt * 4, hipMemcpyDeviceToHost));
                double sm = 0;
                u32 m = 0;
                for (u32 tt = 0; tt < nt; ++tt) { sm += hd[tt]; m = std::max(m, hd[tt]); }
                mean = sm / nt;
                mx = m;
            } else {
                best_nodepth = r == reps ? ms : std::min(best_nodepth, ms);
            }
        }
        printf("lb%-3u       one pass %.3f ms (%.3f with the depth probe); look-back trips per tile: mean %.2f, max %u\n", C, best_nodepth, best,
               mean, mx);
        char name[32];
        snprintf(name, sizeof name, "lb%u", C);
        if (check(name, &ref_sum)) return 1;
        // the joint table against the bucket starts: J[b][d] = end of bucket (b, d)
        std::vector<u32> hj((size_t)BINS * BINS), hs(BINS);
        CK(hipMemcpy(hj.data(), J, hj.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs.data(), start, BINS * 4, hipMemcpyDeviceToHost));
        u32 jbad = 0;
        for (u32 bb = 0; bb < BINS; ++bb) {
            u32 prev = hs[bb];
            for (u32 d = 0; d < BINS; ++d) {
                if (hj[(size_t)bb * BINS + d] < prev) ++jbad;
                prev = hj[(size_t)bb * BINS + d];
            }
            if (prev != (bb + 1 < BINS ? hs[bb + 1] : n)) ++jbad;
        }
        if (jbad) printf("    joint table: %u inconsistencies\n", jbad);
    }
    }
    return 0;
}
