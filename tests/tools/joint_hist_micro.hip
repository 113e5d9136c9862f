// Micro-benchmark behind docs/history/DESIGN_rounds_1-5.md 13 item 10 (round-4 review: "accumulate the joint 20-bit histogram while G1 scatters").
// What would be ADDED to the first scatter pass of the `lines` build is one atomic per element into a table of 2^20
// 32-bit counters (4 MiB, L2-resident); what would be SAVED is msd_hist_kernel<false> (0.84 ms at n = 2^29).  This
// program times exactly that addition in isolation, the cheapest way it can be done:
//   raw        one atomicAdd (no return) per element, bins as random as the second digit of a suffix is
//   wave-agg   equal bins inside a wavefront merged first (the match_digit ballots of the scatter kernels)
//   lds-agg    a workgroup's 8192 elements first counted in an LDS hash of 4096 slots, then flushed
// against a plain streaming read of the same 8-byte elements (what msd_hist_kernel<false> does).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/jh tests/tools/joint_hist_micro.hip && /tmp/jh
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (uint32_t)x;
}

// the elements a G1 workgroup holds: 8 bytes each, streamed; bin = 20 bits of the key
__global__ __launch_bounds__(256) void fill(uint64_t *e, uint64_t n)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) e[i] = ((uint64_t)mix(i) << 32) | (uint32_t)i;
}
__global__ __launch_bounds__(256) void stream_only(const uint64_t *e, uint64_t n, uint32_t *sink)
{
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) acc ^= (uint32_t)(e[i] >> 44);
    if (acc == 0x12345) sink[0] = acc;
}
__global__ __launch_bounds__(256) void raw(const uint64_t *e, uint64_t n, uint32_t *table)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        __hip_atomic_fetch_add(&table[e[i] >> 44], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void wave_agg(const uint64_t *e, uint64_t n, uint32_t *table)
{
    for (uint64_t i0 = blockIdx.x * 256ull; i0 < n; i0 += (uint64_t)gridDim.x * 256) {
        const uint64_t i = i0 + threadIdx.x;
        const bool live = i < n;
        const uint32_t bin = live ? (uint32_t)(e[i] >> 44) : 0xffffffffu;
        // peers with my bin: 20 rounds of ballots
        uint64_t peers = __ballot(live);
        for (int b = 0; b < 20; ++b) {
            const uint64_t m = __ballot((bin >> b) & 1u);
            peers &= ((bin >> b) & 1u) ? m : ~m;
        }
        const int lane = threadIdx.x & 63;
        if (live && (peers & ((1ull << lane) - 1ull)) == 0)
            __hip_atomic_fetch_add(&table[bin], (uint32_t)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ __launch_bounds__(1024) void lds_agg(const uint64_t *e, uint64_t n, uint32_t *table)
{
    constexpr uint32_t HS = 16384;      // 8192 elements, load factor 1/2
    __shared__ uint32_t key[HS], cnt[HS];
    for (uint64_t t0 = blockIdx.x * 8192ull; t0 < n; t0 += (uint64_t)gridDim.x * 8192) {
        for (uint32_t k = threadIdx.x; k < HS; k += 1024) { key[k] = 0xffffffffu; cnt[k] = 0; }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < 8192; k += 1024) {
            const uint64_t i = t0 + k;
            if (i >= n) break;
            const uint32_t bin = (uint32_t)(e[i] >> 44);
            uint32_t s = (bin * 2654435761u) >> 18;
            for (;;) {
                const uint32_t old = atomicCAS(&key[s], 0xffffffffu, bin);
                if (old == 0xffffffffu || old == bin) break;
                s = (s + 1) & (HS - 1);
            }
            atomicAdd(&cnt[s], 1u);
        }
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < HS; k += 1024)
            if (cnt[k]) __hip_atomic_fetch_add(&table[key[k]], cnt[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
}

int main()
{
    const uint64_t n = 1ull << 29;
    uint64_t *e;
    uint32_t *table;
    CK(hipMalloc(&e, n * 8));
    CK(hipMalloc(&table, (1u << 20) * 4 + 64));
    hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, e, n);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto run = [&](const char *name, auto launch) -> int {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipMemset(table, 0, (1u << 20) * 4));
            CK(hipEventRecord(a, 0));
            launch();
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        std::vector<uint32_t> h(1u << 20);
        CK(hipMemcpy(h.data(), table, h.size() * 4, hipMemcpyDeviceToHost));
        uint64_t sum = 0;
        for (uint32_t x : h) sum += x;
        printf("%-12s %8.3f ms   (table sum %llu)\n", name, best, (unsigned long long)sum);
        return 0;
    };
    run("stream_only", [&] { hipLaunchKernelGGL(stream_only, dim3(8192), dim3(256), 0, 0, e, n, table); });
    run("raw", [&] { hipLaunchKernelGGL(raw, dim3(8192), dim3(256), 0, 0, e, n, table); });
    run("wave_agg", [&] { hipLaunchKernelGGL(wave_agg, dim3(8192), dim3(256), 0, 0, e, n, table); });
    run("lds_agg", [&] { hipLaunchKernelGGL(lds_agg, dim3(4096), dim3(1024), 0, 0, e, n, table); });
    printf("n = 2^29 elements of 8 bytes, 2^20 bins; msd_hist_kernel<false> of the lines build: 0.84 ms\n");
    return 0;
}
