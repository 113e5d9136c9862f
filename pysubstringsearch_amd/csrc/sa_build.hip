// sa_build.hip -- suffix-array construction on gfx950 by prefix doubling over
// the device radix sort (radix_sort.hip).  Replaces libsais() as called from
// construct_suffix_array (reference src/lib.rs:24-40, contract
// src/libsais/libsais.h:57-65): same output, different algorithm -- libsais'
// induced sorting is a serial dependency chain (libsais.c:2105-2136,
// 4565-4585), this is a data-parallel sort whose SA is identical because the
// suffix array of a text is unique.
//
// Pipeline (all arrays resident in HBM, u32 indices, n < 2^31):
//
//   0. sa_symbols      which byte values occur (+ sampled counts)   (reads n)
//      sa_recode       text -> dense codes 1..sigma, 0 = "past the end"
//                      (b = bits(sigma) per symbol)                 (reads n, writes n)
//   1. initial sort    radix sort of all n suffixes by a 64-bit key made of their
//                      first h0 symbols, packed on the fly from the codes (the
//                      end-of-text code 0 sorts first, so a suffix that is a proper
//                      prefix of another sorts before it and keys are never
//                      ambiguous); h0 from the symbol statistics (choose_key_chars);
//                      the pass that finishes the sort writes SA in place
//   2. rerank          equal keys = one group; rank = 1 + SA position of the group
//                      head; suffixes in groups of size 1 are final, the rest is
//                      compacted into the ACTIVE list (SA slot, suffix, group rank)
//   3. rounds on the active list only, until it is empty (see the host loop):
//        sparse  few ties (lines): doubling, ranks from a hash table of the tied
//                suffixes + binary search in the sorted initial keys; no ISA
//        text    natural text: every group extended by the next 64/b symbols
//                packed from the text at offset h; no ranks, no ISA
//        rank    repetitive data / after text rounds stop paying: ISA built once,
//                key = ISA[i+h], h doubles (Larsson-Sadakane)
//      text and rank rounds share one machinery: groups of <= 512 members are
//      ranked in LDS (group_sort_kernel), larger ones go through two chained
//      stable radix sorts; a rank round falls back to one global (group, rank)
//      radix sort while large groups dominate.
//
// Wave-level work uses 64-bit ballots throughout: group heads and active
// flags are ballot masks, head positions come from msb(mask), compaction
// offsets from v_mbcnt.
#include "prims.h"
#include "msd_sort.h"
#include "rle_build.h"
#include "radix_sort.h"
#include "sa_build.h"
#include "scan.h"
#include "anchor_impl.h"

#include <algorithm>
#include <cmath>
#include <string>
#include <thread>

namespace pss {

// ---------------------------------------------------------------- alphabet --

// Are there copies in the text?  8192 positions, the 16 bytes at each hashed to 32 bits and put into an LDS table:
// *dups = positions whose fingerprint was there already.  Random text has none (false matches: 0.008 expected); a text
// that holds the same megabyte a few hundred times has dozens.  A first chunk whose symbol counts look like log lines is
// sent to the MSD sort without a sizing sample (below) -- unless this says that the bucket check will refuse it.
constexpr u32 DUP_SAMPLES = 8192, DUP_SLOTS = 16384;
__global__ __launch_bounds__(1024) void dup_screen_kernel(const u8 *T, u32 n, u32 *dups)
{
    __shared__ u32 table[DUP_SLOTS];
    for (u32 i = threadIdx.x; i < DUP_SLOTS; i += 1024) table[i] = 0;
    __syncthreads();
    u32 mine = 0;
    if (n >= 64) {
        const u32 stride = (n - 32) / DUP_SAMPLES;
        for (u32 k = threadIdx.x; k < DUP_SAMPLES; k += 1024) {
            u64 x = ((u64)k + 1) * 0x9E3779B97F4A7C15ull;
            x ^= x >> 29;
            x *= 0xBF58476D1CE4E5B9ull;
            x ^= x >> 32;
            const u32 pos = stride ? k * stride + (u32)(x % stride) : k % (n - 32);
            const u64 a = load_u64_unaligned(T + pos), b = load_u64_unaligned(T + pos + 8);
            u64 hsh = (a ^ (b * 0x9E3779B97F4A7C15ull)) * 0xD6E8FEB86659FD93ull;
            hsh ^= hsh >> 32;
            u32 f = (u32)hsh;
            if (f == 0) f = 1;
            u32 slot = (f * 2654435761u) >> (32 - 14);
            static_assert(DUP_SLOTS == (1u << 14), "14-bit slot");
            for (;;) {
                const u32 old = atomicCAS(&table[slot], 0u, f);
                if (old == 0u) break;
                if (old == f) { ++mine; break; }
                slot = (slot + 1) & (DUP_SLOTS - 1);
            }
        }
    }
    const u32 tot = wave_incl_sum(mine);
    if (lane_id() == kWave - 1 && tot) atomicAdd(dups, tot);
}

// present[c] = 1 for every byte value that occurs (exact); counts[c] += its
// occurrences inside a 1/16 sample of the 16-byte vectors (for the entropy
// estimate that sizes the initial key).
// *runs += maximal runs of equal bytes (positions whose byte differs from the one before, and position 0):
// texts made of long runs take the run-length path (rle_build.hip).
__global__ __launch_bounds__(256) void sa_symbols_kernel(const u8 *T, u32 n, u32 *present, u32 *counts, u32 *runs)
{
    __shared__ u32 seen[256];
    __shared__ u32 cnt[256];
    const u32 tid = threadIdx.x;
    seen[tid] = 0;
    cnt[tid] = 0;
    u32 nruns = 0;
    __syncthreads();
    const u32 nvec = n / 16;
    const uint4 *Tv = reinterpret_cast<const uint4 *>(T);
    const bool aligned = ((uintptr_t)T & 15) == 0;
    if (aligned) {
        auto take = [&](const uint4 v, u32 i) {
            const u32 w[4] = {v.x, v.y, v.z, v.w};
            const bool sample = (i & 15u) == 0;
            // byte before the vector: the neighbouring lane holds it (lane k - 1 reads vector i - 1), lane 0 loads it
            u32 prev = __shfl_up(v.w >> 24, 1);
            if (lane_id() == 0) prev = i ? (u32)T[i * 16 - 1] : (~v.x & 0xffu);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32 diff = w[k] ^ ((w[k] << 8) | prev);       // byte j: T[j] ^ T[j - 1]
                prev = w[k] >> 24;
                nruns += (u32)__popc((((diff & 0x7f7f7f7fu) + 0x7f7f7f7fu) | diff) & 0x80808080u);
#pragma unroll
                for (int s = 0; s < 32; s += 8) {
                    const u32 c = (w[k] >> s) & 0xffu;
                    if (!seen[c]) seen[c] = 1;
                    if (sample) atomicAdd(&cnt[c], 1u);
                }
            }
        };
        // four loads in flight per thread: one 16-byte load per trip to HBM leaves the kernel at 2.8 TB/s
        const u32 stride = gridDim.x * blockDim.x;
        u32 i = blockIdx.x * blockDim.x + tid;
        // (the bound is taken at the wave's last lane: all its lanes leave this loop together, so the neighbouring
        // lane keeps holding the neighbouring vector in the loop below)
        for (; (u64)(i - lane_id() + kWave - 1) + 3ull * stride < nvec; i += 4 * stride) {
            const uint4 v0 = Tv[i], v1 = Tv[i + stride], v2 = Tv[i + 2 * stride], v3 = Tv[i + 3 * stride];
            take(v0, i);
            take(v1, i + stride);
            take(v2, i + 2 * stride);
            take(v3, i + 3 * stride);
        }
        for (; i < nvec; i += stride) take(Tv[i], i);
    }
    const u32 tail0 = aligned ? nvec * 16 : 0;
    for (u32 i = tail0 + blockIdx.x * blockDim.x + tid; i < n; i += gridDim.x * blockDim.x) {
        const u32 c = T[i];
        if (!seen[c]) seen[c] = 1;
        atomicAdd(&cnt[c], 1u);
        if (i == 0 || (u32)T[i - 1] != c) ++nruns;
    }
    __syncthreads();
    if (seen[tid]) present[tid] = 1;
    if (cnt[tid]) atomicAdd(&counts[tid], cnt[tid]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nruns += __shfl_xor(nruns, o);
    if (lane_id() == 0 && nruns) atomicAdd(runs, nruns);
}

// codes[i] = lut[T[i]] for i < n, 0 for n <= i < n_pad (n_pad % 16 == 0).
__global__ __launch_bounds__(256) void sa_recode_kernel(const u8 *T, u32 n, u32 n_pad, const u8 *lut, u8 *codes)
{
    __shared__ u8 s_lut[256];
    s_lut[threadIdx.x] = lut[threadIdx.x];
    __syncthreads();
    const u32 nvec = n_pad / 16;
    const bool aligned = ((uintptr_t)T & 15) == 0;
    for (u32 v = blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += gridDim.x * blockDim.x) {
        const u32 i0 = v * 16;
        u32 w[4] = {0, 0, 0, 0};
        if (aligned && i0 + 16 <= n) {
            const uint4 x = reinterpret_cast<const uint4 *>(T)[v];
            w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u32 o = 0;
#pragma unroll
                for (int s = 0; s < 32; s += 8) o |= (u32)s_lut[(w[k] >> s) & 0xffu] << s;
                w[k] = o;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u32 o = 0;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const u32 i = i0 + k * 4 + s;
                    const u32 c = (i < n) ? (u32)s_lut[T[i]] : 0u;
                    o |= c << (s * 8);
                }
                w[k] = o;
            }
        }
        reinterpret_cast<uint4 *>(codes)[v] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

#include "sa_rerank_impl.h"

#include "sa_rounds_impl.h"

#include "sa_plan_impl.h"

#include "sa_refine_impl.h"

// Suffix array of an INTEGER string of m symbols (rle_build.hip: one symbol per run of the text).  K[cur] / V[cur]:
// the (symbol key, index) pairs sorted by key; both buffer pairs hold m elements and are scratch afterwards.
// The end of the string is smaller than every symbol.  SA_out: m entries.  st: rounds / passes are added.
int suffix_rounds_integer(DeviceCtx *ctx, uint32_t m, uint64_t *K[2], uint32_t *V[2], int cur, uint32_t *SA_out,
                          pss_sa_stats *st)
{
    const Knobs knobs = Knobs::read();
    PSS_TRY(ctx->slot[S_ISA].reserve((size_t)m * 4 + 64));
    PSS_TRY(ctx->slot[S_P0].reserve((size_t)m * 4));
    PSS_TRY(ctx->slot[S_P1].reserve((size_t)m * 4));
    PSS_TRY(ctx->slot[S_GRP].reserve((size_t)m * 4));
    const size_t sort_ws = radix_sort_workspace_bytes();
    PSS_TRY(ctx->slot[S_WORK].reserve(sort_ws + 65536));
    u8 *work = ctx->slot[S_WORK].as<u8>();
    u8 *small = work + sort_ws;                       // the same 64 KiB of small device state as in sa_build_device
    RoundsIO io;
    io.n = m;
    io.SA = SA_out;
    io.K[0] = K[0]; io.K[1] = K[1];
    io.V[0] = V[0]; io.V[1] = V[1];
    io.ISA = ctx->slot[S_ISA].as<u32>();
    io.P[0] = ctx->slot[S_P0].as<u32>(); io.P[1] = ctx->slot[S_P1].as<u32>();
    io.GRP = ctx->slot[S_GRP].as<u32>();
    io.codes = nullptr;
    io.b = 8; io.plus_one = 0; io.key_chars = 1; io.key_drop = 0;
    io.h0 = 1;
    io.cur = cur;
    io.final_buf = -1;
    io.v_scratch = nullptr;
    io.ties = false;
    io.msd_fused = false;
    io.msd_active = 0;
    io.no_sparse = false;
    io.work = work;
    io.d_agg_head = reinterpret_cast<u32 *>(small + 4096);
    io.d_agg_cnt = reinterpret_cast<u32 *>(small + 8192);
    io.d_red = reinterpret_cast<u64 *>(small + 12288);
    io.d_counters = reinterpret_cast<u32 *>(small + 12288 + 64);
    io.h_small = static_cast<u32 *>(ctx->pinned);
    io.profile = false;
    SortStats ss;
    pss_sa_stats local;
    memset(&local, 0, sizeof local);
    PSS_TRY(refine_rounds(ctx, knobs, io, ss, st ? *st : local));
    return PSS_OK;
}

int sa_build_device(DeviceCtx *ctx, const void *d_T, void *d_SA, int32_t n_in, uint32_t flags, pss_sa_stats *stats)
{
    pss_sa_stats st;
    memset(&st, 0, sizeof st);
    if (n_in < 0 || (n_in > 0 && (d_T == nullptr || d_SA == nullptr))) {
        set_error("pss_sa_build: bad arguments");
        return PSS_EINVAL;
    }
    const u32 n = (u32)n_in;
    const bool profile = flags & 1u;
    if (flags & 8u) {                // a cold build: nothing of earlier builds on this device is used
        ctx->plan_path = 0;
        ctx->ss_plan_skip = ctx->ss_plan_backoff = 0;
        ctx->side_plan_heff = 0;
        flags &= ~8u;                // (a restart of THIS build keeps what it has learnt)
    }
    const Knobs knobs = Knobs::read();
    hipStream_t s = ctx->stream;
    if (n < 2) {
        if (n == 1) PSS_HIP(hipMemsetAsync(d_SA, 0, 4, s));
        PSS_HIP(hipStreamSynchronize(s));
        if (stats) *stats = st;
        return PSS_OK;
    }
    const u8 *T = static_cast<const u8 *>(d_T);
    u32 *SA = static_cast<u32 *>(d_SA);

    const size_t n_pad = round_up((size_t)n, 16) + 64;
    PSS_TRY(ctx->slot[S_CODES].reserve(n_pad));
    PSS_TRY(ctx->slot[S_K0].reserve((size_t)n * 8));
    PSS_TRY(ctx->slot[S_K1].reserve((size_t)n * 8));
    PSS_TRY(ctx->slot[S_V0].reserve((size_t)n * 4));
    PSS_TRY(ctx->slot[S_V1].reserve((size_t)n * 4));
    PSS_TRY(ctx->slot[S_ISA].reserve((size_t)n * 4 + 64));
    PSS_TRY(ctx->slot[S_P0].reserve(std::max((size_t)n * 4, msd_workspace_bytes(n))));   // also the tables of the MSD sort
    PSS_TRY(ctx->slot[S_P1].reserve((size_t)n * 4));
    PSS_TRY(ctx->slot[S_GRP].reserve((size_t)n * 4));
    const size_t sort_ws = radix_sort_workspace_bytes();
    PSS_TRY(ctx->slot[S_WORK].reserve(sort_ws + 65536));
    u8 *work = ctx->slot[S_WORK].as<u8>();
    u8 *small = work + sort_ws;                       // 64 KiB of small device state
    u32 *d_present = reinterpret_cast<u32 *>(small);              // [256] presence, [256] sampled counts
    u8 *d_lut = small + 2048;                                     // [256]
    u32 *d_agg_head = reinterpret_cast<u32 *>(small + 4096);      // [1024]
    u32 *d_agg_cnt = reinterpret_cast<u32 *>(small + 8192);       // [1024]
    u64 *d_red = reinterpret_cast<u64 *>(small + 12288);          // [2]
    u32 *d_counters = reinterpret_cast<u32 *>(small + 12288 + 64);

    u8 *codes = ctx->slot[S_CODES].as<u8>();
    u64 *K[2] = {ctx->slot[S_K0].as<u64>(), ctx->slot[S_K1].as<u64>()};
    u32 *V[2] = {ctx->slot[S_V0].as<u32>(), ctx->slot[S_V1].as<u32>()};
    u32 *ISA = ctx->slot[S_ISA].as<u32>();
    u32 *P[2] = {ctx->slot[S_P0].as<u32>(), ctx->slot[S_P1].as<u32>()};
    u32 *GRP = ctx->slot[S_GRP].as<u32>();
    u32 *h_small = static_cast<u32 *>(ctx->pinned);

    BuildTimer timer;
    PSS_HIP(hipEventCreate(&timer.ev0));
    PSS_HIP(hipEventCreate(&timer.ev1));
    PSS_HIP(hipEventCreate(&timer.ev_mid));
    PSS_HIP(hipEventRecord(timer.ev0, s));
    // A build that starts over (a plan that did not hold) reports the time of the attempts it gave up as well.
    if (ctx->restart_depth == 0) {
        ctx->restart_ms = 0.0;
        ctx->ss_refused_note = false;
    }
    SideAnchors side;
    auto start_over = [&](uint32_t new_flags) -> int {
        side.join();                         // (the next attempt has a side line of its own, in the same helper context)
        PSS_HIP(hipEventRecord(timer.ev1, s));
        PSS_HIP(hipStreamSynchronize(s));
        float gone = 0.f;
        PSS_HIP(hipEventElapsedTime(&gone, timer.ev0, timer.ev1));
        ctx->restart_ms += gone;
        ctx->restart_depth += 1;
        const int rc = sa_build_device(ctx, d_T, d_SA, n_in, new_flags, stats);
        ctx->restart_depth -= 1;
        return rc;
    };

    // ---- 0. alphabet ----
    const int grid_stream = ctx->num_cus * 8;
    u32 *d_runs = d_counters + 40;
    // The plan of the previous build on this device (see step 1): when it took the MSD sort on a text of this size class
    // and every switch is at its default, this build does not look at its alphabet first either -- it recodes with the
    // remembered byte -> code table inside the sort's first histogram pass (msd_hist_raw_kernel: the alphabet pass, the
    // recode pass and a host round trip fold into it, 0.4 ms at n = 2^29), which also checks that every byte has a code
    // there.  A text over a subset of the remembered alphabet is sorted correctly with the larger table; a new byte, a
    // crowded bucket, or anything else the sort declines for starts the build over without the plan -- the run-length
    // and periodic-text checks, which this shortcut skips, then take place as always.
    uint32_t logn = 0;
    while ((2u << logn) <= n && logn < 31) ++logn;
    const bool plain = knobs.key_chars == 0 && !knobs.no_sample && !knobs.no_flags && knobs.msd < 0 && knobs.ss < 0 &&
                       knobs.key_drop < 0 && knobs.mode < 0 && !knobs.no_msd_fuse && !knobs.no_plan && knobs.rle < 0 &&
                       knobs.period != 0 && (flags & 2u) == 0;
    const bool fronted = plain && !knobs.no_front && n >= (1u << 24) && ctx->plan_path == 1 && ctx->plan_logn == logn &&
                         (reinterpret_cast<uintptr_t>(T) & 15u) == 0;
    if (!fronted) {
        PSS_HIP(hipMemsetAsync(d_present, 0, 2048, s));
        PSS_HIP(hipMemsetAsync(d_runs, 0, 4, s));
        hipLaunchKernelGGL(sa_symbols_kernel, dim3(grid_stream), dim3(256), 0, s, T, n, d_present, d_present + 256, d_runs);
        const bool screen_dups = plain && !knobs.no_front && (flags & 4u) == 0 && n >= (1u << 24);      // (only the shortcut of a first chunk asks)
        PSS_HIP(hipMemsetAsync(d_runs + 1, 0, 4, s));
        if (screen_dups) hipLaunchKernelGGL(dup_screen_kernel, dim3(1), dim3(1024), 0, s, T, n, d_runs + 1);
        PSS_HIP(hipMemcpyAsync(h_small, d_present, 2048, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipMemcpyAsync(h_small + 512, d_runs, 8, hipMemcpyDeviceToHost, s));
        // (the head of the text rides along: does it repeat one word?  -- see below)
        u8 *h_head = static_cast<u8 *>(ctx->pinned) + 32768;         // the search path's query staging; builds and searches take turns
        const u32 head_len = std::min<u32>(n, kPeriodProbe);
        const bool look_for_period = knobs.period != 0 && n >= 4 * kPeriodProbe;
        if (look_for_period) PSS_HIP(hipMemcpyAsync(h_head, T, head_len, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
        // Long runs of equal bytes (every suffix inside a run is tied with its neighbours for as long as the run
        // lasts: the worst case of prefix doubling): sort the run heads as a string of one symbol per run, then
        // every other suffix falls into place with one short radix sort (rle_build.hip).
        const u32 text_runs = h_small[512];
        st.runs = text_runs;
        if (knobs.rle == 1 || (knobs.rle < 0 && n >= 4096 && (u64)text_runs * 8 <= (u64)n)) {
            RleStats rls;
            PSS_TRY(rle_suffix_array(ctx, T, n, text_runs, SA, profile, &rls, &st));
            st.rle = rls.columns ? 2 : 1;
            st.rle_id_bits = rls.id_bits;
            st.rle_ms_table = rls.ms_table;
            st.rle_ms_reduced = rls.ms_reduced;
            st.rle_ms_expand = rls.ms_expand;
            for (int c = 0; c < 256; ++c) st.sigma += h_small[c] ? 1u : 0u;
            PSS_HIP(hipEventRecord(timer.ev1, s));
            PSS_HIP(hipStreamSynchronize(s));
            float ms = 0.f;
            PSS_HIP(hipEventElapsedTime(&ms, timer.ev0, timer.ev1));
            st.ms_total = ms + ctx->restart_ms;
            st.ms_restarts = ctx->restart_ms;
            if (stats) *stats = st;
            return PSS_OK;
        }
        // One word written over and over (a text whose first m bytes have a small period, with at most a few bytes behind):
        // the worst case of prefix doubling that is not a run of one byte.  The head of the text says whether it is worth
        // a pass to find out how far the repetition goes; if it covers the text, the suffix array has a closed form
        // (rle_build.h).
        if (look_for_period) {
            const u32 p = period_of_head(h_head, head_len);
            if (p) {
                u32 m = 0;
                PSS_TRY(period_extent(ctx, T, n, p, &m));
                st.period = p;
                st.period_extent = m;
                bool accepted = false;
                if (n - m <= kPeriodTailMax) {
                    std::vector<u8> word(h_head, h_head + p);           // (period_extent reuses the pinned scratch)
                    PSS_TRY(period_suffix_array(ctx, T, n, p, m, word.data(), SA, &accepted));
                }
                if (accepted) {
                    st.period_path = 1;
                    for (int c = 0; c < 256; ++c) st.sigma += h_small[c] ? 1u : 0u;
                    PSS_HIP(hipEventRecord(timer.ev1, s));
                    PSS_HIP(hipStreamSynchronize(s));
                    float ms = 0.f;
                    PSS_HIP(hipEventElapsedTime(&ms, timer.ev0, timer.ev1));
                    st.ms_total = ms + ctx->restart_ms;
            st.ms_restarts = ctx->restart_ms;
                    if (stats) *stats = st;
                    return PSS_OK;
                }
            }
        }
    }
    u8 lut[256];
    u32 sigma = 0;
    if (fronted) {
        memcpy(lut, ctx->plan_lut, 256);
        sigma = ctx->plan_sigma;
    } else {
        for (int c = 0; c < 256; ++c) lut[c] = h_small[c] ? (u8)(++sigma) : 0;   // codes 1..sigma
    }
    int b = 1;
    while ((1u << b) <= sigma) ++b;               // codes 0..sigma need b bits
    int plus_one = 0;
    if (sigma == 256) {
        // 257 code points do not fit a byte: keep the raw bytes and let the key
        // packer add 1 to every in-text symbol (9-bit codes, 0 = past the end).
        b = 9;
        plus_one = 1;
        for (int c = 0; c < 256; ++c) lut[c] = (u8)c;
    }
    int kmax = 64 / b;
    if (kmax > 16) kmax = 16;
    // A first chunk (no plan yet) whose symbols are close to uniform -- log lines, identifiers, hex, base64: sum p^2 below
    // 0.035; natural language sits at 0.065 and above -- goes to the MSD sort the way a planned chunk does: the sort's
    // first histogram pass recodes the raw text with the table just built, no recode pass, no sizing sample (0.4 ms and a
    // host round trip: what the sample would say is what the symbol counts say already).  The sort's exact bucket check
    // still decides; a decline starts the build over along the long road (flags bit 2), at the price of the passes made.
    bool fresh = false;
    if (!fronted && plain && !knobs.no_front && (flags & 4u) == 0 && n >= (1u << 24) && sigma >= 2 && sigma < 256 &&
        (reinterpret_cast<uintptr_t>(T) & 15u) == 0) {
        double tot = 0, c2 = 0;
        for (int c = 0; c < 256; ++c) tot += h_small[256 + c];
        for (int c = 0; c < 256 && tot > 0; ++c) {
            const double pr = h_small[256 + c] / tot;
            c2 += pr * pr;
        }
        // ... and no copies among 8192 sampled places (dup_screen_kernel): a text of log lines that repeats itself would be
        // refused by the sort's bucket check after 3.5 ms of passes
        fresh = tot > 0 && c2 < 0.035 && h_small[513] < 8;
        st.dup_screen = h_small[513];
    }
    const bool front_any = fronted || fresh;
    int key_chars = front_any ? kmax : choose_key_chars(h_small + 256, n, b, kmax);
    const bool forced_chars = knobs.key_chars >= 1 && knobs.key_chars <= kmax;
    if (forced_chars) key_chars = knobs.key_chars;
    st.sigma = sigma;
    st.code_bits = (u32)b;
    st.key_chars = (u32)key_chars;
    memcpy(h_small + 1024, lut, 256);
    PSS_HIP(hipMemcpyAsync(d_lut, h_small + 1024, 256, hipMemcpyHostToDevice, s));
    u32 *d_bad = d_counters + 44;
    if (front_any) {
        // the codes are made by the sort (below); their padding past n and the flag for a byte without a code
        const size_t tail0 = (size_t)n & ~(size_t)15;
        PSS_HIP(hipMemsetAsync(codes + tail0, 0, n_pad - tail0, s));
        PSS_HIP(hipMemsetAsync(d_bad, 0, 4, s));
    } else {
        hipLaunchKernelGGL(sa_recode_kernel, dim3(grid_stream), dim3(256), 0, s, T, n, (u32)n_pad, d_lut, codes);
        // The previous chunk on this device sorted its anchors beside its text round (text with copies in it), and this
        // chunk will reach the same depth the same way (same size class, same bits per symbol, the sample sort's key of
        // as many symbols): its side line starts NOW, beside the initial sort as well.  Should the build take another
        // road -- no ties worth an anchor round, a shallower depth -- the result is thrown away (anchor_side = 2).
        if (plain && knobs.side != 0 && knobs.anchor != 0 && ctx->side_plan_heff != 0 && ctx->side_plan_logn == logn &&
            ctx->side_plan_b == b && n >= (1u << 24) && (flags & 2u) == 0) {
            int kt0 = 64 / b;
            if (kt0 > 16) kt0 = 16;
            const int kc0 = ss_key_chars(n, plus_one ? 257u : sigma + 1u);
            if (kc0 > 0 && (u64)kc0 + (u64)kt0 == ctx->side_plan_heff) {
                PSS_HIP(hipEventRecord(timer.ev_mid, s));         // (recorded again below, where the initial sort ends)
                PSS_TRY(side_start(ctx, knobs, side, n, codes, b, plus_one, ctx->side_plan_heff, timer.ev_mid));
            }
        }
    }

    // ---- 1. initial sort on the first key_chars symbols ----
    SortStats ss;
    int key_drop = 0;
    bool msd_screen_ok = false, sampled = false;
    // The plan of the previous build on this device, if it was for the same kind of text (same byte values present, same
    // size class) and every switch is at its default: 1 = it took the MSD sort.  The sizing sample (0.4 ms and a host
    // round trip at n = 2^29) would only repeat what it said then; the MSD sort's own exact bucket check still decides.
    // When it declines, the sample sort is next as always; if that declines too, the build starts over with the sample
    // (the LSD passes want the key length it measures).
    uint32_t present_bits[8] = {};
    for (int c = 0; c < 256; ++c)
        if (plus_one || lut[c]) present_bits[c >> 5] |= 1u << (c & 31);
    int hint = 0;
    if (front_any) hint = 1;
    else if (plain && n >= (1u << 24) && ctx->plan_path && ctx->plan_logn == logn && memcmp(ctx->plan_present, present_bits, 32) == 0)
        hint = ctx->plan_path;
    st.plan_hint = (uint64_t)(fronted ? 2 : (fresh ? 3 : hint));
    if (hint) {
        sampled = true;
        msd_screen_ok = hint == 1;
    } else if (n >= (1u << 24) && !forced_chars && knobs.key_chars == 0 && !knobs.no_sample) {
        PSS_TRY(size_initial_key(ctx, codes, n, b, kmax, plus_one, K, V, work, d_counters + 16, h_small, profile, &ss,
                                 &key_chars, &key_drop, &msd_screen_ok));
        st.key_chars = (u32)key_chars;
        sampled = true;
    }
    if (knobs.key_drop >= 0 && knobs.key_drop < b && key_chars > 1) key_drop = knobs.key_drop;
    TextKeys tk{codes, b, key_chars, plus_one, key_drop};
    const int key_bits0 = key_chars * b - key_drop;
    st.key_bits = (u64)key_bits0;
    // The sorted suffix indices of the initial sort ARE the suffix array (ties are reordered
    // later, inside their slots): let the pass that finishes the sort write straight into the
    // caller's SA buffer.  The text pass writes buffer 0 and the passes alternate, so the
    // buffer that receives the last pass is known up front.
    const int passes0 = (key_bits0 + 7) / 8;
    const int final_buf = (passes0 - 1) & 1;
    u32 *const v_scratch = V[final_buf];
    V[final_buf] = SA;
    int cur = 0;
    // With >= 2 passes the sort carries tie flags instead of consumed digits (radix_sort.hip, fs_*):
    // the last pass writes only the suffix indices with bit 31 = "tied with my predecessor", and
    // the rerank reads 4-byte flagged values instead of comparing 8-byte keys.
    const bool ties = passes0 >= 2 && !knobs.no_flags;
    // Hybrid MSD sort (msd_sort.hip): two global partition passes over 8-byte elements, then every joint
    // bucket sorted in LDS.  Taken when the key the sizing asked for (<= 48 bits) is covered by what an
    // element can carry, and the sorted sample shows no crowded 20-bit prefix; the exact bucket check
    // inside can still decline, then the LSD passes run as before.
    bool msd_done = false, msd_fused = false;
    u32 msd_active = 0;
    if (ties && knobs.msd != 0) {
        // (at n = 2^29 an element holds 45 key bits after the first pass; smaller texts leave room for up to 48)
        const char *cap_env = knob("PSS_MSD_KEY_CAP");
        const int key_cap = cap_env ? atoi(cap_env) : 48;
        // Whole symbols by default.  PSS_MSD_PARTIAL_SYMBOL=1 fills the element with the high bits of one more symbol (a monotone
        // coarsening, like the LSD path's; the depth the rounds start from counts the whole symbols only): `lines` at 2^29 then
        // sorts by 45 bits = 7 symbols + 3 bits of the 8th and leaves 0.42 M suffixes tied instead of 2.09 M -- which sends
        // them to the sparse mode's global passes (<= n / 1024 ties: eight launch-bound radix passes, 0.7 ms) where the text
        // round took 0.3 ms for five times as many: 8.55 vs 8.02 ms, measured in round 6 and left off.
        const bool whole = knob("PSS_MSD_PARTIAL_SYMBOL") == nullptr;
        int kb = std::min(msd_max_key_bits(n), std::max(key_cap, 21));
        int kc = std::min(whole ? kb / b : (kb + b - 1) / b, kmax);
        int mdrop = kc * b - kb;
        if (mdrop < 0 || kc <= 1) {                               // (kmax symbols do not fill the element: all of their bits)
            kb = kc * b;
            mdrop = 0;
        }
        const bool fits = kc >= 1 && kb >= 21 && !plus_one;
        const bool auto_ok = sampled && msd_screen_ok && (hint == 1 || (key_bits0 <= 48 && kb + 8 >= key_bits0));
        if (fits && (knobs.msd == 1 || (knobs.msd < 0 && auto_ok))) {
            TextKeys mk{codes, b, kc, plus_one, mdrop};
            MsdStats ms;
            bool accepted = false;
            // The local sort hands over the active list of the first rerank (SA slot, suffix, group rank of every
            // suffix tied with a neighbour) in the buffers round 0 would fill: P[1], the free value buffer, G[1].
            // Staging: the first element buffer (free once the second partition pass has read it).
            PSS_TRY(ctx->slot[S_GRP2].reserve((size_t)n * 4));
            MsdActive act;
            act.pos = ctx->slot[S_P1].as<u32>();
            act.idx = (final_buf == 0) ? V[1] : V[0];
            act.grp = ctx->slot[S_GRP2].as<u32>();
            act.st_pos = reinterpret_cast<u32 *>(K[0]);
            act.st_idx = reinterpret_cast<u32 *>(K[0]) + n;
            const MsdFront front{T, d_lut, d_bad};
            PSS_TRY(msd_suffix_sort(ctx, &mk, n, kb, K, SA, ctx->slot[S_P0].p, h_small, profile, &ms, &accepted,
                                    knobs.no_msd_fuse ? nullptr : &act, front_any ? &front : nullptr));
            if (front_any && !accepted) {
                // the plan did not hold for this text (a byte outside the remembered alphabet, a crowded bucket): all
                // over again without it -- the alphabet pass, the run-length and periodic-text checks, the sample
                // (a first chunk taken on its symbol counts alone: the same, with that shortcut switched off)
                ctx->plan_path = 0;
                return start_over(fresh ? (flags | 4u) : flags);
            }
            msd_fused = accepted && !knobs.no_msd_fuse;
            msd_active = act.count;
            st.msd_buckets = ms.buckets;
            st.msd_max_bucket = ms.max_bucket;
            if (accepted) {
                msd_done = true;
                key_chars = kc;
                key_drop = mdrop;
                st.key_chars = (u32)kc;
                st.key_bits = (u64)kb;
                st.msd = 1;
                st.msd_tiles = ms.tiles;
                st.msd_slow_tiles = ms.slow_tiles;
                st.msd_lookback = ms.lookback;
                st.msd_ms_g1 = ms.ms_g1;
                st.msd_ms_g2 = ms.ms_g2;
                st.msd_ms_local = ms.ms_local;
                cur = final_buf;                                 // V[final_buf] is the caller's SA buffer
                ss.launches = 3;
                ss.elems = 3ull * n;
            }
        }
    }
    if (front_any && !msd_done) {            // (the sort was not even tried: nothing has made the codes)
        ctx->plan_path = 0;
        return start_over(fresh ? (flags | 4u) : flags);
    }
    // Natural text (some 20-bit prefix holds far more suffixes than a tile, and a 64-bit key leaves most suffixes tied
    // anyway): sample sort over 16-byte [key | index] elements (ss_sort_impl.h) -- splitters from a sorted sample cut the
    // text's own distribution into tile-sized buckets, the key is twice as long.
    if (!msd_done && ties && knobs.ss != 0 && (knobs.ss == 1 || (n >= (1u << 24) && !forced_chars && knobs.key_chars == 0)) &&
        ss_sample_count(n) != 0) {
        const u32 S = ss_sample_count(n);
        // The two element buffers (16 n bytes each: 17 GB at n = 2^29) are the largest allocation of the build.  Where HBM
        // is short -- a Reader resident on the same device -- the sort DECLINES instead of failing the build: the LSD
        // passes below need nothing beyond the buffers every build has.
        bool ss_room = ctx->slot[S_SSA].reserve((size_t)n * 16 + 256) == PSS_OK && ctx->slot[S_SSB].reserve((size_t)n * 16 + 256) == PSS_OK;
        if (!ss_room) {
            (void)hipGetLastError();
            ctx->slot[S_SSA].release();
            ctx->slot[S_SSB].release();
            set_error("%s", "");
            st.ss_declined_nomem = 1;
        }
        if (ss_room) {
        PSS_TRY(ctx->slot[S_GRP2].reserve((size_t)n * 4));
        SsBuffers sb;
        sb.A[0] = ctx->slot[S_SSA].p;
        sb.A[1] = ctx->slot[S_SSB].p;
        sb.digits = reinterpret_cast<uint16_t *>(ctx->slot[S_P1].p);      // 2 n + 128 bytes of the 4 n
        sb.E0 = GRP;                                                       // 16 S <= 4 n bytes each
        sb.E = ISA;
        sb.K[0] = K[0]; sb.K[1] = K[1];
        sb.V[0] = v_scratch; sb.V[1] = (final_buf == 0) ? V[1] : V[0];     // (never the caller's SA buffer)
        sb.sort_work = work;
        // the plan: the previous chunk of this corpus (same size, same alphabet) left its sorted sample behind
        const u32 radix = plus_one ? 257u : sigma + 1u;
        // (the same sample size, bucket counts, index bits and symbols per key: chunks of one Writer differ by an entry or two)
        const u64 tag = ss_geometry_tag(n, radix);
        const bool planned = hint == 2 && tag != 0 && ctx->ss_plan_tag == tag && ctx->ss_plan_radix == radix && ctx->slot[S_SSPLAN].p != nullptr &&
                             ctx->ss_plan_skip == 0;
        if (planned) sb.sample_in = ctx->slot[S_SSPLAN].p;
        else if (plain && ctx->slot[S_SSPLAN].reserve((size_t)S * 16) == PSS_OK) sb.sample_keep = ctx->slot[S_SSPLAN].p;
        ctx->ss_plan_tag = 0;                                              // (valid again once this sort has been accepted)
        MsdActive act;
        act.pos = ctx->slot[S_P1].as<u32>();
        act.idx = (final_buf == 0) ? V[1] : V[0];
        act.grp = ctx->slot[S_GRP2].as<u32>();
        act.st_pos = reinterpret_cast<u32 *>(K[0]);
        act.st_idx = reinterpret_cast<u32 *>(K[0]) + n;
        TextKeys sk{codes, b, 0, plus_one, 0};
        SsStats sst;
        bool accepted = false;
        PSS_TRY(ss_suffix_sort(ctx, &sk, plus_one ? 257u : sigma + 1u, n, sb, SA, ctx->slot[S_P0].p, h_small, profile, &sst, &accepted,
                               &act));
        if (planned && !accepted) {
            // The previous chunk's splitters left a bucket beyond a tile (real files: millions of suffixes with one key --
            // blanks -- sit where the files put them, not where the last chunk had them).  The build starts over without the
            // plan, and the next chunks do not try it again at once: 1, 3, 7, ... builds go by first.
            ctx->ss_plan_backoff = std::min(63u, 2 * ctx->ss_plan_backoff + 1);
            ctx->ss_plan_skip = ctx->ss_plan_backoff;
            ctx->ss_refused_note = true;
            ctx->plan_path = 0;               // all over again without the plan: this may not be natural text at all
            return start_over(flags);
        } else if (accepted) {
            if (planned) ctx->ss_plan_backoff = 0;
            else if (hint == 2 && ctx->ss_plan_skip) --ctx->ss_plan_skip;      // (a chunk that went by without trying)
        }
        st.ss_buckets = sst.buckets;
        st.ss_max_bucket = sst.max_bucket;
        st.ss_samples = sst.samples;
        if (accepted) {
            msd_done = true;
            msd_fused = false;          // ties come back as flags in the suffix array (groups cross the tiles of this sort)
            key_chars = sst.key_chars;
            key_drop = 0;
            st.key_chars = (u32)key_chars;
            st.key_bits = (u64)key_chars * b;
            st.ss = 1;
            st.ss_tiles = sst.tiles;
            st.ss_ms_sample = sst.ms_sample;
            st.ss_ms_g1 = sst.ms_g1;
            st.ss_ms_g2 = sst.ms_g2;
            st.ss_ms_local = sst.ms_local;
            cur = final_buf;
            ss.launches = 4;
            ss.elems = 4ull * n;
            if (sb.sample_in || sb.sample_keep) {
                ctx->ss_plan_tag = tag;
                ctx->ss_plan_radix = radix;
            }
            st.ss_planned = sb.sample_in ? 1 : 0;
        }
        }
    }
    if (hint && !msd_done) {
        // Neither the remembered sort nor the sample sort took this text: forget the plan and size the key the long way.
        ctx->plan_path = 0;
        return start_over(flags);
    }
    if (plain && n >= (1u << 24) && !fronted) {
        // (the MSD sort's own exact check can refuse a text; the sample sort takes any text, and a bucket beyond a tile --
        // with remembered splitters as unlikely as with fresh ones -- declines: both start over without the plan)
        ctx->plan_path = st.msd ? 1 : (st.ss && ctx->ss_plan_tag != 0 ? 2 : 0);
        ctx->plan_logn = logn;
        memcpy(ctx->plan_present, present_bits, 32);
        memcpy(ctx->plan_lut, lut, 256);
        ctx->plan_sigma = sigma;
    }
    if (msd_done) {
    } else if (ties) PSS_TRY(suffix_sort_flags(ctx, K, V, n, key_bits0, &tk, work, &cur, profile, &ss));
    else PSS_TRY(radix_sort_pairs(ctx, K, V, n, key_bits0, 0xffffffffu, &tk, 0, work, &cur, profile, &ss));
    st.initial_passes = (u32)ss.launches;
    PSS_HIP(hipEventRecord(timer.ev_mid, s));
    // ---- 2. rerank + compaction, 3. doubling rounds ----
    RoundsIO io;
    io.n = n;
    io.SA = SA;
    io.K[0] = K[0]; io.K[1] = K[1];
    io.V[0] = V[0]; io.V[1] = V[1];
    io.ISA = ISA;
    io.P[0] = P[0]; io.P[1] = P[1];
    io.GRP = GRP;
    io.codes = codes;
    io.b = b; io.plus_one = plus_one; io.key_chars = key_chars; io.key_drop = key_drop;
    io.h0 = (u64)(key_drop ? key_chars - 1 : key_chars);
    io.cur = cur;
    io.final_buf = final_buf;
    io.v_scratch = v_scratch;
    io.ties = ties;
    io.msd_fused = msd_fused;
    io.msd_active = msd_active;
    io.no_sparse = st.ss != 0;
    io.work = work;
    io.d_agg_head = d_agg_head; io.d_agg_cnt = d_agg_cnt; io.d_red = d_red; io.d_counters = d_counters; io.h_small = h_small;
    io.profile = profile;
    io.side = &side;
    PSS_TRY(refine_rounds(ctx, knobs, io, ss, st));
    side.join();
    if (side.started && st.anchor_side == 0) st.anchor_side = 2;      // started for nothing
    if (plain && n >= (1u << 24)) {
        ctx->side_plan_heff = st.anchor_side == 1 ? side.h_eff : 0;
        ctx->side_plan_logn = logn;
        ctx->side_plan_b = b;
    }
    PSS_HIP(hipEventRecord(timer.ev1, s));
    PSS_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    PSS_HIP(hipEventElapsedTime(&ms, timer.ev0, timer.ev1));
    st.ss_plan_refused = ctx->ss_refused_note ? 1 : 0;
    st.ms_total = ms + ctx->restart_ms;
            st.ms_restarts = ctx->restart_ms;
    PSS_HIP(hipEventElapsedTime(&ms, timer.ev0, timer.ev_mid));
    st.ms_initial = ms;
    st.ms_sort = ss.ms;
    st.sort_launches = ss.launches;
    st.sort_elems = ss.elems;
    st.ms_pairs = ss.ms_pairs;
    st.pairs_launches = ss.pairs_launches;
    st.pairs_elems = ss.pairs_elems;
    st.ms_text = ss.ms_text;
    st.text_launches = ss.text_launches;
    for (int i = 0; i < 9; ++i) {
        st.fs_ms[i] = ss.fs_ms[i];
        st.fs_launches[i] = ss.fs_launches[i];
        st.fs_elems[i] = ss.fs_elems[i];
    }
    if (stats) *stats = st;
    return PSS_OK;
}

}  // namespace pss
