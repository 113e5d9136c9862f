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

// ------------------------------------------------------------------ rerank --

constexpr int RR_BLOCK = 256;
constexpr int RR_WAVES = RR_BLOCK / kWave;
#ifndef PSS_RR_ROWS
#define PSS_RR_ROWS 8
#endif
constexpr int RR_ROWS = PSS_RR_ROWS;               // rows of 64 elements per wave
constexpr int RR_WSEG = RR_ROWS * kWave;           // 512 elements per wave
constexpr int RR_TILE = RR_WSEG * RR_WAVES;        // 2048 elements per tile
constexpr u32 RR_MAX_RANGES = 1024;

struct RerankArgs {
    const u64 *keys;     // sorted keys of the m elements
    const u32 *idx;      // their suffix indices
    const u32 *pos;      // their SA positions (nullptr: element t sits at SA position t)
    const u32 *grp;      // text rounds: current group rank of every element (keys alone do not
                         // identify the group); nullptr when the key carries the group
    const u32 *tied_sa;  // initial rerank after a TIES final pass: no keys; element j's suffix is
                         // tied_sa[j] & 0x7fffffff, bit 31 = same key as element j-1
    u32 m;
    u32 num_tiles, tiles_per_range, num_ranges;
    u32 *agg_head;       // [ranges] 1 + last group-head index of the range (0 = none)
    u32 *agg_cnt;        // [ranges] active elements of the range
    u32 *SA;
    u32 *ISA;
    u32 *pos_out, *idx_out, *grp_out;
    u32 *counters;       // [0] total active
    u64 *ht;             // sparse mode: suffix -> rank hash table (see ht_*)
    u32 ht_mask;
    int rank_bits;       // doubling rounds: key = (old group rank << rank_bits) | rank2
};

struct WaveFlags {
    u64 head[RR_ROWS];   // ballot: element starts a group
    u64 act[RR_ROWS];    // ballot: element's group has more than one member
    u64 valid[RR_ROWS];
};

// Loads the wave's 512-element segment (element (r, lane) = wbase + 64 r + lane)
// and derives group-head / active ballots from neighbouring keys.
// Variant for the initial rerank after a TIES final pass: heads come from bit 31 of the
// flagged suffix array, no neighbour comparison is needed.  v[r] receives the raw values.
__device__ __forceinline__ void wave_flags_tied(const u32 *tied_sa, u32 m, u32 wbase, WaveFlags &f, u32 (&v)[RR_ROWS])
{
    const u32 lane = lane_id();
#pragma unroll
    for (int r = 0; r < RR_ROWS; ++r) {
        const u32 j = wbase + r * kWave + lane;
        v[r] = (j < m) ? tied_sa[j] : 0;
    }
    const u32 jn = wbase + RR_WSEG;
    u32 edge = 0;
    if (lane == 63 && jn < m) edge = tied_sa[jn];
#pragma unroll
    for (int r = 0; r < RR_ROWS; ++r) {
        const u32 j = wbase + r * kWave + lane;
        const bool valid = j < m;
        f.head[r] = __ballot(valid && (j == 0 || !(v[r] >> 31)));
        f.valid[r] = __ballot(valid);
    }
    const bool next_seg_head = (jn >= m) || !(edge >> 31);
    const u64 nsh = (__ballot(next_seg_head) >> 63) & 1ull;
#pragma unroll
    for (int r = 0; r < RR_ROWS; ++r) {
        const u64 hv = f.head[r] | ~f.valid[r];
        const u64 first_next = (r + 1 < RR_ROWS) ? ((f.head[r + 1] | ~f.valid[r + 1]) & 1ull) : nsh;
        const u64 next = (hv >> 1) | (first_next << 63);
        f.act[r] = f.valid[r] & ~(f.head[r] & next);
    }
}

__device__ __forceinline__ void wave_flags(const u64 *keys, const u32 *grp, u32 m, u32 wbase, WaveFlags &f,
                                           u64 (&key)[RR_ROWS])
{
    const u32 lane = lane_id();
    u32 g[RR_ROWS];
#pragma unroll
    for (int r = 0; r < RR_ROWS; ++r) {
        const u32 j = wbase + r * kWave + lane;
        key[r] = (j < m) ? keys[j] : 0;
        g[r] = (grp && j < m) ? grp[j] : 0;
    }
    // element just before the segment (lane 0) and just after it (lane 63)
    u64 edge = 0;
    u32 gedge = 0;
    const u32 jn = wbase + RR_WSEG;
    if (lane == 0 && wbase > 0 && wbase < m) {
        edge = keys[wbase - 1];
        if (grp) gedge = grp[wbase - 1];
    }
    if (lane == 63 && jn < m) {
        edge = keys[jn];
        if (grp) gedge = grp[jn];
    }
    u64 last = 0;   // key / group of lane 63 of the previous row
    u32 glast = 0;
#pragma unroll
    for (int r = 0; r < RR_ROWS; ++r) {
        const u32 j = wbase + r * kWave + lane;
        u64 pk = __shfl_up(key[r], 1);
        u32 pg = __shfl_up(g[r], 1);
        if (lane == 0) {
            pk = (r == 0) ? edge : last;
            pg = (r == 0) ? gedge : glast;
        }
        last = __shfl(key[r], 63);
        glast = __shfl(g[r], 63);
        const bool valid = j < m;
        const bool head = valid && (j == 0 || key[r] != pk || g[r] != pg);
        f.head[r] = __ballot(head);
        f.valid[r] = __ballot(valid);
    }
    // is the element right after the segment a head (or the end of the array)?
    const bool next_seg_head = (jn >= m) || (key[RR_ROWS - 1] != edge) || (g[RR_ROWS - 1] != gedge);
    const u64 nsh = (__ballot(next_seg_head) >> 63) & 1ull;   // lane 63's verdict
#pragma unroll
    for (int r = 0; r < RR_ROWS; ++r) {
        // "head or nothing" mask: invalid slots count as heads for the element before them
        const u64 hv = f.head[r] | ~f.valid[r];
        const u64 first_next = (r + 1 < RR_ROWS) ? ((f.head[r + 1] | ~f.valid[r + 1]) & 1ull) : nsh;
        const u64 next = (hv >> 1) | (first_next << 63);
        f.act[r] = f.valid[r] & ~(f.head[r] & next);
    }
}

__global__ __launch_bounds__(RR_BLOCK) void rr_reduce_kernel(RerankArgs a)
{
    __shared__ u32 s_head, s_cnt;
    const u32 g = blockIdx.x;
    if (threadIdx.x == 0) { s_head = 0; s_cnt = 0; }
    __syncthreads();
    const u32 tile0 = g * a.tiles_per_range, tile1 = min(tile0 + a.tiles_per_range, a.num_tiles);
    u32 whead = 0, wcnt = 0;
    for (u32 tile = tile0; tile < tile1; ++tile) {
        const u32 wbase = tile * RR_TILE + wave_id() * RR_WSEG;
        if (wbase >= a.m) break;
        WaveFlags f;
        u64 key[RR_ROWS];
        u32 tv[RR_ROWS];
        if (a.tied_sa) wave_flags_tied(a.tied_sa, a.m, wbase, f, tv);
        else wave_flags(a.keys, a.grp, a.m, wbase, f, key);
#pragma unroll
        for (int r = 0; r < RR_ROWS; ++r) {
            if (f.head[r]) whead = wbase + r * kWave + (63 - __builtin_clzll(f.head[r])) + 1;
            wcnt += (u32)__popcll(f.act[r]);
        }
    }
    if (lane_id() == 0) {
        atomicMax(&s_head, whead);
        atomicAdd(&s_cnt, wcnt);
    }
    __syncthreads();
    if (threadIdx.x == 0) { a.agg_head[g] = s_head; a.agg_cnt[g] = s_cnt; }
}

// rr_reduce for the flagged suffix array of a TIES final pass: only bit 31 matters, so every
// lane takes four consecutive elements with one 16-byte load (element j is active iff it or its
// successor is flagged; it is a head iff it is not flagged).
__global__ __launch_bounds__(RR_BLOCK) void rr_reduce_tied_kernel(RerankArgs a)
{
    __shared__ u32 s_head, s_cnt;
    const u32 g = blockIdx.x, tid = threadIdx.x, lane = lane_id();
    if (tid == 0) { s_head = 0; s_cnt = 0; }
    __syncthreads();
    const u64 e0 = (u64)g * a.tiles_per_range * RR_TILE;
    const u64 e1 = min(e0 + (u64)a.tiles_per_range * RR_TILE, (u64)a.m);
    u32 cnt = 0, head = 0;
    for (u64 jb = e0; jb < e1; jb += 4 * RR_BLOCK) {     // uniform trip count: the shuffles need whole waves
        const u64 j = jb + 4ull * tid;
        u32 v[4] = {0, 0, 0, 0};
        if (j + 4 <= (u64)a.m) {
            const uint4 q = *reinterpret_cast<const uint4 *>(a.tied_sa + j);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            for (int c = 0; c < 4; ++c)
                if (j + c < (u64)a.m) v[c] = a.tied_sa[j + c];
        }
        if (j == 0) v[0] &= 0x7fffffffu;
        u32 nxt = __shfl_down(v[0], 1);
        if (lane == 63) nxt = (j + 4 < (u64)a.m) ? a.tied_sa[j + 4] : 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (j + c < (u64)a.m) {
                const u32 tn = (c < 3) ? v[c + 1] : nxt;
                const bool next_tied = (j + c + 1 < (u64)a.m) && (tn >> 31);
                cnt += ((v[c] >> 31) || next_tied) ? 1u : 0u;
                if (!(v[c] >> 31)) head = (u32)(j + c) + 1u;
            }
        }
    }
    cnt = wave_incl_sum(cnt);
    head = wave_incl_max(head);
    if (lane == 63) {
        atomicMax(&s_head, head);
        atomicAdd(&s_cnt, cnt);
    }
    __syncthreads();
    if (tid == 0) { a.agg_head[g] = s_head; a.agg_cnt[g] = s_cnt; }
}

// Exclusive max-scan of agg_head and sum-scan of agg_cnt over <= 1024 ranges.
__global__ __launch_bounds__(1024) void rr_scan_kernel(u32 *agg_head, u32 *agg_cnt, u32 num_ranges, u32 *counters)
{
    __shared__ u32 s_h[16], s_c[16];
    const u32 t = threadIdx.x, lane = lane_id(), w = wave_id();
    const u32 h = (t < num_ranges) ? agg_head[t] : 0, c = (t < num_ranges) ? agg_cnt[t] : 0;
    const u32 ih = wave_incl_max(h), ic = wave_incl_sum(c);
    if (lane == 63) { s_h[w] = ih; s_c[w] = ic; }
    __syncthreads();
    u32 bh = 0, bc = 0, tot = 0;
    for (u32 k = 0; k < 16; ++k) {
        if (k < w) { bh = max(bh, s_h[k]); bc += s_c[k]; }
        tot += s_c[k];
    }
    u32 eh = __shfl_up(ih, 1), ec = ic - c;
    if (lane == 0) eh = 0;
    if (t < num_ranges) { agg_head[t] = max(bh, eh); agg_cnt[t] = bc + ec; }
    if (t == 0) counters[0] = tot;
}

// ---- sparse mode: ranks without an inverse suffix array ----------------------
// When the initial sort leaves only a sliver of the suffixes unresolved
// (m0 <= n / 1024), scattering a full n-entry ISA (4 B random writes, ~16x HBM
// sector amplification) would cost more than the rest of the build.  Instead:
//   * every initially-active suffix lives in an open-addressing hash table
//     (entry = (suffix+1) << 32 | rank), refreshed each round;
//   * any other suffix j was unique after the initial sort, so its rank is
//     1 + lower_bound(sorted initial keys, key(j)) -- a binary search over the
//     still-intact sorted key array, no text comparison, depth independent of h.

__device__ __forceinline__ u32 ht_slot(u32 idx, u32 mask) { return (idx * 0x9E3779B1u) & mask; }

__device__ __forceinline__ void ht_insert(u64 *ht, u32 mask, u32 idx, u32 rank)
{
    const u64 entry = ((u64)(idx + 1u) << 32) | rank;
    u32 h = ht_slot(idx, mask);
    for (;;) {
        const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(&ht[h]), 0ull,
                                                 (unsigned long long)entry);
        if (old == 0ull) return;
        h = (h + 1u) & mask;
    }
}

__device__ __forceinline__ void ht_update(u64 *ht, u32 mask, u32 idx, u32 rank)
{
    u32 h = ht_slot(idx, mask);
    for (;;) {
        const u64 e = ht[h];
        if ((u32)(e >> 32) == idx + 1u) {
            ht[h] = ((u64)(idx + 1u) << 32) | rank;
            return;
        }
        if (e == 0) return;   // not an initially-active suffix: cannot happen
        h = (h + 1u) & mask;
    }
}

// rank of suffix j, or 0 if j is not in the table
__device__ __forceinline__ u32 ht_lookup(const u64 *ht, u32 mask, u32 idx)
{
    u32 h = ht_slot(idx, mask);
    for (;;) {
        const u64 e = ht[h];
        if ((u32)(e >> 32) == idx + 1u) return (u32)e;
        if (e == 0) return 0;
        h = (h + 1u) & mask;
    }
}

__global__ __launch_bounds__(256) void ht_insert_kernel(u64 *ht, u32 mask, const u32 *idx, const u32 *grp, u32 m)
{
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < m; t += gridDim.x * blockDim.x)
        ht_insert(ht, mask, idx[t], grp[t]);
}

constexpr int MODE_ISA = 0;    // rank rounds: ISA[suffix] = rank (only where it changed)
constexpr int MODE_NONE = 1;   // no rank storage: initial rerank of the sparse / text paths, text rounds
constexpr int MODE_HT = 2;     // sparse rounds: refresh the hash table

template <int MODE>
__global__ __launch_bounds__(RR_BLOCK) void rr_apply_kernel(RerankArgs a)
{
    __shared__ u32 s_wh[RR_WAVES], s_wc[RR_WAVES];
    __shared__ u32 s_carry_h, s_carry_c;
    const u32 g = blockIdx.x, lane = lane_id(), w = wave_id();
    if (threadIdx.x == 0) { s_carry_h = a.agg_head[g]; s_carry_c = a.agg_cnt[g]; }
    const u32 tile0 = g * a.tiles_per_range, tile1 = min(tile0 + a.tiles_per_range, a.num_tiles);
    for (u32 tile = tile0; tile < tile1; ++tile) {
        const u32 wbase = tile * RR_TILE + w * RR_WSEG;
        WaveFlags f;
        u64 key[RR_ROWS];
        u32 tv[RR_ROWS] = {};
        if (a.tied_sa) wave_flags_tied(a.tied_sa, a.m, min(wbase, a.m), f, tv);
        else wave_flags(a.keys, a.grp, a.m, min(wbase, a.m), f, key);
        u32 whead = 0, wcnt = 0;
#pragma unroll
        for (int r = 0; r < RR_ROWS; ++r) {
            if (f.head[r]) whead = wbase + r * kWave + (63 - __builtin_clzll(f.head[r])) + 1;
            wcnt += (u32)__popcll(f.act[r]);
        }
        if (lane == 0) { s_wh[w] = whead; s_wc[w] = wcnt; }
        __syncthreads();
        u32 carry_h = s_carry_h, carry_c = s_carry_c;
        u32 tile_h = carry_h, tile_c = carry_c;
#pragma unroll
        for (int k = 0; k < RR_WAVES; ++k) {
            if (k < (int)w) { carry_h = max(carry_h, s_wh[k]); carry_c += s_wc[k]; }
            tile_h = max(tile_h, s_wh[k]);
            tile_c += s_wc[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) { s_carry_h = tile_h; s_carry_c = tile_c; }
        // outputs
#pragma unroll
        for (int r = 0; r < RR_ROWS; ++r) {
            const u32 rowbase = wbase + r * kWave;
            const u32 j = rowbase + lane;
            const u64 le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
            const u64 hm = f.head[r] & le;
            const u32 hd1 = hm ? rowbase + (63 - __builtin_clzll(hm)) + 1 : carry_h;   // 1 + head index
            if (j < a.m) {
                const u32 hd = hd1 - 1;
                const u32 newrank = (a.pos ? a.pos[hd] : hd) + 1;
                const u32 pj = a.pos ? a.pos[j] : j;
                const bool is_act = (f.act[r] >> lane) & 1ull;
                // the suffix index is only needed where something is written with it
                const bool need_idx = is_act || a.SA != nullptr || MODE == MODE_ISA || MODE == MODE_HT;
                u32 ij;
                if (a.tied_sa) {
                    // The flag bit is NOT cleared here (a neighbouring workgroup may still be reading
                    // it).  Every flagged element is tied, hence active, hence rewritten clean by the
                    // next round's `SA[slot] = suffix`; readers in between mask bit 31.
                    ij = tv[r] & 0x7fffffffu;
                } else {
                    ij = need_idx ? a.idx[j] : 0u;
                }
                if (a.SA) a.SA[pj] = ij;
                // a suffix whose rank did not change (e.g. every old group's head) needs no ISA write
                if (MODE == MODE_ISA && (a.pos == nullptr || a.grp == nullptr || newrank != a.grp[j]))
                    a.ISA[ij] = newrank;
                if (MODE == MODE_HT) ht_update(a.ht, a.ht_mask, ij, newrank);
                if (is_act) {
                    const u32 u = carry_c + mbcnt(f.act[r]);
                    a.pos_out[u] = pj;
                    a.idx_out[u] = ij;
                    a.grp_out[u] = newrank;
                }
            }
            if (f.head[r]) carry_h = rowbase + (63 - __builtin_clzll(f.head[r])) + 1;
            carry_c += (u32)__popcll(f.act[r]);
        }
    }
}

// rr_apply for the flagged suffix array of a TIES final pass when nothing but the active list
// is written (MODE_NONE, SA already in place, element t sits at SA position t).  Same tiling as
// rr_apply_kernel, but every lane owns 2 x 4 consecutive elements (16-byte loads): heads and
// compaction offsets come from two wave scans per half instead of ballots.
__global__ __launch_bounds__(RR_BLOCK) void rr_apply_tied_kernel(RerankArgs a)
{
    static_assert(RR_WSEG == 512, "two halves of 64 lanes x 4 elements");
    __shared__ u32 s_wh[RR_WAVES], s_wc[RR_WAVES];
    __shared__ u32 s_carry_h, s_carry_c;
    const u32 g = blockIdx.x, lane = lane_id(), w = wave_id();
    if (threadIdx.x == 0) { s_carry_h = a.agg_head[g]; s_carry_c = a.agg_cnt[g]; }
    const u32 tile0 = g * a.tiles_per_range, tile1 = min(tile0 + a.tiles_per_range, a.num_tiles);
    const u64 m = a.m;
    for (u32 tile = tile0; tile < tile1; ++tile) {
        const u64 wbase = (u64)tile * RR_TILE + (u64)w * RR_WSEG;
        u32 v[2][4];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const u64 j = wbase + 256u * hf + 4u * lane;
            if (j + 4 <= m) {
                const uint4 q = *reinterpret_cast<const uint4 *>(a.tied_sa + j);
                v[hf][0] = q.x; v[hf][1] = q.y; v[hf][2] = q.z; v[hf][3] = q.w;
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[hf][c] = (j + c < m) ? a.tied_sa[j + c] : 0u;
            }
        }
        if (wbase == 0 && lane == 0) v[0][0] &= 0x7fffffffu;
        u32 after = 0;                                     // first element past the wave's segment
        if (lane == 63 && wbase + RR_WSEG < m) after = a.tied_sa[wbase + RR_WSEG];
        u32 lane_cnt[2], nxt[2];
        u32 wcnt = 0, whead = 0;
        u32 excl_c[2], excl_h[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const u64 j = wbase + 256u * hf + 4u * lane;
            u32 nx = __shfl_down(v[hf][0], 1);
            const u32 first_b = __shfl(v[1][0], 0);
            if (lane == 63) nx = (hf == 0) ? first_b : after;
            nxt[hf] = nx;
            u32 c_ = 0, h_ = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (j + c < m) {
                    const u32 tn = (c < 3) ? v[hf][c + 1] : nx;
                    const bool next_tied = (j + c + 1 < m) && (tn >> 31);
                    c_ += ((v[hf][c] >> 31) || next_tied) ? 1u : 0u;
                    if (!(v[hf][c] >> 31)) h_ = (u32)(j + c) + 1u;
                }
            }
            lane_cnt[hf] = c_;
            const u32 ic = wave_incl_sum(c_), ih = wave_incl_max(h_);
            excl_c[hf] = wcnt + ic - c_;
            u32 eh = __shfl_up(ih, 1);
            if (lane == 0) eh = 0;
            excl_h[hf] = max(whead, eh);
            wcnt += __shfl(ic, 63);
            whead = max(whead, __shfl(ih, 63));
        }
        if (lane == 0) { s_wh[w] = whead; s_wc[w] = wcnt; }
        __syncthreads();
        u32 carry_h = s_carry_h, carry_c = s_carry_c;
        u32 tile_h = carry_h, tile_c = carry_c;
#pragma unroll
        for (int k = 0; k < RR_WAVES; ++k) {
            if (k < (int)w) { carry_h = max(carry_h, s_wh[k]); carry_c += s_wc[k]; }
            tile_h = max(tile_h, s_wh[k]);
            tile_c += s_wc[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) { s_carry_h = tile_h; s_carry_c = tile_c; }
        if (wcnt == 0) continue;                           // nothing active in this wave's segment
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (lane_cnt[hf] == 0) continue;
            const u64 j = wbase + 256u * hf + 4u * lane;
            u32 u = carry_c + excl_c[hf];
            u32 hd1 = max(carry_h, excl_h[hf]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (j + c < m) {
                    const u32 x = v[hf][c];
                    if (!(x >> 31)) hd1 = (u32)(j + c) + 1u;
                    const u32 tn = (c < 3) ? v[hf][c + 1] : nxt[hf];
                    const bool next_tied = (j + c + 1 < m) && (tn >> 31);
                    if ((x >> 31) || next_tied) {
                        a.pos_out[u] = (u32)(j + c);
                        a.idx_out[u] = x & 0x7fffffffu;
                        a.grp_out[u] = hd1;
                        ++u;
                    }
                }
            }
        }
    }
}

// key(t) = (group rank << rank_bits) | rank of suffix idx[t]+h (0 past the end).
// Also reduces OR / AND of all keys so the host can skip constant digits.
struct KeyArgs {
    const u32 *idx;
    const u32 *grp;
    const u32 *ISA;       // dense mode
    const u64 *ht;        // sparse mode
    u32 ht_mask;
    const u32 *sa;        // sparse mode: suffix array after the initial sort (every initial group
                          // occupies its final slots, so the key order along it is the initial key order)
    const u8 *codes;
    int code_bits, key_chars, plus_one;
    u32 m, n, h;
    int rank_bits;
    u64 *keys;
    u64 *red;
};

__device__ __forceinline__ u64 text_key_at(const u8 *codes, u32 j, int b, int k, int plus_one, u32 n)
{
    // k <= 16 symbols starting at j (codes are zero padded past n).  The address is random per lane,
    // and a scattered load costs the address unit one cycle per lane and instruction whatever its
    // width: two aligned 16-byte loads and a funnel shift instead of six 4-byte loads
    // (`words` 2^29: 78.4 -> 76.0 ms).
    const uint4 *q = reinterpret_cast<const uint4 *>(codes + (j & ~15u));
    const uint4 a = q[0], c = q[1];
    const u64 x0 = (u64)a.x | ((u64)a.y << 32), x1 = (u64)a.z | ((u64)a.w << 32);
    const u64 x2 = (u64)c.x | ((u64)c.y << 32), x3 = (u64)c.z | ((u64)c.w << 32);
    const bool up = (j & 8u) != 0;
    const u32 s8 = (j & 7u) * 8u;
    const u64 l0 = up ? x1 : x0, l1 = up ? x2 : x1, l2 = up ? x3 : x2;
    const u64 w0 = s8 ? (l0 >> s8) | (l1 << (64 - s8)) : l0;
    const u64 w1 = s8 ? (l1 >> s8) | (l2 << (64 - s8)) : l1;
    u64 key = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (c < k) {
            u32 v = (u32)((c < 8 ? w0 : w1) >> ((c & 7) * 8)) & 0xffu;
            if (plus_one) v = ((u64)j + c < n) ? v + 1u : 0u;
            key = (key << b) | v;
        }
    }
    return key;
}

template <bool SPARSE>
__global__ __launch_bounds__(256) void build_keys_kernel(KeyArgs a)
{
    u64 vor = 0, vand = ~0ull;
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < a.m; t += gridDim.x * blockDim.x) {
        const u64 i2 = (u64)a.idx[t] + a.h;
        u32 r2 = 0;
        if (i2 < a.n) {
            if (SPARSE) {
                r2 = ht_lookup(a.ht, a.ht_mask, (u32)i2);
                if (r2 == 0) {
                    const u64 key = text_key_at(a.codes, (u32)i2, a.code_bits, a.key_chars, a.plus_one, a.n);
                    u32 lo = 0, hi = a.n;
                    while (lo < hi) {
                        const u32 mid = lo + ((hi - lo) >> 1);
                        if (text_key_at(a.codes, a.sa[mid] & 0x7fffffffu, a.code_bits, a.key_chars, a.plus_one, a.n) < key) lo = mid + 1;
                        else hi = mid;
                    }
                    r2 = lo + 1;
                }
            } else {
                r2 = a.ISA[i2];
            }
        }
        const u64 key = ((u64)a.grp[t] << a.rank_bits) | r2;
        a.keys[t] = key;
        vor |= key;
        vand &= key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        vor |= __shfl_xor(vor, o);
        vand &= __shfl_xor(vand, o);
    }
    // one pair of atomics per workgroup: they all hit the same two words (with one pair per wave a list of
    // 262 144 keys spent 90 us here, 85 of them queueing)
    __shared__ u64 s_or[256 / kWave], s_and[256 / kWave];
    if (lane_id() == 0) {
        s_or[wave_id()] = vor;
        s_and[wave_id()] = vand;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 256 / kWave; ++w) {
            vor |= s_or[w];
            vand &= s_and[w];
        }
        atomicOr(reinterpret_cast<unsigned long long *>(&a.red[0]), (unsigned long long)vor);
        atomicAnd(reinterpret_cast<unsigned long long *>(&a.red[1]), (unsigned long long)vand);
    }
}

// ---- text rounds: extend every tied group by the NEXT symbols of the text ----
// Natural text leaves most suffixes tied after the initial sort, but in small
// groups and only for a few dozen more symbols.  Instead of ranks (which need
// an n-entry inverse suffix array: n random 4-byte writes plus m random reads
// per round) a round then sorts each group by a 64-bit key packed from the text
// at offset h: small groups (<= GS_CAP members) are ranked inside LDS by direct
// counting, the few large groups go through two chained radix sorts
// (text key, then group).  No ISA exists in this mode; if ties survive
// TEXT_ROUNDS_MAX rounds (repetitive data) the ISA is built once and the
// doubling rounds take over.

// sub_pos (anchors, anchor_impl.h): element value v stands for the suffix at text position sub_pos[v].
__global__ __launch_bounds__(256) void text_keys_kernel(const u32 *idx, u32 m, u32 n, u32 h, const u8 *codes, int b,
                                                          int k, int plus_one, u64 *keys, const u32 *sub_pos)
{
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < m; t += gridDim.x * blockDim.x) {
        const u32 v = idx[t];
        const u64 j = (u64)(sub_pos ? sub_pos[v] : v) + h;
        keys[t] = (j < n) ? text_key_at(codes, (u32)j, b, k, plus_one, n) : 0ull;
    }
}

// Initial keys of a subset sort: the first k symbols of the suffixes at pos[0 .. m), value = ordinal.
__global__ __launch_bounds__(256) void subset_keys_kernel(const u32 *pos, u32 m, u32 n, const u8 *codes, int b, int k,
                                                            int plus_one, u64 *keys, u32 *vals)
{
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < m; t += gridDim.x * blockDim.x) {
        keys[t] = text_key_at(codes, pos[t], b, k, plus_one, n);
        vals[t] = t;
    }
}

// Doubling-round key of the group-local rounds: rank of suffix idx[t]+h (0 past the end).
__global__ __launch_bounds__(256) void rank_keys_kernel(const u32 *idx, u32 m, u32 n, u32 h, const u32 *ISA, u64 *keys)
{
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < m; t += gridDim.x * blockDim.x) {
        const u64 j = (u64)idx[t] + h;
        keys[t] = (j < n) ? (u64)ISA[j] : 0ull;
    }
}

constexpr int GS_T = 2048;      // elements per workgroup window
#ifndef PSS_GS_CAP
#define PSS_GS_CAP 512
#endif
constexpr int GS_CAP = PSS_GS_CAP;     // largest group ranked in LDS (= halo on both sides)
constexpr int GS_LDS = GS_T + 2 * GS_CAP;

// Sorts every group of <= GS_CAP members by key (ties keep their order) into
// okey/oidx; members of larger groups are copied through and flagged in big[].
// blk_big[b] / blk_heads[b] = flagged elements / flagged group heads of window b.
// Group extents come from two workgroup scans over the head flags of the LDS
// range (last head at or before i, first head after i), so every element knows
// its group in O(1); only members of small groups run the O(size) counting loop.
constexpr int GS_PER = GS_LDS / 256;   // LDS elements owned by one thread in the extent scans
static_assert(GS_LDS % 256 == 0, "extent scans assume an even split");

// K32: the keys are ranks (rank rounds: < 2^32, the high word is zero) -- 4-byte keys in LDS, 32-bit compares in the
// counting loop, which is all the kernel does on groups of hundreds (duplicated blocks: 33.5 ms per round with 8-byte keys).
template <bool K32>
__global__ __launch_bounds__(256) void group_sort_kernel(const u64 *key, const u32 *idx, const u32 *grp, u32 m,
                                                           u64 *okey, u32 *oidx, u8 *big, u32 *blk_big, u32 *blk_heads)
{
    using KT = typename std::conditional<K32, u32, u64>::type;
    __shared__ KT s_key[GS_LDS];
    __shared__ u32 s_grp[GS_LDS];
    __shared__ u16 s_start[GS_LDS];   // LDS index of the head of i's group
    __shared__ u16 s_end[GS_LDS];     // LDS index one past the last member of i's group
    __shared__ u8 s_mixed[GS_LDS];    // at a group's head: some member's key differs from its predecessor's
    __shared__ u32 s_wave[2][4];
    __shared__ u32 s_cnt[2];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u32 base = blockIdx.x * GS_T;
    const u32 lo = base >= (u32)GS_CAP ? base - GS_CAP : 0;
    const u32 hi_want = base + GS_T + GS_CAP;
    const u32 hi = hi_want < m ? hi_want : m;
    const u32 cnt = hi - lo;                       // valid LDS elements
    for (u32 i = tid; i < (u32)GS_LDS; i += 256) {
        s_key[i] = (i < cnt) ? (KT)key[lo + i] : (KT)0;
        s_mixed[i] = 0;
        s_grp[i] = (i < cnt) ? grp[lo + i] : 0xffffffffu;
    }
    if (tid < 2) s_cnt[tid] = 0;
    __syncthreads();
    // head flags of my GS_PER consecutive elements; index 0 and everything past the data count as heads
    const u32 i0 = tid * GS_PER;
    u32 hm = 0;
#pragma unroll
    for (int q = 0; q < GS_PER; ++q) {
        const u32 i = i0 + q;
        const bool head = i == 0 || i >= cnt || s_grp[i] != s_grp[i - 1];
        hm |= (head ? 1u : 0u) << q;
    }
    // last head at or before i: exclusive max-scan over threads of (1 + index of my last head)
    const u32 my_last = hm ? i0 + (31 - __builtin_clz(hm)) + 1 : 0;
    u32 incl = wave_incl_max(my_last);
    if (lane == 63) s_wave[0][wave] = incl;
    // first head after i: exclusive min-scan from the right of my first head -> max-scan of (GS_LDS - index)
    const u32 my_first_rev = hm ? GS_LDS - (i0 + (u32)__builtin_ctz(hm)) : 0;
    // reverse lane order inside the wave so that a forward max-scan runs right-to-left
    u32 rincl = wave_incl_max(__shfl(my_first_rev, 63 - (int)lane));
    if (lane == 63) s_wave[1][3 - wave] = rincl;
    __syncthreads();
    u32 carry = 0;
    for (u32 w = 0; w < wave; ++w) carry = max(carry, s_wave[0][w]);
    u32 excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 0;
    u32 last_head1 = max(carry, excl);             // 1 + LDS index of the last head before my block of elements
    u32 rcarry = 0;
    for (u32 w = 0; w < 3 - wave; ++w) rcarry = max(rcarry, s_wave[1][w]);
    u32 rexcl = __shfl_up(rincl, 1);
    if (lane == 0) rexcl = 0;
    // rexcl belongs to reversed lane (63 - lane); bring it back
    const u32 rmine = __shfl(rexcl, 63 - (int)lane);
    const u32 next_rev = max(rcarry, rmine);       // GS_LDS - (LDS index of the first head after my elements), 0 = none
    u32 next_head = next_rev ? GS_LDS - next_rev : GS_LDS;
#pragma unroll
    for (int q = 0; q < GS_PER; ++q) {
        if ((hm >> q) & 1u) last_head1 = i0 + q + 1;
        s_start[i0 + q] = (u16)(last_head1 - 1);
    }
#pragma unroll
    for (int q = GS_PER - 1; q >= 0; --q) {
        s_end[i0 + q] = (u16)next_head;
        if ((hm >> q) & 1u) next_head = i0 + q;
    }
    __syncthreads();
    // A group whose members all carry the same key stays as it is (ties keep their order): no counting.  That is the
    // common case where whole blocks of text are duplicated -- every copy of a suffix has the same rank h symbols on --
    // and it is cheap to know: one pass over neighbouring members.
#pragma unroll
    for (int q = 0; q < GS_PER; ++q) {
        const u32 i = i0 + q;
        if (i > 0 && i < cnt && s_grp[i] == s_grp[i - 1] && s_key[i] != s_key[i - 1]) s_mixed[s_start[i]] = 1;
    }
    __syncthreads();
    const u32 wend = (base + GS_T < m) ? base + GS_T : m;
    u32 nb = 0, nh = 0;
    for (u32 j = base + tid; j < wend; j += 256) {
        const u32 i = j - lo;
        const KT k = s_key[i];
        const u32 gs = s_start[i], ge = s_end[i];          // LDS indices, [gs, ge)
        // a group touching the edge of the LDS range continues outside (unless that edge is the data's edge)
        const bool open = (gs == 0 && lo > 0) || (ge >= cnt && hi < m);
        if (open || ge - gs > (u32)GS_CAP) {
            okey[j] = k;
            oidx[j] = idx[j];
            big[j] = 1;
            ++nb;
            if (gs == i) ++nh;
        } else {
            u32 rank = i - gs;
            if (s_mixed[gs]) {
                rank = 0;
                for (u32 q = gs; q < ge; ++q) {
                    const KT kq = s_key[q];
                    rank += (kq < k || (kq == k && q < i)) ? 1u : 0u;
                }
            }
            okey[lo + gs + rank] = k;
            oidx[lo + gs + rank] = idx[j];
            big[j] = 0;
        }
    }
    if (nb) atomicAdd(&s_cnt[0], nb);
    if (nh) atomicAdd(&s_cnt[1], nh);
    __syncthreads();
    if (tid == 0) {
        blk_big[blockIdx.x] = s_cnt[0];
        blk_heads[blockIdx.x] = s_cnt[1];
    }
}

// ---- rank rounds: the same job by a segmented MERGE sort --------------------------------------------------
// group_sort_kernel ranks a member by counting the smaller members of its group: O(group) LDS reads per member --
// built for the groups of a few suffixes natural text leaves.  Repeats make groups as large as the number of copies,
// and a rank round over duplicated blocks (512 copies: 512 reads per member) spent 40 ms in it at n = 2^29.  Here the
// whole LDS range (the window and its halos, 3072 elements) is sorted ONCE by the 56-bit number
//     [ LDS index of the element's group head : 12 | key : 32 | the element's own LDS index : 12 ]
// -- groups are contiguous and their heads ascend, so the sort permutes every group inside its own slots and nothing
// else; ties keep their order (the index), members of open or oversized groups carry key 0 and stay where they are.
// Eight elements per thread through a sorting network, then merge rounds with a merge-path search per thread (the
// scheme of ss_local_kernel, on 8-byte elements); a pair of runs whose border is a group border is in order already
// and is skipped.  The cost does not depend on the group sizes.  Every window writes the slots of its own 2048
// positions (a group that straddles two windows is sorted by both, identically).
constexpr int GM_BLOCK = 384, GM_IPT = 8, GM_WAVES = GM_BLOCK / kWave;
static_assert(GM_BLOCK * GM_IPT == GS_LDS, "one thread per eight elements of the LDS range");
static_assert(GS_LDS <= 4096, "12-bit LDS indices");
__device__ __forceinline__ u32 gm_slot(u32 p) { return p + (p >> 3); }      // a thread's eight elements: 72-byte stride, no bank conflicts
__device__ __forceinline__ void gm_cswap(u64 &a, u64 &b)
{
    const bool sw = b < a;
    const u64 x = sw ? b : a, y = sw ? a : b;
    a = x;
    b = y;
}

__global__ __launch_bounds__(GM_BLOCK) void group_msort32_kernel(const u64 *key, const u32 *idx, const u32 *grp, u32 m, u64 *okey,
                                                                   u32 *oidx, u8 *big, u32 *blk_big, u32 *blk_heads)
{
    __shared__ u64 s_e[GS_LDS + GS_LDS / 8];
    __shared__ u32 s_idx[GS_LDS];
    __shared__ u32 s_bigm[GS_LDS / 32];
    __shared__ u32 s_wave[2][GM_WAVES];
    __shared__ u32 s_cnt[2];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u32 base = blockIdx.x * GS_T;
    const u32 lo = base >= (u32)GS_CAP ? base - GS_CAP : 0;
    const u32 hi_want = base + GS_T + GS_CAP;
    const u32 hi = hi_want < m ? hi_want : m;
    const u32 cnt = hi - lo;                       // valid LDS elements
    const u32 wend = (base + GS_T < m) ? base + GS_T : m;
    for (u32 i = tid; i < (u32)GS_LDS; i += GM_BLOCK) s_idx[i] = (i < cnt) ? idx[lo + i] : 0u;
    if (tid < GS_LDS / 32) s_bigm[tid] = 0;
    if (tid < 2) s_cnt[tid] = 0;
    // my eight consecutive elements: group ranks (and the one before), head flags
    const u32 i0 = tid * GM_IPT;
    u32 g[GM_IPT + 1];
    g[0] = (i0 > 0 && i0 - 1 < cnt) ? grp[lo + i0 - 1] : 0xffffffffu;
    u32 k32[GM_IPT];
#pragma unroll
    for (int q = 0; q < GM_IPT; ++q) {
        const u32 i = i0 + q;
        g[q + 1] = (i < cnt) ? grp[lo + i] : 0xffffffffu;
        k32[q] = (i < cnt) ? (u32)key[lo + i] : 0u;
    }
    u32 hm = 0;
#pragma unroll
    for (int q = 0; q < GM_IPT; ++q) {
        const u32 i = i0 + q;
        const bool head = i == 0 || i >= cnt || g[q + 1] != g[q];
        hm |= (head ? 1u : 0u) << q;
    }
    // last head at or before i / first head after i: the two scans of group_sort_kernel, over GM_WAVES waves
    const u32 my_last = hm ? i0 + (31 - __builtin_clz(hm)) + 1 : 0;
    const u32 incl = wave_incl_max(my_last);
    if (lane == 63) s_wave[0][wave] = incl;
    const u32 my_first_rev = hm ? GS_LDS - (i0 + (u32)__builtin_ctz(hm)) : 0;
    const u32 rincl = wave_incl_max(__shfl(my_first_rev, 63 - (int)lane));
    if (lane == 63) s_wave[1][GM_WAVES - 1 - wave] = rincl;
    __syncthreads();
    u32 carry = 0;
    for (u32 w = 0; w < wave; ++w) carry = max(carry, s_wave[0][w]);
    u32 excl = __shfl_up(incl, 1);
    if (lane == 0) excl = 0;
    u32 last_head1 = max(carry, excl);
    u32 rcarry = 0;
    for (u32 w = 0; w < GM_WAVES - 1 - wave; ++w) rcarry = max(rcarry, s_wave[1][w]);
    u32 rexcl = __shfl_up(rincl, 1);
    if (lane == 0) rexcl = 0;
    const u32 rmine = __shfl(rexcl, 63 - (int)lane);
    const u32 next_rev = max(rcarry, rmine);
    u32 next_head = next_rev ? GS_LDS - next_rev : GS_LDS;
    u32 gs[GM_IPT], ge[GM_IPT];
#pragma unroll
    for (int q = 0; q < GM_IPT; ++q) {
        if ((hm >> q) & 1u) last_head1 = i0 + q + 1;
        gs[q] = last_head1 - 1;
    }
#pragma unroll
    for (int q = GM_IPT - 1; q >= 0; --q) {
        ge[q] = next_head;
        if ((hm >> q) & 1u) next_head = i0 + q;
    }
    u64 v[GM_IPT];
    u32 bigbits = 0, nb = 0, nh = 0;
#pragma unroll
    for (int q = 0; q < GM_IPT; ++q) {
        const u32 i = i0 + q;
        if (i >= cnt) {
            v[q] = ~0ull;
            continue;
        }
        // a group touching the edge of the LDS range continues outside (unless that edge is the data's edge)
        const bool open = (gs[q] == 0 && lo > 0) || (ge[q] >= cnt && hi < m);
        const bool isb = open || ge[q] - gs[q] > (u32)GS_CAP;
        v[q] = ((u64)gs[q] << 44) | ((u64)(isb ? 0u : k32[q]) << 12) | (u64)i;
        if (isb) {
            bigbits |= 1u << q;
            const u32 j = lo + i;
            if (j >= base && j < wend) {
                ++nb;
                if (gs[q] == i) ++nh;
            }
        }
    }
    if (bigbits) atomicOr(&s_bigm[i0 >> 5], bigbits << (i0 & 31u));
    // eight elements in registers: odd-even merge sort network (19 compare-exchanges)
    gm_cswap(v[0], v[1]); gm_cswap(v[2], v[3]); gm_cswap(v[4], v[5]); gm_cswap(v[6], v[7]);
    gm_cswap(v[0], v[2]); gm_cswap(v[1], v[3]); gm_cswap(v[4], v[6]); gm_cswap(v[5], v[7]);
    gm_cswap(v[1], v[2]); gm_cswap(v[5], v[6]);
    gm_cswap(v[0], v[4]); gm_cswap(v[1], v[5]); gm_cswap(v[2], v[6]); gm_cswap(v[3], v[7]);
    gm_cswap(v[2], v[4]); gm_cswap(v[3], v[5]);
    gm_cswap(v[1], v[2]); gm_cswap(v[3], v[4]); gm_cswap(v[5], v[6]);
#pragma unroll
    for (int q = 0; q < GM_IPT; ++q) s_e[gm_slot(i0 + q)] = v[q];
    __syncthreads();
    const bool live = i0 < cnt;                    // the padding stays at the end of every run
    for (u32 L = GM_IPT; L < (u32)GS_LDS; L <<= 1) {
        const u32 pair0 = i0 & ~(2 * L - 1);
        const u32 d = i0 - pair0;
        const u32 A = pair0, B = pair0 + L;
        const u32 lenA = min(L, (u32)GS_LDS - A), lenB = B < (u32)GS_LDS ? min(L, (u32)GS_LDS - B) : 0u;
        // nothing to merge: no second run, or the border between the runs is a border between groups
        bool work = live && lenB != 0;
        if (work) work = (s_e[gm_slot(B - 1)] >> 44) == (s_e[gm_slot(B)] >> 44);
        if (work) {
            u32 a_lo = d > lenB ? d - lenB : 0, a_hi = d < lenA ? d : lenA;
            while (a_lo < a_hi) {
                const u32 mid = (a_lo + a_hi) >> 1;
                const u64 x = s_e[gm_slot(A + mid)], y = s_e[gm_slot(B + d - 1 - mid)];
                if (x < y) a_lo = mid + 1; else a_hi = mid;
            }
            u32 ai = a_lo, bi = d - a_lo;
            u64 va = ai < lenA ? s_e[gm_slot(A + ai)] : ~0ull;
            u64 vb = bi < lenB ? s_e[gm_slot(B + bi)] : ~0ull;
#pragma unroll
            for (int q = 0; q < GM_IPT; ++q) {
                const bool ta = !(vb < va);
                v[q] = ta ? va : vb;
                ai += ta ? 1u : 0u;
                bi += ta ? 0u : 1u;
                if (q + 1 < GM_IPT) {
                    const u32 ni = ta ? ai : bi;
                    const u32 len = ta ? lenA : lenB;
                    const u64 nx = ni < len ? s_e[gm_slot((ta ? A : B) + min(ni, len - 1))] : ~0ull;
                    va = ta ? nx : va;
                    vb = ta ? vb : nx;
                }
            }
        }
        __syncthreads();
        if (work) {
#pragma unroll
            for (int q = 0; q < GM_IPT; ++q) s_e[gm_slot(i0 + q)] = v[q];
        }
        __syncthreads();
    }
    // output: the slots of my own window
#pragma unroll
    for (int q = 0; q < GM_IPT; ++q) {
        const u32 r = q * GM_BLOCK + tid;
        const u32 j = lo + r;
        if (r < cnt && j >= base && j < wend) {
            if ((s_bigm[r >> 5] >> (r & 31u)) & 1u) {
                okey[j] = key[j];
                oidx[j] = s_idx[r];
                big[j] = 1;
            } else {
                const u64 e = s_e[gm_slot(r)];
                okey[j] = (e >> 12) & 0xffffffffull;
                oidx[j] = s_idx[(u32)e & 0xfffu];
                big[j] = 0;
            }
        }
    }
    if (nb) atomicAdd(&s_cnt[0], nb);
    if (nh) atomicAdd(&s_cnt[1], nh);
    __syncthreads();
    if (tid == 0) {
        blk_big[blockIdx.x] = s_cnt[0];
        blk_heads[blockIdx.x] = s_cnt[1];
    }
}

// ---- middle tier: groups of up to MID_CAP members sorted by one workgroup in LDS -----------------
// group_sort ranks groups of <= GS_CAP members by direct counting (O(size) LDS reads per member) and hands
// everything larger to two chained global radix sorts (key, then group): a dozen passes of 24 B per member.
// On natural text a third of the tied suffixes sit in groups of a few hundred to a few thousand members
// (`words` round 1: 120 M of 366 M), far too many for that price and far too few per group to need it.
// mid_collect finds the extent of every flagged group (group ranks never decrease along the list: a binary
// search from the head); mid_sort sorts one group per workgroup with the counting scheme of the MSD local
// sort (msd_sort.hip): one pass of LDS atomics over the top 12 key bits, then every member counts the smaller
// ones of its bin, ties by list position (stable, like group_sort).  A group with a crowded bin (many equal
// keys) stays flagged and takes the chained sorts as before.  Sorted groups are un-flagged and taken out of
// the per-window counts big_compact works from.
constexpr u32 MID_CAP = 4096;
constexpr int MID_BLOCK = 512;
constexpr int MID_IPT = MID_CAP / MID_BLOCK;
constexpr int MID_WAVES = MID_BLOCK / kWave;
constexpr u32 MID_BINS = 4096, MID_WORDS = MID_BINS / 2;      // 16-bit counters, two per LDS word
#ifndef PSS_MID_KMAX
#define PSS_MID_KMAX 512
#endif
constexpr u32 MID_KMAX = PSS_MID_KMAX;

struct MidGroup {
    u32 start, size;
};

__global__ __launch_bounds__(256) void mid_collect_kernel(const u8 *big, const u32 *grp, u32 m, MidGroup *list, u32 *count)
{
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x) {
        if (!big[j]) continue;
        const u32 g = grp[j];
        if (j > 0 && grp[j - 1] == g) continue;                  // not a head
        // first index behind the group, looked for in (j, j + MID_CAP]
        const u32 limit = min(m, j + MID_CAP + 1);
        u32 lo = j + 1, hi = limit;
        while (lo < hi) {
            const u32 mid = lo + ((hi - lo) >> 1);
            if (grp[mid] == g) lo = mid + 1; else hi = mid;
        }
        const u32 size = lo - j;
        if (size <= MID_CAP && size >= 2) list[atomicAdd(count, 1u)] = MidGroup{j, size};
    }
}

// The group leaves the per-window tallies of flagged members / flagged heads (one thread).
__device__ __forceinline__ void mid_untally(u32 gs, u32 size, u32 *blk_big, u32 *blk_heads)
{
    atomicSub(&blk_heads[gs / GS_T], 1u);
    for (u32 w = gs / GS_T; w * GS_T < gs + size; ++w) {
        const u32 a = max(gs, w * (u32)GS_T), b = min(gs + size, (w + 1) * (u32)GS_T);
        atomicSub(&blk_big[w], b - a);
    }
}

// fail_list / fail_count: the groups with a crowded bin (many equal keys -- copies of a stretch of text), for the merge
// sort below (round 5; they used to stay flagged and take the chained radix sorts, a dozen global passes).
__global__ __launch_bounds__(MID_BLOCK) void mid_sort_kernel(const u64 *key, const u32 *idx, const MidGroup *list, const u32 *count,
                                                               int key_bits, u64 *okey, u32 *oidx, u8 *big, u32 *blk_big,
                                                               u32 *blk_heads, MidGroup *fail_list, u32 *fail_count)
{
    __shared__ u64 s_key[MID_CAP];
    __shared__ u16 s_perm[MID_CAP];                    // slot (bin order) -> member
    __shared__ u32 hist[MID_WORDS], hist2[MID_WORDS];
    __shared__ u32 scr[MID_WAVES + 1];
    __shared__ u32 s_fail;
    __shared__ u64 s_diff;
    const u32 tid = threadIdx.x;
    const u32 total = *count;
    (void)key_bits;
    for (u32 gi = blockIdx.x; gi < total; gi += gridDim.x) {
        const u32 gs = list[gi].start, size = list[gi].size;
        const u32 rows = (size + MID_BLOCK - 1) / MID_BLOCK;
        for (u32 i = tid; i < MID_WORDS; i += MID_BLOCK) hist[i] = 0;
        if (tid == 0) {
            s_fail = 0;
            s_diff = 0;
        }
        __syncthreads();
        u64 k[MID_IPT];
        const u64 k_first = key[gs];
        u64 diff = 0;
#pragma unroll
        for (int r = 0; r < MID_IPT; ++r) {
            k[r] = 0;
            const u32 p = r * MID_BLOCK + tid;
            if ((u32)r < rows && p < size) {
                k[r] = key[gs + p];
                s_key[p] = k[r];
                diff |= k[r] ^ k_first;
            }
        }
        // The members of a group often share the next symbols too (the rest of a word): bin on the 12 bits right
        // below the keys' common prefix, not on the top 12 bits of the key.
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) diff |= __shfl_xor(diff, o);
        if ((tid & 63u) == 0 && diff) atomicOr(reinterpret_cast<unsigned long long *>(&s_diff), (unsigned long long)diff);
        __syncthreads();
        const u64 dall = s_diff;
        if (dall == 0) {
            // every member carries the same key (copies of one stretch of text, h symbols on): the group is in order as
            // it stands -- the pass-through copy of the LDS sort is its output -- and only leaves the flagged set
            for (u32 p = tid; p < size; p += MID_BLOCK) big[gs + p] = 0;
            if (tid == 0) mid_untally(gs, size, blk_big, blk_heads);
            __syncthreads();
            continue;
        }
        const int top = dall ? 64 - __builtin_clzll(dall) : 0;          // bits [0, top) vary
        const int shift = top > 12 ? top - 12 : 0;
#pragma unroll
        for (int r = 0; r < MID_IPT; ++r) {
            const u32 p = r * MID_BLOCK + tid;
            if ((u32)r < rows && p < size) {
                const u32 bin = (u32)(k[r] >> shift) & (MID_BINS - 1u);
                atomicAdd(&hist[bin >> 1], 1u << (16u * (bin & 1u)));
            }
        }
        __syncthreads();
        {
            u32 c[8];
            u32 sum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32 wv = hist[4 * tid + j];
                c[2 * j] = wv & 0xffffu;
                c[2 * j + 1] = wv >> 16;
                sum += c[2 * j] + c[2 * j + 1];
            }
            u32 ex = block_excl_sum<MID_WAVES>(sum, scr, nullptr);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32 lo = ex, hi = ex + c[2 * j];
                hist[4 * tid + j] = hist2[4 * tid + j] = lo | (hi << 16);
                ex = hi + c[2 * j + 1];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MID_IPT; ++r) {
            const u32 p = r * MID_BLOCK + tid;
            if ((u32)r < rows && p < size) {
                const u32 bin = (u32)(k[r] >> shift) & (MID_BINS - 1u), sh = 16u * (bin & 1u);
                s_perm[(atomicAdd(&hist2[bin >> 1], 1u << sh) >> sh) & 0xffffu] = (u16)p;
            }
        }
        __syncthreads();
        // thread <-> slot: neighbouring lanes sit in the same bin; rank = smaller members of the bin (ties by list position)
        u32 fin[MID_IPT], who[MID_IPT];
#pragma unroll
        for (int r = 0; r < MID_IPT; ++r) {
            fin[r] = who[r] = 0;
            const u32 q0 = r * MID_BLOCK + tid;
            if ((u32)r < rows && q0 < size) {
                const u32 i = s_perm[q0];
                const u64 x = s_key[i];
                const u32 bin = (u32)(x >> shift) & (MID_BINS - 1u);
                const u32 s0 = (hist[bin >> 1] >> (16u * (bin & 1u))) & 0xffffu;
                const u32 s1 = bin + 1 < MID_BINS ? ((hist[(bin + 1) >> 1] >> (16u * ((bin + 1) & 1u))) & 0xffffu) : size;
                u32 smaller = 0;
                if (s1 - s0 > MID_KMAX) {
                    s_fail = 1;
                } else {
                    for (u32 q = s0; q < s1; ++q) {
                        const u32 j = s_perm[q];
                        const u64 y = s_key[j];
                        smaller += (y < x || (y == x && j < i)) ? 1u : 0u;
                    }
                }
                fin[r] = s0 + smaller;
                who[r] = i;
                k[r] = x;
            }
        }
        __syncthreads();
        if (!s_fail) {
#pragma unroll
            for (int r = 0; r < MID_IPT; ++r) {
                const u32 q0 = r * MID_BLOCK + tid;
                if ((u32)r < rows && q0 < size) {
                    okey[gs + fin[r]] = k[r];
                    oidx[gs + fin[r]] = idx[gs + who[r]];
                    big[gs + q0] = 0;
                }
            }
            if (tid == 0) mid_untally(gs, size, blk_big, blk_heads);
        } else if (tid == 0 && fail_list) {
            fail_list[atomicAdd(fail_count, 1u)] = list[gi];
        }
        __syncthreads();
    }
}

// ---- middle tier, second chance: a merge sort in LDS for the groups the counting scheme gave up ----------------------
// Copies make keys EQUAL: a group of 600 .. 4096 suffixes of which most share the next symbols crowds one bin of
// mid_sort_kernel, and the chained radix sorts it then fell to cost a dozen global passes per member (real files, first
// text round: 111 M of 364 M members went that way; a third of the anchors' own text rounds).  A comparison sort does not
// care: elements (key : 64, position in the group : 12), eight per thread through a sorting network, then merge rounds
// with a merge-path search per thread -- the scheme of group_msort32_kernel and ss_local_kernel -- in 40 KiB of LDS.
constexpr int MM_IPT = MID_CAP / MID_BLOCK;      // 8
static_assert(MM_IPT == 8, "the register network below sorts eight elements");
__device__ __forceinline__ u32 mm_slot(u32 p) { return p + (p >> 3); }
// (keys and positions in separate scalars throughout: an array of {u64, u32} structs went to scratch memory -- 448 bytes
// per lane -- and the kernel took 19 ms where 1 was expected)
#define MM_LT(ak, ap, bk, bp) ((ak) < (bk) || ((ak) == (bk) && (ap) < (bp)))
#define MM_CSWAP(i, j)                                                   \
    {                                                                     \
        const bool sw = MM_LT(vk[j], vp[j], vk[i], vp[i]);                \
        const u64 xk = sw ? vk[j] : vk[i], yk = sw ? vk[i] : vk[j];       \
        const u32 xp = sw ? vp[j] : vp[i], yp = sw ? vp[i] : vp[j];       \
        vk[i] = xk; vk[j] = yk; vp[i] = xp; vp[j] = yp;                   \
    }

__global__ __launch_bounds__(MID_BLOCK) void mid_msort_kernel(const u64 *key, const u32 *idx, const MidGroup *list, const u32 *count,
                                                                u64 *okey, u32 *oidx, u8 *big, u32 *blk_big, u32 *blk_heads)
{
    __shared__ u64 s_k[MID_CAP + MID_CAP / 8];
    __shared__ u16 s_p[MID_CAP + MID_CAP / 8];
    const u32 tid = threadIdx.x;
    const u32 total = *count;
    const u32 i0 = tid * MM_IPT;
    for (u32 gi = blockIdx.x; gi < total; gi += gridDim.x) {
        const u32 gs = list[gi].start, size = list[gi].size;
        u64 vk[MM_IPT];
        u32 vp[MM_IPT];
        // coalesced load through LDS: position r of the group by thread r mod 512
        for (u32 r = tid; r < (u32)MID_CAP; r += MID_BLOCK) {
            s_k[mm_slot(r)] = r < size ? key[gs + r] : ~0ull;
            s_p[mm_slot(r)] = (u16)(r < size ? r : 0xffffu);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MM_IPT; ++q) {
            vk[q] = s_k[mm_slot(i0 + q)];
            vp[q] = s_p[mm_slot(i0 + q)];
        }
        MM_CSWAP(0, 1) MM_CSWAP(2, 3) MM_CSWAP(4, 5) MM_CSWAP(6, 7)
        MM_CSWAP(0, 2) MM_CSWAP(1, 3) MM_CSWAP(4, 6) MM_CSWAP(5, 7)
        MM_CSWAP(1, 2) MM_CSWAP(5, 6)
        MM_CSWAP(0, 4) MM_CSWAP(1, 5) MM_CSWAP(2, 6) MM_CSWAP(3, 7)
        MM_CSWAP(2, 4) MM_CSWAP(3, 5)
        MM_CSWAP(1, 2) MM_CSWAP(3, 4) MM_CSWAP(5, 6)
#pragma unroll
        for (int q = 0; q < MM_IPT; ++q) {
            s_k[mm_slot(i0 + q)] = vk[q];
            s_p[mm_slot(i0 + q)] = (u16)vp[q];
        }
        __syncthreads();
        const bool live = i0 < size;                  // the padding stays at the end of every run
        for (u32 L = MM_IPT; L < MID_CAP; L <<= 1) {
            if (L >= size) break;                     // (uniform: one run holds every element already)
            const u32 pair0 = i0 & ~(2 * L - 1);
            const u32 d = i0 - pair0;
            const u32 A = pair0, B = pair0 + L;
            const bool work = live && B < size;       // no element in the second run: the first is the merge
            if (work) {
                u32 lo = d > L ? d - L : 0, hi = d < L ? d : L;
                while (lo < hi) {
                    const u32 mid = (lo + hi) >> 1;
                    const u32 sa = mm_slot(A + mid), sb = mm_slot(B + d - 1 - mid);
                    const u64 xk = s_k[sa], yk = s_k[sb];
                    const u32 xp = s_p[sa], yp = s_p[sb];
                    if (MM_LT(xk, xp, yk, yp)) lo = mid + 1; else hi = mid;
                }
                u32 ai = lo, bi = d - lo;
                u64 ak = ~0ull, bk = ~0ull;
                u32 ap = 0xffffu, bp = 0xffffu;
                if (ai < L) { ak = s_k[mm_slot(A + ai)]; ap = s_p[mm_slot(A + ai)]; }
                if (bi < L) { bk = s_k[mm_slot(B + bi)]; bp = s_p[mm_slot(B + bi)]; }
#pragma unroll
                for (int q = 0; q < MM_IPT; ++q) {
                    const bool ta = !MM_LT(bk, bp, ak, ap);
                    vk[q] = ta ? ak : bk;
                    vp[q] = ta ? ap : bp;
                    ai += ta ? 1u : 0u;
                    bi += ta ? 0u : 1u;
                    if (q + 1 < MM_IPT) {
                        const u32 ni = ta ? ai : bi;
                        const u32 at = mm_slot((ta ? A : B) + min(ni, L - 1));
                        const u64 nk = ni < L ? s_k[at] : ~0ull;
                        const u32 np = ni < L ? (u32)s_p[at] : 0xffffu;
                        ak = ta ? nk : ak;
                        ap = ta ? np : ap;
                        bk = ta ? bk : nk;
                        bp = ta ? bp : np;
                    }
                }
            }
            __syncthreads();
            if (work) {
#pragma unroll
                for (int q = 0; q < MM_IPT; ++q) {
                    s_k[mm_slot(i0 + q)] = vk[q];
                    s_p[mm_slot(i0 + q)] = (u16)vp[q];
                }
            }
            __syncthreads();
        }
        for (u32 r = tid; r < size; r += MID_BLOCK) {
            okey[gs + r] = s_k[mm_slot(r)];
            oidx[gs + r] = idx[gs + s_p[mm_slot(r)]];
            big[gs + r] = 0;
        }
        if (tid == 0) mid_untally(gs, size, blk_big, blk_heads);
        __syncthreads();
    }
}

// Ordered compaction of the flagged elements of window b: their list index
// bt[], text key and dense group number (0-based ordinal of the big group).
__global__ __launch_bounds__(256) void big_compact_kernel(const u8 *big, const u32 *grp, const u64 *key, u32 m,
                                                            const u64 *blk_big_off, const u64 *blk_head_off, u32 *bt,
                                                            u64 *bkey, u32 *bgid)
{
    __shared__ u32 scr[4 + 1];
    const u32 tid = threadIdx.x;
    const u32 base = blockIdx.x * GS_T;
    u32 run_b = (u32)blk_big_off[blockIdx.x];
    u32 run_h = (u32)blk_head_off[blockIdx.x];
    for (u32 c = 0; c < (u32)GS_T; c += 256) {
        const u32 j = base + c + tid;
        const bool isb = j < m && big[j];
        const bool ish = isb && (j == 0 || grp[j] != grp[j - 1]);
        u32 tot_b, tot_h;
        const u32 eb = block_excl_sum<4>(isb ? 1u : 0u, scr, &tot_b);
        const u32 eh = block_excl_sum<4>(ish ? 1u : 0u, scr, &tot_h);
        if (isb) {
            const u32 u = run_b + eb;
            bt[u] = j;
            bkey[u] = key[j];
            bgid[u] = run_h + eh + (ish ? 1u : 0u) - 1u;   // heads seen so far, this one included
        }
        run_b += tot_b;
        run_h += tot_h;
    }
}

// Second key of the chained sort: the dense group number of the v-th element in
// text-key order.  `unique` appends v so that an unstable sorter (the one-workgroup
// bitonic path for tiny lists) still keeps the text-key order inside a group.
__global__ __launch_bounds__(256) void gather_gid_kernel(const u32 *order, const u32 *bgid, u32 nbig, bool unique,
                                                           u64 *key2)
{
    for (u32 v = blockIdx.x * blockDim.x + threadIdx.x; v < nbig; v += gridDim.x * blockDim.x) {
        const u64 g = bgid[order[v]];
        key2[v] = unique ? ((g << 32) | v) : g;
    }
}

// v-th element of the (group, key)-sorted big list goes to the v-th big slot.
__global__ __launch_bounds__(256) void big_writeback_kernel(const u32 *order, const u32 *bt, const u64 *tkey,
                                                              const u32 *idx, u32 nbig, u64 *okey, u32 *oidx)
{
    for (u32 v = blockIdx.x * blockDim.x + threadIdx.x; v < nbig; v += gridDim.x * blockDim.x) {
        const u32 src = bt[order[v]], dst = bt[v];
        okey[dst] = tkey[src];
        oidx[dst] = idx[src];
    }
}


// ---- large groups (beyond the LDS tiers): a segmented MERGE sort in global memory (round 5) ---------------------------
// Groups of more than 4096 members went through two chained global radix sorts -- by key (eight passes for a 64-bit
// text key), then by dense group number -- with the elements addressed through their list positions: ten to eleven
// scatter passes and three random reads per element on the way back (real files: 98 M such elements per build, 65 GB of
// radix passes, 19 GB of write-back gathers; `source`: 107 + 63 + 25 GB).  But the groups are CONTIGUOUS in the compacted
// list and need sorting only inside themselves.  So: every 4096-element tile of a group is sorted in LDS (the merge
// sort of the middle tier, on (key, suffix) pairs), then runs of L = 4096, 8192, ... are merged pairwise INSIDE their
// group -- one workgroup per 4096 outputs: two merge-path searches in global memory give its share of both runs, the
// share is merged in LDS and written out in order.  ceil(log2(size / 4096)) sequential passes of 12 bytes in / 12 out
// per element instead of eleven scatter passes; no group keys, no positions, nothing gathered: the (key, suffix) pairs
// ARE the payload, and the write-back is a sequential read.  Total order (key, then suffix index): no two elements are
// equal, every phase uses the same comparison.
// MEASURED AND LEFT OFF (PSS_BIG_MERGE=1 switches it on; test_large_groups_take_the_segmented_merge_sort runs it): a wash on
// real files (111.8 / 112.7 ms against 112.6 / 113.5), 3 % on `source`, 4 % SLOWER on `mixed`, whose groups of millions
// need twelve passes where 32-bit rank keys cost the radix sorts seven.  The passes are sequential but not fast -- a tile
// is loaded, merged and stored behind three barriers by a workgroup that spends the first microseconds of each on two
// searches in global memory -- and the traffic they save was not what bounded the build.
constexpr u32 BG_TILE = 4096;
struct BigTile {
    u32 gstart, gsize, t;      // tile t of the group whose members are [gstart, gstart + gsize) of the compacted list
};

__global__ __launch_bounds__(256) void bg_gstart_kernel(const u32 *bgid, u32 nbig, u32 ngroups, u32 *gstart)
{
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nbig; i += gridDim.x * blockDim.x)
        if (i == 0 || bgid[i] != bgid[i - 1]) gstart[bgid[i]] = i;
    if (blockIdx.x == 0 && threadIdx.x == 0) gstart[ngroups] = nbig;
}
struct InTileCount {
    const u32 *gstart;
    __device__ u64 operator()(u64 g) const { return (u64)((gstart[g + 1] - gstart[g] + BG_TILE - 1) / BG_TILE); }
};
__global__ __launch_bounds__(256) void bg_tiles_kernel(const u32 *gstart, const u64 *toff, u32 ngroups, BigTile *tiles)
{
    // one wave per group: lane l writes tiles l, l + 64, ...
    const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const u32 nwaves = (gridDim.x * blockDim.x) >> 6;
    for (u32 g = wave; g < ngroups; g += nwaves) {
        const u32 s0 = gstart[g], size = gstart[g + 1] - s0;
        const u32 nt = (size + BG_TILE - 1) / BG_TILE;
        const u64 off = toff[g];
        for (u32 t = lane; t < nt; t += 64) tiles[off + t] = BigTile{s0, size, t};
    }
}
__global__ __launch_bounds__(256) void bg_gather_kernel(const u32 *bt, const u32 *idx, u32 nbig, u32 *out)
{
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nbig; i += gridDim.x * blockDim.x) out[i] = idx[bt[i]];
}

#define BG_LT(ak, ai, bk, bi) ((ak) < (bk) || ((ak) == (bk) && (ai) < (bi)))
#define BG_CSWAP(i, j)                                                   \
    {                                                                     \
        const bool sw = BG_LT(vk[j], vi[j], vk[i], vi[i]);                \
        const u64 xk = sw ? vk[j] : vk[i], yk = sw ? vk[i] : vk[j];       \
        const u32 xi = sw ? vi[j] : vi[i], yi = sw ? vi[i] : vi[j];       \
        vk[i] = xk; vk[j] = yk; vi[i] = xi; vi[j] = yi;                   \
    }
constexpr int BG_BLOCK = 512, BG_IPT = BG_TILE / BG_BLOCK;
static_assert(BG_IPT == 8, "eight elements per thread");
__device__ __forceinline__ u32 bg_slot(u32 p) { return p + (p >> 3); }

// Two sorted runs in LDS -- A = [0, na), B = [na, na + nb) of the slot space -- merged: thread t gets outputs 8 t .. 8 t + 7.
__device__ __forceinline__ void bg_merge_lds(const u64 *s_k, const u32 *s_i, u32 na, u32 nb, u32 tid, u64 (&vk)[BG_IPT], u32 (&vi)[BG_IPT])
{
    const u32 d = tid * BG_IPT;
    u32 lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        const u32 sa = bg_slot(mid), sb = bg_slot(na + d - 1 - mid);
        if (BG_LT(s_k[sa], s_i[sa], s_k[sb], s_i[sb])) lo = mid + 1; else hi = mid;
    }
    u32 ai = lo, bi = d - lo;
    u64 ak = ~0ull, bk = ~0ull;
    u32 ax = 0xffffffffu, bx = 0xffffffffu;
    if (ai < na) { ak = s_k[bg_slot(ai)]; ax = s_i[bg_slot(ai)]; }
    if (bi < nb) { bk = s_k[bg_slot(na + bi)]; bx = s_i[bg_slot(na + bi)]; }
#pragma unroll
    for (int q = 0; q < BG_IPT; ++q) {
        const bool ta = !BG_LT(bk, bx, ak, ax);
        vk[q] = ta ? ak : bk;
        vi[q] = ta ? ax : bx;
        ai += ta ? 1u : 0u;
        bi += ta ? 0u : 1u;
        if (q + 1 < BG_IPT) {
            const u32 ni = ta ? ai : bi, lim = ta ? na : nb;
            const u32 at = bg_slot((ta ? 0u : na) + min(ni, lim ? lim - 1 : 0u));
            const u64 nk = ni < lim ? s_k[at] : ~0ull;
            const u32 nx = ni < lim ? s_i[at] : 0xffffffffu;
            ak = ta ? nk : ak;
            ax = ta ? nx : ax;
            bk = ta ? bk : nk;
            bx = ta ? bx : nx;
        }
    }
}

// Every tile sorted by (key, suffix) in LDS: in -> out at the same positions.
__global__ __launch_bounds__(BG_BLOCK) void bg_tile_sort_kernel(const u64 *ik, const u32 *ii, const BigTile *tiles, u32 bound, u64 *ok, u32 *oi)
{
    __shared__ u64 s_k[BG_TILE + BG_TILE / 8];
    __shared__ u32 s_i[BG_TILE + BG_TILE / 8];
    const u32 tid = threadIdx.x;
    const u32 i0 = tid * BG_IPT;
    for (u32 ti = blockIdx.x; ti < bound; ti += gridDim.x) {
        const BigTile T = tiles[ti];
        if (T.gsize == 0) continue;
        const u32 start = T.gstart + T.t * BG_TILE;
        const u32 size = min(BG_TILE, T.gsize - T.t * BG_TILE);
        for (u32 r = tid; r < BG_TILE; r += BG_BLOCK) {
            s_k[bg_slot(r)] = r < size ? ik[start + r] : ~0ull;
            s_i[bg_slot(r)] = r < size ? ii[start + r] : 0xffffffffu;
        }
        __syncthreads();
        u64 vk[BG_IPT];
        u32 vi[BG_IPT];
#pragma unroll
        for (int q = 0; q < BG_IPT; ++q) {
            vk[q] = s_k[bg_slot(i0 + q)];
            vi[q] = s_i[bg_slot(i0 + q)];
        }
        BG_CSWAP(0, 1) BG_CSWAP(2, 3) BG_CSWAP(4, 5) BG_CSWAP(6, 7)
        BG_CSWAP(0, 2) BG_CSWAP(1, 3) BG_CSWAP(4, 6) BG_CSWAP(5, 7)
        BG_CSWAP(1, 2) BG_CSWAP(5, 6)
        BG_CSWAP(0, 4) BG_CSWAP(1, 5) BG_CSWAP(2, 6) BG_CSWAP(3, 7)
        BG_CSWAP(2, 4) BG_CSWAP(3, 5)
        BG_CSWAP(1, 2) BG_CSWAP(3, 4) BG_CSWAP(5, 6)
#pragma unroll
        for (int q = 0; q < BG_IPT; ++q) {
            s_k[bg_slot(i0 + q)] = vk[q];
            s_i[bg_slot(i0 + q)] = vi[q];
        }
        __syncthreads();
        const bool live = i0 < size;
        for (u32 L = BG_IPT; L < BG_TILE; L <<= 1) {
            if (L >= size) break;
            const u32 pair0 = i0 & ~(2 * L - 1);
            const u32 d = i0 - pair0;
            const u32 A = pair0, B = pair0 + L;
            const bool work = live && B < size;
            if (work) {
                u32 lo = d > L ? d - L : 0, hi = d < L ? d : L;
                while (lo < hi) {
                    const u32 mid = (lo + hi) >> 1;
                    const u32 sa = bg_slot(A + mid), sb = bg_slot(B + d - 1 - mid);
                    if (BG_LT(s_k[sa], s_i[sa], s_k[sb], s_i[sb])) lo = mid + 1; else hi = mid;
                }
                u32 ai = lo, bi = d - lo;
                u64 ak = ~0ull, bk = ~0ull;
                u32 ax = 0xffffffffu, bx = 0xffffffffu;
                if (ai < L) { ak = s_k[bg_slot(A + ai)]; ax = s_i[bg_slot(A + ai)]; }
                if (bi < L) { bk = s_k[bg_slot(B + bi)]; bx = s_i[bg_slot(B + bi)]; }
#pragma unroll
                for (int q = 0; q < BG_IPT; ++q) {
                    const bool ta = !BG_LT(bk, bx, ak, ax);
                    vk[q] = ta ? ak : bk;
                    vi[q] = ta ? ax : bx;
                    ai += ta ? 1u : 0u;
                    bi += ta ? 0u : 1u;
                    if (q + 1 < BG_IPT) {
                        const u32 ni = ta ? ai : bi;
                        const u32 at = bg_slot((ta ? A : B) + min(ni, L - 1));
                        const u64 nk = ni < L ? s_k[at] : ~0ull;
                        const u32 nx = ni < L ? s_i[at] : 0xffffffffu;
                        ak = ta ? nk : ak;
                        ax = ta ? nx : ax;
                        bk = ta ? bk : nk;
                        bx = ta ? bx : nx;
                    }
                }
            }
            __syncthreads();
            if (work) {
#pragma unroll
                for (int q = 0; q < BG_IPT; ++q) {
                    s_k[bg_slot(i0 + q)] = vk[q];
                    s_i[bg_slot(i0 + q)] = vi[q];
                }
            }
            __syncthreads();
        }
        for (u32 r = tid; r < size; r += BG_BLOCK) {
            ok[start + r] = s_k[bg_slot(r)];
            oi[start + r] = s_i[bg_slot(r)];
        }
        __syncthreads();
    }
}

// One merge pass: runs of L elements (sorted, inside their group, counted from the group's start) become runs of 2 L.
// Tile t of a group of more than L members = outputs [4096 (t mod R), + 4096) of pair t / R, R = 2 L / 4096.  A pair
// without a second run is copied (the group changes buffers as a whole).
__global__ __launch_bounds__(BG_BLOCK) void bg_merge_kernel(const u64 *sk, const u32 *si, u64 *dk, u32 *di, const BigTile *tiles, u32 bound, u32 L)
{
    __shared__ u64 s_k[BG_TILE + BG_TILE / 8];
    __shared__ u32 s_i[BG_TILE + BG_TILE / 8];
    __shared__ u32 s_part[2];
    const u32 tid = threadIdx.x;
    const u32 R = 2 * (L / BG_TILE);
    for (u32 ti = blockIdx.x; ti < bound; ti += gridDim.x) {
        const BigTile T = tiles[ti];
        if (T.gsize <= L) continue;                        // (unused descriptor, or a group that is sorted already)
        const u32 gend = T.gstart + T.gsize;
        const u32 base = T.gstart + (T.t / R) * 2 * L;
        const u32 left = gend - base;                      // elements of this pair of runs
        const u32 lenA = min(L, left), lenB = left > L ? min(L, left - L) : 0u;
        const u32 diag0 = (T.t % R) * BG_TILE, diag1 = min(diag0 + BG_TILE, lenA + lenB);
        if (lenB == 0) {
            for (u32 r = diag0 + tid; r < diag1; r += BG_BLOCK) {
                dk[base + r] = sk[base + r];
                di[base + r] = si[base + r];
            }
            continue;
        }
        if (tid < 128) {
            // the two merge-path searches, one wave each, SIXTY-FOUR probes at a time (the predicate "A[mid] < B[d - 1 - mid]"
            // is true up to the split and false from it on: a ballot over evenly spaced probes narrows the range 64-fold --
            // three or four rounds of dependent global reads where a binary search by one thread made two dozen)
            const u32 lane = tid & 63u;
            const u32 d = tid < 64 ? diag0 : diag1;
            u32 lo = d > lenB ? d - lenB : 0, hi = d < lenA ? d : lenA;
            while (lo < hi) {
                const u32 span = hi - lo, step = (span + 63u) / 64u;
                const u32 mid = lo + lane * step;
                bool pred = false;
                if (mid < hi) {
                    const u32 a = base + mid, b = base + L + d - 1 - mid;
                    pred = BG_LT(sk[a], si[a], sk[b], si[b]);
                }
                const u32 ncand = (span + step - 1) / step;                 // probes inside [lo, hi)
                const u32 cnt = (u32)__popcll(__ballot(pred));              // leading probes that are true
                const u32 nlo = cnt ? lo + (cnt - 1) * step + 1 : lo;
                const u32 nhi = cnt < ncand ? lo + cnt * step : hi;
                lo = nlo;
                hi = nhi;
            }
            if (lane == 0) s_part[tid < 64 ? 0 : 1] = lo;
        }
        __syncthreads();
        const u32 a0 = s_part[0], a1 = s_part[1];
        const u32 b0 = diag0 - a0, b1 = diag1 - a1;
        const u32 na = a1 - a0, nb = b1 - b0;              // na + nb = diag1 - diag0 <= 4096
        for (u32 r = tid; r < na + nb; r += BG_BLOCK) {
            const u32 src = r < na ? base + a0 + r : base + L + b0 + (r - na);
            s_k[bg_slot(r)] = sk[src];
            s_i[bg_slot(r)] = si[src];
        }
        __syncthreads();
        u64 vk[BG_IPT];
        u32 vi[BG_IPT];
        const bool live = tid * BG_IPT < na + nb;
        if (live) bg_merge_lds(s_k, s_i, na, nb, tid, vk, vi);
        __syncthreads();
        if (live) {
#pragma unroll
            for (int q = 0; q < BG_IPT; ++q) {
                s_k[bg_slot(tid * BG_IPT + q)] = vk[q];
                s_i[bg_slot(tid * BG_IPT + q)] = vi[q];
            }
        }
        __syncthreads();
        for (u32 r = tid; r < na + nb; r += BG_BLOCK) {
            dk[base + diag0 + r] = s_k[bg_slot(r)];
            di[base + diag0 + r] = s_i[bg_slot(r)];
        }
        __syncthreads();
    }
}

// The sorted groups back into the round's output arrays: compacted element i of a group lives in the buffer its
// number of merge passes left it in, and goes to list position bt[i].
__global__ __launch_bounds__(256) void bg_writeback_kernel(const u64 *k0, const u32 *i0, const u64 *k1, const u32 *i1, const BigTile *tiles,
                                                             u32 bound, const u32 *bt, u64 *okey, u32 *oidx)
{
    for (u32 ti = blockIdx.x; ti < bound; ti += gridDim.x) {
        const BigTile T = tiles[ti];
        if (T.gsize == 0) continue;
        u32 passes = 0;
        for (u64 L = BG_TILE; L < (u64)T.gsize; L <<= 1) ++passes;
        const bool in1 = (passes & 1u) == 0;               // the tile sort wrote buffer 1, every pass changes sides
        const u64 *k = in1 ? k1 : k0;
        const u32 *x = in1 ? i1 : i0;
        const u32 start = T.gstart + T.t * BG_TILE, size = min(BG_TILE, T.gsize - T.t * BG_TILE);
        for (u32 r = threadIdx.x; r < size; r += blockDim.x) {
            const u32 dst = bt[start + r];
            okey[dst] = k[start + r];
            oidx[dst] = x[start + r];
        }
    }
}
#undef BG_LT
#undef BG_CSWAP

// ---- periodic runs inside a rank round (round 4) ----------------------------------------------------------------
// Prefix doubling resolves a run of period p and length L in log2(L / h) rounds, every one of them over nearly all of
// the run: at depth h the suffixes of one phase form one group, their keys ISA[i + h] are the (equal) ranks of another
// phase, and only those within 2 h of the run's end come apart.  The order inside such a group is known without
// looking further than the end of the run, though.  Let the group's common h-prefix have period p <= h, and let
// l(i) >= h be how far that period goes on from member i (T[i + x] = T[i + x - p] for p <= x < l(i), not at x = l(i)).
// Members i, j with l(i) < l(j) agree on l(i) symbols -- both continue the same prefix periodically -- and then i has
// its break symbol T[i + l(i)] where j has the periodic one, T[i + l(i) - p]: i < j iff the break symbol is the smaller
// (type L; the end of the string is the smallest symbol), whatever l(j) is.  So the group is ordered by
//     ( type L: 0, l ascending | type G: 1, l descending ),  then the rank of the suffix at the break, i + l(i),
// and members that tie on all three share l + h >= 2 h symbols: a valid doubling round, finer than it need be.
// The members of a run are found from their positions: i and i + p (p <= h) in one group means T[i .. i + p + h) has
// period p, so in position order the members of a run are a chain of steps p, and l(i) = l(z) + z - i for the chain's
// last member z, whose l(z) < h + p comes from at most p symbol comparisons.  A group takes the periodic key when its
// steps <= h all equal one p and at least half of its members have such a step; every other group keeps ISA[i + h].
// Only the groups beyond the LDS sorts (> 3072 members) are looked at: shorter runs need a dozen rounds at most.
struct PerSyms {
    const u32 *names;     // the symbols of an integer string, or
    const u8 *codes;      // the codes of the text
    u32 n;
};
__device__ __forceinline__ long long per_sym(const PerSyms &y, u64 i)
{
    if (i >= y.n) return -1;
    return y.names ? (long long)y.names[i] : (long long)y.codes[i];
}

__global__ __launch_bounds__(256) void per_pack_kernel(const u32 *bt, const u32 *bgid, const u32 *idx, u32 nbig, int idx_bits,
                                                         u64 *pk, u32 *pv)
{
    for (u32 e = blockIdx.x * blockDim.x + threadIdx.x; e < nbig; e += gridDim.x * blockDim.x) {
        pk[e] = ((u64)bgid[e] << idx_bits) | (u64)idx[bt[e]];
        pv[e] = e;
    }
}

// step[r] = distance to the next member of the same group in position order (0: none); pmin[g] = smallest step <= h;
// gsize[g] = members.  The list is sorted by group and every wave walks a contiguous piece of it, keeping the tallies
// of the group it is in and handing them over when the group changes: a handful of atomics per wave, not one per row
// (1.1 M atomics on seven addresses cost 23 ms at 18 M members).
__device__ __forceinline__ u32 per_wave_min(u32 v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = min(v, (u32)__shfl_xor((int)v, o));
    return v;
}

__global__ __launch_bounds__(256) void per_steps_kernel(const u64 *pk, u32 nbig, int idx_bits, u32 h, u32 *step, u32 *pmin,
                                                          u32 *gsize)
{
    const u64 mask = (1ull << idx_bits) - 1;
    const u32 waves = gridDim.x * (blockDim.x / kWave), wid = blockIdx.x * (blockDim.x / kWave) + wave_id();
    const u32 nrow = (nbig + kWave - 1) / kWave, per = (nrow + waves - 1) / waves;
    const u32 row0 = min(nrow, wid * per), row1 = min(nrow, row0 + per);
    u32 cg = 0xffffffffu, csize = 0, cmin = 0xffffffffu;
    auto flush = [&]() {
        if (cg != 0xffffffffu && lane_id() == 0) {
            atomicAdd(&gsize[cg], csize);
            if (cmin != 0xffffffffu) atomicMin(&pmin[cg], cmin);
        }
    };
    for (u32 row = row0; row < row1; ++row) {
        const u32 r = row * kWave + lane_id();
        const bool valid = r < nbig;
        u32 g = 0xffffffffu, d = 0;
        if (valid) {
            const u64 a = pk[r];
            g = (u32)(a >> idx_bits);
            if (r + 1 < nbig) {
                const u64 c = pk[r + 1];
                if ((u32)(c >> idx_bits) == g) d = (u32)((c & mask) - (a & mask));
            }
            step[r] = d;
        }
        u64 todo = __ballot(valid);
        while (todo) {
            const int first = __ffsll((unsigned long long)todo) - 1;
            const u32 g0 = __shfl(g, first);
            const bool in = valid && g == g0;
            const u64 same = __ballot(in);
            const u32 mn = per_wave_min((in && d != 0 && d <= h) ? d : 0xffffffffu);
            if (g0 != cg) {
                flush();
                cg = g0;
                csize = 0;
                cmin = 0xffffffffu;
            }
            csize += (u32)__popcll(same);
            cmin = min(cmin, mn);
            todo &= ~same;
        }
    }
    flush();
}

// links[g] = members whose step is pmin[g]; bad[g] = some step <= h is another one; out[1] += members whose step
// equals their successor's (a run whose period the depth has not reached yet shows up like that), out[2] = the smallest such step.
__global__ __launch_bounds__(256) void per_check_kernel(const u64 *pk, const u32 *step, u32 nbig, int idx_bits, u32 h,
                                                          const u32 *pmin, u32 *links, u32 *bad, u32 *out)
{
    const u32 waves = gridDim.x * (blockDim.x / kWave), wid = blockIdx.x * (blockDim.x / kWave) + wave_id();
    const u32 nrow = (nbig + kWave - 1) / kWave, per = (nrow + waves - 1) / waves;
    const u32 row0 = min(nrow, wid * per), row1 = min(nrow, row0 + per);
    u32 cg = 0xffffffffu, clinks = 0, cbad = 0, carith = 0, cstep = 0xffffffffu;
    auto flush = [&]() {
        if (cg != 0xffffffffu && lane_id() == 0) {
            if (clinks) atomicAdd(&links[cg], clinks);
            if (cbad) atomicOr(&bad[cg], 1u);
        }
    };
    for (u32 row = row0; row < row1; ++row) {
        const u32 r = row * kWave + lane_id();
        const bool valid = r < nbig;
        u32 g = 0xffffffffu, d = 0;
        bool link = false, wrong = false, arith = false;
        if (valid) {
            g = (u32)(pk[r] >> idx_bits);
            d = step[r];
            const u32 p = pmin[g];
            link = d != 0 && d == p;
            wrong = d != 0 && d <= h && d != p;
            arith = d != 0 && r + 1 < nbig && step[r + 1] == d;
        }
        carith += (u32)__popcll(__ballot(arith));
        cstep = min(cstep, per_wave_min(arith ? d : 0xffffffffu));
        u64 todo = __ballot(valid);
        while (todo) {
            const int first = __ffsll((unsigned long long)todo) - 1;
            const u32 g0 = __shfl(g, first);
            const u64 same = __ballot(valid && g == g0);
            const u64 bl = __ballot(link && g == g0), bw = __ballot(wrong && g == g0);
            if (g0 != cg) {
                flush();
                cg = g0;
                clinks = 0;
                cbad = 0;
            }
            clinks += (u32)__popcll(bl);
            cbad |= bw ? 1u : 0u;
            todo &= ~same;
        }
    }
    flush();
    if (lane_id() == 0 && carith) {
        atomicAdd(&out[1], carith);
        atomicMin(&out[2], cstep);      // the shortest step that repeats: no period below it can show up later
    }
}

constexpr u32 PER_MAX_PERIOD = 1u << 16;
// pg[g] = the period the group's key is made with, or 0: the group keeps the plain key.  out[0] += members of periodic groups.
__global__ __launch_bounds__(256) void per_decide_kernel(const u32 *pmin, const u32 *gsize, const u32 *links, const u32 *bad,
                                                           u32 ngroups, u32 *pg, u32 *out)
{
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    u32 mine = 0;
    if (g < ngroups) {
        const u32 p = pmin[g];
        const bool yes = p != 0xffffffffu && p <= PER_MAX_PERIOD && !bad[g] && (u64)links[g] * 2 >= (u64)gsize[g];
        pg[g] = yes ? p : 0u;
        if (yes) mine = gsize[g];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mine += (u32)__shfl_xor((int)mine, o);
    if (lane_id() == 0 && mine) atomicAdd(&out[0], mine);
}

// flag[r] = member r ends a chain of a periodic group (no successor at the group's step)
__global__ __launch_bounds__(256) void per_flag_kernel(const u64 *pk, const u32 *step, u32 nbig, int idx_bits, const u32 *pg,
                                                         u32 *flag)
{
    for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < nbig; r += gridDim.x * blockDim.x) {
        const u32 p = pg[(u32)(pk[r] >> idx_bits)];
        flag[r] = (p != 0 && step[r] != p) ? 1u : 0u;
    }
}

// One wave per chain end z (the c[r]-th): l(z) by comparing symbols from h on (fewer than p of them hold), the type of the
// break, the rank of the suffix at the break.
__global__ __launch_bounds__(256) void per_ends_kernel(const u64 *pk, const u32 *flag, const u64 *c, u32 nbig, int idx_bits,
                                                         const u32 *pg, u32 h, PerSyms y, const u32 *ISA, u32 *epos, u32 *eell,
                                                         u64 *etail)
{
    const u64 mask = (1ull << idx_bits) - 1;
    const u32 lane = lane_id();
    const u32 waves = gridDim.x * (blockDim.x / kWave);
    const u32 nrow = (nbig + kWave - 1) / kWave;
    for (u32 row = blockIdx.x * (blockDim.x / kWave) + wave_id(); row < nrow; row += waves) {
        const u32 r = row * kWave + lane;
        const bool mine = r < nbig && flag[r];
        u64 todo = __ballot(mine);
        while (todo) {
            const int src = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const u32 rr = row * kWave + (u32)src;
            const u64 a = pk[rr];
            const u32 z = (u32)(a & mask), p = pg[(u32)(a >> idx_bits)];
            const u64 lim = (u64)y.n - z;      // (l(z) < h + p when the groups are the classes of depth h; they may be finer)
            u64 ell = lim;
            for (u64 x0 = h; x0 < lim; x0 += kWave) {
                const u64 x = x0 + lane;
                const bool differs = x < lim && per_sym(y, z + x) != per_sym(y, z + x - p);
                const u64 bd = __ballot(differs);
                if (bd) { ell = x0 + (u64)(__ffsll((unsigned long long)bd) - 1); break; }
            }
            if ((int)lane == src) {
                const u64 k = c[rr];
                const long long brk = per_sym(y, (u64)z + ell), per = per_sym(y, (u64)z + ell - p);
                const u64 type = brk < per ? 0ull : 1ull;
                const u64 rank = ((u64)z + ell < y.n) ? (u64)ISA[(u64)z + ell] : 0ull;
                epos[k] = z;
                eell[k] = (u32)ell;
                etail[k] = (type << 63) | rank;
            }
        }
    }
}

// The key of every member of a periodic group: ( type | l or its complement | rank at the break ), into the list of the
// large groups and into the key plane of the round (the write-back and the regrouping read it there).
__global__ __launch_bounds__(256) void per_keys_kernel(const u64 *pk, const u32 *pv, const u64 *c, u32 nbig, int idx_bits,
                                                         const u32 *pg, const u32 *epos, const u32 *eell, const u64 *etail,
                                                         const u32 *bt, u64 *bkey, u64 *slot_key)
{
    const u64 mask = (1ull << idx_bits) - 1;
    for (u32 r = blockIdx.x * blockDim.x + threadIdx.x; r < nbig; r += gridDim.x * blockDim.x) {
        const u64 a = pk[r];
        if (pg[(u32)(a >> idx_bits)] == 0) continue;
        const u64 k = c[r];                       // chain ends before r = the ordinal of the end of r's chain
        const u64 ell = (u64)(epos[k] - (u32)(a & mask)) + eell[k];
        const u64 t = etail[k];
        const u64 field = (t >> 63) ? (0x7fffffffull - ell) : ell;
        const u64 key = (t & (1ull << 63)) | (field << 32) | (t & 0xffffffffull);
        const u32 e = pv[r];
        bkey[e] = key;
        slot_key[bt[e]] = key;
    }
}

__global__ __launch_bounds__(256) void iota_kernel(u32 *v, u32 n)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}

// Switching from text rounds to doubling rounds: rank of every suffix.
__global__ __launch_bounds__(256) void isa_from_sa_kernel(const u32 *SA, u32 n, u32 *ISA)
{
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x)
        ISA[SA[j] & 0x7fffffffu] = j + 1;   // bit 31 may still carry a tie flag of the initial sort
}
__global__ __launch_bounds__(256) void isa_active_kernel(const u32 *idx, const u32 *grp, u32 m, u32 *ISA)
{
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < m; t += gridDim.x * blockDim.x) ISA[idx[t]] = grp[t];
}

// ---- are the ties repeats? -------------------------------------------------------------------------------------
// Before the first text round: a few thousand neighbours of the active list that sit in the same group are compared
// for 48 symbols beyond what they are known to share.  Natural text parts ways within a dozen symbols (a text round
// resolves most of its ties); copies of a block or of a line do not, and every text round over them is a pass over
// the whole list for nothing (33 ms at n = 2^29) -- those go straight to the anchor round.
// out[0] = pairs looked at, out[1] = pairs equal on all 48 symbols.
__global__ __launch_bounds__(256) void probe_repeats_kernel(const u32 *idx, const u32 *grp, u32 m, u32 samples, u32 n, u32 h,
                                                              const u8 *codes, u32 *out)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 pair = 0, same = 0;
    if (t < samples && m >= 2) {
        const u32 stride = (m - 1) / samples;
        u64 x = ((u64)t + 1) * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29;
        const u32 at = stride ? t * stride + (u32)(x % stride) : t % (m - 1);
        if (grp[at] == grp[at + 1]) {
            pair = 1;
            const u64 i = (u64)idx[at] + h, j = (u64)idx[at + 1] + h;
            if (i + 48 <= n && j + 48 <= n) {
                same = 1;
                for (u32 c = 0; c < 48; ++c)
                    if (codes[i + c] != codes[j + c]) { same = 0; break; }
            }
        }
    }
    const u64 bp = __ballot(pair != 0), bs = __ballot(same != 0);
    if (lane_id() == 0) {
        if (bp) atomicAdd(&out[0], (u32)__popcll(bp));
        if (bs) atomicAdd(&out[1], (u32)__popcll(bs));
    }
}

// ---- sizing the initial sort from a sample ---------------------------------
// The initial sort costs one pass per 8 key bits; every suffix it leaves tied
// costs about ten times a pass's per-element price in the rounds.  How many
// suffixes W key bits leave tied depends on the data, not only on the symbol
// frequencies (natural text repeats far more than i.i.d. symbols do), so it is
// measured: S stratified random suffixes, their full-width keys sorted, and for
// every W = 8 P the sample members that share their top W bits with a sorted
// neighbour counted.  A member collides inside the sample with probability
// (group size - 1) * S / n, hence tied fraction ~= count / S * n / S (an
// overestimate when groups are large -- the safe direction).
__global__ __launch_bounds__(256) void sample_keys_kernel(const u8 *codes, u32 n, u32 S, int b, int kmax, int plus_one,
                                                            u64 *keys, u32 *vals)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S) return;
    const u32 stride = n / S;
    u64 x = ((u64)t + 1) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32;
    const u32 pos = t * stride + (u32)(x % stride);
    const int bits = kmax * b;
    keys[t] = text_key_at(codes, pos, b, kmax, plus_one, n) << (64 - bits);
    vals[t] = t;
}

// tied[8]: sample members whose 48th successor still shares their top 20 key bits -- a joint bucket of the
// MSD path (msd_sort.hip) with >= 49 of the S sample members holds about 49 n / S suffixes, far beyond
// what a workgroup sorts in LDS: any such member rules that path out before it starts.
// tied[9]: distinct 20-bit prefixes in the sample; n / distinct estimates the average non-empty bucket
// (exact when there are far fewer buckets than sample members), and a path whose AVERAGE bucket is close
// to the tile limit will not pass the exact check either (`lines` at n = 2^30: 6 000 per bucket).
constexpr u32 kMsdScreenRun = 48;
__global__ __launch_bounds__(256) void sample_ties_kernel(const u64 *keys, u32 S, u32 *tied /* [8]: W = 8, 16, .. 64; [8] screen */)
{
    __shared__ u32 s_c[8];
    if (threadIdx.x < 8) s_c[threadIdx.x] = 0;
    __syncthreads();
    u32 c[8] = {};
    u32 crowded = 0, distinct = 0;
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < S; t += gridDim.x * blockDim.x) {
        const u64 k = keys[t];
        const u64 dp = t > 0 ? (keys[t - 1] ^ k) : ~0ull, dn = t + 1 < S ? (keys[t + 1] ^ k) : ~0ull;
        // equal top W bits with a neighbour <=> its xor has at least W leading zeros
        const int lz = max(dp ? __builtin_clzll(dp) : 64, dn ? __builtin_clzll(dn) : 64);
#pragma unroll
        for (int w = 0; w < 8; ++w) c[w] += lz >= 8 * (w + 1) ? 1u : 0u;
        if (t + kMsdScreenRun < S && ((keys[t + kMsdScreenRun] ^ k) >> 44) == 0) ++crowded;
        if (t == 0 || (dp >> 44) != 0) ++distinct;          // first sample member of its 20-bit prefix
    }
    if (crowded) atomicAdd(&tied[8], crowded);
    if (distinct) atomicAdd(&tied[9], distinct);
#pragma unroll
    for (int w = 0; w < 8; ++w)
        if (c[w]) atomicAdd(&s_c[w], c[w]);
    __syncthreads();
    if (threadIdx.x < 8 && s_c[threadIdx.x]) atomicAdd(&tied[threadIdx.x], s_c[threadIdx.x]);
}

// -------------------------------------------------------------------- host --

static void rerank_geometry(u32 m, RerankArgs &a)
{
    a.m = m;
    a.num_tiles = (u32)(((u64)m + RR_TILE - 1) / RR_TILE);
    a.tiles_per_range = (a.num_tiles + RR_MAX_RANGES - 1) / RR_MAX_RANGES;
    if (a.tiles_per_range == 0) a.tiles_per_range = 1;
    a.num_ranges = (a.num_tiles + a.tiles_per_range - 1) / a.tiles_per_range;
}

enum Slot { S_CODES = 0, S_K0, S_K1, S_V0, S_V1, S_ISA, S_P0, S_P1, S_GRP, S_WORK, S_GRP2 = 26, S_BIG = 27, S_SCR = 29, S_SSA = 30, S_SSB = 31, S_ANC = 32 /* .. 38: one per level */, S_X0 = 39 /* .. 45 */, S_SSPLAN = 48, S_PER = 49, S_BGT = 54, S_ANCW = 55 };      // (10 .. 23, 28: search.hip; 24, 25: capi.cpp)

// Initial key width.  Model the text as i.i.d. with per-symbol collision
// probability c = sum p_i^2 (from the sampled counts): two suffixes agree on k
// symbols with probability c^k, so about n * c^k of the suffixes stay tied.
// Pick the smallest k that leaves <= 1/4096 of them tied (the sparse path
// finishes those almost for free), then widen k to fill the last radix pass.
// A wrong guess costs speed only: whatever stays tied goes to the doubling rounds.
static int choose_key_chars(const u32 *counts, u32 n, int b, int kmax)
{
    double tot = 0, c = 0;
    for (int i = 0; i < 256; ++i) tot += counts[i];
    if (tot <= 0) return kmax;
    for (int i = 0; i < 256; ++i) {
        const double p = counts[i] / tot;
        c += p * p;
    }
    if (c >= 0.999999) return kmax;
    const double need = (12.0 + log2((double)n)) / -log2(c);
    int k = (int)ceil(need);
    if (k < 1) k = 1;
    if (k > kmax) k = kmax;
    const int passes = (k * b + 7) / 8;
    k = (passes * 8) / b;
    if (k > kmax) k = kmax;
    // Within one pass of the full 64 bits the model is not trusted to save that pass: real text
    // repeats far more than i.i.d. symbols do, and then every extra initial symbol pays
    // (measured on `words`: 12 symbols / 8 passes beats 11 / 7 by 4 %).
    if ((kmax * b + 7) / 8 - passes <= 1) k = kmax;
    return k;
}

// Environment switches of the builder (exploration and tests; read on every call so a test can
// flip them between builds).  None of them changes the result.
struct Knobs {
    int key_chars = 0;          // PSS_KEY_CHARS  force the symbols packed into the initial key (0 = choose)
    int key_drop = -1;          // PSS_KEY_DROP   force the low bits of the last symbol left out (-1 = choose)
    bool no_sample = false;     // PSS_NO_SAMPLE  size the initial key from symbol counts even for large n
    bool no_flags = false;      // PSS_NO_TIES_PASS  plain 8-byte-key passes + key comparison in the rerank
    int mode = -1;              // PSS_MODE       dense / sparse / text tie resolution (-1 = choose)
    int text_rounds_max = 5;    // PSS_TEXT_ROUNDS
    int msd = -1;               // PSS_MSD        0: never the MSD initial sort, 1: whenever the key fits, unset: screened
    bool no_plan = false;       // PSS_NO_PLAN_CACHE  always take the sizing sample (never reuse the previous build's choice of sort)
    bool no_front = false;      // PSS_NO_PLAN_FRONT  reuse the choice of sort, but not the alphabet (separate alphabet and recode passes)
    int ss = -1;                // PSS_SS         0: never the sample sort over 16-byte elements, 1: whenever the text has the size for it,
                                //                unset: n >= 2^24 and the MSD sort did not take the text
    bool no_msd_fuse = false;   // PSS_MSD_NO_FUSE  MSD sort flags ties in the suffix array; the rerank kernels read them
    bool no_mid_tier = false;   // PSS_NO_MID_TIER  groups above 512 members all take the chained radix sorts
    int big_merge = 0;          // PSS_BIG_MERGE  1: groups above 4096 members through the segmented merge sort (bg_*_kernel) instead of the
                                //                chained radix sorts, 2: in text rounds only.  Measured at 2^29 and left OFF: real files
                                //                111.8 / 112.7 vs 112.6 / 113.5 ms, `source` 166.6 vs 171.5, `mixed` 90.7 vs 87.2 (its
                                //                groups of millions take twelve merge passes where the radix sorts take seven)
    bool no_mid_merge = false;  // PSS_NO_MID_MERGE  groups of 513 .. 4096 members with a crowded bin take the chained sorts (no LDS merge sort)
    int period = -1;            // PSS_PERIOD     0: never the closed form for texts that repeat one word (rle_build.h)
    int rle = -1;               // PSS_RLE        0: never the run-length path, 1: always, unset: when runs average >= 8 bytes
    int anchor = -1;            // PSS_ANCHOR     0: never the anchor round for ties that outlive the text rounds (rank rounds over the
                                //                whole text instead), 1: whenever ties outlive them, unset: texts of >= 2^20 bytes
    int anchor_omega = 0;       // PSS_ANCHOR_OMEGA  force the window of the minimizers (0 = as wide as the known common prefix allows)
    bool no_probe = false;      // PSS_NO_PROBE   always a text round before the anchor round (no sampling of the ties)
    int anchor_min_omega = 11;  // PSS_ANCHOR_MIN_OMEGA  narrowest window the anchor round accepts by itself
    int probe_skip_pct = 50;    // PSS_PROBE_SKIP_PCT  no text rounds when more than this share of the sampled tied pairs are repeats
    int side = -1;              // PSS_ANCHOR_SIDE  0: never sort the anchors beside the text round, 1: whenever a text round precedes the
                                //                anchor round, unset: texts of >= 2^24 bytes whose sampled ties show copies
    int side_pct = 8;           // PSS_ANCHOR_SIDE_PCT  ... at least this share of the sampled tied pairs
    int anchor_cap_div = 5;     // PSS_ANCHOR_CAP_DIV  the anchor round declines when the windows choose more than n / this many anchors
    bool count_sort = false;    // PSS_COUNT_SORT  rank rounds: groups ranked by counting (group_sort_kernel) instead of the merge sort
    bool no_periodic = false;   // PSS_PERIODIC=0  rank rounds: no periodic keys for the large groups (per_*_kernel)
    bool timing = false;        // PSS_TIMING     per-round trace on stderr
    static Knobs read()
    {
        Knobs k;
        if (const char *e = getenv("PSS_KEY_CHARS")) k.key_chars = atoi(e);
        if (const char *e = getenv("PSS_KEY_DROP")) k.key_drop = atoi(e);
        k.no_sample = getenv("PSS_NO_SAMPLE") != nullptr;
        k.no_flags = getenv("PSS_NO_TIES_PASS") != nullptr;
        if (const char *e = getenv("PSS_MODE")) {
            if (!strcmp(e, "dense")) k.mode = 0;
            else if (!strcmp(e, "sparse")) k.mode = 1;
            else if (!strcmp(e, "text")) k.mode = 2;
        }
        if (const char *e = getenv("PSS_TEXT_ROUNDS")) k.text_rounds_max = atoi(e);
        if (const char *e = getenv("PSS_MSD")) k.msd = atoi(e);
        if (const char *e = getenv("PSS_SS")) k.ss = atoi(e);
        k.no_plan = getenv("PSS_NO_PLAN_CACHE") != nullptr;
        k.no_front = getenv("PSS_NO_PLAN_FRONT") != nullptr;
        k.no_msd_fuse = getenv("PSS_MSD_NO_FUSE") != nullptr;
        k.no_mid_tier = getenv("PSS_NO_MID_TIER") != nullptr;
        k.no_mid_merge = getenv("PSS_NO_MID_MERGE") != nullptr;
        if (const char *e = getenv("PSS_BIG_MERGE")) k.big_merge = atoi(e);
        if (getenv("PSS_NO_BIG_MERGE")) k.big_merge = 0;
        if (const char *e = getenv("PSS_RLE")) k.rle = atoi(e);
        if (const char *e = getenv("PSS_PERIOD")) k.period = atoi(e);
        if (const char *e = getenv("PSS_ANCHOR")) k.anchor = atoi(e);
        if (const char *e = getenv("PSS_ANCHOR_OMEGA")) k.anchor_omega = atoi(e);
        k.no_probe = getenv("PSS_NO_PROBE") != nullptr;
        if (const char *e = getenv("PSS_PROBE_SKIP_PCT")) k.probe_skip_pct = atoi(e);
        if (const char *e = getenv("PSS_ANCHOR_MIN_OMEGA")) k.anchor_min_omega = std::max(2, atoi(e));
        if (const char *e = getenv("PSS_ANCHOR_SIDE")) k.side = atoi(e);
        if (const char *e = getenv("PSS_ANCHOR_SIDE_PCT")) k.side_pct = atoi(e);
        if (const char *e = getenv("PSS_ANCHOR_CAP_DIV")) k.anchor_cap_div = std::min(5, std::max(3, atoi(e)));
        k.count_sort = getenv("PSS_COUNT_SORT") != nullptr;
        { const char *e = getenv("PSS_PERIODIC"); k.no_periodic = e && atoi(e) == 0; }
        k.timing = getenv("PSS_TIMING") != nullptr;
        return k;
    }
};

// start / stop events of one build, destroyed on every exit path
struct BuildTimer {
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mid = nullptr;
    ~BuildTimer()
    {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (ev_mid) (void)hipEventDestroy(ev_mid);
    }
};

// Sizing of the initial sort from a sorted sample (see sample_keys_kernel): the fewest passes
// that leave <= 2 % of the suffixes tied, else the full kmax symbols.  K / V are free scratch.
static int size_initial_key(DeviceCtx *ctx, const u8 *codes, u32 n, int b, int kmax, int plus_one, u64 *K[2], u32 *V[2],
                            void *work, u32 *d_tied, u32 *h_small, bool profile, SortStats *ss, int *key_chars,
                            int *key_drop, bool *msd_screen_ok)
{
    hipStream_t s = ctx->stream;
    const u32 S = 1u << 21;
    const int bits_max = kmax * b, pmax = (bits_max + 7) / 8;
    PSS_HIP(hipMemsetAsync(d_tied, 0, 64, s));
    hipLaunchKernelGGL(sample_keys_kernel, dim3(S / 256), dim3(256), 0, s, codes, n, S, b, kmax, plus_one, K[0], V[0]);
    u32 mask = 0;
    for (int p = 0; p < 8; ++p)
        if (8 * (p + 1) > 64 - 8 * (pmax - 1)) mask |= 1u << p;   // only the top 8 (pmax - 1) bits are ever compared
    int sd = 0;
    const u64 launches = ss->launches, elems = ss->elems;
    PSS_TRY(radix_sort_pairs(ctx, K, V, S, 64, mask, nullptr, 0, work, &sd, profile, ss));
    ss->launches = launches;   // not passes of the suffix sort (their profile figures stay in: same kernel, same stream)
    ss->elems = elems;
    hipLaunchKernelGGL(sample_ties_kernel, dim3(256), dim3(256), 0, s, K[sd], S, d_tied);
    PSS_HIP(hipMemcpyAsync(h_small, d_tied, 64, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipStreamSynchronize(s));
    *msd_screen_ok = h_small[8] == 0 && h_small[9] != 0 && (double)n / (double)h_small[9] <= 3400.0;
    *key_chars = kmax;
    *key_drop = 0;
    for (int p = 2; p < pmax; ++p) {
        const double est = (double)h_small[p - 1] / S * ((double)n / S);
        if (est <= 0.02) {
            *key_chars = (8 * p + b - 1) / b;
            *key_drop = *key_chars * b - 8 * p;
            break;
        }
    }
    return PSS_OK;
}

// Everything after the initial sort: rerank + compaction of the tied suffixes, then rounds until no
// ties are left.  Its own function since round 2: the run-length path (rle_build below) sorts the
// suffixes of an INTEGER string (one symbol per run of the text) with the same rounds -- there is no
// text to pack keys from then (codes == nullptr): rank rounds only, starting from h0 = 1.
struct RoundsIO {
    u32 n;
    u32 *SA;
    u64 *K[2];
    u32 *V[2];
    u32 *ISA;
    u32 *P[2];
    u32 *GRP;
    const u8 *codes;            // recoded text (nullptr: rank rounds only)
    int b, plus_one, key_chars, key_drop;
    u64 h0;                     // symbols every group of the initial sort is known to share
    int cur;                    // K[cur] / V[cur]: sorted keys / suffixes of the initial sort
    int final_buf;              // V[final_buf] was redirected to SA for the initial sort (-1: not)
    u32 *v_scratch;             // ... and this is the buffer it stands for
    bool ties;                  // V[cur] carries tie flags in bit 31 (no keys)
    bool msd_fused;             // the MSD sort already produced the first active list
    u32 msd_active;
    bool no_sparse;             // the initial key does not fit 64 bits (sample sort): the sparse mode's key search cannot be used
    u8 *work;
    u32 *d_agg_head, *d_agg_cnt;
    u64 *d_red;
    u32 *d_counters;
    u32 *h_small;
    bool profile;
    // Subset sort (anchors, anchor_impl.h): the n elements are the suffixes at text positions sub_pos[0 .. n) of a text of
    // text_n symbols; element values are the ordinals.  Text rounds run until every group shares stop_text_h symbols,
    // whatever they resolve; then the group ranks are the symbols of an integer string (element v is followed by v + 1)
    // whose suffixes the rank rounds sort, starting over from h = 1.
    const u32 *sub_pos = nullptr;
    u32 text_n = 0;
    u64 stop_text_h = 0;
    u32 *grp2 = nullptr;        // second group-rank buffer (nullptr: slot S_GRP2 of the context)
    int level = 0;              // 0: the text (or the run-length path's reduced string); k: the names of level k - 1's anchors
    struct SideAnchors *side = nullptr;      // the side line of the build these rounds belong to (it may be under way already)
};

static int anchor_rank_keys(DeviceCtx *ctx, const Knobs &knobs, const RoundsIO &outer, u64 h, const u32 *syms,
                            const u32 *cur_ranks, u32 *akey, pss_sa_stats &st, bool *ok);

// The anchors' own sort BESIDE the text round (round 5).  Which positions are anchors, their names and the order of the
// anchor suffixes depend on the text and on the window only -- not on the active list the text round is busy with.  On
// text with copies in it (source code: 412 MB of real files spent 34 ms in the text round and 50 ms in the anchors'
// sort, one after the other) the anchors are therefore sorted by a second host thread on a second stream, in a context
// of its own (DeviceCtx::helper: its stream, pinned scratch and slots), while the main line runs the text round that
// raises the depth to what the window needs.  Half of the anchors' sort is a long row of small launches (rank rounds over
// lists of 10^4 .. 10^6 elements on three levels) that leave the device all but empty: they fill the gaps of the other
// stream instead of standing in line.  The window is fixed in advance (the depth the text round WILL reach); a text round
// that gives up half-way leaves the depth where it was, and the result of the side line is then thrown away.
struct SideAnchors {
    std::thread th;
    bool started = false, joined = false, ok = false;
    int rc = PSS_OK;
    std::string err;
    u64 h_eff = 0;
    pss_sa_stats st;
    u32 *akey = nullptr;
    void join()
    {
        if (started && !joined) {
            if (th.joinable()) th.join();
            joined = true;
        }
    }
    ~SideAnchors() { join(); }
};

// Starts the side line: anchors of the text `codes` for windows that fit depth h_eff, sorted in ctx's helper context.
// after (optional): an event on the main stream that the helper's stream waits for first (the codes are being written).
// No room for the buffers is not an error: the anchors then wait their turn on the main line as before.
static int side_start(DeviceCtx *ctx, const Knobs &knobs, SideAnchors &side, u32 n, const u8 *codes, int b, int plus_one, u64 h_eff,
                      hipEvent_t after)
{
    if (side.started) return PSS_OK;
    DeviceCtx *hc = nullptr;
    const size_t sort_ws = radix_sort_workspace_bytes();
    int rs = get_helper_ctx(ctx, &hc);
    if (rs == PSS_OK) rs = hc->slot[S_K0].reserve((size_t)n * 8);
    if (rs == PSS_OK) rs = hc->slot[S_K1].reserve((size_t)n * 8);
    if (rs == PSS_OK) rs = hc->slot[S_ISA].reserve((size_t)n * 4 + 64);
    if (rs == PSS_OK) rs = hc->slot[S_WORK].reserve(sort_ws + 65536);
    if (rs != PSS_OK) {
        (void)hipGetLastError();
        set_error("%s", "");
        return PSS_OK;
    }
    if (after) PSS_HIP(hipStreamWaitEvent(hc->stream, after, 0));
    u8 *hw = hc->slot[S_WORK].as<u8>();
    u8 *hsmall = hw + sort_ws;
    RoundsIO o2;
    memset(&o2, 0, sizeof o2);
    o2.n = n;
    o2.K[0] = hc->slot[S_K0].as<u64>();
    o2.K[1] = hc->slot[S_K1].as<u64>();
    o2.codes = codes;
    o2.b = b;
    o2.plus_one = plus_one;
    o2.work = hw;
    o2.d_agg_head = reinterpret_cast<u32 *>(hsmall + 4096);
    o2.d_agg_cnt = reinterpret_cast<u32 *>(hsmall + 8192);
    o2.d_red = reinterpret_cast<u64 *>(hsmall + 12288);
    o2.d_counters = reinterpret_cast<u32 *>(hsmall + 12288 + 64);
    o2.h_small = static_cast<u32 *>(hc->pinned);
    o2.level = 0;
    side.h_eff = h_eff;
    side.akey = hc->slot[S_ISA].as<u32>();
    memset(&side.st, 0, sizeof side.st);
    const int dev = ctx->device;
    SideAnchors *sp = &side;
    // (std::thread's constructor may throw -- no more threads to be had: the flag goes up only once the thread exists, so
    // that the destructor never joins what was never started; the exception travels to the C ABI's catch-all)
    side.th = std::thread([hc, knobs, o2, h_eff, sp, dev]() {
        if (hipSetDevice(dev) != hipSuccess) {
            sp->rc = PSS_EDEVICE;
            sp->err = "hipSetDevice failed in the anchors' side line";
            return;
        }
        try {
            sp->rc = anchor_rank_keys(hc, knobs, o2, h_eff, nullptr, nullptr, sp->akey, sp->st, &sp->ok);
            if (sp->rc != PSS_OK) sp->err = last_error();
        } catch (const std::bad_alloc &) {
            sp->rc = PSS_ENOMEM;
            sp->err = "host allocation failed in the anchors' side line";
        } catch (...) {                      // (nothing may leave a thread's function)
            sp->rc = PSS_EDEVICE;
            sp->err = "internal error in the anchors' side line";
        }
    });
    side.started = true;
    return PSS_OK;
}

static int refine_rounds(DeviceCtx *ctx, const Knobs &knobs, RoundsIO &io, SortStats &ss, pss_sa_stats &st)
{
    hipStream_t s = ctx->stream;
    const u32 n = io.n;
    u32 *SA = io.SA;
    u64 *K[2] = {io.K[0], io.K[1]};
    u32 *V[2] = {io.V[0], io.V[1]};
    u32 *ISA = io.ISA;
    u32 *P[2] = {io.P[0], io.P[1]};
    u32 *GRP = io.GRP;
    const u8 *codes = io.codes;
    const bool rank_only = codes == nullptr;
    const bool subset = io.sub_pos != nullptr;
    const u32 text_n = subset ? io.text_n : n;
    bool anchored = false;       // the anchor round has run: nothing may be left tied
    const int b = io.b, plus_one = io.plus_one, key_chars = io.key_chars;
    const bool profile = io.profile;
    u8 *work = io.work;
    u32 *d_agg_head = io.d_agg_head, *d_agg_cnt = io.d_agg_cnt, *d_counters = io.d_counters, *h_small = io.h_small;
    u64 *d_red = io.d_red;
    const int grid_stream = ctx->num_cus * 8;
    int cur = io.cur;
    const int final_buf = io.final_buf;
    u32 *const v_scratch = io.v_scratch;
    const bool ties = io.ties, msd_fused = io.msd_fused;
    const u32 msd_active = io.msd_active;
    const bool sa_in_place = (final_buf >= 0 && cur == final_buf);

    int rank_bits = 1;
    while ((1ull << rank_bits) <= (u64)n) ++rank_bits;      // ranks 0..n
    RerankArgs ra;
    memset(&ra, 0, sizeof ra);
    ra.agg_head = d_agg_head;
    ra.agg_cnt = d_agg_cnt;
    ra.SA = SA;
    ra.ISA = ISA;
    ra.counters = d_counters;
    ra.ht = reinterpret_cast<u64 *>(ISA);     // the two modes never coexist
    ra.rank_bits = rank_bits;
    u32 m = n;
    int pcur = 0;                // P[pcur] holds the SA positions of the active list (after round 0)
    int gcur = 0;                // G[gcur] holds its group ranks
    u32 *G[2] = {GRP, nullptr};
    bool identity_pos = true;
    enum Mode { M_DENSE = 0, M_SPARSE = 1, M_TEXT = 2 };
    Mode mode = M_DENSE;
    bool was_text = false;
    int text_rounds = 0;
    u32 m_text_prev = 0;
    bool keyed_grp = true;       // the sorted keys of the previous round carry the group rank in their high half
    u32 global_above = 0;        // rank rounds use ONE global (group, rank) sort while m stays above this
    double last_big_frac = 0.0;
    const int text_rounds_max = knobs.text_rounds_max;
    int kt = 64 / b;             // symbols per text-round key
    if (kt > 16) kt = 16;
    const int k0buf = cur;       // K[k0buf] = sorted initial keys (kept intact in sparse mode)
    u64 *SK[2] = {nullptr, nullptr};   // sparse mode: small ping-pong key buffers
    u64 **Kr = K;
    u64 h = io.h0;               // symbols every group is known to share
    const u32 grid_all = (u32)grid_stream;
    // Integer strings (rank rounds only): the ranks the first round leaves ARE the string, up to renaming -- kept for the
    // minimizers of an anchor level on top of this one (anchor_impl.h), should the rounds reach depth 32 with much left tied.
    SideAnchors own_side;
    SideAnchors &side = io.side ? *io.side : own_side;
    u32 *X0 = nullptr;
    const u32 *Xsym = nullptr;   // the same snapshot for the periodic keys of the rank rounds (per_*_kernel), never given up
    bool last_per = false;       // the last rank round met periodic runs among its large groups, or chains that will be
    u64 per_wait_h = 0;          // ... and the depth from which their period can be seen
    auto snapshot_symbols = [&]() -> int {
        const bool for_levels = !(knobs.anchor == 0 || io.level >= 6 || n < (knobs.anchor == 1 ? 64u : (1u << 20)));
        if (!for_levels && (knobs.no_periodic || io.level >= 6 || n <= 3072u)) return PSS_OK;
        PSS_TRY(ctx->slot[S_X0 + io.level].reserve((size_t)n * 4));
        u32 *snap = ctx->slot[S_X0 + io.level].as<u32>();
        PSS_HIP(hipMemcpyAsync(snap, ISA, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        if (for_levels) X0 = snap;
        Xsym = snap;
        return PSS_OK;
    };
    for (int round = 0;; ++round) {
        if (round > 96) {
            set_error("sa_build: no convergence after 96 rounds (internal error)");
            return PSS_EDEVICE;
        }
        rerank_geometry(m, ra);
        ra.keys = Kr[cur];
        ra.idx = V[cur];
        ra.SA = (round == 0 && sa_in_place) ? nullptr : SA;      // round 0: the sort already wrote SA
        ra.tied_sa = (round == 0 && ties && !msd_fused) ? V[cur] : nullptr;
        ra.pos = identity_pos ? nullptr : P[pcur];
        ra.grp = (round > 0 && !keyed_grp) ? G[gcur] : nullptr;   // group-local rounds: keys do not carry the group
        ra.pos_out = P[pcur ^ 1];
        ra.idx_out = V[cur ^ 1];
        if (round == 0) {
            if (io.grp2) G[1] = io.grp2;
            else {
                PSS_TRY(ctx->slot[S_GRP2].reserve((size_t)n * 4));
                G[1] = ctx->slot[S_GRP2].as<u32>();
            }
        }
        ra.grp_out = G[gcur ^ 1];
        const bool fused0 = round == 0 && msd_fused;     // the MSD local sort already produced this round's active list
        if (!fused0) {
            if (ra.tied_sa) hipLaunchKernelGGL(rr_reduce_tied_kernel, dim3(ra.num_ranges), dim3(RR_BLOCK), 0, s, ra);
            else hipLaunchKernelGGL(rr_reduce_kernel, dim3(ra.num_ranges), dim3(RR_BLOCK), 0, s, ra);
            hipLaunchKernelGGL(rr_scan_kernel, dim3(1), dim3(1024), 0, s, d_agg_head, d_agg_cnt, ra.num_ranges, d_counters);
            PSS_HIP(hipMemcpyAsync(h_small, d_counters, 4, hipMemcpyDeviceToHost, s));
            PSS_HIP(hipStreamSynchronize(s));
        }
        const u32 m_next = fused0 ? msd_active : h_small[0];
        if (round == 0) {
            // few ties: sparse (hash + key search); otherwise extend the ties from the text first
            mode = ((u64)m_next * 1024 <= (u64)n) ? M_SPARSE : M_TEXT;
            if (knobs.mode >= 0) mode = (Mode)knobs.mode;
            if (mode == M_SPARSE && (u64)m_next * 16 > (u64)n) mode = M_TEXT;   // hash table must fit the ISA buffer
            if (mode == M_SPARSE && io.no_sparse && m_next) mode = M_TEXT;
            if (m_next == 0) mode = M_SPARSE;                                   // nothing left: no ISA at all
            if (mode == M_TEXT && text_rounds_max <= 0) mode = M_DENSE;
            if (rank_only && m_next) mode = M_DENSE;                            // no text to pack keys from
            if (subset && m_next) mode = h < io.stop_text_h ? M_TEXT : M_DENSE;
            was_text = mode == M_TEXT;
        }
        if (fused0) {
            if (mode == M_DENSE && m_next) {      // rank rounds from the start: they need the inverse suffix array
                hipLaunchKernelGGL(isa_from_sa_kernel, dim3(grid_all), dim3(256), 0, s, SA, n, ISA);
                hipLaunchKernelGGL(isa_active_kernel, dim3(grid_all), dim3(256), 0, s, ra.idx_out, ra.grp_out, m_next, ISA);
            }
        } else if (mode == M_DENSE) hipLaunchKernelGGL(rr_apply_kernel<MODE_ISA>, dim3(ra.num_ranges), dim3(RR_BLOCK), 0, s, ra);
        else if (mode == M_SPARSE && round > 0) hipLaunchKernelGGL(rr_apply_kernel<MODE_HT>, dim3(ra.num_ranges), dim3(RR_BLOCK), 0, s, ra);
        else if (ra.tied_sa && ra.SA == nullptr && ra.pos == nullptr)
            hipLaunchKernelGGL(rr_apply_tied_kernel, dim3(ra.num_ranges), dim3(RR_BLOCK), 0, s, ra);
        else hipLaunchKernelGGL(rr_apply_kernel<MODE_NONE>, dim3(ra.num_ranges), dim3(RR_BLOCK), 0, s, ra);
        PSS_HIP(hipGetLastError());
        if (round == 0 && final_buf >= 0) V[final_buf] = v_scratch;   // later rounds must not scribble over SA
        if (m_next == 0) break;
        if (anchored && mode == M_DENSE) {
            // cannot happen: the anchor round leaves no ties (and its keys took the place of the inverse array)
            st.anchor_left += m_next;
            set_error("sa_build: %u elements tied after the anchor round of level %d (internal error)", m_next, io.level);
            return PSS_EDEVICE;
        }
        if (round == 0 && mode == M_DENSE && subset) h = 1;           // (no text round was needed: the elements are symbols already)
        if (round == 0 && mode == M_DENSE && (subset || rank_only)) PSS_TRY(snapshot_symbols());
        // (subset mode counts h in SYMBOLS of the text while its text rounds run, in elements afterwards)
        if (h >= (subset && mode == M_TEXT ? (u64)text_n : (u64)n)) {
            set_error("sa_build: %u suffixes unresolved at h=%llu >= n (internal error)", m_next,
                      (unsigned long long)h);
            return PSS_EDEVICE;
        }
        m = m_next;
        pcur ^= 1;
        gcur ^= 1;
        identity_pos = false;
        const int src = cur ^ 1;             // V[src] = compacted suffix indices, G[gcur] their groups, P[pcur] their slots
        const u32 grid = (u32)std::min<u64>((u64)grid_stream, ((u64)m + 255) / 256);

        // ------------------------------------------------------ group-local round --
        // One round over the active list: a 64-bit key per suffix, every group sorted by it.
        //   text round : key = next symbols packed from the text at offset h      (h += kt)
        //   rank round : key = rank of suffix i+h from the inverse suffix array    (h *= 2)
        // Groups of <= GS_CAP members are ranked in LDS (group_sort); members of larger groups
        // go through two chained stable radix sorts (key, then dense group number).
        // Text rounds advance h linearly; they pay off while each round resolves most ties
        // (natural-language LCPs).  When a round leaves more than 60 % of its list tied, or large
        // groups dominate, the data is repetitive: rank rounds, logarithmic in the LCP, take over.
        auto local_round = [&](bool use_text, bool *bail, const u32 *key_of_suffix = nullptr) -> int {
            *bail = false;
            const u32 nblk = (m + GS_T - 1) / GS_T;
            PSS_TRY(ctx->slot[S_SCR].reserve((size_t)m + (size_t)nblk * 24 + (SC_MAX_BLOCKS + 8) * 8 + 4096 +
                                             ((size_t)m / 2 + 16) * sizeof(MidGroup) + ((size_t)m / 512 + 16) * sizeof(MidGroup) + 512));
            u8 *scr = ctx->slot[S_SCR].as<u8>();
            size_t o = 0;
            auto carve = [&](size_t bytes) { u8 *p = scr + o; o = round_up(o + bytes, 64); return p; };
            u8 *d_big = carve(m);
            u32 *d_blk_big = reinterpret_cast<u32 *>(carve((size_t)nblk * 4));
            u32 *d_blk_heads = reinterpret_cast<u32 *>(carve((size_t)nblk * 4));
            u64 *d_off_big = reinterpret_cast<u64 *>(carve(((size_t)nblk + 1) * 8));
            u64 *d_off_heads = reinterpret_cast<u64 *>(carve(((size_t)nblk + 1) * 8));
            u64 *d_partial = reinterpret_cast<u64 *>(carve((SC_MAX_BLOCKS + 2) * 8));
            u64 *d_total = d_partial + SC_MAX_BLOCKS;
            const u32 h32 = (u32)std::min<u64>(h, 0xffffffffull);
            if (use_text)
                hipLaunchKernelGGL(text_keys_kernel, dim3(grid), dim3(256), 0, s, V[src], m, text_n, h32, codes, b, kt, plus_one,
                                   K[src], io.sub_pos);
            else if (key_of_suffix)      // anchor round: the key of suffix i is key_of_suffix[i] (the rank of the anchor its window chose)
                hipLaunchKernelGGL(rank_keys_kernel, dim3(grid), dim3(256), 0, s, V[src], m, n, 0u, key_of_suffix, K[src]);
            else
                hipLaunchKernelGGL(rank_keys_kernel, dim3(grid), dim3(256), 0, s, V[src], m, n, h32, ISA, K[src]);
            if (use_text)
                hipLaunchKernelGGL(group_sort_kernel<false>, dim3(nblk), dim3(256), 0, s, K[src], V[src], G[gcur], m, K[src ^ 1],
                                   V[src ^ 1], d_big, d_blk_big, d_blk_heads);
            else if (knobs.count_sort)
                hipLaunchKernelGGL(group_sort_kernel<true>, dim3(nblk), dim3(256), 0, s, K[src], V[src], G[gcur], m, K[src ^ 1],
                                   V[src ^ 1], d_big, d_blk_big, d_blk_heads);
            else
                hipLaunchKernelGGL(group_msort32_kernel, dim3(nblk), dim3(GM_BLOCK), 0, s, K[src], V[src], G[gcur], m, K[src ^ 1],
                                   V[src ^ 1], d_big, d_blk_big, d_blk_heads);
            if (!knobs.no_mid_tier) {
                // groups of up to MID_CAP members: one workgroup each, in LDS (no host round trip: the list and its
                // length stay on the device, the workgroups persist and walk over it)
                MidGroup *d_mid = reinterpret_cast<MidGroup *>(carve(((size_t)m / 2 + 16) * sizeof(MidGroup)));
                MidGroup *d_mid_fail = reinterpret_cast<MidGroup *>(carve(((size_t)m / 512 + 16) * sizeof(MidGroup)));   // (groups of > 512 members)
                u32 *d_mid_count = reinterpret_cast<u32 *>(carve(64));
                PSS_HIP(hipMemsetAsync(d_mid_count, 0, 8, s));
                hipLaunchKernelGGL(mid_collect_kernel, dim3(grid), dim3(256), 0, s, d_big, G[gcur], m, d_mid, d_mid_count);
                hipLaunchKernelGGL(mid_sort_kernel, dim3((u32)ctx->num_cus * 4), dim3(MID_BLOCK), 0, s, K[src], V[src], d_mid,
                                   (const u32 *)d_mid_count, use_text ? kt * b : rank_bits, K[src ^ 1], V[src ^ 1], d_big, d_blk_big,
                                   d_blk_heads, knobs.no_mid_merge ? (MidGroup *)nullptr : d_mid_fail, d_mid_count + 1);
                if (!knobs.no_mid_merge)
                    hipLaunchKernelGGL(mid_msort_kernel, dim3((u32)ctx->num_cus * 3), dim3(MID_BLOCK), 0, s, K[src], V[src], d_mid_fail,
                                       (const u32 *)(d_mid_count + 1), K[src ^ 1], V[src ^ 1], d_big, d_blk_big, d_blk_heads);
            }
            PSS_TRY(device_excl_scan(ctx, InU32{d_blk_big}, nblk, d_partial, d_total, d_off_big));
            PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
            PSS_HIP(hipStreamSynchronize(s));
            const u32 nbig = h_small[0];
            if (knobs.timing)
                fprintf(stderr, "[pss] %s round: h=%llu m=%u large-group members=%u (%.1f%%)\n", use_text ? "text" : "rank",
                        (unsigned long long)h, m, nbig, 100.0 * nbig / m);
            last_big_frac = (double)nbig / (double)m;
            if (nbig == 0) return PSS_OK;
            if (use_text && !subset && (u64)nbig * 2 > (u64)m && (text_rounds > 0 || (u64)nbig * 4 > (u64)m * 3)) {
                *bail = true;
                return PSS_OK;
            }
            PSS_TRY(ctx->slot[S_BIG].reserve((size_t)nbig * (4 + 4 + 16 + 8) + 1024));
            u8 *bscr = ctx->slot[S_BIG].as<u8>();
            size_t bo = 0;
            auto bcarve = [&](size_t bytes) { u8 *p = bscr + bo; bo = round_up(bo + bytes, 64); return p; };
            u32 *d_bt = reinterpret_cast<u32 *>(bcarve((size_t)nbig * 4));
            u32 *d_bgid = reinterpret_cast<u32 *>(bcarve((size_t)nbig * 4));
            u64 *BK[2] = {reinterpret_cast<u64 *>(bcarve((size_t)nbig * 8)), reinterpret_cast<u64 *>(bcarve((size_t)nbig * 8))};
            u32 *BV[2] = {reinterpret_cast<u32 *>(bcarve((size_t)nbig * 4)), reinterpret_cast<u32 *>(bcarve((size_t)nbig * 4))};
            PSS_TRY(device_excl_scan(ctx, InU32{d_blk_heads}, nblk, d_partial, d_total, d_off_heads));
            PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
            hipLaunchKernelGGL(big_compact_kernel, dim3(nblk), dim3(256), 0, s, d_big, G[gcur], K[src], m, d_off_big,
                               d_off_heads, d_bt, BK[0], d_bgid);
            PSS_HIP(hipStreamSynchronize(s));
            const u32 nbig_groups = h_small[0];
            int gid_bits = 1;
            while ((1ull << gid_bits) < (u64)nbig_groups) ++gid_bits;
            int big_key_bits = use_text ? kt * b : rank_bits;
            last_per = false;
            const bool per_syms = (subset || rank_only) ? Xsym != nullptr : codes != nullptr;
            if (per_wait_h > h && nbig >= 2) last_per = true;      // (chains seen, their period still ahead of the depth)
            if (!use_text && !key_of_suffix && !knobs.no_periodic && per_syms && nbig >= 2 && h < (1ull << 31) && h >= per_wait_h) {
                // periodic runs among the large groups: their members get the key that orders them at once (per_*_kernel)
                int idx_bits = 1;
                while ((1ull << idx_bits) < (u64)n) ++idx_bits;
                const size_t g4 = round_up((size_t)nbig_groups * 4, 64), e4 = round_up((size_t)nbig * 4, 64), e8 = round_up((size_t)nbig * 8 + 8, 64);
                if (ctx->slot[S_PER].reserve(e8 + e4 + e4 + e4 + e8 + e4 + e4 + e8 + 5 * g4 + 256) == PSS_OK) {
                    u8 *pb = ctx->slot[S_PER].as<u8>();
                    size_t po = 0;
                    auto pcarve = [&](size_t bytes) { u8 *q = pb + po; po += bytes; return q; };
                    u64 *PK[2] = {reinterpret_cast<u64 *>(pcarve(e8)), BK[1]};
                    u32 *PV[2] = {reinterpret_cast<u32 *>(pcarve(e4)), BV[1]};
                    u32 *d_step = reinterpret_cast<u32 *>(pcarve(e4)), *d_flag = reinterpret_cast<u32 *>(pcarve(e4));
                    u64 *d_c = reinterpret_cast<u64 *>(pcarve(e8));
                    u32 *d_epos = reinterpret_cast<u32 *>(pcarve(e4)), *d_eell = reinterpret_cast<u32 *>(pcarve(e4));
                    u64 *d_etail = reinterpret_cast<u64 *>(pcarve(e8));
                    u32 *d_pmin = reinterpret_cast<u32 *>(pcarve(g4));
                    u32 *d_gsize = reinterpret_cast<u32 *>(pcarve(4 * g4 + 256));        // gsize, links, bad, pg, out: zeroed together
                    u32 *d_links = d_gsize + g4 / 4, *d_bad = d_links + g4 / 4, *d_pg = d_bad + g4 / 4, *d_out = d_pg + g4 / 4;
                    const u32 pgrid = std::min<u32>((nbig + 255) / 256, (u32)ctx->num_cus * 16);
                    const u32 wgrid = std::min<u32>((nbig + 255) / 256, (u32)ctx->num_cus * 4);      // (waves that walk contiguous pieces)
                    PSS_HIP(hipMemsetAsync(d_pmin, 0xff, g4, s));
                    PSS_HIP(hipMemsetAsync(d_gsize, 0, 4 * g4 + 256, s));
                    PSS_HIP(hipMemsetAsync(d_out + 2, 0xff, 4, s));
                    hipLaunchKernelGGL(per_pack_kernel, dim3(pgrid), dim3(256), 0, s, d_bt, d_bgid, V[src], nbig, idx_bits, PK[0], PV[0]);
                    int dp = 0;
                    SortStats sp;
                    PSS_TRY(radix_sort_pairs(ctx, PK, PV, nbig, gid_bits + idx_bits, 0xffffffffu, nullptr, 0, work, &dp, false, &sp));
                    hipLaunchKernelGGL(per_steps_kernel, dim3(wgrid), dim3(256), 0, s, PK[dp], nbig, idx_bits, h32, d_step, d_pmin, d_gsize);
                    hipLaunchKernelGGL(per_check_kernel, dim3(wgrid), dim3(256), 0, s, PK[dp], d_step, nbig, idx_bits, h32, d_pmin,
                                       d_links, d_bad, d_out);
                    hipLaunchKernelGGL(per_decide_kernel, dim3((nbig_groups + 255) / 256), dim3(256), 0, s, d_pmin, d_gsize, d_links,
                                       d_bad, nbig_groups, d_pg, d_out);
                    PSS_HIP(hipMemcpyAsync(h_small, d_out, 12, hipMemcpyDeviceToHost, s));
                    PSS_HIP(hipStreamSynchronize(s));
                    const u32 per_members = h_small[0], arith = h_small[1];
                    // most of the list in chains whose step the depth has not reached: nothing to find before it has
                    per_wait_h = (per_members == 0 && (u64)arith * 2 >= (u64)nbig && h_small[2] != 0xffffffffu) ? h_small[2] : 0;
                    if (knobs.timing)
                        fprintf(stderr, "[pss] rank round: h=%llu large-group members=%u in periodic groups=%u, equal steps=%u\n",
                                (unsigned long long)h, nbig, per_members, arith);
                    last_per = per_members != 0 || (u64)arith * 4 >= (u64)nbig;
                    if (per_members) {
                        const PerSyms y{(subset || rank_only) ? Xsym : nullptr, (subset || rank_only) ? nullptr : codes, n};
                        hipLaunchKernelGGL(per_flag_kernel, dim3(pgrid), dim3(256), 0, s, PK[dp], d_step, nbig, idx_bits, d_pg, d_flag);
                        PSS_TRY(device_excl_scan(ctx, InU32{d_flag}, nbig, d_partial, d_total, d_c));
                        hipLaunchKernelGGL(per_ends_kernel, dim3(pgrid), dim3(256), 0, s, PK[dp], d_flag, d_c, nbig, idx_bits, d_pg, h32,
                                           y, ISA, d_epos, d_eell, d_etail);
                        hipLaunchKernelGGL(per_keys_kernel, dim3(pgrid), dim3(256), 0, s, PK[dp], PV[dp], d_c, nbig, idx_bits, d_pg, d_epos,
                                           d_eell, d_etail, d_bt, BK[0], K[src]);
                        big_key_bits = 64;
                        st.periodic_rounds += 1;
                        st.periodic_members += per_members;
                    }
                    st.round_passes += (u32)sp.launches;
                } else {
                    (void)hipGetLastError();
                    set_error("%s", "");
                }
            }
            if ((knobs.big_merge == 1 || (knobs.big_merge == 2 && use_text && big_key_bits > 32)) && nbig < 0x7fffffffu) {
                // segmented merge sort of the large groups (bg_*_kernel): tiles in LDS, then merge passes inside every group
                const u32 bound = nbig / BG_TILE + nbig_groups + 2;
                const size_t g4 = round_up(((size_t)nbig_groups + 2) * 4, 64), g8 = round_up(((size_t)nbig_groups + 2) * 8, 64);
                PSS_TRY(ctx->slot[S_BGT].reserve(g4 + g8 + round_up((size_t)bound * sizeof(BigTile), 64) + 256));
                u8 *tb = ctx->slot[S_BGT].as<u8>();
                u32 *d_gstart = reinterpret_cast<u32 *>(tb);
                u64 *d_toff = reinterpret_cast<u64 *>(tb + g4);
                BigTile *d_tiles = reinterpret_cast<BigTile *>(tb + g4 + g8);
                const u32 eg = std::min<u32>((nbig + 255) / 256, (u32)ctx->num_cus * 16);
                hipLaunchKernelGGL(bg_gstart_kernel, dim3(eg), dim3(256), 0, s, d_bgid, nbig, nbig_groups, d_gstart);
                PSS_TRY(device_excl_scan(ctx, InTileCount{d_gstart}, nbig_groups, d_partial, d_total, d_toff));
                PSS_HIP(hipMemsetAsync(d_tiles, 0, (size_t)bound * sizeof(BigTile), s));
                hipLaunchKernelGGL(bg_tiles_kernel, dim3(std::min<u32>((nbig_groups + 3) / 4, (u32)ctx->num_cus * 8)), dim3(256), 0, s, d_gstart,
                                   d_toff, nbig_groups, d_tiles);
                hipLaunchKernelGGL(bg_gather_kernel, dim3(eg), dim3(256), 0, s, d_bt, V[src], nbig, BV[0]);
                const u32 wg = std::min<u32>(bound, (u32)ctx->num_cus * 2);
                hipLaunchKernelGGL(bg_tile_sort_kernel, dim3(wg), dim3(BG_BLOCK), 0, s, BK[0], BV[0], d_tiles, bound, BK[1], BV[1]);
                int from = 1;
                for (u64 L = BG_TILE; L < (u64)nbig; L <<= 1) {
                    hipLaunchKernelGGL(bg_merge_kernel, dim3(wg), dim3(BG_BLOCK), 0, s, BK[from], BV[from], BK[from ^ 1], BV[from ^ 1], d_tiles,
                                       bound, (u32)L);
                    from ^= 1;
                }
                hipLaunchKernelGGL(bg_writeback_kernel, dim3(std::min<u32>(bound, (u32)ctx->num_cus * 8)), dim3(256), 0, s, BK[0], BV[0], BK[1], BV[1],
                                   d_tiles, bound, d_bt, K[src ^ 1], V[src ^ 1]);
                st.big_elems += nbig;
                return PSS_OK;
            }
            hipLaunchKernelGGL(iota_kernel, dim3((nbig + 255) / 256), dim3(256), 0, s, BV[0], nbig);
            SortStats s1, s2;
            int d1 = 0, d2 = 0;
            PSS_TRY(radix_sort_pairs(ctx, BK, BV, nbig, big_key_bits, 0xffffffffu, nullptr, 0, work, &d1, profile, &s1));
            hipLaunchKernelGGL(gather_gid_kernel, dim3((nbig + 255) / 256), dim3(256), 0, s, BV[d1], d_bgid, nbig,
                               nbig <= 4096u, BK[d1]);
            if (nbig_groups > 1)
                PSS_TRY(radix_sort_pairs(ctx, BK, BV, nbig, gid_bits, 0xffffffffu, nullptr, d1, work, &d2, profile, &s2));
            else d2 = d1;
            hipLaunchKernelGGL(big_writeback_kernel, dim3((nbig + 255) / 256), dim3(256), 0, s, BV[d2], d_bt, K[src],
                               V[src], nbig, K[src ^ 1], V[src ^ 1]);
            st.round_passes += (u32)(s1.launches + s2.launches);
            ss.launches += s1.launches + s2.launches;
            ss.elems += s1.elems + s2.elems;
            ss.ms += s1.ms + s2.ms;
            ss.ms_pairs += s1.ms_pairs + s2.ms_pairs;
            ss.pairs_launches += s1.pairs_launches + s2.pairs_launches;
            ss.pairs_elems += s1.pairs_elems + s2.pairs_elems;
            st.big_elems += nbig;
            return PSS_OK;
        };
        if (mode == M_TEXT) {
            const bool text_progress = text_rounds == 0 || (u64)m * 10 <= (u64)m_text_prev * 6;
            bool bail = true;
            const bool anchors_on = !subset && knobs.anchor != 0 && (knobs.anchor == 1 || n >= (1u << 20));
            bool skip_text = false;
            const bool probe_skips = h >= (u64)knobs.anchor_min_omega + 3;       // deep enough for the anchor round to run at once
            const bool side_on = knobs.side != 0 && !rank_only && (knobs.side == 1 || n >= (1u << 24)) && io.level == 0;
            bool side_wanted = false;
            if (anchors_on && !anchored && text_rounds == 0 && !knobs.no_probe && m >= 4096 && (u64)m * 16 >= (u64)n && (probe_skips || side_on)) {
                // are these ties repeats (see probe_repeats_kernel)?  Then no text round will resolve them.
                u32 *d_probe = d_counters + 48;
                PSS_HIP(hipMemsetAsync(d_probe, 0, 8, s));
                const u32 samples = 8192;
                hipLaunchKernelGGL(probe_repeats_kernel, dim3(samples / 256), dim3(256), 0, s, V[src], G[gcur], m, samples, n,
                                   (u32)std::min<u64>(h, 0xffffffffull), codes, d_probe);
                PSS_HIP(hipMemcpyAsync(h_small, d_probe, 8, hipMemcpyDeviceToHost, s));
                PSS_HIP(hipStreamSynchronize(s));
                st.probe_pairs = h_small[0];
                st.probe_same = h_small[1];
                skip_text = probe_skips && h_small[0] >= 64 && (u64)h_small[1] * 100 > (u64)h_small[0] * (u64)knobs.probe_skip_pct;
                side_wanted = side_on && !skip_text && h_small[0] >= 64 && (u64)h_small[1] * 100 >= (u64)h_small[0] * (u64)knobs.side_pct;
                if (knobs.timing) fprintf(stderr, "[pss] probe: %u of %u sampled pairs share 48 more symbols%s\n", h_small[1], h_small[0], skip_text ? ": no text rounds" : (side_wanted ? ": anchors beside the text round" : ""));
            }
            if (knobs.side == 1 && side_on && anchors_on && !anchored && text_rounds == 0 && !skip_text) side_wanted = true;
            if (side_wanted && !side.started && text_rounds < text_rounds_max)
                // the depth the coming text round will reach decides the window; the anchors' sort starts now, on the side
                PSS_TRY(side_start(ctx, knobs, side, n, codes, b, plus_one, h + (u64)kt, nullptr));
            if (anchored) {
                // cannot happen: the anchor round leaves no ties.  Counted (tests assert zero) and resolved by rank rounds.
                st.anchor_left += m;
            } else if (skip_text) {
            } else if (subset) {
                if (h < io.stop_text_h) PSS_TRY(local_round(true, &bail));
            } else if (text_rounds < text_rounds_max && text_progress) {
                m_text_prev = m;
                PSS_TRY(local_round(true, &bail));
            }
            if (bail && !anchored && anchors_on) {
                // Ties that outlive the text rounds are repeats: one round keyed by the ranks of the anchors (anchor_impl.h)
                // instead of log2(length of the repeat) rank rounds over the whole text.  The key array takes the place
                // of the inverse suffix array, which this path never builds.
                bool ok = false;
                const u32 *akey = ISA;
                if (side.started) {
                    side.join();
                    if (side.rc != PSS_OK) {
                        set_error("%s", side.err.c_str());
                        return side.rc;
                    }
                    if (side.ok && h >= side.h_eff) {
                        // sorted beside the text round: its keys are valid for every depth from h_eff on
                        const pss_sa_stats &t = side.st;
                        st.anchor = 1;
                        st.anchor_count = t.anchor_count;
                        st.anchor_omega = t.anchor_omega;
                        st.anchor_w = t.anchor_w;
                        st.anchor_ms += t.anchor_ms;
                        st.anchor_depth = h;
                        st.anchor_text_rounds += t.anchor_text_rounds;
                        st.anchor_rounds += t.anchor_rounds;
                        st.anchor_sum_active += t.anchor_sum_active;
                        st.anchor_left += t.anchor_left;
                        st.periodic_rounds += t.periodic_rounds;
                        st.periodic_members += t.periodic_members;
                        st.anchor_levels = std::max<uint64_t>(st.anchor_levels, t.anchor_levels);
                        st.anchor_side = 1;
                        akey = side.akey;
                        ok = true;
                    } else {
                        st.anchor_side = 2;      // thrown away: the text round gave up before it reached the window's depth, or the round declined
                    }
                }
                if (!ok) PSS_TRY(anchor_rank_keys(ctx, knobs, io, h, nullptr, nullptr, ISA, st, &ok));
                if (ok) {
                    st.anchor_active = m;
                    bool b2 = false;
                    PSS_TRY(local_round(false, &b2, akey));
                    anchored = true;
                    keyed_grp = false;
                    cur = src ^ 1;
                    st.rounds += 1;
                    st.sum_active += m;
                    PSS_HIP(hipGetLastError());
                    continue;
                }
            }
            if (!bail) {
                keyed_grp = false;
                cur = src ^ 1;
                st.rounds += 1;
                st.text_rounds += 1;
                st.sum_active += m;
                text_rounds += 1;
                h += (u64)kt;
                PSS_HIP(hipGetLastError());
                continue;
            }
            // ties outlived the text rounds: build the inverse suffix array once, continue with rank rounds
            hipLaunchKernelGGL(isa_from_sa_kernel, dim3(grid_all), dim3(256), 0, s, SA, n, ISA);
            hipLaunchKernelGGL(isa_active_kernel, dim3(grid), dim3(256), 0, s, V[src], G[gcur], m, ISA);
            mode = M_DENSE;
            if (subset) {
                h = 1;               // from here on the elements are the symbols of an integer string
                PSS_TRY(snapshot_symbols());
            }
        }
        if (mode == M_DENSE && X0 && !anchored && h >= 32 && (u64)m * 16 >= (u64)n && m >= (knobs.anchor == 1 ? 64u : (1u << 19))) {
            // a level up: minimizers of this string of symbols, named by the ranks the rounds have reached
            bool ok = false;
            PSS_TRY(anchor_rank_keys(ctx, knobs, io, h, X0, ISA, ISA, st, &ok));
            if (ok) {
                bool b2 = false;
                PSS_TRY(local_round(false, &b2, ISA));
                anchored = true;
                keyed_grp = false;
                cur = src ^ 1;
                st.rounds += 1;
                st.sum_active += m;
                PSS_HIP(hipGetLastError());
                continue;
            }
            X0 = nullptr;            // declined: plain rounds to the end
        }
        // Rank rounds: group-local unless large groups dominate (repetitive data) -- then one
        // global radix sort on (group rank, rank) with constant digits skipped is cheaper than
        // ranking in LDS + compaction + two chained sorts over nearly everything.
        if (mode == M_DENSE && m <= global_above) global_above = 0;
        if (mode == M_DENSE && global_above == 0) {
            bool bail = false;
            PSS_TRY(local_round(false, &bail));
            if (last_big_frac > 0.5 && !last_per) global_above = m / 2;      // (periodic runs: the local rounds know a shortcut)
            keyed_grp = false;
            cur = src ^ 1;
            st.rounds += 1;
            st.sum_active += m;
            h *= 2;
            PSS_HIP(hipGetLastError());
            continue;
        }

        // ------------- global doubling round: sparse (hash table + key search) or dense (ISA) --
        keyed_grp = true;
        if (mode == M_SPARSE && round == 0) {
            // hash table over the initially-active suffixes, in the (unused) ISA buffer
            u32 cap = 1024;
            while (cap < 4u * m) cap <<= 1;
            ra.ht_mask = cap - 1;
            PSS_HIP(hipMemsetAsync(ra.ht, 0, (size_t)cap * 8, s));
            hipLaunchKernelGGL(ht_insert_kernel, dim3(grid), dim3(256), 0, s, ra.ht, ra.ht_mask, V[src], G[gcur], m);
            // small key buffers carved out of the free big key buffer
            SK[0] = K[k0buf ^ 1];
            SK[1] = K[k0buf ^ 1] + (size_t)m;
            Kr = SK;
        }
        h_small[0] = 0; h_small[1] = 0; h_small[2] = 0xffffffffu; h_small[3] = 0xffffffffu;
        PSS_HIP(hipMemcpyAsync(d_red, h_small, 16, hipMemcpyHostToDevice, s));
        KeyArgs ka;
        ka.idx = V[src];
        ka.grp = G[gcur];
        ka.ISA = ISA;
        ka.ht = ra.ht;
        ka.ht_mask = ra.ht_mask;
        ka.sa = SA;
        ka.codes = codes;
        ka.code_bits = b;
        ka.key_chars = key_chars;
        ka.plus_one = plus_one;
        ka.m = m;
        ka.n = n;
        ka.h = (u32)std::min<u64>(h, 0xffffffffull);
        ka.rank_bits = rank_bits;
        ka.keys = Kr[src];
        ka.red = d_red;
        const u32 grid_keys = std::max(1u, std::min(grid, (m + 2047u) / 2048u));   // >= 8 keys per thread: fewer atomics on the two words
        if (mode == M_SPARSE) hipLaunchKernelGGL(build_keys_kernel<true>, dim3(grid), dim3(256), 0, s, ka);   // latency-bound key searches: every wave helps
        else hipLaunchKernelGGL(build_keys_kernel<false>, dim3(grid_keys), dim3(256), 0, s, ka);
        PSS_HIP(hipMemcpyAsync(h_small, d_red, 16, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
        const u64 vor = (u64)h_small[0] | ((u64)h_small[1] << 32);
        const u64 vand = (u64)h_small[2] | ((u64)h_small[3] << 32);
        const u64 varying = vor & ~vand;
        const int key_bits = 2 * rank_bits;
        u32 mask = 0;
        for (int p = 0; p < (key_bits + 7) / 8; ++p)
            if ((varying >> (8 * p)) & 0xffull) mask |= 1u << p;
        SortStats rs;
        PSS_TRY(radix_sort_pairs(ctx, Kr, V, m, key_bits, mask, nullptr, src, work, &cur, profile, &rs));
        st.rounds += 1;
        st.round_passes += (u32)rs.launches;
        st.sum_active += m;
        ss.launches += rs.launches;
        ss.elems += rs.elems;
        ss.ms += rs.ms;
        ss.ms_pairs += rs.ms_pairs;
        ss.pairs_launches += rs.pairs_launches;
        ss.pairs_elems += rs.pairs_elems;
        h *= 2;
    }
    st.mode = was_text ? (mode == M_TEXT ? 2u : 3u) : (u64)mode;
    return PSS_OK;
}

// The key of the anchor round (anchor_impl.h): akey[i] = rank, among the anchor suffixes, of the anchor the window at i
// chose -- for every position i of the string.  `h`: symbols every tied group of the caller's active list shares.
//   syms == nullptr  the string is the text (outer.codes): w = 4 or 8 bytes hashed per position, omega + w - 1 <= h; the
//                    anchors are named by a sort of their own (text rounds to 2 omega + w - 1 symbols), then their names
//                    are a string of 32-bit symbols whose suffixes the rank rounds sort;
//   syms != nullptr  the string is that array of 32-bit symbols (a level up: the names of a coarser level's anchors):
//                    w = 1, omega = h / 2, and the anchors' names are the ranks the caller's rounds have reached
//                    (cur_ranks, depth h >= 2 omega).
// The caller's key buffers K[0], K[1] (8 n bytes each, scratch between two rounds) hold the anchors' own sort; `akey`:
// 4 n bytes (may be cur_ranks).  *ok = false: declined (window too narrow, too many anchors) -- nothing is lost but the
// time of the selection pass.
__global__ __launch_bounds__(256) void gather_names_kernel(const u32 *pos, u32 m, const u32 *ranks, u64 *keys, u32 *vals)
{
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < m; t += gridDim.x * blockDim.x) {
        keys[t] = ranks[pos[t]];
        vals[t] = t;
    }
}

static int anchor_rank_keys(DeviceCtx *ctx, const Knobs &knobs, const RoundsIO &outer, u64 h, const u32 *syms,
                            const u32 *cur_ranks, u32 *akey, pss_sa_stats &st, bool *ok)
{
    *ok = false;
    hipStream_t s = ctx->stream;
    const u32 n = outer.n;
    const u8 *codes = outer.codes;
    const bool forced = knobs.anchor == 1;
    int w;
    u64 omega64;
    if (syms) {
        w = 1;
        omega64 = h / 2;
    } else {
        w = h >= 28 ? 8 : 4;
        omega64 = h >= (u64)w ? h - (u64)w + 1 : 0;
    }
    if (knobs.anchor_omega > 0) omega64 = std::min<u64>(omega64, (u64)knobs.anchor_omega);
    const u32 omega = (u32)std::min<u64>(omega64, 64);       // wider windows: fewer anchors, but names of 2 omega + w - 1 symbols
    if (omega < (forced ? 2u : (u32)knobs.anchor_min_omega) || n < 64 || outer.level >= 6) return PSS_OK;
    const u32 num_tiles = (n + ANC_TILE - 1) / ANC_TILE;
    const size_t n16 = round_up((size_t)n, 16) + 16;
    const u32 cap_div = outer.level == 0 ? (u32)knobs.anchor_cap_div : 5u;
    const u32 m_cap = n / cap_div + 64;
    DevBuf &slot = ctx->slot[S_ANC + outer.level];
    PSS_TRY(slot.reserve(n16 + round_up((size_t)num_tiles * 4, 64) + ((size_t)num_tiles + 2) * 8 + (SC_MAX_BLOCKS + 8) * 8 +
                         (size_t)m_cap * 4 + 1024));
    u8 *base = slot.as<u8>();
    size_t o = 0;
    auto carve = [&](size_t bytes) { u8 *p = base + o; o = round_up(o + bytes, 64); return p; };
    u8 *d_dist = carve(n16);
    u32 *d_tile_cnt = reinterpret_cast<u32 *>(carve((size_t)num_tiles * 4));
    u64 *d_tile_off = reinterpret_cast<u64 *>(carve(((size_t)num_tiles + 2) * 8));
    u64 *d_partial = reinterpret_cast<u64 *>(carve((SC_MAX_BLOCKS + 8) * 8));
    u64 *d_total = d_partial + SC_MAX_BLOCKS;
    u32 *d_Q = reinterpret_cast<u32 *>(carve((size_t)m_cap * 4));
    BuildTimer tm;
    PSS_HIP(hipEventCreate(&tm.ev0));
    PSS_HIP(hipEventCreate(&tm.ev1));
    PSS_HIP(hipEventRecord(tm.ev0, s));
    const u32 n_read = (u32)(round_up((size_t)n, 16) + 64);      // the recoded text's padding (zero)
    const u32 grid = std::min<u32>(num_tiles, (u32)ctx->num_cus * 4);
    if (syms)
        hipLaunchKernelGGL(anc_select_kernel<true>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const u8 *>(syms), n, n, omega, w,
                           d_dist, d_tile_cnt, num_tiles);
    else
        hipLaunchKernelGGL(anc_select_kernel<false>, dim3(grid), dim3(256), 0, s, codes, n, n_read, omega, w, d_dist, d_tile_cnt,
                           num_tiles);
    PSS_TRY(device_excl_scan(ctx, InU32{d_tile_cnt}, num_tiles, d_partial, d_total, d_tile_off));
    u32 *h_small = outer.h_small;
    PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipStreamSynchronize(s));
    const u32 m = h_small[0];
    if (outer.level == 0) {
        st.anchor_count = m;
        st.anchor_omega = omega;
        st.anchor_w = (u64)w;
    }
    if (knobs.timing)
        fprintf(stderr, "[pss] anchors (level %d): n=%u h=%llu omega=%u w=%d anchors=%u (n / %.1f)\n", outer.level, n,
                (unsigned long long)h, omega, w, m, (double)n / std::max(1u, m));
    if (m == 0 || m > n / cap_div) return PSS_OK;
    hipLaunchKernelGGL(anc_walk_kernel<false>, dim3(grid), dim3(256), 0, s, d_dist, n, d_tile_off, num_tiles, d_Q,
                       (const u32 *)nullptr, (u32 *)nullptr);
    // the anchors' own sort, in the caller's two key buffers
    u8 *b0 = reinterpret_cast<u8 *>(outer.K[0]), *b1 = reinterpret_cast<u8 *>(outer.K[1]);
    const size_t m8 = round_up((size_t)m * 8, 256), m4 = round_up((size_t)m * 4 + 64, 256);
    u64 *AK[2] = {reinterpret_cast<u64 *>(b0), reinterpret_cast<u64 *>(b0 + m8)};
    if (outer.level == 0 && 9 * m4 > (size_t)n * 8 && 2 * m8 <= (size_t)n * 8) {
        // more anchors than the caller's second key buffer holds nine arrays of: a slot of their own
        PSS_TRY(ctx->slot[S_ANCW].reserve(9 * m4));
        b1 = ctx->slot[S_ANCW].as<u8>();
    }
    u32 *AV[2] = {reinterpret_cast<u32 *>(b1), reinterpret_cast<u32 *>(b1 + m4)};
    u32 *A_isa = reinterpret_cast<u32 *>(b1 + 2 * m4);
    u32 *AP[2] = {reinterpret_cast<u32 *>(b1 + 3 * m4), reinterpret_cast<u32 *>(b1 + 4 * m4)};
    u32 *A_grp = reinterpret_cast<u32 *>(b1 + 5 * m4);
    u32 *A_grp2 = reinterpret_cast<u32 *>(b1 + 6 * m4);
    u32 *A_sa = reinterpret_cast<u32 *>(b1 + 7 * m4);
    u32 *A_rank = reinterpret_cast<u32 *>(b1 + 8 * m4);
    if (2 * m8 > (size_t)n * 8 || (b1 == reinterpret_cast<u8 *>(outer.K[1]) && 9 * m4 > (size_t)n * 8)) return PSS_OK;      // (tiny strings)
    int kt = 64 / outer.b;
    if (kt > 16) kt = 16;
    const u32 gk = (u32)std::min<u64>((u64)ctx->num_cus * 8, ((u64)m + 255) / 256);
    int cur = 0;
    SortStats ss;
    RoundsIO io;
    if (syms) {
        hipLaunchKernelGGL(gather_names_kernel, dim3(gk), dim3(256), 0, s, d_Q, m, cur_ranks, AK[0], AV[0]);
        int bits = 1;
        while ((1ull << bits) <= (u64)n) ++bits;
        PSS_TRY(radix_sort_pairs(ctx, AK, AV, m, bits, 0xffffffffu, nullptr, 0, outer.work, &cur, false, &ss));
        io.codes = nullptr;
        io.b = 8; io.plus_one = 0; io.key_chars = 1;
        io.h0 = 1;
    } else {
        hipLaunchKernelGGL(subset_keys_kernel, dim3(gk), dim3(256), 0, s, d_Q, m, n, codes, outer.b, kt, outer.plus_one, AK[0], AV[0]);
        PSS_TRY(radix_sort_pairs(ctx, AK, AV, m, kt * outer.b, 0xffffffffu, nullptr, 0, outer.work, &cur, false, &ss));
        io.codes = codes;
        io.b = outer.b; io.plus_one = outer.plus_one; io.key_chars = kt;
        io.h0 = (u64)kt;
        io.sub_pos = d_Q;
        io.text_n = n;
        io.stop_text_h = 2ull * omega + (u64)w - 1;
    }
    io.n = m;
    io.SA = A_sa;
    io.K[0] = AK[0]; io.K[1] = AK[1];
    io.V[0] = AV[0]; io.V[1] = AV[1];
    io.ISA = A_isa;
    io.P[0] = AP[0]; io.P[1] = AP[1];
    io.GRP = A_grp;
    io.grp2 = A_grp2;
    io.key_drop = 0;
    io.cur = cur;
    io.final_buf = -1;
    io.v_scratch = nullptr;
    io.ties = false;
    io.msd_fused = false;
    io.msd_active = 0;
    io.no_sparse = true;
    io.work = outer.work;
    io.d_agg_head = outer.d_agg_head; io.d_agg_cnt = outer.d_agg_cnt; io.d_red = outer.d_red; io.d_counters = outer.d_counters;
    io.h_small = outer.h_small;
    io.profile = false;
    io.level = outer.level + 1;
    pss_sa_stats sub;
    memset(&sub, 0, sizeof sub);
    SortStats ss2;
    PSS_TRY(refine_rounds(ctx, knobs, io, ss2, sub));
    st.anchor_text_rounds += sub.text_rounds;
    st.anchor_rounds += sub.rounds - sub.text_rounds + sub.anchor_rounds;
    st.anchor_sum_active += sub.sum_active + sub.anchor_sum_active;
    st.anchor_left += sub.anchor_left;
    st.periodic_rounds += sub.periodic_rounds;
    st.periodic_members += sub.periodic_members;
    st.anchor_levels = std::max<uint64_t>(st.anchor_levels, 1 + sub.anchor_levels);
    hipLaunchKernelGGL(isa_from_sa_kernel, dim3(gk), dim3(256), 0, s, A_sa, m, A_rank);
    hipLaunchKernelGGL(anc_walk_kernel<true>, dim3(grid), dim3(256), 0, s, d_dist, n, d_tile_off, num_tiles, (u32 *)nullptr,
                       (const u32 *)A_rank, akey);
    PSS_HIP(hipEventRecord(tm.ev1, s));
    PSS_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    PSS_HIP(hipEventElapsedTime(&ms, tm.ev0, tm.ev1));
    if (outer.level == 0) {
        st.anchor_ms += ms;
        st.anchor_depth = h;
    }
    st.anchor = 1;
    *ok = true;
    return PSS_OK;
}

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
        static const int key_cap = [] { const char *e = getenv("PSS_MSD_KEY_CAP"); return e ? atoi(e) : 48; }();
        int kb = std::min(msd_max_key_bits(n), std::max(key_cap, 21));
        const int kc = std::min(kb / b, kmax);                   // whole symbols only
        kb = kc * b;
        const bool fits = kc >= 1 && kb >= 21 && !plus_one;
        const bool auto_ok = sampled && msd_screen_ok && (hint == 1 || (key_bits0 <= 48 && kb + 8 >= key_bits0));
        if (fits && (knobs.msd == 1 || (knobs.msd < 0 && auto_ok))) {
            TextKeys mk{codes, b, kc, plus_one, 0};
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
                key_drop = 0;
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
