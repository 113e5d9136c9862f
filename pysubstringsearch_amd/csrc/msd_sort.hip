// msd_sort.hip -- initial suffix sort as a hybrid MSD radix sort, for gfx950.
//
// The LSD passes of radix_sort.hip move every suffix through HBM once per 8 key bits
// (5 scatter passes + 5 histogram reads for the 40-bit key of the `lines` corpus), and a
// scatter pass is bound by where its short runs land, not by bytes.  This
// path spends TWO global partition passes and finishes inside LDS:
//
//   G1  text -> A1   partition all n suffixes by the top 10 bits of their packed key.  The
//                    element written is 8 bytes: [ remaining key bits | suffix index ], so no
//                    second plane and nothing is gathered from the text again.
//   G2  A1 -> A2     partition every G1 bucket by the next 10 key bits ("segmented": a
//                    workgroup's range never crosses a G1 bucket).  The scan of the
//                    (range, digit) counts IS the table of the 2^20 joint bucket starts.
//   L   A2 -> SA     joint buckets are at most 4096 suffixes (checked exactly, see below):
//                    consecutive buckets are packed into tiles of <= 8192 elements, one
//                    workgroup sorts a tile by (bucket, remaining key bits) entirely in LDS
//                    (stable 8-bit LSD passes, the wave-ballot ranking of radix_sort.hip) and
//                    writes the suffix indices sequentially -- full lines, no scatter -- with
//                    bit 31 = "same key as my predecessor" (the contract of suffix_sort_flags).
//
// Traffic: (1 + 1 + 8) + (8 + 8 + 8) + (8 + 4) = 46 bytes per suffix instead of 86, and two
// scattered passes instead of five.
//
// Neither global pass needs a stable order (whatever order a bucket arrives in, L sorts it),
// so the per-tile ranking is one returning LDS atomic per element -- no ballots, no per-wave
// histograms -- which is what makes 1024 bins affordable: 8192-element tiles keep the runs at
// 8 elements x 8 bytes.
//
// The path needs every joint bucket to fit a tile.  That is a property of the text (high-entropy
// text: yes; natural language: no, "the " alone overflows it), known exactly after G2's
// histogram.  The caller screens with the sorted key sample it already has; when the exact
// check fails the caller falls back to the LSD path (G1 is lost, ~4 ms at n = 2^29).
//
// HBM-bound integer work: no MFMA anywhere by design.
#include "msd_sort.h"

#include "prims.h"
#include "scan.h"
#include "text_keys.h"

namespace pss {

constexpr int MSD_D = 10;
constexpr u32 MSD_BINS = 1u << MSD_D;
constexpr int MSD_BLOCK = 512;
constexpr int MSD_WAVES = MSD_BLOCK / kWave;
constexpr int MSD_IPT = 16;
constexpr u32 MSD_TILE = MSD_BLOCK * MSD_IPT;        // 8192 elements
constexpr u32 MSD_CAPH = 4096;                       // buckets whose start falls into one window of this size share a tile
constexpr u32 MSD_MAX_BUCKET = 4096;                 // => a tile holds < CAPH + MAX_BUCKET = 8192 elements
constexpr u32 MSD_TILE_BUCKETS = 1024;               // and at most this many buckets (10 bits of the LDS sort key)
constexpr u32 MSD_G1_RANGES = 1024;
constexpr u32 MSD_G2_RANGE = 16 * MSD_TILE;          // elements per G2 range (a piece of one G1 bucket)

struct MsdRange {
    u32 seg, start, end;
};

struct MsdArgs {
    // text source (G1)
    const u8 *codes;
    int code_bits, key_chars, plus_one, key_drop;
    u32 n;
    int key_bits;        // K: bits of (packed key >> key_drop) that take part in the sort
    int idx_bits;        // ib: bits of a suffix index
    // G1 geometry
    u32 tiles_per_range1, num_ranges1;
    // tables
    u32 *T;              // [ranges][1024] counts, then global offsets
    u32 *J1;             // [1025] G1 bucket starts
    u32 *J;              // [2^20 + 1] joint bucket starts
    MsdRange *ranges2;   // G2 range descriptors
    u32 *seg_first;      // [1025] first G2 range of every G1 bucket
    u32 *counters;       // [0] number of G2 ranges, [1] largest joint bucket, [2] non-empty buckets, [3] tiles
    const u64 *in;
    u64 *out;
};

// ---- tile machinery of the two global passes -----------------------------------------------------

template <bool FROM_TEXT>
__device__ __forceinline__ void msd_load_tile(const MsdArgs &a, u32 base, u32 valid, u64 (&elem)[MSD_IPT], u32 (&dig)[MSD_IPT],
                                              int shift2)
{
    const u32 tid = threadIdx.x;
    if (FROM_TEXT) {
        // 16 consecutive suffixes per thread; element = [key bits below the top 10 | index]
        const u32 i0 = base + tid * MSD_IPT;
        u64 key[MSD_IPT] = {};
        if (i0 < a.n) text_keys16(a.codes, i0, a.code_bits, a.key_chars, a.plus_one, a.key_drop, a.n, key);
        const int rest_bits = a.key_bits - MSD_D;
        const u64 rest_mask = (1ull << rest_bits) - 1ull;
#pragma unroll
        for (int r = 0; r < MSD_IPT; ++r) {
            dig[r] = (u32)(key[r] >> rest_bits) & (MSD_BINS - 1u);
            elem[r] = ((key[r] & rest_mask) << a.idx_bits) | (u64)(i0 + r);
        }
    } else {
#pragma unroll
        for (int r = 0; r < MSD_IPT; ++r) {
            const u32 p = r * MSD_BLOCK + tid;
            elem[r] = p < valid ? a.in[base + p] : 0ull;
            dig[r] = (u32)(elem[r] >> shift2) & (MSD_BINS - 1u);
        }
    }
}

template <bool FROM_TEXT>
__device__ __forceinline__ bool msd_valid(u32 base, u32 valid, int r, u32 n)
{
    const u32 tid = threadIdx.x;
    (void)base;
    (void)n;
    return FROM_TEXT ? (tid * MSD_IPT + r < valid) : (r * MSD_BLOCK + tid < valid);
}

template <bool FROM_TEXT>
__global__ __launch_bounds__(MSD_BLOCK) void msd_hist_kernel(MsdArgs a)
{
    __shared__ u32 hist[MSD_BINS];
    const u32 tid = threadIdx.x, r = blockIdx.x;
    u32 e0, e1;
    if (FROM_TEXT) {
        if (r >= a.num_ranges1) return;
        e0 = r * a.tiles_per_range1 * MSD_TILE;
        const u64 end = (u64)(r + 1) * a.tiles_per_range1 * MSD_TILE;
        e1 = end < a.n ? (u32)end : a.n;
    } else {
        if (r >= a.counters[0]) return;
        e0 = a.ranges2[r].start;
        e1 = a.ranges2[r].end;
    }
    for (u32 i = tid; i < MSD_BINS; i += MSD_BLOCK) hist[i] = 0;
    __syncthreads();
    const int shift2 = a.idx_bits + a.key_bits - 2 * MSD_D;
    for (u32 base = e0; base < e1; base += MSD_TILE) {
        const u32 valid = min(MSD_TILE, e1 - base);
        u64 elem[MSD_IPT];
        u32 dig[MSD_IPT];
        msd_load_tile<FROM_TEXT>(a, base, valid, elem, dig, shift2);
#pragma unroll
        for (int k = 0; k < MSD_IPT; ++k)
            if (msd_valid<FROM_TEXT>(base, valid, k, a.n)) atomicAdd(&hist[dig[k]], 1u);
    }
    __syncthreads();
    for (u32 i = tid; i < MSD_BINS; i += MSD_BLOCK) a.T[(size_t)r * MSD_BINS + i] = hist[i];
}

// One workgroup per segment (G1: the whole input; G2: one G1 bucket), thread = digit:
// T[r][d] := global offset of (range r, digit d); joint[seg * 1024 + d] := start of bucket (seg, d).
__global__ __launch_bounds__(MSD_BINS) void msd_offsets_kernel(u32 *T, const u32 *seg_first, const u32 *seg_start, u32 *joint,
                                                                u32 num_ranges_fixed, u32 n, u32 nseg)
{
    __shared__ u32 scr[MSD_BINS / kWave + 1];
    const u32 seg = blockIdx.x, d = threadIdx.x;
    const u32 r0 = seg_first ? seg_first[seg] : 0u;
    const u32 r1 = seg_first ? seg_first[seg + 1] : num_ranges_fixed;
    const u32 base = seg_start ? seg_start[seg] : 0u;
    u32 run = 0;
    for (u32 r = r0; r < r1; ++r) {
        const u32 c = T[(size_t)r * MSD_BINS + d];
        T[(size_t)r * MSD_BINS + d] = run;
        run += c;
    }
    const u32 binbase = base + block_excl_sum<MSD_BINS / kWave>(run, scr, nullptr);
    joint[(size_t)seg * MSD_BINS + d] = binbase;
    if (seg == nseg - 1 && d == 0) joint[(size_t)nseg * MSD_BINS] = n;
    for (u32 r = r0; r < r1; ++r) T[(size_t)r * MSD_BINS + d] += binbase;
}

// G2 ranges: pieces of <= MSD_G2_RANGE elements of one G1 bucket.  One workgroup, thread = bucket.
__global__ __launch_bounds__(MSD_BINS) void msd_ranges_kernel(const u32 *J1, MsdRange *ranges, u32 *seg_first, u32 *counters)
{
    __shared__ u32 scr[MSD_BINS / kWave + 1];
    const u32 seg = threadIdx.x;
    const u32 s = J1[seg], e = J1[seg + 1];
    const u32 nr = (e - s + MSD_G2_RANGE - 1) / MSD_G2_RANGE;
    u32 total = 0;
    const u32 first = block_excl_sum<MSD_BINS / kWave>(nr, scr, &total);
    seg_first[seg] = first;
    if (seg == MSD_BINS - 1) seg_first[MSD_BINS] = total;
    if (seg == 0) counters[0] = total;
    for (u32 k = 0; k < nr; ++k) {
        const u32 rs = s + k * MSD_G2_RANGE;
        ranges[first + k] = MsdRange{seg, rs, min(e, rs + MSD_G2_RANGE)};
    }
}

template <bool FROM_TEXT>
__global__ __launch_bounds__(MSD_BLOCK) void msd_scatter_kernel(MsdArgs a)
{
    __shared__ __attribute__((aligned(16))) u64 exch[MSD_TILE];
    __shared__ u16 exd[FROM_TEXT ? MSD_TILE : 1];      // G1 strips the digit from the element: kept beside it
    __shared__ u32 hist[MSD_BINS], s_delta[MSD_BINS], s_off[MSD_BINS];
    __shared__ u16 s_start[MSD_BINS];                 // 16-bit (a tile has 8192 slots): the G2 kernel then fits twice into a CU's LDS
    __shared__ u32 scr[MSD_WAVES + 1];
    const u32 tid = threadIdx.x, r = blockIdx.x;
    u32 e0, e1;
    if (FROM_TEXT) {
        if (r >= a.num_ranges1) return;
        e0 = r * a.tiles_per_range1 * MSD_TILE;
        const u64 end = (u64)(r + 1) * a.tiles_per_range1 * MSD_TILE;
        e1 = end < a.n ? (u32)end : a.n;
    } else {
        if (r >= a.counters[0]) return;
        e0 = a.ranges2[r].start;
        e1 = a.ranges2[r].end;
    }
    for (u32 i = tid; i < MSD_BINS; i += MSD_BLOCK) {
        hist[i] = 0;
        s_off[i] = a.T[(size_t)r * MSD_BINS + i];
    }
    __syncthreads();
    const int shift2 = a.idx_bits + a.key_bits - 2 * MSD_D;
    for (u32 base = e0; base < e1; base += MSD_TILE) {
        const u32 valid = min(MSD_TILE, e1 - base);
        u64 elem[MSD_IPT];
        u32 dig[MSD_IPT];
        u32 rank[MSD_IPT];
        msd_load_tile<FROM_TEXT>(a, base, valid, elem, dig, shift2);
#pragma unroll
        for (int k = 0; k < MSD_IPT; ++k)
            rank[k] = msd_valid<FROM_TEXT>(base, valid, k, a.n) ? atomicAdd(&hist[dig[k]], 1u) : 0u;
        __syncthreads();                                    // (A) counts complete; previous tile fully written out
        {
            // exclusive scan over the 1024 bins, two adjacent bins per thread
            const u32 c0 = hist[2 * tid], c1 = hist[2 * tid + 1];
            const u32 ex = block_excl_sum<MSD_WAVES>(c0 + c1, scr, nullptr);
            s_start[2 * tid] = (u16)ex;
            s_start[2 * tid + 1] = (u16)(ex + c0);
            const u32 o0 = s_off[2 * tid], o1 = s_off[2 * tid + 1];
            s_delta[2 * tid] = o0 - ex;
            s_delta[2 * tid + 1] = o1 - (ex + c0);
            s_off[2 * tid] = o0 + c0;
            s_off[2 * tid + 1] = o1 + c1;
            hist[2 * tid] = 0;
            hist[2 * tid + 1] = 0;
        }
        __syncthreads();                                    // (B) bin starts published
#pragma unroll
        for (int k = 0; k < MSD_IPT; ++k) {
            if (msd_valid<FROM_TEXT>(base, valid, k, a.n)) {
                const u32 lp = (u32)s_start[dig[k]] + rank[k];
                exch[lp] = elem[k];
                if (FROM_TEXT) exd[lp] = (u16)dig[k];
            }
        }
        __syncthreads();                                    // (C) tile in bin order
#pragma unroll
        for (int k = 0; k < MSD_IPT; ++k) {
            const u32 p = k * MSD_BLOCK + tid;
            if (p < valid) {
                const u64 e = exch[p];
                const u32 d = FROM_TEXT ? (u32)exd[p] : ((u32)(e >> shift2) & (MSD_BINS - 1u));
                a.out[s_delta[d] + p] = e;
            }
        }
    }
}

// ---- tile plan ---------------------------------------------------------------------------------

struct InNonEmpty {
    const u32 *J;
    __device__ u64 operator()(u64 j) const { return J[j + 1] > J[j] ? 1u : 0u; }
};

// compact the starts of the non-empty joint buckets; largest bucket
__global__ __launch_bounds__(256) void msd_compact_kernel(const u32 *J, u32 nb, const u64 *rank, u32 *cstart, u32 *counters)
{
    u32 mx = 0;
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < nb; j += gridDim.x * blockDim.x) {
        const u32 s = J[j], e = J[j + 1];
        if (e > s) {
            cstart[rank[j]] = s;
            mx = max(mx, e - s);
        }
    }
    mx = wave_incl_max(mx);
    if (lane_id() == kWave - 1 && mx) atomicMax(&counters[1], mx);
}

struct InTileHead {
    const u32 *cstart;
    __device__ u64 operator()(u64 k) const
    {
        if (k == 0 || (k % MSD_TILE_BUCKETS) == 0) return 1u;
        return (cstart[k] / MSD_CAPH != cstart[k - 1] / MSD_CAPH) ? 1u : 0u;
    }
};

__global__ __launch_bounds__(256) void msd_tiles_kernel(const u32 *cstart, u32 ne, const u64 *rank, const u64 *total,
                                                          u32 *tile_first)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) tile_first[*total] = ne;      // sentinel behind the last tile
    for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < ne; k += gridDim.x * blockDim.x) {
        const bool head = k == 0 || (k % MSD_TILE_BUCKETS) == 0 || cstart[k] / MSD_CAPH != cstart[k - 1] / MSD_CAPH;
        if (head) tile_first[rank[k]] = k;
    }
}

// ---- local sort --------------------------------------------------------------------------------

// One workgroup per tile: <= 8192 elements of <= 1024 consecutive joint buckets.  Sort key in LDS:
// [ bucket number inside the tile | remaining key bits | suffix index ] -- stable 8-bit LSD passes
// over the bucket and key bits (the index bits ride along), wave-ballot ranking as in radix_sort.hip.
// This is the general (slower) form: it takes whatever the fast kernel below hands back.
__global__ __launch_bounds__(MSD_BLOCK) void msd_local_sort_kernel(const u64 *in, const u32 *cstart, const u32 *tile_first,
                                                                     u32 ne, u32 n, int rem_bits, int idx_bits, u32 *sa_out,
                                                                     const u32 *tile_list)
{
    __shared__ __attribute__((aligned(16))) u64 exch[MSD_TILE];
    __shared__ u32 wave_hist[MSD_WAVES][256];
    __shared__ u32 s_bstart[MSD_TILE_BUCKETS + 1];
    __shared__ u32 scr[MSD_WAVES + 1];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u32 t = tile_list ? tile_list[blockIdx.x] : blockIdx.x;
    const u32 k0 = tile_first[t], k1 = tile_first[t + 1];
    const u32 nb = k1 - k0;
    const u32 e0 = cstart[k0];
    const u32 e1 = k1 < ne ? cstart[k1] : n;
    const u32 count = e1 - e0;
    for (u32 i = tid; i <= nb; i += MSD_BLOCK) s_bstart[i] = (k0 + i < ne) ? cstart[k0 + i] : n;
    for (u32 i = tid; i < MSD_WAVES * 256; i += MSD_BLOCK) (&wave_hist[0][0])[i] = 0;
    __syncthreads();
    int seg_bits = 0;
    while ((1u << seg_bits) < nb) ++seg_bits;
    const u64 low_mask = (1ull << (rem_bits + idx_bits)) - 1ull;
    // element r of this thread sits at tile position wave * 1024 + r * 64 + lane (order = position)
    u64 key[MSD_IPT];
#pragma unroll
    for (int r = 0; r < MSD_IPT; ++r) {
        const u32 p = wave * (kWave * MSD_IPT) + r * kWave + lane;
        if (p < count) {
            const u64 e = in[e0 + p];
            // bucket of position e0 + p: last start <= it
            u32 lo = 0, hi = nb;
            const u32 at = e0 + p;
            while (hi - lo > 1) {
                const u32 mid = (lo + hi) >> 1;
                if (s_bstart[mid] <= at) lo = mid; else hi = mid;
            }
            key[r] = ((u64)lo << (rem_bits + idx_bits)) | (e & low_mask);
        } else {
            key[r] = ~0ull;                                   // padding
        }
    }
    const int sort_bits = rem_bits + seg_bits;
    for (int shift = idx_bits; shift < idx_bits + sort_bits; shift += 8) {
        const int left = idx_bits + sort_bits - shift;
        const u32 dmask = left >= 8 ? 0xffu : ((1u << left) - 1u);
        u32 rank[MSD_IPT], prev[MSD_IPT];
#pragma unroll
        for (int r = 0; r < MSD_IPT; ++r) {
            // padding (positions >= count) carries the largest digit of every pass: a stable sort keeps it last
            const bool pad = wave * (kWave * MSD_IPT) + r * kWave + lane >= count;
            const u32 d = pad ? dmask : ((u32)(key[r] >> shift) & dmask);
            const u64 peers = match_digit8(d, ~0ull);
            const u32 below = mbcnt(peers);
            prev[r] = 0;
            if (below == 0) prev[r] = atomicAdd(&wave_hist[wave][d], (u32)__popcll(peers));
            rank[r] = below | ((u32)__builtin_ctzll(peers) << 16);
        }
#pragma unroll
        for (int r = 0; r < MSD_IPT; ++r) rank[r] = __shfl(prev[r], (int)(rank[r] >> 16)) + (rank[r] & 0xffffu);
        __syncthreads();
        {
            // digit d is owned by thread d (the first four waves); exclusive prefix over the 256 digits
            u32 c[MSD_WAVES];
            u32 total = 0;
            if (tid < 256) {
#pragma unroll
                for (int w = 0; w < MSD_WAVES; ++w) {
                    c[w] = wave_hist[w][tid];
                    total += c[w];
                }
            }
            const u32 incl = wave_incl_sum(total);
            if (lane == kWave - 1) scr[wave] = incl;
            __syncthreads();
            if (tid < 256) {
                u32 run = incl - total;
                for (u32 w = 0; w < wave; ++w) run += scr[w];
#pragma unroll
                for (int w = 0; w < MSD_WAVES; ++w) {
                    wave_hist[w][tid] = run;
                    run += c[w];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MSD_IPT; ++r) {
            const bool pad = wave * (kWave * MSD_IPT) + r * kWave + lane >= count;
            const u32 d = pad ? dmask : ((u32)(key[r] >> shift) & dmask);
            exch[wave_hist[wave][d] + rank[r]] = key[r];
        }
        __syncthreads();
        for (u32 i = tid; i < MSD_WAVES * 256; i += MSD_BLOCK) (&wave_hist[0][0])[i] = 0;
#pragma unroll
        for (int r = 0; r < MSD_IPT; ++r) key[r] = exch[wave * (kWave * MSD_IPT) + r * kWave + lane];
        __syncthreads();
    }
    if (sort_bits <= 0) {            // nothing to sort (one bucket, no key bits left): still goes through LDS for the flags
#pragma unroll
        for (int r = 0; r < MSD_IPT; ++r) exch[wave * (kWave * MSD_IPT) + r * kWave + lane] = key[r];
        __syncthreads();
    }
    // sequential output; bit 31 = same (bucket, key) as my predecessor
    const u32 imask = (u32)((1ull << idx_bits) - 1ull);
#pragma unroll
    for (int k = 0; k < MSD_IPT; ++k) {
        const u32 p = k * MSD_BLOCK + tid;
        if (p < count) {
            const u64 e = exch[p];
            const bool tie = p > 0 && (exch[p - 1] >> idx_bits) == (e >> idx_bits);
            sa_out[e0 + p] = ((u32)e & imask) | (tie ? 0x80000000u : 0u);
        }
    }
}


// The fast form.  One counting pass on the top 11 bits of (bucket, key) with returning LDS atomics (no
// order to preserve: the whole element, index included, is the sort key, so the result is a total order
// anyway) leaves bins of a handful of elements; every element then finds its place inside its bin by
// counting the smaller ones -- neighbouring lanes sit in the same bin, so those LDS reads are broadcasts.
// A tile with a bin above LS_KMAX elements (many equal or nearly equal keys) is handed to the general
// kernel instead.
constexpr int LS_BIN_BITS = 11;
constexpr u32 LS_BINS = 1u << LS_BIN_BITS;
constexpr u32 LS_KMAX = 64;

__global__ __launch_bounds__(MSD_BLOCK) void msd_local_fast_kernel(const u64 *in, const u32 *cstart, const u32 *tile_first,
                                                                     u32 ne, u32 n, int rem_bits, int idx_bits, u32 *sa_out,
                                                                     u32 *fail_list, u32 *fail_count)
{
    __shared__ __attribute__((aligned(16))) u64 exch[MSD_TILE];
    __shared__ u32 hist[LS_BINS + 4];
    __shared__ u32 s_bstart[MSD_TILE_BUCKETS + 1];
    __shared__ u32 scr[MSD_WAVES + 1];
    __shared__ u32 s_fail;
    const u32 tid = threadIdx.x;
    const u32 t = blockIdx.x;
    const u32 k0 = tile_first[t], k1 = tile_first[t + 1];
    const u32 nb = k1 - k0;
    const u32 e0 = cstart[k0];
    const u32 e1 = k1 < ne ? cstart[k1] : n;
    const u32 count = e1 - e0;
    for (u32 i = tid; i <= nb; i += MSD_BLOCK) s_bstart[i] = (k0 + i < ne) ? cstart[k0 + i] : n;
    for (u32 i = tid; i < LS_BINS + 4; i += MSD_BLOCK) hist[i] = 0;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    int seg_bits = 0;
    while ((1u << seg_bits) < nb) ++seg_bits;
    const int sort_bits = rem_bits + seg_bits;
    const int bin_shift = idx_bits + (sort_bits > LS_BIN_BITS ? sort_bits - LS_BIN_BITS : 0);
    const u64 low_mask = (1ull << (rem_bits + idx_bits)) - 1ull;
    const u32 rows = (count + MSD_BLOCK - 1) / MSD_BLOCK;          // uniform over the workgroup
    u64 e[MSD_IPT];
    u32 rk[MSD_IPT];
#pragma unroll
    for (int r = 0; r < MSD_IPT; ++r) {
        if ((u32)r < rows) {
            const u32 p = r * MSD_BLOCK + tid;
            e[r] = ~0ull;
            rk[r] = 0;
            if (p < count) {
                const u64 x = in[e0 + p];
                u32 lo = 0, hi = nb;
                const u32 at = e0 + p;
                while (hi - lo > 1) {
                    const u32 mid = (lo + hi) >> 1;
                    if (s_bstart[mid] <= at) lo = mid; else hi = mid;
                }
                e[r] = ((u64)lo << (rem_bits + idx_bits)) | (x & low_mask);
                rk[r] = atomicAdd(&hist[(u32)(e[r] >> bin_shift)], 1u);
            }
        }
    }
    __syncthreads();
    {
        // exclusive scan over the 2048 bins in place, four adjacent bins per thread
        const u32 c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
        const u32 ex = block_excl_sum<MSD_WAVES>(c0 + c1 + c2 + c3, scr, nullptr);
        hist[4 * tid] = ex;
        hist[4 * tid + 1] = ex + c0;
        hist[4 * tid + 2] = ex + c0 + c1;
        hist[4 * tid + 3] = ex + c0 + c1 + c2;
        if (tid == 0) hist[LS_BINS] = count;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MSD_IPT; ++r) {
        if ((u32)r < rows) {
            const u32 p = r * MSD_BLOCK + tid;
            if (p < count) exch[hist[(u32)(e[r] >> bin_shift)] + rk[r]] = e[r];
        }
    }
    __syncthreads();
    // place inside the bin = number of smaller elements there (thread <-> position: neighbours share the bin)
#pragma unroll
    for (int r = 0; r < MSD_IPT; ++r) {
        if ((u32)r < rows) {
            const u32 p = r * MSD_BLOCK + tid;
            if (p < count) {
                const u64 x = exch[p];
                const u32 bin = (u32)(x >> bin_shift);
                const u32 s0 = hist[bin], s1 = hist[bin + 1];
                u32 smaller = 0;
                if (s1 - s0 > LS_KMAX) {
                    s_fail = 1;
                } else {
                    for (u32 q = s0; q < s1; ++q) smaller += exch[q] < x ? 1u : 0u;
                }
                e[r] = x;
                rk[r] = s0 + smaller;
            }
        }
    }
    __syncthreads();
    if (s_fail) {
        if (tid == 0) fail_list[atomicAdd(fail_count, 1u)] = t;
        return;
    }
#pragma unroll
    for (int r = 0; r < MSD_IPT; ++r) {
        if ((u32)r < rows) {
            const u32 p = r * MSD_BLOCK + tid;
            if (p < count) exch[rk[r]] = e[r];
        }
    }
    __syncthreads();
    const u32 imask = (u32)((1ull << idx_bits) - 1ull);
#pragma unroll
    for (int r = 0; r < MSD_IPT; ++r) {
        if ((u32)r < rows) {
            const u32 p = r * MSD_BLOCK + tid;
            if (p < count) {
                const u64 x = exch[p];
                const bool tie = p > 0 && (exch[p - 1] >> idx_bits) == (x >> idx_bits);
                sa_out[e0 + p] = ((u32)x & imask) | (tie ? 0x80000000u : 0u);
            }
        }
    }
}

// ---- host --------------------------------------------------------------------------------------

size_t msd_workspace_bytes(uint32_t n)
{
    const size_t max_ranges2 = (size_t)n / MSD_G2_RANGE + MSD_BINS + 8;
    const size_t nbk = (size_t)MSD_BINS * MSD_BINS;
    return max_ranges2 * MSD_BINS * 4                 // T
           + (MSD_BINS + 8) * 4                       // J1
           + (nbk + 8) * 4                            // J
           + max_ranges2 * sizeof(MsdRange) + (MSD_BINS + 8) * 4 + 64   // ranges, seg_first, counters
           + (nbk + 8) * 8                            // scan output (ranks)
           + (nbk + 8) * 4                            // compacted starts
           + ((size_t)n / MSD_CAPH + nbk / MSD_TILE_BUCKETS + 32) * 8   // tile_first, then the tiles left to the general kernel
           + (SC_MAX_BLOCKS + 8) * 8 + 4096;
}

int msd_max_key_bits(uint32_t n)
{
    int ib = 1;
    while ((1ull << ib) < (u64)n) ++ib;
    return 64 + MSD_D - ib;      // [K - 10 key bits | ib index bits] must fit 64 bits after G1
}

int msd_suffix_sort(DeviceCtx *ctx, const TextKeys *text, uint32_t n, int key_bits, uint64_t *A[2], uint32_t *sa_out,
                    void *work, uint32_t *h_small, bool profile, MsdStats *stats, bool *accepted)
{
    *accepted = false;
    hipStream_t s = ctx->stream;
    int ib = 1;
    while ((1ull << ib) < (u64)n) ++ib;
    if (key_bits < 2 * MSD_D + 1 || key_bits > 64 + MSD_D - ib || n < 2) {
        set_error("msd_suffix_sort: key of %d bits does not fit (n = %u)", key_bits, n);
        return PSS_EINVAL;
    }
    const size_t max_ranges2 = (size_t)n / MSD_G2_RANGE + MSD_BINS + 8;
    const size_t nbk = (size_t)MSD_BINS * MSD_BINS;
    u8 *w = static_cast<u8 *>(work);
    size_t o = 0;
    auto carve = [&](size_t bytes) { u8 *p = w + o; o = round_up(o + bytes, 256); return p; };
    u32 *T = reinterpret_cast<u32 *>(carve(max_ranges2 * MSD_BINS * 4));
    u32 *J1 = reinterpret_cast<u32 *>(carve((MSD_BINS + 8) * 4));
    u32 *J = reinterpret_cast<u32 *>(carve((nbk + 8) * 4));
    MsdRange *ranges2 = reinterpret_cast<MsdRange *>(carve(max_ranges2 * sizeof(MsdRange)));
    u32 *seg_first = reinterpret_cast<u32 *>(carve((MSD_BINS + 8) * 4));
    u32 *counters = reinterpret_cast<u32 *>(carve(64));
    u64 *ranks = reinterpret_cast<u64 *>(carve((nbk + 8) * 8));
    u32 *cstart = reinterpret_cast<u32 *>(carve((nbk + 8) * 4));
    const size_t max_tiles = (size_t)n / MSD_CAPH + nbk / MSD_TILE_BUCKETS + 8;
    u32 *tile_first = reinterpret_cast<u32 *>(carve((max_tiles + 16) * 8));
    u64 *partial = reinterpret_cast<u64 *>(carve((SC_MAX_BLOCKS + 8) * 8));
    u64 *d_total = partial + SC_MAX_BLOCKS;

    MsdArgs a;
    memset(&a, 0, sizeof a);
    a.codes = text->codes;
    a.code_bits = text->code_bits;
    a.key_chars = text->key_chars;
    a.plus_one = text->plus_one;
    a.key_drop = text->drop;
    a.n = n;
    a.key_bits = key_bits;
    a.idx_bits = ib;
    const u32 num_tiles = (u32)(((u64)n + MSD_TILE - 1) / MSD_TILE);
    a.tiles_per_range1 = (num_tiles + MSD_G1_RANGES - 1) / MSD_G1_RANGES;
    a.num_ranges1 = (num_tiles + a.tiles_per_range1 - 1) / a.tiles_per_range1;
    a.T = T;
    a.J1 = J1;
    a.J = J;
    a.ranges2 = ranges2;
    a.seg_first = seg_first;
    a.counters = counters;

    hipEvent_t ev[8] = {};
    int nev = 0;
    struct EvGuard {
        hipEvent_t *e;
        int *n;
        ~EvGuard()
        {
            for (int i = 0; i < *n; ++i) (void)hipEventDestroy(e[i]);
        }
    } guard{ev, &nev};
    auto mark = [&]() -> int {
        if (profile && nev < 8) {
            PSS_HIP(hipEventCreate(&ev[nev]));
            PSS_HIP(hipEventRecord(ev[nev], s));
            ++nev;
        }
        return PSS_OK;
    };

    PSS_HIP(hipMemsetAsync(counters, 0, 64, s));
    // ---- G1: text -> A[0] by the top 10 key bits ----
    a.out = A[0];
    hipLaunchKernelGGL(msd_hist_kernel<true>, dim3(a.num_ranges1), dim3(MSD_BLOCK), 0, s, a);
    hipLaunchKernelGGL(msd_offsets_kernel, dim3(1), dim3(MSD_BINS), 0, s, T, (const u32 *)nullptr, (const u32 *)nullptr, J1,
                       a.num_ranges1, n, 1u);
    PSS_TRY(mark());
    hipLaunchKernelGGL(msd_scatter_kernel<true>, dim3(a.num_ranges1), dim3(MSD_BLOCK), 0, s, a);
    PSS_TRY(mark());
    // ---- G2: A[0] -> A[1], every G1 bucket by the next 10 bits ----
    hipLaunchKernelGGL(msd_ranges_kernel, dim3(1), dim3(MSD_BINS), 0, s, J1, ranges2, seg_first, counters);
    a.in = A[0];
    a.out = A[1];
    hipLaunchKernelGGL(msd_hist_kernel<false>, dim3((u32)max_ranges2), dim3(MSD_BLOCK), 0, s, a);
    hipLaunchKernelGGL(msd_offsets_kernel, dim3(MSD_BINS), dim3(MSD_BINS), 0, s, T, (const u32 *)seg_first, (const u32 *)J1, J,
                       0u, n, MSD_BINS);
    // ---- plan: non-empty joint buckets, largest bucket, tiles ----
    PSS_TRY(device_excl_scan(ctx, InNonEmpty{J}, nbk, partial, d_total, ranks));
    hipLaunchKernelGGL(msd_compact_kernel, dim3(1024), dim3(256), 0, s, J, (u32)nbk, ranks, cstart, counters);
    PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipMemcpyAsync(h_small + 2, counters, 16, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipStreamSynchronize(s));
    const u32 ne = h_small[0];
    const u32 maxb = h_small[3];
    if (stats) {
        stats->buckets = ne;
        stats->max_bucket = maxb;
    }
    if (maxb > MSD_MAX_BUCKET) return PSS_OK;      // not this text: the caller takes the LSD path
    PSS_TRY(mark());
    hipLaunchKernelGGL(msd_scatter_kernel<false>, dim3((u32)max_ranges2), dim3(MSD_BLOCK), 0, s, a);
    PSS_TRY(mark());
    PSS_TRY(device_excl_scan(ctx, InTileHead{cstart}, ne, partial, d_total, ranks));
    hipLaunchKernelGGL(msd_tiles_kernel, dim3(1024), dim3(256), 0, s, cstart, ne, ranks, (const u64 *)d_total, tile_first);
    PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipStreamSynchronize(s));
    const u32 nt = h_small[0];
    PSS_TRY(mark());
    // counters[4] = tiles the fast kernel declined; their numbers go behind the tile table
    u32 *fail_list = tile_first + nt + 8;
    if (getenv("PSS_MSD_SLOW_LOCAL")) {
        hipLaunchKernelGGL(msd_local_sort_kernel, dim3(nt), dim3(MSD_BLOCK), 0, s, A[1], cstart, tile_first, ne, n,
                           key_bits - 2 * MSD_D, ib, sa_out, (const u32 *)nullptr);
    } else {
        hipLaunchKernelGGL(msd_local_fast_kernel, dim3(nt), dim3(MSD_BLOCK), 0, s, A[1], cstart, tile_first, ne, n,
                           key_bits - 2 * MSD_D, ib, sa_out, fail_list, counters + 4);
        PSS_HIP(hipMemcpyAsync(h_small, counters + 4, 4, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
        const u32 nfail = h_small[0];
        if (stats) stats->slow_tiles = nfail;
        if (nfail)
            hipLaunchKernelGGL(msd_local_sort_kernel, dim3(nfail), dim3(MSD_BLOCK), 0, s, A[1], cstart, tile_first, ne, n,
                               key_bits - 2 * MSD_D, ib, sa_out, (const u32 *)fail_list);
    }
    PSS_TRY(mark());
    PSS_HIP(hipGetLastError());
    if (stats) stats->tiles = nt;
    if (profile && nev >= 6) {
        PSS_HIP(hipStreamSynchronize(s));
        float ms = 0.f;
        PSS_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
        if (stats) stats->ms_g1 = ms;
        PSS_HIP(hipEventElapsedTime(&ms, ev[2], ev[3]));
        if (stats) stats->ms_g2 = ms;
        PSS_HIP(hipEventElapsedTime(&ms, ev[4], ev[5]));
        if (stats) stats->ms_local = ms;
    }
    *accepted = true;
    return PSS_OK;
}

}  // namespace pss
