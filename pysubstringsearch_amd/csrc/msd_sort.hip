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
//                    consecutive buckets are packed into tiles of < 8192 elements, one
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
// Round 6 -- the two digits in LSD ORDER, the second pass in ONE sweep (decoupled look-back).  The order above makes
// the second pass wait for a histogram of its own: where bucket (b, d) starts inside G1 bucket b is only known once all
// of b has been counted -- 8 n bytes read a second time (msd_hist_kernel<false>, 0.85 ms at n = 2^29).  Taken the other
// way round the passes need the two 1024-bin histograms of the TEXT only, which one pass over its n bytes gives:
//
//   H   text          counts of the second digit d per range (the first pass's offsets) and of the first digit b
//   P1  text -> A1    partition by d; the element keeps b: [ b | remaining key bits | suffix index ]
//   P2  A1 -> A2      partition by b, tiles in order: a tile lies inside one d-region, so inside every b-bucket the tiles
//                     arrive in d order and the joint buckets (b, d) come out contiguous whatever the order inside a tile
//                     (still one returning LDS atomic per element).  Where (tile, b) goes = start of b + what the tiles
//                     before it hold of b: every tile counts its own 16384 elements (it ranks them anyway), publishes
//                     the 1024 counts and adds up what its predecessors published, back to the nearest tile that has
//                     published its running total -- no second read, no table.  Tiles are handed out by an atomic ticket,
//                     so every predecessor of a waiting tile is running.  The last tile of d-region d leaves the ends of
//                     the buckets (b, d): the table of joint bucket starts falls out of the pass.
//
// The joint table only exists after P2, so P2 cannot tag an element with its bucket's number among the non-empty ones.
// The d-regions are renumbered first (msd_tiles2_kernel: the non-empty ones 0 .. D - 1 in digit order -- a digit value
// that never occurs leaves no hole), P2 tags with that number mod 64 (a constant of the tile), the plan never lets a
// local-sort tile cross a multiple of 64 joint buckets, and the local sort takes "tag - tag of the tile's first bucket"
// as the bucket's number inside its tile: one subtraction.
// Measured in isolation before it was built: tests/tools/lookback_micro.hip, profiles/r06_lookback_micro.txt.
//
// HBM-bound integer work: no MFMA anywhere by design.
#include "msd_sort.h"

#include <algorithm>
#include <vector>

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
constexpr u32 MSD_WIN = 6144;                        // buckets whose start falls into one window of this size share a tile ...
#ifndef PSS_LS_WINDOW
#define PSS_LS_WINDOW 8                              // (members of its bin every element of the local sort reads unconditionally)
#endif
constexpr u32 MSD_TILE_CAP = 8184 - PSS_LS_WINDOW;   // ... unless that is more than a tile holds: then the window's last bucket goes alone
                                                     // (the window's worth of slots behind a tile's last element hold the ranking's
                                                     //  sentinels, the eight after those the fast kernel's scalars)
constexpr u32 MSD_MAX_BUCKET = 4088;                 // a bucket must fit a tile on its own (the last eight slots of the LDS tile
                                                     // carry the fast kernel's scalars: MSD_TILE_CAP)
constexpr int MSD_TAG_BITS = MSD_D + 1;              // an element entering the local sort carries the low 11 bits of its joint bucket number ...
constexpr u32 MSD_TAG_SPAN = 1u << MSD_TAG_BITS;     // ... so a tile never crosses a multiple of 2048 buckets: inside it the tags only grow
constexpr int MSD_RAW_TAG_BITS = 6;                  // LSD order: the tag is the joint bucket number mod 64 (= d mod 64) ...
constexpr u32 MSD_RAW_TAG_SPAN = 1u << MSD_RAW_TAG_BITS;   // ... and a tile stays inside one aligned block of 64 joint buckets
constexpr u32 MSD_G1_RANGES = 1024;
constexpr u32 MSD_G2_RANGE = 16 * MSD_TILE;          // elements per G2 range (a piece of one G1 bucket)

struct MsdRange {
    u32 seg, start, end;
};

// A tile of the look-back pass: a piece of one d-region (count 0: an empty region, which still owes its row of the
// joint table), the region's digit, bit 31 = the region's last tile
struct MsdTile2 {
    u32 start, count, d_last;
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
    const u64 *dense;    // G2 scatter: [2^20] number of every joint bucket among the non-empty ones
    const u64 *in;
    u64 *out;
    // LSD order (round 6): the first pass partitions by the SECOND digit, the second pass is the look-back kernel
    int lsd;
    u32 *T2;             // [1024][ranges] counts of the first digit per range (only their totals are used)
    const u32 *Jb;       // [1025] starts of the first digit's buckets
    const struct MsdTile2 *tiles2;
    u32 *status;         // [tiles2][1024] look-back words
    // MsdFront: the first histogram pass recodes the raw text on its way through
    const u8 *raw;
    const u8 *lut;
    u8 *codes_out;
    u32 *bad;
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
    __shared__ u32 histb[FROM_TEXT ? MSD_BINS : 1];      // LSD order: the first digit, counted in the same pass
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
        if (FROM_TEXT) histb[i] = 0;
    }
    __syncthreads();
    const int shift2 = a.idx_bits + a.key_bits - 2 * MSD_D;
    if (FROM_TEXT) {
        // the digit is the top 10 bits of the key: the first m = ceil(10 / b) symbols are all it takes -- a 32-bit
        // sliding window instead of the 64-bit keys of text_keys16 (0.41 -> 0.16 ms at n = 2^29).  LSD order: the top
        // 20 bits, m = ceil(20 / b) symbols (<= 24 bits), the pass's own digit is the LOWER ten of them.
        const int b = a.code_bits;
        const int top = a.lsd ? 2 * MSD_D : MSD_D;
        const int m = (top + b - 1) / b;
        const u32 wmask = (m * b >= 32) ? ~0u : ((1u << (m * b)) - 1u);
        const int down = m * b - top;
        // (the next tile's 32 bytes are loaded before this tile's atomics: two trips to HBM in flight per thread)
        uint4 nlo = make_uint4(0, 0, 0, 0), nhi = nlo;
        if (e0 + tid * MSD_IPT < e1) {
            const uint4 *p = reinterpret_cast<const uint4 *>(a.codes + e0 + tid * MSD_IPT);
            nlo = p[0];
            nhi = p[1];
        }
        for (u32 base = e0; base < e1; base += MSD_TILE) {
            const u32 i0 = base + tid * MSD_IPT;
            const uint4 lo = nlo, hi = nhi;
            if ((u64)i0 + MSD_TILE < e1) {
                const uint4 *p = reinterpret_cast<const uint4 *>(a.codes + i0 + MSD_TILE);
                nlo = p[0];
                nhi = p[1];
            }
            if (i0 >= e1) continue;
            const u64 q[4] = {(u64)lo.x | ((u64)lo.y << 32), (u64)lo.z | ((u64)lo.w << 32),
                              (u64)hi.x | ((u64)hi.y << 32), (u64)hi.z | ((u64)hi.w << 32)};
            auto sym = [&](u32 j) -> u32 { return (u32)(q[j >> 3] >> ((j & 7u) * 8u)) & 0xffu; };   // codes are 0 past the text
            u32 win = 0;
            for (int t = 0; t < m; ++t) win = (win << b) | sym((u32)t);
#pragma unroll
            for (int r = 0; r < MSD_IPT; ++r) {
                if (r > 0) win = ((win << b) | sym((u32)(m - 1 + r))) & wmask;
                if (i0 + r < e1) {
                    const u32 tp = win >> down;
                    atomicAdd(&hist[tp & (MSD_BINS - 1u)], 1u);
                    if (a.lsd) atomicAdd(&histb[tp >> MSD_D], 1u);
                }
            }
        }
    } else {
        for (u32 base = e0; base < e1; base += MSD_TILE) {
            const u32 valid = min(MSD_TILE, e1 - base);
            u64 elem[MSD_IPT];
            u32 dig[MSD_IPT];
            msd_load_tile<FROM_TEXT>(a, base, valid, elem, dig, shift2);
#pragma unroll
            for (int k = 0; k < MSD_IPT; ++k)
                if (msd_valid<FROM_TEXT>(base, valid, k, a.n)) atomicAdd(&hist[dig[k]], 1u);
        }
    }
    __syncthreads();
    // G1: digit-major (one row per digit: its offsets kernel scans rows); G2: range-major (one row per range)
    for (u32 i = tid; i < MSD_BINS; i += MSD_BLOCK) {
        if (FROM_TEXT) {
            a.T[(size_t)i * a.num_ranges1 + r] = hist[i];
            if (a.lsd) a.T2[(size_t)i * a.num_ranges1 + r] = histb[i];
        } else {
            a.T[(size_t)r * MSD_BINS + i] = hist[i];
        }
    }
}

// The first histogram pass when the codes do not exist yet (MsdFront): the same windows, read from the raw text through
// the byte -> code table (in LDS), and the codes of the thread's own 16 bytes written out on the way -- the recode pass
// and the alphabet pass before it (0.4 ms and a host round trip at n = 2^29) fold into this one.  A byte without a code
// raises *bad; the caller then throws the sort away.
__global__ __launch_bounds__(MSD_BLOCK) void msd_hist_raw_kernel(MsdArgs a)
{
    __shared__ u32 hist[MSD_BINS];
    __shared__ u32 histb[MSD_BINS];      // LSD order: the first digit (see msd_hist_kernel)
    __shared__ u8 s_lut[256];
    const u32 tid = threadIdx.x, r = blockIdx.x;
    if (r >= a.num_ranges1) return;
    const u32 e0 = r * a.tiles_per_range1 * MSD_TILE;
    const u64 end = (u64)(r + 1) * a.tiles_per_range1 * MSD_TILE;
    const u32 e1 = end < a.n ? (u32)end : a.n;
    for (u32 i = tid; i < MSD_BINS; i += MSD_BLOCK) hist[i] = histb[i] = 0;
    if (tid < 256) s_lut[tid] = a.lut[tid];
    __syncthreads();
    const int b = a.code_bits;
    const int top = a.lsd ? 2 * MSD_D : MSD_D;
    const int m = (top + b - 1) / b;
    const u32 wmask = (m * b >= 32) ? ~0u : ((1u << (m * b)) - 1u);
    const int down = m * b - top;
    auto tr = [&](u32 w) -> u32 {
        return (u32)s_lut[w & 0xffu] | ((u32)s_lut[(w >> 8) & 0xffu] << 8) | ((u32)s_lut[(w >> 16) & 0xffu] << 16) |
               ((u32)s_lut[w >> 24] << 24);
    };
    // 32 raw bytes at i: two 16-byte loads when they lie inside the text, byte by byte (zero past the end) at its end
    auto load32 = [&](u32 i, uint4 &lo, uint4 &hi) {
        if ((u64)i + 32 <= a.n) {
            const uint4 *p = reinterpret_cast<const uint4 *>(a.raw + i);
            lo = p[0];
            hi = p[1];
        } else {
            u32 w[8] = {};
            for (u32 k = 0; k < 32 && i + k < a.n; ++k) w[k >> 2] |= (u32)a.raw[i + k] << (8u * (k & 3u));
            lo = make_uint4(w[0], w[1], w[2], w[3]);
            hi = make_uint4(w[4], w[5], w[6], w[7]);
        }
    };
    bool bad = false;
    uint4 nlo = make_uint4(0, 0, 0, 0), nhi = nlo;
    if (e0 + tid * MSD_IPT < e1) load32(e0 + tid * MSD_IPT, nlo, nhi);
    for (u32 base = e0; base < e1; base += MSD_TILE) {
        const u32 i0 = base + tid * MSD_IPT;
        uint4 lo = nlo, hi = nhi;
        if ((u64)i0 + MSD_TILE < e1) load32(i0 + MSD_TILE, nlo, nhi);
        if (i0 >= e1) continue;
        lo = make_uint4(tr(lo.x), tr(lo.y), tr(lo.z), tr(lo.w));
        // of the sixteen bytes after my own the windows read the first m - 1: one word of them when that is <= 4 symbols
        // (every alphabet of >= 5 bits per symbol) -- the table lookups are what this pass spends its time on
        if (m <= 5) hi = make_uint4(tr(hi.x), 0u, 0u, 0u);
        else hi = make_uint4(tr(hi.x), tr(hi.y), tr(hi.z), tr(hi.w));
        // my own 16 bytes: codes out, and every one of them inside the text must have got a code
        const u32 own = min(16u, a.n - i0);
        if (own == 16) {
            *reinterpret_cast<uint4 *>(a.codes_out + i0) = lo;
            const u32 z = ((lo.x - 0x01010101u) & ~lo.x) | ((lo.y - 0x01010101u) & ~lo.y) | ((lo.z - 0x01010101u) & ~lo.z) |
                          ((lo.w - 0x01010101u) & ~lo.w);
            bad |= (z & 0x80808080u) != 0;
        } else {
            const u32 w[4] = {lo.x, lo.y, lo.z, lo.w};
            for (u32 k = 0; k < own; ++k) {
                const u8 c = (u8)(w[k >> 2] >> (8u * (k & 3u)));
                a.codes_out[i0 + k] = c;
                bad |= c == 0;
            }
            // (a raw byte past the end reads as 0 and translates to lut[0]: the window must see 0 there)
            u32 ww[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            for (u32 k = own; k < 32; ++k) ww[k >> 2] &= ~(0xffu << (8u * (k & 3u)));
            lo = make_uint4(ww[0], ww[1], ww[2], ww[3]);
            hi = make_uint4(ww[4], ww[5], ww[6], ww[7]);
        }
        if (own == 16 && (u64)i0 + 32 > a.n) {            // the window's second half crosses the end of the text
            u32 ww[4] = {hi.x, hi.y, hi.z, hi.w};
            for (u32 k = 0; k < 16; ++k)
                if (i0 + 16 + k >= a.n) ww[k >> 2] &= ~(0xffu << (8u * (k & 3u)));
            hi = make_uint4(ww[0], ww[1], ww[2], ww[3]);
        }
        const u64 q[4] = {(u64)lo.x | ((u64)lo.y << 32), (u64)lo.z | ((u64)lo.w << 32),
                          (u64)hi.x | ((u64)hi.y << 32), (u64)hi.z | ((u64)hi.w << 32)};
        auto sym = [&](u32 j) -> u32 { return (u32)(q[j >> 3] >> ((j & 7u) * 8u)) & 0xffu; };
        u32 win = 0;
        for (int t = 0; t < m; ++t) win = (win << b) | sym((u32)t);
#pragma unroll
        for (int k = 0; k < MSD_IPT; ++k) {
            if (k > 0) win = ((win << b) | sym((u32)(m - 1 + k))) & wmask;
            if (i0 + k < e1) {
                const u32 tp = win >> down;
                atomicAdd(&hist[tp & (MSD_BINS - 1u)], 1u);
                if (a.lsd) atomicAdd(&histb[tp >> MSD_D], 1u);
            }
        }
    }
    if (__ballot(bad) && (tid & 63u) == 0) atomicOr(a.bad, 1u);
    __syncthreads();
    for (u32 i = tid; i < MSD_BINS; i += MSD_BLOCK) {
        a.T[(size_t)i * a.num_ranges1 + r] = hist[i];
        if (a.lsd) a.T2[(size_t)i * a.num_ranges1 + r] = histb[i];
    }
}

// G1 offsets, step 1: one workgroup per digit turns its row of per-range counts into exclusive prefixes
// and leaves the digit's total; step 2 (one workgroup) scans the totals into the bucket starts J1.
__global__ __launch_bounds__(256) void msd_offsets1_kernel(u32 *T, u32 num_ranges, u32 *totals)
{
    __shared__ u32 scr[256 / kWave + 1];
    const u32 d = blockIdx.x, tid = threadIdx.x;
    u32 *row = T + (size_t)d * num_ranges;
    u32 v[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 r = tid * 4 + j;
        v[j] = r < num_ranges ? row[r] : 0u;
        sum += v[j];
    }
    u32 total = 0;
    u32 run = block_excl_sum<256 / kWave>(sum, scr, &total);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32 r = tid * 4 + j;
        if (r < num_ranges) row[r] = run;
        run += v[j];
    }
    if (tid == 0) totals[d] = total;
}
__global__ __launch_bounds__(MSD_BINS) void msd_offsets1b_kernel(const u32 *totals, u32 *J1, u32 n)
{
    __shared__ u32 scr[MSD_BINS / kWave + 1];
    const u32 d = threadIdx.x;
    J1[d] = block_excl_sum<MSD_BINS / kWave>(totals[d], scr, nullptr);
    if (d == 0) J1[MSD_BINS] = n;
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
__global__ __launch_bounds__(MSD_BLOCK, 4) void msd_scatter_kernel(MsdArgs a)
{
    // (79.9 KB of LDS: both instantiations fit twice into a CU)
    __shared__ __attribute__((aligned(16))) u64 exch[MSD_TILE];
    __shared__ u32 hist[MSD_BINS], s_delta[MSD_BINS], s_off[MSD_BINS];
    __shared__ u16 s_start[MSD_BINS];                 // 16-bit (a tile has 8192 slots): the G2 kernel then fits twice into a CU's LDS
    __shared__ u32 scr[MSD_WAVES + 1];
    const u32 tid = threadIdx.x, r = blockIdx.x;
    u32 e0, e1;
    u32 tag0 = 0, tag1 = 0;      // G2: what the elements of bins 2 tid, 2 tid + 1 are tagged with
    if (FROM_TEXT) {
        if (r >= a.num_ranges1) return;
        e0 = r * a.tiles_per_range1 * MSD_TILE;
        const u64 end = (u64)(r + 1) * a.tiles_per_range1 * MSD_TILE;
        e1 = end < a.n ? (u32)end : a.n;
    } else {
        if (r >= a.counters[0]) return;
        e0 = a.ranges2[r].start;
        e1 = a.ranges2[r].end;
        // The element leaves G2 as [ tag | remaining key bits | index ], tag = low 11 bits of its joint bucket's number
        // among the non-empty buckets (known before this pass: the plan runs on G2's histogram): the local sort
        // reads the bucket of an element from the element (msd_local_*: no table of bucket starts, no search).
        // The tag of a bin rides in the high half of its counter word -- the returning atomic of the ranking brings it
        // along for nothing, the output loop reads it with the bin's offset.
        const u64 *dn = a.dense + (size_t)a.ranges2[r].seg * MSD_BINS;
        tag0 = ((u32)dn[2 * tid] & (MSD_TAG_SPAN - 1u)) << 16;
        tag1 = ((u32)dn[2 * tid + 1] & (MSD_TAG_SPAN - 1u)) << 16;
        hist[2 * tid] = tag0;
        hist[2 * tid + 1] = tag1;
    }
    for (u32 i = tid; i < MSD_BINS; i += MSD_BLOCK) {
        if (FROM_TEXT) hist[i] = 0;
        s_off[i] = FROM_TEXT ? a.T[(size_t)i * a.num_ranges1 + r] + a.J1[i] : a.T[(size_t)r * MSD_BINS + i];
    }
    __syncthreads();
    const int shift2 = a.idx_bits + a.key_bits - 2 * MSD_D;
    const u64 low_mask2 = (1ull << shift2) - 1ull;
    for (u32 base = e0; base < e1; base += MSD_TILE) {
        const u32 valid = min(MSD_TILE, e1 - base);
        u64 elem[MSD_IPT];
        u32 dig[MSD_IPT];
        u32 rank[MSD_IPT];
        msd_load_tile<FROM_TEXT>(a, base, valid, elem, dig, shift2);
#pragma unroll
        for (int k = 0; k < MSD_IPT; ++k)
            rank[k] = msd_valid<FROM_TEXT>(base, valid, k, a.n) ? (atomicAdd(&hist[dig[k]], 1u) & 0xffffu) : 0u;
        __syncthreads();                                    // (A) counts complete; previous tile fully written out
        {
            // exclusive scan over the 1024 bins, two adjacent bins per thread
            const u32 c0 = hist[2 * tid] & 0xffffu, c1 = hist[2 * tid + 1] & 0xffffu;
            const u32 ex = block_excl_sum<MSD_WAVES>(c0 + c1, scr, nullptr);
            s_start[2 * tid] = (u16)ex;
            s_start[2 * tid + 1] = (u16)(ex + c0);
            const u32 o0 = s_off[2 * tid], o1 = s_off[2 * tid + 1];
            s_delta[2 * tid] = o0 - ex;
            s_delta[2 * tid + 1] = o1 - (ex + c0);
            s_off[2 * tid] = o0 + c0;
            s_off[2 * tid + 1] = o1 + c1;
            hist[2 * tid] = tag0;
            hist[2 * tid + 1] = tag1;
        }
        __syncthreads();                                    // (B) bin starts published
#pragma unroll
        for (int k = 0; k < MSD_IPT; ++k) {
            if (msd_valid<FROM_TEXT>(base, valid, k, a.n)) {
                const u32 lp = (u32)s_start[dig[k]] + rank[k];
                // G1's element no longer holds its digit, and the output loop needs it: through LDS it travels as
                // [key rest | digit | position in the tile] (13 bits instead of the suffix index, which is base + position)
                if (FROM_TEXT) exch[lp] = ((elem[k] >> a.idx_bits) << 23) | ((u64)dig[k] << 13) | (u64)(tid * MSD_IPT + k);
                else exch[lp] = elem[k];
            }
        }
        __syncthreads();                                    // (C) tile in bin order
#pragma unroll
        for (int k = 0; k < MSD_IPT; ++k) {
            const u32 p = k * MSD_BLOCK + tid;
            if (p < valid) {
                u64 e = exch[p];
                u32 d;
                if (FROM_TEXT) {
                    d = (u32)(e >> 13) & (MSD_BINS - 1u);
                    e = ((e >> 23) << a.idx_bits) | (u64)(base + ((u32)e & (MSD_TILE - 1u)));
                } else {
                    d = (u32)(e >> shift2) & (MSD_BINS - 1u);
                    e = (e & low_mask2) | ((u64)(hist[d] >> 16) << shift2);
                }
                a.out[s_delta[d] + p] = e;
            }
        }
    }
}

// The same pass over tiles of 16384 elements (1024 threads x 16): twice as many elements per bin and tile, so the runs
// the output loop writes are 16 x 8 = 128 bytes on average -- whole lines instead of half lines; the pass is bound by
// the 128-byte lines it touches (docs/history 4.3).  LDS stages 8192 elements at a time: the tile goes through in two
// pieces by destination position.
constexpr u32 MSD_TILE2 = 16384;
constexpr int MSD_PIECES2 = MSD_TILE2 / MSD_TILE;    // 2

template <bool FROM_TEXT, int BLOCK, bool LSD = false>
__global__ __launch_bounds__(BLOCK) void msd_scatter2_kernel(MsdArgs a)
{
    constexpr int IPT = MSD_TILE2 / BLOCK;               // 16 (1024 threads) or 32 (512)
    constexpr int BPT = MSD_BINS / BLOCK;                // bins per thread in the scan
    __shared__ __attribute__((aligned(16))) u64 exch[MSD_TILE];
    __shared__ u32 hist[MSD_BINS], s_delta[MSD_BINS], s_off[MSD_BINS];
    __shared__ u16 s_start[MSD_BINS];
    __shared__ u32 scr[BLOCK / kWave + 1];
    const u32 tid = threadIdx.x, r = blockIdx.x;
    u32 e0, e1;
    u32 tag[BPT];
#pragma unroll
    for (int j = 0; j < BPT; ++j) tag[j] = 0;
    if (FROM_TEXT) {
        if (r >= a.num_ranges1) return;
        e0 = r * a.tiles_per_range1 * MSD_TILE;
        const u64 end = (u64)(r + 1) * a.tiles_per_range1 * MSD_TILE;
        e1 = end < a.n ? (u32)end : a.n;
    } else {
        if (r >= a.counters[0]) return;
        e0 = a.ranges2[r].start;
        e1 = a.ranges2[r].end;
#pragma unroll
        for (int j = 0; j < BPT; ++j)      // see msd_scatter_kernel
            tag[j] = ((u32)a.dense[(size_t)a.ranges2[r].seg * MSD_BINS + BPT * tid + j] & (MSD_TAG_SPAN - 1u)) << 16;
    }
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
        const u32 i = BPT * tid + j;
        hist[i] = tag[j];
        s_off[i] = FROM_TEXT ? a.T[(size_t)i * a.num_ranges1 + r] + a.J1[i] : a.T[(size_t)r * MSD_BINS + i];
    }
    __syncthreads();
    const int shift2 = a.idx_bits + a.key_bits - 2 * MSD_D;
    const u64 low_mask2 = (1ull << shift2) - 1ull;
    const int rest_bits = a.key_bits - MSD_D;
    const u64 rest_mask = (1ull << rest_bits) - 1ull;
    for (u32 base = e0; base < e1; base += MSD_TILE2) {
        const u32 valid = min(MSD_TILE2, e1 - base);
        u64 elem[IPT];
        u32 lp[IPT];          // position in the tile (before the scan: rank inside the bin), digit in the top 10 bits
        if (FROM_TEXT) {
#pragma unroll
            for (int g = 0; g < IPT / 16; ++g) {
                const u32 i0 = base + tid * IPT + g * 16;
                u64 key[16] = {};
                if (i0 < a.n) text_keys16(a.codes, i0, a.code_bits, a.key_chars, a.plus_one, a.key_drop, a.n, key);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const u32 pos = tid * IPT + g * 16 + k;
                    u32 d;
                    u64 keep;                                   // the K - 10 key bits the element carries on
                    if (LSD) {                                  // the pass's digit is the SECOND ten bits; the first ten stay
                        d = (u32)(key[k] >> (rest_bits - MSD_D)) & (MSD_BINS - 1u);
                        keep = ((key[k] >> rest_bits) << (rest_bits - MSD_D)) | (key[k] & (rest_mask >> MSD_D));
                    } else {
                        d = (u32)(key[k] >> rest_bits) & (MSD_BINS - 1u);
                        keep = key[k] & rest_mask;
                    }
                    // through LDS: [key rest | digit | position in the tile]; the suffix index is base + position
                    elem[g * 16 + k] = (keep << 24) | ((u64)d << 14) | (u64)pos;
                    lp[g * 16 + k] = pos < valid ? ((atomicAdd(&hist[d], 1u) & 0xffffu) | (d << 22)) : 0xffffffffu;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                const u32 p = k * BLOCK + tid;
                elem[k] = p < valid ? a.in[base + p] : 0ull;
            }
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                const u32 p = k * BLOCK + tid;
                const u32 d = (u32)(elem[k] >> shift2) & (MSD_BINS - 1u);
                lp[k] = p < valid ? ((atomicAdd(&hist[d], 1u) & 0xffffu) | (d << 22)) : 0xffffffffu;
            }
        }
        __syncthreads();                                    // (A) counts complete; previous tile fully written out
        {
            u32 c[BPT], sum = 0;
#pragma unroll
            for (int j = 0; j < BPT; ++j) {
                c[j] = hist[BPT * tid + j] & 0xffffu;
                sum += c[j];
            }
            u32 ex = block_excl_sum<BLOCK / kWave>(sum, scr, nullptr);
#pragma unroll
            for (int j = 0; j < BPT; ++j) {
                const u32 i = BPT * tid + j;
                s_start[i] = (u16)ex;
                const u32 o = s_off[i];
                s_delta[i] = o - ex;
                s_off[i] = o + c[j];
                hist[i] = tag[j];
                ex += c[j];
            }
        }
        __syncthreads();                                    // (B) bin starts published
#pragma unroll
        for (int k = 0; k < IPT; ++k)
            if (lp[k] != 0xffffffffu) lp[k] = (lp[k] & 0xffffu) + (u32)s_start[lp[k] >> 22];
#pragma unroll
        for (int h = 0; h < MSD_PIECES2; ++h) {
            if (h * MSD_TILE >= valid) break;
            if (h) __syncthreads();                         // the piece before this one is written out
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                const u32 q = lp[k] - h * MSD_TILE;
                if (q < MSD_TILE) exch[q] = elem[k];
            }
            __syncthreads();                                // (C) the piece in bin order
#pragma unroll
            for (int k = 0; k < (int)(MSD_TILE / BLOCK); ++k) {
                const u32 q = k * BLOCK + tid, p = h * MSD_TILE + q;
                if (p < valid) {
                    u64 e = exch[q];
                    u32 d;
                    if (FROM_TEXT) {
                        d = (u32)(e >> 14) & (MSD_BINS - 1u);
                        e = ((e >> 24) << a.idx_bits) | (u64)(base + ((u32)e & (MSD_TILE2 - 1u)));
                    } else {
                        d = (u32)(e >> shift2) & (MSD_BINS - 1u);
                        e = (e & low_mask2) | ((u64)(hist[d] >> 16) << shift2);
                    }
                    a.out[s_delta[d] + p] = e;
                }
            }
        }
    }
}

// ---- decoupled look-back: where does (tile, bin) go? -------------------------------------------------------
// A look-back word: two state bits over a 30-bit number (n < 2^30 on this path).  A: the tile's own count of the bin;
// P: where the bin stands after the tile -- start of the bin's bucket + everything up to and including the tile.
// One 4-byte word per (tile, bin) carries state and number, so nothing needs ordering beyond the word itself: relaxed
// device-scope accesses.  Tiles are handed out by an atomic ticket: every predecessor of a waiting tile is running.
constexpr u32 LB_A = 1u << 30, LB_P = 2u << 30, LB_STATE = 3u << 30, LB_VALUE = LB_A - 1u;
constexpr int LB_WINDOW = 4;                         // predecessors asked per trip (their loads are in flight together)

// thread = bin.  lb_begin publishes the tile's count c and asks the first window of predecessors; lb_finish adds up what
// they published, back to the nearest running total (asking again while it must), publishes the tile's own running total
// and returns where the tile's elements of the bin start.  first: the bin's bucket start (tile 0).  Between the two the
// caller does what does not depend on the answer (scanning its bins, staging its first piece into LDS): the round trip of
// the first window costs nothing then (tests/tools/lookback_micro.hip: 2.34 -> 2.26 ms for the pass).
struct LbAsk {
    u32 v[LB_WINDOW];
};
__device__ __forceinline__ void lb_ask(const u32 *status, u32 p, LbAsk &q)
{
#pragma unroll
    for (u32 w = 0; w < (u32)LB_WINDOW; ++w) {
        const u32 r = p > w ? p - 1 - w : 0u;
        q.v[w] = __hip_atomic_load(&status[(size_t)r * MSD_BINS + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void lb_begin(u32 *status, u32 t, u32 c, LbAsk &q)
{
    if (t == 0) return;
    __hip_atomic_store(&status[(size_t)t * MSD_BINS + threadIdx.x], LB_A | c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lb_ask(status, t, q);
}
__device__ __forceinline__ u32 lb_finish(u32 *status, u32 t, u32 c, u32 first, LbAsk &q)
{
    u32 off = first;
    if (t != 0) {
        u32 sum = 0, p = t;                           // predecessors not yet added: the next one is p - 1
        for (;;) {
            const u32 p0 = p;
            bool stop = false, done = false;
#pragma unroll
            for (u32 w = 0; w < (u32)LB_WINDOW; ++w) {
                if (!stop && w < p0) {
                    const u32 st = q.v[w] & LB_STATE;
                    if (st == 0) {
                        stop = true;                  // not published yet: ask again from here
                    } else {
                        sum += q.v[w] & LB_VALUE;
                        --p;
                        if (st == LB_P) done = stop = true;
                    }
                }
            }
            if (done || p == 0) break;                // (tile 0 publishes P: p == 0 is never the way out)
            lb_ask(status, p, q);
        }
        off = sum;
    }
    __hip_atomic_store(&status[(size_t)t * MSD_BINS + threadIdx.x], LB_P | (off + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return off;
}

// ---- second pass in one sweep: decoupled look-back (LSD order) ---------------------------------------

// Tiles of the look-back pass: every d-region cut into pieces of MSD_TILE2 elements.  The regions are RENUMBERED on the way:
// the non-empty ones first, 0 .. D - 1 in digit order, then the empty ones (each still gets one tile of nothing: it owes
// its row of the joint table).  From here on "d" is that number: a digit value that never occurs in the text (a quarter
// of the 1024 for a 39-symbol alphabet in 6-bit codes) leaves no hole in the numbering of the joint buckets (b, d), which
// is what lets the local sort take "tag - tag of the tile's first bucket" as the bucket's number inside its tile.
// One workgroup, thread = region.
__global__ __launch_bounds__(MSD_BINS) void msd_tiles2_kernel(const u32 *J1, MsdTile2 *tiles, u32 *counters)
{
    __shared__ u32 scr[MSD_BINS / kWave + 1];
    const u32 d = threadIdx.x;
    const u32 s = J1[d], e = J1[d + 1];
    const u32 nr = (e - s + MSD_TILE2 - 1) / MSD_TILE2;
    u32 total = 0, D = 0;
    const u32 first = block_excl_sum<MSD_BINS / kWave>(nr, scr, &total);
    const u32 dn = block_excl_sum<MSD_BINS / kWave>(nr ? 1u : 0u, scr, &D);      // non-empty regions below this one
    if (d == 0) counters[0] = total + (MSD_BINS - D);
    if (nr == 0) {
        const u32 dd = D + (d - dn);                                               // (d - dn: empty regions below)
        tiles[total + (d - dn)] = MsdTile2{s, 0u, dd | 0x80000000u};
    }
    for (u32 k = 0; k < nr; ++k) {
        const u32 ts = s + k * MSD_TILE2;
        tiles[first + k] = MsdTile2{ts, min(MSD_TILE2, e - ts), dn | (k + 1 == nr ? 0x80000000u : 0u)};
    }
}

// Persistent workgroups, tiles by ticket (a waiting tile's predecessors all hold earlier tickets, hence run).  Per tile:
// load, rank with one returning LDS atomic per element, publish the 1024 counts, add up the predecessors' (thread =
// bin; one 4-byte word per (tile, bin) carries state and number, so nothing needs ordering beyond the word itself:
// relaxed device-scope accesses), publish the running totals, stage through LDS in two pieces, write.  Measured in
// tests/tools/lookback_micro.hip at n = 2^29: 2.5 ms against 0.88 + 2.2 ms for histogram pass + scatter from a table;
// a tile asks 5 trips of 4 rows on average (one chain over all tiles; 4 .. 64 chains side by side read fewer rows and
// were no faster: neighbouring tiles then write to far-apart places).
__global__ __launch_bounds__(1024) void msd_scatter_lb_kernel(MsdArgs a)
{
    constexpr int BLOCK = 1024;
    constexpr int IPT = MSD_TILE2 / BLOCK;               // 16
    __shared__ __attribute__((aligned(16))) u64 exch[MSD_TILE];
    __shared__ u32 hist[MSD_BINS], s_delta[MSD_BINS];
    __shared__ u16 s_start[MSD_BINS];
    __shared__ u32 scr[BLOCK / kWave + 1];
    __shared__ u32 s_ticket;
    const u32 tid = threadIdx.x;
    const u32 nt = a.counters[0];
    const int shift2 = a.idx_bits + a.key_bits - 2 * MSD_D;      // the first digit sits right above [remaining key bits | index]
    const u64 low_mask2 = (1ull << shift2) - 1ull;
    hist[tid] = 0;
    __syncthreads();
    for (;;) {
        if (tid == 0) s_ticket = atomicAdd(&a.counters[8], 1u);
        __syncthreads();
        const u32 t = s_ticket;
        if (t >= nt) break;
        const MsdTile2 td = a.tiles2[t];
        const u32 base = td.start, valid = td.count;
        u64 elem[IPT];
        u32 lp[IPT];
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const u32 p = k * BLOCK + tid;
            elem[k] = p < valid ? a.in[base + p] : 0ull;
        }
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const u32 p = k * BLOCK + tid;
            const u32 d = (u32)(elem[k] >> shift2) & (MSD_BINS - 1u);
            lp[k] = p < valid ? (atomicAdd(&hist[d], 1u) | (d << 22)) : 0xffffffffu;
        }
        __syncthreads();                                    // (A) counts complete; previous tile fully written out
        const u32 c = hist[tid];
        LbAsk ask;
        lb_begin(a.status, t, c, ask);                      // my counts are out, the first window of predecessors is on its way
        const u32 ex = block_excl_sum<BLOCK / kWave>(c, scr, nullptr);
        s_start[tid] = (u16)ex;
        hist[tid] = 0;
        __syncthreads();                                    // (B) bin starts published
        // what the element leaves as: [ tag | remaining key bits | index ], tag = joint bucket number mod 64 = d mod 64
        const u64 tag = (u64)(td.d_last & (MSD_RAW_TAG_SPAN - 1u)) << shift2;
#pragma unroll
        for (int k = 0; k < IPT; ++k)
            if (lp[k] != 0xffffffffu) lp[k] = (lp[k] & 0xffffu) + (u32)s_start[lp[k] >> 22];
#pragma unroll
        for (int h = 0; h < MSD_PIECES2; ++h) {
            if (h && h * MSD_TILE >= valid) break;
            if (h) __syncthreads();                         // the piece before this one is written out
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                const u32 q = lp[k] - h * MSD_TILE;
                if (q < MSD_TILE) exch[q] = elem[k];
            }
            if (h == 0) {
                // only the OUTPUT needs to know where the bins go: the look-back is settled behind the staging
                if (t == 0 && tid == 0) a.J[0] = 0;
                const u32 off = lb_finish(a.status, t, c, a.Jb[tid], ask);
                // the last tile of d-region d: where bin b stands now is the end of joint bucket (b, d)
                if (td.d_last >> 31) a.J[(size_t)tid * MSD_BINS + (td.d_last & (MSD_BINS - 1u)) + 1u] = off + c;
                s_delta[tid] = off - ex;
            }
            __syncthreads();                                // (C) the piece in bin order
#pragma unroll
            for (int k = 0; k < (int)(MSD_TILE / BLOCK); ++k) {
                const u32 q = k * BLOCK + tid, p = h * MSD_TILE + q;
                if (p < valid) {
                    const u64 e = exch[q];
                    const u32 d = (u32)(e >> shift2) & (MSD_BINS - 1u);
                    a.out[s_delta[d] + p] = (e & low_mask2) | tag;
                }
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
// (cj, LSD order: the joint bucket number of every non-empty bucket -- the tile rule and the tiles' tags need it)
__global__ __launch_bounds__(256) void msd_compact_kernel(const u32 *J, u32 nb, const u64 *rank, u32 *cstart, u32 *counters, u32 *cj)
{
    u32 mx = 0;
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < nb; j += gridDim.x * blockDim.x) {
        const u32 s = J[j], e = J[j + 1];
        if (e > s) {
            cstart[rank[j]] = s;
            if (cj) cj[rank[j]] = j;
            mx = max(mx, e - s);
        }
    }
    mx = wave_incl_max(mx);
    if (lane_id() == kWave - 1 && mx) atomicMax(&counters[1], mx);
}

// Tiles: the (non-empty) buckets that start inside one MSD_WIN window form a tile (< MSD_WIN + MSD_MAX_BUCKET
// elements); when that exceeds MSD_TILE_CAP the window's last bucket becomes a tile of its own (what is left
// ends before the window does: < MSD_WIN).  A tile stays inside one aligned block of MSD_TAG_SPAN buckets.  Every decision looks at
// one window only, so all of them are taken in parallel -- and tiles come out at ~5 500 elements on `lines`
// instead of the 4 096 of a plain "one tile per 4096-slot window" rule (a fifth fewer tiles).
struct TilePlan {
    u32 win, cap;      // window of the rule above; elements a tile holds at most
    const u32 *cj;     // LSD order: joint bucket numbers of the non-empty buckets -- a tile stays inside one aligned block of
                       // MSD_RAW_TAG_SPAN joint buckets (else: of MSD_TAG_SPAN non-empty ones)
};
__device__ __forceinline__ bool msd_tile_head(const u32 *cstart, u32 ne, u32 n, u32 k, TilePlan tp)
{
    if (tp.cj) {
        if (k == 0 || (tp.cj[k] >> MSD_RAW_TAG_BITS) != (tp.cj[k - 1] >> MSD_RAW_TAG_BITS)) return true;
    } else if ((k & (MSD_TAG_SPAN - 1u)) == 0) {
        return true;
    }
    const u32 w = cstart[k] / tp.win;
    if (cstart[k - 1] / tp.win != w) return true;                        // first bucket of its window
    if (k + 1 < ne && cstart[k + 1] / tp.win == w) return false;         // neither first nor last
    u32 lo = 0, hi = k;                                                  // first bucket of the window: start >= w * WIN
    const u32 ws = w * tp.win;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (cstart[mid] < ws) lo = mid + 1; else hi = mid;
    }
    const u32 end = k + 1 < ne ? cstart[k + 1] : n;
    return end - cstart[lo] > tp.cap;                                    // the window does not fit: its last bucket goes alone
}

struct InTileHead {
    const u32 *cstart;
    u32 ne, n;
    TilePlan tp;
    __device__ u64 operator()(u64 k) const { return msd_tile_head(cstart, ne, n, (u32)k, tp) ? 1u : 0u; }
};

// The same with the number of non-empty buckets read on the device (LSD order: the whole plan is launched without a
// host round trip; the scan runs over all 2^20 bucket numbers, those from *nep on count nothing).
struct InTileHeadDev {
    const u32 *cstart;
    const u64 *nep;
    u32 n;
    TilePlan tp;
    __device__ u64 operator()(u64 k) const
    {
        const u32 ne = (u32)*nep;
        return (k < ne && msd_tile_head(cstart, ne, n, (u32)k, tp)) ? 1u : 0u;
    }
};
__global__ __launch_bounds__(256) void msd_tiles_dev_kernel(const u32 *cstart, const u64 *nep, u32 n, const u64 *rank, const u64 *total,
                                                              u32 *tile_first, TilePlan tp)
{
    const u32 ne = (u32)*nep;
    if (blockIdx.x == 0 && threadIdx.x == 0) tile_first[*total] = ne;      // sentinel behind the last tile
    for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < ne; k += gridDim.x * blockDim.x) {
        if (msd_tile_head(cstart, ne, n, k, tp)) tile_first[rank[k]] = k;
    }
}

__global__ __launch_bounds__(256) void msd_tiles_kernel(const u32 *cstart, u32 ne, u32 n, const u64 *rank, const u64 *total,
                                                          u32 *tile_first, TilePlan tp)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) tile_first[*total] = ne;      // sentinel behind the last tile
    for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < ne; k += gridDim.x * blockDim.x) {
        if (msd_tile_head(cstart, ne, n, k, tp)) tile_first[rank[k]] = k;
    }
}

struct MsdTile {
    u32 e0, count, tag0, nb;      // first element, elements, tag of the first bucket, buckets
};

// ---- output of a sorted tile ------------------------------------------------------------------------

// Where the still-tied suffixes go when the caller wants the active list of the first rerank straight
// from the local sort (instead of re-reading 4 n bytes of flagged suffix array twice).  Ties are rare on
// the texts this path takes (a handful per tile), so the sorting workgroup only drops a record per tied
// element -- SA slot and suffix, bit 31 of the suffix word = "tied with my predecessor" -- into the
// tile's block of the staging arrays, in whatever order the lanes get there (one LDS atomic each; the
// block starts at the tile's own first SA slot: no allocation, no overlap).  msd_gather_kernel then
// orders every block by slot with a bitmap, derives the group heads and packs the blocks in tile order.
struct MsdEmit {
    u32 *st_pos, *st_idx;            // staging (capacity n each)
    u32 *blk_cnt;                    // [tiles] records of every tile
};

// exch[0 .. count) is the sorted tile (position p <-> thread p % 512, row p / 512).  Writes the suffix
// indices to sa_out[e0 ..]; without `em` bit 31 marks "same key as my predecessor" (the contract of
// suffix_sort_flags), with it the array is written clean and a record goes out for every element that
// is tied with a neighbour (it or its successor carries the flag; a tile starts at a bucket boundary,
// so groups never cross tiles).  s_count: one zeroed word of LDS; the caller's next barrier publishes it.
__device__ __forceinline__ void msd_emit_tile(const u64 *exch, u32 count, u32 e0, int idx_bits, u32 *sa_out, const MsdEmit *em,
                                              u32 *s_count)
{
    const u32 tid = threadIdx.x, lane = tid & 63u;
    const u32 imask = (u32)((1ull << idx_bits) - 1ull);
    const u32 rows = (count + MSD_BLOCK - 1) / MSD_BLOCK;
#pragma unroll 1
    for (u32 r = 0; r < rows; ++r) {
        const u32 p = r * MSD_BLOCK + tid;
        const bool valid = p < count;
        u64 x = 0;
        bool tie = false;
        if (valid) {
            x = exch[p];
            tie = p > 0 && (exch[p - 1] >> idx_bits) == (x >> idx_bits);
            sa_out[e0 + p] = ((u32)x & imask) | ((tie && em == nullptr) ? 0x80000000u : 0u);
        }
        if (em != nullptr) {
            const u64 tm = __ballot(tie);
            // my successor's flag: the next lane's, or (lane 63) one more look into LDS
            const bool tie_next = lane < 63 ? ((tm >> (lane + 1)) & 1ull) != 0
                                            : (p + 1 < count && (exch[p + 1] >> idx_bits) == (x >> idx_bits));
            if (valid && (tie || tie_next)) {
                const u32 slot = e0 + atomicAdd(s_count, 1u);
                em->st_pos[slot] = e0 + p;
                em->st_idx[slot] = ((u32)x & imask) | (tie ? 0x80000000u : 0u);
            }
        }
    }
}

struct InBlkCnt {
    const u32 *c;
    __device__ u64 operator()(u64 t) const { return c[t]; }
};

// One wavefront per tile: the tile's records (unordered) -> the active list of the first rerank in slot
// order: (SA slot, suffix, 1 + SA slot of the group's head).  A bitmap of the tile's occupied slots ranks
// a record by the bits below its slot; a second bitmap of the untied ones (group heads -- the head of a
// tied element is itself active, so it has a record) gives its head as the nearest set bit at or below.
// O(records) whatever the tile looks like.
constexpr u32 GA_WORDS = MSD_TILE / 32;      // 256 words per bitmap
__global__ __launch_bounds__(256) void msd_gather_kernel(const MsdTile *tiles, const u32 *blk_cnt, const u64 *dst_off, u32 nt,
                                                           const u32 *st_pos, const u32 *st_idx, u32 *pos, u32 *idx, u32 *grp)
{
    __shared__ u32 s_occ[4][GA_WORDS], s_head[4][GA_WORDS], s_pre[4][GA_WORDS];
    const u32 w = wave_id(), lane = lane_id();
    const u32 t = blockIdx.x * 4 + w;
    if (t >= nt) return;
    const u32 c = blk_cnt[t];
    if (c == 0) return;
    const u32 e0 = tiles[t].e0;
    const u64 d = dst_off[t];
    u32 *occ = s_occ[w], *head = s_head[w], *pre = s_pre[w];
    for (u32 i = lane; i < GA_WORDS; i += kWave) occ[i] = head[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (u32 i = lane; i < c; i += kWave) {
        const u32 q = st_pos[e0 + i] - e0;
        atomicOr(&occ[q >> 5], 1u << (q & 31u));
        if (!(st_idx[e0 + i] >> 31)) atomicOr(&head[q >> 5], 1u << (q & 31u));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {   // exclusive prefix of the popcounts: four consecutive words per lane
        const u32 c0 = __popc(occ[4 * lane]), c1 = __popc(occ[4 * lane + 1]), c2 = __popc(occ[4 * lane + 2]),
                  c3 = __popc(occ[4 * lane + 3]);
        const u32 ex = wave_incl_sum(c0 + c1 + c2 + c3) - (c0 + c1 + c2 + c3);
        pre[4 * lane] = ex;
        pre[4 * lane + 1] = ex + c0;
        pre[4 * lane + 2] = ex + c0 + c1;
        pre[4 * lane + 3] = ex + c0 + c1 + c2;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (u32 i = lane; i < c; i += kWave) {
        const u32 ps = st_pos[e0 + i], q = ps - e0;
        const u32 below = (q & 31u) ? (occ[q >> 5] & ((1u << (q & 31u)) - 1u)) : 0u;
        const u32 rank = pre[q >> 5] + __popc(below);
        // nearest head at or below q (exists: the first element of a group is never flagged)
        u32 wi = q >> 5;
        u32 m = head[wi] & ((q & 31u) == 31u ? ~0u : ((2u << (q & 31u)) - 1u));
        while (m == 0 && wi > 0) m = head[--wi];
        const u32 hq = wi * 32 + (31u - (u32)__builtin_clz(m | 1u));
        pos[d + rank] = ps;
        idx[d + rank] = st_idx[e0 + i] & 0x7fffffffu;
        grp[d + rank] = e0 + hq + 1;
    }
}

// ---- local sort --------------------------------------------------------------------------------

// One workgroup per tile: <= 8192 elements of <= 1024 consecutive joint buckets.  Sort key in LDS:
// [ bucket number inside the tile | remaining key bits | suffix index ] -- stable 8-bit LSD passes
// over the bucket and key bits (the index bits ride along), wave-ballot ranking as in radix_sort.hip.
// This is the general (slower) form: it takes whatever the fast kernel below hands back.
__global__ __launch_bounds__(MSD_BLOCK) void msd_local_sort_kernel(const u64 *in, const MsdTile *tiles, int rem_bits, int idx_bits,
                                                                     u32 *sa_out, const u32 *tile_list, int fused, MsdEmit em_val)
{
    const MsdEmit *em = fused ? &em_val : nullptr;
    __shared__ __attribute__((aligned(16))) u64 exch[MSD_TILE];
    __shared__ u32 wave_hist[MSD_WAVES][256];
    __shared__ u32 scr[MSD_WAVES + 1];
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u32 t = tile_list ? tile_list[blockIdx.x] : blockIdx.x;
    const MsdTile td = tiles[t];
    const u32 e0 = td.e0, count = td.count;
    for (u32 i = tid; i < MSD_WAVES * 256; i += MSD_BLOCK) (&wave_hist[0][0])[i] = 0;
    __syncthreads();
    int seg_bits = 0;
    while ((1u << seg_bits) < td.nb) ++seg_bits;
    const u64 tag_base = (u64)td.tag0 << (rem_bits + idx_bits);
    // element r of this thread sits at tile position wave * 1024 + r * 64 + lane (order = position)
    u64 key[MSD_IPT];
#pragma unroll
    for (int r = 0; r < MSD_IPT; ++r) {
        const u32 p = wave * (kWave * MSD_IPT) + r * kWave + lane;
        // [bucket inside the tile | remaining key bits | index] (G2 tagged the element with its bucket's number); ~0: padding
        key[r] = p < count ? in[e0 + p] - tag_base : ~0ull;
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
    if (tid == 0) scr[0] = 0;
    __syncthreads();
    msd_emit_tile(exch, count, e0, idx_bits, sa_out, em, &scr[0]);
    __syncthreads();
    if (em && tid == 0) em->blk_cnt[t] = scr[0];
}


// The fast form.  One counting pass on the top 12 bits of (bucket, key) with returning LDS atomics (no
// order to preserve: the whole element, index included, is the sort key, so the result is a total order
// anyway) leaves bins of a handful of elements; every element then finds its place inside its bin by
// counting the smaller ones -- neighbouring lanes sit in the same bin, so those LDS reads are broadcasts.
// A tile with a bin above LS_KMAX elements (many equal or nearly equal keys) is handed to the general
// kernel instead.
#ifndef PSS_LS_BIN_BITS
#define PSS_LS_BIN_BITS 12
#endif
constexpr int LS_BIN_BITS = PSS_LS_BIN_BITS;
constexpr u32 LS_BINS = 1u << LS_BIN_BITS;          // 16-bit counters, two per LDS word (a tile has < 8192 elements)
constexpr u32 LS_WORDS = LS_BINS / 2;
constexpr u32 LS_KMAX = 64;
#ifndef PSS_LS_GROUP
#define PSS_LS_GROUP 1
#endif
constexpr int LS_GROUP = PSS_LS_GROUP;              // rows ranked / written out together: their LDS reads are issued back to back
constexpr int LS_WINDOW = PSS_LS_WINDOW;                        // members of its bin every element reads unconditionally (bins average 1.3)

__global__ __launch_bounds__(256) void msd_tile_desc_kernel(const u32 *cstart, const u32 *tile_first, u32 nt, u32 ne, u32 n,
                                                              MsdTile *tiles, const u32 *cj, const u64 *ntp, const u64 *nep)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (ntp) {                     // (counts on the device: see InTileHeadDev)
        nt = (u32)*ntp;
        ne = (u32)*nep;
    }
    if (t >= nt) return;
    const u32 k0 = tile_first[t], k1 = tile_first[t + 1];
    const u32 e0 = cstart[k0], e1 = k1 < ne ? cstart[k1] : n;
    if (cj) {
        // LSD order: the tag is the joint bucket number mod 64, the tile lies inside one block of 64: "buckets" = the span of
        // its tags (a joint bucket that is empty between two of them just takes a number)
        const u32 t0 = cj[k0] & (MSD_RAW_TAG_SPAN - 1u), t1 = cj[k1 - 1] & (MSD_RAW_TAG_SPAN - 1u);
        tiles[t] = MsdTile{e0, e1 - e0, t0, t1 - t0 + 1u};
    } else {
        tiles[t] = MsdTile{e0, e1 - e0, k0 & (MSD_TAG_SPAN - 1u), k1 - k0};
    }
}

// Workgroup barrier that waits for this wave's LDS traffic only: global loads issued before it (the
// prefetch of the next tile) stay in flight across it.
// Pins a loaded value: the load that produced it cannot be sunk into a later branch.
__device__ __forceinline__ void keep_load(u64 &v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// A tile lives ~3 us, and what it needs first -- its descriptor, then the starts of its buckets and its
// elements -- are dependent trips to HBM of ~1 us each: one workgroup per tile leaves the CU waiting for
// memory most of the time.  The workgroups therefore persist (two per CU) and walk over the tiles with the
// loads one tile ahead: descriptor two tiles ahead, bucket starts and elements of the next tile issued as
// soon as the registers of the current one are free, landing while the current tile is being sorted.
// nt_dev (optional): the number of tiles is read on the device, and the kernel does nothing when the plan it would run on
// is not to be used -- a bucket beyond a tile (counters[1]) or a byte without a code (*bad_dev): the host learns both after
// the launch, in the one round trip of the sort.
__global__ __launch_bounds__(MSD_BLOCK, 4) void msd_local_fast_kernel(const u64 *in, const MsdTile *tiles, u32 nt, int rem_bits,
                                                                     int idx_bits, u32 *sa_out, u32 *fail_list, u32 *fail_count,
                                                                     int fused, MsdEmit em_val, const u64 *nt_dev,
                                                                     const u32 *counters_dev, const u32 *bad_dev)
{
    if (nt_dev) {
        if (counters_dev[1] > MSD_MAX_BUCKET || (bad_dev && *bad_dev)) return;
        nt = (u32)*nt_dev;
    }
    const MsdEmit *em = fused ? &em_val : nullptr;
    // 80 KiB of LDS to the byte: two workgroups per CU.  One array: the tile, then the counters -- the ranking below reads
    // up to LS_WINDOW - 1 elements past the tile's last slot (masked out afterwards), and those reads must stay inside it.
    __shared__ __attribute__((aligned(16))) u64 lds_all[MSD_TILE + LS_WORDS];
    u64 *const exch = lds_all;
    u32 *const hist = reinterpret_cast<u32 *>(lds_all + MSD_TILE);      // counts, then bin starts (hist) / running slots (hist2), packed
    u32 *const hist2 = hist + LS_WORDS;
    u32 *const scr = reinterpret_cast<u32 *>(&exch[MSD_TILE - 8]);      // a tile never reaches these slots (MSD_MAX_BUCKET)
    u32 &s_fail = scr[MSD_WAVES + 2];
    const u32 tid = threadIdx.x;
    const u32 stride = gridDim.x;
    u32 t = blockIdx.x;
    if (t >= nt) return;
    MsdTile cur = tiles[t];
    MsdTile nxt = t + stride < nt ? tiles[t + stride] : MsdTile{0, 0, 0, 0};
    u64 pe[MSD_IPT];          // prefetched raw elements of the tile about to be sorted
    auto prefetch = [&](const MsdTile &d) {
#pragma unroll
        for (int r = 0; r < MSD_IPT; ++r) {
            const u32 p = r * MSD_BLOCK + tid;
            pe[r] = p < d.count ? in[d.e0 + p] : 0ull;
        }
    };
    prefetch(cur);
    // Six barriers per tile, not nine: the counters are zeroed for the NEXT tile while this one is permuted (they are not
    // read after the ranking), the tile's bookkeeping (failure flag, record count -> blk_cnt) is settled behind the next
    // tile's first barrier -- by then every wave has left this tile's output -- and nothing separates the tiles otherwise
    // (the next write to the element array is three barriers into the next tile).
    for (u32 i = tid; i < LS_WORDS; i += MSD_BLOCK) hist[i] = 0;
    if (tid == 0) {
        s_fail = 0;
        scr[MSD_WAVES + 3] = 0;            // records the tile has emitted
    }
    lds_barrier();
    u32 prev_t = 0;
    bool prev_ok = false;
    for (;;) {
        const u32 count = cur.count, e0 = cur.e0;
        const u32 rows = (count + MSD_BLOCK - 1) / MSD_BLOCK;          // uniform over the workgroup
        int seg_bits = 0;
        while ((1u << seg_bits) < cur.nb) ++seg_bits;
        const u64 tag_base = (u64)cur.tag0 << (rem_bits + idx_bits);
        const int sort_bits = rem_bits + seg_bits;
        const int bin_shift = idx_bits + (sort_bits > LS_BIN_BITS ? sort_bits - LS_BIN_BITS : 0);
        u64 e[MSD_IPT];
#pragma unroll
        for (int r = 0; r < MSD_IPT; ++r) {
            e[r] = ~0ull;
            if ((u32)r < rows) {
                const u32 p = r * MSD_BLOCK + tid;
                if (p < count) {
                    // the element carries its bucket (G2 put it there): [bucket inside the tile | remaining key bits | index]
                    e[r] = pe[r] - tag_base;
                    const u32 bin = (u32)(e[r] >> bin_shift);
                    atomicAdd(&hist[bin >> 1], 1u << (16u * (bin & 1u)));    // count now; the slot is taken after the scan
                }
            }
        }
        // the raw elements are consumed: their registers take the next tile while this one is sorted
        const bool more = t + stride < nt;
        if (more) prefetch(nxt);
        lds_barrier();
        if (tid == 0) {                    // the previous tile is over for every wave: its record count, the flags for this one
            if (em && prev_ok) em->blk_cnt[prev_t] = scr[MSD_WAVES + 3];
            s_fail = 0;
            scr[MSD_WAVES + 3] = 0;
        }
        {
            // exclusive scan over the bins in place: LS_WPT adjacent words = 2 LS_WPT bins per thread (four words at 4096 bins)
            constexpr int LS_WPT = LS_WORDS / MSD_BLOCK;
            static_assert(LS_WPT * MSD_BLOCK == (int)LS_WORDS && LS_WPT >= 1, "bins split evenly over the threads");
            u32 c[2 * LS_WPT];
            u32 sum = 0;
#pragma unroll
            for (int j = 0; j < LS_WPT; ++j) {
                const u32 wv = hist[LS_WPT * tid + j];
                c[2 * j] = wv & 0xffffu;
                c[2 * j + 1] = wv >> 16;
                sum += c[2 * j] + c[2 * j + 1];
            }
            const u32 incl = wave_incl_sum(sum);
            if (lane_id() == kWave - 1) scr[wave_id()] = incl;
            lds_barrier();
            u32 ex = incl - sum;
            for (int w = 0; w < wave_id(); ++w) ex += scr[w];
#pragma unroll
            for (int j = 0; j < LS_WPT; ++j) {
                const u32 lo = ex, hi = ex + c[2 * j];
                hist[LS_WPT * tid + j] = hist2[LS_WPT * tid + j] = lo | (hi << 16);
                ex = hi + c[2 * j + 1];
            }
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < MSD_IPT; ++r) {
            if ((u32)r < rows) {
                const u32 p = r * MSD_BLOCK + tid;
                // (a second returning atomic on the bin's running start hands out the slot: one more LDS atomic per
                // element, sixteen fewer registers held across the prefetch)
                if (p < count) {
                    const u32 bin = (u32)(e[r] >> bin_shift), sh = 16u * (bin & 1u);
                    exch[(atomicAdd(&hist2[bin >> 1], 1u << sh) >> sh) & 0xffffu] = e[r];
                }
            }
        }
        // sentinels behind the tile: the ranking window of the last bins reads them (count + 7 < the scalars' slots: MSD_TILE_CAP)
        if (tid < (u32)LS_WINDOW) exch[count + tid] = ~0ull;
        lds_barrier();
        // Place inside the bin = number of smaller elements there (thread <-> position: neighbours share the bin, their
        // reads are broadcasts).  The kernel is bound by the latency of dependent LDS reads, not by their number: every
        // element reads a fixed window of LS_WINDOW members of its bin (no loop, no branch -- a read past the bin is
        // masked out), LS_GROUP rows at a time with all their reads in flight; a longer bin (rare: bins average 1.3
        // elements) finishes in a loop.
        const u16 *const starts16 = reinterpret_cast<const u16 *>(hist);      // starts16[bin] = first slot of the bin
        u32 rk[MSD_IPT];
#pragma unroll
        for (int g = 0; g < MSD_IPT; g += LS_GROUP) {
            if ((u32)g < rows) {
                u64 x[LS_GROUP];
                u32 s0[LS_GROUP], sm[LS_GROUP];
#pragma unroll
                for (int j = 0; j < LS_GROUP; ++j) x[j] = exch[(g + j) * MSD_BLOCK + tid];
#pragma unroll
                for (int j = 0; j < LS_GROUP; ++j) {
                    const u32 bin = (u32)(x[j] >> bin_shift) & (LS_BINS - 1u);      // (a slot past the tile holds anything)
                    s0[j] = starts16[bin];
                }
                u64 y[LS_GROUP][LS_WINDOW];
#pragma unroll
                for (int j = 0; j < LS_GROUP; ++j) {
#pragma unroll
                    for (int k = 0; k < LS_WINDOW; ++k) y[j][k] = exch[s0[j] + k];
                }
#pragma unroll
                for (int j = 0; j < LS_GROUP; ++j) {
                    u32 c = 0;
#pragma unroll
                    for (int k = 0; k < LS_WINDOW; ++k) {
                        keep_load(y[j][k]);      // (or the compiler makes every read conditional on k < len: a branch and a wait per read)
                        // (no "k < len": what lies behind my bin in the window belongs to LATER bins -- larger numbers, the
                        // elements being in bin order -- or is a sentinel; eight compares and eight mask operations fewer per element)
                        c += y[j][k] < x[j] ? 1u : 0u;
                    }
                    sm[j] = c;
                }
#pragma unroll
                for (int j = 0; j < LS_GROUP; ++j) {
                    // the window's last member still in my bin: the bin may go on (rare: bins average 1.3 elements) -- only then
                    // is its length looked up (the bin's end = the next bin's start)
                    const u32 p = (g + j) * MSD_BLOCK + tid;
                    if (p < count && ((y[j][LS_WINDOW - 1] ^ x[j]) >> bin_shift) == 0) {
                        const u32 bin = (u32)(x[j] >> bin_shift) & (LS_BINS - 1u);
                        const u32 s1 = bin + 1u < LS_BINS ? (u32)starts16[min(bin + 1u, LS_BINS - 1u)] : count;
                        const u32 len = s1 - s0[j];
                        if (len > LS_KMAX) {
                            s_fail = 1;
                        } else {
                            for (u32 q = s0[j] + LS_WINDOW; q < s0[j] + len; ++q) sm[j] += exch[q] < x[j] ? 1u : 0u;
                        }
                    }
                    e[g + j] = x[j];
                    rk[g + j] = s0[j] + sm[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < LS_GROUP; ++j) rk[g + j] = 0;
            }
        }
        lds_barrier();
        const bool failed = s_fail != 0;
        if (failed) {
            if (tid == 0) fail_list[atomicAdd(fail_count, 1u)] = t;
        } else {
#pragma unroll
            for (int r = 0; r < MSD_IPT; ++r) {
                if ((u32)r < rows) {
                    const u32 p = r * MSD_BLOCK + tid;
                    if (p < count) exch[rk[r]] = e[r];
                }
            }
        }
        for (u32 i = tid; i < LS_WORDS; i += MSD_BLOCK) hist[i] = 0;      // (bin starts: last read in the ranking) for the next tile
        lds_barrier();
        if (!failed) {
            // Output (msd_emit_tile with the loads of LS_GROUP rows in flight): suffix indices to sa_out[e0 ..]; a record for
            // every element tied with a neighbour when the first rerank is fused, else bit 31 = "same key as my predecessor".
            const u32 imask = (u32)((1ull << idx_bits) - 1ull);
            u32 *const s_count = &scr[MSD_WAVES + 3];
#pragma unroll
            for (int g = 0; g < MSD_IPT; g += LS_GROUP) {
                if ((u32)g < rows) {
                    u64 cur[LS_GROUP], prv[LS_GROUP], nxt[LS_GROUP];
#pragma unroll
                    for (int j = 0; j < LS_GROUP; ++j) {
                        const u32 p = (g + j) * MSD_BLOCK + tid;
                        cur[j] = exch[p];
                        prv[j] = exch[p ? p - 1 : 0];
                        nxt[j] = exch[p + 1];                                   // (position 8191 + 1 is the first counter word)
                    }
#pragma unroll
                    for (int j = 0; j < LS_GROUP; ++j) {
                        keep_load(cur[j]);
                        keep_load(prv[j]);
                        keep_load(nxt[j]);
                    }
#pragma unroll
                    for (int j = 0; j < LS_GROUP; ++j) {
                        const u32 p = (g + j) * MSD_BLOCK + tid;
                        if (p < count) {
                            const u64 kx = cur[j] >> idx_bits;
                            const bool tie = p > 0 && (prv[j] >> idx_bits) == kx;
                            const u32 sfx = (u32)cur[j] & imask;
                            sa_out[e0 + p] = sfx | ((tie && em == nullptr) ? 0x80000000u : 0u);
                            if (em != nullptr) {
                                const bool tie_next = p + 1 < count && (nxt[j] >> idx_bits) == kx;
                                if (tie || tie_next) {
                                    const u32 slot = e0 + atomicAdd(s_count, 1u);
                                    em->st_pos[slot] = e0 + p;
                                    em->st_idx[slot] = sfx | (tie ? 0x80000000u : 0u);
                                }
                            }
                        }
                    }
                }
            }
        }
        prev_t = t;
        prev_ok = !failed;
        if (!more) break;
        t += stride;
        cur = nxt;
        nxt = t + stride < nt ? tiles[t + stride] : MsdTile{0, 0, 0, 0};
    }
    if (em) {
        lds_barrier();
        if (tid == 0 && prev_ok) em->blk_cnt[prev_t] = scr[MSD_WAVES + 3];
    }
}

#include "ss_sort_impl.h"

// ---- host --------------------------------------------------------------------------------------

// Upper bound on the tiles of the plan: at most two per MSD_WIN window (msd_tile_head) plus one per aligned block
// of MSD_TAG_SPAN non-empty / MSD_RAW_TAG_SPAN joint buckets.
static size_t msd_max_tiles(uint32_t n)
{
    return (size_t)n / (SS_WIN / 2) + ((size_t)MSD_BINS * MSD_BINS) / MSD_RAW_TAG_SPAN + 8;  // (the smaller window of the two paths, the
                                                                                             //  finer block rule: LSD order)
}

static size_t msd_max_tiles2(uint32_t n) { return (size_t)n / MSD_TILE2 + MSD_BINS + 8; }

// LSD order + look-back: the default wherever its 30-bit look-back words hold the positions.  PSS_MSD_LSD=0: the two
// passes in MSD order with the second histogram pass (rounds 2-5).
static bool msd_use_lsd(uint32_t n)
{
    const char *e = knob("PSS_MSD_LSD");
    return !(e && atoi(e) == 0) && n < (1u << 30);
}

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
           + (msd_max_tiles(n) + 32) * 8              // tile_first, then the tiles left to the general kernel
           + (msd_max_tiles(n) + 32) * 16             // blocks of active records per tile + their final offsets
           + (msd_max_tiles(n) + 32) * sizeof(MsdTile)
           + (SC_MAX_BLOCKS + 8) * 8 + 8192
           // LSD order: counts of the first digit per range, the look-back tiles and their words, joint numbers of the buckets
           + (size_t)MSD_BINS * MSD_G1_RANGES * 4 + (MSD_BINS + 8) * 4
           + msd_max_tiles2(n) * (sizeof(MsdTile2) + (size_t)MSD_BINS * 4)
           + (nbk + 8) * 4 + 4096;
}

int msd_max_key_bits(uint32_t n)
{
    int ib = 1;
    while ((1ull << ib) < (u64)n) ++ib;
    // [bucket tag | K - 20 key bits | ib index bits] must fit 64 bits after the second pass (an 11-bit tag, or 6 bits in LSD
    // order), and [K - 10 key bits | ib index bits] after the first
    const int after2 = 64 + MSD_D - (msd_use_lsd(n) ? MSD_RAW_TAG_BITS : MSD_TAG_BITS) - ib + MSD_D;
    return std::min(after2, 64 + MSD_D - ib);
}

int msd_suffix_sort(DeviceCtx *ctx, const TextKeys *text, uint32_t n, int key_bits, uint64_t *A[2], uint32_t *sa_out,
                    void *work, uint32_t *h_small, bool profile, MsdStats *stats, bool *accepted, MsdActive *active,
                    const MsdFront *front)
{
    *accepted = false;
    hipStream_t s = ctx->stream;
    int ib = 1;
    while ((1ull << ib) < (u64)n) ++ib;
    if (key_bits < 2 * MSD_D + 1 || key_bits > msd_max_key_bits(n) || n < 2) {
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
    const size_t max_tiles = msd_max_tiles(n);
    MsdTile *tiles_all = reinterpret_cast<MsdTile *>(carve((max_tiles + 16) * sizeof(MsdTile)));
    u32 *tile_first = reinterpret_cast<u32 *>(carve((max_tiles + 16) * 8));
    u32 *blk_cnt = reinterpret_cast<u32 *>(carve((max_tiles + 8) * 4));
    u64 *dst_off = reinterpret_cast<u64 *>(carve((max_tiles + 8) * 8));
    u64 *partial = reinterpret_cast<u64 *>(carve((SC_MAX_BLOCKS + 8) * 8));
    u64 *d_total = partial + SC_MAX_BLOCKS;
    const bool lsd = msd_use_lsd(n);
    const size_t max_tiles2 = msd_max_tiles2(n);
    u32 *T2 = nullptr, *Jb = nullptr, *status = nullptr, *cj = nullptr;
    MsdTile2 *tiles2 = nullptr;
    if (lsd) {
        T2 = reinterpret_cast<u32 *>(carve((size_t)MSD_BINS * MSD_G1_RANGES * 4));
        Jb = reinterpret_cast<u32 *>(carve((MSD_BINS + 8) * 4));
        tiles2 = reinterpret_cast<MsdTile2 *>(carve(max_tiles2 * sizeof(MsdTile2)));
        status = reinterpret_cast<u32 *>(carve(max_tiles2 * MSD_BINS * 4));
        cj = reinterpret_cast<u32 *>(carve((nbk + 8) * 4));
    }

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
    a.lsd = lsd ? 1 : 0;
    a.T2 = T2;
    a.Jb = Jb;
    a.tiles2 = tiles2;
    a.status = status;

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
    if (front) {
        a.raw = front->raw;
        a.lut = front->lut;
        a.codes_out = const_cast<u8 *>(text->codes);
        a.bad = front->bad;
        hipLaunchKernelGGL(msd_hist_raw_kernel, dim3(a.num_ranges1), dim3(MSD_BLOCK), 0, s, a);
    } else {
        hipLaunchKernelGGL(msd_hist_kernel<true>, dim3(a.num_ranges1), dim3(MSD_BLOCK), 0, s, a);
    }
    static_assert(MSD_G1_RANGES <= 1024, "msd_offsets1_kernel takes four ranges per thread");
    hipLaunchKernelGGL(msd_offsets1_kernel, dim3(MSD_BINS), dim3(256), 0, s, T, a.num_ranges1, seg_first);   // totals: scratch
    hipLaunchKernelGGL(msd_offsets1b_kernel, dim3(1), dim3(MSD_BINS), 0, s, (const u32 *)seg_first, J1, n);
    if (lsd) {
        // the first digit's bucket starts from the same pass over the text (the per-range prefixes in T2 are not used)
        hipLaunchKernelGGL(msd_offsets1_kernel, dim3(MSD_BINS), dim3(256), 0, s, T2, a.num_ranges1, seg_first);
        hipLaunchKernelGGL(msd_offsets1b_kernel, dim3(1), dim3(MSD_BINS), 0, s, (const u32 *)seg_first, Jb, n);
        // (the words of the look-back pass: zero = nothing published)
        PSS_HIP(hipMemsetAsync(status, 0, max_tiles2 * MSD_BINS * 4, s));
    }
    PSS_TRY(mark());
    // 16384-element scatter tiles (whole-line runs), 1024 threads: 2.2 -> 1.9 ms (from the text) and 2.8 -> 2.45 ms
    // (second pass, with its sixteen loads per thread issued before the ranking atomics) at 2^29; 512 threads x 32 elements
    // spill.  PSS_MSD_SCATTER=1: the 8192-element kernels.
    const bool wide = lsd || !(knob("PSS_MSD_SCATTER") && atoi(knob("PSS_MSD_SCATTER")) == 1);
    if (lsd) hipLaunchKernelGGL((msd_scatter2_kernel<true, 1024, true>), dim3(a.num_ranges1), dim3(1024), 0, s, a);
    else if (wide) hipLaunchKernelGGL((msd_scatter2_kernel<true, 1024>), dim3(a.num_ranges1), dim3(1024), 0, s, a);
    else hipLaunchKernelGGL(msd_scatter_kernel<true>, dim3(a.num_ranges1), dim3(MSD_BLOCK), 0, s, a);
    PSS_TRY(mark());
    a.in = A[0];
    a.out = A[1];
    if (lsd) {
        // ---- second pass in one sweep: A[0] -> A[1] by the first digit, tile order kept (look-back); leaves J ----
        hipLaunchKernelGGL(msd_tiles2_kernel, dim3(1), dim3(MSD_BINS), 0, s, (const u32 *)J1, tiles2, counters);
        PSS_TRY(mark());
        const u32 grid2 = std::min<u32>((u32)max_tiles2, (u32)ctx->num_cus);      // one 1024-thread workgroup per CU (~100 VGPRs)
        hipLaunchKernelGGL(msd_scatter_lb_kernel, dim3(grid2), dim3(1024), 0, s, a);
        PSS_TRY(mark());
    } else {
        // ---- G2: A[0] -> A[1], every G1 bucket by the next 10 bits ----
        hipLaunchKernelGGL(msd_ranges_kernel, dim3(1), dim3(MSD_BINS), 0, s, J1, ranges2, seg_first, counters);
        a.dense = ranks;
        hipLaunchKernelGGL(msd_hist_kernel<false>, dim3((u32)max_ranges2), dim3(MSD_BLOCK), 0, s, a);
        hipLaunchKernelGGL(msd_offsets_kernel, dim3(MSD_BINS), dim3(MSD_BINS), 0, s, T, (const u32 *)seg_first, (const u32 *)J1, J,
                           0u, n, MSD_BINS);
    }
    // ---- plan: non-empty joint buckets, largest bucket, tiles ----
    PSS_TRY(device_excl_scan(ctx, InNonEmpty{J}, nbk, partial, d_total, ranks));
    hipLaunchKernelGGL(msd_compact_kernel, dim3(1024), dim3(256), 0, s, J, (u32)nbk, ranks, cstart, counters, cj);
    const int rem_bits = key_bits - 2 * MSD_D;
    MsdEmit em{};
    const int fused = active != nullptr;
    if (fused) em = MsdEmit{active->st_pos, active->st_idx, blk_cnt};
    // LSD order: nothing between here and the local sort needs the host -- the counts stay on the device (the scans run
    // over all 2^20 bucket numbers), the local sort checks the plan's verdict itself, ONE round trip afterwards tells the
    // host everything (rounds 2-5: three, ~0.1 ms of idle GPU each at n = 2^29).
    const bool one_trip = lsd && !knob("PSS_MSD_SLOW_LOCAL");
    u64 *d_total2 = d_total + 1;
    u32 *const fail_list_dev = tile_first + max_tiles + 16;      // (tile_first has 2 (max_tiles + 16) slots)
    if (one_trip) {
        const TilePlan tpd{MSD_WIN, MSD_TILE_CAP, cj};
        PSS_TRY(device_excl_scan(ctx, InTileHeadDev{cstart, d_total, n, tpd}, nbk, partial, d_total2, ranks));
        hipLaunchKernelGGL(msd_tiles_dev_kernel, dim3(1024), dim3(256), 0, s, cstart, (const u64 *)d_total, n, ranks, (const u64 *)d_total2,
                           tile_first, tpd);
        hipLaunchKernelGGL(msd_tile_desc_kernel, dim3((u32)((max_tiles + 255) / 256)), dim3(256), 0, s, cstart, tile_first, 0u, 0u, n,
                           tiles_all, (const u32 *)cj, (const u64 *)d_total2, (const u64 *)d_total);
        PSS_TRY(mark());            // (events 2 .. 5: second pass, then the local sort)
        hipLaunchKernelGGL(msd_local_fast_kernel, dim3(2u * (u32)ctx->num_cus), dim3(MSD_BLOCK), 0, s, A[1], (const MsdTile *)tiles_all, 0u,
                           rem_bits, ib, sa_out, fail_list_dev, counters + 4, fused, em, (const u64 *)d_total2, (const u32 *)counters,
                           front ? (const u32 *)front->bad : (const u32 *)nullptr);
        PSS_TRY(mark());
        PSS_HIP(hipMemcpyAsync(h_small + 8, d_total2, 8, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipMemcpyAsync(h_small + 10, counters + 4, 4, hipMemcpyDeviceToHost, s));
    }
    PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipMemcpyAsync(h_small + 2, counters, 16, hipMemcpyDeviceToHost, s));
    if (front) PSS_HIP(hipMemcpyAsync(h_small + 6, front->bad, 4, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipStreamSynchronize(s));
    const u32 ne = h_small[0];
    const u32 maxb = h_small[3];
    if (stats) {
        stats->buckets = ne;
        stats->max_bucket = maxb;
        stats->bad_symbol = front ? h_small[6] : 0u;
    }
    if (front && h_small[6]) return PSS_OK;        // a byte the table has no code for: nothing of this sort can be used
    if (maxb > MSD_MAX_BUCKET) return PSS_OK;      // not this text: the caller takes the LSD path
    if (!lsd) {
        PSS_TRY(mark());
        if (wide) hipLaunchKernelGGL((msd_scatter2_kernel<false, 1024>), dim3((u32)max_ranges2), dim3(1024), 0, s, a);
        else hipLaunchKernelGGL(msd_scatter_kernel<false>, dim3((u32)max_ranges2), dim3(MSD_BLOCK), 0, s, a);
        PSS_TRY(mark());
    }
    const auto local_sort = msd_local_sort_kernel;
    const auto local_fast = msd_local_fast_kernel;
    u32 nt = 0;
    u32 *fail_list = fail_list_dev;
    if (one_trip) {
        nt = h_small[8];
        if ((size_t)nt > max_tiles) {
            set_error("msd_suffix_sort: %u tiles planned, tables hold %zu (internal error)", nt, max_tiles);
            return PSS_EDEVICE;
        }
        const u32 nfail = h_small[10];
        if (stats) stats->slow_tiles = nfail;
        if (nfail)
            hipLaunchKernelGGL(local_sort, dim3(nfail), dim3(MSD_BLOCK), 0, s, A[1], (const MsdTile *)tiles_all, rem_bits,
                               ib, sa_out, (const u32 *)fail_list, fused, em);
    } else {
    const TilePlan tp{MSD_WIN, MSD_TILE_CAP, cj};
    PSS_TRY(device_excl_scan(ctx, InTileHead{cstart, ne, n, tp}, ne, partial, d_total, ranks));
    hipLaunchKernelGGL(msd_tiles_kernel, dim3(1024), dim3(256), 0, s, cstart, ne, n, ranks, (const u64 *)d_total, tile_first, tp);
    PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipStreamSynchronize(s));
    nt = h_small[0];
    if ((size_t)nt > max_tiles) {       // cannot happen (msd_max_tiles is an upper bound of the plan): never run past the tables
        set_error("msd_suffix_sort: %u tiles planned, tables hold %zu (internal error)", nt, max_tiles);
        return PSS_EDEVICE;
    }
    PSS_TRY(mark());
    // counters[4] = tiles the fast kernel declined (their numbers go behind the tile table), [5] = active records
    fail_list = tile_first + nt + 8;
    hipLaunchKernelGGL(msd_tile_desc_kernel, dim3((nt + 255) / 256), dim3(256), 0, s, cstart, tile_first, nt, ne, n, tiles_all,
                       (const u32 *)cj, (const u64 *)nullptr, (const u64 *)nullptr);
    if (knob("PSS_MSD_SLOW_LOCAL")) {
        hipLaunchKernelGGL(local_sort, dim3(nt), dim3(MSD_BLOCK), 0, s, A[1], (const MsdTile *)tiles_all, rem_bits, ib,
                           sa_out, (const u32 *)nullptr, fused, em);
    } else {
        MsdTile *tiles = tiles_all;
        const u32 grid = std::min<u32>(nt, 2u * (u32)ctx->num_cus);
        hipLaunchKernelGGL(local_fast, dim3(grid), dim3(MSD_BLOCK), 0, s, A[1], (const MsdTile *)tiles, nt, rem_bits, ib,
                           sa_out, fail_list, counters + 4, fused, em, (const u64 *)nullptr, (const u32 *)nullptr, (const u32 *)nullptr);
        PSS_TRY(mark());            // (profile mode) the events bracket this launch alone
        PSS_HIP(hipMemcpyAsync(h_small, counters + 4, 4, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
        const u32 nfail = h_small[0];
        if (stats) stats->slow_tiles = nfail;
        if (nfail)
            hipLaunchKernelGGL(local_sort, dim3(nfail), dim3(MSD_BLOCK), 0, s, A[1], (const MsdTile *)tiles_all, rem_bits,
                               ib, sa_out, (const u32 *)fail_list, fused, em);
    }
    if (knob("PSS_MSD_SLOW_LOCAL")) PSS_TRY(mark());
    }
    if (fused) {
        PSS_TRY(device_excl_scan(ctx, InBlkCnt{blk_cnt}, nt, partial, d_total, dst_off));
        hipLaunchKernelGGL(msd_gather_kernel, dim3((nt + 3) / 4), dim3(256), 0, s, (const MsdTile *)tiles_all, blk_cnt, dst_off, nt,
                           active->st_pos, active->st_idx, active->pos, active->idx, active->grp);
        PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
        active->count = h_small[0];
    }
    PSS_HIP(hipGetLastError());
    if (stats) {
        stats->tiles = nt;
        stats->lookback = lsd ? 1u : 0u;
    }
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

// ---- sample sort: host ---------------------------------------------------------------------------------------------

struct SsGeometry {
    u32 B1, B2, os, S;
    int ib, kc, key_bits;
    u64 plo, phi;          // radix^(kc - 1)
};
static bool ss_geometry(uint32_t n, uint32_t radix, SsGeometry *g)
{
    if (n < (1u << 16) || n > (1u << 30) || radix < 2 || radix > 257) return false;
    int ib = 1;
    while ((1ull << ib) < (u64)n) ++ib;
    // symbols per key: the largest kc with radix^kc <= 2^(128 - ib), at most SS_MAX_CHARS
    const unsigned __int128 limit = (unsigned __int128)1 << (128 - ib);
    unsigned __int128 pw = 1;
    int kc = 0;
    while (kc < SS_MAX_CHARS && pw <= limit / radix) {
        pw *= radix;
        ++kc;
    }
    if (kc < 2) return false;
    unsigned __int128 lead = 1;
    for (int i = 0; i + 1 < kc; ++i) lead *= radix;
    g->plo = (u64)lead;
    g->phi = (u64)(lead >> 64);
    int kb = 0;
    while (kb < 128 && ((pw - 1) >> kb) != 0) ++kb;      // bits of the largest key
    g->key_bits = kb;
    // joint buckets of ~512 elements (a tile holds 4088): B1 x B2 of them, both powers of two <= 1024
    int lb = 4;
    while (((u64)512 << lb) < (u64)n && lb < 20) ++lb;
    const u32 B1 = 1u << ((lb + 1) / 2), B2 = 1u << (lb / 2);
    // sample members per bucket: the bucket sizes follow a gamma distribution of that shape -- 4 puts a bucket of 512 on
    // average beyond 4088 with probability 1e-10, an average of 1024 (n = 2^30) needs 8
    const u32 os = ((u64)n > (u64)768 * B1 * B2) ? 8u : 4u;
    g->B1 = B1;
    g->B2 = B2;
    g->os = os;
    g->S = os * B1 * B2;
    g->ib = ib;
    g->kc = kc;
    return (u64)g->S * 4 <= (u64)n;
}
uint32_t ss_sample_count(uint32_t n)
{
    SsGeometry g;
    return ss_geometry(n, 257, &g) ? g.S : 0u;
}
// Everything about the sort that depends on n and the alphabet, in one number (0: the sort does not take this text): two
// texts with the same tag can be cut by the same sorted sample (sa_build.hip, the plan of the sample sort).
uint64_t ss_geometry_tag(uint32_t n, uint32_t radix)
{
    SsGeometry g;
    if (!ss_geometry(n, radix, &g)) return 0;
    return ((uint64_t)g.S << 32) | ((uint64_t)g.B1 << 20) | ((uint64_t)g.B2 << 8) | ((uint64_t)g.ib << 2) | (uint64_t)(g.os == 8) | ((uint64_t)g.kc << 58);
}
int ss_key_chars(uint32_t n, uint32_t radix)
{
    SsGeometry g;
    return ss_geometry(n, radix, &g) ? g.kc : 0;
}

int ss_suffix_sort(DeviceCtx *ctx, const TextKeys *text, uint32_t radix, uint32_t n, const SsBuffers &buf, uint32_t *sa_out,
                   void *work, uint32_t *h_small, bool profile, SsStats *stats, bool *accepted, MsdActive *active)
{
    *accepted = false;
    hipStream_t s = ctx->stream;
    SsGeometry g;
    if (!ss_geometry(n, radix, &g)) return PSS_OK;      // not this text
    static_assert(SS_SBLOCK == (int)MSD_BINS, "one bin per thread in the scatter passes");
    const size_t max_ranges2 = (size_t)n / MSD_G2_RANGE + MSD_BINS + 8;
    const size_t nbk = (size_t)MSD_BINS * MSD_BINS;
    u8 *w = static_cast<u8 *>(work);
    size_t o = 0;
    auto carve = [&](size_t bytes) { u8 *p = w + o; o = round_up(o + bytes, 256); return p; };
    // (the same carving as msd_suffix_sort: msd_workspace_bytes covers it)
    u32 *T = reinterpret_cast<u32 *>(carve(max_ranges2 * MSD_BINS * 4));
    u32 *J1 = reinterpret_cast<u32 *>(carve((MSD_BINS + 8) * 4));
    u32 *J = reinterpret_cast<u32 *>(carve((nbk + 8) * 4));
    MsdRange *ranges2 = reinterpret_cast<MsdRange *>(carve(max_ranges2 * sizeof(MsdRange)));
    u32 *seg_first = reinterpret_cast<u32 *>(carve((MSD_BINS + 8) * 4));
    u32 *counters = reinterpret_cast<u32 *>(carve(64));
    u64 *ranks = reinterpret_cast<u64 *>(carve((nbk + 8) * 8));
    u32 *cstart = reinterpret_cast<u32 *>(carve((nbk + 8) * 4));
    const size_t max_tiles = msd_max_tiles(n);
    MsdTile *tiles_all = reinterpret_cast<MsdTile *>(carve((max_tiles + 16) * sizeof(MsdTile)));
    u32 *tile_first = reinterpret_cast<u32 *>(carve((max_tiles + 16) * 8));
    u32 *blk_cnt = reinterpret_cast<u32 *>(carve((max_tiles + 8) * 4));
    u64 *dst_off = reinterpret_cast<u64 *>(carve((max_tiles + 8) * 8));
    u64 *partial = reinterpret_cast<u64 *>(carve((SC_MAX_BLOCKS + 8) * 8));
    u64 *d_total = partial + SC_MAX_BLOCKS;

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

    SsText tx{text->codes, n, text->code_bits, g.kc, text->plus_one, g.ib, radix, g.plo, g.phi};
    const u32 S = g.S;
    E16 *E0 = static_cast<E16 *>(buf.E0), *E = static_cast<E16 *>(buf.E);
    PSS_TRY(mark());                                                                       // [0]
    // ---- the sample, sorted as 128-bit numbers ----
    if (buf.sample_in) {
        E = const_cast<E16 *>(static_cast<const E16 *>(buf.sample_in));      // (read only from here on)
    } else {
    hipLaunchKernelGGL(ss_sample_kernel, dim3((S + 255) / 256), dim3(256), 0, s, tx, S, E0, buf.K[0], buf.V[0]);
    if (S <= 8192) {
        // (small texts, tests: the device sort's one-workgroup path is not stable, which the second of the chained sorts needs)
        std::vector<E16> hs(S);
        PSS_HIP(hipMemcpyAsync(hs.data(), E0, (size_t)S * 16, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
        std::sort(hs.begin(), hs.end(), [](const E16 &x, const E16 &y) { return x.hi < y.hi || (x.hi == y.hi && x.lo < y.lo); });
        PSS_HIP(hipMemcpyAsync(E, hs.data(), (size_t)S * 16, hipMemcpyHostToDevice, s));
        PSS_HIP(hipStreamSynchronize(s));
    } else {
        u64 *K[2] = {buf.K[0], buf.K[1]};
        u32 *V[2] = {buf.V[0], buf.V[1]};
        SortStats ss1;
        int d1 = 0, d2 = 0;
        PSS_TRY(radix_sort_pairs(ctx, K, V, S, 64, 0xffu, nullptr, 0, buf.sort_work, &d1, false, &ss1));
        hipLaunchKernelGGL(ss_gather_hi_kernel, dim3((S + 255) / 256), dim3(256), 0, s, E0, (const u32 *)V[d1], S, K[d1]);
        const int hi_bits = std::min(64, std::max(1, g.key_bits + g.ib - 64));
        PSS_TRY(radix_sort_pairs(ctx, K, V, S, hi_bits, 0xffu, nullptr, d1, buf.sort_work, &d2, false, &ss1));
        hipLaunchKernelGGL(ss_gather_elems_kernel, dim3((S + 255) / 256), dim3(256), 0, s, E0, (const u32 *)V[d2], S, E);
    }
    if (buf.sample_keep) PSS_HIP(hipMemcpyAsync(buf.sample_keep, E, (size_t)S * 16, hipMemcpyDeviceToDevice, s));
    }
    PSS_TRY(mark());                                                                       // [1]

    SsArgs a;
    memset(&a, 0, sizeof a);
    a.text = tx;
    a.sample = E;
    a.B1 = g.B1;
    a.B2 = g.B2;
    a.spb = S / g.B1;
    a.st2 = g.os;
    a.zl = (u32)std::max(0, std::min(64 - SS_CELL_BITS, 128 - g.key_bits - g.ib));
    const u32 num_tiles = (u32)(((u64)n + SS_DTILE1 - 1) / SS_DTILE1);
    a.tiles_per_range1 = (num_tiles + MSD_G1_RANGES - 1) / MSD_G1_RANGES;
    a.num_ranges1 = (num_tiles + a.tiles_per_range1 - 1) / a.tiles_per_range1;
    a.T = T;
    a.J1 = J1;
    a.ranges2 = ranges2;
    a.counters = counters;
    a.digits = buf.digits;
    PSS_HIP(hipMemsetAsync(counters, 0, 64, s));
    // ---- G1: text -> A[0] by the first-level splitters ----
    hipLaunchKernelGGL(ss_digits1_kernel, dim3(a.num_ranges1), dim3(SS_DBLOCK), 0, s, a);
    hipLaunchKernelGGL(msd_offsets1_kernel, dim3(MSD_BINS), dim3(256), 0, s, T, a.num_ranges1, seg_first);
    hipLaunchKernelGGL(msd_offsets1b_kernel, dim3(1), dim3(MSD_BINS), 0, s, (const u32 *)seg_first, J1, n);
    a.out = static_cast<E16 *>(buf.A[0]);
    hipLaunchKernelGGL(ss_scatter_kernel<true>, dim3(a.num_ranges1), dim3(SS_SBLOCK), 0, s, a);
    PSS_TRY(mark());                                                                       // [2]
    // ---- G2: A[0] -> A[1], every first-level bucket by its own splitters ----
    hipLaunchKernelGGL(msd_ranges_kernel, dim3(1), dim3(MSD_BINS), 0, s, (const u32 *)J1, ranges2, seg_first, counters);
    a.in = static_cast<const E16 *>(buf.A[0]);
    a.out = static_cast<E16 *>(buf.A[1]);
    hipLaunchKernelGGL(ss_digits2_kernel, dim3((u32)max_ranges2), dim3(SS_DBLOCK), 0, s, a);
    hipLaunchKernelGGL(msd_offsets_kernel, dim3(MSD_BINS), dim3(MSD_BINS), 0, s, T, (const u32 *)seg_first, (const u32 *)J1, J,
                       0u, n, MSD_BINS);
    PSS_TRY(device_excl_scan(ctx, InNonEmpty{J}, nbk, partial, d_total, ranks));
    hipLaunchKernelGGL(msd_compact_kernel, dim3(1024), dim3(256), 0, s, J, (u32)nbk, ranks, cstart, counters, (u32 *)nullptr);
    PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipMemcpyAsync(h_small + 2, counters, 16, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipStreamSynchronize(s));
    const u32 ne = h_small[0];
    const u32 maxb = h_small[3];
    if (stats) {
        stats->buckets = ne;
        stats->max_bucket = maxb;
        stats->b1 = g.B1;
        stats->b2 = g.B2;
        stats->samples = S;
        stats->key_chars = g.kc;
    }
    if (maxb > SS_MAX_BUCKET && knob("PSS_SS_DEBUG")) {
        // diagnostic: where did the crowded bucket come from?  (first-level bucket sizes, the largest joint buckets)
        std::vector<u32> hj1(MSD_BINS + 1), hj((size_t)nbk + 1);
        (void)hipMemcpy(hj1.data(), J1, (MSD_BINS + 1) * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hj.data(), J, ((size_t)nbk + 1) * 4, hipMemcpyDeviceToHost);
        u32 m1 = 0, a1 = 0;
        for (u32 i = 0; i < g.B1; ++i) { const u32 c = hj1[i + 1] - hj1[i]; if (c > m1) { m1 = c; a1 = i; } }
        fprintf(stderr, "[pss] ss declined: n=%u S=%u B1=%u B2=%u kc=%d ib=%d | largest first-level bucket %u (#%u, average %u) | joint buckets above the cap:", n, S,
                g.B1, g.B2, g.kc, g.ib, m1, a1, n / g.B1);
        int shown = 0;
        for (u64 i = 0; i < nbk && shown < 12; ++i) {
            const u32 c = hj[i + 1] - hj[i];
            if (c > SS_MAX_BUCKET) { fprintf(stderr, " #%llu:%u@%u", (unsigned long long)i, c, hj[i]); ++shown; }
        }
        fprintf(stderr, "\n");
    }
    if (maxb > SS_MAX_BUCKET) return PSS_OK;      // (a sampling accident, probability ~1e-10 per bucket: the caller falls back)
    hipLaunchKernelGGL(ss_scatter_kernel<false>, dim3((u32)max_ranges2), dim3(SS_SBLOCK), 0, s, a);
    PSS_TRY(mark());                                                                       // [3]
    // bucket-by-bucket local sort (ss_local_seg_kernel): the plan counts every bucket's length rounded up to eight
    const char *seg_env = knob("PSS_SS_SEG");
    const bool seg = !knob("PSS_SS_WINDOW_PLAN") && !(seg_env && atoi(seg_env) == 0);
    u32 *pstart = nullptr;
    if (knob("PSS_SS_WINDOW_PLAN")) {
        const TilePlan tp{SS_WIN, SS_TILE_CAP, nullptr};
        PSS_TRY(device_excl_scan(ctx, InTileHead{cstart, ne, n, tp}, ne, partial, d_total, ranks));
        hipLaunchKernelGGL(msd_tiles_kernel, dim3(1024), dim3(256), 0, s, cstart, ne, n, ranks, (const u64 *)d_total, tile_first, tp);
    } else {
        // greedy plan (ss_sort_impl.h); its tables live in the first element buffer, which the second scatter has read
        const size_t row = (size_t)ne + 1;
        u32 *pl = reinterpret_cast<u32 *>(buf.A[0]);
        const u32 *starts = cstart;
        u32 total = n, cap = SS_TILE_CAP;
        if (seg) {
            PSS_TRY(device_excl_scan(ctx, InPad8{cstart, ne, n}, ne, partial, d_total, ranks));
            PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
            pstart = pl;                                                  // the first row of the tables; read by the local sort too
            hipLaunchKernelGGL(ss_pad_starts_kernel, dim3((u32)((row + 255) / 256)), dim3(256), 0, s, (const u64 *)ranks, (const u64 *)d_total, ne,
                               pstart);
            PSS_HIP(hipStreamSynchronize(s));
            total = h_small[0];
            starts = pstart;
            cap = SS_TILE;
            pl += round_up(row, 64);
        }
        const u32 nseg = (u32)(((u64)total + SS_PLAN_SEG - 1) / SS_PLAN_SEG);
        u32 levels = 0;
        while ((1u << levels) < nseg) ++levels;
        u32 *nxt = pl, *heads = pl + row, *jumps = pl + 2 * row;          // jumps: `levels` rows (at least one)
        const u32 gb = (u32)((row + 255) / 256);
        PSS_HIP(hipMemsetAsync(heads, 0, row * 4, s));
        hipLaunchKernelGGL(ss_plan_next_kernel, dim3(gb), dim3(256), 0, s, starts, ne, total, cap, nxt);
        hipLaunchKernelGGL(ss_plan_exit_kernel, dim3(gb), dim3(256), 0, s, starts, ne, (const u32 *)nxt, jumps);
        for (u32 i = 1; i < levels; ++i)
            hipLaunchKernelGGL(ss_plan_double_kernel, dim3(gb), dim3(256), 0, s, (const u32 *)(jumps + (size_t)(i - 1) * row), (u32)row,
                               jumps + (size_t)i * row);
        hipLaunchKernelGGL(ss_plan_mark_kernel, dim3((nseg + 255) / 256), dim3(256), 0, s, starts, ne, (const u32 *)nxt,
                           (const u32 *)jumps, levels, nseg, heads);
        PSS_TRY(device_excl_scan(ctx, InU32{heads}, ne, partial, d_total, ranks));
        hipLaunchKernelGGL(ss_plan_tiles_kernel, dim3(1024), dim3(256), 0, s, (const u32 *)heads, ne, (const u64 *)ranks,
                           (const u64 *)d_total, tile_first);
    }
    PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipStreamSynchronize(s));
    const u32 nt = h_small[0];
    if ((size_t)nt > max_tiles) {
        set_error("ss_suffix_sort: %u tiles planned, tables hold %zu (internal error)", nt, max_tiles);
        return PSS_EDEVICE;
    }
    hipLaunchKernelGGL(msd_tile_desc_kernel, dim3((nt + 255) / 256), dim3(256), 0, s, cstart, tile_first, nt, ne, n, tiles_all,
                       (const u32 *)nullptr, (const u64 *)nullptr, (const u64 *)nullptr);
    // Ties go out as flags (bit 31 = "same key as my predecessor", the contract of suffix_sort_flags), not as the records
    // of the fused first rerank: groups of equal keys cross tile boundaries here, and the flag of a tile's first element
    // takes a look at the tile before it (ss_boundary_kernel, once every tile is sorted).
    (void)active;
    (void)blk_cnt;
    (void)dst_off;
    PSS_TRY(mark());                                                                       // [4]
    if (pstart)
        hipLaunchKernelGGL(ss_local_seg_kernel, dim3(nt), dim3(SL_BLOCK), 0, s, (const E16 *)buf.A[1], (const MsdTile *)tiles_all, nt, g.ib,
                           (const u32 *)cstart, (const u32 *)pstart, (const u32 *)tile_first, ne, n, sa_out);
    else
        hipLaunchKernelGGL(ss_local_kernel, dim3(nt), dim3(SL_BLOCK), 0, s, (const E16 *)buf.A[1], (const MsdTile *)tiles_all, nt, g.ib,
                           sa_out);
    PSS_TRY(mark());                                                                       // [5]
    hipLaunchKernelGGL(ss_boundary_kernel, dim3((nt + 255) / 256), dim3(256), 0, s, (const MsdTile *)tiles_all, nt, tx, sa_out);
    PSS_HIP(hipGetLastError());
    if (stats) stats->tiles = nt;
    if (profile && nev >= 6 && stats) {
        PSS_HIP(hipStreamSynchronize(s));
        float ms = 0.f;
        PSS_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
        stats->ms_sample = ms;
        PSS_HIP(hipEventElapsedTime(&ms, ev[1], ev[2]));
        stats->ms_g1 = ms;
        PSS_HIP(hipEventElapsedTime(&ms, ev[2], ev[3]));
        stats->ms_g2 = ms;
        PSS_HIP(hipEventElapsedTime(&ms, ev[4], ev[5]));
        stats->ms_local = ms;
    }
    *accepted = true;
    return PSS_OK;
}

}  // namespace pss
