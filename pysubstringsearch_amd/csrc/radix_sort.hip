// radix_sort.hip -- LSD radix sort of (u64 key, u32 value) pairs for gfx950.
//
// One pass = three launches over R contiguous "ranges" of the input
// (R <= 1024 workgroups, a multiple of 8, one range per workgroup):
//
//   rs_hist     every workgroup counts the 256 digit values of its range in LDS
//               -> table[digit][range], totals[digit]          (reads 8 B/elem)
//   rs_scan     256 workgroups turn table[][] into exclusive global offsets
//   rs_scatter  every workgroup walks its range tile by tile (4096 pairs):
//                 * coalesced wave-striped loads (8 B keys, 4 B values)
//                 * per-wave digit ranking with 64-bit ballots: four 4-way
//                   (2-bit) bucket refinements give the lanes sharing a digit,
//                   v_mbcnt gives the lane's rank among them, a per-wave LDS
//                   counter row gives the running count  -> stable rank
//                 * workgroup prefix sum over the 4x256 wave counters
//                 * keys, then values, are staged through LDS in digit order
//                   so each wave writes contiguous runs per digit bucket
//               (reads 12 B/elem, writes 12 B/elem)
//
// No inter-workgroup communication happens inside a launch, so nothing depends
// on dispatch order or XCD placement; the range->workgroup map is XCD-aware
// only for L2 write-combining (prims.h: xcd_range_of_block).
//
// A sort whose source is the text never materialises its keys: the first pass
// packs them on the fly from the recoded text (TextKeys), 1 B/elem read instead
// of 12.
//
// The initial sort of ALL suffixes has its own scatter family further down
// (fs_scatter_kernel<KIN, KOUT>, suffix_sort_flags): same tiling and ranking, but
// every pass drops the digit it consumed and carries a tie bit instead, so the
// key plane shrinks from 8 to 4 to 0 bytes per element.
//
// HBM-bound integer work: no MFMA anywhere by design.
#include "prims.h"
#include "radix_sort.h"
#include "text_keys.h"

namespace pss {

#ifndef PSS_RS_BLOCK
#define PSS_RS_BLOCK 256
#endif
#ifndef PSS_RS_IPT
#define PSS_RS_IPT 16
#endif
#ifndef PSS_RS_MINWAVES
#define PSS_RS_MINWAVES 1
#endif
constexpr int RS_BLOCK = PSS_RS_BLOCK;
constexpr int RS_WAVES = RS_BLOCK / kWave;
constexpr int RS_IPT = PSS_RS_IPT;
constexpr int RS_TILE = RS_BLOCK * RS_IPT;   // 4096 pairs per tile
static_assert(RS_IPT == TK_IPT, "text tiles pack 16 suffixes per thread");
#ifndef PSS_RS_MAX_RANGES
#define PSS_RS_MAX_RANGES 1024
#endif
constexpr u32 RS_MAX_RANGES = PSS_RS_MAX_RANGES;   // <= 1024 (one thread per range in the scan / fix kernels)

struct PassArgs {
    const u64 *kin;
    const u32 *vin;
    u64 *kout;
    u32 *vout;
    const u8 *codes;
    int code_bits;
    int key_chars;
    int plus_one;
    int key_drop;   // text pass: sort key = packed key >> key_drop
    u32 n;
    u32 num_tiles;
    u32 tiles_per_range;
    u32 num_ranges;
    int shift;
    u32 *table;    // [256][num_ranges]
    u32 *totals;   // [256]
    // flag-carrying passes of the initial suffix sort (fs_*): 4-byte key planes and the tables
    // from which fs_fix settles the one element per (range, digit) whose predecessor lives in
    // another range
    const u32 *kin32;
    u32 *kout32;
    u32 *has;         // [num_ranges][256] the range holds the digit
    u32 *first_z;     // [num_ranges][256] group number of its first / last element of the digit
    u32 *last_z;
    u32 *zeros;       // [num_ranges] group heads seen in the range
};

// Text sorted by its lower digits has clustered next digits (neighbours share the following symbols
// too), and dozens of lanes adding to ONE LDS counter serialise.  The lanes therefore spread their
// adds over HIST_COPIES histograms (`words` 2^29: 79.7 -> 76.2 ms with 4, no gain beyond; random
// digits unchanged; a match-any pre-aggregation costs more than the conflicts it removes).
#ifndef PSS_HIST_COPIES
#define PSS_HIST_COPIES 4
#endif
constexpr int HIST_COPIES = PSS_HIST_COPIES;                 // lanes spread their adds over this many histograms
constexpr int HIST_STRIDE = HIST_COPIES > 1 ? 257 : 256;     // odd stride: the copies of one digit sit in different banks
__device__ __forceinline__ u32 hist_total(const u32 *h, u32 d)
{
    u32 c = 0;
#pragma unroll
    for (int k = 0; k < HIST_COPIES; ++k) c += h[k * HIST_STRIDE + d];
    return c;
}
__device__ __forceinline__ void hist_add(u32 *h, u32 d, bool valid)
{
    // wave-uniform digit (sorted or low-entropy input): one LDS add per wave
    const u32 d0 = __builtin_amdgcn_readfirstlane(d);
    const u64 vm = __ballot(valid);
    const u64 same = __ballot(valid && d == d0);
    if (vm != 0 && same == vm) {
        if (mbcnt(vm) == 0 && valid) atomicAdd(&h[d0], (u32)__popcll(vm));
    } else if (valid) {
        atomicAdd(&h[(HIST_COPIES > 1 ? (threadIdx.x % HIST_COPIES) * HIST_STRIDE : 0) + d], 1u);
    }
}

template <bool FROM_TEXT>
__global__ __launch_bounds__(RS_BLOCK) void rs_hist_kernel(PassArgs a)
{
    __shared__ u32 h[HIST_COPIES * HIST_STRIDE];
    const u32 tid = threadIdx.x;
    const u32 g = blockIdx.x;
    for (u32 i = tid; i < (u32)(HIST_COPIES * HIST_STRIDE); i += RS_BLOCK) h[i] = 0;
    __syncthreads();
    const u32 tile0 = g * a.tiles_per_range;
    const u32 tile1 = min(tile0 + a.tiles_per_range, a.num_tiles);
    for (u32 tile = tile0; tile < tile1; ++tile) {
        const u32 base = tile * RS_TILE;
        if (FROM_TEXT) {
            const u32 i0 = base + tid * RS_IPT;
            u64 key[RS_IPT] = {};
            if (i0 < a.n) text_keys16(a.codes, i0, a.code_bits, a.key_chars, a.plus_one, a.key_drop, a.n, key);
#pragma unroll
            for (int r = 0; r < RS_IPT; ++r) {
                const bool valid = (i0 + r) < a.n;
                hist_add(h, valid ? (u32)(key[r] >> a.shift) & 0xffu : 0u, valid);
            }
        } else {
#pragma unroll
            for (int r = 0; r < RS_IPT / 2; ++r) {
                const u32 i = base + r * (2 * RS_BLOCK) + tid * 2;
                u64 k0 = 0, k1 = 0;
                if (i + 1 < a.n) {
                    const ulonglong2 kk = *reinterpret_cast<const ulonglong2 *>(a.kin + i);
                    k0 = kk.x;
                    k1 = kk.y;
                } else if (i < a.n) {
                    k0 = a.kin[i];
                }
                hist_add(h, (u32)(k0 >> a.shift) & 0xffu, i < a.n);
                hist_add(h, (u32)(k1 >> a.shift) & 0xffu, i + 1 < a.n);
            }
        }
    }
    __syncthreads();
    if (tid < 256) {
        const u32 c = hist_total(h, tid);
        a.table[tid * a.num_ranges + g] = c;
        if (c) atomicAdd(&a.totals[tid], c);
    }
}

// table[d][g] -> exclusive offset of (digit d, range g) in (d-major, g-minor) order.
__global__ __launch_bounds__(256) void rs_scan_kernel(u32 *table, const u32 *totals, u32 num_ranges)
{
    __shared__ u32 scr[4 + 1];
    const u32 tid = threadIdx.x, d = blockIdx.x;
    u32 base = 0;
    (void)block_excl_sum<4>(tid < d ? totals[tid] : 0u, scr, &base);
    const u32 per = (num_ranges + 255) / 256;
    u32 *row = table + (size_t)d * num_ranges;
    const u32 i0 = tid * per, i1 = min(i0 + per, num_ranges);
    u32 local = 0;
    for (u32 i = i0; i < i1; ++i) local += row[i];
    u32 run = base + block_excl_sum<4>(local, scr, nullptr);
    for (u32 i = i0; i < i1; ++i) {
        const u32 v = row[i];
        row[i] = run;
        run += v;
    }
}

// One tile of the scatter pass.  FULL = every slot of the tile holds an element
// (all but the last tile of the input): the per-item bounds predicates fold away.
template <bool FROM_TEXT, bool FULL, int IPT>
__device__ __forceinline__ void scatter_tile(const PassArgs &a, u32 base, u32 valid_count, u64 *exch,
                                             u32 (*wave_hist)[256], u32 *s_off, u32 *s_delta, u32 *s_scr)
{
    const u32 tid = threadIdx.x;
    const u32 lane = tid & 63u, wave = tid >> 6;
    u64 key[IPT] = {};
    u32 val[IPT];
    u32 rank[IPT];
    // position of item r inside the tile
    auto pos_of = [&](int r) -> u32 {
        return FROM_TEXT ? tid * IPT + r : wave * (kWave * IPT) + r * kWave + lane;
    };
    auto is_valid = [&](int r) -> bool { return FULL || pos_of(r) < valid_count; };
    if (FROM_TEXT) {
        const u32 i0 = base + tid * IPT;
        if (FULL || i0 < a.n) {
            static_assert(!FROM_TEXT || IPT == RS_IPT, "text tiles are 16 items per thread");
            u64 (&k16)[RS_IPT] = reinterpret_cast<u64 (&)[RS_IPT]>(key);
            text_keys16(a.codes, i0, a.code_bits, a.key_chars, a.plus_one, a.key_drop, a.n, k16);
        }
#pragma unroll
        for (int r = 0; r < IPT; ++r) val[r] = i0 + r;
    } else {
#pragma unroll
        for (int r = 0; r < IPT; ++r) key[r] = is_valid(r) ? a.kin[base + pos_of(r)] : 0;
    }
    for (u32 i = tid; i < RS_WAVES * 256; i += RS_BLOCK) (&wave_hist[0][0])[i] = 0;
    __syncthreads();

    // ---- per-wave stable ranking ----
    // Round r ranks item r of every lane.  Lanes sharing a digit are found
    // with ballots (match_digit8); the lowest of them adds the group size
    // to the wave's LDS counter with a returning atomic (ds_add_rtn), so the
    // 16 rounds pipeline instead of waiting on a load-modify-store each.
    u32 prev[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const bool valid = is_valid(r);
        const u32 d = (u32)(key[r] >> a.shift) & 0xffu;
        const u64 peers = match_digit8(d, FULL ? ~0ull : __ballot(valid));
        const u32 below = mbcnt(peers);
        prev[r] = 0;
        if (valid && below == 0) prev[r] = atomicAdd(&wave_hist[wave][d], (u32)__popcll(peers));
        const u32 leader = valid ? (u32)__builtin_ctzll(peers) : lane;
        rank[r] = below | (leader << 16);
    }
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const u32 p = __shfl(prev[r], (int)(rank[r] >> 16));
        rank[r] = p + (rank[r] & 0xffffu);
    }
    __syncthreads();

    // ---- workgroup prefix over digits (thread d < 256 owns digit d) ----
    {
        u32 c[RS_WAVES];
        u32 total = 0;
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) {
                c[w] = wave_hist[w][tid];
                total += c[w];
            }
        }
        const u32 dstart = block_excl_sum<RS_WAVES>(total, s_scr, nullptr);
        if (tid < 256) {
            u32 run = dstart;
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) {
                wave_hist[w][tid] = run;
                run += c[w];
            }
            const u32 off = s_off[tid];
            s_delta[tid] = off - dstart;
            s_off[tid] = off + total;
        }
    }
    __syncthreads();

    // ---- keys through LDS in digit order, then out in contiguous runs ----
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const u32 d = (u32)(key[r] >> a.shift) & 0xffu;
        const u32 lp = wave_hist[wave][d] + rank[r];
        rank[r] = lp;
        if (is_valid(r)) exch[lp] = key[r];
    }
    __syncthreads();
    u32 gpos[IPT];
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const u32 p = i * RS_BLOCK + tid;
        gpos[i] = 0;
        if (FULL || p < valid_count) {
            const u64 k = exch[p];
            const u32 d = (u32)(k >> a.shift) & 0xffu;
            gpos[i] = s_delta[d] + p;
#ifdef PSS_EXPERIMENT_SEQ_STORES
            gpos[i] = base + p;      // timing experiment only (wrong output): same work, sequential stores
#endif
            a.kout[gpos[i]] = k;
        }
    }
    __syncthreads();
#ifdef PSS_EXPERIMENT_NO_VALUES
    return;                          // timing experiment only (wrong output): the key plane alone
#endif
    u32 *exv = reinterpret_cast<u32 *>(exch);
    if (!FROM_TEXT) {
#pragma unroll
        for (int r = 0; r < IPT; ++r) val[r] = is_valid(r) ? a.vin[base + pos_of(r)] : 0;
    }
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        if (is_valid(r)) exv[rank[r]] = val[r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const u32 p = i * RS_BLOCK + tid;
        if (FULL || p < valid_count) a.vout[gpos[i]] = exv[p];
    }
    __syncthreads();
}

#ifndef PSS_RS_PAIR_IPT
#define PSS_RS_PAIR_IPT 16
#endif
constexpr int RS_PAIR_IPT = PSS_RS_PAIR_IPT;   // items per thread of the (key, value) scatter; divides RS_IPT

template <bool FROM_TEXT>
__global__ __launch_bounds__(RS_BLOCK, PSS_RS_MINWAVES) void rs_scatter_kernel(PassArgs a)
{
    constexpr int IPT = FROM_TEXT ? RS_IPT : RS_PAIR_IPT;
    constexpr u32 SUB = RS_BLOCK * IPT;                               // elements per sub-tile
    __shared__ __attribute__((aligned(16))) u64 exch[SUB];            // reused for values
    __shared__ u32 wave_hist[RS_WAVES][256];
    __shared__ u32 s_off[256];     // running global offset of each digit for this range
    __shared__ u32 s_delta[256];   // s_off - (start of the digit inside the tile)
    __shared__ u32 s_scr[RS_WAVES + 1];

    const u32 tid = threadIdx.x;
    const u32 g = xcd_range_of_block(blockIdx.x, gridDim.x);
    if (tid < 256) s_off[tid] = a.table[tid * a.num_ranges + g];

    const u32 tile0 = g * a.tiles_per_range;
    const u32 tile1 = min(tile0 + a.tiles_per_range, a.num_tiles);
    if (tile0 >= tile1) return;
    const u32 e0 = tile0 * (u32)RS_TILE;                       // n < 2^31: no overflow
    const u32 e1_full = tile1 * (u32)RS_TILE;
    const u32 e1 = e1_full < a.n ? e1_full : a.n;
    for (u32 base = e0; base < e1; base += SUB) {
        const u32 left = e1 - base;
        const u32 valid_count = left < SUB ? left : SUB;
        if (valid_count == SUB)
            scatter_tile<FROM_TEXT, true, IPT>(a, base, valid_count, exch, wave_hist, s_off, s_delta, s_scr);
        else
            scatter_tile<FROM_TEXT, false, IPT>(a, base, valid_count, exch, wave_hist, s_off, s_delta, s_scr);
    }
}

// =====================================================================================
// Initial suffix sort: flag-carrying passes with shrinking keys (fs_*)
//
// An LSD pass never looks at a digit again once it has consumed it, but the suffix sort
// must end with "is my full key equal to my predecessor's?" for every element.  Instead
// of carrying the consumed digits along for that one comparison, every pass carries ONE
// BIT per element (bit 31 of the value): t_p(x) = "the digits consumed so far, L_p(x),
// equal those of my predecessor in the current order" (the order is sorted by L_p, so
// equal L_p are neighbours).  With Z(x) = number of elements up to and including x whose
// bit is clear (the dense rank of L_p(x)), two elements have L_p(x) == L_p(y) iff
// Z(x) == Z(y).  The next pass moves x behind y = its predecessor inside its digit bucket,
// and t_{p+1}(x) = [Z(x) == Z(y)].  Z is a prefix count in tile order: ballots inside the
// tile, a running carry across the tiles of a range; the one element per (range, digit)
// whose predecessor lives in another range is settled by an epilogue from per-range tables
// (first / last Z per digit, heads per range).
//
// So a pass stores only the digits it has not consumed yet: the key plane is 8 bytes while
// more than 32 bits remain, 4 bytes afterwards, and absent in the last pass.  `lines`
// (40-bit key): 1 B -> 4+4 B, three passes of 4+4 -> 4+4 B (the last writes 0+4), instead
// of 8+4 B in and out everywhere.
// =====================================================================================

// digit histogram of a 4-byte key plane (low 8 bits), 16 items per thread
__global__ __launch_bounds__(RS_BLOCK) void fs_hist32_kernel(PassArgs a)
{
    __shared__ u32 h[HIST_COPIES * HIST_STRIDE];
    const u32 tid = threadIdx.x;
    const u32 g = blockIdx.x;
    for (u32 i = tid; i < (u32)(HIST_COPIES * HIST_STRIDE); i += RS_BLOCK) h[i] = 0;
    __syncthreads();
    const u32 tile0 = g * a.tiles_per_range;
    const u32 tile1 = min(tile0 + a.tiles_per_range, a.num_tiles);
    for (u32 tile = tile0; tile < tile1; ++tile) {
        const u32 base = tile * RS_TILE;
#pragma unroll
        for (int r = 0; r < RS_IPT / 4; ++r) {
            const u32 i = base + r * (4 * RS_BLOCK) + tid * 4;
            u32 k[4] = {0, 0, 0, 0};
            if (i + 3 < a.n) {
                const uint4 q = *reinterpret_cast<const uint4 *>(a.kin32 + i);
                k[0] = q.x; k[1] = q.y; k[2] = q.z; k[3] = q.w;
            } else {
                for (int c = 0; c < 4; ++c)
                    if (i + c < a.n) k[c] = a.kin32[i + c];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) hist_add(h, k[c] & 0xffu, i + c < a.n);
        }
    }
    __syncthreads();
    if (tid < 256) {
        const u32 c = hist_total(h, tid);
        a.table[tid * a.num_ranges + g] = c;
        if (c) atomicAdd(&a.totals[tid], c);
    }
}

// digit histogram of the text pass: only the symbols that reach into bits [drop, drop + 8) of the
// packed key are assembled (a 32-bit sliding window instead of the full 64-bit key of text_keys16)
__global__ __launch_bounds__(RS_BLOCK) void fs_hist_text_kernel(PassArgs a)
{
    __shared__ u32 h[HIST_COPIES * HIST_STRIDE];
    const u32 tid = threadIdx.x;
    const u32 g = blockIdx.x;
    for (u32 i = tid; i < (u32)(HIST_COPIES * HIST_STRIDE); i += RS_BLOCK) h[i] = 0;
    __syncthreads();
    const int b = a.code_bits, k = a.key_chars;
    int nch = (a.key_drop + 8 + b - 1) / b;          // symbols covering the digit (<= 26 bits of window)
    if (nch > k) nch = k;
    const u32 off = (u32)(k - nch);                  // first of them, relative to the suffix start
    const u32 wmask = (nch * b >= 32) ? ~0u : ((1u << (nch * b)) - 1u);
    const u32 tile0 = g * a.tiles_per_range;
    const u32 tile1 = min(tile0 + a.tiles_per_range, a.num_tiles);
    for (u32 tile = tile0; tile < tile1; ++tile) {
        const u32 i0 = tile * RS_TILE + tid * RS_IPT;
        if (i0 >= a.n) continue;
        const uint4 *p = reinterpret_cast<const uint4 *>(a.codes + i0);
        const uint4 lo = p[0], hi = p[1];
        const u64 q[4] = {(u64)lo.x | ((u64)lo.y << 32), (u64)lo.z | ((u64)lo.w << 32),
                          (u64)hi.x | ((u64)hi.y << 32), (u64)hi.z | ((u64)hi.w << 32)};
        auto sym = [&](u32 j) -> u32 {               // symbol at text position i0 + j, j < 32
            u32 c = (u32)(q[j >> 3] >> ((j & 7u) * 8u)) & 0xffu;
            if (a.plus_one) c = ((u64)i0 + j < a.n) ? c + 1u : 0u;
            return c;
        };
        u32 win = 0;
        for (int t = 0; t < nch; ++t) win = (win << b) | sym(off + (u32)t);
#pragma unroll
        for (int r = 0; r < RS_IPT; ++r) {
            if (r > 0) win = ((win << b) | sym(off + (u32)nch - 1u + (u32)r)) & wmask;
            hist_add(h, (win >> a.key_drop) & 0xffu, i0 + r < a.n);
        }
    }
    __syncthreads();
    if (tid < 256) {
        const u32 c = hist_total(h, tid);
        a.table[tid * a.num_ranges + g] = c;
        if (c) atomicAdd(&a.totals[tid], c);
    }
}

template <int BYTES> struct FsKey { using type = u64; };
template <> struct FsKey<4> { using type = u32; };

// One tile.  KIN = bytes of the incoming key plane (0: packed from the text), KOUT = bytes of
// the outgoing one (0: none, last pass).  zcarry = group heads seen in the earlier tiles of
// the range (uniform over the workgroup); par = tile parity (s_tz / s_tv are double-buffered so
// that the digit owners can publish this tile's last group numbers while the output loop still
// reads the previous tile's).
//
// Four workgroup barriers per tile: keys, group numbers and values pass through separate LDS
// buffers in ONE exchange, every wave clears its own counter row for the next tile, and the
// next tile's barriers order everything else (its loads and ranking touch only per-wave state).
template <int KIN, int KOUT, bool FULL>
__device__ __forceinline__ void fs_tile(const PassArgs &a, u32 base, u32 valid_count,
                                        typename FsKey<KIN == 4 ? 4 : 8>::type *exk, u32 *exz, u32 *exv,
                                        u32 (*wave_hist)[256], u32 *s_off, u32 *s_delta, u32 *s_scr, u32 g,
                                        u32 *s_dstart, u32 *s_cnt, u32 (*s_tz)[256], u32 (*s_tv)[256], u32 *s_zwave,
                                        u32 &zcarry, u32 par)
{
    constexpr bool FROM_TEXT = KIN == 0;
    constexpr int IPT = RS_IPT;
    using RK = typename FsKey<KIN == 4 ? 4 : 8>::type;
    const u32 tid = threadIdx.x;
    const u32 lane = tid & 63u, wave = tid >> 6;
    RK key[IPT] = {};
    u32 val[IPT];
    u32 zin[IPT];      // group number of the element (before the carries are added)
    u32 rank[IPT];
    auto pos_of = [&](int r) -> u32 {
        return FROM_TEXT ? tid * IPT + r : wave * (kWave * IPT) + r * kWave + lane;
    };
    auto is_valid = [&](int r) -> bool { return FULL || pos_of(r) < valid_count; };
    u32 zwave_total = 0;
    if (FROM_TEXT) {
        const u32 i0 = base + tid * IPT;
        if (FULL || i0 < a.n) {
            u64 (&k16)[RS_IPT] = reinterpret_cast<u64 (&)[RS_IPT]>(key);
            text_keys16(a.codes, i0, a.code_bits, a.key_chars, a.plus_one, a.key_drop, a.n, k16);
        }
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            val[r] = i0 + r;
            zin[r] = 0;          // nothing consumed yet: every element belongs to group 0
        }
    } else {
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const bool valid = is_valid(r);
            const u32 at = base + pos_of(r);
            if (KIN == 4) key[r] = valid ? (RK)a.kin32[at] : (RK)0;
            else key[r] = valid ? (RK)a.kin[at] : (RK)0;
            val[r] = valid ? a.vin[at] : 0x80000000u;
        }
        // Z in tile order (wave-major, then row, then lane)
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const bool head = !(val[r] >> 31);
            const u64 m = __ballot(head);
            zin[r] = zwave_total + mbcnt(m) + (head ? 1u : 0u);
            zwave_total += (u32)__popcll(m);
        }
        if (lane == 0) s_zwave[wave] = zwave_total;
    }

    // ---- per-wave stable ranking (as rs scatter_tile); the wave's counter row is zero on entry ----
    u32 prev[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const bool valid = is_valid(r);
        const u32 d = (u32)key[r] & 0xffu;
        const u64 peers = match_digit8(d, FULL ? ~0ull : __ballot(valid));
        const u32 below = mbcnt(peers);
        prev[r] = 0;
        if (valid && below == 0) prev[r] = atomicAdd(&wave_hist[wave][d], (u32)__popcll(peers));
        const u32 leader = valid ? (u32)__builtin_ctzll(peers) : lane;
        rank[r] = below | (leader << 16);
    }
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const u32 p = __shfl(prev[r], (int)(rank[r] >> 16));
        rank[r] = p + (rank[r] & 0xffffu);
    }
    __syncthreads();                                            // (A) counters and s_zwave complete

    u32 zbase = zcarry, ztile = 0;
    if (!FROM_TEXT) {
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            const u32 c = s_zwave[w];
            if (w < (int)wave) zbase += c;
            ztile += c;
        }
    }
    // ---- workgroup prefix over digits (thread d < 256 owns digit d) ----
    {
        u32 c[RS_WAVES];
        u32 total = 0;
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) {
                c[w] = wave_hist[w][tid];
                total += c[w];
            }
        }
        const u32 incl = wave_incl_sum(total);
        if (lane == kWave - 1) s_scr[wave] = incl;
        __syncthreads();                                        // (P) wave totals of the digit counts
        u32 dstart = incl - total;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w)
            if (w < (int)wave) dstart += s_scr[w];
        if (tid < 256) {
            u32 run = dstart;
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) {
                wave_hist[w][tid] = run;
                run += c[w];
            }
            const u32 off = s_off[tid];
            s_delta[tid] = off - dstart;
            s_off[tid] = off + total;
            s_dstart[tid] = dstart;
            s_cnt[tid] = total;
        }
    }
    __syncthreads();                                            // (B) digit offsets published

    // ---- keys, group numbers and values through LDS in digit order ----
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const u32 d = (u32)key[r] & 0xffu;
        const u32 lp = wave_hist[wave][d] + rank[r];
        if (is_valid(r)) {
            exk[lp] = key[r];
            if (!FROM_TEXT) exz[lp] = zbase + zin[r];
            exv[lp] = val[r] & 0x7fffffffu;
        }
    }
    __syncthreads();                                            // (C) exchange buffers complete
    // the counter row is consumed: clear it for the next tile (own wave only)
#pragma unroll
    for (int q = 0; q < 256 / kWave; ++q) wave_hist[wave][q * kWave + lane] = 0;
    // digit owners publish the last group number of their digit for the next tile
    if (tid < 256) {
        const u32 c = s_cnt[tid];
        s_tz[par ^ 1u][tid] = c ? (FROM_TEXT ? 0u : exz[s_dstart[tid] + c - 1]) : s_tz[par][tid];
        s_tv[par ^ 1u][tid] = c ? 1u : s_tv[par][tid];
    }
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const u32 p = i * RS_BLOCK + tid;
        if (FULL || p < valid_count) {
            const RK k = exk[p];
            const u32 d = (u32)k & 0xffu;
            const u32 z = FROM_TEXT ? 0u : exz[p];
            const u32 gp = s_delta[d] + p;
            const u32 ds = s_dstart[d];
            bool t;
            if (p > ds) {
                t = FROM_TEXT ? true : exz[p - 1] == z;
            } else if (s_tv[par][d]) {
                t = s_tz[par][d] == z;                         // last of this digit in an earlier tile
            } else {
                t = false;                                     // first of its digit in the whole range:
                a.first_z[(size_t)g * 256 + d] = z;            // settled by fs_fix
            }
            if (KOUT == 8) a.kout[gp] = (u64)(k >> 8);
            if (KOUT == 4) a.kout32[gp] = (u32)(k >> 8);
            a.vout[gp] = exv[p] | ((t ? 1u : 0u) << 31);
        }
    }
    zcarry += ztile;
}

template <int KIN, int KOUT>
__global__ __launch_bounds__(RS_BLOCK, PSS_RS_MINWAVES) void fs_scatter_kernel(PassArgs a)
{
    using RK = typename FsKey<KIN == 4 ? 4 : 8>::type;
    __shared__ __attribute__((aligned(16))) RK exk[RS_TILE];
    __shared__ u32 exz[KIN == 0 ? 1 : RS_TILE];
    __shared__ u32 exv[RS_TILE];
    __shared__ u32 wave_hist[RS_WAVES][256];
    __shared__ u32 s_off[256], s_delta[256], s_dstart[256], s_cnt[256], s_tz[2][256], s_tv[2][256];
    __shared__ u32 s_scr[RS_WAVES], s_zwave[RS_WAVES];

    const u32 tid = threadIdx.x;
    const u32 g = xcd_range_of_block(blockIdx.x, gridDim.x);
    if (tid < 256) {
        s_off[tid] = a.table[tid * a.num_ranges + g];
        s_tv[0][tid] = 0;
        s_tz[0][tid] = 0;
    }
    for (u32 i = tid; i < RS_WAVES * 256; i += RS_BLOCK) (&wave_hist[0][0])[i] = 0;
    const u32 tile0 = g * a.tiles_per_range;
    const u32 tile1 = min(tile0 + a.tiles_per_range, a.num_tiles);
    if (tile0 >= tile1) {
        if (tid < 256) a.has[(size_t)g * 256 + tid] = 0;
        if (tid == 0) a.zeros[g] = 0;
        return;
    }
    __syncthreads();
    const u32 e0 = tile0 * (u32)RS_TILE;
    const u32 e1_full = tile1 * (u32)RS_TILE;
    const u32 e1 = e1_full < a.n ? e1_full : a.n;
    u32 zcarry = 0, par = 0;
    for (u32 base = e0; base < e1; base += RS_TILE, par ^= 1u) {
        const u32 left = e1 - base;
        const u32 valid_count = left < (u32)RS_TILE ? left : (u32)RS_TILE;
        if (valid_count == (u32)RS_TILE)
            fs_tile<KIN, KOUT, true>(a, base, valid_count, exk, exz, exv, wave_hist, s_off, s_delta, s_scr, g, s_dstart, s_cnt,
                                     s_tz, s_tv, s_zwave, zcarry, par);
        else
            fs_tile<KIN, KOUT, false>(a, base, valid_count, exk, exz, exv, wave_hist, s_off, s_delta, s_scr, g, s_dstart,
                                      s_cnt, s_tz, s_tv, s_zwave, zcarry, par);
    }
    __syncthreads();                                            // the last tile's s_tz / s_tv [par]
    if (tid < 256) {
        a.has[(size_t)g * 256 + tid] = s_tv[par][tid];
        if (s_tv[par][tid]) a.last_z[(size_t)g * 256 + tid] = s_tz[par][tid];
    }
    if (tid == 0) a.zeros[g] = zcarry;
}

// Epilogue: the first element of digit d in range g follows, in the output, the last element of
// digit d of the nearest earlier range that had one; their group numbers are range-local, so the
// heads of the ranges in between are added (exclusive scan of zeros[]).  One workgroup per digit,
// one thread per range.
__global__ __launch_bounds__(1024) void fs_fix_kernel(const u32 *first_z, const u32 *last_z, const u32 *has,
                                                        const u32 *zeros, const u32 *table, u32 num_ranges, u32 *vout)
{
    __shared__ u32 s_wave[16], s_zw[16];
    __shared__ u32 s_base[1024];
    const u32 d = blockIdx.x, g = threadIdx.x, lane = g & 63u, wave = g >> 6;
    const bool mine = g < num_ranges && has[(size_t)g * 256 + d] != 0;
    const u32 incl = wave_incl_max(mine ? g + 1 : 0u);        // 1 + last range with digit d, up to and including g
    const u32 zc = g < num_ranges ? zeros[g] : 0u;
    const u32 zincl = wave_incl_sum(zc);
    if (lane == 63) { s_wave[wave] = incl; s_zw[wave] = zincl; }
    __syncthreads();
    u32 prev1 = __shfl_up(incl, 1);
    if (lane == 0) prev1 = 0;
    u32 zb = zincl - zc;
    for (u32 w = 0; w < wave; ++w) { prev1 = max(prev1, s_wave[w]); zb += s_zw[w]; }
    s_base[g] = zb;                                            // heads in the ranges before g
    __syncthreads();
    if (mine && prev1) {
        const u32 gp = prev1 - 1;
        if (last_z[(size_t)gp * 256 + d] + s_base[gp] == first_z[(size_t)g * 256 + d] + zb)
            vout[table[d * num_ranges + g]] |= 0x80000000u;
    }
}

// Tiny inputs (<= one tile): one workgroup, bitonic network in LDS over the
// whole 64-bit key, ties broken by value (so the order is fully determined and the
// padding never overtakes a real element).
__global__ __launch_bounds__(RS_BLOCK) void rs_small_sort_kernel(u64 *keys, u32 *vals, u32 n)
{
    __shared__ u64 sk[RS_TILE];
    __shared__ u32 sv[RS_TILE];
    const u32 tid = threadIdx.x;
    u32 np2 = 2;
    while (np2 < n) np2 <<= 1;
    for (u32 i = tid; i < np2; i += RS_BLOCK) {
        sk[i] = i < n ? keys[i] : ~0ull;
        sv[i] = i < n ? vals[i] : 0xffffffffu;
    }
    __syncthreads();
    for (u32 k = 2; k <= np2; k <<= 1) {
        for (u32 j = k >> 1; j > 0; j >>= 1) {
            for (u32 i = tid; i < np2; i += RS_BLOCK) {
                const u32 x = i ^ j;
                if (x > i) {
                    const bool up = (i & k) == 0;
                    const u64 a = sk[i], b = sk[x];
                    const u32 va = sv[i], vb = sv[x];
                    // equal keys order by value: the padding (value 0xffffffff) must stay behind real
                    // elements even when a real key is all ones (16 symbols of the largest 4-bit code)
                    if ((a > b || (a == b && va > vb)) == up) {
                        sk[i] = b;
                        sk[x] = a;
                        sv[i] = vb;
                        sv[x] = va;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (u32 i = tid; i < n; i += RS_BLOCK) {
        keys[i] = sk[i];
        vals[i] = sv[i];
    }
}

// profiling events of one sort call: destroyed on every exit path
struct EventSet {
    hipEvent_t ev[2 * 16] = {};
    int created = 0;
    ~EventSet()
    {
        for (int i = 0; i < created; ++i) (void)hipEventDestroy(ev[i]);
    }
};

size_t radix_sort_workspace_bytes()
{
    // digit table + 16 totals rows + the (has, first_z, last_z) tables and zeros[] of the fs passes
    return (size_t)256 * RS_MAX_RANGES * 4 + 256 * 4 * 16 + (size_t)256 * RS_MAX_RANGES * 12 + RS_MAX_RANGES * 4;
}

int radix_sort_pairs(DeviceCtx *ctx, uint64_t *keys[2], uint32_t *vals[2], uint32_t n, int key_bits,
                     uint32_t pass_mask, const TextKeys *text, int src, void *work, int *dst,
                     bool profile, SortStats *stats)
{
    const int passes = (key_bits + 7) / 8;
    int cur = src;
    bool from_text = text != nullptr;
    if (n == 0) {
        *dst = from_text ? 0 : src;
        return PSS_OK;
    }
    if (!from_text && n <= (u32)RS_TILE) {
        hipLaunchKernelGGL(rs_small_sort_kernel, dim3(1), dim3(RS_BLOCK), 0, ctx->stream, keys[src], vals[src], n);
        PSS_HIP(hipGetLastError());
        if (stats) stats->small_launches += 1;
        *dst = src;
        return PSS_OK;
    }
    const u32 num_tiles = (u32)(((u64)n + RS_TILE - 1) / RS_TILE);
    const u32 tpr = (num_tiles + RS_MAX_RANGES - 1) / RS_MAX_RANGES;
    u32 num_ranges = (num_tiles + tpr - 1) / tpr;
    num_ranges = (num_ranges + 7u) & ~7u;

    u32 *table = static_cast<u32 *>(work);
    u32 *totals_base = table + (size_t)256 * RS_MAX_RANGES;
    PSS_HIP(hipMemsetAsync(totals_base, 0, 256 * 4 * 16, ctx->stream));

    EventSet evs;
    hipEvent_t *ev = evs.ev;
    int ev_kind[16];   // 0 = text pass, 1 = (key, value) pass
    int nev = 0;
    int &nev_created = evs.created;
    if (profile) {   // created up front: a hipEventCreate between launch and record would idle the GPU
        for (; nev_created < 32; ++nev_created) PSS_HIP(hipEventCreate(&ev[nev_created]));
    }
    int executed = 0;
    for (int p = 0; p < passes; ++p) {
        if (!((pass_mask >> p) & 1u)) continue;
        PassArgs a;
        a.n = n;
        a.num_tiles = num_tiles;
        a.tiles_per_range = tpr;
        a.num_ranges = num_ranges;
        a.shift = p * 8;
        a.table = table;
        a.totals = totals_base + (size_t)256 * (executed & 15);
        if (executed >= 16) PSS_HIP(hipMemsetAsync(a.totals, 0, 256 * 4, ctx->stream));
        int out;
        if (from_text) {
            a.codes = text->codes;
            a.code_bits = text->code_bits;
            a.key_chars = text->key_chars;
            a.plus_one = text->plus_one;
            a.key_drop = text->drop;
            a.kin = nullptr;
            a.vin = nullptr;
            out = 0;
        } else {
            a.codes = nullptr;
            a.code_bits = a.key_chars = a.plus_one = a.key_drop = 0;
            a.kin = keys[cur];
            a.vin = vals[cur];
            out = cur ^ 1;
        }
        a.kout = keys[out];
        a.vout = vals[out];
        if (from_text) hipLaunchKernelGGL(rs_hist_kernel<true>, dim3(num_ranges), dim3(RS_BLOCK), 0, ctx->stream, a);
        else hipLaunchKernelGGL(rs_hist_kernel<false>, dim3(num_ranges), dim3(RS_BLOCK), 0, ctx->stream, a);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(256), dim3(256), 0, ctx->stream, a.table, a.totals, num_ranges);
        if (profile && nev < 32) {
            ev_kind[nev / 2] = from_text ? 0 : 1;
            PSS_HIP(hipEventRecord(ev[nev++], ctx->stream));
        }
        if (from_text) hipLaunchKernelGGL(rs_scatter_kernel<true>, dim3(num_ranges), dim3(RS_BLOCK), 0, ctx->stream, a);
        else hipLaunchKernelGGL(rs_scatter_kernel<false>, dim3(num_ranges), dim3(RS_BLOCK), 0, ctx->stream, a);
        if (profile && nev < 32) PSS_HIP(hipEventRecord(ev[nev++], ctx->stream));
        PSS_HIP(hipGetLastError());
        cur = out;
        from_text = false;
        ++executed;
        if (stats) {
            stats->launches += 1;
            stats->elems += n;
        }
    }
    if (nev) {
        PSS_HIP(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i + 1 < nev; i += 2) {
            float ms = 0.f;
            PSS_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            if (stats) {
                stats->ms += ms;
                if (ev_kind[i / 2] == 0) { stats->ms_text += ms; stats->text_launches += 1; }
                else if (ev_kind[i / 2] == 1) { stats->ms_pairs += ms; stats->pairs_launches += 1; stats->pairs_elems += n; }
            }
        }
    }
    if (executed == 0 && text != nullptr) {
        set_error("radix_sort_pairs: text source needs at least one pass");
        return PSS_EINVAL;
    }
    *dst = cur;
    return PSS_OK;
}


int suffix_sort_flags(DeviceCtx *ctx, uint64_t *keys[2], uint32_t *vals[2], uint32_t n, int key_bits,
                      const TextKeys *text, void *work, int *dst, bool profile, SortStats *stats)
{
    const int passes = (key_bits + 7) / 8;
    if (passes < 2 || text == nullptr || n == 0) {
        set_error("suffix_sort_flags: needs a text source, n > 0 and more than 8 key bits");
        return PSS_EINVAL;
    }
    const u32 num_tiles = (u32)(((u64)n + RS_TILE - 1) / RS_TILE);
    const u32 tpr = (num_tiles + RS_MAX_RANGES - 1) / RS_MAX_RANGES;
    u32 num_ranges = (num_tiles + tpr - 1) / tpr;
    num_ranges = (num_ranges + 7u) & ~7u;

    u32 *table = static_cast<u32 *>(work);
    u32 *totals_base = table + (size_t)256 * RS_MAX_RANGES;
    u32 *first_z = totals_base + 256 * 16;
    u32 *last_z = first_z + (size_t)256 * RS_MAX_RANGES;
    u32 *has = last_z + (size_t)256 * RS_MAX_RANGES;
    u32 *zeros = has + (size_t)256 * RS_MAX_RANGES;
    PSS_HIP(hipMemsetAsync(totals_base, 0, 256 * 4 * 16, ctx->stream));

    EventSet evs;
    hipEvent_t *ev = evs.ev;
    int ev_idx[16];
    int nev = 0;
    int &nev_created = evs.created;
    if (profile) {
        for (; nev_created < 2 * passes && nev_created < 32; ++nev_created) PSS_HIP(hipEventCreate(&ev[nev_created]));
    }
    int cur = 0, kin = 0;
    for (int p = 0; p < passes; ++p) {
        const int rem = key_bits - 8 * (p + 1);
        const int kout = rem <= 0 ? 0 : (rem <= 32 ? 4 : 8);
        const int out = (p == 0) ? 0 : cur ^ 1;
        PassArgs a;
        memset(&a, 0, sizeof a);
        a.n = n;
        a.num_tiles = num_tiles;
        a.tiles_per_range = tpr;
        a.num_ranges = num_ranges;
        a.shift = 0;
        a.table = table;
        a.totals = totals_base + 256 * p;
        a.first_z = first_z;
        a.last_z = last_z;
        a.has = has;
        a.zeros = zeros;
        if (p == 0) {
            a.codes = text->codes;
            a.code_bits = text->code_bits;
            a.key_chars = text->key_chars;
            a.plus_one = text->plus_one;
            a.key_drop = text->drop;
        } else {
            a.kin = keys[cur];
            a.kin32 = reinterpret_cast<const u32 *>(keys[cur]);
            a.vin = vals[cur];
        }
        a.kout = keys[out];
        a.kout32 = reinterpret_cast<u32 *>(keys[out]);
        a.vout = vals[out];
        const dim3 grid(num_ranges), block(RS_BLOCK);
        if (p == 0) hipLaunchKernelGGL(fs_hist_text_kernel, grid, block, 0, ctx->stream, a);
        else if (kin == 8) hipLaunchKernelGGL(rs_hist_kernel<false>, grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL(fs_hist32_kernel, grid, block, 0, ctx->stream, a);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(256), dim3(256), 0, ctx->stream, a.table, a.totals, num_ranges);
        if (profile && nev + 1 < nev_created) {
            ev_idx[nev / 2] = (kin / 4) * 3 + kout / 4;
            PSS_HIP(hipEventRecord(ev[nev++], ctx->stream));
        }
        if (kin == 0 && kout == 8) hipLaunchKernelGGL((fs_scatter_kernel<0, 8>), grid, block, 0, ctx->stream, a);
        else if (kin == 0 && kout == 4) hipLaunchKernelGGL((fs_scatter_kernel<0, 4>), grid, block, 0, ctx->stream, a);
        else if (kin == 8 && kout == 8) hipLaunchKernelGGL((fs_scatter_kernel<8, 8>), grid, block, 0, ctx->stream, a);
        else if (kin == 8 && kout == 4) hipLaunchKernelGGL((fs_scatter_kernel<8, 4>), grid, block, 0, ctx->stream, a);
        else if (kin == 4 && kout == 4) hipLaunchKernelGGL((fs_scatter_kernel<4, 4>), grid, block, 0, ctx->stream, a);
        else if (kin == 4 && kout == 0) hipLaunchKernelGGL((fs_scatter_kernel<4, 0>), grid, block, 0, ctx->stream, a);
        else {
            set_error("suffix_sort_flags: no kernel for key planes %d -> %d", kin, kout);
            return PSS_EINVAL;
        }
        if (profile && (nev & 1)) PSS_HIP(hipEventRecord(ev[nev++], ctx->stream));
        hipLaunchKernelGGL(fs_fix_kernel, dim3(256), dim3(1024), 0, ctx->stream, first_z, last_z, has, zeros, table,
                           num_ranges, a.vout);
        PSS_HIP(hipGetLastError());
        cur = out;
        kin = kout;
        if (stats) {
            stats->launches += 1;
            stats->elems += n;
        }
    }
    if (nev) {
        PSS_HIP(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i + 1 < nev; i += 2) {
            float ms = 0.f;
            PSS_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            if (stats) {
                stats->ms += ms;
                stats->fs_ms[ev_idx[i / 2]] += ms;
                stats->fs_launches[ev_idx[i / 2]] += 1;
                stats->fs_elems[ev_idx[i / 2]] += n;
            }
        }
    }
    *dst = cur;
    return PSS_OK;
}

}  // namespace pss
