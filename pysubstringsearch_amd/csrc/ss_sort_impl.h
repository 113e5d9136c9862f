// ss_sort_impl.h -- initial suffix sort of natural text as a SAMPLE SORT over 16-byte elements, for gfx950.
// Included by msd_sort.hip (it shares that file's partition tables, tile plan and tie gathering).
//
// The radix partition of msd_sort.hip needs every 20-bit key prefix to hold at most a tile of suffixes, which is a
// property of high-entropy text.  Natural text fails it by three orders of magnitude ("the " alone), and a
// 64-bit key (12 letters) leaves two thirds of its suffixes tied anyway: the LSD sort + text rounds that text
// took before spent 29 + 41 ms on a 512 MiB chunk of the `words` corpus.  Two changes fix both ends:
//
//   * the element is 16 bytes, [ packed key | suffix index ] read as ONE 128-bit unsigned number: 99 key bits at
//     n = 2^29 (19 five-bit symbols), and -- the index being part of the number -- no two elements are equal;
//   * the partition is by SPLITTERS, not by digits: a sorted random sample of the elements (4 per final bucket) gives
//     1023 first-level and 1023 x 1024 second-level splitters that cut the text's own distribution into 2^20 buckets
//     of 512 +- 256 elements, whatever the text looks like (equal keys included: the index breaks the tie).
//
// Pipeline (n = 2^29: B1 = B2 = 1024, S = 2^22 samples):
//   sample   S stratified random suffixes -> elements -> sorted (two chained stable 64-bit radix sorts)
//   G1       text -> digits (binary search among the first-level splitters in LDS, 10 steps of 16 bytes; the digit
//            of every suffix is kept, 2 bytes, so that the scatter pass need not search again) -> A0
//   G2       A0 -> digits inside every first-level bucket (its 1023 splitters in LDS) -> A1; the scan of the counts is
//            the table of joint bucket starts, as in msd_sort.hip
//   local    tiles of consecutive buckets (<= 4096 elements, 64 KiB of LDS) sorted by a merge sort in LDS -- eight
//            elements per thread sorted in registers, then nine rounds of pairwise merges, every thread finding its
//            eight outputs with a merge-path search: comparison based, so no distribution can crowd a bin --
//            and written out as suffix indices plus a record for every element tied with a neighbour (equal key
//            bits), exactly what msd_local_fast_kernel emits; msd_gather_kernel lines the records up.
//
// HBM traffic per suffix: (1 + 2) + (1 + 2 + 16) + (16 + 2) + (16 + 2 + 16) + (16 + 4) = 94 bytes.

constexpr int SS_MAX_CHARS = 32;                     // symbols packed into a key at most (byte window of the packers)
constexpr u32 SS_TILE = 4096;                        // elements of one workgroup tile (64 KiB of LDS)
constexpr int SS_CELL_BITS = 12;                     // ss_digits1: lookup on the top bits of the key before the search
constexpr u32 SS_CELLS = 1u << SS_CELL_BITS;
constexpr int SS_SBLOCK = 1024;                      // scatter passes: one workgroup per CU (the tile fills most of its LDS)
constexpr int SS_SIPT = SS_TILE / SS_SBLOCK;         // 4
constexpr int SS_DBLOCK = 512;                       // digit passes
constexpr u32 SS_DTILE1 = SS_DBLOCK * 16;            // G1 digits: 16 consecutive suffixes per thread
// Tile plan (msd_tile_head): the buckets that start inside one window of SS_WIN slots form a tile -- W elements on
// average, give or take what the buckets at its two ends overhang -- unless that exceeds the tile, in which case the
// window's last bucket becomes a tile of its own.  The merge sort costs per tile, not per element, so the window is as
// wide as the overflows allow: at 2^29 (buckets of 512 +- 256) 3072 -> 175 995 tiles, 3456 -> 161 659 (local sort 11.0 ->
// 10.0 ms), 3584 -> 160 905 but more of them single buckets (10.2 ms).
#ifndef PSS_SS_WIN
#define PSS_SS_WIN 3456
#endif
constexpr u32 SS_WIN = PSS_SS_WIN;
constexpr u32 SS_TILE_CAP = 4088;
constexpr u32 SS_MAX_BUCKET = 4088;

struct __attribute__((aligned(16))) E16 {
    u64 lo, hi;
};
__device__ __forceinline__ bool e16_lt(const E16 &a, const E16 &b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
__device__ __forceinline__ E16 e16_inf() { return E16{~0ull, ~0ull}; }
// c ? a : b, word by word (a select of the aggregate goes through the stack)
__device__ __forceinline__ E16 e16_sel(bool c, const E16 &a, const E16 &b) { return E16{c ? a.lo : b.lo, c ? a.hi : b.hi}; }
__device__ __forceinline__ E16 e16_load(const E16 *p)
{
    const uint4 v = *reinterpret_cast<const uint4 *>(p);
    return E16{(u64)v.x | ((u64)v.y << 32), (u64)v.z | ((u64)v.w << 32)};
}
__device__ __forceinline__ void e16_store(E16 *p, const E16 &e)
{
    *reinterpret_cast<uint4 *>(p) = make_uint4((u32)e.lo, (u32)(e.lo >> 32), (u32)e.hi, (u32)(e.hi >> 32));
}
// same key bits (everything above the index)?
__device__ __forceinline__ bool e16_same_key(const E16 &a, const E16 &b, int ib)
{
    return a.hi == b.hi && ((a.lo ^ b.lo) >> ib) == 0;
}

struct SsText {
    const u8 *codes;       // recoded text, zero padded for >= 64 bytes past n
    u32 n;
    int b, kc, plus_one;   // bits per symbol (text rounds), symbols per key, raw bytes + 1 (sigma == 256)
    int ib;                // index bits
    // The key is the number sum c_j R^(kc - 1 - j) of its symbols in base R = code values (sigma + 1, the end-of-text
    // code 0 included), not a string of b-bit fields: order preserving all the same, and denser whenever R is not a
    // power of two -- 20 symbols of a 28-letter alphabet in 99 bits where 5-bit fields hold 19.
    u32 radix;
    u64 plo, phi;          // R^(kc - 1): what the leading symbol of a key is worth (the sliding window takes it out)
};

// (hi, lo) = (hi, lo) * m + c for a small multiplier: 128-bit arithmetic on two words
__device__ __forceinline__ void mul128_add(u64 &hi, u64 &lo, u32 m, u32 c)
{
    const u64 carry = __umul64hi(lo, (u64)m);
    lo *= (u64)m;
    hi = hi * (u64)m + carry;
    const u64 nl = lo + c;
    hi += nl < lo ? 1ull : 0ull;
    lo = nl;
}

// NE consecutive suffixes whose symbols start at byte 0 of the little-endian byte stream q (NQ words; byte j of the
// stream = symbol j of the first suffix): element r = [ symbols r .. r + kc - 1 | idx0 + r ].
template <int NE, int NQ>
__device__ __forceinline__ void ss_pack(const u64 (&q)[NQ], u32 idx0, const SsText &t, E16 (&out)[NE])
{
    const int kc = t.kc;
    auto sym = [&](int j) -> u32 { return (u32)(q[j >> 3] >> ((j & 7) * 8)) & 0xffu; };
    u64 wlo = 0, whi = 0;      // the 128-bit key window
#pragma unroll
    for (int j = 0; j < SS_MAX_CHARS; ++j) {
        if (j < kc) {
            u32 c = sym(j);
            if (t.plus_one) c = ((u64)idx0 + j < t.n) ? c + 1u : 0u;
            mul128_add(whi, wlo, t.radix, c);
        }
    }
    // the stream from byte kc on (kc is uniform): symbol kc + j is byte j of s
    constexpr int NS = (NE + 6) / 8;      // words holding bytes 0 .. NE - 2
    u64 s[NS > 0 ? NS : 1];
    {
        const int qs = kc >> 3, sh = (kc & 7) * 8;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            u64 a0 = 0, a1 = 0;
#pragma unroll
            for (int c = 0; c <= SS_MAX_CHARS / 8; ++c) {
                if (c == qs) {
                    a0 = (c + i < NQ) ? q[(c + i < NQ) ? c + i : 0] : 0ull;
                    a1 = (c + i + 1 < NQ) ? q[(c + i + 1 < NQ) ? c + i + 1 : 0] : 0ull;
                }
            }
            s[i] = sh ? (a0 >> sh) | (a1 << (64 - sh)) : a0;
        }
    }
    const int ib = t.ib;
#pragma unroll
    for (int r = 0; r < NE; ++r) {
        if (r > 0) {
            const int j = r - 1;
            u32 c = (u32)(s[j >> 3] >> ((j & 7) * 8)) & 0xffu;
            if (t.plus_one) c = ((u64)idx0 + j + kc < t.n) ? c + 1u : 0u;
            // slide: take the leading symbol (symbol j of the stream) out, append the new one
            u32 lead = sym(j);
            if (t.plus_one) lead = ((u64)idx0 + j < t.n) ? lead + 1u : 0u;
            u64 lhi = t.phi, llo = t.plo;
            mul128_add(lhi, llo, lead, 0u);
            whi = whi - lhi - (wlo < llo ? 1ull : 0ull);
            wlo -= llo;
            mul128_add(whi, wlo, t.radix, c);
        }
        out[r].hi = (whi << ib) | (wlo >> (64 - ib));
        out[r].lo = (wlo << ib) | (u64)(idx0 + (u32)r);
    }
}

// The byte stream of the text from (possibly unaligned) position pos on, NQ words of it, out of three aligned 16-byte
// loads: q[0] starts at byte pos.  Needs (pos & 15) + 8 NQ <= 56 to be exact; words past that are zero-extended reads of
// the padding and never used by the callers.
template <int NQ>
__device__ __forceinline__ void ss_stream_at(const u8 *codes, u32 pos, u64 (&q)[NQ])
{
    const uint4 *p = reinterpret_cast<const uint4 *>(codes + (pos & ~15u));
    const uint4 v0 = p[0], v1 = p[1], v2 = p[2];
    u64 w[8] = {(u64)v0.x | ((u64)v0.y << 32), (u64)v0.z | ((u64)v0.w << 32), (u64)v1.x | ((u64)v1.y << 32),
                (u64)v1.z | ((u64)v1.w << 32), (u64)v2.x | ((u64)v2.y << 32), (u64)v2.z | ((u64)v2.w << 32), 0ull, 0ull};
    const bool up = (pos & 8u) != 0;
    const u32 sh = (pos & 7u) * 8u;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const u64 a0 = up ? w[i + 1 < 8 ? i + 1 : 7] : w[i < 8 ? i : 7];
        const u64 a1 = up ? w[i + 2 < 8 ? i + 2 : 7] : w[i + 1 < 8 ? i + 1 : 7];
        q[i] = sh ? (a0 >> sh) | (a1 << (64 - sh)) : a0;
    }
}

// The searches are bound by LDS bank conflicts (SQ_LDS_BANK_CONFLICT: 85 % of the LDS cycles -- every lane reads its own
// random address of the table), and the price is per read instruction, not per byte (measured: one 8-byte read per step
// instead of one 16-byte read, 4.1 -> 2.9 ms; two 8-byte reads per step, 7.9 ms).  So the first level searches the HIGH
// WORDS of the splitters alone: order preserving, and the few splitters that share the element's high word (a run of
// equal keys twelve symbols long) are settled by a walk over the full numbers.  Inside a first-level bucket high words
// -- any 64-bit window -- collide all the time (the splitters of a long run of equal keys differ in their index bits
// only): the second level reads the 16-byte numbers.
// after the high-word search left `pos` = splitters whose high word is <= the element's: step back over the splitters
// that share the high word but are larger as full numbers
// (a walk, because it is nearly always zero or one step -- but a text of long repeats gives hundreds of splitters the
// same high word, and every element of the repeat then walked over all of them: 322 ms for this kernel on a periodic
// text.  After eight steps the rest of the run of equal high words is bisected.)
__device__ __forceinline__ u32 ss_settle(const u64 *win, const E16 *spl, u32 pos, u64 w, const E16 &e)
{
    u32 steps = 0;
    while (pos > 0 && win[pos - 1] == w && e16_lt(e, spl[pos - 1])) {
        --pos;
        if (++steps == 8) {
            u32 a = 0, b = pos;                    // first splitter of the run of high words equal to w
            while (a < b) {
                const u32 m = (a + b) >> 1;
                if (win[m] < w) a = m + 1; else b = m;
            }
            b = pos;                                // first splitter in [a, pos) above e as a full number (pos: none)
            while (a < b) {
                const u32 m = (a + b) >> 1;
                if (e16_lt(e, spl[m])) b = m; else a = m + 1;
            }
            return a;
        }
    }
    return pos;
}

// ---- sample ---------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void ss_sample_kernel(SsText t, u32 S, E16 *E0, u64 *klo, u32 *vals)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    // one sample from every stratum [i n / S, (i + 1) n / S): the strata cover the WHOLE text (round 4: with a stride of
    // floor(n / S) the last n mod S positions -- 1.3 MB of a 412 MB chunk of real files -- were never sampled, a run of
    // 18 432 equal bytes there made one bucket of 18 432 elements and the sort declined)
    const u64 s0 = (u64)i * t.n / S, s1 = (u64)(i + 1) * t.n / S;
    u64 x = ((u64)i + 1) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32;
    const u32 pos = (u32)(s0 + x % (s1 - s0));
    u64 q[5];
    ss_stream_at<5>(t.codes, pos, q);      // 40 bytes >= SS_MAX_CHARS
    E16 e[1];
    ss_pack<1, 5>(q, pos, t, e);
    e16_store(&E0[i], e[0]);
    klo[i] = e[0].lo;
    vals[i] = i;
}
__global__ __launch_bounds__(256) void ss_gather_hi_kernel(const E16 *E0, const u32 *order, u32 S, u64 *khi)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S) khi[i] = E0[order[i]].hi;
}
__global__ __launch_bounds__(256) void ss_gather_elems_kernel(const E16 *E0, const u32 *order, u32 S, E16 *E)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S) e16_store(&E[i], e16_load(&E0[order[i]]));
}

// ---- digit passes ----------------------------------------------------------------------------------------------

struct SsArgs {
    SsText text;
    const E16 *sample;     // the sorted sample
    u32 B1, B2;            // buckets of the two levels (powers of two, <= 1024)
    u32 spb, st2;          // sample members per first-level bucket; stride of the second-level splitters inside them
    u32 zl;                // leading bits of an element's high word that are zero for every key (128 - key bits - index bits)
    u32 tiles_per_range1, num_ranges1;      // G1 ranges, in tiles of SS_DTILE1 suffixes
    u32 *T;
    const u32 *J1;
    const MsdRange *ranges2;
    const u32 *counters;
    u16 *digits;           // [n] digit of every element of the pass (by text position in G1, by position in A0 in G2)
    const E16 *in;
    E16 *out;
};

__global__ __launch_bounds__(SS_DBLOCK) void ss_digits1_kernel(SsArgs a)
{
    __shared__ E16 spl[MSD_BINS];
    __shared__ u64 win[MSD_BINS];
    __shared__ u32 hist[MSD_BINS];
    // cell[c] = splitters whose high word lies below c << zs: the top twelve SIGNIFICANT bits of an element's high word
    // say which few splitters it can fall between -- a lookup and a short search instead of ten steps over all of them.
    // (Round 4: the twelve bits used to be the top bits of the word; twelve symbols over 213 byte values fill 122 of the
    // 128 bits, so real files saw 64 cells instead of 4096 and this pass took 5.5 ms where `words` takes 2.4.)
    __shared__ u16 cell[SS_CELLS + 1];
    const int zs = 64 - SS_CELL_BITS - (int)a.zl;
    const u32 tid = threadIdx.x, r = blockIdx.x;
    if (r >= a.num_ranges1) return;
    for (u32 i = tid; i < MSD_BINS; i += SS_DBLOCK) {
        hist[i] = 0;
        const E16 sp = e16_sel(i + 1 < a.B1, e16_load(&a.sample[(size_t)min(i + 1, a.B1 - 1) * a.spb]), e16_inf());
        spl[i] = sp;
        win[i] = sp.hi;
    }
    __syncthreads();
    for (u32 c = tid; c <= SS_CELLS; c += SS_DBLOCK) {
        u32 lo = 0, hi = a.B1 - 1;                      // first splitter with high word >= c << 52 (all of them: B1 - 1)
        if (c < SS_CELLS) {
            const u64 v = (u64)c << zs;
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                if (win[mid] < v) lo = mid + 1; else hi = mid;
            }
        } else {
            lo = a.B1 - 1;
        }
        cell[c] = (u16)lo;
    }
    __syncthreads();
    const u32 n = a.text.n;
    const u64 e0 = (u64)r * a.tiles_per_range1 * SS_DTILE1;
    const u64 e1 = min(e0 + (u64)a.tiles_per_range1 * SS_DTILE1, (u64)n);
    for (u64 base = e0; base < e1; base += SS_DTILE1) {
        const u32 i0 = (u32)base + tid * 16;
        // (a thread past the end of the text goes through the motions on the start of the text: the wave-wide maximum
        // below needs every lane)
        const bool live = i0 < n;
        u64 q[6];
        {
            const uint4 *p = reinterpret_cast<const uint4 *>(a.text.codes + (live ? i0 : 0u));
            const uint4 v0 = p[0], v1 = p[1], v2 = p[2];
            q[0] = (u64)v0.x | ((u64)v0.y << 32); q[1] = (u64)v0.z | ((u64)v0.w << 32);
            q[2] = (u64)v1.x | ((u64)v1.y << 32); q[3] = (u64)v1.z | ((u64)v1.w << 32);
            q[4] = (u64)v2.x | ((u64)v2.y << 32); q[5] = (u64)v2.z | ((u64)v2.w << 32);
        }
        u32 dg[16];
#pragma unroll
        for (int h = 0; h < 2; ++h) {      // two halves of eight: sixteen 128-bit elements at once do not fit the registers
            u64 qh[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) qh[i] = h ? (i + 1 < 6 ? q[i + 1] : 0ull) : q[i];
            E16 e[8];
            ss_pack<8, 6>(qh, (live ? i0 : 0u) + 8 * h, a.text, e);
            // pos = splitters whose high word is <= the element's: those of the cells below (table), then an upper-bound
            // search among the splitters of the element's own cell -- as many steps as the fullest cell of the wave needs
            u32 pos[8], cnt[8], most = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const u32 c = min((u32)(e[k].hi >> zs), SS_CELLS - 1u);
                pos[k] = cell[c];
                cnt[k] = cell[c + 1] - pos[k];
                most = max(most, cnt[k]);
            }
            most = (u32)__shfl((int)wave_incl_max(most), 63);
            u32 off[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) off[k] = 0;
            for (u32 step = most ? 1u << (31 - __builtin_clz(most)) : 0u; step; step >>= 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const u32 t = off[k] + step;
                    if (t <= cnt[k] && !(e[k].hi < win[pos[k] + t - 1])) off[k] = t;
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) pos[k] += off[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                pos[k] = ss_settle(win, spl, pos[k], e[k].hi, e[k]);
                dg[8 * h + k] = pos[k];
                if (live && i0 + 8 * h + k < n) atomicAdd(&hist[pos[k]], 1u);
            }
        }
        uint4 d0, d1;
        d0.x = dg[0] | (dg[1] << 16); d0.y = dg[2] | (dg[3] << 16); d0.z = dg[4] | (dg[5] << 16); d0.w = dg[6] | (dg[7] << 16);
        d1.x = dg[8] | (dg[9] << 16); d1.y = dg[10] | (dg[11] << 16); d1.z = dg[12] | (dg[13] << 16); d1.w = dg[14] | (dg[15] << 16);
        if (live) {
            uint4 *dp = reinterpret_cast<uint4 *>(a.digits + i0);  // (the digit array is padded to a multiple of 16 entries)
            dp[0] = d0;
            dp[1] = d1;
        }
    }
    __syncthreads();
    for (u32 i = tid; i < MSD_BINS; i += SS_DBLOCK) a.T[(size_t)i * a.num_ranges1 + r] = hist[i];      // digit-major (msd_offsets1)
}

// cell of a 128-bit number relative to `base`, 2^sh numbers per cell: 0 below the base, the last cell from its start on
__device__ __forceinline__ u32 ss_cell_of(const E16 &x, const E16 &base, int sh)
{
    if (e16_lt(x, base)) return 0u;
    const u64 lo = x.lo - base.lo;
    const u64 hi = x.hi - base.hi - (x.lo < base.lo ? 1ull : 0ull);
    u64 t_lo, t_hi;
    if (sh == 0) {
        t_lo = lo;
        t_hi = hi;
    } else if (sh < 64) {
        t_lo = (lo >> sh) | (hi << (64 - sh));
        t_hi = hi >> sh;
    } else {
        t_lo = hi >> (sh - 64);
        t_hi = 0;
    }
    return (t_hi || t_lo >= SS_CELLS) ? SS_CELLS - 1u : (u32)t_lo;
}

__global__ __launch_bounds__(SS_DBLOCK) void ss_digits2_kernel(SsArgs a)
{
    __shared__ E16 spl[MSD_BINS];
    __shared__ u32 hist[MSD_BINS];
    // the same shortcut as in ss_digits1, on the key relative to the bucket: the splitters of this first-level bucket
    // span [spl[0], spl[B2 - 2]], cut into SS_CELLS cells of 2^sh numbers; cell[c] = splitters in the cells below c
    __shared__ u16 cell[SS_CELLS + 1];
    const u32 tid = threadIdx.x, r = blockIdx.x;
    if (r >= a.counters[0]) return;
    const u32 seg = a.ranges2[r].seg, e0 = a.ranges2[r].start, e1 = a.ranges2[r].end;
    for (u32 i = tid; i < MSD_BINS; i += SS_DBLOCK) {
        hist[i] = 0;
        spl[i] = e16_sel(i + 1 < a.B2, e16_load(&a.sample[(size_t)seg * a.spb + (size_t)min(i + 1, a.B2 - 1) * a.st2]), e16_inf());
    }
    __syncthreads();
    const u32 nspl = a.B2 - 1;                          // real splitters
    const E16 cbase = spl[0];
    int csh = 0;
    {
        const E16 top = spl[nspl ? nspl - 1 : 0];
        const u64 dlo = top.lo - cbase.lo, dhi = top.hi - cbase.hi - (top.lo < cbase.lo ? 1ull : 0ull);
        const int bits = dhi ? 128 - __builtin_clzll(dhi) : (dlo ? 64 - __builtin_clzll(dlo) : 0);
        csh = bits > SS_CELL_BITS ? bits - SS_CELL_BITS : 0;
    }
    for (u32 c = tid; c <= SS_CELLS; c += SS_DBLOCK) {
        u32 lo = 0, hi = nspl;                          // first splitter whose cell is >= c
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (ss_cell_of(spl[mid], cbase, csh) < c) lo = mid + 1; else hi = mid;
        }
        cell[c] = (u16)lo;
    }
    __syncthreads();
    for (u32 base = e0; base < e1; base += SS_DBLOCK * 8) {
        E16 e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u32 p = base + k * SS_DBLOCK + tid;
            e[k] = e16_sel(p < e1, e16_load(&a.in[min(p, e1 - 1)]), e16_inf());
        }
        // pos = splitters <= the element: those of the cells below its own (table), then an upper-bound search among
        // the splitters of its cell, as many steps as the fullest cell of the wave needs
        u32 pos[8], cnt[8], off[8], most = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u32 c = ss_cell_of(e[k], cbase, csh);
            pos[k] = cell[c];
            cnt[k] = cell[c + 1] - pos[k];
            off[k] = 0;
            most = max(most, cnt[k]);
        }
        most = (u32)__shfl((int)wave_incl_max(most), 63);
        for (u32 step = most ? 1u << (31 - __builtin_clz(most)) : 0u; step; step >>= 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const u32 t = off[k] + step;
                if (t <= cnt[k] && !e16_lt(e[k], spl[pos[k] + t - 1])) off[k] = t;
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) pos[k] += off[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u32 p = base + k * SS_DBLOCK + tid;
            if (p < e1) {
                a.digits[p] = (u16)pos[k];
                atomicAdd(&hist[pos[k]], 1u);
            }
        }
    }
    __syncthreads();
    for (u32 i = tid; i < MSD_BINS; i += SS_DBLOCK) a.T[(size_t)r * MSD_BINS + i] = hist[i];      // range-major (msd_offsets)
}

// ---- scatter passes --------------------------------------------------------------------------------------------

#ifndef PSS_SS_STILE
#define PSS_SS_STILE 8192
#endif
// A scatter tile is larger than what LDS can stage at once: it goes through in pieces by destination position
// (positions [0, 2048), [2048, 4096), ...), 32 KiB of staging whatever the tile size.  What the tile size buys is the
// length of the runs the output loop writes: the pass is bound by the 128-byte lines it touches (docs/history 4.3), and 8192
// elements over 1024 bins are runs of 8 x 16 bytes -- whole lines -- where 4096 gave half lines.
constexpr u32 SS_STILE = PSS_SS_STILE;
constexpr int SS_SIPT2 = SS_STILE / SS_SBLOCK;      // elements per thread and tile
constexpr u32 SS_PIECE = 2048;
constexpr int SS_PIECES = SS_STILE / SS_PIECE;

template <bool FROM_TEXT>
__global__ __launch_bounds__(SS_SBLOCK) void ss_scatter_kernel(SsArgs a)
{
    __shared__ E16 exch[SS_PIECE];
    __shared__ u16 dstage[SS_PIECE];
    __shared__ u32 hist[MSD_BINS], s_delta[MSD_BINS], s_off[MSD_BINS];
    __shared__ u16 s_start[MSD_BINS];
    __shared__ u32 scr[SS_SBLOCK / kWave + 1];
    const u32 tid = threadIdx.x, r = blockIdx.x;
    u32 e0, e1;
    if (FROM_TEXT) {
        if (r >= a.num_ranges1) return;
        const u64 s = (u64)r * a.tiles_per_range1 * SS_DTILE1;
        if (s >= a.text.n) return;
        e0 = (u32)s;
        e1 = (u32)min(s + (u64)a.tiles_per_range1 * SS_DTILE1, (u64)a.text.n);
    } else {
        if (r >= a.counters[0]) return;
        e0 = a.ranges2[r].start;
        e1 = a.ranges2[r].end;
    }
    hist[tid] = 0;      // SS_SBLOCK == MSD_BINS
    s_off[tid] = FROM_TEXT ? a.T[(size_t)tid * a.num_ranges1 + r] + a.J1[tid] : a.T[(size_t)r * MSD_BINS + tid];
    __syncthreads();
    for (u32 base = e0; base < e1; base += SS_STILE) {
        const u32 valid = min(SS_STILE, e1 - base);
        E16 e[SS_SIPT2];
        u32 dig[SS_SIPT2], lp[SS_SIPT2];
        if (FROM_TEXT) {
            // SS_SIPT2 consecutive suffixes per thread (base is a multiple of 16)
            const u32 i0 = base + tid * SS_SIPT2;
            if (tid * SS_SIPT2 < valid) {
                if (SS_SIPT2 >= 8) {
                    u64 q[6];
                    ss_stream_at<6>(a.text.codes, i0, q);
                    ss_pack<SS_SIPT2, 6>(q, i0, a.text, e);
                    const uint4 *dp = reinterpret_cast<const uint4 *>(a.digits + i0);
                    const uint4 dd = dp[0], d2 = SS_SIPT2 > 8 ? dp[1] : dd;
                    const u32 w[8] = {dd.x, dd.y, dd.z, dd.w, d2.x, d2.y, d2.z, d2.w};
#pragma unroll
                    for (int k = 0; k < SS_SIPT2; ++k) dig[k] = (w[(k >> 1) & 7] >> (16 * (k & 1))) & 0xffffu;
                } else {
                    u64 q[5];
                    ss_stream_at<5>(a.text.codes, i0, q);
                    ss_pack<SS_SIPT2, 5>(q, i0, a.text, e);
                    const u64 dd = *reinterpret_cast<const u64 *>(a.digits + i0);
#pragma unroll
                    for (int k = 0; k < SS_SIPT2; ++k) dig[k] = (u32)(dd >> (16 * (k & 3))) & 0xffffu;
                }
            }
#pragma unroll
            for (int k = 0; k < SS_SIPT2; ++k) lp[k] = (tid * SS_SIPT2 + k < valid) ? atomicAdd(&hist[dig[k]], 1u) : 0xffffffffu;
        } else {
#pragma unroll
            for (int k = 0; k < SS_SIPT2; ++k) {      // (every load of the tile in flight before the first ranking atomic)
                const u32 p = k * SS_SBLOCK + tid;
                e[k] = e16_load(&a.in[base + min(p, valid - 1)]);
                dig[k] = a.digits[base + min(p, valid - 1)];
            }
#pragma unroll
            for (int k = 0; k < SS_SIPT2; ++k) {
                const u32 p = k * SS_SBLOCK + tid;
                lp[k] = p < valid ? atomicAdd(&hist[dig[k]], 1u) : 0xffffffffu;
            }
        }
        __syncthreads();                                    // (A) counts complete; previous tile fully written out
        {
            const u32 c = hist[tid];
            const u32 ex = block_excl_sum<SS_SBLOCK / kWave>(c, scr, nullptr);
            s_start[tid] = (u16)ex;
            const u32 o = s_off[tid];
            s_delta[tid] = o - ex;
            s_off[tid] = o + c;
            hist[tid] = 0;
        }
        __syncthreads();                                    // (B) bin starts published
#pragma unroll
        for (int k = 0; k < SS_SIPT2; ++k)
            if (lp[k] != 0xffffffffu) lp[k] += (u32)s_start[dig[k]];      // rank inside the bin -> position in the tile
#pragma unroll
        for (int h = 0; h < SS_PIECES; ++h) {
            if (h * SS_PIECE >= valid) break;
            if (h) __syncthreads();                         // the piece before this one is written out
#pragma unroll
            for (int k = 0; k < SS_SIPT2; ++k) {
                const u32 q = lp[k] - h * SS_PIECE;         // (an element of another piece, or none: q >= SS_PIECE)
                if (q < SS_PIECE) {
                    e16_store(&exch[q], e[k]);
                    dstage[q] = (u16)dig[k];
                }
            }
            __syncthreads();                                // (C) the piece in bin order
#pragma unroll
            for (int k = 0; k < (int)(SS_PIECE / SS_SBLOCK); ++k) {
                const u32 q = k * SS_SBLOCK + tid, p = h * SS_PIECE + q;
                if (p < valid) e16_store(&a.out[(size_t)s_delta[dstage[q]] + p], e16_load(&exch[q]));
            }
        }
    }
}

// ---- local sort: merge sort of a tile in LDS ----------------------------------------------------------------------

#ifndef PSS_SL_BLOCK
#define PSS_SL_BLOCK 512
#endif
constexpr int SL_BLOCK = PSS_SL_BLOCK;          // threads of the sorting workgroup
constexpr int SL_IPT = SS_TILE / SL_BLOCK;      // elements every thread merges per round (4 or 8)
static_assert(SL_IPT == 4 || SL_IPT == 8, "the register sort below is written for 4 or 8 elements");

// LDS slot of tile position p: consecutive positions (one thread's run of outputs) spread over eight 16-byte
// columns, so that the threads' b128 accesses to "their" k-th element do not all land in the same banks.
__device__ __forceinline__ u32 sl_slot(u32 p) { return p ^ (((p >> 3) ^ (p >> 6)) & 7u); }

__device__ __forceinline__ void sl_cswap(E16 &a, E16 &b)
{
    const bool sw = e16_lt(b, a);
    const E16 x = e16_sel(sw, b, a), y = e16_sel(sw, a, b);
    a = x;
    b = y;
}

__global__ __launch_bounds__(SL_BLOCK, SL_BLOCK == 512 ? 4 : 8) void ss_local_kernel(const E16 *in, const MsdTile *tiles, u32 nt, int ib, u32 *sa_out)
{
    __shared__ E16 buf[SS_TILE];
    const u32 tid = threadIdx.x;
    const u32 t = blockIdx.x;
    if (t >= nt) return;
    const u32 e0 = tiles[t].e0, count = tiles[t].count;
    // coalesced load, positions past the tile = +infinity
#pragma unroll
    for (int k = 0; k < SL_IPT; ++k) {
        const u32 p = k * SL_BLOCK + tid;
        e16_store(&buf[sl_slot(p)], e16_sel(p < count, e16_load(&in[e0 + min(p, count - 1)]), e16_inf()));
    }
    __syncthreads();
    E16 v[SL_IPT];
    const u32 o0 = tid * SL_IPT;                 // this thread's run of positions
#pragma unroll
    for (int k = 0; k < SL_IPT; ++k) v[k] = e16_load(&buf[sl_slot(o0 + k)]);
    if (SL_IPT == 8) {
        // eight elements in registers: odd-even merge sort network (19 compare-exchanges)
        sl_cswap(v[0], v[1]); sl_cswap(v[2], v[3]); sl_cswap(v[4 % SL_IPT], v[5 % SL_IPT]); sl_cswap(v[6 % SL_IPT], v[7 % SL_IPT]);
        sl_cswap(v[0], v[2]); sl_cswap(v[1], v[3]); sl_cswap(v[4 % SL_IPT], v[6 % SL_IPT]); sl_cswap(v[5 % SL_IPT], v[7 % SL_IPT]);
        sl_cswap(v[1], v[2]); sl_cswap(v[5 % SL_IPT], v[6 % SL_IPT]);
        sl_cswap(v[0], v[4 % SL_IPT]); sl_cswap(v[1], v[5 % SL_IPT]); sl_cswap(v[2], v[6 % SL_IPT]); sl_cswap(v[3], v[7 % SL_IPT]);
        sl_cswap(v[2], v[4 % SL_IPT]); sl_cswap(v[3], v[5 % SL_IPT]);
        sl_cswap(v[1], v[2]); sl_cswap(v[3], v[4 % SL_IPT]); sl_cswap(v[5 % SL_IPT], v[6 % SL_IPT]);
    } else {
        sl_cswap(v[0], v[1]); sl_cswap(v[2], v[3]);
        sl_cswap(v[0], v[2]); sl_cswap(v[1], v[3]);
        sl_cswap(v[1], v[2]);
    }
#pragma unroll
    for (int k = 0; k < SL_IPT; ++k) e16_store(&buf[sl_slot(o0 + k)], v[k]);
    __syncthreads();
    // merge rounds: runs of L become runs of 2 L; thread -> outputs o0 .. o0 + SL_IPT - 1 of its pair of runs (merge path)
    for (u32 L = SL_IPT; L < SS_TILE; L <<= 1) {
        const u32 pair0 = o0 & ~(2 * L - 1);             // first position of the pair of runs
        const u32 d = o0 - pair0;                        // outputs of the pair before mine
        const u32 A = pair0, B = pair0 + L;
        // Positions from `count` on hold +infinity at every level (the elements fill a prefix of the tile, and a sorted
        // run keeps its padding at its end): a thread whose outputs all lie there has nothing to merge -- at the top
        // levels that is a quarter of the waves of an average tile (3050 of 4096 slots): 11.0 -> 10.8 ms.  (Sorting
        // tiles of <= 3072 / 3584 elements with six / seven elements per thread instead -- three instantiations of the
        // rounds in one kernel -- was slower: 11.7 ms.)
        const bool live = o0 < count;
        if (live) {
            // merge path: the smallest a in [lo, hi] with NOT A[a] < B[d - 1 - a].  The kernel is bound by VALU issue (128-bit
            // compares and selects: ~450 instructions per thread and round) with the LDS busy half of the time; measured
            // and left out, all at 2^29: a 4-ary search (three probes per step, half as many dependent steps), 12.5 vs
            // 11.0 ms; 1024 threads with four elements each, 15.2 ms; merging only inside the bucket that straddles the
            // border of two long runs (the buckets of a tile are in order already), 12.1 ms -- the skipped lanes save
            // nothing while the rounds stay barrier-synchronised and the extra predicates cost in every round.
            u32 lo = d > L ? d - L : 0, hi = d < L ? d : L;
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                const E16 x = e16_load(&buf[sl_slot(A + mid)]), y = e16_load(&buf[sl_slot(B + d - 1 - mid)]);
                if (e16_lt(x, y)) lo = mid + 1; else hi = mid;
            }
            u32 ai = lo, bi = d - lo;
            E16 va = e16_sel(ai < L, e16_load(&buf[sl_slot(A + min(ai, L - 1))]), e16_inf());
            E16 vb = e16_sel(bi < L, e16_load(&buf[sl_slot(B + min(bi, L - 1))]), e16_inf());
#pragma unroll
            for (int k = 0; k < SL_IPT; ++k) {
                const bool ta = !e16_lt(vb, va);         // take from A (both +infinity: padding, either will do)
                v[k] = e16_sel(ta, va, vb);
                ai += ta ? 1u : 0u;
                bi += ta ? 0u : 1u;
                if (k + 1 < SL_IPT) {
                    const u32 ni = ta ? ai : bi;
                    const E16 nx = e16_sel(ni < L, e16_load(&buf[sl_slot((ta ? A : B) + min(ni, L - 1))]), e16_inf());
                    va = e16_sel(ta, nx, va);
                    vb = e16_sel(ta, vb, nx);
                }
            }
        }
        __syncthreads();                                 // every read of this round is done
        if (live) {
#pragma unroll
            for (int k = 0; k < SL_IPT; ++k) e16_store(&buf[sl_slot(o0 + k)], v[k]);
        }
        __syncthreads();
    }
    // output: suffix indices, coalesced, bit 31 = "same key as my predecessor" (ties come back as flags, not as the
    // records of the MSD sort's fused first rerank: a group of equal keys can cross a tile border here -- ss_boundary)
    const u32 imask = (u32)((1ull << ib) - 1ull);
#pragma unroll
    for (int k = 0; k < SL_IPT; ++k) {
        const u32 p = k * SL_BLOCK + tid;
        if (p < count) {
            const E16 x = e16_load(&buf[sl_slot(p)]);
            const bool tie = p > 0 && e16_same_key(e16_load(&buf[sl_slot(p - 1)]), x, ib);
            const u32 sfx = (u32)x.lo & imask;
            sa_out[e0 + p] = sfx | (tie ? 0x80000000u : 0u);
        }
    }
}

// ---- local sort, bucket by bucket (round 5) --------------------------------------------------------------------------
// A tile is a run of consecutive buckets, and the buckets are in order among themselves already (the splitters cut
// [key | index] numbers: every element of a bucket is smaller than every element of the next).  Sorting the tile as one
// array of 4096 slots takes nine merge rounds whatever it holds; sorting every bucket on its own takes log2 of the
// bucket's length in runs of eight -- six rounds for the average bucket of 512 elements -- and the workgroup stops after
// the rounds its LARGEST bucket needs.  For that every bucket starts at a multiple of eight slots (the plan counts
// padded lengths: ss_pad_starts), so that a thread's eight outputs never straddle a border, the runs of a round are
// counted from the bucket's start, and the last run of a bucket is simply shorter.  Threads of a bucket that is done
// sit out the remaining rounds: the kernel is bound by VALU issue, and an idle wave leaves the issue slots to the others.
constexpr u32 SL_MAXB = SS_TILE / 8;            // buckets of a tile at most (every bucket takes >= 8 slots)

struct InPad8 {
    const u32 *cstart;
    u32 ne, n;
    __device__ u64 operator()(u64 k) const
    {
        const u32 c = ((u32)k + 1 < ne ? cstart[k + 1] : n) - cstart[k];
        return (u64)((c + 7u) & ~7u);
    }
};
// padded starts as 32-bit numbers, ne + 1 of them (n + 7 ne < 2^32 is checked by the caller)
__global__ __launch_bounds__(256) void ss_pad_starts_kernel(const u64 *scan, const u64 *total, u32 ne, u32 *pstart)
{
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < ne) pstart[k] = (u32)scan[k];
    else if (k == ne) pstart[ne] = (u32)*total;
}

__global__ __launch_bounds__(SL_BLOCK, SL_BLOCK == 512 ? 4 : 8) void ss_local_seg_kernel(const E16 *in, const MsdTile *tiles, u32 nt, int ib,
                                                                                        const u32 *cstart, const u32 *pstart,
                                                                                        const u32 *tile_first, u32 ne, u32 n, u32 *sa_out)
{
    static_assert(SL_IPT == 8, "runs of eight: the padding of the plan");
    __shared__ E16 buf[SS_TILE];
    __shared__ u32 s_bs[SL_MAXB + 1];            // element offset of bucket k inside the tile (k = nb: count)
    __shared__ u32 s_ps[SL_MAXB + 1];            // slot offset of bucket k (k = nb: slots used)
    __shared__ u32 s_pmax;
    const u32 tid = threadIdx.x;
    const u32 t = blockIdx.x;
    if (t >= nt) return;
    const u32 e0 = tiles[t].e0, count = tiles[t].count, nb = tiles[t].nb, k0 = tile_first[t];
    if (tid == 0) s_pmax = 0;
    const u32 p0 = pstart[k0];
    for (u32 k = tid; k <= nb; k += SL_BLOCK) {
        const u32 kk = k0 + k;
        s_bs[k] = (kk < ne ? cstart[kk] : n) - e0;
        s_ps[k] = pstart[kk] - p0;
    }
    __syncthreads();
    for (u32 k = tid; k < nb; k += SL_BLOCK) atomicMax(&s_pmax, s_ps[k + 1] - s_ps[k]);
    const u32 o0 = tid * SL_IPT;                 // this thread's slots, all of one bucket
    const bool mine = o0 < s_ps[nb];
    u32 b = 0;
    if (mine) {
        u32 lo = 0, hi = nb - 1;                 // the last bucket that starts at or before o0
        while (lo < hi) {
            const u32 mid = (lo + hi + 1) >> 1;
            if (s_ps[mid] <= o0) lo = mid; else hi = mid - 1;
        }
        b = lo;
    }
    const u32 S = s_ps[b], P = s_ps[b + 1] - S;  // the bucket's slots: [S, S + P), P a multiple of eight
    const u32 eb = s_bs[b], sz = s_bs[b + 1] - eb;
    const u32 rel0 = o0 - S;                     // (meaningless unless mine)
    E16 v[SL_IPT];
#pragma unroll
    for (int k = 0; k < SL_IPT; ++k) {
        const u32 idx = rel0 + k;
        const bool real = mine && idx < sz;
        v[k] = e16_sel(real, e16_load(&in[e0 + (real ? eb + idx : 0u)]), e16_inf());
    }
    sl_cswap(v[0], v[1]); sl_cswap(v[2], v[3]); sl_cswap(v[4], v[5]); sl_cswap(v[6], v[7]);
    sl_cswap(v[0], v[2]); sl_cswap(v[1], v[3]); sl_cswap(v[4], v[6]); sl_cswap(v[5], v[7]);
    sl_cswap(v[1], v[2]); sl_cswap(v[5], v[6]);
    sl_cswap(v[0], v[4]); sl_cswap(v[1], v[5]); sl_cswap(v[2], v[6]); sl_cswap(v[3], v[7]);
    sl_cswap(v[2], v[4]); sl_cswap(v[3], v[5]);
    sl_cswap(v[1], v[2]); sl_cswap(v[3], v[4]); sl_cswap(v[5], v[6]);
#pragma unroll
    for (int k = 0; k < SL_IPT; ++k) e16_store(&buf[sl_slot(o0 + k)], v[k]);
    __syncthreads();
    const u32 pmax = s_pmax;
    for (u32 L = SL_IPT; L < pmax; L <<= 1) {
        // runs of L slots counted from the bucket's start become runs of 2 L; the last run of a bucket may be short or missing
        const u32 pr = rel0 & ~(2 * L - 1);      // first slot of my pair of runs, relative to the bucket
        const u32 d = rel0 - pr;                 // outputs of the pair before mine
        const u32 left = P - pr;                 // slots of the bucket from the pair's start on
        const u32 lenA = min(L, left), lenB = left > L ? min(L, left - L) : 0u;
        const bool live = mine && P > L && lenB != 0;      // (a run without a partner stays as it is: v[] holds it)
        const u32 A = S + pr, B = A + L;
        if (live) {
            u32 lo = d > lenB ? d - lenB : 0, hi = min(d, lenA);
            while (lo < hi) {
                const u32 mid = (lo + hi) >> 1;
                const E16 x = e16_load(&buf[sl_slot(A + mid)]), y = e16_load(&buf[sl_slot(B + d - 1 - mid)]);
                if (e16_lt(x, y)) lo = mid + 1; else hi = mid;
            }
            u32 ai = lo, bi = d - lo;
            E16 va = e16_sel(ai < lenA, e16_load(&buf[sl_slot(A + min(ai, lenA - 1))]), e16_inf());
            E16 vb = e16_sel(bi < lenB, e16_load(&buf[sl_slot(B + min(bi, lenB - 1))]), e16_inf());
#pragma unroll
            for (int k = 0; k < SL_IPT; ++k) {
                const bool ta = !e16_lt(vb, va);
                v[k] = e16_sel(ta, va, vb);
                ai += ta ? 1u : 0u;
                bi += ta ? 0u : 1u;
                if (k + 1 < SL_IPT) {
                    const u32 ni = ta ? ai : bi, len = ta ? lenA : lenB;
                    const E16 nx = e16_sel(ni < len, e16_load(&buf[sl_slot((ta ? A : B) + min(ni, len - 1))]), e16_inf());
                    va = e16_sel(ta, nx, va);
                    vb = e16_sel(ta, vb, nx);
                }
            }
        }
        __syncthreads();                         // every read of this round is done
        if (live) {
#pragma unroll
            for (int k = 0; k < SL_IPT; ++k) e16_store(&buf[sl_slot(o0 + k)], v[k]);
        }
        __syncthreads();
    }
    // v[] = my eight slots of the sorted bucket.  Suffix words (bit 31 = "same key as my predecessor") go through LDS so
    // that the suffix array is written in order; the predecessor of a bucket's first element is the last real element
    // of the bucket before it, the first element of the tile is looked at by ss_boundary_kernel.
    const u32 imask = (u32)((1ull << ib) - 1ull);
    u32 w[SL_IPT];
    {
        E16 prev = v[0];
        bool have = false;
        if (mine) {
            if (rel0 != 0) {
                prev = e16_load(&buf[sl_slot(o0 - 1)]);
                have = true;
            } else if (b != 0) {
                prev = e16_load(&buf[sl_slot(s_ps[b - 1] + (eb - s_bs[b - 1]) - 1)]);
                have = true;
            }
        }
#pragma unroll
        for (int k = 0; k < SL_IPT; ++k) {
            const bool tie = have && e16_same_key(prev, v[k], ib);
            w[k] = ((u32)v[k].lo & imask) | (tie ? 0x80000000u : 0u);
            prev = v[k];
            have = true;
        }
    }
    __syncthreads();                             // the last reads of the elements are done: their LDS is the output stage now
    u32 *s_out = reinterpret_cast<u32 *>(buf);
    if (mine) {
#pragma unroll
        for (int k = 0; k < SL_IPT; ++k)
            if (rel0 + k < sz) s_out[eb + rel0 + k] = w[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SL_IPT; ++k) {
        const u32 p = k * SL_BLOCK + tid;
        if (p < count) sa_out[e0 + p] = s_out[p];
    }
}

// ---- greedy tile plan ----------------------------------------------------------------------------------------------
// The merge sort costs per tile, so tiles should be as full as the buckets allow: tile = the longest run of consecutive
// buckets that fits (SS_TILE_CAP).  That rule is a chain -- where a tile ends depends on where it began -- but every hop
// of the chain is shorter than a tile, so the chain enters every SEGMENT of SS_PLAN_SEG slots, and what happens inside
// a segment depends only on the bucket it is entered at:
//   nxt[k]   the bucket the tile that starts at bucket k ends before (binary search in the bucket starts),
//   exit[k]  where the chain that passes through k enters the next segment (a walk of <= SS_PLAN_SEG / 1 hops, ~17),
//   entry[t] = exit^t(0) by pointer doubling (jump tables J_i = exit^(2^i)), one thread per segment,
//   heads    every segment walked from its entry: the buckets tiles start at.
// A dozen small launches, ~0.2 ms at 2^29; 139 k tiles instead of the 162 k of the window plan.
constexpr u32 SS_PLAN_SEG = 65536;

__global__ __launch_bounds__(256) void ss_plan_next_kernel(const u32 *cstart, u32 ne, u32 n, u32 cap, u32 *nxt)
{
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > ne) return;
    if (k == ne) {
        nxt[ne] = ne;
        return;
    }
    // largest j in (k, ne] with start(j) - start(k) <= cap, start(ne) = n  (a bucket alone always fits: j >= k + 1)
    const u64 lim = (u64)cstart[k] + cap;
    u32 lo = k + 1, hi = ne;
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo + 1) >> 1);
        const u64 sm = mid < ne ? cstart[mid] : n;
        if (sm <= lim) lo = mid; else hi = mid - 1;
    }
    nxt[k] = lo;
}
__global__ __launch_bounds__(256) void ss_plan_exit_kernel(const u32 *cstart, u32 ne, const u32 *nxt, u32 *ex)
{
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > ne) return;
    if (k == ne) {
        ex[ne] = ne;
        return;
    }
    const u64 seg_end = ((u64)cstart[k] / SS_PLAN_SEG + 1) * SS_PLAN_SEG;
    u32 j = nxt[k];
    while (j < ne && cstart[j] < seg_end) j = nxt[j];
    ex[k] = j;
}
__global__ __launch_bounds__(256) void ss_plan_double_kernel(const u32 *jin, u32 cnt, u32 *jout)
{
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < cnt) jout[k] = jin[jin[k]];
}
// one thread per segment: its entry bucket, then the walk that marks the tile heads inside it
__global__ __launch_bounds__(256) void ss_plan_mark_kernel(const u32 *cstart, u32 ne, const u32 *nxt, const u32 *jumps, u32 levels,
                                                             u32 nseg, u32 *heads)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nseg) return;
    u32 k = 0;
    for (u32 i = 0; i < levels; ++i)
        if ((t >> i) & 1u) k = jumps[(size_t)i * (ne + 1) + k];
    const u64 seg_end = ((u64)t + 1) * SS_PLAN_SEG;
    while (k < ne && cstart[k] < seg_end) {
        heads[k] = 1;
        k = nxt[k];
    }
}
__global__ __launch_bounds__(256) void ss_plan_tiles_kernel(const u32 *heads, u32 ne, const u64 *rank, const u64 *total, u32 *tile_first)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) tile_first[*total] = ne;      // sentinel behind the last tile
    for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < ne; k += gridDim.x * blockDim.x)
        if (heads[k]) tile_first[rank[k]] = k;
}

// Equal keys may sit on both sides of a bucket boundary (the index is part of the number the splitters cut), hence of a
// tile boundary, where the sorting workgroup cannot see its predecessor: one thread per tile compares the keys of the
// last suffix of the tile before it and of its own first suffix (packed from the text again) and sets the flag.
__global__ __launch_bounds__(256) void ss_boundary_kernel(const MsdTile *tiles, u32 nt, SsText t, u32 *sa)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 || i >= nt) return;
    const u32 e0 = tiles[i].e0;
    const u32 pa = sa[e0 - 1] & 0x7fffffffu, pc = sa[e0] & 0x7fffffffu;
    u64 qa[5], qc[5];
    ss_stream_at<5>(t.codes, pa, qa);
    ss_stream_at<5>(t.codes, pc, qc);
    E16 a[1], c[1];
    ss_pack<1, 5>(qa, pa, t, a);
    ss_pack<1, 5>(qc, pc, t, c);
    if (e16_same_key(a[0], c[0], t.ib)) sa[e0] = pc | 0x80000000u;
}

