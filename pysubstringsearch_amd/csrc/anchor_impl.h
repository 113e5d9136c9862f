// anchor_impl.h -- anchors: a content-defined sample of the text's positions that breaks long ties in ONE round.
// Included by sa_build.hip (device kernels; the host side, anchor_rank_keys, sits there next to refine_rounds).
//
// Prefix doubling pays log2(longest repeat) rounds over every suffix that is still tied, and inside duplicated
// stretches of text (copies of a block, a line written over and over) every suffix stays tied until h reaches the end
// of the copy: `dup_blocks` took 16 full-size rank rounds (libsais, the reference's builder, is linear whatever the
// repeats: src/libsais/libsais.c:6458-6498).  The rounds below replace them for every text the run-length and
// one-word closed forms (rle_build.hip) do not take:
//
//   * H(q) = a hash of the w bytes at q.  M(s) = the LEFTMOST position of the window [s, s + omega) with the smallest
//     H (a minimizer).  The anchors are the positions some window chose: A = { M(s) }.  M never decreases with s, so
//     the anchors in text order are the distinct values of M, and "which anchor did window s choose" is a prefix sum
//     over [M(s) != M(s - 1)].
//   * M(s) - s depends on the omega + w - 1 bytes from s on and on nothing else.  Two suffixes i, j that share their
//     first h >= omega + w - 1 symbols therefore chose anchors at the same distance d < omega, agree on everything
//     before them, and compare exactly as the suffixes AT their anchors do: one key per suffix -- the rank of the
//     anchor its window chose -- sorts every group of the active list completely, whatever the length of the repeat.
//   * The ranks of the anchor suffixes among themselves: the next anchor behind an anchor q is a function of the
//     2 omega + w - 1 bytes from q on (the first position > q that some window starting in [q, ...] chooses; a window
//     that starts before q and chooses p > q has H(p) < H(q), so the window starting AT q chooses p as well).  Name
//     every anchor by the rank of its first tau >= 2 omega + w - 1 symbols (a sort of the anchors alone: text rounds
//     over m ~ 2 n / (omega + 1) elements): equal names mean equal text up to and including the way to the next anchor,
//     unequal names decide the comparison -- the order of the anchor suffixes is the lexicographic order of their name
//     sequences, i.e. the suffix array of an integer string of m symbols: rank rounds over m elements instead of n
//     (the last position of the text is always an anchor and its name holds the end of the text, so no name sequence
//     is a prefix of another).
//
// The same construction serves one level up (round 4, late): the anchors' names ARE a string of 32-bit symbols; when its
// rank rounds have reached depth 32 with many elements still tied, minimizers of THAT string (w = 1 symbol, omega = 16:
// names at depth 2 omega = 32 are the ranks the rounds have just produced) give a string 8.5 times shorter, and so on:
// every level costs five rank rounds and one anchor round over its own length, the lengths fall geometrically, and the
// number of passes over any level no longer depends on the length of the repeats.
//
// Everything is exact for every text (PSS_ANCHOR=1 sends any text through it in the tests and the fuzzer); a text whose
// windows choose more than n / 5 anchors (a hash that keeps falling, a stretch of period 2) declines, and the rank rounds
// over the whole text run as before.
#pragma once
#include "prims.h"

namespace pss {

constexpr u32 ANC_TILE = 4096;        // window starts per workgroup pass (256 threads x 16)
constexpr u32 ANC_MAX_OMEGA = 224;    // M(s) - s fits a byte
constexpr u32 ANC_HMAX = 0xffffffffu; // H of a position past the end of the text: never the smallest of a window that starts inside

__device__ __forceinline__ u32 anc_hash(u64 x, int w)
{
    if (w == 4) x &= 0xffffffffull;
    return (u32)((x * 0x9E3779B97F4A7C15ull) >> 33);      // 31 bits: below ANC_HMAX
}

// Leftmost minima of the G consecutive windows [j0 + k, j0 + k + omega), k < G <= omega, of h[]: the windows share the core
// [j0 + G - 1, j0 + omega); to its left the positions j0 + k .. j0 + G - 2, to its right j0 + omega .. j0 + omega + k - 1 --
// omega + 2 (G - 1) reads instead of G omega.
template <int G>
__device__ __forceinline__ void anc_window_mins(const u32 *h, u32 j0, u32 omega, u32 (&out)[G])
{
    u32 cv = h[j0 + G - 1], ci = j0 + G - 1;
    for (u32 q = j0 + G; q < j0 + omega; ++q) {
        const u32 hq = h[q];
        if (hq < cv) { cv = hq; ci = q; }
    }
    u32 lv[G], li[G];
    u32 v = ANC_HMAX, vi = 0;
    bool have = false;
#pragma unroll
    for (int k = G - 2; k >= 0; --k) {
        const u32 hq = h[j0 + k];
        if (!have || hq <= v) { v = hq; vi = j0 + k; have = true; }      // <=: the leftmost of equals
        lv[k] = v;
        li[k] = vi;
    }
    u32 rv = 0, ri = 0;
    bool rhave = false;
#pragma unroll
    for (int k = 0; k < G; ++k) {
        if (k > 0) {
            const u32 q = j0 + omega + k - 1;
            const u32 hq = h[q];
            if (!rhave || hq < rv) { rv = hq; ri = q; rhave = true; }
        }
        u32 bv = cv, bi = ci;                                            // core, unless the left part is as small ...
        if (k < G - 1 && lv[k] <= cv) { bv = lv[k]; bi = li[k]; }
        if (rhave && rv < bv) { bv = rv; bi = ri; }                      // ... or the right part smaller
        out[k] = bi;
    }
}

// d[s] = M(s) - s for every window start s < n; tile_cnt[t] = anchors first chosen by a window of tile t
// (windows s with s == 0 or M(s) != M(s - 1)).  `text`: the recoded text, 16-byte aligned, readable (and zero)
// up to n_read >= n, n_read % 16 == 0.
// SYMS: the string is an array of 32-bit symbols (the names of a coarser level's anchors, or the symbols of the run-length
// path's reduced string) and H hashes ONE symbol (w = 1); `text` is then that array.
template <bool SYMS>
__global__ __launch_bounds__(256) void anc_select_kernel(const u8 *text, u32 n, u32 n_read, u32 omega, int w, u8 *d,
                                                           u32 *tile_cnt, u32 num_tiles)
{
    // bytes of positions [tile0 - 16, tile0 + ANC_TILE + omega + 16): H is needed for [tile0 - 1, tile0 + ANC_TILE + omega - 1)
    __shared__ uint4 s_txt4[(ANC_TILE + ANC_MAX_OMEGA + 48) / 16];
    __shared__ u32 s_h[ANC_TILE + ANC_MAX_OMEGA + 32];
    __shared__ u32 s_red[4];
    const u8 *s_txt = reinterpret_cast<const u8 *>(s_txt4);
    const u32 tid = threadIdx.x;
    for (u32 tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const u32 tile0 = tile * ANC_TILE;
        if (SYMS) {
            const u32 *sym = reinterpret_cast<const u32 *>(text);
            for (u32 j = tid; j < ANC_TILE + omega; j += 256) {
                const long long p = (long long)tile0 - 1 + j;
                s_h[j] = (p < 0 || p >= (long long)n) ? ANC_HMAX : anc_hash((u64)sym[p], 4);
            }
        } else {
        const u32 nvec = (ANC_TILE + omega + 16 + 8 + 15) / 16 + 1;
        for (u32 v = tid; v < nvec; v += 256) {
            const long long pos = (long long)tile0 - 16 + 16ll * v;
            uint4 x = make_uint4(0, 0, 0, 0);
            if (pos >= 0 && pos < (long long)n_read) x = *reinterpret_cast<const uint4 *>(text + pos);
            s_txt4[v] = x;
        }
        __syncthreads();
        // H of position tile0 - 1 + j, j in [0, ANC_TILE + omega): 17 consecutive j per thread (17 * 256 = 4352 >= 4096 + 224)
        {
            const u32 jn = ANC_TILE + omega;
            const u32 j0 = tid * 17u;
            if (j0 < jn) {
                u64 x = 0;
#pragma unroll
                for (int c = 0; c < 8; ++c) x |= (u64)s_txt[15 + j0 + c] << (8 * c);
#pragma unroll
                for (u32 c = 0; c < 17; ++c) {
                    const u32 j = j0 + c;
                    if (j < jn) {
                        const long long p = (long long)tile0 - 1 + j;
                        s_h[j] = (p < 0 || p >= (long long)n) ? ANC_HMAX : anc_hash(x, w);
                        x = (x >> 8) | ((u64)s_txt[15 + j + 8] << 56);
                    }
                }
            }
        }
        }
        __syncthreads();
        // my 17 windows: starts A + k, A = tile0 + 16 tid - 1, k = 0 (the neighbour's last, for the flag of my first) .. 16
        const u32 jA = 16u * tid;                   // index of position A in s_h
        u32 mk[17];                                 // M(A + k) as an index into s_h
        if (omega >= 17) {
            anc_window_mins<17>(s_h, jA, omega, mk);
        } else if (omega >= 9) {
            u32 m0[9], m1[8];
            anc_window_mins<9>(s_h, jA, omega, m0);
            anc_window_mins<8>(s_h, jA + 9, omega, m1);
#pragma unroll
            for (int k = 0; k < 9; ++k) mk[k] = m0[k];
#pragma unroll
            for (int k = 0; k < 8; ++k) mk[9 + k] = m1[k];
        } else {
#pragma unroll
            for (int k = 0; k <= 16; ++k) {
                u32 bv = s_h[jA + k], bi = jA + k;
                for (u32 q = 1; q < omega; ++q) {
                    const u32 hq = s_h[jA + k + q];
                    if (hq < bv) { bv = hq; bi = jA + k + q; }
                }
                mk[k] = bi;
            }
        }
        const u32 s0 = tile0 + 16u * tid;
        u32 dw[4] = {0, 0, 0, 0};
        u32 cnt = 0;
#pragma unroll
        for (int k = 1; k <= 16; ++k) {
            const u32 s = s0 + (u32)k - 1u;
            if (s < n) {
                const u32 dist = mk[k] - (jA + (u32)k);
                dw[(k - 1) >> 2] |= dist << (8 * ((k - 1) & 3));
                cnt += (s == 0 || mk[k] != mk[k - 1]) ? 1u : 0u;
            }
        }
        if (s0 < n) *reinterpret_cast<uint4 *>(d + s0) = make_uint4(dw[0], dw[1], dw[2], dw[3]);
        cnt = wave_incl_sum(cnt);
        if (lane_id() == 63) s_red[wave_id()] = cnt;
        __syncthreads();
        if (tid == 0) tile_cnt[tile] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
        __syncthreads();
    }
}

// The anchors in text order: Q[tile_off[t] + ...] = s + d[s] for every window s that is the first to choose its anchor.
// FILL: instead, akey[s] = rank_a[index of the anchor window s chose] for every s (rank_a: 1-based ranks of the anchor
// suffixes) -- the key of the one round that finishes the suffix array.
template <bool FILL>
__global__ __launch_bounds__(256) void anc_walk_kernel(const u8 *d, u32 n, const u64 *tile_off, u32 num_tiles, u32 *Q,
                                                         const u32 *rank_a, u32 *akey)
{
    __shared__ u32 scr[4 + 1];
    const u32 tid = threadIdx.x;
    for (u32 tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const u32 s0 = tile * ANC_TILE + 16u * tid;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (s0 < n) x = *reinterpret_cast<const uint4 *>(d + s0);
        const u32 dw[4] = {x.x, x.y, x.z, x.w};
        u32 prev = (s0 > 0 && s0 <= n) ? (s0 - 1u) + (u32)d[s0 - 1] : 0xffffffffu;       // M(s0 - 1)
        u32 flags = 0, mloc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const u32 s = s0 + (u32)k;
            const u32 m = s + ((dw[k >> 2] >> (8 * (k & 3))) & 0xffu);
            mloc[k] = m;
            if (s < n && (s == 0 || m != prev)) flags |= 1u << k;
            prev = m;
        }
        u32 tot;
        const u32 ex = block_excl_sum<4>((u32)__popc(flags), scr, &tot);
        u32 at = (u32)tile_off[tile] + ex;          // anchors before my first window
        if (FILL) {
            u32 out[16];
            u32 cur = at ? rank_a[at - 1] : 0u;     // the anchor chosen by the window before mine (none before window 0)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if ((flags >> k) & 1u) cur = rank_a[at++];
                out[k] = cur;
            }
            if (s0 + 16 <= n) {
#pragma unroll
                for (int k = 0; k < 16; k += 4)
                    *reinterpret_cast<uint4 *>(akey + s0 + k) = make_uint4(out[k], out[k + 1], out[k + 2], out[k + 3]);
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (s0 + (u32)k < n) akey[s0 + k] = out[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if ((flags >> k) & 1u) Q[at++] = mloc[k];
        }
    }
}

}  // namespace pss
