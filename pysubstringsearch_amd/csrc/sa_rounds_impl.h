// sa_rounds_impl.h -- kernels of the rounds that resolve ties: text rounds, rank rounds by a segmented merge sort, the LDS middle tiers, the segmented merge sort of the large groups, periodic runs, the probe of the ties.
// Included by sa_build.hip (inside namespace pss, after the alphabet kernels): one translation unit, split by route.

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
