// sa_rerank_impl.h -- the first rerank of a sorted key array (flags, ranks, the active list of tied suffixes), the sparse mode (ranks without an inverse suffix array) and the keys of a rank round.
// Included by sa_build.hip (inside namespace pss, after the alphabet kernels): one translation unit, split by route.

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
