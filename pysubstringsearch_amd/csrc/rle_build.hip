// rle_build.hip -- suffix array of a text made of long runs of equal bytes, for gfx950.
//
// Prefix doubling pays log2(longest repeat) rounds over everything that is still tied, and inside a
// run c^L every suffix is tied with its neighbours until h reaches what is left of the run: the
// adversarial corpora of bench.py (`runs`: runs of up to 8192 equal bytes; `periodic`: a^4095 \n
// repeated) took 11 and 25 full-size rank rounds, 0.5 and 1.3 s at n = 2^29 (round 1 and 2 figures).
// This path sorts them in what two radix passes cost:
//
//  1. Run table.  start[k] = first position of the k-th maximal run (S runs, start[S] = n).
//  2. Reduced string.  Run k = (byte c, length L, next byte d; d = "end of text" < every byte after the
//     last run) becomes ONE symbol   meta(k) = (c, type, type ? -L : L),   type = (d > c).
//     For two suffixes that start at run heads the order of the texts equals the lexicographic order
//     of the meta strings: same c and L1 < L2 means the shorter run's successor d1 meets a c of the
//     longer run, so the shorter one is smaller iff d1 < c (type 0: ascending L) and larger iff
//     d1 > c (type 1: descending L); a type-0 run is below a type-1 run of any length (d1 < c < d2 at
//     the first difference); equal symbols mean equal runs AND the same side of c, and the comparison
//     moves on to the next runs, which is exactly what the next symbols compare.  The end of the
//     reduced string is the end of the text (smaller than everything) in both orders.
//     Its suffix array SAr comes from the same machinery as every other build: the symbols are
//     sorted (radix sort of 41-bit keys), then rank rounds over an inverse array resolve the ties
//     (refine_rounds of sa_build.hip with h0 = 1) -- on S elements instead of n.
//  3. Expansion.  The suffix at position x inside run k, r = start[k+1] - x bytes before its end, is
//     c^r followed by the suffix at the head of run k+1.  By the argument above suffixes are ordered by
//     (c, type, type ? -r : r) first -- a dense number id(x) < n: per class (c, type) the longest run
//     gives the range -- and by the rank of the NEXT run head among equals.  So the suffixes are
//     generated in the order of their next run head (one pass over SAr: the runs before SAr[0], SAr[1],
//     ...; the last run, whose successor is the end of the text, goes first) and a STABLE radix sort by
//     id(x) alone -- 13 bits for `periodic`, 15 for `runs`: two passes -- leaves the suffix array.
//
//     When the table fits (below), the radix sort is not needed either: the expansion is a matrix walk.  Put the
//     runs in generation order into columns and the ids into rows: entry (id, t) exists iff run t covers
//     that id, i.e. lo_t <= id < hi_t (a run covers one id per remaining length), and the suffix array is the
//     row-major walk over the existing entries.  Blocks of 64 columns (one wave): a ballot gives the entries
//     of a row inside a block, popcounts give the table count[block][id]; its column-wise exclusive sums plus
//     the row starts are where every (block, row) segment begins, and the second walk stores
//     SA[start + lanes below me] = position -- contiguous segments, no sort, no keys (col_* kernels).
//
// Every step is exact for every text (a text without runs reduces to itself and pays for it: the
// caller takes this path when runs average >= 8 bytes).  HBM-bound integer work, no MFMA.
#include "rle_build.h"

#include "prims.h"
#include "radix_sort.h"
#include "sa_build.h"
#include "scan.h"

namespace pss {

constexpr int RL_BLOCK = 256;
constexpr u32 RL_TILE = RL_BLOCK * 16;       // bytes of text per workgroup
constexpr int S_RLE = 30;                    // workspace slot (sa_build.hip: 0-9, 26-29; search: 10-22, 28; writer: 24, 25)

// bit j: a run starts at byte i0 + j (i0 % 16 == 0).  Bytes at or past n never start a run.
__device__ __forceinline__ u32 run_start_mask16(const u8 *T, u32 n, u32 i0, bool aligned)
{
    if (i0 >= n) return 0u;
    u32 mask = 0;
    if (aligned && i0 + 16 <= n) {
        const uint4 v = *reinterpret_cast<const uint4 *>(T + i0);
        const u32 w[4] = {v.x, v.y, v.z, v.w};
        u32 prev = i0 ? (u32)T[i0 - 1] : (~v.x & 0xffu);       // position 0 always starts a run
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u32 diff = w[k] ^ ((w[k] << 8) | prev);       // byte j: T[j] ^ T[j - 1]
            prev = w[k] >> 24;
            const u32 t = (((diff & 0x7f7f7f7fu) + 0x7f7f7f7fu) | diff) & 0x80808080u;   // bit 7 of every non-zero byte
            mask |= (((t >> 7) & 1u) | ((t >> 14) & 2u) | ((t >> 21) & 4u) | ((t >> 28) & 8u)) << (4 * k);
        }
    } else {
        u32 prev = i0 ? (u32)T[i0 - 1] : 0x100u;
        for (u32 j = 0; j < 16 && i0 + j < n; ++j) {
            const u32 c = T[i0 + j];
            if (c != prev) mask |= 1u << j;
            prev = c;
        }
    }
    return mask;
}

__global__ __launch_bounds__(RL_BLOCK) void rle_count_kernel(const u8 *T, u32 n, u32 *blk_cnt)
{
    __shared__ u32 scr[RL_BLOCK / kWave + 1];
    const bool aligned = ((uintptr_t)T & 15) == 0;
    const u32 i0 = blockIdx.x * RL_TILE + threadIdx.x * 16;
    const u32 c = (u32)__popc(run_start_mask16(T, n, i0, aligned));
    u32 total = 0;
    (void)block_excl_sum<RL_BLOCK / kWave>(c, scr, &total);
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = total;
}

__global__ __launch_bounds__(RL_BLOCK) void rle_starts_kernel(const u8 *T, u32 n, const u64 *blk_off, u32 *start, u32 S)
{
    __shared__ u32 scr[RL_BLOCK / kWave + 1];
    const bool aligned = ((uintptr_t)T & 15) == 0;
    const u32 i0 = blockIdx.x * RL_TILE + threadIdx.x * 16;
    u32 mask = run_start_mask16(T, n, i0, aligned);
    u32 at = (u32)blk_off[blockIdx.x] + block_excl_sum<RL_BLOCK / kWave>((u32)__popc(mask), scr, nullptr);
    while (mask) {
        const u32 j = (u32)__ffs(mask) - 1u;
        mask &= mask - 1u;
        start[at++] = i0 + j;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) start[S] = n;
}

// meta symbol of every run (see the header) and the longest run of every class (c, type)
__global__ __launch_bounds__(RL_BLOCK) void rle_meta_kernel(const u8 *T, u32 n, const u32 *start, u32 S, u64 *keys, u32 *vals,
                                                            u32 *cls_max)
{
    __shared__ u32 s_max[512];
    for (u32 i = threadIdx.x; i < 512; i += RL_BLOCK) s_max[i] = 0;
    __syncthreads();
    for (u32 k = blockIdx.x * RL_BLOCK + threadIdx.x; k < S; k += gridDim.x * RL_BLOCK) {
        const u32 s0 = start[k], e = start[k + 1];
        const u32 c = T[s0], L = e - s0;
        const u32 type = (e < n && (u32)T[e] > c) ? 1u : 0u;
        keys[k] = ((u64)c << 33) | ((u64)type << 32) | (u64)(type ? ~L : L);
        vals[k] = k;
        atomicMax(&s_max[c * 2 + type], L);
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < 512; i += RL_BLOCK)
        if (s_max[i]) atomicMax(&cls_max[i], s_max[i]);
}

// cls_base[q] = first id of class q (exclusive sum of the longest runs), cls_base[512] = number of ids
__global__ __launch_bounds__(512) void rle_class_kernel(const u32 *cls_max, u32 *cls_base)
{
    __shared__ u32 scr[512 / kWave + 1];
    u32 total = 0;
    const u32 ex = block_excl_sum<512 / kWave>(cls_max[threadIdx.x], scr, &total);
    cls_base[threadIdx.x] = ex;
    if (threadIdx.x == 0) cls_base[512] = total;
}

// Per run: how the id of a suffix r bytes before the run's end is formed.
//   type 0 (ascending r):  id = base + r - 1          stored: base
//   type 1 (descending r): id = base + maxL - r       stored: (base + maxL) | 1 << 31
__global__ __launch_bounds__(RL_BLOCK) void rle_runinfo_kernel(const u8 *T, u32 n, const u32 *start, u32 S, const u32 *cls_max,
                                                               const u32 *cls_base, u32 *sbase)
{
    for (u32 k = blockIdx.x * RL_BLOCK + threadIdx.x; k < S; k += gridDim.x * RL_BLOCK) {
        const u32 s0 = start[k], e = start[k + 1];
        const u32 c = T[s0];
        const u32 type = (e < n && (u32)T[e] > c) ? 1u : 0u;
        const u32 q = c * 2 + type;
        sbase[k] = type ? ((cls_base[q] + cls_max[q]) | 0x80000000u) : cls_base[q];
    }
}

__global__ __launch_bounds__(RL_BLOCK) void rle_j0_kernel(const u32 *SAr, u32 S, u32 *j0)
{
    for (u32 j = blockIdx.x * RL_BLOCK + threadIdx.x; j < S; j += gridDim.x * RL_BLOCK)
        if (SAr[j] == 0u) *j0 = j;
}

// t-th run in generation order: the last run first (its successor is the end of the text, the smallest
// suffix of all), then the predecessors of SAr[0], SAr[1], ... (run 0 has none: its slot j0 is skipped).
__device__ __forceinline__ u32 rle_ord(const u32 *SAr, u32 j0, u32 S, u32 t)
{
    if (t == 0) return S - 1u;
    const u32 j = (t - 1u < j0) ? t - 1u : t;
    return SAr[j] - 1u;
}

struct InOrdLen {
    const u32 *SAr, *start, *j0;
    u32 S;
    __device__ u64 operator()(u64 t) const
    {
        const u32 k = rle_ord(SAr, *j0, S, (u32)t);
        return (u64)(start[k + 1] - start[k]);
    }
};

constexpr int EX_IPT = 8;

// Slot p of the generated sequence (off[t] <= p < off[t + 1]: the (p - off[t])-th byte of run ord(t)):
// K[p] = id of that suffix, V[p] = its position.  `with_slot`: the key also carries p below the id (inputs
// of <= one tile go to the single-workgroup sort, which orders equal keys by value, not by arrival).
__global__ __launch_bounds__(RL_BLOCK) void rle_expand_kernel(const u32 *SAr, const u32 *j0p, const u32 *start, const u32 *sbase,
                                                              const u64 *off, u32 S, u32 n, u64 *K, u32 *V, int with_slot)
{
    const u32 j0 = *j0p;
    const u32 chunks = (n + EX_IPT - 1) / EX_IPT;
    for (u32 c = blockIdx.x * RL_BLOCK + threadIdx.x; c < chunks; c += gridDim.x * RL_BLOCK) {
        const u32 p0 = c * EX_IPT;
        u32 lo = 0, hi = S;                          // off[0] = 0 <= p0 < n = off[S]
        while (hi - lo > 1u) {
            const u32 mid = lo + ((hi - lo) >> 1);
            if ((u32)off[mid] <= p0) lo = mid; else hi = mid;
        }
        u32 t = lo;
        u32 k = rle_ord(SAr, j0, S, t);
        u32 s0 = start[k], e = start[k + 1], sb = sbase[k];
        u32 o0 = (u32)off[t], o1 = o0 + (e - s0);
        u64 kk[EX_IPT];
        u32 vv[EX_IPT];
#pragma unroll
        for (int i = 0; i < EX_IPT; ++i) {
            const u32 p = p0 + (u32)i;
            kk[i] = 0;
            vv[i] = 0;
            if (p < n) {
                while (p >= o1) {
                    ++t;
                    k = rle_ord(SAr, j0, S, t);
                    s0 = start[k];
                    e = start[k + 1];
                    sb = sbase[k];
                    o0 = o1;
                    o1 = o0 + (e - s0);
                }
                const u32 x = s0 + (p - o0), r = e - x;
                const u32 id = (sb >> 31) ? (sb & 0x7fffffffu) - r : sb + r - 1u;
                kk[i] = with_slot ? (((u64)id << 32) | p) : (u64)id;
                vv[i] = x;
            }
        }
        if (p0 + EX_IPT <= n) {
            ulonglong2 *Kv = reinterpret_cast<ulonglong2 *>(K + p0);
#pragma unroll
            for (int i = 0; i < EX_IPT / 2; ++i) Kv[i] = make_ulonglong2(kk[2 * i], kk[2 * i + 1]);
            uint4 *Vv = reinterpret_cast<uint4 *>(V + p0);
#pragma unroll
            for (int i = 0; i < EX_IPT / 4; ++i) Vv[i] = make_uint4(vv[4 * i], vv[4 * i + 1], vv[4 * i + 2], vv[4 * i + 3]);
        } else {
            for (int i = 0; i < EX_IPT && p0 + (u32)i < n; ++i) {
                K[p0 + i] = kk[i];
                V[p0 + i] = vv[i];
            }
        }
    }
}


// ---- expansion as a matrix walk (see the header) ----------------------------------------------------

constexpr u32 CB = 64;           // runs per block = lanes of a wave
constexpr u32 CSEG = 64;         // blocks per segment of the column scan

// Per run in generation order: the ids it covers, [lo, hi), and how the position follows from the id:
// type 0: id = base + r - 1  ->  x = e - r = (e - 1 + base) - id;   type 1: id = top - r  ->  x = (e - top) + id.
__global__ __launch_bounds__(RL_BLOCK) void col_prep_kernel(const u32 *SAr, const u32 *j0p, const u32 *start, const u32 *sbase, u32 S,
                                                            u32 *lo, u32 *hi, u32 *c0)
{
    const u32 j0 = *j0p;
    for (u32 t = blockIdx.x * RL_BLOCK + threadIdx.x; t < S; t += gridDim.x * RL_BLOCK) {
        const u32 k = rle_ord(SAr, j0, S, t);
        const u32 s0 = start[k], e = start[k + 1], sb = sbase[k], L = e - s0;
        if (sb >> 31) {
            const u32 top = sb & 0x7fffffffu;
            lo[t] = top - L;
            hi[t] = top | 0x80000000u;          // bit 31: x = c0 + id
            c0[t] = e - top;
        } else {
            lo[t] = sb;
            hi[t] = sb + L;                     // x = c0 - id
            c0[t] = e - 1u + sb;
        }
    }
}

// wave (b, c): block b of 64 runs, chunk c of 64 ids.  count[b][id] = runs of the block that cover id.
__global__ __launch_bounds__(256) void col_count_kernel(const u32 *lo, const u32 *hi, u32 S, u32 idpad, u32 nchunks, u32 nwaves,
                                                        u32 *table)
{
    const u32 w = blockIdx.x * 4u + (u32)wave_id(), lane = (u32)lane_id();
    if (w >= nwaves) return;
    const u32 c = w % nchunks, b = w / nchunks;
    const u32 t = b * CB + lane;
    const u32 mylo = t < S ? lo[t] : 0xffffffffu, myhi = t < S ? (hi[t] & 0x7fffffffu) : 0u;
    const u32 id0 = c * 64u;
    const bool touch = mylo < id0 + 64u && myhi > id0, full = mylo <= id0 && myhi >= id0 + 64u;
    if (__ballot(touch) == 0ull) return;                                  // the table was zeroed
    u32 mine = 0;
    if (__ballot(touch && !full) == 0ull) {
        mine = (u32)__popcll(__ballot(full));                             // no run starts or ends inside the chunk: one count for all 64 ids
    } else {
#pragma unroll 8
        for (u32 j = 0; j < 64u; ++j) {
            const u32 id = id0 + j;
            const u32 cnt = (u32)__popcll(__ballot(mylo <= id && id < myhi));
            if (lane == j) mine = cnt;
        }
    }
    table[(size_t)b * idpad + id0 + lane] = mine;
}

// exclusive sums down the columns, two levels: inside segments of CSEG blocks, then over the segments
__global__ __launch_bounds__(256) void col_scan1_kernel(u32 *table, u32 idpad, u32 NB, u32 *segtot)
{
    const u32 id = blockIdx.x * 256u + threadIdx.x, seg = blockIdx.y;
    if (id >= idpad) return;
    const u32 b0 = seg * CSEG, b1 = min(NB, b0 + CSEG);
    u32 run = 0;
    for (u32 b = b0; b < b1; ++b) {
        const u32 v = table[(size_t)b * idpad + id];
        table[(size_t)b * idpad + id] = run;
        run += v;
    }
    segtot[(size_t)seg * idpad + id] = run;
}
__global__ __launch_bounds__(256) void col_scan2_kernel(u32 *segtot, u32 idpad, u32 nseg, u32 *tot)
{
    const u32 id = blockIdx.x * 256u + threadIdx.x;
    if (id >= idpad) return;
    u32 run = 0;
    for (u32 seg = 0; seg < nseg; ++seg) {
        const u32 v = segtot[(size_t)seg * idpad + id];
        segtot[(size_t)seg * idpad + id] = run;
        run += v;
    }
    tot[id] = run;
}

__global__ __launch_bounds__(256) void col_scatter_kernel(const u32 *lo, const u32 *hi, const u32 *c0, u32 S, u32 idpad, u32 nchunks,
                                                          u32 nwaves, const u32 *table, const u32 *segtot, const u64 *idoff, u32 *SA)
{
    const u32 w = blockIdx.x * 4u + (u32)wave_id(), lane = (u32)lane_id();
    if (w >= nwaves) return;
    const u32 c = w % nchunks, b = w / nchunks;
    const u32 t = b * CB + lane;
    const u32 mylo = t < S ? lo[t] : 0xffffffffu, hraw = t < S ? hi[t] : 0u, myc0 = t < S ? c0[t] : 0u;
    const u32 myhi = hraw & 0x7fffffffu;
    const bool plus = (hraw >> 31) != 0u;
    const u32 id0 = c * 64u;
    const bool touch = mylo < id0 + 64u && myhi > id0, full = mylo <= id0 && myhi >= id0 + 64u;
    if (__ballot(touch) == 0ull) return;
    const u32 offs = table[(size_t)b * idpad + id0 + lane] + segtot[(size_t)(b / CSEG) * idpad + id0 + lane] + (u32)idoff[id0 + lane];
    if (__ballot(touch && !full) == 0ull) {
        // the same runs cover all 64 ids of the chunk: one ballot, 64 stores per covering lane
        const u32 rank = mbcnt(__ballot(full));
        u32 x = plus ? myc0 + id0 : myc0 - id0;
        const u32 step = plus ? 1u : 0xffffffffu;
#pragma unroll 8
        for (u32 j = 0; j < 64u; ++j) {
            const u32 o = (u32)__builtin_amdgcn_readlane((int)offs, (int)j);      // (every lane takes part in the readlane)
            if (full) SA[o + rank] = x;
            x += step;
        }
        return;
    }
#pragma unroll 4
    for (u32 j = 0; j < 64u; ++j) {
        const u32 id = id0 + j;
        const bool act = mylo <= id && id < myhi;
        const u64 m = __ballot(act);
        if (m == 0ull) continue;
        const u32 o = (u32)__builtin_amdgcn_readlane((int)offs, (int)j);
        if (act) SA[o + mbcnt(m)] = plus ? myc0 + id : myc0 - id;
    }
}

struct RleEvents {
    hipEvent_t ev[4] = {};
    int created = 0;
    ~RleEvents()
    {
        for (int i = 0; i < created; ++i) (void)hipEventDestroy(ev[i]);
    }
};

int rle_suffix_array(DeviceCtx *ctx, const uint8_t *T, uint32_t n, uint32_t S, uint32_t *SA, bool profile, RleStats *rs,
                     pss_sa_stats *st)
{
    hipStream_t s = ctx->stream;
    if (n < 2 || S == 0 || S > n) {
        set_error("rle_suffix_array: bad arguments");
        return PSS_EINVAL;
    }
    const u32 ntiles = (u32)(((u64)n + RL_TILE - 1) / RL_TILE);
    const int grid_stream = ctx->num_cus * 8;
    const u32 grid_runs = (u32)std::min<u64>((u64)grid_stream, ((u64)S + RL_BLOCK - 1) / RL_BLOCK);

    PSS_TRY(ctx->slot[1].reserve((size_t)n * 8));      // S_K0, S_K1, S_V0, S_V1, S_WORK of sa_build.hip
    PSS_TRY(ctx->slot[2].reserve((size_t)n * 8));
    PSS_TRY(ctx->slot[3].reserve((size_t)n * 4));
    PSS_TRY(ctx->slot[4].reserve((size_t)n * 4));
    const size_t sort_ws = radix_sort_workspace_bytes();
    PSS_TRY(ctx->slot[9].reserve(sort_ws + 65536));
    void *work = ctx->slot[9].p;
    u64 *K[2] = {ctx->slot[1].as<u64>(), ctx->slot[2].as<u64>()};
    u32 *V[2] = {ctx->slot[3].as<u32>(), ctx->slot[4].as<u32>()};

    size_t need = 0;
    auto plan = [&](size_t bytes) { const size_t o = need; need = round_up(need + bytes, 256); return o; };
    const size_t o_start = plan(((size_t)S + 1) * 4), o_sbase = plan((size_t)S * 4), o_sar = plan((size_t)S * 4);
    const size_t o_off = plan(((size_t)S + 1) * 8), o_bcnt = plan((size_t)ntiles * 4), o_boff = plan(((size_t)ntiles + 1) * 8);
    const size_t o_part = plan((SC_MAX_BLOCKS + 2) * 8), o_cmax = plan(512 * 4), o_cbase = plan(513 * 4), o_j0 = plan(64);
    PSS_TRY(ctx->slot[S_RLE].reserve(need));
    u8 *base = ctx->slot[S_RLE].as<u8>();
    u32 *d_start = reinterpret_cast<u32 *>(base + o_start), *d_sbase = reinterpret_cast<u32 *>(base + o_sbase);
    u32 *d_sar = reinterpret_cast<u32 *>(base + o_sar);
    u64 *d_off = reinterpret_cast<u64 *>(base + o_off);
    u32 *d_bcnt = reinterpret_cast<u32 *>(base + o_bcnt);
    u64 *d_boff = reinterpret_cast<u64 *>(base + o_boff);
    u64 *d_part = reinterpret_cast<u64 *>(base + o_part), *d_total = d_part + SC_MAX_BLOCKS;
    u32 *d_cmax = reinterpret_cast<u32 *>(base + o_cmax), *d_cbase = reinterpret_cast<u32 *>(base + o_cbase);
    u32 *d_j0 = reinterpret_cast<u32 *>(base + o_j0);
    u32 *h_small = static_cast<u32 *>(ctx->pinned);

    RleEvents evs;
    if (profile)
        for (; evs.created < 4; ++evs.created) PSS_HIP(hipEventCreate(&evs.ev[evs.created]));
    auto mark = [&](int i) { if (profile) (void)hipEventRecord(evs.ev[i], s); };
    mark(0);

    // 1. run table
    hipLaunchKernelGGL(rle_count_kernel, dim3(ntiles), dim3(RL_BLOCK), 0, s, T, n, d_bcnt);
    PSS_TRY(device_excl_scan(ctx, InU32{d_bcnt}, ntiles, d_part, d_total, d_boff));
    // S (counted by sa_symbols_kernel) sizes every run table; the starts come from rle_count / rle_starts, a second
    // implementation of the same definition.  Their totals are compared at the next host sync (below): a mismatch
    // means one of the two kernels is wrong, and the tables would be too.
    PSS_HIP(hipMemcpyAsync(h_small + 8, d_total, 8, hipMemcpyDeviceToHost, s));
    hipLaunchKernelGGL(rle_starts_kernel, dim3(ntiles), dim3(RL_BLOCK), 0, s, T, n, d_boff, d_start, S);

    // 2. reduced string: symbols, their sort, ties by rank rounds
    PSS_HIP(hipMemsetAsync(d_cmax, 0, 512 * 4, s));
    hipLaunchKernelGGL(rle_meta_kernel, dim3(grid_runs), dim3(RL_BLOCK), 0, s, T, n, d_start, S, K[0], V[0], d_cmax);
    hipLaunchKernelGGL(rle_class_kernel, dim3(1), dim3(512), 0, s, d_cmax, d_cbase);
    hipLaunchKernelGGL(rle_runinfo_kernel, dim3(grid_runs), dim3(RL_BLOCK), 0, s, T, n, d_start, S, d_cmax, d_cbase, d_sbase);
    PSS_HIP(hipGetLastError());
    mark(1);
    int cur = 0;
    SortStats ss;
    PSS_TRY(radix_sort_pairs(ctx, K, V, S, 41, 0x3fu, nullptr, 0, work, &cur, false, &ss));
    PSS_TRY(suffix_rounds_integer(ctx, S, K, V, cur, d_sar, st));
    mark(2);

    // 3. expansion
    hipLaunchKernelGGL(rle_j0_kernel, dim3(grid_runs), dim3(RL_BLOCK), 0, s, d_sar, S, d_j0);
    PSS_HIP(hipMemcpyAsync(h_small, d_cbase + 512, 4, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipStreamSynchronize(s));
    {
        const u64 counted = (u64)h_small[8] | ((u64)h_small[9] << 32);
        if (counted != (u64)S) {
            set_error("rle_suffix_array: %llu run starts found, %u runs counted (internal error)", (unsigned long long)counted, S);
            return PSS_EDEVICE;
        }
    }
    const u32 num_ids = h_small[0];
    int id_bits = 1;
    while ((1ull << id_bits) < (u64)num_ids) ++id_bits;
    SortStats fs;
    const u32 NB = (S + CB - 1) / CB, idpad = (u32)round_up((size_t)num_ids, 64), nchunks = idpad / 64u;
    const u32 nseg = (NB + CSEG - 1) / CSEG;
    // (the bounds keep the table inside the first key buffer and everything else inside the second)
    const bool columns = !knob("PSS_RLE_SORT") && n >= 4096u && (u64)S * 4 <= (u64)n && (u64)idpad * 4 <= (u64)n &&
                         (u64)NB * idpad * 4 <= (u64)n * 8 && (u64)NB * nchunks < (1ull << 31);
    if (columns) {
        // matrix walk: table in the first key buffer, everything else in the second
        u32 *table = reinterpret_cast<u32 *>(K[0]);
        u8 *q = reinterpret_cast<u8 *>(K[1]);
        size_t qo = 0;
        auto qcarve = [&](size_t bytes) { u8 *p = q + qo; qo = round_up(qo + bytes, 256); return p; };
        u32 *segtot = reinterpret_cast<u32 *>(qcarve((size_t)nseg * idpad * 4));
        u32 *tot = reinterpret_cast<u32 *>(qcarve((size_t)idpad * 4));
        u64 *idoff = reinterpret_cast<u64 *>(qcarve(((size_t)idpad + 1) * 8));
        u32 *lo = reinterpret_cast<u32 *>(qcarve((size_t)S * 4)), *hi = reinterpret_cast<u32 *>(qcarve((size_t)S * 4));
        u32 *c0 = reinterpret_cast<u32 *>(qcarve((size_t)S * 4));
        const u32 nwaves = NB * nchunks;
        PSS_HIP(hipMemsetAsync(table, 0, (size_t)NB * idpad * 4, s));
        hipLaunchKernelGGL(col_prep_kernel, dim3(grid_runs), dim3(RL_BLOCK), 0, s, d_sar, d_j0, d_start, d_sbase, S, lo, hi, c0);
        hipLaunchKernelGGL(col_count_kernel, dim3((nwaves + 3) / 4), dim3(256), 0, s, lo, hi, S, idpad, nchunks, nwaves, table);
        hipLaunchKernelGGL(col_scan1_kernel, dim3((idpad + 255) / 256, nseg), dim3(256), 0, s, table, idpad, NB, segtot);
        hipLaunchKernelGGL(col_scan2_kernel, dim3((idpad + 255) / 256), dim3(256), 0, s, segtot, idpad, nseg, tot);
        PSS_TRY(device_excl_scan(ctx, InU32{tot}, idpad, d_part, d_total, idoff));
        hipLaunchKernelGGL(col_scatter_kernel, dim3((nwaves + 3) / 4), dim3(256), 0, s, lo, hi, c0, S, idpad, nchunks, nwaves, table,
                           segtot, idoff, SA);
        PSS_HIP(hipGetLastError());
    } else {
        PSS_TRY(device_excl_scan(ctx, InOrdLen{d_sar, d_start, d_j0, S}, S, d_part, d_total, d_off));
        const bool small = n <= 4096u;                      // single-workgroup sort: whole 64-bit key, ties by value
        const int passes = (id_bits + 7) / 8;
        const int final_buf = small ? 0 : (passes & 1);     // pass p reads buffer p & 1 (starting from 0) and writes the other
        u32 *const v_scratch = V[final_buf];
        V[final_buf] = SA;
        const u32 chunks = (n + EX_IPT - 1) / EX_IPT;
        const u32 grid_ex = (u32)std::min<u64>((u64)grid_stream * 4, ((u64)chunks + RL_BLOCK - 1) / RL_BLOCK);
        hipLaunchKernelGGL(rle_expand_kernel, dim3(grid_ex), dim3(RL_BLOCK), 0, s, d_sar, d_j0, d_start, d_sbase, d_off, S, n, K[0],
                           V[0], small ? 1 : 0);
        PSS_HIP(hipGetLastError());
        int dst = 0;
        PSS_TRY(radix_sort_pairs(ctx, K, V, n, small ? 64 : id_bits, small ? 0xffu : ((1u << passes) - 1u), nullptr, 0, work, &dst,
                                 profile, &fs));
        if (V[dst] != SA) PSS_HIP(hipMemcpyAsync(SA, V[dst], (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        V[final_buf] = v_scratch;
    }
    mark(3);
    if (st && !columns) {
        st->sort_launches += fs.launches;
        st->sort_elems += fs.elems;
        st->ms_sort += fs.ms;
        st->ms_pairs += fs.ms_pairs;
        st->pairs_launches += fs.pairs_launches;
        st->pairs_elems += fs.pairs_elems;
    }
    if (rs) {
        rs->runs = S;
        rs->id_bits = (u32)id_bits;
        rs->columns = columns;
        if (profile) {
            PSS_HIP(hipStreamSynchronize(s));
            float ms = 0.f;
            PSS_HIP(hipEventElapsedTime(&ms, evs.ev[0], evs.ev[1]));
            rs->ms_table = ms;
            PSS_HIP(hipEventElapsedTime(&ms, evs.ev[1], evs.ev[2]));
            rs->ms_reduced = ms;
            PSS_HIP(hipEventElapsedTime(&ms, evs.ev[2], evs.ev[3]));
            rs->ms_expand = ms;
        }
    }
    return PSS_OK;
}

// ------------------------------------------------------------------ periodic prefix (rle_build.h) --

uint32_t period_of_head(const uint8_t *head, uint32_t len)
{
    // (a head of one byte repeated has period 1: runs are the run-length path's business, and the word found below
    // must be primitive -- the smallest period -- for its rotations to be distinct)
    bool run = len > 1;
    for (u32 i = 1; i < len && run; ++i) run = head[i] == head[0];
    if (run) return 0;
    for (u32 p = 2; p <= kPeriodMax && 4 * p <= len; ++p) {
        bool ok = true;
        for (u32 i = 0; i + p < len; ++i)
            if (head[i] != head[i + p]) {
                ok = false;
                break;
            }
        if (ok) return p;
    }
    return 0;
}

// first i with T[i] != T[i + p] (n - p when there is none), into *first by atomicMin
__global__ __launch_bounds__(256) void period_extent_kernel(const u8 *T, u32 n, u32 p, u32 *first)
{
    const u64 lim = (u64)n - p;
    u32 best = 0xffffffffu;
    for (u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < lim; i += (u64)gridDim.x * blockDim.x * 16) {
        if (best != 0xffffffffu) break;
        const u64 e = i + 16 < lim ? i + 16 : lim;
        for (u64 j = i; j < e; ++j)
            if (T[j] != T[j + p]) {
                best = (u32)j;
                break;
            }
    }
    best = ~wave_incl_max(~best);                      // lane 63: the wave's smallest
    if ((threadIdx.x & 63u) == 63u && best != 0xffffffffu) atomicMin(first, best);
}

int period_extent(DeviceCtx *ctx, const uint8_t *T, uint32_t n, uint32_t p, uint32_t *m)
{
    hipStream_t s = ctx->stream;
    PSS_TRY(ctx->slot[S_RLE].reserve(1u << 20));
    u32 *d_first = ctx->slot[S_RLE].as<u32>();
    // (a word of the pinned scratch that neither the alphabet counts nor the head of the text, both still needed, sit in)
    u32 *h = reinterpret_cast<u32 *>(static_cast<u8 *>(ctx->pinned) + 32768 + 16384 - 64);
    PSS_HIP(hipMemsetAsync(d_first, 0xff, 4, s));
    hipLaunchKernelGGL(period_extent_kernel, dim3(ctx->num_cus * 8), dim3(256), 0, s, T, n, p, d_first);
    PSS_HIP(hipMemcpyAsync(h, d_first, 4, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipStreamSynchronize(s));
    *m = h[0] == 0xffffffffu ? n : h[0] + p;
    return PSS_OK;
}

struct PeriodBlock {      // one rotation's suffixes in the suffix array: SA[out .. out + count)
    u32 out, count, cls;
};

// SA[o] for every o inside a rotation block: block k holds the long suffixes of class cls -- cls, cls + p, ... --
// ascending or descending.  Eight consecutive outputs per thread: one search of the block table, then a walk.
__global__ __launch_bounds__(256) void period_fill_kernel(const PeriodBlock *blocks, u32 nb, u32 p, u32 desc, u32 n, u32 *SA)
{
    __shared__ PeriodBlock s_blk[kPeriodMax];
    for (u32 i = threadIdx.x; i < nb; i += blockDim.x) s_blk[i] = blocks[i];
    __syncthreads();
    for (u64 o0 = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 8; o0 < n; o0 += (u64)gridDim.x * blockDim.x * 8) {
        u32 lo = 0, hi = nb;                                    // last block that starts at or before o0
        while (hi - lo > 1) {
            const u32 mid = (lo + hi) >> 1;
            if (s_blk[mid].out <= (u32)o0) lo = mid; else hi = mid;
        }
        u32 k = lo;
        const u32 oe = (u32)(o0 + 8 < n ? o0 + 8 : n);
        for (u32 o = (u32)o0; o < oe; ++o) {
            while (k + 1 < nb && s_blk[k + 1].out <= o) ++k;
            const PeriodBlock b = s_blk[k];
            if (o < b.out || o >= b.out + b.count) continue;  // a slot of one of the late suffixes (period_late_kernel)
            const u32 q = o - b.out;
            SA[o] = b.cls + p * (desc ? b.count - 1 - q : q);
        }
    }
}

__global__ __launch_bounds__(256) void period_late_kernel(const u32 *pos, const u32 *val, u32 cnt, u32 *SA)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) SA[pos[i]] = val[i];
}

int period_suffix_array(DeviceCtx *ctx, const uint8_t *T, uint32_t n, uint32_t p, uint32_t m, const uint8_t *W, uint32_t *SA,
                        bool *accepted)
{
    *accepted = false;
    hipStream_t s = ctx->stream;
    if (p < 2 || p > kPeriodMax || m > n || n - m > kPeriodTailMax) return PSS_OK;
    const u32 t = n - m;
    // A suffix is "long" when the repetition goes on for more than `margin` bytes behind its start: every comparison
    // with a late suffix, and with a long one of another rotation, is then decided inside the repetition.
    const u32 margin = 2 * p + 2 * t + 1;
    if ((u64)margin * 4 > m) return PSS_OK;
    const u32 long_end = m - margin;                 // long suffixes: [0, long_end); late ones: [long_end, n)
    const u32 nl = n - long_end;
    std::vector<u8> late(nl);
    PSS_HIP(hipMemcpyAsync(late.data(), T + long_end, nl, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipStreamSynchronize(s));
    // Inside a rotation class, suffix j = i + k p agrees with suffix i for as long as j's repetition lasts; then j sees
    // the byte that ends the repetition (or the end of the text, which sorts first) and i another turn of the word.
    const bool desc = m == n || late[m - long_end] < W[m % p];
    // items to order: the p rotation blocks and the nl late suffixes
    struct Item {
        u32 late;       // 1: a late suffix, start = text offset; 0: the rotation block of class start
        u32 start;
    };
    std::vector<Item> items;
    items.reserve((size_t)p + nl);
    for (u32 c = 0; c < p; ++c)
        if (c < long_end) items.push_back(Item{0, c});
    for (u32 i = long_end; i < n; ++i) items.push_back(Item{1, i});
    auto rot = [&](u32 c, u32 k) { return W[(c + k) % p]; };
    auto at = [&](u32 i, u32 k) { return late[i - long_end + k]; };
    auto less = [&](const Item &a, const Item &b) {
        if (!a.late && !b.late) {                    // two rotations of a primitive word differ within p bytes
            for (u32 k = 0; k < p; ++k) {
                const u8 x = rot(a.start, k), y = rot(b.start, k);
                if (x != y) return x < y;
            }
            return false;
        }
        if (a.late && b.late) {
            const u32 la = n - a.start, lb = n - b.start, l = std::min(la, lb);
            const int c = memcmp(&late[a.start - long_end], &late[b.start - long_end], l);
            return c ? c < 0 : la < lb;
        }
        // a late suffix against a block: against the word repeated for as long as the late suffix lasts; a late
        // suffix that is a prefix of that is the shorter one
        const Item &x = a.late ? a : b, &blk = a.late ? b : a;
        const u32 lx = n - x.start;
        int c = 0;
        for (u32 k = 0; k < lx && c == 0; ++k) {
            const u8 u = at(x.start, k), v = rot(blk.start, k);
            if (u != v) c = u < v ? -1 : 1;
        }
        const bool x_first = c <= 0;
        return a.late ? x_first : !x_first;
    };
    std::sort(items.begin(), items.end(), less);
    std::vector<PeriodBlock> blocks;
    std::vector<u32> pos, val;
    u32 out = 0;
    for (const Item &it : items) {
        if (it.late) {
            pos.push_back(out);
            val.push_back(it.start);
            ++out;
        } else {
            const u32 cnt = (long_end - it.start + p - 1) / p;         // suffixes it.start, it.start + p, ... < long_end
            blocks.push_back(PeriodBlock{out, cnt, it.start});
            out += cnt;
        }
    }
    if (out != n || blocks.empty()) {
        set_error("period_suffix_array: internal count mismatch");
        return PSS_EDEVICE;
    }
    const size_t need = blocks.size() * sizeof(PeriodBlock) + 2 * pos.size() * 4 + 256;
    PSS_TRY(ctx->slot[S_RLE].reserve(std::max<size_t>(need, 1u << 20)));
    u8 *d = ctx->slot[S_RLE].as<u8>();
    PeriodBlock *d_blocks = reinterpret_cast<PeriodBlock *>(d);
    u32 *d_pos = reinterpret_cast<u32 *>(d + round_up(blocks.size() * sizeof(PeriodBlock), 256));
    u32 *d_val = d_pos + pos.size();
    PSS_HIP(hipMemcpyAsync(d_blocks, blocks.data(), blocks.size() * sizeof(PeriodBlock), hipMemcpyHostToDevice, s));
    PSS_HIP(hipMemcpyAsync(d_pos, pos.data(), pos.size() * 4, hipMemcpyHostToDevice, s));
    PSS_HIP(hipMemcpyAsync(d_val, val.data(), val.size() * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(period_fill_kernel, dim3(ctx->num_cus * 8), dim3(256), 0, s, d_blocks, (u32)blocks.size(), p, desc ? 1u : 0u, n, SA);
    hipLaunchKernelGGL(period_late_kernel, dim3((u32)((pos.size() + 255) / 256)), dim3(256), 0, s, d_pos, d_val, (u32)pos.size(), SA);
    PSS_HIP(hipStreamSynchronize(s));      // (the tables on the host side are about to go away)
    *accepted = true;
    return PSS_OK;
}

}  // namespace pss
