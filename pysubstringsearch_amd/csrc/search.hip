// search.hip -- batched substring search over device-resident chunks
// (text + suffix array in HBM).  Replaces Reader::search (reference
// src/lib.rs:201-287) and the Python loop of Reader.search_multiple
// (pysubstringsearch/__init__.py:61-73) by ONE launch sequence per batch.
//
//   K1 search_interval   one wavefront per (query, chunk).  The reference's two
//                        binary searches (lib.rs:212-252, ~2*log2(n) dependent
//                        probes) become two 64-ary searches: every lane probes
//                        its own suffix, a 64-bit ballot of the comparison
//                        results picks the sub-interval -> ~2*log65(n) dependent
//                        steps (5+5 at n = 2^29 instead of 29+29).
//   scan                 hit counts -> hit offsets (total H)
//   K2 hit_lines         one thread per suffix-array hit: newline scan back to
//                        the entry start (lib.rs:270-273) and forward to its end
//                        (lib.rs:266-269).  Per-(query, chunk) dedupe on the
//                        entry start (lib.rs:262,274) without a hash set: a hit
//                        is kept iff it is the LEFTMOST occurrence of the query
//                        inside its entry -- exactly one hit per distinct entry
//                        start satisfies this, so the kept multiset equals the
//                        reference's.
//   scan x2              kept flags -> entry index, entry lengths -> byte offset
//   K3 emit              copies entry bytes into the packed result
//
// Output order: query-major, inside a query chunk-major, inside a chunk
// suffix-array order of the kept hit (the reference's inter-chunk order is
// nondeterministic, lib.rs:207,280; results are compared as multisets).
#include "prims.h"
#include "scan.h"
#include "search.h"

namespace pss {

constexpr u32 kSkip = 0xffffffffu;

struct InKept {
    const u32 *len;
    __device__ u64 operator()(u64 i) const { return len[i] != kSkip ? 1u : 0u; }
};
struct InLen {
    const u32 *len;
    __device__ u64 operator()(u64 i) const { const u32 l = len[i]; return l != kSkip ? l : 0u; }
};


// -1: suffix < pattern, 0: pattern is a prefix of the suffix, +1: suffix > pattern
// (unsigned bytes, a proper prefix sorts first -- Rust slice cmp, lib.rs:224,246).
// text and pat must be readable 16 bytes past their ends.
__device__ __forceinline__ int cmp_suffix(const u8 *text, u32 n, u32 s, const u8 *pat, u32 plen)
{
    const u32 avail = n - s;
    const u32 L = plen < avail ? plen : avail;
    u32 i = 0;
    while (i < L) {
        u64 a = load_u64_unaligned(text + s + i);
        u64 b = load_u64_unaligned(pat + i);
        const u32 rem = L - i;
        if (rem < 8) {
            const u64 mask = (1ull << (8 * rem)) - 1ull;
            a &= mask;
            b &= mask;
        }
        if (a != b) {
            const int sh = __builtin_ctzll(a ^ b) & ~7;
            return ((a >> sh) & 0xffu) < ((b >> sh) & 0xffu) ? -1 : 1;
        }
        i += 8;
    }
    return (L == plen) ? 0 : -1;
}

// First index in [lo, hi) whose suffix is NOT before the bound; wave-cooperative.
// upper == false: suffixes < pattern are "before"; upper == true: suffixes that
// are < pattern or start with it are "before".
__device__ __forceinline__ u32 wave_bound(const u8 *text, u32 n, const u32 *sa, const u8 *pat, u32 plen, u32 lo,
                                          u32 hi, bool upper)
{
    const u32 lane = lane_id();
    while (hi > lo) {
        const u32 s = hi - lo;
        if (s <= kWave) {
            bool before = false;
            if (lane < s) {
                const int c = cmp_suffix(text, n, sa[lo + lane], pat, plen);
                before = upper ? (c <= 0) : (c < 0);
            }
            return lo + (u32)__popcll(__ballot(before));
        }
        const u32 p = lo + (u32)(((u64)(lane + 1) * s) / (kWave + 1));
        const int c = cmp_suffix(text, n, sa[p], pat, plen);
        const bool before = upper ? (c <= 0) : (c < 0);
        const u32 k = (u32)__popcll(__ballot(before));
        const u32 nlo = (k == 0) ? lo : (u32)__shfl((int)p, (int)k - 1) + 1;
        const u32 nhi = (k == kWave) ? hi : (u32)__shfl((int)p, (int)k);
        lo = nlo;
        hi = nhi;
    }
    return lo;
}

__global__ __launch_bounds__(256) void search_interval_kernel(const ChunkDesc *chunks, u32 nc, const u8 *qbytes,
                                                                const u64 *qoff, u64 nvq, u32 *lo_out, u32 *cnt_out)
{
    const u64 vq = (u64)blockIdx.x * (blockDim.x / kWave) + wave_id();
    if (vq >= nvq) return;
    const u32 q = (u32)(vq / nc), c = (u32)(vq % nc);
    const ChunkDesc ch = chunks[c];
    const u8 *pat = qbytes + qoff[q];
    const u32 plen = (u32)(qoff[q + 1] - qoff[q]);
    const u32 L = wave_bound(ch.text, ch.n, ch.sa, pat, plen, 0, ch.n, false);
    const u32 U = wave_bound(ch.text, ch.n, ch.sa, pat, plen, L, ch.n, true);
    if (lane_id() == 0) {
        lo_out[vq] = L;
        cnt_out[vq] = U - L;
    }
}

// ------------------------------------------------------------- hit -> entry --

// High bit of every byte of x that is zero (exact, no cross-byte carries).
__device__ __forceinline__ u64 zero_bytes(u64 x)
{
    const u64 m = 0x7f7f7f7f7f7f7f7full;
    return ~(((x & m) + m) | x | m);
}

__global__ __launch_bounds__(256) void hit_lines_kernel(const ChunkDesc *chunks, u32 nc, const u8 *qbytes,
                                                          const u64 *qoff, u64 nvq, const u32 *lo, const u64 *hit_off,
                                                          u64 H, u32 *start_out, u32 *len_out)
{
    const u64 NL = 0x0a0a0a0a0a0a0a0aull;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < H; t += (u64)gridDim.x * blockDim.x) {
        // owning (query, chunk): last vq with hit_off[vq] <= t
        u64 a = 0, b = nvq;
        while (b - a > 1) {
            const u64 mid = a + (b - a) / 2;
            if (hit_off[mid] <= t) a = mid; else b = mid;
        }
        const u64 vq = a;
        const u32 q = (u32)(vq / nc), c = (u32)(vq % nc);
        const ChunkDesc ch = chunks[c];
        const u8 *pat = qbytes + qoff[q];
        const u32 plen = (u32)(qoff[q + 1] - qoff[q]);
        const u32 di = ch.sa[lo[vq] + (u32)(t - hit_off[vq])];
        // Backwards, 8 bytes at a time, to the entry start (lib.rs:270-273).  Any
        // earlier occurrence of the query inside the entry makes this hit a duplicate
        // (lib.rs:262,274): candidates are the bytes equal to the query's first byte.
        const u64 first = plen ? 0x0101010101010101ull * pat[0] : 0;
        u32 p = di;            // scan frontier: bytes [p, di) hold no newline
        bool dup = false, at_start = false;
        while (p >= 8 && !dup && !at_start) {
            const u64 w = load_u64_unaligned(ch.text + p - 8);      // byte k of w = text[p-8+k]
            const u64 nlm = zero_bytes(w ^ NL);
            u32 keep_from = 0;                                       // first byte index of w inside the entry
            if (nlm) {
                keep_from = (u32)((63 - __builtin_clzll(nlm)) >> 3) + 1;
                at_start = true;
            }
            if (plen == 0) {
                dup = keep_from < 8;                                 // an earlier position exists in the entry
            } else {
                u64 cand = zero_bytes(w ^ first);
                if (keep_from) cand &= keep_from < 8 ? ~0ull << (8 * keep_from) : 0ull;
                while (cand && !dup) {
                    const u32 k = (u32)(__builtin_ctzll(cand) >> 3);
                    cand &= cand - 1;
                    dup = cmp_suffix(ch.text, ch.n, p - 8 + k, pat, plen) == 0;
                }
            }
            p = at_start ? p - 8 + keep_from : p - 8;
        }
        while (p > 0 && !dup && !at_start) {                         // the first < 8 bytes of the chunk
            const u8 cb = ch.text[p - 1];
            if (cb == '\n') break;
            --p;
            dup = plen == 0 || (cb == pat[0] && cmp_suffix(ch.text, ch.n, p, pat, plen) == 0);
        }
        if (dup) {
            len_out[t] = kSkip;
            start_out[t] = 0;
            continue;
        }
        const u32 line_start = p;
        // forwards to the entry end (lib.rs:266-269; no newline: len - 1); text is zero padded past n
        u32 e = di;
        for (;;) {
            const u64 nlm = zero_bytes(load_u64_unaligned(ch.text + e) ^ NL);
            if (nlm) {
                e += (u32)(__builtin_ctzll(nlm) >> 3);
                break;
            }
            e += 8;
            if (e >= ch.n) break;
        }
        if (e >= ch.n) e = ch.n - 1;
        start_out[t] = line_start;
        len_out[t] = e >= line_start ? e - line_start : 0;
    }
}

__global__ __launch_bounds__(256) void emit_kernel(const ChunkDesc *chunks, u32 nc, u64 nvq, const u64 *hit_off, u64 H,
                                                     const u32 *start, const u32 *len, const u64 *eidx,
                                                     const u64 *boff, u64 *ent_off, u8 *out)
{
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < H; t += (u64)gridDim.x * blockDim.x) {
        const u32 l = len[t];
        if (l == kSkip) continue;
        u64 a = 0, b = nvq;
        while (b - a > 1) {
            const u64 mid = a + (b - a) / 2;
            if (hit_off[mid] <= t) a = mid; else b = mid;
        }
        const ChunkDesc ch = chunks[(u32)(a % nc)];
        const u64 o = boff[t];
        ent_off[eidx[t]] = o;
        const u8 *src = ch.text + start[t];
        u32 i = 0;
        for (; i + 8 <= l; i += 8) {                       // 8 bytes per step, unaligned on both sides
            const u64 v = load_u64_unaligned(src + i);
            __builtin_memcpy(out + o + i, &v, 8);
        }
        for (; i < l; ++i) out[o + i] = src[i];
    }
}

// entries per query = sum over its chunks of kept hits
__global__ __launch_bounds__(256) void query_counts_kernel(u32 nc, u32 nq, const u64 *hit_off, const u64 *eidx,
                                                             u64 *qcount)
{
    const u32 q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const u64 v0 = (u64)q * nc, v1 = v0 + nc;
    qcount[q] = eidx[hit_off[v1]] - eidx[hit_off[v0]];
}

// --------------------------------------------------------------------- host --

enum SSlot { Q_BYTES = 10, Q_OFF, Q_LO, Q_CNT, Q_HITOFF, Q_START, Q_LEN, Q_EIDX, Q_BOFF, Q_ENTOFF, Q_OUT, Q_SMALL, Q_QCOUNT };

int search_batch_device(DeviceCtx *ctx, const ChunkDesc *d_chunks, u32 nc, const uint8_t *qbytes,
                        const uint64_t *qoffsets, uint32_t nq, HostResult *res, pss_search_stats *st)
{
    hipStream_t s = ctx->stream;
    memset(st, 0, sizeof *st);
    st->queries = nq;
    res->nq = nq;
    res->qcount = (u64 *)calloc(nq ? nq : 1, sizeof(u64));
    res->offsets = nullptr;
    res->bytes = nullptr;
    res->n_entries = 0;
    if (!res->qcount) return PSS_ENOMEM;
    if (nq == 0 || nc == 0) {
        res->offsets = (u64 *)calloc(1, sizeof(u64));
        return res->offsets ? PSS_OK : PSS_ENOMEM;
    }
    const u64 qtotal = qoffsets[nq];
    const u64 nvq = (u64)nq * nc;
    PSS_TRY(ctx->slot[Q_BYTES].reserve(qtotal + 32));
    PSS_TRY(ctx->slot[Q_OFF].reserve(((size_t)nq + 1) * 8));
    PSS_TRY(ctx->slot[Q_LO].reserve(nvq * 4));
    PSS_TRY(ctx->slot[Q_CNT].reserve(nvq * 4));
    PSS_TRY(ctx->slot[Q_HITOFF].reserve((nvq + 1) * 8));
    PSS_TRY(ctx->slot[Q_SMALL].reserve(SC_MAX_BLOCKS * 8 + 64));
    PSS_TRY(ctx->slot[Q_QCOUNT].reserve((size_t)nq * 8));
    u8 *d_q = ctx->slot[Q_BYTES].as<u8>();
    u64 *d_qoff = ctx->slot[Q_OFF].as<u64>();
    u32 *d_lo = ctx->slot[Q_LO].as<u32>();
    u32 *d_cnt = ctx->slot[Q_CNT].as<u32>();
    u64 *d_hitoff = ctx->slot[Q_HITOFF].as<u64>();
    u64 *d_partial = ctx->slot[Q_SMALL].as<u64>();
    u64 *d_total = d_partial + SC_MAX_BLOCKS;
    u64 *d_qcount = ctx->slot[Q_QCOUNT].as<u64>();
    u64 *h_small = static_cast<u64 *>(ctx->pinned);

    hipEvent_t e0, e1, e2;
    PSS_HIP(hipEventCreate(&e0));
    PSS_HIP(hipEventCreate(&e1));
    PSS_HIP(hipEventCreate(&e2));
    PSS_HIP(hipMemsetAsync(d_q + qtotal, 0, 32, s));
    if (qtotal) PSS_HIP(hipMemcpyAsync(d_q, qbytes, qtotal, hipMemcpyHostToDevice, s));
    PSS_HIP(hipMemcpyAsync(d_qoff, qoffsets, ((size_t)nq + 1) * 8, hipMemcpyHostToDevice, s));
    PSS_HIP(hipEventRecord(e0, s));
    const u64 waves_per_block = 256 / kWave;
    hipLaunchKernelGGL(search_interval_kernel, dim3((u32)((nvq + waves_per_block - 1) / waves_per_block)), dim3(256), 0,
                       s, d_chunks, nc, d_q, d_qoff, nvq, d_lo, d_cnt);
    PSS_HIP(hipEventRecord(e1, s));
    PSS_TRY(device_excl_scan(ctx, InU32{d_cnt}, nvq, d_partial, d_total, d_hitoff));
    PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipStreamSynchronize(s));
    const u64 H = h_small[0];
    st->hits = H;
    u64 E = 0, B = 0;
    if (H) {
        PSS_TRY(ctx->slot[Q_START].reserve(H * 4));
        PSS_TRY(ctx->slot[Q_LEN].reserve(H * 4));
        PSS_TRY(ctx->slot[Q_EIDX].reserve((H + 1) * 8));
        PSS_TRY(ctx->slot[Q_BOFF].reserve((H + 1) * 8));
        u32 *d_start = ctx->slot[Q_START].as<u32>();
        u32 *d_len = ctx->slot[Q_LEN].as<u32>();
        u64 *d_eidx = ctx->slot[Q_EIDX].as<u64>();
        u64 *d_boff = ctx->slot[Q_BOFF].as<u64>();
        const u32 grid = (u32)std::min<u64>((u64)ctx->num_cus * 16, (H + 255) / 256);
        hipLaunchKernelGGL(hit_lines_kernel, dim3(grid), dim3(256), 0, s, d_chunks, nc, d_q, d_qoff, nvq, d_lo,
                           d_hitoff, H, d_start, d_len);
        PSS_TRY(device_excl_scan(ctx, InKept{d_len}, H, d_partial, d_total, d_eidx));
        PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
        PSS_TRY(device_excl_scan(ctx, InLen{d_len}, H, d_partial, d_total, d_boff));
        PSS_HIP(hipMemcpyAsync(h_small + 1, d_total, 8, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
        E = h_small[0];
        B = h_small[1];
        PSS_TRY(ctx->slot[Q_ENTOFF].reserve((E + 1) * 8));
        PSS_TRY(ctx->slot[Q_OUT].reserve(B + 16));
        u64 *d_entoff = ctx->slot[Q_ENTOFF].as<u64>();
        u8 *d_out = ctx->slot[Q_OUT].as<u8>();
        hipLaunchKernelGGL(emit_kernel, dim3(grid), dim3(256), 0, s, d_chunks, nc, nvq, d_hitoff, H, d_start, d_len,
                           d_eidx, d_boff, d_entoff, d_out);
        hipLaunchKernelGGL(query_counts_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, nc, nq, d_hitoff, d_eidx,
                           d_qcount);
        PSS_HIP(hipEventRecord(e2, s));
        res->offsets = (u64 *)malloc((E + 1) * sizeof(u64));
        res->bytes = (u8 *)malloc(B ? B : 1);
        if (!res->offsets || !res->bytes) return PSS_ENOMEM;
        if (E) PSS_HIP(hipMemcpyAsync(res->offsets, d_entoff, E * 8, hipMemcpyDeviceToHost, s));
        if (B) PSS_HIP(hipMemcpyAsync(res->bytes, d_out, B, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipMemcpyAsync(res->qcount, d_qcount, (size_t)nq * 8, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
        res->offsets[E] = B;
    } else {
        PSS_HIP(hipEventRecord(e2, s));
        PSS_HIP(hipStreamSynchronize(s));
        res->offsets = (u64 *)calloc(1, sizeof(u64));
        if (!res->offsets) return PSS_ENOMEM;
    }
    PSS_HIP(hipGetLastError());
    res->n_entries = E;
    st->entries = E;
    st->result_bytes = B;
    float ms = 0.f;
    PSS_HIP(hipEventElapsedTime(&ms, e0, e2));
    st->ms_device = ms;
    PSS_HIP(hipEventElapsedTime(&ms, e0, e1));
    st->ms_interval = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipEventDestroy(e2);
    return PSS_OK;
}

}  // namespace pss
