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

#include <vector>

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
    if (upper && hi - lo > kWave) {
        // the interval of a query is short far more often than not: look at the 64 suffixes
        // right behind the lower bound first (one step when the query has < 64 hits here)
        const int c = cmp_suffix(text, n, sa[lo + lane], pat, plen);
        const u32 k = (u32)__popcll(__ballot(c <= 0));
        if (k < kWave) return lo + k;
        lo += kWave;
    }
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

// ---- key samples: confine a query to a window of the suffix array -----------------------
//
// key8(i) = first 8 bytes of suffix sa[i], big-endian, zero padded past the end of the text, is
// non-decreasing in i (zero is the smallest byte, so the padding never breaks the order).  With
// P8 = the query's first min(8, plen) bytes: every suffix whose key8 < P8|00.. is smaller than
// the query and every suffix whose key8 > P8|ff.. is larger and does not start with it, so both
// ends of the query's interval lie between A = first i with key8(i) >= P8|00.. and B = first i
// with key8(i) > P8|ff...  The table holds key8 of every 2^shift-th suffix: jA / jB searched
// there give A > (jA - 1) << shift and B <= jB << shift.

__device__ __forceinline__ u64 key8_be(const u8 *p) { return __builtin_bswap64(load_u64_unaligned(p)); }

// P8|00.. and P8|ff.. of a query (pat is readable 16 bytes past its end)
__device__ __forceinline__ void query_keys(const u8 *pat, u32 plen, u64 &k_lo, u64 &k_hi)
{
    const u64 mask = plen >= 8 ? ~0ull : (plen ? ~0ull << (8 * (8 - plen)) : 0ull);
    k_lo = key8_be(pat) & mask;
    k_hi = k_lo | ~mask;
}

// first j in [lo, hi) with k[j] >= key (upper: > key); wave-cooperative 64-ary search
__device__ __forceinline__ u32 wave_bound_key(const u64 *k, u32 lo, u32 hi, u64 key, bool upper)
{
    const u32 lane = lane_id();
    while (hi > lo) {
        const u32 s = hi - lo;
        if (s <= kWave) {
            bool before = false;
            if (lane < s) {
                const u64 v = k[lo + lane];
                before = upper ? (v <= key) : (v < key);
            }
            return lo + (u32)__popcll(__ballot(before));
        }
        const u32 p = lo + (u32)(((u64)(lane + 1) * s) / (kWave + 1));
        const u64 v = k[p];
        const bool before = upper ? (v <= key) : (v < key);
        const u32 c = (u32)__popcll(__ballot(before));
        const u32 nlo = (c == 0) ? lo : (u32)__shfl((int)p, (int)c - 1) + 1;
        const u32 nhi = (c == kWave) ? hi : (u32)__shfl((int)p, (int)c);
        lo = nlo;
        hi = nhi;
    }
    return lo;
}

// window [lo, hi) of the suffix array that holds the query's interval (whole array without a table)
__device__ __forceinline__ void sample_window_wave(const ChunkDesc &ch, const u8 *pat, u32 plen, u32 &lo, u32 &hi)
{
    lo = 0;
    hi = ch.n;
    if (ch.skeys == nullptr) return;
    const u32 ns = (u32)(((u64)ch.n + (1u << ch.shift) - 1) >> ch.shift);
    u64 k_lo, k_hi;
    query_keys(pat, plen, k_lo, k_hi);
    const u32 ja = wave_bound_key(ch.skeys, 0, ns, k_lo, false);
    u32 jb = ja;
    if (ja < ns) {
        // jB is jA or jA + 1 unless the query's first 8 bytes are frequent: 64 consecutive samples first
        const u32 s = min(ns - ja, (u32)kWave);
        const u32 lane = lane_id();
        const bool before = lane < s && ch.skeys[ja + lane] <= k_hi;
        const u32 c = (u32)__popcll(__ballot(before));
        jb = ja + c;
        if (c == kWave) jb = wave_bound_key(ch.skeys, ja + kWave, ns, k_hi, true);
    }
    lo = ja ? (ja - 1) << ch.shift : 0u;
    hi = jb < ns ? jb << ch.shift : ch.n;
}

__device__ __forceinline__ void sample_window_lane(const ChunkDesc &ch, const u8 *pat, u32 plen, u32 &lo, u32 &hi)
{
    lo = 0;
    hi = ch.n;
    if (ch.skeys == nullptr) return;
    const u32 ns = (u32)(((u64)ch.n + (1u << ch.shift) - 1) >> ch.shift);
    u64 k_lo, k_hi;
    query_keys(pat, plen, k_lo, k_hi);
    u32 a = 0, b = ns;
    while (a < b) {
        const u32 mid = a + ((b - a) >> 1);
        if (ch.skeys[mid] < k_lo) a = mid + 1; else b = mid;
    }
    const u32 ja = a;
    b = ns;                                       // gallop: jB is almost always jA or jA + 1
    for (u32 step = 1; a < b; step <<= 1) {
        const u32 p = a + step - 1;
        if (p >= b) break;
        if (ch.skeys[p] <= k_hi) a = p + 1; else { b = p; break; }
    }
    while (a < b) {
        const u32 mid = a + ((b - a) >> 1);
        if (ch.skeys[mid] <= k_hi) a = mid + 1; else b = mid;
    }
    lo = ja ? (ja - 1) << ch.shift : 0u;
    hi = a < ns ? a << ch.shift : ch.n;
}

__global__ __launch_bounds__(256) void key_samples_kernel(const u8 *text, const u32 *sa, u32 n, u32 shift, u64 *skeys)
{
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 i = j << shift;
    if (i < n) skeys[j] = key8_be(text + sa[i]);
}

int build_key_samples(DeviceCtx *ctx, const uint8_t *d_text, const uint32_t *d_sa, uint32_t n, uint32_t shift,
                      uint64_t *d_skeys)
{
    if (n == 0) return PSS_OK;
    const u64 ns = sample_count(n, shift);
    hipLaunchKernelGGL(key_samples_kernel, dim3((u32)((ns + 255) / 256)), dim3(256), 0, ctx->stream, d_text, d_sa, n,
                       shift, d_skeys);
    PSS_HIP(hipGetLastError());
    return PSS_OK;
}

// Large batches: one LANE per (query, chunk) and plain binary searches.  The 64-ary wave
// search above minimises latency (5+5 dependent steps) but touches 64 random SA + text
// sectors per step, ~80 KB per pair -- at 1.5 M pairs (100 k queries x 15 chunks) that is
// HBM-bound.  A binary search reads ~2 sectors per step (29+29 steps, the top levels shared in
// L2), 20x less traffic; with tens of thousands of pairs in flight the longer dependent chain
// is hidden by parallelism instead.
__global__ __launch_bounds__(256) void search_interval_lane_kernel(const ChunkDesc *chunks, u32 nc, const u8 *qbytes,
                                                                     const u64 *qoff, u64 nvq, u32 *lo_out,
                                                                     u32 *cnt_out)
{
    const u64 vq = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (vq >= nvq) return;
    const u32 q = (u32)(vq / nc), c = (u32)(vq % nc);
    const ChunkDesc ch = chunks[c];
    const u8 *pat = qbytes + qoff[q];
    const u32 plen = (u32)(qoff[q + 1] - qoff[q]);
    u32 lo, hi0;
    sample_window_lane(ch, pat, plen, lo, hi0);
    u32 hi = hi0;                                // lower bound: first suffix not < pattern
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (cmp_suffix(ch.text, ch.n, ch.sa[mid], pat, plen) < 0) lo = mid + 1; else hi = mid;
    }
    const u32 L = lo;
    hi = hi0;                                    // upper bound: first suffix > pattern and not prefixed by it;
    for (u32 step = 1; lo < hi; step <<= 1) {    // galloping from L (most intervals are short)
        const u32 p = lo + step - 1;
        if (p >= hi) break;
        if (cmp_suffix(ch.text, ch.n, ch.sa[p], pat, plen) <= 0) lo = p + 1; else { hi = p; break; }
    }
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (cmp_suffix(ch.text, ch.n, ch.sa[mid], pat, plen) <= 0) lo = mid + 1; else hi = mid;
    }
    lo_out[vq] = L;
    cnt_out[vq] = lo - L;
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
    u32 w0, w1;
    sample_window_wave(ch, pat, plen, w0, w1);
    const u32 L = wave_bound(ch.text, ch.n, ch.sa, pat, plen, w0, w1, false);
    const u32 U = wave_bound(ch.text, ch.n, ch.sa, pat, plen, L, w1, true);
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

// Entry around the hit at text offset di: returns false when an earlier
// occurrence of the query inside the same entry exists (duplicate for the
// per-(query, chunk) dedupe, lib.rs:262,274); else the entry's [start, start+len).
__device__ __forceinline__ bool hit_entry(const ChunkDesc &ch, const u8 *pat, u32 plen, u32 di, u32 &line_start,
                                          u32 &line_len)
{
    const u64 NL = 0x0a0a0a0a0a0a0a0aull;
    // Backwards, 8 bytes at a time, to the entry start (lib.rs:270-273); candidates for an
    // earlier occurrence are the bytes equal to the query's first byte.
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
    if (dup) return false;
    line_start = p;
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
    line_len = e >= line_start ? e - line_start : 0;
    return true;
}

__device__ __forceinline__ void copy_entry(u8 *dst, const u8 *src, u32 l)
{
    u32 i = 0;
    for (; i + 8 <= l; i += 8) {                       // 8 bytes per step, unaligned on both sides
        const u64 v = load_u64_unaligned(src + i);
        __builtin_memcpy(dst + i, &v, 8);
    }
    for (; i < l; ++i) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void hit_lines_kernel(const ChunkDesc *chunks, u32 nc, const u8 *qbytes,
                                                          const u64 *qoff, u64 nvq, const u32 *lo, const u64 *hit_off,
                                                          u64 H, u32 *start_out, u32 *len_out)
{
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
        u32 ls = 0, ll = 0;
        if (hit_entry(ch, pat, plen, di, ls, ll)) {
            start_out[t] = ls;
            len_out[t] = ll;
        } else {
            start_out[t] = 0;
            len_out[t] = kSkip;
        }
    }
}

// ---- fused path for small batches (single-query latency) ----------------------
// One launch does everything for up to SM_MAX_VQ (query, chunk) pairs: interval
// search, entry recovery, dedupe, and packing into a small arena whose space is
// handed out with two atomic cursors.  A pair's entries are contiguous; the host
// re-orders pairs query-major.  Anything that does not fit (arena full, more
// than SM_MAX_HITS hits for one pair) sets the overflow flag and the general
// multi-kernel path runs instead.
constexpr u32 SM_MAX_VQ = 1024;
constexpr u32 SM_MAX_HITS = 1024;
constexpr u32 SM_ENT_CAP = 4096;
constexpr u32 SM_BYTE_PREFIX = 8000;   // result bytes fetched together with the header in the first copy
constexpr u32 SM_BYTE_CAP = 2u << 20;

struct SmallHeader {
    u32 ent_cursor, byte_cursor, overflow, pad;
};
struct SmallRecord {
    u32 ent_start, ent_count;
};
struct SmallEntry {
    u32 byte_off, len;
};

__global__ __launch_bounds__(256) void search_small_kernel(const ChunkDesc *chunks, u32 nc, const u8 *qbytes,
                                                             const u64 *qoff, u32 nvq, SmallHeader *hdr,
                                                             SmallRecord *rec, SmallEntry *ent, u8 *bytes)
{
    const u32 vq = blockIdx.x * (blockDim.x / kWave) + wave_id();
    if (vq >= nvq) return;
    const u32 lane = lane_id();
    const u32 q = vq / nc, c = vq % nc;
    const ChunkDesc ch = chunks[c];
    const u8 *pat = qbytes + qoff[q];
    const u32 plen = (u32)(qoff[q + 1] - qoff[q]);
    u32 w0, w1;
    sample_window_wave(ch, pat, plen, w0, w1);
    const u32 L = wave_bound(ch.text, ch.n, ch.sa, pat, plen, w0, w1, false);
    const u32 U = wave_bound(ch.text, ch.n, ch.sa, pat, plen, L, w1, true);
    const u32 cnt = U - L;
    if (cnt == 0) {
        if (lane == 0) rec[vq] = SmallRecord{0, 0};
        return;
    }
    if (cnt > SM_MAX_HITS) {
        if (lane == 0) hdr->overflow = 1;
        return;
    }
    // pass 1: entry bounds of every hit (kept in LDS), entries / bytes this pair produces
    __shared__ u32 s_ls[256 / kWave][SM_MAX_HITS];
    __shared__ u32 s_ll[256 / kWave][SM_MAX_HITS];
    u32 *my_ls = s_ls[wave_id()], *my_ll = s_ll[wave_id()];
    u32 n_ent = 0, n_bytes = 0;
    for (u32 base = 0; base < cnt; base += kWave) {
        const u32 j = base + lane;
        u32 ls = 0, ll = 0;
        const bool keep = j < cnt && hit_entry(ch, pat, plen, ch.sa[L + j], ls, ll);
        if (j < cnt) {
            my_ls[j] = ls;
            my_ll[j] = keep ? ll : kSkip;
        }
        n_ent += (u32)__popcll(__ballot(keep));
        n_bytes += wave_incl_sum(keep ? ll : 0u);      // lane 63 holds the chunk total
    }
    n_bytes = __shfl(n_bytes, 63);
    u32 e0 = 0, b0 = 0;
    if (lane == 0) {
        e0 = atomicAdd(&hdr->ent_cursor, n_ent);
        b0 = atomicAdd(&hdr->byte_cursor, n_bytes);
        if (e0 + n_ent > SM_ENT_CAP || b0 + n_bytes > SM_BYTE_CAP) hdr->overflow = 1;
        rec[vq] = SmallRecord{e0, n_ent};
    }
    e0 = __shfl(e0, 0);
    b0 = __shfl(b0, 0);
    if (e0 + n_ent > SM_ENT_CAP || b0 + n_bytes > SM_BYTE_CAP) return;
    // pass 2: pack (each lane reads back only what it wrote: no barrier needed)
    for (u32 base = 0; base < cnt; base += kWave) {
        const u32 j = base + lane;
        const u32 ll = j < cnt ? my_ll[j] : kSkip;
        const bool keep = ll != kSkip;
        const u64 km = __ballot(keep);
        const u32 incl = wave_incl_sum(keep ? ll : 0u);
        if (keep) {
            const u32 e = e0 + mbcnt(km);
            const u32 o = b0 + incl - ll;
            ent[e] = SmallEntry{o, ll};
            copy_entry(bytes + o, ch.text + my_ls[j], ll);
        }
        e0 += (u32)__popcll(km);
        b0 += __shfl(incl, 63);
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
        copy_entry(out + o, ch.text + start[t], l);
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

enum SSlot { Q_BYTES = 10, Q_OFF, Q_LO, Q_CNT, Q_HITOFF, Q_START, Q_LEN, Q_EIDX, Q_BOFF, Q_ENTOFF, Q_OUT, Q_SMALL, Q_QCOUNT, Q_ARENA = 28 };

int search_batch_device(DeviceCtx *ctx, const ChunkDesc *d_chunks, u32 nc, const uint8_t *qbytes,
                        const uint64_t *qoffsets, uint32_t nq, HostResult *res, pss_search_stats *st, bool counts_only)
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

    struct Events {   // destroyed on every exit path
        hipEvent_t e[3] = {nullptr, nullptr, nullptr};
        ~Events()
        {
            for (hipEvent_t x : e)
                if (x) (void)hipEventDestroy(x);
        }
    } evs;
    for (hipEvent_t &x : evs.e) PSS_HIP(hipEventCreate(&x));
    const hipEvent_t e0 = evs.e[0], e1 = evs.e[1], e2 = evs.e[2];
    const size_t off_bytes = ((size_t)nq + 1) * 8;
    if (qtotal + 32 <= 8192 && off_bytes <= 8192) {
        // tiny batch: stage in pinned memory (pageable H2D copies are synchronous and slow to start)
        u8 *stg = static_cast<u8 *>(ctx->pinned) + 49152;          // last 16 KiB of the pinned scratch
        memcpy(stg, qbytes, qtotal);
        memset(stg + qtotal, 0, 32);
        memcpy(stg + 8192, qoffsets, off_bytes);
        PSS_HIP(hipMemcpyAsync(d_q, stg, qtotal + 32, hipMemcpyHostToDevice, s));
        PSS_HIP(hipMemcpyAsync(d_qoff, stg + 8192, off_bytes, hipMemcpyHostToDevice, s));
    } else {
        PSS_HIP(hipMemsetAsync(d_q + qtotal, 0, 32, s));
        if (qtotal) PSS_HIP(hipMemcpyAsync(d_q, qbytes, qtotal, hipMemcpyHostToDevice, s));
        PSS_HIP(hipMemcpyAsync(d_qoff, qoffsets, off_bytes, hipMemcpyHostToDevice, s));
    }
    PSS_HIP(hipEventRecord(e0, s));
    const u64 waves_per_block = 256 / kWave;
    if (nvq <= SM_MAX_VQ && !counts_only && !getenv("PSS_NO_SMALL_PATH")) {
        // ---- fused small-batch path: one kernel, two small copies ----
        const size_t rec_bytes = (size_t)SM_MAX_VQ * sizeof(SmallRecord);
        const size_t ent_bytes = (size_t)SM_ENT_CAP * sizeof(SmallEntry);
        PSS_TRY(ctx->slot[Q_ARENA].reserve(64 + rec_bytes + ent_bytes + SM_BYTE_CAP + 64));
        u8 *arena = ctx->slot[Q_ARENA].as<u8>();
        SmallHeader *d_hdr = reinterpret_cast<SmallHeader *>(arena);
        SmallRecord *d_rec = reinterpret_cast<SmallRecord *>(arena + 64);
        SmallEntry *d_ent = reinterpret_cast<SmallEntry *>(arena + 64 + rec_bytes);
        u8 *d_bytes = arena + 64 + rec_bytes + ent_bytes;
        PSS_HIP(hipMemsetAsync(d_hdr, 0, 64, s));
        hipLaunchKernelGGL(search_small_kernel, dim3((u32)((nvq + waves_per_block - 1) / waves_per_block)), dim3(256), 0,
                           s, d_chunks, nc, d_q, d_qoff, (u32)nvq, d_hdr, d_rec, d_ent, d_bytes);
        PSS_HIP(hipEventRecord(e2, s));
        // first copy: header + records + entry table + the first SM_BYTE_PREFIX result bytes
        // (48 KiB of the pinned scratch); most small results need nothing more
        u8 *h_arena = static_cast<u8 *>(ctx->pinned);
        const size_t prefix = 64 + rec_bytes + ent_bytes + SM_BYTE_PREFIX;
        static_assert(64 + SM_MAX_VQ * sizeof(SmallRecord) + SM_ENT_CAP * sizeof(SmallEntry) + SM_BYTE_PREFIX <= 49152,
                      "first copy must fit the pinned scratch");
        PSS_HIP(hipMemcpyAsync(h_arena, arena, prefix, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
        const SmallHeader hh = *reinterpret_cast<SmallHeader *>(h_arena);
        if (!hh.overflow) {
            const u32 E = hh.ent_cursor, B = hh.byte_cursor;
            const SmallEntry *h_ent = reinterpret_cast<const SmallEntry *>(h_arena + 64 + rec_bytes);
            std::vector<u8> h_more;
            const u8 *h_bytes = h_arena + 64 + rec_bytes + ent_bytes;
            if (B > SM_BYTE_PREFIX) {
                h_more.resize(B);
                PSS_HIP(hipMemcpyAsync(h_more.data(), d_bytes, B, hipMemcpyDeviceToHost, s));
                PSS_HIP(hipStreamSynchronize(s));
                h_bytes = h_more.data();
            }
            const SmallRecord *h_rec = reinterpret_cast<const SmallRecord *>(h_arena + 64);
            res->offsets = (u64 *)malloc(((size_t)E + 1) * sizeof(u64));
            res->bytes = (u8 *)malloc(B ? B : 1);
            if (!res->offsets || !res->bytes) return PSS_ENOMEM;
            u64 e_out = 0, b_out = 0;
            for (u64 vq = 0; vq < nvq; ++vq) {               // pairs in (query, chunk) order
                const SmallRecord r = h_rec[vq];
                res->qcount[vq / nc] += r.ent_count;
                for (u32 k = 0; k < r.ent_count; ++k) {
                    const SmallEntry en = h_ent[r.ent_start + k];
                    res->offsets[e_out++] = b_out;
                    memcpy(res->bytes + b_out, h_bytes + en.byte_off, en.len);
                    b_out += en.len;
                }
            }
            res->offsets[e_out] = b_out;
            res->n_entries = e_out;
            st->entries = e_out;
            st->result_bytes = b_out;
            st->hits = e_out;   // hits before dedupe are not counted on this path
            float ms = 0.f;
            PSS_HIP(hipEventElapsedTime(&ms, e0, e2));
            st->ms_device = ms;
            st->ms_interval = ms;
            return PSS_OK;
        }
        // overflow: fall through to the general path (qcount is still all zero)
        for (u32 i = 0; i < nq; ++i) res->qcount[i] = 0;
    }
    if (nvq >= 32768 && !getenv("PSS_WAVE_SEARCH"))
        hipLaunchKernelGGL(search_interval_lane_kernel, dim3((u32)((nvq + 255) / 256)), dim3(256), 0, s, d_chunks, nc,
                           d_q, d_qoff, nvq, d_lo, d_cnt);
    else
        hipLaunchKernelGGL(search_interval_kernel, dim3((u32)((nvq + waves_per_block - 1) / waves_per_block)), dim3(256),
                           0, s, d_chunks, nc, d_q, d_qoff, nvq, d_lo, d_cnt);
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
        if (counts_only) {
            // entries per query without materialising one: interval search, dedupe flags, two scans
            hipLaunchKernelGGL(query_counts_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, nc, nq, d_hitoff, d_eidx,
                               d_qcount);
            PSS_HIP(hipEventRecord(e2, s));
            PSS_HIP(hipMemcpyAsync(res->qcount, d_qcount, (size_t)nq * 8, hipMemcpyDeviceToHost, s));
            PSS_HIP(hipStreamSynchronize(s));
            res->offsets = (u64 *)calloc(1, sizeof(u64));
            if (!res->offsets) return PSS_ENOMEM;
            st->entries = h_small[0];
            float msc = 0.f;
            PSS_HIP(hipEventElapsedTime(&msc, e0, e2));
            st->ms_device = msc;
            PSS_HIP(hipEventElapsedTime(&msc, e0, e1));
            st->ms_interval = msc;
            return PSS_OK;
        }
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
    return PSS_OK;
}

}  // namespace pss
