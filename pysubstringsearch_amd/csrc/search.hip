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
#include "radix_sort.h"

#include <chrono>
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
// 8 bytes at the unaligned address p of the TEXT as two aligned 8-byte loads (every lane probes its
// own random suffix, and a scattered load costs the address unit a cycle per lane and instruction
// whatever its width: 2 instead of 3).  Reads up to 15 bytes past p (the text is zero padded).
__device__ __forceinline__ u64 load_text8(const u8 *p)
{
    const uintptr_t a = (uintptr_t)p;
    const u64 *q = reinterpret_cast<const u64 *>(a & ~(uintptr_t)7);
    const u32 sh = (u32)(a & 7) * 8;
    const u64 w0 = q[0], w1 = q[1];
    return sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
}

// pat0 = the query's first 8 bytes (load_u64_unaligned(pat)), fetched once per pair by the caller
// instead of once per probe
__device__ __forceinline__ int cmp_suffix(const u8 *text, u32 n, u32 s, const u8 *pat, u32 plen, u64 pat0)
{
    const u32 avail = n - s;
    const u32 L = plen < avail ? plen : avail;
    u32 i = 0;
    while (i < L) {
        u64 a = load_text8(text + s + i);
        u64 b = i ? load_u64_unaligned(pat + i) : pat0;
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
__device__ __forceinline__ int cmp_suffix(const u8 *text, u32 n, u32 s, const u8 *pat, u32 plen)
{
    return cmp_suffix(text, n, s, pat, plen, load_u64_unaligned(pat));
}
// The same with the suffix's first 8 bytes already fetched (a0 = load_text8(text + s)): two probes of one lane can have
// their loads in flight together.
__device__ __forceinline__ int cmp_suffix_a0(const u8 *text, u32 n, u32 s, const u8 *pat, u32 plen, u64 pat0, u64 a0)
{
    const u32 avail = n - s;
    const u32 L = plen < avail ? plen : avail;
    u32 i = 0;
    while (i < L) {
        u64 a = i ? load_text8(text + s + i) : a0;
        u64 b = i ? load_u64_unaligned(pat + i) : pat0;
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

// Both bounds of the query's interval inside [lo, hi) at once: a probe that says "smaller", "starts with the query" or
// "larger" serves both searches, and while the two bounds sit in the same gap between probes (a query with no or few
// hits: nearly always) one chain of dependent loads finds both -- half the round trips of two wave_bound calls.
__device__ __forceinline__ void wave_bounds(const u8 *text, u32 n, const u32 *sa, const u8 *pat, u32 plen, u32 lo, u32 hi,
                                            u32 &L, u32 &U)
{
    const u32 lane = lane_id();
    const u64 pat0 = load_u64_unaligned(pat);
    while (hi > lo) {
        const u32 s = hi - lo;
        if (s <= kWave) {
            int c = 1;
            if (lane < s) c = cmp_suffix(text, n, sa[lo + lane], pat, plen, pat0);
            L = lo + (u32)__popcll(__ballot(c < 0));
            U = lo + (u32)__popcll(__ballot(c <= 0));
            return;
        }
        const u32 p = lo + (u32)(((u64)(lane + 1) * s) / (kWave + 1));
        const int c = cmp_suffix(text, n, sa[p], pat, plen, pat0);
        const u32 kl = (u32)__popcll(__ballot(c < 0)), ku = (u32)__popcll(__ballot(c <= 0));
        const u32 l_lo = (kl == 0) ? lo : (u32)__shfl((int)p, (int)kl - 1) + 1;
        const u32 u_hi = (ku == kWave) ? hi : (u32)__shfl((int)p, (int)ku);
        if (kl == ku) {
            lo = l_lo;
            hi = u_hi;
            continue;
        }
        // the bounds part: L in (p[kl - 1], p[kl]], U in (p[ku - 1], p[ku]]
        const u32 l_hi = (u32)__shfl((int)p, (int)kl), u_lo = (u32)__shfl((int)p, (int)ku - 1) + 1;
        if (l_hi - l_lo <= kWave && u_hi - u_lo <= kWave) {
            // both gaps fit a wave (they do whenever the window came from the key samples): one round trip for the two
            // of them -- every lane fetches a suffix of each gap before it looks at either (round 4: a query with hits
            // paid the second chain of loads, ~1 us)
            const bool inl = lane < l_hi - l_lo, inu = lane < u_hi - u_lo;
            const u32 sl = inl ? sa[l_lo + lane] : 0u, su = inu ? sa[u_lo + lane] : 0u;
            const u64 tl = inl ? load_text8(text + sl) : 0ull, tu = inu ? load_text8(text + su) : 0ull;
            const int cl = inl ? cmp_suffix_a0(text, n, sl, pat, plen, pat0, tl) : 1;
            const int cu = inu ? cmp_suffix_a0(text, n, su, pat, plen, pat0, tu) : 1;
            L = l_lo + (u32)__popcll(__ballot(cl < 0));
            U = u_lo + (u32)__popcll(__ballot(cu <= 0));
            return;
        }
        L = wave_bound(text, n, sa, pat, plen, l_lo, l_hi, false);
        U = wave_bound(text, n, sa, pat, plen, u_lo, u_hi, true);
        return;
    }
    L = U = lo;
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
    const u64 pat0 = load_u64_unaligned(pat);
    u32 lo, hi0;
    sample_window_lane(ch, pat, plen, lo, hi0);
    u32 hi = hi0;                                // lower bound: first suffix not < pattern
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (cmp_suffix(ch.text, ch.n, ch.sa[mid], pat, plen, pat0) < 0) lo = mid + 1; else hi = mid;
    }
    const u32 L = lo;
    hi = hi0;                                    // upper bound: first suffix > pattern and not prefixed by it;
    for (u32 step = 1; lo < hi; step <<= 1) {    // galloping from L (most intervals are short)
        const u32 p = lo + step - 1;
        if (p >= hi) break;
        if (cmp_suffix(ch.text, ch.n, ch.sa[p], pat, plen, pat0) <= 0) lo = p + 1; else { hi = p; break; }
    }
    while (lo < hi) {
        const u32 mid = lo + ((hi - lo) >> 1);
        if (cmp_suffix(ch.text, ch.n, ch.sa[mid], pat, plen, pat0) <= 0) lo = mid + 1; else hi = mid;
    }
    lo_out[vq] = L;
    cnt_out[vq] = lo - L;
}

// Mid-size batches (thousands of pairs): SG = 16 lanes per (query, chunk) pair, four pairs per
// wavefront, 17-ary searches.  One wave per pair finishes in the fewest dependent steps but probes 64
// suffixes per step -- with ten thousand pairs in flight the address units and the random sectors are
// the limit, not the latency; 16 lanes per pair need ~1.5x the steps for a quarter of the probes
// (10 000 queries on a 512 MiB chunk: 76 -> ~40 us).  All 64 lanes run every step together; the
// groups of a wave that are done idle.
constexpr u32 SG = 16;

// first index in [lo, hi) for which before(i) is false (before is monotone: true ... true false ...);
// `near`: look at the SG positions right behind lo first (short intervals end there in one step)
template <typename Before>
__device__ __forceinline__ u32 group_search(u32 lo, u32 hi, bool active, bool near, u32 gl, u32 gbase, Before before)
{
    bool done = !active;
    u32 res = lo;
    for (;;) {
        if (!done && hi <= lo) {
            res = lo;
            done = true;
        }
        const bool work = !done;
        if (__ballot(work) == 0) break;
        const u32 s = work ? hi - lo : 0;
        const bool last = s <= SG;                  // the remaining range fits the group: one probe each
        const bool first_near = near && !last;
        u32 p = lo;
        bool b = false;
        if (work) {
            p = (last || first_near) ? lo + gl : lo + (u32)(((u64)(gl + 1) * s) / (SG + 1));
            b = (!last || gl < s) && before(p);
        }
        const u32 k = (u32)__popc((u32)((__ballot(b) >> gbase) & 0xffffu));
        const u32 p_below = (u32)__shfl((int)p, (int)(gbase + (k ? k - 1 : 0)));     // last probe that is "before"
        const u32 p_at = (u32)__shfl((int)p, (int)(gbase + (k < SG ? k : SG - 1)));   // first probe that is not
        if (work) {
            if (last) {
                res = lo + k;
                done = true;
            } else if (first_near) {
                if (k < SG) {
                    res = lo + k;
                    done = true;
                } else {
                    lo += SG;
                }
            } else {
                const u32 nlo = k ? p_below + 1 : lo;
                hi = (k == SG) ? hi : p_at;
                lo = nlo;
            }
        }
        near = false;
    }
    return res;
}

__global__ __launch_bounds__(256) void search_interval_group_kernel(const ChunkDesc *chunks, u32 nc, const u8 *qbytes,
                                                                      const u64 *qoff, u64 nvq, u32 *lo_out, u32 *cnt_out)
{
    const u32 lane = lane_id(), gl = lane & (SG - 1), gbase = lane & ~(SG - 1);
    const u64 vq0 = ((u64)blockIdx.x * (blockDim.x / kWave) + wave_id()) * (kWave / SG);
    if (vq0 >= nvq) return;                                   // wave-uniform
    const u64 vq = vq0 + (lane / SG);
    const bool active = vq < nvq;
    const u64 vqc = active ? vq : vq0;                        // idle groups read valid memory
    const u32 q = (u32)(vqc / nc), c = (u32)(vqc % nc);
    const ChunkDesc ch = chunks[c];
    const u8 *pat = qbytes + qoff[q];
    const u32 plen = (u32)(qoff[q + 1] - qoff[q]);
    u32 w0 = 0, w1 = ch.n;
    const bool tab = active && ch.skeys != nullptr;
    {
        const u32 ns = (u32)(((u64)ch.n + (1u << ch.shift) - 1) >> ch.shift);
        u64 k_lo, k_hi;
        query_keys(pat, plen, k_lo, k_hi);
        const u32 ja = group_search(0, ns, tab, false, gl, gbase, [&](u32 j) { return ch.skeys[j] < k_lo; });
        const u32 jb = group_search(ja, ns, tab, true, gl, gbase, [&](u32 j) { return ch.skeys[j] <= k_hi; });
        if (tab) {
            w0 = ja ? (ja - 1) << ch.shift : 0u;
            w1 = jb < ns ? jb << ch.shift : ch.n;
        }
    }
    const u64 pat0 = load_u64_unaligned(pat);
    const u32 L = group_search(w0, w1, active, false, gl, gbase,
                               [&](u32 i) { return cmp_suffix(ch.text, ch.n, ch.sa[i], pat, plen, pat0) < 0; });
    const u32 U = group_search(L, w1, active, true, gl, gbase,
                               [&](u32 i) { return cmp_suffix(ch.text, ch.n, ch.sa[i], pat, plen, pat0) <= 0; });
    if (active && gl == 0) {
        lo_out[vq] = L;
        cnt_out[vq] = U - L;
    }
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
    u32 L, U;
    wave_bounds(ch.text, ch.n, ch.sa, pat, plen, w0, w1, L, U);
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

// 64 bytes starting at the (unaligned) address p as eight little-endian words, fetched as the
// five aligned 16-byte pieces around them: scattered 4-byte loads cost the address unit one
// cycle per lane and instruction, so the scans below move 64 bytes with 5 load instructions
// instead of 24.  Reads up to 16 bytes before and 79 bytes after p.
__device__ __forceinline__ void load_words64(const u8 *p, u64 (&out)[8])
{
    const uintptr_t a = (uintptr_t)p;
    const uint4 *q = reinterpret_cast<const uint4 *>(a & ~(uintptr_t)15);
    u64 w[10];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const uint4 v = q[i];
        w[2 * i] = (u64)v.x | ((u64)v.y << 32);
        w[2 * i + 1] = (u64)v.z | ((u64)v.w << 32);
    }
    const bool up = (a & 8) != 0;
    const u32 s8 = (u32)(a & 7) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u64 lo = up ? w[k + 1] : w[k];
        const u64 hi = up ? w[k + 2] : w[k + 1];
        out[k] = s8 ? (lo >> s8) | (hi << (64 - s8)) : lo;
    }
}

// Entry around the hit at text offset di: returns false when an earlier
// occurrence of the query inside the same entry exists (duplicate for the
// per-(query, chunk) dedupe, lib.rs:262,274); else the entry's [start, start+len).
__device__ __forceinline__ bool hit_entry(const ChunkDesc &ch, const u8 *pat, u32 plen, u32 di, u32 &line_start,
                                          u32 &line_len)
{
    const u64 NL = 0x0a0a0a0a0a0a0a0aull;
    // The scans are chains of dependent loads from a random place in the text, and the slowest lane
    // of a wave sets the pace, so they move HE_W * 8 = 64 bytes per step and the first step forwards
    // is issued before the backward scan starts.  text is zero padded 128 bytes past n (a step from
    // e < n reads at most 79 bytes ahead).
    constexpr int HE_W = 8;
    u64 fw[HE_W];
    load_words64(ch.text + di, fw);
    // Backwards to the entry start (lib.rs:270-273); candidates for an earlier occurrence are
    // the bytes equal to the query's first byte.  A candidate is checked against the query's first
    // min(8, plen) bytes IN REGISTERS (the word just scanned plus the word after it in the text, which
    // the scan has seen before); only a query longer than 8 bytes whose first 8 match goes back to
    // memory.  (Checking every candidate with a load made the slowest lane of a wave pay one memory
    // round trip per candidate and word: 20 us for 70 hits.)
    const u64 first = plen ? 0x0101010101010101ull * pat[0] : 0;
    const u64 pmask = plen >= 8 ? ~0ull : (plen ? (1ull << (8 * plen)) - 1ull : 0ull);
    const u64 pk = load_u64_unaligned(pat) & pmask;            // pat is readable 16 bytes past its end
    u64 nxt = fw[0];       // the 8 bytes behind the word being scanned: text[p, p + 8)
    u32 p = di;            // scan frontier: bytes [p, di) hold no newline
    bool dup = false, at_start = false;
    // one 8-byte word w = text[p-8, p): byte k of w = text[p-8+k]
    auto word = [&](u64 w) {
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
                const u64 x = k ? (w >> (8 * k)) | (nxt << (64 - 8 * k)) : w;     // text[p-8+k, p+k)
                if ((x & pmask) == pk) dup = plen <= 8 || cmp_suffix(ch.text, ch.n, p - 8 + k, pat, plen) == 0;
            }
        }
        nxt = w;
        p = at_start ? p - 8 + keep_from : p - 8;
    };
    while (p >= 8 * HE_W && !dup && !at_start) {
        u64 w[HE_W];                                             // w[j] = text[p - 64 + 8 j, + 8)
        load_words64(ch.text + p - 8 * HE_W, w);
#pragma unroll
        for (int k = HE_W - 1; k >= 0; --k)
            if (!dup && !at_start) word(w[k]);
    }
    while (p >= 8 && !dup && !at_start) word(load_u64_unaligned(ch.text + p - 8));
    while (p > 0 && !dup && !at_start) {                         // the first < 8 bytes of the chunk
        const u8 cb = ch.text[p - 1];
        if (cb == '\n') break;
        --p;
        dup = plen == 0 || (cb == pat[0] && cmp_suffix(ch.text, ch.n, p, pat, plen) == 0);
    }
    if (dup) return false;
    line_start = p;
    // forwards to the entry end (lib.rs:266-269; no newline: len - 1)
    u32 e = di;
    for (;;) {
        bool found = false;
#pragma unroll
        for (int k = 0; k < HE_W; ++k) {
            const u64 nlm = zero_bytes(fw[k] ^ NL);
            if (nlm && !found) {
                e += 8 * k + (u32)(__builtin_ctzll(nlm) >> 3);
                found = true;
            }
        }
        if (found) break;
        e += 8 * HE_W;
        if (e >= ch.n) break;
        load_words64(ch.text + e, fw);
    }
    if (e >= ch.n) e = ch.n - 1;
    line_len = e >= line_start ? e - line_start : 0;
    return true;
}

// The entry around text offset di, bounds only (lib.rs:266-273): [start, start + len).  Whether the hit is the first
// of its entry is the caller's business (block_pair settles that among the hits themselves).  Both 64-byte loads --
// behind the hit and ahead of it -- are issued before either is looked at: one round trip for entries of < 64 bytes
// on each side.
__device__ __forceinline__ void entry_bounds(const ChunkDesc &ch, u32 di, u32 &line_start, u32 &line_len)
{
    const u64 NL = 0x0a0a0a0a0a0a0a0aull;
    u64 fw[8], bw[8];
    load_words64(ch.text + di, fw);
    u32 p = di;
    if (p >= 64) load_words64(ch.text + p - 64, bw);
    for (;;) {
        if (p < 64) {                                            // the first bytes of the chunk
            while (p > 0 && ch.text[p - 1] != '\n') --p;
            break;
        }
        bool found = false;
#pragma unroll
        for (int k = 7; k >= 0; --k) {
            const u64 nlm = zero_bytes(bw[k] ^ NL);
            if (nlm && !found) {
                p = p - 64 + 8 * k + (u32)((63 - __builtin_clzll(nlm)) >> 3) + 1;
                found = true;
            }
        }
        if (found) break;
        p -= 64;
        if (p >= 64) load_words64(ch.text + p - 64, bw);
    }
    line_start = p;
    u32 e = di;
    for (;;) {
        bool found = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u64 nlm = zero_bytes(fw[k] ^ NL);
            if (nlm && !found) {
                e += 8 * k + (u32)(__builtin_ctzll(nlm) >> 3);
                found = true;
            }
        }
        if (found) break;
        e += 64;
        if (e >= ch.n) break;
        load_words64(ch.text + e, fw);
    }
    if (e >= ch.n) e = ch.n - 1;
    line_len = e >= line_start ? e - line_start : 0;
}

// Copies one entry (l bytes, unaligned on both sides) to dst and, when dst2 is given, to a second
// destination from the same loads.  64 bytes per step: eight independent loads are in flight at
// once, so a long entry costs a few memory round trips instead of one per 8 bytes (the slowest
// lane of a wave sets the pace: 150-byte entries took 19 dependent steps).
__device__ __forceinline__ void copy_entry(u8 *dst, const u8 *src, u32 l, u8 *dst2 = nullptr)
{
    u32 i = 0;
    for (; i + 64 <= l; i += 64) {
        u64 v[8];
        load_words64(src + i, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) __builtin_memcpy(dst + i + 8 * k, &v[k], 8);
        if (dst2) {
#pragma unroll
            for (int k = 0; k < 8; ++k) __builtin_memcpy(dst2 + i + 8 * k, &v[k], 8);
        }
    }
    if (i < l) {
        // tail of < 64 bytes: the same loads (the text is readable 128 bytes past its end), stores by length
        u64 v[8];
        load_words64(src + i, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u32 at = i + 8 * k;
            if (at + 8 <= l) {
                __builtin_memcpy(dst + at, &v[k], 8);
                if (dst2) __builtin_memcpy(dst2 + at, &v[k], 8);
            } else if (at < l) {
                for (u32 b = 0; b < l - at; ++b) {
                    const u8 x = (u8)(v[k] >> (8 * b));
                    dst[at + b] = x;
                    if (dst2) dst2[at + b] = x;
                }
            }
        }
    }
}

// ---- mid pipeline: batches of <= MID_MAX pairs with <= MID_MAX hits -----------------------
// The general pipeline waits for the host twice in the middle (hit total, then entry / byte totals:
// sizes of the next allocations and grids) and spends nine launches on three scans.  When the pairs
// and the hits fit one workgroup's scan, the totals stay on the device (MidState), every scan is one
// launch, the buffers are sized for the caps, and the host waits once for the totals and once for the
// result.  More hits or bytes than the caps raise `flag`; the later kernels then do nothing and the
// general pipeline runs instead.
constexpr u32 MID_MAX = 65536;
constexpr u32 MID_BLOCK = 1024;
struct MidState {
    u64 hits, entries, bytes;
    u32 flag, pad;
};

// exclusive sum of one value per thread over a MID_BLOCK workgroup (u64); *total = block sum
__device__ __forceinline__ u64 mid_block_excl(u64 v, u64 *scr /* [16] */, u64 *total)
{
    const u64 incl = wave_incl_sum64(v);
    if (lane_id() == kWave - 1) scr[wave_id()] = incl;
    __syncthreads();
    u64 base = 0, tot = 0;
#pragma unroll
    for (u32 w = 0; w < MID_BLOCK / kWave; ++w) {
        const u64 x = scr[w];
        if (w < (u32)wave_id()) base += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// hit counts -> hit offsets (nvq + 1 entries); total and overflow flag into MidState
__global__ __launch_bounds__(MID_BLOCK) void mid_hitoff_kernel(const u32 *cnt, u32 nvq, u64 *hit_off, MidState *ms)
{
    __shared__ u64 scr[MID_BLOCK / kWave];
    const u32 per = (nvq + MID_BLOCK - 1) / MID_BLOCK;
    const u32 i0 = min(threadIdx.x * per, nvq), i1 = min(i0 + per, nvq);
    u64 local = 0;
    for (u32 i = i0; i < i1; ++i) local += cnt[i];
    u64 total;
    u64 run = mid_block_excl(local, scr, &total);
    for (u32 i = i0; i < i1; ++i) {
        hit_off[i] = run;
        run += cnt[i];
    }
    if (threadIdx.x == 0) {
        hit_off[nvq] = total;
        ms->hits = total;
        ms->entries = 0;
        ms->bytes = 0;
        ms->flag = total > MID_MAX ? 1u : 0u;
    }
}

// kept flags -> entry index, entry lengths -> byte offset (both hits + 1 entries), in one launch
__global__ __launch_bounds__(MID_BLOCK) void mid_hit_scans_kernel(const u32 *len, MidState *ms, u64 byte_cap, u64 *eidx,
                                                                    u64 *boff)
{
    __shared__ u64 scr[MID_BLOCK / kWave];
    if (ms->flag) return;
    const u32 H = (u32)ms->hits;
    const u32 per = (H + MID_BLOCK - 1) / MID_BLOCK;
    const u32 i0 = min(threadIdx.x * per, H), i1 = min(i0 + per, H);
    u64 le = 0, lb = 0;
    for (u32 i = i0; i < i1; ++i) {
        const u32 l = len[i];
        if (l != kSkip) {
            ++le;
            lb += l;
        }
    }
    u64 te, tb;
    u64 re = mid_block_excl(le, scr, &te);
    u64 rb = mid_block_excl(lb, scr, &tb);
    for (u32 i = i0; i < i1; ++i) {
        const u32 l = len[i];
        eidx[i] = re;
        boff[i] = rb;
        if (l != kSkip) {
            ++re;
            rb += l;
        }
    }
    if (threadIdx.x == 0) {
        eidx[H] = te;
        boff[H] = tb;
        ms->entries = te;
        ms->bytes = tb;
        if (tb > byte_cap) ms->flag = 1u;
    }
}

__global__ __launch_bounds__(256) void hit_lines_kernel(const ChunkDesc *chunks, u32 nc, const u8 *qbytes,
                                                          const u64 *qoff, u64 nvq, const u32 *lo, const u64 *hit_off,
                                                          u64 H, const MidState *mid, u32 *start_out, u32 *len_out)
{
    if (mid) H = mid->flag ? 0 : mid->hits;      // mid pipeline: the hit count never visits the host
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
// One launch does everything for up to SM_MAX_VQ (query, chunk) pairs: interval search, entry
// recovery, dedupe, and packing into a small arena whose space is handed out with two atomic
// cursors.  A pair's entries are contiguous; the host re-orders pairs query-major.
// No copy engine is involved: the kernel READS the queries from pinned host memory (once, into
// LDS) and WRITES the per-pair records, the entry table and the first SM_BYTE_PREFIX result
// bytes straight into pinned host memory, so a call is memcpy -> launch -> wait.  Results larger
// than the prefix are fetched from the device arena with one copy.  Anything that does not fit
// (arena full, more than SM_MAX_HITS hits for one pair) sets the overflow flag and the general
// multi-kernel path runs instead.
constexpr u32 SM_MAX_VQ = 1024;
constexpr u32 SM_MAX_HITS = 1024;
constexpr u32 SM_ENT_CAP = 65536;
#ifndef PSS_SM_BYTE_PREFIX
#define PSS_SM_BYTE_PREFIX 65536
#endif
constexpr u32 SM_BYTE_PREFIX = PSS_SM_BYTE_PREFIX;   // result bytes that also go to pinned host memory
constexpr u32 SM_BYTE_CAP = 8u << 20;
constexpr u32 SM_MAX_PLEN = 256;       // longest query the path takes (kept in LDS)

struct SmallHeader {   // device memory; all zero between launches (the last wave to finish resets it)
    u32 ent_cursor, byte_cursor, done;
    u32 leave;          // resident kernel: its first workgroup tells the others to leave
    u32 echo;           // resident kernel: SUM (mod 2^32) of the workgroups' query checksums (ResidentMailbox::echo)
};
struct SmallRecord {
    u32 ent_start, ent_count;
};
// Space for n_ent entries and n_bytes result bytes: both cursors move with ONE atomic (they share a 64-bit word), one
// round trip to L2 instead of two in a row.
__device__ __forceinline__ void small_take(SmallHeader *hdr, u32 n_ent, u32 n_bytes, u32 &e0, u32 &b0)
{
    static_assert(offsetof(SmallHeader, ent_cursor) == 0 && offsetof(SmallHeader, byte_cursor) == 4, "the cursors share a word");
    const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long *>(hdr), (unsigned long long)n_ent | ((unsigned long long)n_bytes << 32));
    e0 = (u32)old;
    b0 = (u32)(old >> 32);
}
struct SmallEntry {
    u32 byte_off, len;
};
// layout of the pinned scratch (DeviceCtx::pinned) on this path
constexpr size_t SM_OFF_FLAGS = 0;                                              // u32 overflow
constexpr size_t SM_OFF_REC = 64;                                               // 2048 records (one per pair, or per (pair, sub-block))
constexpr size_t SM_OFF_MAILBOX = DeviceCtx::kResidentMailboxOff;               // resident kernel only: ResidentMailbox
constexpr size_t SM_OFF_QUERY = 32768;                                          // query bytes, then offsets at + 8192
constexpr size_t SM_OFF_BYTES = 65536;                                          // first SM_BYTE_PREFIX result bytes
constexpr size_t SM_OFF_ENT = SM_OFF_BYTES + SM_BYTE_PREFIX;                    // entry table
static_assert(SM_OFF_REC + 2048 * sizeof(SmallRecord) <= SM_OFF_QUERY && SM_MAX_VQ <= 2048, "records must fit below the query staging");
static_assert(SM_OFF_QUERY + 16384 <= SM_OFF_BYTES, "query staging must fit below the result prefix");
static_assert(SM_OFF_REC + 2048 * sizeof(SmallRecord) <= SM_OFF_MAILBOX && SM_OFF_MAILBOX + sizeof(ResidentMailbox) <= SM_OFF_QUERY,
              "the mailbox sits between the records and the query staging");
static_assert(sizeof(ResidentMailbox::query) >= SM_MAX_PLEN + 32 && offsetof(ResidentMailbox, query) % 8 == 0,
              "the mailbox holds the longest query of the path, padded");
static_assert(SM_OFF_ENT + SM_ENT_CAP * sizeof(SmallEntry) <= DeviceCtx::kPinnedBytes, "entry table must fit the pinned scratch");

__device__ __forceinline__ void small_pair(const ChunkDesc &ch, const u8 *pat, u32 plen, u32 vq, SmallHeader *hdr,
                                           u32 *h_overflow, SmallRecord *rec, SmallEntry *ent, u8 *bytes, u8 *hbytes,
                                           u32 *my_ls, u32 *my_ll)
{
    const u32 lane = lane_id();
    u32 w0, w1;
    sample_window_wave(ch, pat, plen, w0, w1);
    u32 L, U;
    wave_bounds(ch.text, ch.n, ch.sa, pat, plen, w0, w1, L, U);
    const u32 cnt = U - L;
    if (cnt == 0) {
        if (lane == 0) rec[vq] = SmallRecord{0, 0};
        return;
    }
    if (cnt > SM_MAX_HITS) {
        if (lane == 0) *h_overflow = 1;
        return;
    }
    // pass 1: entry bounds of every hit (kept in LDS), entries / bytes this pair produces
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
        small_take(hdr, n_ent, n_bytes, e0, b0);
        if (e0 + n_ent > SM_ENT_CAP || b0 + n_bytes > SM_BYTE_CAP) *h_overflow = 1;
        else rec[vq] = SmallRecord{e0, n_ent};
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
            copy_entry(bytes + o, ch.text + my_ls[j], ll, o + ll <= SM_BYTE_PREFIX ? hbytes + o : nullptr);
        }
        e0 += (u32)__popcll(km);
        b0 += __shfl(incl, 63);
    }
}

// qbytes / qoff / h_overflow / rec / ent / hbytes: pinned host memory; hdr / bytes: device memory
__global__ __launch_bounds__(256) void search_small_kernel(const ChunkDesc *chunks, u32 nc, const u8 *qbytes,
                                                             const u64 *qoff, u32 nvq, SmallHeader *hdr,
                                                             u32 *h_overflow, SmallRecord *rec, SmallEntry *ent,
                                                             u8 *bytes, u8 *hbytes)
{
    __shared__ u32 s_ls[256 / kWave][SM_MAX_HITS];
    __shared__ u32 s_ll[256 / kWave][SM_MAX_HITS];
    __shared__ __attribute__((aligned(8))) u8 s_pat[256 / kWave][SM_MAX_PLEN + 32];
    const u32 vq = blockIdx.x * (blockDim.x / kWave) + wave_id();
    if (vq >= nvq) return;
    const u32 lane = lane_id();
    const u32 q = vq / nc, c = vq % nc;
    const u64 o0 = qoff[q];
    const u32 plen = (u32)(qoff[q + 1] - o0);
    // the query crosses PCIe once; every comparison afterwards reads it from LDS (zero padded)
    u8 *pat = s_pat[wave_id()];
    for (u32 i = lane * 8; i < SM_MAX_PLEN + 32; i += kWave * 8) {
        u64 v = 0;
        if (i < plen) {
            v = load_u64_unaligned(qbytes + o0 + i);
            if (plen - i < 8) v &= (1ull << (8 * (plen - i))) - 1ull;
        }
        *reinterpret_cast<u64 *>(pat + i) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the wave reads what its other lanes wrote
    __builtin_amdgcn_wave_barrier();
    small_pair(chunks[c], pat, plen, vq, hdr, h_overflow, rec, ent, bytes, hbytes, s_ls[wave_id()], s_ll[wave_id()]);
    if (lane == 0 && atomicAdd(&hdr->done, 1u) == nvq - 1) {
        hdr->ent_cursor = 0;       // every pair has taken its space: leave the header zero for the next launch
        hdr->byte_cursor = 0;
        hdr->done = 0;
    }
}

// The same for very few pairs (a single query over <= SM_BLOCK_MAX_VQ chunks): one WORKGROUP
// of 16 wavefronts per pair.  Wave 0 finds the interval; then every thread recovers the entry of
// one hit at a time, so up to 1024 hits cost one round of dependent loads instead of sixteen
// (a query with 245 hits: 137 -> ~45 us of device time).  Entry order inside the pair stays the
// suffix-array order of the kept hits.
constexpr u32 SM_BLOCK = 1024;
constexpr u32 SM_BLOCK_MAX_VQ = 64;
constexpr u32 SM_BLOCK_MAX_HITS = 1024;
// ... and up to SM_SPREAD workgroups per pair: workgroup k of a pair takes hits [1024 k, 1024 (k + 1)) of its
// interval -- one hit per thread, one round of dependent loads -- and every workgroup finds the interval
// itself (a few microseconds, no hand-over), so a single query with up to 32 768 hits per chunk stays on
// this one-kernel path instead of falling back to the multi-kernel pipeline.  The record of (pair, k)
// holds its entries; read in (pair, k) order they are in suffix-array order.
constexpr u32 SM_SPREAD = 32;
constexpr u32 SM_STAGE_BYTES = 48 * 1024;     // LDS stage of a workgroup's result bytes
constexpr u32 SM_MAX_REC = SM_BLOCK_MAX_VQ * SM_SPREAD;          // 2048 records

// One (pair, sub-block) of the block path: the query (plen bytes at qsrc, host memory) against chunk ch, hits
// [1024 sub, 1024 (sub + 1)) of its interval into record ri.
// (Records and the overflow flag go out as system-scope stores: the resident kernel, which does not end, then owes the
// host a write-back of its L2 only when it produced entries.)
__device__ __forceinline__ void put_record(SmallRecord *rec, u32 ent_start, u32 ent_count)
{
    __hip_atomic_store(reinterpret_cast<u64 *>(rec), (u64)ent_start | ((u64)ent_count << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void put_flag(u32 *flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// (Diagnostic build, -DPSS_TRACE_RESIDENT: thread 0 stamps the device clock at the phase borders of a resident query and
// the kernel leaves the stamps in the mailbox's padding; tests/tools/latency_trace.py prints them.)
#ifdef PSS_TRACE_RESIDENT
#define PSS_TR(tr, i) do { if ((tr) && threadIdx.x == 0) (tr)[i] = (u32)wall_clock64(); } while (0)
#else
#define PSS_TR(tr, i) do { } while (0)
#endif
// Returns true when the workgroup wrote entries (entry table and result bytes, ordinary stores) that also goes to pinned host memory (they are
// only in the device arena then).
__device__ __forceinline__ bool block_pair(const ChunkDesc ch, const u8 *s_pat /* LDS, zero padded, visible */, u32 plen, u32 sub,
                                           u32 ri, u32 spread, SmallHeader *hdr, u32 *h_overflow, SmallRecord *rec,
                                           SmallEntry *ent, u8 *bytes, u8 *hbytes, bool alone = false, u32 *tr = nullptr)
{
    // alone: this workgroup is the whole launch (the resident kernel of a one-chunk reader) -- its entries start at 0 without
    // a trip to the cursors, and a result of a few entries goes to the pinned prefix as system-scope stores and nowhere
    // else: nothing is left in L2 for a write-back (the function then returns false)
    __shared__ u32 s_ls[SM_BLOCK_MAX_HITS];
    __shared__ u32 s_ll[SM_BLOCK_MAX_HITS];
    __shared__ u32 s_L, s_cnt, s_e0, s_b0;
    __shared__ u32 s_we[SM_BLOCK / kWave], s_wb[SM_BLOCK / kWave];
    // first hit (smallest text offset) of every entry among the hits of this pair: open-addressing table keyed by the
    // entry's start, 2 slots per possible hit
    constexpr u32 HS = 2 * SM_BLOCK_MAX_HITS;
    __shared__ u32 s_hkey[HS], s_hmin[HS];
    __shared__ __attribute__((aligned(16))) u8 s_stage[SM_STAGE_BYTES];
    const u32 tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    bool wrote = false;
    const u8 *pat = s_pat;
    for (u32 i = tid; i < HS; i += SM_BLOCK) {
        s_hkey[i] = 0;
        s_hmin[i] = 0xffffffffu;
    }
    if (wave == 0) {
        u32 w0, w1;
        sample_window_wave(ch, pat, plen, w0, w1);
        u32 L, U;
        wave_bounds(ch.text, ch.n, ch.sa, pat, plen, w0, w1, L, U);
        if (lane == 0) {
            s_L = L;
            s_cnt = U - L;
        }
    }
    __syncthreads();
    PSS_TR(tr, 1);
    const u32 total = s_cnt;
    const u32 first = sub * SM_BLOCK_MAX_HITS;                 // this workgroup's slice of the interval
    const u32 L = s_L + first;
    const u32 cnt = total > first ? min(total - first, SM_BLOCK_MAX_HITS) : 0u;
    static_assert(SM_MAX_REC <= 2048, "one record per (pair, sub-block)");
    if (total > spread * SM_BLOCK_MAX_HITS) {
        if (tid == 0) put_flag(h_overflow);
    } else if (cnt == 0) {
        if (tid == 0) put_record(rec + ri, 0, 0);
    } else {
        // pass 1: entry bounds of every hit, one hit per thread
        static_assert(SM_BLOCK_MAX_HITS == SM_BLOCK, "one hit per thread");
        if (total <= SM_BLOCK_MAX_HITS) {
            // This workgroup holds every hit of the pair, so "an earlier occurrence of the query in the same entry"
            // (lib.rs:262,274: one result per entry) is a question about the hits themselves: the hit with the
            // smallest text offset among those with the same entry start stays.  Scanning each entry for earlier
            // occurrences instead (hit_entry) is what made 245 hits cost 16 us here: divergent candidate loops, a
            // wave at the pace of the union of its lanes' paths.
            u32 di = 0, ls = 0, ll = 0, slot = 0;
            if (tid < cnt) {
                di = ch.sa[L + tid];
                entry_bounds(ch, di, ls, ll);
                slot = (ls * 2654435761u) >> (32 - 11);
                static_assert(HS == 2048, "11-bit hash");
                for (;;) {
                    const u32 old = atomicCAS(&s_hkey[slot], 0u, ls + 1u);
                    if (old == 0u || old == ls + 1u) break;
                    slot = (slot + 1) & (HS - 1);
                }
                atomicMin(&s_hmin[slot], di);
                s_ls[tid] = ls;
            }
            __syncthreads();
            if (tid < cnt) s_ll[tid] = (s_hmin[slot] == di) ? ll : kSkip;
        } else {
            for (u32 j = tid; j < cnt; j += SM_BLOCK) {
                u32 ls = 0, ll = 0;
                const bool keep = hit_entry(ch, pat, plen, ch.sa[L + j], ls, ll);
                s_ls[j] = ls;
                s_ll[j] = keep ? ll : kSkip;
            }
        }
        __syncthreads();
        PSS_TR(tr, 2);
        // entries / bytes before each thread's run of consecutive hits (suffix-array order)
        const u32 per = (cnt + SM_BLOCK - 1) / SM_BLOCK;
        const u32 j0 = min(tid * per, cnt), j1 = min(j0 + per, cnt);
        u32 my_e = 0, my_b = 0;
        for (u32 j = j0; j < j1; ++j) {
            const u32 ll = s_ll[j];
            if (ll != kSkip) {
                ++my_e;
                my_b += ll;
            }
        }
        const u32 ie = wave_incl_sum(my_e), ib = wave_incl_sum(my_b);
        if (lane == kWave - 1) {
            s_we[wave] = ie;
            s_wb[wave] = ib;
        }
        __syncthreads();
        u32 e_before = ie - my_e, b_before = ib - my_b, n_ent = 0, n_bytes = 0;
#pragma unroll
        for (u32 w = 0; w < SM_BLOCK / kWave; ++w) {
            if (w < wave) {
                e_before += s_we[w];
                b_before += s_wb[w];
            }
            n_ent += s_we[w];
            n_bytes += s_wb[w];
        }
        if (tid == 0) {
            u32 e0 = 0, b0 = 0;
            if (!alone) small_take(hdr, n_ent, n_bytes, e0, b0);
            s_e0 = e0;
            s_b0 = b0;
            if (e0 + n_ent > SM_ENT_CAP || b0 + n_bytes > SM_BYTE_CAP) put_flag(h_overflow);
            else put_record(rec + ri, e0, n_ent);
        }
        __syncthreads();
        PSS_TR(tr, 3);
        const u32 e0 = s_e0, b0 = s_b0;
        if (e0 + n_ent <= SM_ENT_CAP && b0 + n_bytes <= SM_BYTE_CAP) {
            // (only a few entries: every system-scope store is a transaction of its own on the bus -- 10 KiB of result
            // took 260 us that way, 24 through L2 and one write-back)
            const bool direct = alone && n_bytes <= 512 && n_ent <= 16;
            // pass 2: pack.  When the workgroup's bytes fit the LDS stage, entries are assembled there (unaligned
            // pieces, byte tails) and leave as one run of aligned 16-byte stores per destination; else entry by entry.
            const bool staged = n_bytes <= SM_STAGE_BYTES - 16;
            const u32 skew = b0 & 15u;                          // keeps stage and destinations 16-byte congruent
            const bool pinned_all = b0 + n_bytes <= SM_BYTE_PREFIX;
            // Round 4: by OUTPUT pieces.  One entry per thread meant one chain of loads per entry -- the longest of a few
            // hundred sets the pace, 5.6 us for 241 entries -- and a second trip through LDS.  Here the entries only leave
            // their (offset, source, length) in LDS and mark the 16-byte pieces of the output whose first byte is theirs;
            // then every thread builds whole pieces: its owner's bytes from the text with one unaligned 16-byte load, the
            // next entries' if the piece runs on, and stores them aligned -- one round of loads whatever the lengths.  A
            // workgroup on its own whose bytes fit the pinned prefix writes nothing to the device arena: the host reads the
            // pinned copy only.
            const bool coop = staged && !direct && n_ent != 0;
            const bool arena_too = !(alone && pinned_all);
            wrote = n_ent != 0 && !direct;       // (ordinary stores, to the arena or to pinned memory alike, wait in L2 for the write-back)
            if (coop) {
                static_assert(SM_STAGE_BYTES >= 3 * 4 * SM_BLOCK_MAX_HITS + 2 * (SM_STAGE_BYTES / 16 + 2), "tables in the stage");
                u32 *t_off = reinterpret_cast<u32 *>(s_stage), *t_src = t_off + SM_BLOCK_MAX_HITS, *t_len = t_src + SM_BLOCK_MAX_HITS;
                u16 *t_owner = reinterpret_cast<u16 *>(t_len + SM_BLOCK_MAX_HITS);
                u32 k = e_before, off = b_before;                // entry ordinal / byte offset inside this workgroup's result
                for (u32 j = j0; j < j1; ++j) {
                    const u32 ll = s_ll[j];
                    if (ll == kSkip) continue;
                    ent[e0 + k] = SmallEntry{b0 + off, ll};
                    t_off[k] = off;
                    t_src[k] = s_ls[j];
                    t_len[k] = ll;
                    // pieces whose first valid byte lies in [off, off + ll): piece t starts at byte 16 t - skew (piece 0 at 0)
                    if (ll) {
                        const u32 t0 = off == 0 ? 0u : (off + skew + 15u) / 16u, t1 = (off + ll - 1u + skew) / 16u;
                        for (u32 t = t0; t <= t1; ++t) t_owner[t] = (u16)k;
                    }
                    ++k;
                    off += ll;
                }
                __syncthreads();
                PSS_TR(tr, 4);
                const u32 end = skew + n_bytes, pieces = (end + 15u) / 16u;
                u8 *d0 = bytes + (b0 - skew), *d1 = hbytes + (b0 - skew);
                for (u32 t = tid; t < pieces; t += SM_BLOCK) {
                    const u32 pstart = 16u * t;                                   // in stage coordinates (byte x of the result at skew + x)
                    u32 cur = max(pstart, skew) - skew;                           // result byte the piece starts with
                    const u32 stop = min(pstart + 16u, end) - skew;
                    u32 kk = t_owner[t];
                    u64 lo = 0, hi = 0;
                    // 16 bytes of the text from byte `within` of entry k on (readable past the end of the text)
                    auto load16 = [&](u32 k, u32 within, u64 &x0, u64 &x1) {
                        const uintptr_t a = (uintptr_t)(ch.text + t_src[k] + within);
                        const uint4 *q = reinterpret_cast<const uint4 *>(a & ~(uintptr_t)15);
                        const uint4 v0 = q[0], v1 = q[1];
                        const u64 w0 = (u64)v0.x | ((u64)v0.y << 32), w1 = (u64)v0.z | ((u64)v0.w << 32);
                        const u64 w2 = (u64)v1.x | ((u64)v1.y << 32), w3 = (u64)v1.z | ((u64)v1.w << 32);
                        const bool up = (a & 8) != 0;
                        const u32 s8 = (u32)(a & 7) * 8;
                        const u64 l0 = up ? w1 : w0, l1 = up ? w2 : w1, l2 = up ? w3 : w2;
                        x0 = s8 ? (l0 >> s8) | (l1 << (64 - s8)) : l0;
                        x1 = s8 ? (l1 >> s8) | (l2 << (64 - s8)) : l1;
                    };
                    while (cur < stop) {
                        while (t_off[kk] + t_len[kk] <= cur) ++kk;                // (empty entries, and the ones this piece has used up)
                        const u32 within = cur - t_off[kk];
                        const u32 take = min(stop - cur, t_len[kk] - within);
                        u64 x0, x1;
                        load16(kk, within, x0, x1);
                        // keep `take` bytes, move them to byte (cur + skew - pstart) of the piece
                        if (take < 8) { x0 &= (1ull << (8 * take)) - 1ull; x1 = 0; }
                        else if (take < 16) x1 &= take == 8 ? 0ull : (1ull << (8 * (take - 8))) - 1ull;
                        const u32 sh = (cur + skew - pstart) * 8u;
                        if (sh == 0) { lo |= x0; hi |= x1; }
                        else if (sh < 64) { lo |= x0 << sh; hi |= (x1 << sh) | (x0 >> (64 - sh)); }
                        else { hi |= x0 << (sh - 64); }
                        cur += take;
                    }
                    if (pstart >= skew && pstart + 16u <= end) {
                        const uint4 v = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
                        if (arena_too) *reinterpret_cast<uint4 *>(d0 + pstart) = v;
                        if (pinned_all) *reinterpret_cast<uint4 *>(d1 + pstart) = v;
                    } else {
                        for (u32 b = max(pstart, skew); b < min(pstart + 16u, end); ++b) {
                            const u32 i = b - pstart;
                            const u8 x = (u8)(i < 8 ? lo >> (8 * i) : hi >> (8 * (i - 8)));
                            if (arena_too) d0[b] = x;
                            if (pinned_all) d1[b] = x;
                        }
                    }
                }
            } else {
            u32 e = e0 + e_before, o = b0 + b_before;
            for (u32 j = j0; j < j1; ++j) {
                const u32 ll = s_ll[j];
                if (ll == kSkip) continue;
                if (direct) __hip_atomic_store(reinterpret_cast<u64 *>(ent + e), (u64)o | ((u64)ll << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                else ent[e] = SmallEntry{o, ll};
                ++e;
                if (staged) copy_entry(s_stage + skew + (o - b0), ch.text + s_ls[j], ll);
                else copy_entry(bytes + o, ch.text + s_ls[j], ll, o + ll <= SM_BYTE_PREFIX ? hbytes + o : nullptr);
                o += ll;
            }
            if (staged) {
                __syncthreads();
                PSS_TR(tr, 4);
                const u32 end = skew + n_bytes;                  // stage bytes [skew, end) -> arena bytes [b0, b0 + n_bytes)
                u8 *d0 = bytes + (b0 - skew), *d1 = hbytes + (b0 - skew);
                const bool pinned_too = b0 + n_bytes <= SM_BYTE_PREFIX;   // (else the host takes everything from the device arena)
                if (direct) {
                    // (b0 = 0: whole 16-byte blocks, the last one padded with whatever the stage holds)
                    for (u32 at = tid * 16; at < end; at += SM_BLOCK * 16) {
                        const u64 *sp = reinterpret_cast<const u64 *>(s_stage + at);
                        u64 *dp = reinterpret_cast<u64 *>(hbytes + at);
                        __hip_atomic_store(dp, sp[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        __hip_atomic_store(dp + 1, sp[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                } else
                for (u32 at = tid * 16; at < end; at += SM_BLOCK * 16) {
                    if (at >= skew && at + 16 <= end) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(s_stage + at);
                        *reinterpret_cast<uint4 *>(d0 + at) = v;
                        if (pinned_too) *reinterpret_cast<uint4 *>(d1 + at) = v;
                    } else {
                        for (u32 b = max(at, skew); b < min(at + 16, end); ++b) {
                            d0[b] = s_stage[b];
                            if (pinned_too) d1[b] = s_stage[b];
                        }
                    }
                }
            }
            }
        }
    }
    return wrote;
}

__global__ __launch_bounds__(SM_BLOCK) void search_block_kernel(const ChunkDesc *chunks, u32 nc, const u8 *qbytes,
                                                                  const u64 *qoff, u32 nvq, SmallHeader *hdr,
                                                                  u32 *h_overflow, SmallRecord *rec, SmallEntry *ent,
                                                                  u8 *bytes, u8 *hbytes, u32 spread)
{
    __shared__ __attribute__((aligned(8))) u8 s_pat[SM_MAX_PLEN + 32];
    const u32 vq = blockIdx.x / spread, sub = blockIdx.x % spread;
    const u32 q = vq / nc, c = vq % nc;
    const u64 o0 = qoff[q];
    const u32 plen = (u32)(qoff[q + 1] - o0);
    for (u32 i = threadIdx.x * 8; i < SM_MAX_PLEN + 32; i += SM_BLOCK * 8) {
        u64 v = 0;
        if (i < plen) {
            v = load_u64_unaligned(qbytes + o0 + i);
            if (plen - i < 8) v &= (1ull << (8 * (plen - i))) - 1ull;
        }
        *reinterpret_cast<u64 *>(s_pat + i) = v;
    }
    __syncthreads();
    block_pair(chunks[c], s_pat, plen, sub, blockIdx.x, spread, hdr, h_overflow, rec, ent, bytes, hbytes);
    if (threadIdx.x == 0 && atomicAdd(&hdr->done, 1u) == nvq * spread - 1) {
        hdr->ent_cursor = 0;
        hdr->byte_cursor = 0;
        hdr->done = 0;
    }
}

// ---- resident variant (low-latency mode of a reader) ----
// The launch and the completion of a kernel are ~10 of the ~22 us a single query costs on the path above; a word in
// pinned memory makes the round trip host -> running kernel -> host in 1.6 us on the same machine.  This kernel stays:
// workgroup c (one per chunk) waits for a query in the mailbox (fine-grained pinned host memory, common.h), answers
// it exactly as search_block_kernel does, and the last workgroup to finish writes the query's sequence number back --
// the host spins on that word instead of launching and synchronising.  It is a LEASE, not a daemon: the first
// workgroup watches the device's wall clock and makes everyone leave after idle_ticks without a query or life_ticks
// in any case (a crashed host, or a device-wide synchronisation somewhere else in the process, waits for no longer
// than that); `leave` in the header tells the other workgroups.  Leaving races with a query being posted: the first
// workgroup announces `closing`, looks at the mailbox once more and only then writes `exited`; a host that sees
// `exited` without its sequence number takes the query elsewhere (resident_query).
// Memory: every access to the mailbox is a system-scope atomic (uncached, no fence: a system-scope fence writes back
// and invalidates L2, ~4 us each here); results go to fine-grained host memory with ordinary stores, which are
// write-through there, and the sequence number follows once every wave has seen its stores acknowledged
// (s_waitcnt vmcnt(0)).  Entries and result bytes are ordinary stores (to pinned memory and to the device arena, where
// a copy engine fetches what does not fit the pinned prefix): only a workgroup that wrote some pays for the write-back
// of its L2.
__device__ __forceinline__ u32 sys_load(const volatile u32 *p)
{
    return __hip_atomic_load(const_cast<const u32 *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ u64 sys_load64(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void sys_store(volatile u32 *p, u32 v)
{
    __hip_atomic_store(const_cast<u32 *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (The header lives in ordinary device memory, which the eight L2s only agree on between kernels: inside this one the
// cursors are moved by device-scope atomics, so they are also RESET by device-scope stores -- a plain store would sit
// in one XCD's L2 while the other workgroups' atomics keep counting from the old value.)
__device__ __forceinline__ void dev_store(u32 *p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(SM_BLOCK) void search_resident_kernel(const ChunkDesc *chunks, u32 nc, SmallHeader *hdr,
                                                                     u32 *h_overflow, SmallRecord *rec, SmallEntry *ent,
                                                                     u8 *bytes, u8 *hbytes, u32 spread, ResidentMailbox *mb,
                                                                     u32 seen, u64 idle_ticks, u64 life_ticks)
{
    __shared__ __attribute__((aligned(8))) u8 s_pat[SM_MAX_PLEN + 32];
    __shared__ u32 s_seq, s_plen, s_leave;
    const u32 c = blockIdx.x / spread, sub = blockIdx.x % spread;
    const u32 tid = threadIdx.x;
    const bool first = blockIdx.x == 0;
    const ChunkDesc ch = chunks[c];          // (every change of the reader's chunks stops this kernel first)
    const u64 *line = reinterpret_cast<const u64 *>(&mb->post);
    const u64 t_start = wall_clock64();
    u64 t_last = t_start;
    for (;;) {
        if (tid < kWave) {
            // wave 0 polls the posted line: lanes 0 .. 7 read its eight words with one load
            u32 leave = 0, seq = seen, plen = 0;
            u64 w = 0;
            // (the answer waits for the LAST workgroup to notice the query: nothing but the load in the loop -- the clock
            // and the leave flag are looked at every eighth time round)
            for (u32 spin = 0;; ++spin) {
                if (tid < 8) w = sys_load64(line + tid);
                const u32 seq_a = (u32)__shfl(w, 0), seq_b = (u32)(__shfl(w, 7) >> 32);
                plen = (u32)(__shfl(w, 0) >> 32);
                if (seq_a != seen && seq_a == seq_b) {
                    seq = seq_a;
                    break;
                }
                // (the lease is looked at every eighth poll -- and at once when the query just answered ran past it: a host
                // posting back to back would otherwise never let the idle polls, hence the clock, come round)
                if ((spin & 7u) != 7u && !(first && spin == 0 && t_last - t_start > life_ticks)) continue;
                if (first) {
                    const u64 now = wall_clock64();
                    if (now - t_last > idle_ticks || now - t_start > life_ticks) {
                        if (tid == 0) sys_store(&mb->closing, 1u);
                        stores_done();
                        if (tid < 8) w = sys_load64(line + tid);
                        const u32 a2 = (u32)__shfl(w, 0), b2 = (u32)(__shfl(w, 7) >> 32);
                        plen = (u32)(__shfl(w, 0) >> 32);
                        if (a2 != seen && a2 == b2) {       // a query came in while the lease ran out: answer it first
                            if (tid == 0) sys_store(&mb->closing, 0u);
                            seq = a2;
                            break;
                        }
                        leave = 1;
                        break;
                    }
                } else if (__hip_atomic_load(&hdr->leave, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) {
                    leave = 1;
                    break;
                }
            }
            if (!leave && plen == kResidentStop) leave = 1;            // (told to)
            // (an add, not a store: told to leave by the host, the others may have counted themselves out already)
            if (leave && first && tid == 0) __hip_atomic_fetch_add(&hdr->leave, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (!leave) {
                plen = min(plen, SM_MAX_PLEN);
                if (plen <= sizeof mb->post.bytes) {
                    // bytes 8 .. 59 of the line: lane k holds bytes 8 k .. 8 k + 7
                    if (tid >= 1 && tid < 8) {
                        u64 v = w;
                        if (tid == 7) v &= 0xffffffffull;
                        *reinterpret_cast<u64 *>(s_pat + 8 * (tid - 1)) = v;
                    }
                } else {
                    const u64 *src = reinterpret_cast<const u64 *>(mb->query);
                    for (u32 i = tid; i < (SM_MAX_PLEN + 32) / 8; i += kWave) *reinterpret_cast<u64 *>(s_pat + 8 * i) = sys_load64(src + i);
                }
            }
            if (tid == 0) {
                s_seq = seq;
                s_plen = plen;
                s_leave = leave;
            }
        }
        __syncthreads();
        if (s_leave) break;
        const u32 seq = s_seq, plen = s_plen;
        seen = seq;
        if (tid >= plen && tid < SM_MAX_PLEN + 32) s_pat[tid] = 0;     // zero padding behind the query
        __syncthreads();
        u32 ck = 0;
        if (tid == 0) {             // what did THIS workgroup read off the posted line?  (resident_query_checksum, common.h)
            ck = 0x811C9DC5u ^ plen;
            for (u32 i = 0; i < plen; i += 8) {
                const u64 wv = *reinterpret_cast<const u64 *>(s_pat + i);
                ck = (ck ^ (u32)wv) * 0x01000193u;
                ck = (ck ^ (u32)(wv >> 32)) * 0x01000193u;
            }
        }
#ifdef PSS_TRACE_RESIDENT
        __shared__ u32 s_tr[8];
        u32 *tr = blockIdx.x == 0 ? s_tr : nullptr;
#else
        u32 *tr = nullptr;
#endif
        PSS_TR(tr, 0);
        const bool wrote = block_pair(ch, s_pat, plen, sub, blockIdx.x, spread, hdr, h_overflow, rec, ent, bytes, hbytes,
                                      nc * spread == 1, tr);
        PSS_TR(tr, 5);
        stores_done();
        __syncthreads();
        PSS_TR(tr, 6);
        // every wave's stores have reached L2; one wave writes the workgroup's XCD L2 back (the entry table and the
        // result bytes are ordinary stores and stay there otherwise) -- a write-back per wave costs 4 us, one per
        // workgroup ~1, a workgroup without entries none (its record went out as a system-scope store)
        if (wrote && tid < kWave) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
#ifdef PSS_TRACE_RESIDENT
        if (tr && tid == 0) {
            tr[7] = (u32)wall_clock64();
            for (int i = 0; i < 8; ++i) sys_store(&mb->pad2[i], tr[i]);
            stores_done();
        }
#endif
        if (tid == 0 && nc * spread != 1) __hip_atomic_fetch_add(&hdr->echo, ck, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0 && (nc * spread == 1 || atomicAdd(&hdr->done, 1u) == nc * spread - 1)) {
            if (nc * spread != 1) {          // (a workgroup on its own never moves the cursors)
                ck = __hip_atomic_load(&hdr->echo, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                dev_store(&hdr->ent_cursor, 0u);
                dev_store(&hdr->byte_cursor, 0u);
                dev_store(&hdr->done, 0u);
                dev_store(&hdr->echo, 0u);
                stores_done();
            }
            __hip_atomic_store(const_cast<u64 *>(&mb->done_seq_echo), (u64)seq | ((u64)ck << 32), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
        t_last = wall_clock64();
    }
    // Every workgroup is leaving (`leave` counts them: 1 from the first, + 1 per other); the first one says so to the
    // host once nobody can touch the arena any more, and leaves the header zero.  (A workgroup that saw a query the
    // first one did not may have answered its part alone on the way out: the cursors it moved are reset here, the
    // host sees `exited` without that query's sequence number and takes it elsewhere.)
    if (tid == 0) {
        if (!first) {
            __hip_atomic_fetch_add(&hdr->leave, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(&hdr->leave, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != nc * spread)
                __builtin_amdgcn_s_sleep(8);
            dev_store(&hdr->ent_cursor, 0u);
            dev_store(&hdr->byte_cursor, 0u);
            dev_store(&hdr->done, 0u);
            dev_store(&hdr->leave, 0u);
            dev_store(&hdr->echo, 0u);
            stores_done();
            sys_store(&mb->exited, 1u);
        }
    }
}

// ---- opt-in: the reference's order inside a (query, chunk) pair (pss_reader_set_result_order) -------------------------
// src/lib.rs:262-276 walks the hits of a chunk in suffix-array order and pushes an entry when its line start is first
// seen: the entries come out ordered by the suffix-array rank of their FIRST hit.  hit_lines_kernel keeps the LEFTMOST
// match of an entry instead (that is what it can decide from the text alone), so an entry that holds the pattern
// twice may come out at another place of the list -- the same multiset, another order.  Here every hit reports its
// entry, a stable sort of the hits by (pair, line start) brings the hits of one entry together in suffix-array
// order, and all but the first of each run are dropped; the scans and emit_kernel then run as always.
__global__ __launch_bounds__(256) void hit_bounds_kernel(const ChunkDesc *chunks, u32 nc, u64 nvq, const u32 *lo, const u64 *hit_off,
                                                           u64 H, u32 *start_out, u32 *len_out, u64 *key_out, u32 *val_out)
{
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < H; t += (u64)gridDim.x * blockDim.x) {
        u64 a = 0, b = nvq;
        while (b - a > 1) {
            const u64 mid = a + (b - a) / 2;
            if (hit_off[mid] <= t) a = mid; else b = mid;
        }
        const ChunkDesc ch = chunks[(u32)(a % nc)];
        const u32 di = ch.sa[lo[a] + (u32)(t - hit_off[a])];
        u32 ls = 0, ll = 0;
        entry_bounds(ch, di, ls, ll);
        start_out[t] = ls;
        len_out[t] = ll;
        key_out[t] = (a << 31) | (u64)ls;        // line starts are below 2^31, pairs below 2^33
        val_out[t] = (u32)t;
    }
}

__global__ __launch_bounds__(256) void mark_later_hits_kernel(const u64 *key, const u32 *val, u64 H, u32 *len)
{
    for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < H; j += (u64)gridDim.x * blockDim.x)
        if (j > 0 && key[j] == key[j - 1]) len[val[j]] = kSkip;
}

__global__ __launch_bounds__(256) void emit_kernel(const ChunkDesc *chunks, u32 nc, u64 nvq, const u64 *hit_off, u64 H,
                                                     const MidState *mid, const u32 *start, const u32 *len,
                                                     const u64 *eidx, const u64 *boff, u64 *ent_off, u8 *out)
{
    if (mid) H = mid->flag ? 0 : mid->hits;
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
                                                             u64 *qcount, const MidState *mid)
{
    const u32 q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq || (mid && mid->flag)) return;      // mid pipeline over its caps: eidx is not filled
    const u64 v0 = (u64)q * nc, v1 = v0 + nc;
    qcount[q] = eidx[hit_off[v1]] - eidx[hit_off[v0]];
}

// --------------------------------------------------------------------- host --

// device arena of the small paths: header of the launch path, result bytes (shared: the paths never run a query at the
// same time), header of the resident kernel
constexpr size_t SM_ARENA_RHDR = 64 + (size_t)SM_BYTE_CAP + 64;
constexpr size_t SM_ARENA_BYTES = SM_ARENA_RHDR + 64;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "search.hip is written for gfx950 (MI355X): the fused search kernels keep ~72 KiB of LDS per workgroup (160 KiB per CU there; gfx90a / gfx942 stop at 64 KiB)"
#endif

enum SSlot { Q_ORD_K0 = 50, Q_ORD_K1, Q_ORD_V0, Q_ORD_V1, Q_ORD_WORK, Q_BYTES = 10, Q_OFF, Q_LO, Q_CNT, Q_HITOFF, Q_START, Q_LEN, Q_EIDX, Q_BOFF, Q_ENTOFF, Q_OUT, Q_SMALL, Q_QCOUNT, Q_ARENA = 28, Q_HEAT = 46 };

void HostResult::release()
{
    free(qcount);
    if (pinned) {
        pinned_pool_free(pinned, pinned_bytes);
    } else {
        free(offsets);
        free(bytes);
    }
    *this = HostResult{};
}

// Room for E + 1 offsets and B bytes: one pinned block when the result is large, else malloc.
int alloc_host_result(HostResult *res, u64 E, u64 B, bool allow_pinned)
{
    const size_t off_bytes = round_up((size_t)(E + 1) * sizeof(u64), 64);
    if (allow_pinned && off_bytes + B >= ((size_t)8 << 20)) {
        size_t granted = 0;
        void *blk = pinned_pool_alloc(off_bytes + (size_t)B + 64, &granted);
        if (blk) {
            res->pinned = blk;
            res->pinned_bytes = granted;
            res->offsets = static_cast<u64 *>(blk);
            res->bytes = static_cast<u8 *>(blk) + off_bytes;
            return PSS_OK;
        }
    }
    res->offsets = (u64 *)malloc((size_t)(E + 1) * sizeof(u64));
    res->bytes = (u8 *)malloc(B ? (size_t)B : 1);
    return (res->offsets && res->bytes) ? PSS_OK : PSS_ENOMEM;
}

// The packed result of one small-path launch from what the kernel left in the pinned arena (records, entry table,
// the first SM_BYTE_PREFIX result bytes) and, beyond that prefix, in the device arena.
static int small_collect(DeviceCtx *ctx, const u8 *h_arena, const u8 *d_bytes, u64 nrec, u32 spread, u32 nc, HostResult *res,
                         pss_search_stats *st, uint64_t *chunk_hits = nullptr)
{
    const SearchKnobs &knobs = search_knobs();
    hipStream_t s = ctx->stream;
    const SmallRecord *h_rec = reinterpret_cast<const SmallRecord *>(h_arena + SM_OFF_REC);
    const SmallEntry *h_ent = reinterpret_cast<const SmallEntry *>(h_arena + SM_OFF_ENT);
    u64 E = 0, B = 0;
    for (u64 ri = 0; ri < nrec; ++ri) {
        const SmallRecord r = h_rec[ri];
        E += r.ent_count;
        for (u32 k = 0; k < r.ent_count; ++k) B += h_ent[r.ent_start + k].len;
    }
    std::vector<u8> h_more;
    const u8 *h_bytes = h_arena + SM_OFF_BYTES;
    if (B > SM_BYTE_PREFIX) {
        // the rest comes from the device arena with one copy: through the pinned staging when it fits
        u8 *dst = nullptr;
        if (B <= DeviceCtx::kStageR && !knobs.no_search_stage && ctx->ensure_search_stage() == PSS_OK) {
            dst = static_cast<u8 *>(ctx->search_stage) + DeviceCtx::kStageQ;
        } else {
            h_more.resize(B);
            dst = h_more.data();
        }
        PSS_HIP(hipMemcpyAsync(dst, d_bytes, B, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
        h_bytes = dst;
    }
    PSS_TRY(alloc_host_result(res, E, B, false));
    u64 e_out = 0, b_out = 0;
    for (u64 ri = 0; ri < nrec; ++ri) {       // pairs in (query, chunk) order, sub-blocks in interval order
        const SmallRecord r = h_rec[ri];
        res->qcount[(ri / spread) / nc] += r.ent_count;
        if (chunk_hits) chunk_hits[(ri / spread) % nc] += r.ent_count;
        if (r.ent_count == 0) continue;
        // the entries of one record are packed back to back in the arena: one copy for all of them
        const u64 first = b_out;
        for (u32 k = 0; k < r.ent_count; ++k) {
            res->offsets[e_out++] = b_out;
            b_out += h_ent[r.ent_start + k].len;
        }
        memcpy(res->bytes + first, h_bytes + h_ent[r.ent_start].byte_off, b_out - first);
    }
    res->offsets[e_out] = b_out;
    res->n_entries = e_out;
    res->n_bytes = b_out;
    st->entries = e_out;
    st->result_bytes = b_out;
    st->hits = e_out;   // hits before dedupe are not counted on this path
    return PSS_OK;
}

// One query through the resident kernel (low-latency mode): start it if none is running for this chunk table, post the
// query in the mailbox, spin on the answer.  *served = false: the kernel left without answering (its lease ran out as
// the query came in) or the result does not fit the path -- the caller goes on as if this mode did not exist.
static int resident_query(DeviceCtx *ctx, const ChunkDesc *d_chunks, u32 nc, const u8 *q, u32 plen, HostResult *res,
                          pss_search_stats *st, bool *served)
{
    *served = false;
    const SearchKnobs &knobs = search_knobs();
    PSS_TRY(ctx->ensure_resident());
    PSS_TRY(ctx->slot[Q_ARENA].reserve(SM_ARENA_BYTES));
    u8 *arena = ctx->slot[Q_ARENA].as<u8>();
    DeviceCtx::Resident &R = ctx->resident;
    u8 *h_arena = static_cast<u8 *>(R.arena);
    u8 *v_arena = static_cast<u8 *>(R.arena_dev);
    ResidentMailbox *mb = reinterpret_cast<ResidentMailbox *>(h_arena + SM_OFF_MAILBOX);
    volatile u32 *h_overflow = reinterpret_cast<volatile u32 *>(h_arena + SM_OFF_FLAGS);
    // One workgroup per chunk (the launch path gives a pair up to 32): every one of them spins while the kernel stays,
    // the answer waits for the LAST of them to notice the query, and a reader of one chunk then needs no counter of
    // finished workgroups at all.  A chunk with more than 1024 hits goes to the launch path.
    const u32 spread = 1;
    static u64 ticks_per_us = 0;
    if (ticks_per_us == 0) {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || khz <= 0) khz = 100000;
        ticks_per_us = std::max<u64>(1, (u64)khz / 1000);
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (R.running && (R.chunks != d_chunks || R.nc != nc || R.spread != spread || R.d_arena != arena)) ctx->stop_resident();
        if (R.running && __atomic_load_n(&mb->exited, __ATOMIC_ACQUIRE)) {
            PSS_HIP(hipStreamSynchronize(R.stream));
            R.running = false;
        }
        if (!R.running) {
            mb->exited = 0;
            mb->closing = 0;
            __atomic_thread_fence(__ATOMIC_SEQ_CST);
            SmallHeader *d_hdr = reinterpret_cast<SmallHeader *>(arena + SM_ARENA_RHDR);
            PSS_HIP(hipMemsetAsync(d_hdr, 0, 64, R.stream));
            hipLaunchKernelGGL(search_resident_kernel, dim3(nc * spread), dim3(SM_BLOCK), 0, R.stream, d_chunks, nc, d_hdr,
                               reinterpret_cast<u32 *>(v_arena + SM_OFF_FLAGS),
                               reinterpret_cast<SmallRecord *>(v_arena + SM_OFF_REC),
                               reinterpret_cast<SmallEntry *>(v_arena + SM_OFF_ENT), arena + 64, v_arena + SM_OFF_BYTES, spread,
                               reinterpret_cast<ResidentMailbox *>(v_arena + SM_OFF_MAILBOX), R.seq,
                               (u64)knobs.resident_idle_us * ticks_per_us, (u64)knobs.resident_life_us * ticks_per_us);
            PSS_HIP(hipGetLastError());
            R.running = true;
            R.chunks = d_chunks;
            R.nc = nc;
            R.spread = spread;
            R.d_arena = arena;
            ++R.launches;
        }
        *h_overflow = 0;
        const auto t_post = std::chrono::steady_clock::now();
        R.post(q, plen);
        const u32 seq = R.seq;
        bool answered = false;
        u64 done_pair = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (u32 spins = 0;; ++spins) {
            done_pair = __atomic_load_n(&mb->done_seq_echo, __ATOMIC_ACQUIRE);
            if ((u32)done_pair == seq) {
                answered = true;
                break;
            }
            if (__atomic_load_n(&mb->exited, __ATOMIC_ACQUIRE)) {
                done_pair = __atomic_load_n(&mb->done_seq_echo, __ATOMIC_ACQUIRE);   // (written before `exited`)
                answered = (u32)done_pair == seq;
                break;
            }
            __builtin_ia32_pause();
            if ((spins & 0xfffu) == 0xfffu &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) {
                ctx->stop_resident();
                set_error("the resident search kernel did not answer within 5 s");
                return PSS_EDEVICE;
            }
        }
        if (answered) {
            // Did every workgroup work on the bytes that were posted?  (The two sequence numbers at the ends of the line
            // guard a read that the fabric splits in halves; a finer split could leave a stale word in the middle.)
            // The workgroups' checksums are ADDED up (round 5; they used to be xor-ed, and an even number of workgroups that
            // all read the same stale line -- the likely failure, they poll one cache line -- cancelled to the expected 0).
            const u32 one = resident_query_checksum(q, plen);
            const u32 want = (u32)((u64)(nc * spread) * one);
            if ((u32)(done_pair >> 32) != want) {
                ++R.torn;
                return PSS_OK;                   // not served: the launch path answers this query
            }
            // (no HIP events on this path: the time from the post to the answer as the host saw it)
            st->ms_device = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_post).count();
            st->ms_interval = st->ms_device;
#ifdef PSS_TRACE_RESIDENT
            if (knob("PSS_TRACE_RESIDENT")) {
                u32 t[8];
                for (int i = 0; i < 8; ++i) t[i] = mb->pad2[i];
                fprintf(stderr, "[pss] resident trace (10 ns ticks): host %.1f us | interval %u, bounds %u, sums %u, pack %u, copy %u, drain %u, fence %u\n",
                        st->ms_device * 1e3, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[7] - t[6]);
            }
#endif
            ++R.served;
            if (*h_overflow) return PSS_OK;      // too many hits for this path: the launch path and its fallbacks take it
            PSS_TRY(small_collect(ctx, h_arena, arena + 64, (u64)nc * spread, spread, nc, res, st));
            *served = true;
            return PSS_OK;
        }
        PSS_HIP(hipStreamSynchronize(R.stream));     // it left without this query: once more with a fresh one
        R.running = false;
    }
    return PSS_OK;
}

// Suffix-array hits of the batch per chunk (pair p = query p / nc on chunk p % nc): one workgroup per chunk.
__global__ __launch_bounds__(256) void chunk_hits_kernel(const u32 *cnt, u64 nq, u32 nc, u64 *out)
{
    __shared__ u64 s_w[256 / kWave];
    const u32 c = blockIdx.x;
    u64 acc = 0;
    for (u64 q = threadIdx.x; q < nq; q += 256) acc += cnt[q * nc + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane_id() == 0) s_w[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

int search_batch_device(DeviceCtx *ctx, const ChunkDesc *d_chunks, u32 nc, const uint8_t *qbytes,
                        const uint64_t *qoffsets, uint32_t nq, HostResult *res, pss_search_stats *st, SearchMode mode, bool low_latency,
                        uint64_t *chunk_hits, bool sa_order)
{
    if (chunk_hits)
        for (u32 c = 0; c < nc; ++c) chunk_hits[c] = 0;
    const bool counts_only = mode == SEARCH_COUNTS;
    const bool device_only = mode == SEARCH_DEVICE;
    const SearchKnobs &knobs = search_knobs();
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
        if (device_only) {        // a rank that owns no chunk still answers: nq zero counters, no entries
            if (nq) {
                PSS_TRY(ctx->slot[Q_QCOUNT].reserve((size_t)nq * 8));
                PSS_HIP(hipMemsetAsync(ctx->slot[Q_QCOUNT].p, 0, (size_t)nq * 8, s));
                PSS_HIP(hipStreamSynchronize(s));
                res->d_qcount = ctx->slot[Q_QCOUNT].as<u64>();
            }
            return PSS_OK;
        }
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
    PSS_TRY(ctx->slot[Q_SMALL].reserve(SC_MAX_BLOCKS * 8 + 256));      // scan partials, totals, MidState
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

    const hipEvent_t e0 = ctx->search_ev[0], e1 = ctx->search_ev[1], e2 = ctx->search_ev[2];
    const auto t_begin = std::chrono::steady_clock::now();
    auto host_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    const size_t off_bytes = ((size_t)nq + 1) * 8;
    const bool tiny = qtotal + 32 <= 8192 && off_bytes <= 8192;
    u8 *stg = static_cast<u8 *>(ctx->pinned) + SM_OFF_QUERY;          // 16 KiB of the pinned scratch
    if (tiny) {
        memcpy(stg, qbytes, qtotal);
        memset(stg + qtotal, 0, 32);
        memcpy(stg + 8192, qoffsets, off_bytes);
    }
    if (counts_only) sa_order = false;          // (counts do not depend on the order)
    bool small = tiny && nvq <= SM_MAX_VQ && !counts_only && !device_only && !knobs.no_small_path && !sa_order;
    for (u32 i = 0; small && i < nq; ++i) small = qoffsets[i + 1] - qoffsets[i] <= SM_MAX_PLEN;
    const u64 waves_per_block = 256 / kWave;
    if (small) {
        // ---- fused small-batch path: one kernel, queries and results through pinned host memory ----
        if (low_latency && nq == 1 && nvq <= SM_BLOCK_MAX_VQ && !knobs.no_block_path) {
            // a reader in low-latency mode: the resident kernel answers without a launch (or declines)
            bool served = false;
            PSS_TRY(resident_query(ctx, d_chunks, nc, qbytes + qoffsets[0], (u32)(qoffsets[1] - qoffsets[0]), res, st, &served));
            if (served) {
                st->ms_host = host_ms();
                return PSS_OK;
            }
            for (u32 i = 0; i < nq; ++i) res->qcount[i] = 0;
        }
        PSS_TRY(ctx->slot[Q_ARENA].reserve(SM_ARENA_BYTES));
        u8 *arena = ctx->slot[Q_ARENA].as<u8>();
        SmallHeader *d_hdr = reinterpret_cast<SmallHeader *>(arena);
        u8 *d_bytes = arena + 64;
        if (ctx->small_hdr_ready != arena) {
            PSS_HIP(hipMemsetAsync(d_hdr, 0, 64, s));
            ctx->small_hdr_ready = arena;
        }
        u8 *h_arena = static_cast<u8 *>(ctx->pinned);
        u8 *v_arena = static_cast<u8 *>(ctx->pinned_dev);               // the same bytes, device view
        volatile u32 *h_overflow = reinterpret_cast<volatile u32 *>(h_arena + SM_OFF_FLAGS);
        *h_overflow = 0;
        // (the events bracket the kernel for last_stats().ms_device; two records cost the single-query path a couple of
        // microseconds of host time and two marker packets, so it only takes them when asked: PSS_SEARCH_EVENTS=1)
        const bool timed = knobs.small_path_events;
        if (timed) PSS_HIP(hipEventRecord(e0, s));
        const u8 *v_q = v_arena + SM_OFF_QUERY;
        const u64 *v_qoff = reinterpret_cast<const u64 *>(v_arena + SM_OFF_QUERY + 8192);
        u32 *v_flags = reinterpret_cast<u32 *>(v_arena + SM_OFF_FLAGS);
        SmallRecord *v_rec = reinterpret_cast<SmallRecord *>(v_arena + SM_OFF_REC);
        SmallEntry *v_ent = reinterpret_cast<SmallEntry *>(v_arena + SM_OFF_ENT);
        const bool block_path = nvq <= SM_BLOCK_MAX_VQ && !knobs.no_block_path;
        // workgroups per pair: as many as keep the launch at <= 64 workgroups (every one of them searches the
        // interval first; on 15 chunks 32 per pair cost a miss 7 us, 4 per pair nothing measurable)
        // (33 .. 64 pairs -- one query over that many chunks -- get up to 4 per pair, 256 workgroups at most: a pair of
        // 1025 .. 4096 hits stays on this path instead of throwing the launch away for the general pipeline)
        u32 spread = 1;
        if (block_path)
            while (spread < SM_SPREAD && nvq * spread * 2 <= (nvq <= 32 ? 64u : 256u)) spread *= 2;
        if (block_path)
            hipLaunchKernelGGL(search_block_kernel, dim3((u32)nvq * spread), dim3(SM_BLOCK), 0, s, d_chunks, nc, v_q, v_qoff,
                               (u32)nvq, d_hdr, v_flags, v_rec, v_ent, d_bytes, v_arena + SM_OFF_BYTES, spread);
        else
            hipLaunchKernelGGL(search_small_kernel, dim3((u32)((nvq + waves_per_block - 1) / waves_per_block)), dim3(256),
                               0, s, d_chunks, nc, v_q, v_qoff, (u32)nvq, d_hdr, v_flags, v_rec, v_ent, d_bytes,
                               v_arena + SM_OFF_BYTES);
        if (timed) PSS_HIP(hipEventRecord(e2, s));
        PSS_HIP(hipStreamSynchronize(s));
        if (!*h_overflow) {
            PSS_TRY(small_collect(ctx, h_arena, d_bytes, nvq * spread, spread, nc, res, st, chunk_hits));
            float ms = 0.f;
            if (timed) PSS_HIP(hipEventElapsedTime(&ms, e0, e2));
            st->ms_device = ms;
            st->ms_interval = ms;
            st->ms_host = host_ms();
            return PSS_OK;
        }
        // overflow: fall through to the general path (qcount is still all zero)
    }
    if (tiny) {
        // pageable H2D copies are synchronous and slow to start: tiny batches go up from the pinned staging
        PSS_HIP(hipMemcpyAsync(d_q, stg, qtotal + 32, hipMemcpyHostToDevice, s));
        PSS_HIP(hipMemcpyAsync(d_qoff, stg + 8192, off_bytes, hipMemcpyHostToDevice, s));
    } else if (qtotal + 32 + off_bytes + 64 <= DeviceCtx::kStageQ && !knobs.no_search_stage) {
        // mid-size batch: the same through the larger pinned staging
        PSS_TRY(ctx->ensure_search_stage());
        u8 *sq = static_cast<u8 *>(ctx->search_stage);
        u8 *so = sq + round_up(qtotal + 32, 64);
        memcpy(sq, qbytes, qtotal);
        memset(sq + qtotal, 0, 32);
        memcpy(so, qoffsets, off_bytes);
        PSS_HIP(hipMemcpyAsync(d_q, sq, qtotal + 32, hipMemcpyHostToDevice, s));
        PSS_HIP(hipMemcpyAsync(d_qoff, so, off_bytes, hipMemcpyHostToDevice, s));
    } else {
        PSS_HIP(hipMemsetAsync(d_q + qtotal, 0, 32, s));
        if (qtotal) PSS_HIP(hipMemcpyAsync(d_q, qbytes, qtotal, hipMemcpyHostToDevice, s));
        PSS_HIP(hipMemcpyAsync(d_qoff, qoffsets, off_bytes, hipMemcpyHostToDevice, s));
    }
    PSS_HIP(hipEventRecord(e0, s));
    const u64 lane_min = knobs.lane_search_min;   // pairs from which one lane per pair is at least as fast as 16
                                                  // (10 000 pairs: 0.039 ms either way; 30 000: 0.044 vs 0.078 ms)
    if (nvq >= lane_min && !knobs.wave_search)
        hipLaunchKernelGGL(search_interval_lane_kernel, dim3((u32)((nvq + 255) / 256)), dim3(256), 0, s, d_chunks, nc,
                           d_q, d_qoff, nvq, d_lo, d_cnt);
    else if (nvq >= 2048 && !knobs.wave_search && !knobs.no_group_search)
        hipLaunchKernelGGL(search_interval_group_kernel, dim3((u32)((nvq + 15) / 16)), dim3(256), 0, s, d_chunks, nc, d_q,
                           d_qoff, nvq, d_lo, d_cnt);
    else
        hipLaunchKernelGGL(search_interval_kernel, dim3((u32)((nvq + waves_per_block - 1) / waves_per_block)), dim3(256),
                           0, s, d_chunks, nc, d_q, d_qoff, nvq, d_lo, d_cnt);
    PSS_HIP(hipEventRecord(e1, s));
    if (chunk_hits && nc <= 4096) {
        // (a reader with suffix arrays on the host tier: where did this batch's hits land?  One small kernel over the
        // pair counts and a wait -- next to probes over PCIe, nothing)
        PSS_TRY(ctx->slot[Q_HEAT].reserve((size_t)nc * 8));
        u64 *d_heat = ctx->slot[Q_HEAT].as<u64>();
        hipLaunchKernelGGL(chunk_hits_kernel, dim3(nc), dim3(256), 0, s, d_cnt, (u64)nq, nc, d_heat);
        PSS_HIP(hipMemcpyAsync(chunk_hits, d_heat, (size_t)nc * 8, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
    }
    if (nvq <= MID_MAX && !counts_only && !knobs.no_mid_pipeline && !sa_order) {
        // ---- mid pipeline: totals stay on the device, one wait for them, one for the result ----
        const u64 byte_cap = (u64)16 << 20;
        PSS_TRY(ctx->slot[Q_START].reserve((size_t)MID_MAX * 4));
        PSS_TRY(ctx->slot[Q_LEN].reserve((size_t)MID_MAX * 4));
        PSS_TRY(ctx->slot[Q_EIDX].reserve(((size_t)MID_MAX + 1) * 8));
        PSS_TRY(ctx->slot[Q_BOFF].reserve(((size_t)MID_MAX + 1) * 8));
        PSS_TRY(ctx->slot[Q_ENTOFF].reserve(((size_t)MID_MAX + 1) * 8));
        PSS_TRY(ctx->slot[Q_OUT].reserve(byte_cap + 16));
        u32 *d_start = ctx->slot[Q_START].as<u32>();
        u32 *d_len = ctx->slot[Q_LEN].as<u32>();
        u64 *d_eidx = ctx->slot[Q_EIDX].as<u64>();
        u64 *d_boff = ctx->slot[Q_BOFF].as<u64>();
        u64 *d_entoff = ctx->slot[Q_ENTOFF].as<u64>();
        u8 *d_out = ctx->slot[Q_OUT].as<u8>();
        MidState *d_ms = reinterpret_cast<MidState *>(d_total + 8);         // behind the scan partials
        const u32 grid = (u32)ctx->num_cus * 4;
        hipLaunchKernelGGL(mid_hitoff_kernel, dim3(1), dim3(MID_BLOCK), 0, s, d_cnt, (u32)nvq, d_hitoff, d_ms);
        hipLaunchKernelGGL(hit_lines_kernel, dim3(grid), dim3(256), 0, s, d_chunks, nc, d_q, d_qoff, nvq, d_lo, d_hitoff,
                           (u64)0, d_ms, d_start, d_len);
        hipLaunchKernelGGL(mid_hit_scans_kernel, dim3(1), dim3(MID_BLOCK), 0, s, d_len, d_ms, byte_cap, d_eidx, d_boff);
        hipLaunchKernelGGL(emit_kernel, dim3(grid), dim3(256), 0, s, d_chunks, nc, nvq, d_hitoff, (u64)0, d_ms, d_start,
                           d_len, d_eidx, d_boff, d_entoff, d_out);
        hipLaunchKernelGGL(query_counts_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, nc, nq, d_hitoff, d_eidx, d_qcount,
                           (const MidState *)d_ms);
        PSS_HIP(hipEventRecord(e2, s));
        MidState *h_ms = reinterpret_cast<MidState *>(h_small);
        PSS_HIP(hipMemcpyAsync(h_ms, d_ms, sizeof(MidState), hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
        if (!h_ms->flag) {
            const u64 H = h_ms->hits, E = h_ms->entries, B = h_ms->bytes;
            if (device_only) {
                res->d_qcount = d_qcount;
                res->d_offsets = d_entoff;
                res->d_bytes = d_out;
                res->n_entries = E;
                res->n_bytes = B;
                st->hits = H;
                st->entries = E;
                st->result_bytes = B;
                float msd = 0.f;
                PSS_HIP(hipEventElapsedTime(&msd, e0, e2));
                st->ms_device = msd;
                PSS_HIP(hipEventElapsedTime(&msd, e0, e1));
                st->ms_interval = msd;
                st->ms_host = host_ms();
                return PSS_OK;
            }
            PSS_TRY(alloc_host_result(res, E, B, false));
            const size_t need = round_up(E * 8, 64) + round_up(B, 64) + (size_t)nq * 8;
            if (need <= DeviceCtx::kStageR && !knobs.no_search_stage) {
                // down through pinned staging (three DMA copies, one wait), then plain memcpy
                PSS_TRY(ctx->ensure_search_stage());
                u8 *r0 = static_cast<u8 *>(ctx->search_stage) + DeviceCtx::kStageQ;
                u8 *r1 = r0 + round_up(E * 8, 64), *r2 = r1 + round_up(B, 64);
                if (E) PSS_HIP(hipMemcpyAsync(r0, d_entoff, E * 8, hipMemcpyDeviceToHost, s));
                if (B) PSS_HIP(hipMemcpyAsync(r1, d_out, B, hipMemcpyDeviceToHost, s));
                PSS_HIP(hipMemcpyAsync(r2, d_qcount, (size_t)nq * 8, hipMemcpyDeviceToHost, s));
                PSS_HIP(hipStreamSynchronize(s));
                if (E) memcpy(res->offsets, r0, E * 8);
                if (B) memcpy(res->bytes, r1, B);
                memcpy(res->qcount, r2, (size_t)nq * 8);
            } else {
                if (E) PSS_HIP(hipMemcpyAsync(res->offsets, d_entoff, E * 8, hipMemcpyDeviceToHost, s));
                if (B) PSS_HIP(hipMemcpyAsync(res->bytes, d_out, B, hipMemcpyDeviceToHost, s));
                PSS_HIP(hipMemcpyAsync(res->qcount, d_qcount, (size_t)nq * 8, hipMemcpyDeviceToHost, s));
                PSS_HIP(hipStreamSynchronize(s));
            }
            res->offsets[E] = B;
            res->n_entries = E;
            res->n_bytes = B;
            st->hits = H;
            st->entries = E;
            st->result_bytes = B;
            float ms = 0.f;
            PSS_HIP(hipEventElapsedTime(&ms, e0, e2));
            st->ms_device = ms;
            PSS_HIP(hipEventElapsedTime(&ms, e0, e1));
            st->ms_interval = ms;
            st->ms_host = host_ms();
            return PSS_OK;
        }
        // more hits or bytes than the caps: the general pipeline below takes over (intervals are kept)
    }
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
        if (sa_order) {
            if (H >= (1ull << 32) || nvq >= (1ull << 33)) {
                set_error("search: %llu hits of %llu (query, chunk) pairs are more than the suffix-array result order takes in one batch",
                          (unsigned long long)H, (unsigned long long)nvq);
                return PSS_EINVAL;
            }
            PSS_TRY(ctx->slot[Q_ORD_K0].reserve(H * 8));
            PSS_TRY(ctx->slot[Q_ORD_K1].reserve(H * 8));
            PSS_TRY(ctx->slot[Q_ORD_V0].reserve(H * 4));
            PSS_TRY(ctx->slot[Q_ORD_V1].reserve(H * 4));
            PSS_TRY(ctx->slot[Q_ORD_WORK].reserve(radix_sort_workspace_bytes()));
            u64 *OK[2] = {ctx->slot[Q_ORD_K0].as<u64>(), ctx->slot[Q_ORD_K1].as<u64>()};
            u32 *OV[2] = {ctx->slot[Q_ORD_V0].as<u32>(), ctx->slot[Q_ORD_V1].as<u32>()};
            hipLaunchKernelGGL(hit_bounds_kernel, dim3(grid), dim3(256), 0, s, d_chunks, nc, nvq, d_lo, d_hitoff, H, d_start, d_len,
                               OK[0], OV[0]);
            int pair_bits = 1;
            while ((1ull << pair_bits) < nvq) ++pair_bits;
            int od = 0;
            SortStats oss;
            PSS_TRY(radix_sort_pairs(ctx, OK, OV, (u32)H, 31 + pair_bits, 0xffffffffu, nullptr, 0, ctx->slot[Q_ORD_WORK].p, &od, false, &oss));
            hipLaunchKernelGGL(mark_later_hits_kernel, dim3(grid), dim3(256), 0, s, OK[od], OV[od], H, d_len);
        } else
        hipLaunchKernelGGL(hit_lines_kernel, dim3(grid), dim3(256), 0, s, d_chunks, nc, d_q, d_qoff, nvq, d_lo,
                           d_hitoff, H, (const MidState *)nullptr, d_start, d_len);
        PSS_TRY(device_excl_scan(ctx, InKept{d_len}, H, d_partial, d_total, d_eidx));
        PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
        if (counts_only) {
            // entries per query without materialising one: interval search, dedupe flags, two scans
            hipLaunchKernelGGL(query_counts_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, nc, nq, d_hitoff, d_eidx,
                               d_qcount, (const MidState *)nullptr);
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
            st->ms_host = host_ms();
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
        hipLaunchKernelGGL(emit_kernel, dim3(grid), dim3(256), 0, s, d_chunks, nc, nvq, d_hitoff, H,
                           (const MidState *)nullptr, d_start, d_len, d_eidx, d_boff, d_entoff, d_out);
        hipLaunchKernelGGL(query_counts_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, nc, nq, d_hitoff, d_eidx,
                           d_qcount, (const MidState *)nullptr);
        PSS_HIP(hipEventRecord(e2, s));
        if (device_only) {
            res->d_qcount = d_qcount;
            res->d_offsets = d_entoff;
            res->d_bytes = d_out;
            PSS_HIP(hipStreamSynchronize(s));
        } else {
            PSS_TRY(alloc_host_result(res, E, B, !knobs.no_pinned_results));
            if (E) PSS_HIP(hipMemcpyAsync(res->offsets, d_entoff, E * 8, hipMemcpyDeviceToHost, s));
            if (B) PSS_HIP(hipMemcpyAsync(res->bytes, d_out, B, hipMemcpyDeviceToHost, s));
            PSS_HIP(hipMemcpyAsync(res->qcount, d_qcount, (size_t)nq * 8, hipMemcpyDeviceToHost, s));
            PSS_HIP(hipStreamSynchronize(s));
            res->offsets[E] = B;
        }
    } else {
        PSS_HIP(hipEventRecord(e2, s));
        if (device_only) {
            PSS_HIP(hipMemsetAsync(d_qcount, 0, (size_t)nq * 8, s));
            res->d_qcount = d_qcount;
        }
        PSS_HIP(hipStreamSynchronize(s));
        if (!device_only) {
            res->offsets = (u64 *)calloc(1, sizeof(u64));
            if (!res->offsets) return PSS_ENOMEM;
        }
    }
    PSS_HIP(hipGetLastError());
    res->n_entries = E;
    res->n_bytes = B;
    st->entries = E;
    st->result_bytes = B;
    float ms = 0.f;
    PSS_HIP(hipEventElapsedTime(&ms, e0, e2));
    st->ms_device = ms;
    PSS_HIP(hipEventElapsedTime(&ms, e0, e1));
    st->ms_interval = ms;
    st->ms_host = host_ms();
    return PSS_OK;
}

// ---- device-side merge of per-rank packed results (multi-GPU gather) ---------------------------------------------
// The collecting rank of dist.gather_device holds `world` packed results of the same nq queries in its HBM (its own and
// the ones RCCL delivered).  Merging them THERE -- query-major, rank-major inside a query: the order of pss_merge_packed --
// leaves one result to bring down instead of `world`.

struct MergeRank {
    const u64 *counts;     // [nq]
    const u64 *starts;     // [E_r] start of every entry in bytes
    const u8 *bytes;
    u64 E, B;
    const u64 *ebase;      // [nq + 1] entries of this rank before query q (scan of counts)
};
struct MergeArgs {
    MergeRank r[16];
    u32 world;
    u64 nq;
};
struct InCounts {
    const u64 *c;
    __device__ u64 operator()(u64 q) const { return c[q]; }
};
// bytes of segment (q, r), in (q-major, r-minor) order
struct InSegBytes {
    MergeArgs a;
    __device__ u64 operator()(u64 i) const
    {
        const u64 q = i / a.world;
        const MergeRank &m = a.r[i % a.world];
        const u64 c = m.counts[q];
        if (!c) return 0;
        const u64 e0 = m.ebase[q], e1 = e0 + c;
        return (e1 < m.E ? m.starts[e1] : m.B) - m.starts[e0];
    }
};

__global__ __launch_bounds__(256) void merge_counts_kernel(MergeArgs a, u64 *out_counts)
{
    for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < a.nq; q += (u64)gridDim.x * blockDim.x) {
        u64 t = 0;
        for (u32 r = 0; r < a.world; ++r) t += a.r[r].counts[q];
        out_counts[q] = t;
    }
}

// What arrives over RCCL is data from another process: before anything is copied by its offsets, every rank's counts must
// add up to its number of entries and its entry starts must ascend inside its bytes -- what pss_merge_packed checks on the
// host.  *bad = 1 otherwise (the copy kernel would read and write out of bounds).
__global__ __launch_bounds__(256) void merge_validate_kernel(MergeArgs a, u32 *bad)
{
    const u64 t0 = (u64)blockIdx.x * blockDim.x + threadIdx.x, step = (u64)gridDim.x * blockDim.x;
    for (u32 r = 0; r < a.world; ++r) {
        const MergeRank &m = a.r[r];
        if (t0 == 0 && m.ebase[a.nq] != m.E) *bad = 1;
        for (u64 e = t0; e < m.E; e += step) {
            const u64 s0 = m.starts[e], s1 = e + 1 < m.E ? m.starts[e + 1] : m.B;
            if (s0 > s1 || s1 > m.B) *bad = 1;
        }
    }
}

// one wavefront per (query, rank) segment: its entries' offsets, then its bytes
__global__ __launch_bounds__(256) void merge_copy_kernel(MergeArgs a, const u64 *qbase /* [nq + 1] entries before query q */,
                                                           const u64 *sbase /* [nq * world + 1] bytes before segment */, u64 *out_offsets,
                                                           u8 *out_bytes)
{
    const u64 nseg = a.nq * a.world;
    const u32 lane = lane_id();
    for (u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) / kWave; i < nseg; i += ((u64)gridDim.x * blockDim.x) / kWave) {
        const u64 q = i / a.world;
        const u32 r = (u32)(i % a.world);
        const MergeRank &m = a.r[r];
        const u64 c = m.counts[q];
        if (!c) continue;
        u64 before = 0;
        for (u32 x = 0; x < r; ++x) before += a.r[x].counts[q];
        const u64 e0 = m.ebase[q], de = qbase[q] + before;
        const u64 b0 = m.starts[e0], b1 = (e0 + c < m.E ? m.starts[e0 + c] : m.B), db = sbase[i];
        for (u64 e = lane; e < c; e += kWave) out_offsets[de + e] = db + (m.starts[e0 + e] - b0);
        for (u64 b = lane; b < b1 - b0; b += kWave) out_bytes[db + b] = m.bytes[b0 + b];
    }
}

int merge_packed_device(DeviceCtx *ctx, u32 world, u64 nq, const void *const *d_counts, const void *const *d_starts,
                        const void *const *d_bytes, const u64 *num_entries, const u64 *num_bytes, void *d_out_counts,
                        void *d_out_offsets, void *d_out_bytes)
{
    if (world == 0 || world > 16) {
        set_error("pss_merge_packed_device: 1 .. 16 ranks");
        return PSS_EINVAL;
    }
    hipStream_t s = ctx->stream;
    u64 E = 0, B = 0;
    for (u32 r = 0; r < world; ++r) {
        E += num_entries[r];
        B += num_bytes[r];
    }
    u64 *out_counts = static_cast<u64 *>(d_out_counts), *out_offsets = static_cast<u64 *>(d_out_offsets);
    if (!out_counts || !out_offsets || (B && !d_out_bytes)) {
        set_error("pss_merge_packed_device: null output buffer");
        return PSS_EINVAL;
    }
    for (u32 r = 0; r < world; ++r)
        if ((nq && !d_counts[r]) || (num_entries[r] && !d_starts[r]) || (num_bytes[r] && !d_bytes[r])) {
            set_error("pss_merge_packed_device: null buffer of rank %u", r);
            return PSS_EINVAL;
        }
    if (nq == 0) {
        PSS_HIP(hipMemsetAsync(out_offsets, 0, 8, s));
        PSS_HIP(hipStreamSynchronize(s));
        return PSS_OK;
    }
    const u64 nseg = nq * world;
    // scratch: per-rank entry bases (world x (nq + 1)), query bases (nq + 1), segment bases (nseg + 1), scan partials
    PSS_TRY(ctx->slot[Q_SMALL].reserve(SC_MAX_BLOCKS * 8 + 256));
    PSS_TRY(ctx->slot[Q_START].reserve(((size_t)world * (nq + 1) + (nq + 1) + (nseg + 1)) * 8 + 256));
    u64 *partial = ctx->slot[Q_SMALL].as<u64>(), *d_total = partial + SC_MAX_BLOCKS;
    u64 *scr = ctx->slot[Q_START].as<u64>();
    MergeArgs a;
    memset(&a, 0, sizeof a);
    a.world = world;
    a.nq = nq;
    for (u32 r = 0; r < world; ++r) {
        a.r[r].counts = static_cast<const u64 *>(d_counts[r]);
        a.r[r].starts = static_cast<const u64 *>(d_starts[r]);
        a.r[r].bytes = static_cast<const u8 *>(d_bytes[r]);
        a.r[r].E = num_entries[r];
        a.r[r].B = num_bytes[r];
        u64 *eb = scr + (size_t)r * (nq + 1);
        a.r[r].ebase = eb;
        PSS_TRY(device_excl_scan(ctx, InCounts{a.r[r].counts}, nq, partial, d_total, eb));
    }
    u64 *qbase = scr + (size_t)world * (nq + 1), *sbase = qbase + (nq + 1);
    {
        u32 *d_bad = reinterpret_cast<u32 *>(d_total + 4);
        u32 *h_bad = static_cast<u32 *>(ctx->pinned);
        PSS_HIP(hipMemsetAsync(d_bad, 0, 4, s));
        const u32 gv = (u32)std::min<u64>((E + 255) / 256 + 1, (u64)ctx->num_cus * 8);
        hipLaunchKernelGGL(merge_validate_kernel, dim3(gv), dim3(256), 0, s, a, d_bad);
        PSS_HIP(hipMemcpyAsync(h_bad, d_bad, 4, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
        if (*h_bad) {
            set_error("pss_merge_packed_device: a rank's counts do not add up to its entries, or its entry starts do not ascend "
                      "inside its bytes");
            return PSS_EINVAL;
        }
    }
    const u32 grid = (u32)std::min<u64>((nq + 255) / 256, (u64)ctx->num_cus * 8);
    hipLaunchKernelGGL(merge_counts_kernel, dim3(grid), dim3(256), 0, s, a, out_counts);
    PSS_TRY(device_excl_scan(ctx, InCounts{out_counts}, nq, partial, d_total, qbase));
    PSS_TRY(device_excl_scan(ctx, InSegBytes{a}, nseg, partial, d_total, sbase));
    const u32 grid2 = (u32)std::min<u64>((nseg + 3) / 4, (u64)ctx->num_cus * 16);
    hipLaunchKernelGGL(merge_copy_kernel, dim3(grid2 ? grid2 : 1), dim3(256), 0, s, a, (const u64 *)qbase, (const u64 *)sbase, out_offsets,
                       static_cast<u8 *>(d_out_bytes));
    PSS_HIP(hipMemcpyAsync(out_offsets + E, &sbase[nseg], 8, hipMemcpyDeviceToDevice, s));      // total bytes closes the offsets
    PSS_HIP(hipStreamSynchronize(s));
    PSS_HIP(hipGetLastError());
    (void)B;
    return PSS_OK;
}

}  // namespace pss
