// text_keys.h -- packing sort keys of consecutive suffixes from the recoded text (shared by the LSD
// passes of radix_sort.hip and the MSD partition of msd_sort.hip).
#pragma once
#include "prims.h"

namespace pss {

constexpr int TK_IPT = 16;   // suffixes per thread

// Packs the keys of 16 consecutive suffixes i0 .. i0+15 (i0 % 16 == 0) from the
// recoded text.  Sliding window: key(i+1) = ((key(i) << b) | code[i+k]) & mask.
__device__ __forceinline__ void text_keys16(const u8 *codes, u32 i0, int b, int k, int plus_one, int drop, u32 n,
                                            u64 (&key)[TK_IPT])
{
    const uint4 *p = reinterpret_cast<const uint4 *>(codes + i0);
    const uint4 lo = p[0], hi = p[1];
    const u64 q0 = (u64)lo.x | ((u64)lo.y << 32), q1 = (u64)lo.z | ((u64)lo.w << 32);
    const u64 q2 = (u64)hi.x | ((u64)hi.y << 32), q3 = (u64)hi.z | ((u64)hi.w << 32);
    const u64 mask = (k * b >= 64) ? ~0ull : ((1ull << (k * b)) - 1ull);
    // first window: bytes 0 .. k-1
    u64 win = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const u64 src = (j < 8) ? q0 : q1;
        u32 c = (u32)(src >> ((j & 7) * 8)) & 0xffu;
        if (plus_one) c = (i0 + j < n) ? c + 1u : 0u;
        if (j < k) win = (win << b) | c;
    }
    key[0] = win >> drop;
    // byte stream starting at byte k (k is wave-uniform): s0 = bytes k..k+7, s1 = k+8..k+15
    u64 a0, a1, a2;
    if (k >= 16) { a0 = q2; a1 = q3; a2 = 0; }
    else if (k >= 8) { a0 = q1; a1 = q2; a2 = q3; }
    else { a0 = q0; a1 = q1; a2 = q2; }
    const int sh = (k & 7) * 8;
    u64 s0 = a0, s1 = a1;
    if (sh) {
        s0 = (a0 >> sh) | (a1 << (64 - sh));
        s1 = (a1 >> sh) | (a2 << (64 - sh));
    }
#pragma unroll
    for (int r = 1; r < TK_IPT; ++r) {
        const int j = r - 1;
        const u64 src = (j < 8) ? s0 : s1;
        u32 c = (u32)(src >> ((j & 7) * 8)) & 0xffu;
        if (plus_one) c = ((u64)i0 + j + k < n) ? c + 1u : 0u;
        win = ((win << b) | c) & mask;
        key[r] = win >> drop;
    }
}

}  // namespace pss
