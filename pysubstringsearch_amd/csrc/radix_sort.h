// radix_sort.h -- device LSD radix sort of (u64 key, u32 value) pairs, 8-bit
// digits, hand-written for gfx950 (see radix_sort.hip for the kernel design).
#pragma once
#include "common.h"

namespace pss {

// First-pass source: keys are not materialised, they are packed on the fly
// from the recoded text (codes[i] in 1..sigma, zero past the end of text):
//   key(i) = codes[i] . codes[i+1] ... codes[i+key_chars-1], code_bits each,
//   first symbol most significant;  value(i) = i.
struct TextKeys {
    const uint8_t *codes;   // readable for n + 64 bytes
    int code_bits;
    int key_chars;          // <= 16
    int plus_one;           // 1: codes are raw bytes, symbol = byte + 1 inside the text (sigma == 256)
    int drop = 0;           // low bits of the packed key left out of the sort key (< code_bits): the last
                            // symbol then only contributes its high bits -- a monotone coarsening
};

struct SortStats {
    uint64_t launches = 0;   // scatter-kernel launches
    uint64_t elems = 0;      // elements moved, summed over passes
    double ms = 0.0;         // device time of all scatter launches (profile mode)
    // profile mode, split by scatter-kernel instantiation:
    double ms_text = 0.0;    //   rs_scatter_kernel<true>  (first pass, keys packed from text)
    uint64_t text_launches = 0;
    double ms_pairs = 0.0;   //   rs_scatter_kernel<false> (12 B in + 12 B out per element)
    uint64_t pairs_launches = 0;
    uint64_t pairs_elems = 0;
    uint64_t small_launches = 0;   // single-workgroup LDS sorts (n <= 4096)
};

// Workspace the sort needs besides the ping-pong buffers.
size_t radix_sort_workspace_bytes();

// Sorts n pairs by the key bits [0, 8*ceil(key_bits/8)), least significant
// digit first, skipping every pass p whose bit (1<<p) is clear in pass_mask.
// Input: buffer `src` of the two (keys[i], vals[i]) pairs, or `text` (then the
// first executed pass reads the text and writes buffer 0).  *dst receives the
// index of the buffer holding the sorted pairs (== src if no pass ran; with
// `text` and no pass... not allowed: pass_mask must be non-zero).
// Stable.  `work` must hold radix_sort_workspace_bytes().
int radix_sort_pairs(DeviceCtx *ctx, uint64_t *keys[2], uint32_t *vals[2], uint32_t n, int key_bits,
                     uint32_t pass_mask, const TextKeys *text, int src, void *work, int *dst,
                     bool profile, SortStats *stats, bool ties_last = false);
// ties_last: the last executed pass (which must not be the text pass) writes NO keys; bit 31 of
// every output value is set iff the element's full key equals its predecessor's in the output
// ("tied": not the head of its group).  Values must be < 2^31.

}  // namespace pss
