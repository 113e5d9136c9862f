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
    // flag-carrying passes of the initial suffix sort, fs_scatter_kernel<KIN, KOUT>:
    // index = (KIN / 4) * 3 + KOUT / 4   (KIN 0 = keys packed from the text, KOUT 0 = last pass)
    double fs_ms[9] = {};
    uint64_t fs_launches[9] = {};
    uint64_t fs_elems[9] = {};
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
                     bool profile, SortStats *stats);

// Initial suffix sort: value(i) = i for all n suffixes, sorted (stably) by the low key_bits of
// (packed text key >> text->drop).  Every pass stores only the digits it has not consumed yet
// (8-byte key plane while more than 32 bits remain, then 4 bytes, none in the last pass) and
// carries in bit 31 of each value whether the element's consumed digits equal its predecessor's;
// after the last pass that bit says "full key equal to my predecessor's" (tied: not the head of
// its group).  Needs key_bits > 8 (at least two passes).  Pass p writes buffer p & 1; *dst
// receives the buffer index of the final values.
int suffix_sort_flags(DeviceCtx *ctx, uint64_t *keys[2], uint32_t *vals[2], uint32_t n, int key_bits,
                      const TextKeys *text, void *work, int *dst, bool profile, SortStats *stats);

}  // namespace pss
