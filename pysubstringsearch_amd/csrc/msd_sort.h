// msd_sort.h -- initial suffix sort as a hybrid MSD radix sort (see msd_sort.hip).
#pragma once
#include "common.h"
#include "radix_sort.h"

namespace pss {

struct MsdStats {
    uint32_t buckets = 0;      // non-empty joint (20-bit) buckets
    uint32_t max_bucket = 0;   // largest of them
    uint32_t tiles = 0;        // local-sort workgroups
    uint32_t slow_tiles = 0;   // of them, tiles the counting kernel handed to the general (ballot LSD) kernel
    uint32_t bad_symbol = 0;   // MsdFront: a byte of the text had no code in the table it was given
    uint32_t lookback = 0;     // digits in LSD order, second pass in one sweep (look-back)
    double ms_g1 = 0, ms_g2 = 0, ms_local = 0;   // profile mode: the two partition scatters and the local sort
};

// Optional fusion of the first rerank: instead of flagging ties in bit 31 of sa_out (left clean), the
// local sort emits the active list itself -- for every suffix that is tied with a neighbour its SA slot,
// its index and 1 + the slot of its group's head, in slot order -- which is exactly what rr_apply_tied
// would compute from the flags with two more passes over the 4 n-byte array.
struct MsdActive {
    uint32_t *pos, *idx, *grp;               // the list (capacity n each)
    uint32_t *st_pos, *st_idx;               // staging of the same size (per-tile blocks before they are lined up)
    uint32_t count = 0;                      // out: entries in the list
};

// Bytes of workspace msd_suffix_sort needs besides the two 8 n-byte element buffers.
size_t msd_workspace_bytes(uint32_t n);
// Largest key width (bits of packed key >> drop) the 8-byte elements can carry for n suffixes.
int msd_max_key_bits(uint32_t n);

// Same contract as suffix_sort_flags: sa_out[i] = index of the i-th smallest suffix by the low key_bits
// of (packed text key >> text->drop), bit 31 = "same key as my predecessor".  The order among equal
// keys is unspecified (suffix_sort_flags leaves them by index; nothing downstream relies on that).
// *accepted = false (and sa_out untouched) when some joint bucket exceeds what a workgroup sorts in
// LDS: the caller then uses suffix_sort_flags.  h_small: >= 64 bytes of pinned host memory.
// `front` (optional): the codes do not exist yet -- the first histogram pass makes them on its way through the raw text
// (codes[i] = lut[T[i]], the buffer behind text->codes is written; its padding past n must be zero already) and raises
// *bad when a byte has no code in `lut` (then the sort declines: *accepted = false, stats->bad_symbol = 1).
struct MsdFront {
    const uint8_t *raw;     // T, 16-byte aligned
    const uint8_t *lut;     // device, [256]
    uint32_t *bad;          // device, zeroed by the caller
};
int msd_suffix_sort(DeviceCtx *ctx, const TextKeys *text, uint32_t n, int key_bits, uint64_t *A[2], uint32_t *sa_out,
                    void *work, uint32_t *h_small, bool profile, MsdStats *stats, bool *accepted, MsdActive *active = nullptr,
                    const MsdFront *front = nullptr);

// ---- sample sort over 16-byte elements (ss_sort_impl.h): the initial sort of natural text ----

struct SsStats {
    uint32_t buckets = 0, max_bucket = 0, tiles = 0;
    uint32_t b1 = 0, b2 = 0, samples = 0;
    int key_chars = 0;
    double ms_sample = 0, ms_g1 = 0, ms_g2 = 0, ms_local = 0;      // profile mode
};
struct SsBuffers {
    void *A[2];            // 16 n bytes each: the element buffers of the two partition passes
    uint16_t *digits;      // n + 64 entries
    void *E0, *E;          // 16 S bytes each (S = ss_sample_count(n)): the sample, unsorted / sorted
    uint64_t *K[2];        // >= S entries each: scratch of the sample's sort
    uint32_t *V[2];
    void *sort_work;       // radix_sort_workspace_bytes()
    // The plan of the sample sort (round 4): the SORTED sample of the previous chunk of the same corpus cuts this chunk
    // as well as a sample of its own would -- bucket sizes follow the same distribution either way (the noise is in the
    // four sample members per bucket, not in which chunk they came from) -- and any splitters give the exact result.
    const void *sample_in = nullptr;     // use these S sorted sample elements, draw and sort none
    void *sample_keep = nullptr;         // a copy of the sorted sample goes here (16 S bytes)
};
// Sample members drawn for n suffixes (0: the path does not take texts of this size).
uint32_t ss_sample_count(uint32_t n);
// Symbols a 16-byte element carries next to the index of one of n suffixes when the text has radix - 1 distinct bytes
// (the key is a base-radix number, code 0 = past the end of the text).
int ss_key_chars(uint32_t n, uint32_t radix);
uint64_t ss_geometry_tag(uint32_t n, uint32_t radix);
// Same contract as suffix_sort_flags / msd_suffix_sort without `active` (bit 31 of sa_out[i] = "same key as my
// predecessor"); the key is the first ss_key_chars() symbols.  `work`: msd_workspace_bytes(n).
int ss_suffix_sort(DeviceCtx *ctx, const TextKeys *text, uint32_t radix, uint32_t n, const SsBuffers &buf, uint32_t *sa_out,
                   void *work, uint32_t *h_small, bool profile, SsStats *stats, bool *accepted, MsdActive *active);

}  // namespace pss
