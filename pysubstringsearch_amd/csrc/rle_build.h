// rle_build.h -- suffix array of a text that consists of long runs of equal bytes (see rle_build.hip).
#pragma once
#include "common.h"

namespace pss {

struct RleStats {
    uint32_t runs = 0;          // maximal runs of equal bytes
    bool columns = false;       // expansion by the matrix walk (no radix sort)
    uint32_t id_bits = 0;       // bits of the (class, remaining length) key of the expansion sort
    double ms_table = 0, ms_reduced = 0, ms_expand = 0;   // profile mode
};

// SA[0..n) := suffix array of T[0..n), by way of the run-length reduced string (one symbol per run).
// `runs` = number of maximal runs of T (the caller counted them).  Correct for every text; worth it when
// the runs are long.  st receives the rounds of the reduced string's sort.
int rle_suffix_array(DeviceCtx *ctx, const uint8_t *T, uint32_t n, uint32_t runs, uint32_t *SA, bool profile, RleStats *rs,
                     pss_sa_stats *st);

// ---- texts that repeat one word (rle_build.hip, "periodic prefix") ----
// A text whose first m bytes have period p (T[i] == T[i + p] for i + p < m: one line, record or frame written over and
// over) ties every suffix with n / p others for as long as the repetition lasts -- prefix doubling at full activity
// for log2(m) rounds.  Its suffix array has a closed form instead: suffixes that start at least `margin` bytes before
// the repetition ends order by (the rotation of the word they start in, then by position -- ascending or descending,
// decided by the one byte that ends the repetition), and the few that start later (the last margin bytes and the
// n - m bytes behind the repetition) are sorted on the host and slotted between the rotation blocks.
constexpr uint32_t kPeriodMax = 1024;        // longest word looked for
constexpr uint32_t kPeriodTailMax = 1024;    // most bytes behind the repetition
constexpr uint32_t kPeriodProbe = 8192;      // bytes of the head of the text the word is looked for in

// Smallest p in [2, kPeriodMax] with head[i] == head[i + p] for all i + p < len (0: none).  Host, on the probe.
uint32_t period_of_head(const uint8_t *head, uint32_t len);
// m = how far the repetition with word length p reaches from the start of T (device pass; synchronises).
int period_extent(DeviceCtx *ctx, const uint8_t *T, uint32_t n, uint32_t p, uint32_t *m);
// SA := suffix array of T[0..n), T[0..m) of smallest period p, n - m <= kPeriodTailMax.  *accepted = false: the text
// is not of the shape this handles (too short a repetition), SA untouched.
int period_suffix_array(DeviceCtx *ctx, const uint8_t *T, uint32_t n, uint32_t p, uint32_t m, const uint8_t *head /* host: T[0..p) */,
                        uint32_t *SA, bool *accepted);

}  // namespace pss
