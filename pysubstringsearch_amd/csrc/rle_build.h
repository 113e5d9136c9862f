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

}  // namespace pss
