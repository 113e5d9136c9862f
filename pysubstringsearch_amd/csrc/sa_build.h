// sa_build.h -- device suffix-array builder entry point (see sa_build.hip).
#pragma once
#include "common.h"

namespace pss {

// d_T: n bytes, d_SA: n x int32, both resident on ctx's device.
// flags bit 0: profile mode (HIP events around every radix pass).
int sa_build_device(DeviceCtx *ctx, const void *d_T, void *d_SA, int32_t n, uint32_t flags, pss_sa_stats *stats);

// Suffix array of an integer string of m symbols given as (symbol key, index) pairs sorted by key in
// K[cur] / V[cur] (rle_build.hip); the end of the string is smaller than every symbol.
int suffix_rounds_integer(DeviceCtx *ctx, uint32_t m, uint64_t *K[2], uint32_t *V[2], int cur, uint32_t *SA_out,
                          pss_sa_stats *st);

}  // namespace pss
