// sa_build.h -- device suffix-array builder entry point (see sa_build.hip).
#pragma once
#include "common.h"

namespace pss {

// d_T: n bytes, d_SA: n x int32, both resident on ctx's device.
// flags bit 0: profile mode (HIP events around every radix pass).
int sa_build_device(DeviceCtx *ctx, const void *d_T, void *d_SA, int32_t n, uint32_t flags, pss_sa_stats *stats);

}  // namespace pss
