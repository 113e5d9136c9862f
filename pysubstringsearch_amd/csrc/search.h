// search.h -- batched device search entry point (see search.hip).
#pragma once
#include "common.h"

namespace pss {

// One resident chunk: text (n bytes, readable 16 bytes past the end) and its
// suffix array (n x u32), both in HBM.
struct ChunkDesc {
    const uint8_t *text;
    const uint32_t *sa;
    uint32_t n;
    uint32_t pad;
};

// Host-side packed result of one batch (malloc'ed; owned by pss_result).
struct HostResult {
    uint64_t nq = 0;
    uint64_t n_entries = 0;
    uint64_t *qcount = nullptr;    // [nq]
    uint64_t *offsets = nullptr;   // [n_entries + 1]
    uint8_t *bytes = nullptr;
};

int search_batch_device(DeviceCtx *ctx, const ChunkDesc *d_chunks, uint32_t nc, const uint8_t *qbytes,
                        const uint64_t *qoffsets, uint32_t nq, HostResult *res, pss_search_stats *st,
                        bool counts_only = false);
// counts_only: res->qcount only (entries each query would return); no entry is materialised.

}  // namespace pss
