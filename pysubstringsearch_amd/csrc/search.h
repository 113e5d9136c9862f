// search.h -- batched device search entry point (see search.hip).
#pragma once
#include "common.h"

namespace pss {

// One resident chunk: text (n bytes, zero padded and readable 128 bytes past the end), its
// suffix array (n x u32) and a table of key samples: skeys[j] = the first 8 bytes of suffix
// sa[j << shift] as a big-endian integer (ceil(n / 2^shift) entries, nullptr = none).  The
// keys are non-decreasing along the suffix array, so two searches in the small table (L2
// resident) confine a query to a window of one or two strides before the suffix array and
// the text are touched.
struct ChunkDesc {
    const uint8_t *text;
    const uint32_t *sa;
    const uint64_t *skeys;
    uint32_t n;
    uint32_t shift;
};
constexpr uint32_t kSampleShift = 11;   // one key sample per 2048 suffixes: 2 MiB per 512 MiB chunk

// Entries of the key-sample table of an n-byte chunk, and the kernel launch that fills it
// (stream-ordered on ctx->stream; `skeys` must hold sample_count(n, shift) entries).
static inline uint64_t sample_count(uint32_t n, uint32_t shift) { return ((uint64_t)n + (1u << shift) - 1) >> shift; }
int build_key_samples(DeviceCtx *ctx, const uint8_t *d_text, const uint32_t *d_sa, uint32_t n, uint32_t shift,
                      uint64_t *d_skeys);

// Host-side packed result of one batch (owned by pss_result).  Small results are malloc'ed; large
// ones (offsets + bytes) share ONE pinned block from the pool in common.h, so the D2H copy runs at
// link speed.  In SEARCH_DEVICE mode nothing but the totals comes down: d_* point into the device
// context's workspace and stay valid until the next search on that device.
struct HostResult {
    uint64_t nq = 0;
    uint64_t n_entries = 0;
    uint64_t n_bytes = 0;
    uint64_t *qcount = nullptr;    // [nq]
    uint64_t *offsets = nullptr;   // [n_entries + 1]
    uint8_t *bytes = nullptr;
    void *pinned = nullptr;        // when set: offsets and bytes live inside this block
    size_t pinned_bytes = 0;
    const uint64_t *d_qcount = nullptr;    // SEARCH_DEVICE: [nq]
    const uint64_t *d_offsets = nullptr;   //                [n_entries] entry starts
    const uint8_t *d_bytes = nullptr;      //                [n_bytes]
    void release();
};

// Room for E + 1 offsets and B bytes in `res`: one block of the pinned pool when the result is large (the D2H copy then
// runs at link speed), else malloc.
int alloc_host_result(HostResult *res, uint64_t E, uint64_t B, bool allow_pinned);

enum SearchMode {
    SEARCH_FULL = 0,     // packed result on the host
    SEARCH_COUNTS = 1,   // res->qcount only (entries each query would return); no entry is materialised
    SEARCH_DEVICE = 2,   // packed result left on the device (multi-GPU gather over RCCL takes it from there)
};

// chunk_hits (optional, nc entries): how much of the batch's work landed on every chunk -- suffix-array hits per chunk on
// the multi-kernel paths, entries per chunk on the fused small-batch path.  The reader's residency manager feeds on it
// (capi.cpp); it costs one small kernel and a stream synchronisation, so readers whose chunks all live in HBM pass nullptr.
int search_batch_device(DeviceCtx *ctx, const ChunkDesc *d_chunks, uint32_t nc, const uint8_t *qbytes,
                        const uint64_t *qoffsets, uint32_t nq, HostResult *res, pss_search_stats *st,
                        SearchMode mode = SEARCH_FULL, bool low_latency = false, uint64_t *chunk_hits = nullptr,
                        bool sa_order = false);
// sa_order: the entries of one (query, chunk) pair come out in the reference's order -- suffix-array order of the FIRST hit
// inside each entry (src/lib.rs:262-276: the hits are walked in suffix-array order and an entry is pushed when its line
// start is first seen) -- instead of the order of each entry's leftmost match.  Opt-in (pss_reader_set_result_order): it
// takes the general pipeline and one extra sort of the hits.

// Merge of `world` packed results of the same nq queries, all resident on ctx's device, into one (query-major,
// rank-major inside a query -- pss_merge_packed's order) on the same device.  starts[r] = entry starts (no closing
// offset); out_offsets receives sum(num_entries) + 1 entries.  Synchronises the stream.
int merge_packed_device(DeviceCtx *ctx, uint32_t world, uint64_t nq, const void *const *d_counts, const void *const *d_starts,
                        const void *const *d_bytes, const uint64_t *num_entries, const uint64_t *num_bytes, void *d_out_counts,
                        void *d_out_offsets, void *d_out_bytes);

}  // namespace pss
