// common.h -- host-side plumbing shared by the engine: error reporting across
// the C ABI, per-device context (stream + grow-only HBM workspace slots).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/pss.h"

namespace pss {

void set_error(const char *fmt, ...);
const std::string &last_error();

#define PSS_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            pss::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (_e == hipErrorOutOfMemory) ? PSS_ENOMEM : PSS_EDEVICE;              \
        }                                                                               \
    } while (0)

#define PSS_TRY(expr)               \
    do {                            \
        int _rc = (expr);           \
        if (_rc != PSS_OK) return _rc; \
    } while (0)

// A grow-only device allocation, reused across calls.
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

// One per (process, device): a stream and named workspace slots.
struct DeviceCtx {
    // Workspace slots, staging buffers and the stream are shared by every handle on the
    // device; ctypes releases the GIL, so two Python threads can be inside the library at
    // once.  Every entry point that touches the context holds this lock for its duration.
    std::recursive_mutex mu;
    int device = -1;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    static constexpr int kSlots = 32;
    DevBuf slot[kSlots];
    static constexpr size_t kPinnedBytes = (size_t)192 << 10;   // 64 KiB of counters / small tables + 128 KiB of result bytes
    void *pinned = nullptr;   // small pinned host scratch for D2H of counters
    size_t pinned_cap = 0;
    void *pinned_dev = nullptr;          // the same memory as the device sees it (zero-copy reads / writes)
    // Pinned staging of the batched search (queries up, results down): copies from / to pageable
    // memory are synchronous and slow to start, DMA from / to pinned memory is not.
    static constexpr size_t kStageQ = (size_t)2 << 20, kStageR = (size_t)4 << 20;
    void *search_stage = nullptr;        // kStageQ + kStageR bytes, allocated on first use
    int ensure_search_stage();
    hipEvent_t search_ev[3] = {nullptr, nullptr, nullptr};   // timing events of the search path, created once
    void *small_hdr_ready = nullptr;     // arena whose small-path cursors have been zeroed (search.hip)
    // Two pinned staging buffers + a copy stream: file <-> HBM transfers are
    // double-buffered so the PCIe copy of piece i overlaps the file I/O of piece i+1.
    static constexpr size_t kStage = (size_t)64 << 20;
    void *stage[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    int ensure_staging();
};

// Validates `device`, makes it current, returns its context (created lazily).
int get_ctx(int device, DeviceCtx **out);
// Frees every workspace slot of every context (memory pressure relief).
void trim_all();

static inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace pss
