// capi.cpp -- the extern "C" surface declared in include/pss.h: container
// writer / reader (the .idx chunk-record format of reference src/lib.rs:105-124
// and 162-199) around the device suffix-array builder and the device search.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/vfs.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "radix_sort.h"
#include "sa_build.h"
#include "search.h"

using namespace pss;

namespace {

constexpr int W_TEXT = 24, W_SA = 25;

// PSS_TIMING=1: phase timings of the host pipeline on stderr
struct Phase {
    const char *name;
    std::chrono::steady_clock::time_point t0;
    static bool on() { static const bool v = knob("PSS_TIMING") != nullptr; return v; }
    explicit Phase(const char *n) : name(n), t0(std::chrono::steady_clock::now()) {}
    ~Phase()
    {
        if (on())
            fprintf(stderr, "[pss] %-22s %8.2f ms\n", name,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};   // DeviceCtx slots used by the writer / host SA entry point

template <typename F>
int guarded(F &&f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        set_error("host allocation failed");
        return PSS_ENOMEM;
    } catch (const std::exception &e) {
        set_error("internal error: %s", e.what());
        return PSS_EDEVICE;
    } catch (...) {
        set_error("internal error");
        return PSS_EDEVICE;
    }
}

int io_error(const char *what)
{
    const int e = errno ? errno : EIO;
    set_error("%s: %s", what, strerror(e));
    errno = e;
    return PSS_EIO;
}

// Host text -> SA on `ctx` (upload, build, download).
int sa_build_host(DeviceCtx *ctx, const uint8_t *T, int32_t *SA, int32_t n, pss_sa_stats *stats)
{
    if (n < 2) {
        if (n == 1) SA[0] = 0;
        return PSS_OK;
    }
    PSS_TRY(ctx->slot[W_TEXT].reserve((size_t)n + 64));
    PSS_TRY(ctx->slot[W_SA].reserve((size_t)n * 4));
    PSS_HIP(hipMemcpyAsync(ctx->slot[W_TEXT].p, T, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    PSS_TRY(sa_build_device(ctx, ctx->slot[W_TEXT].p, ctx->slot[W_SA].p, n, 0, stats));
    PSS_HIP(hipMemcpyAsync(SA, ctx->slot[W_SA].p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(hipStreamSynchronize(ctx->stream));
    return PSS_OK;
}

}  // namespace

// ------------------------------------------------------------------ library --

extern "C" int pss_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return c;
}

// Which devices a handle opened without an explicit device uses (include/pss.h).  The reference fans every search over
// all the cores of the machine without being told to (rayon's global pool, src/lib.rs:205-207): the unchanged drop-in
// call uses every GPU the process can see, unless a launcher pinned it to one.
extern "C" int32_t pss_default_devices(int32_t *out, int32_t cap)
{
    if (!out || cap < 1) return 0;
    const int count = pss_device_count();
    auto digits = [](const char *v) {
        if (!v || !*v) return false;
        for (const char *c = v; *c; ++c)
            if (*c < '0' || *c > '9') return false;
        return true;
    };
    const char *e = knob("PSS_DEVICES");
    if (e && *e) {
        if (strcmp(e, "all") == 0) {
            int32_t k = 0;
            for (; k < count && k < cap; ++k) out[k] = k;
            if (k == 0) out[k++] = 0;
            return k;
        }
        // a comma-separated list of ordinals (one may be named more than once: "virtual devices")
        int32_t k = 0;
        bool ok = true;
        const char *c = e;
        while (ok && *c && k < cap) {
            char *end = nullptr;
            const long v = strtol(c, &end, 10);
            if (end == c || v < 0 || (count > 0 && v >= count)) { ok = false; break; }
            out[k++] = (int32_t)v;
            c = end;
            if (*c == ',') ++c;
            else if (*c) ok = false;
        }
        if (ok && k > 0 && (!*c || k == cap)) return k;       // (a list longer than the caller's room: its first cap ordinals)
        // A list that does not parse is an ERROR (round 5), not "as if unset": under a launcher every rank would otherwise
        // open its handles on all visible GPUs at once.
        set_error("PSS_DEVICES=%s is neither 'all' nor a comma-separated list of device ordinals below %d", e, count);
        return -1;
    }
    // unset (or set to nothing): one process per GPU under a launcher -- ours, torchrun's, Slurm's, Open MPI's, MVAPICH's
    for (const char *var : {"PSS_DEVICE", "LOCAL_RANK", "SLURM_LOCALID", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK"}) {
        const char *v = getenv(var);      // (launchers' variables, read as they are: not switches of this library)
        if (digits(v)) {
            const long d = strtol(v, nullptr, 10);
            out[0] = (int32_t)(count > 0 ? d % count : d);
            return 1;
        }
    }
    int32_t k = 0;
    for (; k < count && k < cap; ++k) out[k] = k;
    if (k == 0) out[k++] = 0;       // no usable device: the first compute call says so
    return k;
}

extern "C" size_t pss_last_error(char *buf, size_t cap)
{
    const std::string &e = last_error();
    if (buf && cap) {
        const size_t k = e.size() < cap - 1 ? e.size() : cap - 1;
        memcpy(buf, e.data(), k);
        buf[k] = 0;
    }
    return e.size();
}

// Buffers a Writer needs again the next time one is opened in this process (round 4): its pinned stages (8 x 16 MiB) and up
// to three host text buffers of a chunk each.  Allocating, faulting in and unmapping 512 MiB buffers and pinning / unpinning
// the stages cost a Writer of one chunk 0.25 s at close and 0.1 s on the way -- of 0.7 s in all; a process that writes
// index after index pays that once.  pss_release_workspace() gives everything back.
namespace {
std::mutex g_wcache_mu;
std::vector<void *> g_stage_cache;                                  // pinned, DeviceCtx::kIoPiece each
std::vector<std::pair<uint8_t *, size_t>> g_text_cache;             // malloc'ed
constexpr size_t kTextCacheMax = 3, kTextCacheMaxBytes = (size_t)3 << 30, kStageCacheMax = 16;

void *stage_cache_take()
{
    std::lock_guard<std::mutex> lk(g_wcache_mu);
    if (g_stage_cache.empty()) return nullptr;
    void *p = g_stage_cache.back();
    g_stage_cache.pop_back();
    return p;
}
void stage_cache_give(void *p)
{
    {
        std::lock_guard<std::mutex> lk(g_wcache_mu);
        if (g_stage_cache.size() < kStageCacheMax) {
            g_stage_cache.push_back(p);
            return;
        }
    }
    (void)hipHostFree(p);
}
// the smallest cached text buffer of at least `need` bytes (nullptr: none)
uint8_t *text_cache_take(size_t need, size_t *cap)
{
    std::lock_guard<std::mutex> lk(g_wcache_mu);
    int best = -1;
    for (int i = 0; i < (int)g_text_cache.size(); ++i)
        if (g_text_cache[i].second >= need && (best < 0 || g_text_cache[i].second < g_text_cache[best].second)) best = i;
    if (best < 0) return nullptr;
    uint8_t *p = g_text_cache[best].first;
    *cap = g_text_cache[best].second;
    g_text_cache.erase(g_text_cache.begin() + best);
    return p;
}
void text_cache_give(uint8_t *p, size_t cap)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_wcache_mu);
        size_t held = 0;
        for (const auto &t : g_text_cache) held += t.second;
        if (cap >= ((size_t)1 << 20) && g_text_cache.size() < kTextCacheMax && held + cap <= kTextCacheMaxBytes) {
            g_text_cache.emplace_back(p, cap);
            return;
        }
    }
    free(p);
}
void writer_caches_release()
{
    std::vector<void *> st;
    std::vector<std::pair<uint8_t *, size_t>> tx;
    {
        std::lock_guard<std::mutex> lk(g_wcache_mu);
        st.swap(g_stage_cache);
        tx.swap(g_text_cache);
    }
    for (void *p : st) (void)hipHostFree(p);
    for (auto &t : tx) free(t.first);
}
}  // namespace

extern "C" int pss_release_workspace(void)
{
    return guarded([&]() -> int {
        writer_caches_release();
        trim_all();
        return PSS_OK;
    });
}

extern "C" uint64_t pss_workspace_bytes(int32_t device) { return workspace_bytes(device); }

extern "C" int32_t pss_knob_count(void) { return kNumKnobs; }
extern "C" int pss_knob_info(int32_t i, const char **name, const char **dflt, const char **fuzz, const char **what)
{
    if (i < 0 || i >= kNumKnobs) return PSS_EINVAL;
    if (name) *name = kKnobs[i].name;
    if (dflt) *dflt = kKnobs[i].dflt;
    if (fuzz) *fuzz = kKnobs[i].fuzz;
    if (what) *what = kKnobs[i].what;
    return PSS_OK;
}

extern "C" uint64_t pss_sa_stats_size(void) { return sizeof(pss_sa_stats); }
extern "C" uint64_t pss_search_stats_size(void) { return sizeof(pss_search_stats); }

// --------------------------------------------------------------- SA builder --

extern "C" int32_t pss_sa_build(const uint8_t *T, int32_t *SA, int32_t n, int32_t device)
{
    return guarded([&]() -> int {
        if (T == nullptr || SA == nullptr || n < 0) {   // libsais.c:6599-6602
            set_error("pss_sa_build: bad arguments");
            return PSS_EINVAL;
        }
        if (n < 2) {                                    // libsais.c:6603-6607
            if (n == 1) SA[0] = 0;
            return PSS_OK;
        }
        DeviceCtx *ctx;
        PSS_TRY(get_build_ctx(device, &ctx));
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        return sa_build_host(ctx, T, SA, n, nullptr);
    });
}

extern "C" int32_t pss_sa_build_device(const void *d_T, void *d_SA, int32_t n, int32_t device, uint32_t flags,
                                       pss_sa_stats *stats)
{
    return guarded([&]() -> int {
        DeviceCtx *ctx;
        PSS_TRY(get_build_ctx(device, &ctx));
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        return sa_build_device(ctx, d_T, d_SA, n, flags, stats);
    });
}

extern "C" int32_t pss_sort_pairs_device(void *d_keys, void *d_vals, uint32_t n, int32_t key_bits, int32_t device,
                                         double *ms_scatter)
{
    return guarded([&]() -> int {
        if ((n && (!d_keys || !d_vals)) || key_bits < 1 || key_bits > 64) {
            set_error("pss_sort_pairs_device: bad arguments");
            return PSS_EINVAL;
        }
        DeviceCtx *ctx;
        PSS_TRY(get_build_ctx(device, &ctx));
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        if (n == 0) return PSS_OK;
        // slots of the SA builder double as the ping-pong partner and workspace
        PSS_TRY(ctx->slot[1].reserve((size_t)n * 8));
        PSS_TRY(ctx->slot[3].reserve((size_t)n * 4));
        PSS_TRY(ctx->slot[9].reserve(radix_sort_workspace_bytes() + 65536));
        uint64_t *K[2] = {static_cast<uint64_t *>(d_keys), ctx->slot[1].as<uint64_t>()};
        uint32_t *V[2] = {static_cast<uint32_t *>(d_vals), ctx->slot[3].as<uint32_t>()};
        SortStats st;
        int dst = 0;
        PSS_TRY(radix_sort_pairs(ctx, K, V, n, key_bits, 0xffffffffu, nullptr, 0, ctx->slot[9].p, &dst, ms_scatter != nullptr, &st));
        if (dst != 0) {
            PSS_HIP(hipMemcpyAsync(d_keys, K[1], (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream));
            PSS_HIP(hipMemcpyAsync(d_vals, V[1], (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
        PSS_HIP(hipStreamSynchronize(ctx->stream));
        if (ms_scatter) *ms_scatter = st.ms;
        return PSS_OK;
    });
}

#include "capi_writer_impl.h"

#include "capi_reader_impl.h"

#include "capi_comm_impl.h"

}

// Test hook: re-reads the PSS_* environment switches of the search path (they are read once, when the
// library first needs them -- not on every call).
extern "C" int pss_reload_env(void)
{
    reload_search_knobs();
    return PSS_OK;
}

extern "C" int pss_reader_last_stats(const pss_reader *r, pss_search_stats *stats)
{
    if (!r || !stats) return PSS_EINVAL;
    *stats = r->last;
    return PSS_OK;
}

extern "C" int pss_reader_close(pss_reader *r)
{
    return guarded([&]() -> int {
        reader_free(r);
        return PSS_OK;
    });
}

extern "C" uint64_t pss_result_num_queries(const pss_result *res) { return res ? res->r.nq : 0; }
extern "C" uint64_t pss_result_num_entries(const pss_result *res) { return res ? res->r.n_entries : 0; }
extern "C" const uint64_t *pss_result_query_counts(const pss_result *res) { return res ? res->r.qcount : nullptr; }
extern "C" const uint64_t *pss_result_offsets(const pss_result *res) { return res ? res->r.offsets : nullptr; }
extern "C" const uint8_t *pss_result_bytes(const pss_result *res) { return res ? res->r.bytes : nullptr; }
extern "C" void pss_result_free(pss_result *res)
{
    if (!res) return;
    res->r.release();
    delete res;
}
