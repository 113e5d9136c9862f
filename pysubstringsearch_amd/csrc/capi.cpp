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

// ------------------------------------------------------------------- Writer --

// The Writer is a three-stage pipeline so that ingest, suffix-array builds and the file never wait for
// one another more than the data dependencies demand:
//
//   caller thread     fills the host text buffer of chunk k (add_entry / add_entries_from_file_lines) and,
//                     when the chunk is full, queues it as job k and goes on with a fresh buffer;
//   builder threads   one per device of the writer: chunk k is built on device k mod G (upload, device
//                     suffix-array build into one of that device's two SA buffers) -- with G devices, G
//                     chunks are being built at once (reference: one libsais call at a time, lib.rs:105-124);
//   record thread     writes the records strictly in chunk order (lib.rs:112-119), streaming each suffix
//                     array HBM -> pinned double buffer -> file, so the file is byte-identical whatever G is.
//
// A device's SA buffer is reused two chunks of that device later, hence at most 2 G jobs are in flight.
// The first failure (build or write) is sticky: nothing is written after it and every later dump /
// finalize / close reports it.
struct WJob {
    uint64_t seq = 0;
    uint8_t *text = nullptr;      // host text, owned by the job until its record is written
    size_t text_alloc = 0;
    size_t n = 0;
    enum State { QUEUED, BUILDING, BUILT } state = QUEUED;
};

// Striped layout of format 2 (opt-in, round 5): header flags bit 0 set, bits 8..15 = S stripe files, bits 16..23 = log2 of
// the stripe unit.  The records of the index file then hold no suffix array (u64 n | text | u64 4n); the arrays live in
// `<path>.sa0` .. `<path>.sa<S-1>`: every chunk's array starts a new unit, unit u sits in file u mod S at offset
// (u / S) * unit.  Why: ONE file in the page cache takes 11 - 14 GB/s on the test box however many threads write it (the
// inode's lock), a file per writer 47 - 97 GB/s (profiles/r04_pagecache_micro.txt) -- and the suffix arrays are 4/5 of
// the bytes.  The reference container (and format 2 without the flag) stay as they are.
constexpr uint32_t kStripedFlag = 1u;
constexpr int kStripeUnitLog = 24;                     // = DeviceCtx::kIoPiece: one piece of the I/O pool per unit
static_assert(((size_t)1 << kStripeUnitLog) == DeviceCtx::kIoPiece, "a stripe unit is one piece of the I/O pool");
struct Stripes {
    std::vector<int> fd;
    uint64_t next_unit = 0;                            // first unit of the next chunk's suffix array
    int S() const { return (int)fd.size(); }
    // closes every stripe file; returns the errno of the first close() that failed (0: none) -- four fifths of a striped
    // index's bytes live in these files, their close is where a full disk or a lost NFS write shows up
    int close_all()
    {
        int first = 0;
        for (int f : fd)
            if (f >= 0 && close(f) != 0 && first == 0) first = errno ? errno : EIO;
        fd.clear();
        return first;
    }
    static std::string name(const char *path, int j) { return std::string(path) + ".sa" + std::to_string(j); }
};

struct WDevice {
    int device = 0;
    DevBuf sa[2];                 // suffix arrays in HBM: one being written out, one being built
    hipStream_t io_stream = nullptr;
    hipEvent_t ev[8] = {};        // one per staging piece of the record thread (kWPieces)
    std::thread builder;
};

constexpr int kWPieces = 8;       // pinned staging pieces of the record thread (DeviceCtx::kIoPiece bytes each)

struct pss_writer {
    int fd = -1;                  // the index file: records are written with pwrite at offsets known in advance
    int64_t pos = 0;              // where the next record starts
    bool no_mmap = true;          // records through pwrite (false: through a shared mapping -- see write_record)
    Stripes stripes;              // striped layout: the suffix arrays' files (empty: arrays inline, as in the reference)
    int map_fd = -1;              // the same file opened for reading AND writing: a shared mapping needs both (the index file
                                  // itself is opened like File::create, write-only -- mmap on that fd fails with EACCES)
    size_t mmap_min = (size_t)1 << 20;     // records below this go through pwrite (PSS_WRITER_MMAP_MIN)
    std::atomic<uint64_t> records_mapped{0}, records_pwritten{0};      // which way the records went (pss_writer_io_stats)
    uint64_t ingest_direct = 0, ingest_copied = 0;           // file bytes read straight into the chunk / through a block buffer
    uint8_t *buf = nullptr;
    size_t len = 0;
    size_t limit = 0;    // the reference's Vec capacity (src/lib.rs:62), see reserve()
    size_t alloc = 0;
    int version = 1;              // container format: 1 = the reference's (lib.rs:112-119), 2 = 64-bit lengths
    std::vector<WDevice> devs;    // chunk k is built on devs[k % devs.size()]
    // pipeline state, guarded by mu
    std::mutex mu;
    std::condition_variable cv;
    std::deque<WJob> jobs;        // jobs[i].seq == written + i
    uint64_t next_seq = 0, written = 0;
    bool started = false, stop = false;
    int rc = PSS_OK;              // first failure of any stage (sticky)
    int err_no = 0;
    std::string err;
    std::vector<std::pair<uint8_t *, size_t>> free_text;   // host text buffers back from written jobs (at most G + 1 kept)
    size_t inflight_text = 0;            // bytes of host text owned by jobs that are queued, building or being written
    size_t text_budget = (size_t)8 << 30;   // ... bounded by this (PSS_WRITER_HOST_BUDGET), not only by 2 G jobs
    std::thread record_thread;
    void *stage[kWPieces] = {};                             // pinned staging of the record thread
};

namespace {

// The reference's chunk limit is the capacity of a Rust Vec<u8> (lib.rs:62,75,
// 92,96).  Appending past it grows the Vec by the standard amortised rule
// new_cap = max(8, 2*cap, len+additional), which silently raises the limit;
// mirrored here so chunk boundaries stay byte-identical even in that corner.
int w_reserve(pss_writer *w, size_t additional)
{
    if (w->limit - w->len < additional) {
        size_t nc = w->limit * 2;
        if (nc < w->len + additional) nc = w->len + additional;
        if (nc < 8) nc = 8;
        w->limit = nc;
    }
    const size_t need = w->len + additional;
    if (need > w->alloc && w->buf == nullptr && w->len == 0) {
        // (a buffer an earlier Writer of this process left behind: a whole chunk's worth, its pages already there)
        size_t cap = 0;
        if (uint8_t *p = text_cache_take(need, &cap)) {
            w->buf = p;
            w->alloc = cap;
        }
    }
    if (need > w->alloc) {
        size_t na = w->alloc ? w->alloc : 65536;
        while (na < need) na *= 2;
        uint8_t *nb = static_cast<uint8_t *>(realloc(w->buf, na));
        if (!nb) {
            set_error("host allocation of %zu bytes failed", na);
            return PSS_ENOMEM;
        }
        w->buf = nb;
        w->alloc = na;
    }
    return PSS_OK;
}

int w_append(pss_writer *w, const uint8_t *p, size_t l)
{
    PSS_TRY(w_reserve(w, l));
    if (l) memcpy(w->buf + w->len, p, l);
    w->len += l;
    PSS_TRY(w_reserve(w, 1));
    w->buf[w->len++] = '\n';
    return PSS_OK;
}

// Container format 2 (opt-in; the reference format stays the default):
//   file   = "PSSIDX\x02\x00" | u32le flags (0) | u32le reserved (0) | record*
//   record = u64le n | n bytes of text | u64le 4n | n x i32le
// i.e. the reference's record with 64-bit lengths: the u32 at lib.rs:116 wraps from 1 GiB of text on,
// here a chunk may hold up to 2^31 - 1 bytes (the suffix array stays int32).
constexpr uint8_t kMagicV2[8] = {'P', 'S', 'S', 'I', 'D', 'X', 2, 0};
constexpr size_t kHeaderV2 = 16;

void put_u64le(uint8_t *p, uint64_t v)
{
    for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(v >> (8 * i));
}

void put_u32le(uint8_t *p, uint32_t v)
{
    p[0] = (uint8_t)v;
    p[1] = (uint8_t)(v >> 8);
    p[2] = (uint8_t)(v >> 16);
    p[3] = (uint8_t)(v >> 24);
}

int pwrite_all(int fd, const void *buf, size_t len, int64_t off)
{
    const uint8_t *p = static_cast<const uint8_t *>(buf);
    size_t at = 0;
    while (at < len) {
        const ssize_t k = pwrite(fd, p + at, len - at, (off_t)(off + (int64_t)at));
        if (k < 0) {
            if (errno == EINTR) continue;
            return io_error("write");
        }
        at += (size_t)k;
    }
    return PSS_OK;
}

// Streams `bytes` of device memory to the file at `off`: D2H copies into a ring of pinned pieces on the owning device's
// copy stream, every piece handed to the I/O pool (pwrite at its own offset) as soon as it has landed -- the copy of
// piece i + 1 runs while pieces <= i are being written by several threads.
int download_to_file(pss_writer *w, WDevice &d, const void *src, size_t bytes, int64_t off, IoPool::Batch *batch, uint8_t *map,
                     uint64_t unit_base = 0)
{
    const size_t piece = DeviceCtx::kIoPiece;
    const size_t pieces = (bytes + piece - 1) / piece;
    IoPool &pool = IoPool::get();
    std::atomic<int> done[kWPieces];
    for (auto &x : done) x.store(1);
    static const bool drop = knob("PSS_EXPERIMENT_NO_FILE") != nullptr;     // measurement only: the copies without the file
    auto put = [&](size_t i) {          // piece i has landed in its staging buffer: to the pool
        const size_t o = i * piece, k = std::min(piece, bytes - o);
        if (drop) return;
        if (w->stripes.S()) {             // striped layout: piece i is unit unit_base + i of the suffix arrays' files
            const uint64_t u = unit_base + i;
            const int S = w->stripes.S();
            pool.submit(batch, w->stripes.fd[(size_t)(u % (uint64_t)S)], true, w->stage[i % kWPieces], k,
                        (int64_t)((u / (uint64_t)S) * piece), &done[i % kWPieces]);
        } else if (map) pool.submit_copy(batch, map + o, w->stage[i % kWPieces], k, &done[i % kWPieces]);      // map: where `off` is mapped
        else pool.submit(batch, w->fd, true, w->stage[i % kWPieces], k, off + (int64_t)o, &done[i % kWPieces]);
    };
    auto copies = [&]() -> int {
        for (size_t i = 0; i < pieces; ++i) {
            const int slot = (int)(i % kWPieces);
            if (i >= (size_t)kWPieces) IoPool::wait_flag(batch, &done[slot]);  // the write of piece i - kWPieces is through
            const size_t o = i * piece, k = std::min(piece, bytes - o);
            PSS_HIP(hipMemcpyAsync(w->stage[slot], static_cast<const uint8_t *>(src) + o, k, hipMemcpyDeviceToHost, d.io_stream));
            PSS_HIP(hipEventRecord(d.ev[slot], d.io_stream));
            if (i >= 1) {
                PSS_HIP(hipEventSynchronize(d.ev[(i - 1) % kWPieces]));
                put(i - 1);
            }
        }
        if (pieces) {
            PSS_HIP(hipEventSynchronize(d.ev[(pieces - 1) % kWPieces]));
            put(pieces - 1);
        }
        return PSS_OK;
    };
    const int rc = copies();
    const int err = IoPool::wait_all(batch);     // always: the pool's pieces point at `done` and at the staging ring
    if (rc != PSS_OK) return rc;
    if (err) {
        errno = err;
        return io_error("write");
    }
    return PSS_OK;
}

// One chunk record: u32le len | data | u32le 4n | n x i32le  (src/lib.rs:112-119), at w->pos.
int write_record(pss_writer *w, const WJob &job)
{
    uint8_t hdr[8];
    const size_t hl = w->version == 2 ? 8 : 4;
    const size_t n = job.n;
    const size_t sa_bytes = n < 2 ? 4 * n : n * 4;
    const int64_t at = w->pos;
    const bool striped = w->stripes.S() != 0;
    const int64_t total = (int64_t)(2 * hl + n + (striped ? 0 : sa_bytes));
    const uint64_t unit_base = w->stripes.next_unit;
    if (striped) w->stripes.next_unit += (sa_bytes + DeviceCtx::kIoPiece - 1) / DeviceCtx::kIoPiece;
    w->pos += total;                    // whatever happens below, no later record may land here
    errno = 0;
    // On tmpfs large records go into the file through a shared MAPPING of their range: the blocks are reserved first
    // (fallocate: a full file system is reported here, not as a SIGBUS later), then the threads of the pool copy into the
    // mapping and their page faults allocate the pages in parallel.  Elsewhere (and where fallocate or mmap is refused)
    // the pieces are pwritten: write(2) holds the inode's lock exclusively, so the threads take turns at 11 - 14 GB/s on
    // this box whatever their number -- the ceiling of ONE index file in the page cache (profiles/r04_pagecache_micro.txt).
    uint8_t *map = nullptr, *map_base = nullptr;
    size_t map_len = 0;
#ifdef __linux__
    if (!striped && (size_t)total >= w->mmap_min && !w->no_mmap && w->map_fd >= 0 && fallocate(w->fd, 0, (off_t)at, (off_t)total) == 0) {
        const int64_t pg = (int64_t)sysconf(_SC_PAGESIZE);
        const int64_t lo = at & ~(pg - 1);
        map_len = (size_t)(at + total - lo);
        void *m = mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_SHARED, w->map_fd, (off_t)lo);
        if (m != MAP_FAILED) {
            map_base = static_cast<uint8_t *>(m);
            map = map_base + (at - lo);
        }
    }
    errno = 0;
#endif
    struct Unmap {
        uint8_t *p;
        size_t len;
        ~Unmap() { if (p) (void)munmap(p, len); }
    } unmap{map_base, map_len};
    IoPool::Batch batch;
    IoPool &pool = IoPool::get();
    if (map) ++w->records_mapped;
    else ++w->records_pwritten;
    if (w->version == 2) put_u64le(hdr, (uint64_t)n);
    else put_u32le(hdr, (uint32_t)n);
    if (map) memcpy(map, hdr, hl);
    else PSS_TRY(pwrite_all(w->fd, hdr, hl, at));
    {
        Phase ph("record: text -> pool");
        const size_t piece = DeviceCtx::kIoPiece;
        for (size_t o = 0; o < n; o += piece) {
            if (map) pool.submit_copy(&batch, map + hl + o, job.text + o, std::min(piece, n - o));
            else pool.submit(&batch, w->fd, true, job.text + o, std::min(piece, n - o), at + (int64_t)hl + (int64_t)o);
        }
    }
    if (w->version == 2) put_u64le(hdr, (uint64_t)n * 4);
    else put_u32le(hdr, (uint32_t)(n * 4));   // wraps like `as u32` at n >= 2^30 (lib.rs:116)
    int rc = PSS_OK;
    if (map) memcpy(map + hl + n, hdr, hl);
    else rc = pwrite_all(w->fd, hdr, hl, at + (int64_t)hl + (int64_t)n);
    const int64_t sa_at = at + (int64_t)(2 * hl + n);
    if (rc == PSS_OK && n == 1) {              // libsais.c:6603-6607: n == 1 -> SA[0] = 0, no device involved
        const uint8_t zero[4] = {0, 0, 0, 0};
        if (striped) {
            const int S = w->stripes.S();
            rc = pwrite_all(w->stripes.fd[(size_t)(unit_base % (uint64_t)S)], zero, 4, (int64_t)((unit_base / (uint64_t)S) * DeviceCtx::kIoPiece));
        } else rc = pwrite_all(w->fd, zero, 4, sa_at);
    }
    if (rc == PSS_OK && n >= 2) {
        // x86-64 / little-endian host: int32 in memory == i32le on disk (lib.rs:117-119)
        Phase ph("record: SA -> file");
        const size_t G = w->devs.size();
        WDevice &d = w->devs[job.seq % G];
        rc = guarded([&]() -> int {
            PSS_HIP(hipSetDevice(d.device));
            return download_to_file(w, d, d.sa[(job.seq / G) & 1].p, n * 4, sa_at, &batch, map ? map + 2 * hl + n : nullptr, unit_base);
        });
    }
    const int err = IoPool::wait_all(&batch);      // the text pieces (and, after a failure above, whatever was in flight)
    if (rc == PSS_OK && err) {
        errno = err;
        rc = io_error("write");
    }
    return rc;
}

void w_fail(pss_writer *w, int rc)      // with w->mu held
{
    if (w->rc == PSS_OK && rc != PSS_OK) {
        w->rc = rc;
        w->err_no = errno;
        w->err = last_error();
    }
}

// Builder of device slot `di`: takes the jobs with seq % G == di in order.
void builder_main(pss_writer *w, size_t di)
{
    const size_t G = w->devs.size();
    WDevice &d = w->devs[di];
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
        WJob *job = nullptr;
        w->cv.wait(lk, [&] {
            for (auto &j : w->jobs)
                if (j.seq % G == di && j.state == WJob::QUEUED) {
                    job = &j;
                    return true;
                }
            return w->stop;
        });
        if (!job) return;
        job->state = WJob::BUILDING;
        const uint64_t seq = job->seq;
        const uint8_t *text = job->text;
        const size_t n = job->n;
        const bool skip = w->rc != PSS_OK || n < 2;
        lk.unlock();
        int rc = PSS_OK;
        if (!skip) {
            rc = guarded([&]() -> int {
                DeviceCtx *ctx;
                PSS_TRY(get_build_ctx(d.device, &ctx));
                std::lock_guard<std::recursive_mutex> dl(ctx->mu);     // the builder's workspace is shared by every Writer on the device
                PSS_HIP(hipSetDevice(d.device));
                DevBuf &sa = d.sa[(seq / G) & 1];
                PSS_TRY(ctx->slot[W_TEXT].reserve(n + 64));
                PSS_TRY(sa.reserve(n * 4));
                Phase ph("build: upload+build");
                PSS_HIP(hipMemcpyAsync(ctx->slot[W_TEXT].p, text, n, hipMemcpyHostToDevice, ctx->stream));
                return sa_build_device(ctx, ctx->slot[W_TEXT].p, sa.p, (int32_t)n, 0, nullptr);
            });
        }
        lk.lock();
        w_fail(w, rc);
        for (auto &j : w->jobs)          // the deque may have shifted (front jobs written meanwhile)
            if (j.seq == seq) j.state = WJob::BUILT;
        w->cv.notify_all();
    }
}

void record_main(pss_writer *w)
{
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
        w->cv.wait(lk, [&] { return (!w->jobs.empty() && w->jobs.front().state == WJob::BUILT) || w->stop; });
        if (w->jobs.empty() || w->jobs.front().state != WJob::BUILT) {
            if (w->stop) return;
            continue;
        }
        const WJob job = w->jobs.front();
        const bool skip = w->rc != PSS_OK;       // after a failure nothing more is written: no record follows a broken one
        lk.unlock();
        int rc = PSS_OK;
        if (!skip) rc = guarded([&]() -> int { return write_record(w, job); });
        lk.lock();
        w_fail(w, rc);
        w->inflight_text -= job.text_alloc;
        if (w->free_text.size() <= w->devs.size()) w->free_text.emplace_back(job.text, job.text_alloc);
        else free(job.text);
        w->jobs.pop_front();
        w->written += 1;
        w->cv.notify_all();
    }
}

int w_report(pss_writer *w)              // with w->mu held: the sticky failure, if any
{
    if (w->rc == PSS_OK) return PSS_OK;
    set_error("%s", w->err.c_str());
    errno = w->err_no;
    return w->rc;
}

// Blocks until every queued record is in the file; reports the first failure (every time).
int io_wait(pss_writer *w)
{
    Phase ph("writer: wait for records");
    std::unique_lock<std::mutex> lk(w->mu);
    w->cv.wait(lk, [&] { return w->written == w->next_seq; });
    return w_report(w);
}

int pipe_start(pss_writer *w)
{
    if (w->started) return PSS_OK;
    Phase ph("writer: pipeline start");
    for (int i = 0; i < kWPieces; ++i)
        if (!w->stage[i] && !(w->stage[i] = stage_cache_take()))
            PSS_HIP(hipHostMalloc(&w->stage[i], DeviceCtx::kIoPiece, hipHostMallocPortable));
    for (auto &d : w->devs) {
        PSS_HIP(hipSetDevice(d.device));
        if (!d.io_stream) PSS_HIP(hipStreamCreateWithFlags(&d.io_stream, hipStreamNonBlocking));
        for (int i = 0; i < kWPieces; ++i)
            if (!d.ev[i]) PSS_HIP(hipEventCreateWithFlags(&d.ev[i], hipEventDisableTiming));
    }
    // Lanes on different ordinals must really be different devices: their contexts (workspace, streams) and their
    // copy streams may not coincide, or two builders would scribble over one workspace / serialise on one stream.
    for (size_t i = 0; i < w->devs.size(); ++i) {
        DeviceCtx *ci = nullptr;
        PSS_TRY(get_build_ctx(w->devs[i].device, &ci));
        if (ci->device != w->devs[i].device) {
            set_error("writer lane %zu: context of device %d answers for device %d", i, w->devs[i].device, ci->device);
            return PSS_EDEVICE;
        }
        for (size_t j = 0; j < i; ++j) {
            if (w->devs[j].device == w->devs[i].device) continue;
            DeviceCtx *cj = nullptr;
            PSS_TRY(get_build_ctx(w->devs[j].device, &cj));
            if (ci == cj || ci->stream == cj->stream || w->devs[i].io_stream == w->devs[j].io_stream) {
                set_error("writer lanes %zu and %zu (devices %d, %d) share a context or a stream", j, i, w->devs[j].device,
                          w->devs[i].device);
                return PSS_EDEVICE;
            }
        }
    }
    w->started = true;
    for (size_t di = 0; di < w->devs.size(); ++di) w->devs[di].builder = std::thread(builder_main, w, di);
    w->record_thread = std::thread(record_main, w);
    return PSS_OK;
}

void pipe_stop(pss_writer *w)
{
    if (w->started) {
        {
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [&] { return w->written == w->next_seq; });
            w->stop = true;
            w->cv.notify_all();
        }
        for (auto &d : w->devs)
            if (d.builder.joinable()) d.builder.join();
        if (w->record_thread.joinable()) w->record_thread.join();
    }
    for (auto &d : w->devs) {
        bool touched = d.io_stream || d.sa[0].p || d.sa[1].p;
        if (touched) (void)hipSetDevice(d.device);
        for (int i = 0; i < kWPieces; ++i)
            if (d.ev[i]) (void)hipEventDestroy(d.ev[i]);
        if (d.io_stream) (void)hipStreamDestroy(d.io_stream);
        for (auto &b : d.sa) b.release();
    }
    for (int i = 0; i < kWPieces; ++i)
        if (w->stage[i]) {
            stage_cache_give(w->stage[i]);
            w->stage[i] = nullptr;
        }
    for (auto &t : w->free_text) text_cache_give(t.first, t.second);
    w->free_text.clear();
}

// src/lib.rs:105-124
int w_dump(pss_writer *w)
{
    if (w->len == 0) return PSS_OK;
    if (w->len >= ((size_t)1 << 31)) {
        set_error("chunk of %zu bytes exceeds the 32-bit suffix array", w->len);
        return PSS_EINVAL;
    }
    if (w->version == 1 && w->len >= ((size_t)1 << 30)) {
        // the reference writes (4 n) as u32 here and wraps (src/lib.rs:116): a file its own Reader cannot walk.  Refused.
        set_error("chunk of %zu bytes: the reference container stores the suffix array's byte length in a u32 (src/lib.rs:116), "
                  "chunks must stay below 2^30 bytes -- format_version 2 holds larger ones", w->len);
        return PSS_EINVAL;
    }
    const size_t G = w->devs.size();
    if (w->len >= 2) {
        // no usable device is reported here and now, not by a later call
        DeviceCtx *ctx;
        PSS_TRY(get_build_ctx(w->devs[w->next_seq % G].device, &ctx));
    }
    if (w->len >= 2 || w->started) PSS_TRY(pipe_start(w));
    if (!w->started) {
        // a one-byte chunk before anything touched a device: written in place (libsais.c:6603-6607)
        WJob job;
        job.text = w->buf;
        job.n = w->len;
        PSS_TRY(write_record(w, job));
        w->len = 0;
        return PSS_OK;
    }
    std::unique_lock<std::mutex> lk(w->mu);
    // the SA buffer this chunk builds into was last used by chunk k - 2 G: its record must be out
    // ... and the host text of the chunks in flight stays inside the budget (chunks of 2 GiB on eight devices would
    // otherwise park 16 x 2 GiB of text that the builders have long uploaded); one job always goes through
    w->cv.wait(lk, [&] {
        return w->next_seq - w->written < 2 * G && (w->jobs.empty() || w->inflight_text + w->alloc <= w->text_budget);
    });
    PSS_TRY(w_report(w));
    WJob job;
    job.seq = w->next_seq++;
    job.text = w->buf;
    job.text_alloc = w->alloc;
    w->inflight_text += w->alloc;
    job.n = w->len;
    job.state = WJob::QUEUED;
    w->jobs.push_back(job);
    // go on filling a buffer that a written job gave back (or a fresh one, allocated on demand)
    w->buf = nullptr;
    w->alloc = 0;
    if (!w->free_text.empty()) {
        w->buf = w->free_text.back().first;
        w->alloc = w->free_text.back().second;
        w->free_text.pop_back();
    }
    w->len = 0;
    w->cv.notify_all();
    return PSS_OK;
}

}  // namespace

extern "C" int pss_writer_open_multi(const char *path, int64_t max_chunk_len, const int32_t *devices, int32_t n_devices,
                                     int32_t format_version, pss_writer **out)
{
    return guarded([&]() -> int {
        const bool striped = (format_version & PSS_FORMAT_STRIPED) != 0;
        format_version &= ~PSS_FORMAT_STRIPED;
        if (!path || !out || (format_version != 1 && format_version != 2) || (striped && format_version != 2) || !devices ||
            n_devices < 1 || n_devices > 64) {
            set_error("pss_writer_open: bad arguments (format_version must be 1 or 2 -- 2 | PSS_FORMAT_STRIPED for the striped layout -- "
                      "and 1..64 devices)");
            return PSS_EINVAL;
        }
        int32_t defaults[64];
        if (n_devices == 1 && devices[0] == -1) {      // the default list (PSS_DEVICES / a launcher's pin / every visible device)
            n_devices = pss_default_devices(defaults, 64);
            if (n_devices < 1) return PSS_EINVAL;      // (a PSS_DEVICES that does not parse: the message is set)
            devices = defaults;
        }
        for (int i = 0; i < n_devices; ++i)
            if (devices[i] < 0) {
                set_error("pss_writer_open: device %d out of range", devices[i]);
                return PSS_EINVAL;
            }
        if (format_version == 2 && max_chunk_len > (int64_t)INT32_MAX) {
            set_error("max_chunk_len %lld: a chunk holds at most 2^31 - 1 bytes (32-bit suffix array)", (long long)max_chunk_len);
            return PSS_EINVAL;
        }
        errno = 0;
        const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);   // File::create truncates, lib.rs:55
        if (fd < 0) return io_error(path);
        int64_t pos = 0;
        Stripes stripes;
        // stripe files an earlier striped Writer left beside this path and this one will not rewrite (it has fewer stripes,
        // or none): a Reader must never find arrays that belong to another index there
        {
            int keep = 0;
            if (striped) {
                keep = 8;
                if (const char *e = knob("PSS_STRIPES")) keep = std::min(64, std::max(1, atoi(e)));
            }
            for (int j = keep; j < 64; ++j) (void)unlink(Stripes::name(path, j).c_str());
        }
        if (striped) {
            int S = 8;
            if (const char *e = knob("PSS_STRIPES")) S = std::min(64, std::max(1, atoi(e)));
            for (int j = 0; j < S; ++j) {
                const int sf = open(Stripes::name(path, j).c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0666);
                if (sf < 0) {
                    const int rc = io_error(Stripes::name(path, j).c_str());
                    stripes.close_all();
                    close(fd);
                    return rc;
                }
                stripes.fd.push_back(sf);
            }
        }
        if (format_version == 2) {
            uint8_t hdr[kHeaderV2] = {};
            memcpy(hdr, kMagicV2, 8);
            if (striped) put_u32le(hdr + 8, kStripedFlag | ((uint32_t)stripes.S() << 8) | ((uint32_t)kStripeUnitLog << 16));
            if (pwrite(fd, hdr, kHeaderV2, 0) != (ssize_t)kHeaderV2) {
                stripes.close_all();
                const int rc = io_error(path);
                close(fd);
                return rc;
            }
            pos = (int64_t)kHeaderV2;
        }
        pss_writer *w = new pss_writer();
        w->fd = fd;
        w->pos = pos;
        w->stripes = stripes;
        // Which way large records go into the page cache is a property of the file system (tests/tools/pagecache_micro.c
        // on the GPU box, 16 threads, one file): tmpfs takes 18.6 GB/s through a shared mapping and 5.8 through pwrite;
        // overlayfs / ext4 take 11 - 14 GB/s through pwrite -- the inode's lock lets one thread copy at a time -- and
        // 2 - 7 through a mapping.  PSS_WRITER_MMAP=0|1 overrides.
        {
            struct statfs sf;
            w->no_mmap = !(fstatfs(fd, &sf) == 0 && (unsigned long)sf.f_type == 0x01021994ul /* TMPFS_MAGIC */);
            if (const char *e = knob("PSS_WRITER_MMAP")) w->no_mmap = atoi(e) == 0;
            if (const char *e = knob("PSS_WRITER_MMAP_MIN")) w->mmap_min = (size_t)strtoull(e, nullptr, 0);
            if (!w->no_mmap) {
                w->map_fd = open(path, O_RDWR | O_CLOEXEC);       // (a file this user may not read: records are pwritten)
                if (w->map_fd < 0) w->no_mmap = true;
                errno = 0;
            }
        }
        w->limit = max_chunk_len < 0 ? (size_t)512 * 1024 * 1024 : (size_t)max_chunk_len;   // lib.rs:57
        w->devs.resize((size_t)n_devices);
        for (int i = 0; i < n_devices; ++i) w->devs[(size_t)i].device = devices[i];
        w->version = format_version;
        if (const char *ev = knob("PSS_WRITER_HOST_BUDGET")) w->text_budget = (size_t)strtoull(ev, nullptr, 0);
        *out = w;
        return PSS_OK;
    });
}

extern "C" int pss_writer_open_format(const char *path, int64_t max_chunk_len, int32_t device, int32_t format_version,
                                      pss_writer **out)
{
    return pss_writer_open_multi(path, max_chunk_len, &device, 1, format_version, out);
}

extern "C" int pss_writer_open(const char *path, int64_t max_chunk_len, int32_t device, pss_writer **out)
{
    return pss_writer_open_format(path, max_chunk_len, device, 1, out);
}

extern "C" int pss_writer_add_entry(pss_writer *w, const uint8_t *text, uint64_t len)
{
    return guarded([&]() -> int {
        if (!w || (!text && len)) return PSS_EINVAL;
        if (len > w->limit) {   // lib.rs:92-94
            set_error("entry is too big");
            return PSS_ETOOBIG;
        }
        if (w->len + len + 1 > w->limit) PSS_TRY(w_dump(w));   // lib.rs:96-98
        return w_append(w, text, (size_t)len);                 // lib.rs:99-100
    });
}

// src/lib.rs:67-86.  Line rule of bstr 0.2 `for_byte_line` (Cargo.toml pins
// bstr = "0.2"; not vendored): a line ends after each '\n'; that '\n' and then
// one preceding '\r' are stripped; a final unterminated line is delivered
// as is; an empty file has no lines.  No size check, no UTF-8 validation.
extern "C" int pss_writer_add_file_lines(pss_writer *w, const char *path)
{
    return guarded([&]() -> int {
        if (!w || !path) return PSS_EINVAL;
        Phase ph_all("add_file_lines total");
        errno = 0;
        const int in = open(path, O_RDONLY | O_CLOEXEC);
        if (in < 0) return io_error(path);
        struct CloseIn {
            int fd;
            ~CloseIn() { close(fd); }
        } close_in{in};
        std::vector<uint8_t> line;                       // carry: the unterminated tail of the previous block
        std::vector<uint8_t> block((size_t)4 << 20), aside;
        int rc = PSS_OK;
        // read(2) until `want` bytes or the end of the file
        auto rd = [&](uint8_t *dst, size_t want, size_t *got) -> int {
            size_t at = 0;
            while (at < want) {
                const ssize_t k = read(in, dst + at, want - at);
                if (k < 0) {
                    if (errno == EINTR) continue;
                    return io_error(path);
                }
                if (k == 0) break;
                at += (size_t)k;
            }
            *got = at;
            return PSS_OK;
        };
        auto deliver = [&](const uint8_t *p, size_t l, bool terminated) -> int {
            if (terminated && l && p[l - 1] == '\r') --l;
            if (w->len + l + 1 > w->limit) PSS_TRY(w_dump(w));   // lib.rs:75-77
            return w_append(w, p, l);
        };
        // Whole '\n'-terminated lines without any '\r' are appended in bulk: the
        // per-line rule "flush when the next line does not fit, then append" is the
        // same as "append the longest run of whole lines that fits, flush, go on".
        auto bulk = [&](const uint8_t *p, size_t size) -> int {
            size_t pos = 0;
            while (pos < size) {
                const size_t room = w->limit > w->len ? w->limit - w->len : 0;
                size_t k = 0;
                if (size - pos <= room) {
                    k = size - pos;
                } else if (room) {
                    const void *q = memrchr(p + pos, '\n', room);
                    if (q) k = (size_t)(static_cast<const uint8_t *>(q) - (p + pos)) + 1;
                }
                if (k) {
                    PSS_TRY(w_reserve(w, k));
                    memcpy(w->buf + w->len, p + pos, k);
                    w->len += k;
                    pos += k;
                } else {   // the next line does not fit: per-line rule (flush, then append, growing if it must)
                    const uint8_t *nl = static_cast<const uint8_t *>(memchr(p + pos, '\n', size - pos));
                    const size_t l = (size_t)(nl - (p + pos));
                    PSS_TRY(deliver(p + pos, l, false));
                    pos += l + 1;
                }
            }
            return PSS_OK;
        };
        // one block of the file, wherever it was read to
        auto process = [&](const uint8_t *blk, size_t got) -> int {
            size_t p = 0;
            if (!line.empty()) {   // finish the carried line first
                const uint8_t *nl = static_cast<const uint8_t *>(memchr(blk, '\n', got));
                const size_t e = nl ? (size_t)(nl - blk) : got;
                line.insert(line.end(), blk, blk + e);
                if (!nl) return PSS_OK;
                PSS_TRY(deliver(line.data(), line.size(), true));
                line.clear();
                p = e + 1;
            }
            const void *last = p < got ? memrchr(blk + p, '\n', got - p) : nullptr;
            const size_t whole_end = last ? (size_t)(static_cast<const uint8_t *>(last) - blk) + 1 : p;
            if (whole_end > p) {
                if (memchr(blk + p, '\r', whole_end - p) == nullptr) {
                    PSS_TRY(bulk(blk + p, whole_end - p));
                } else {
                    while (p < whole_end) {
                        const uint8_t *nl = static_cast<const uint8_t *>(memchr(blk + p, '\n', whole_end - p));
                        const size_t e = (size_t)(nl - blk);
                        PSS_TRY(deliver(blk + p, e - p, true));
                        p = e + 1;
                    }
                }
            }
            line.insert(line.end(), blk + whole_end, blk + got);
            return PSS_OK;
        };
        // `pend`: bytes of an unterminated line that sit IN PLACE at w->buf + w->len (the tail of the last direct read).
        // Round 4 carried that tail over in `line`, and a non-empty `line` sent every later block through the copying
        // path: the direct read engaged once per file.  The next block is now read right behind the tail, which then
        // finishes where it lies.
        size_t pend = 0;
        size_t direct_block = (size_t)32 << 20, direct_min_room = (size_t)1 << 20;     // tests shrink both: PSS_INGEST_BLOCK, _MIN_ROOM
        if (const char *e = knob("PSS_INGEST_BLOCK")) direct_block = std::max<size_t>(16, (size_t)strtoull(e, nullptr, 0));
        if (const char *e = knob("PSS_INGEST_MIN_ROOM")) direct_min_room = std::max<size_t>(1, (size_t)strtoull(e, nullptr, 0));
        auto pend_to_line = [&]() {
            if (pend) line.assign(w->buf + w->len, w->buf + w->len + pend);
            pend = 0;
        };
        for (;;) {
            const size_t room = w->limit > w->len ? w->limit - w->len : 0;
            size_t got = 0;
            if (line.empty() && room >= pend + direct_min_room) {
                // The file is read STRAIGHT into the chunk being filled (round 4: one copy of every byte instead of two).
                // Whatever is read fits the chunk, so its whole lines are exactly what the per-line rule would have
                // appended; the unterminated tail stays where it is and the next read continues it.  A block with a
                // '\r' in it is set aside (with the tail) and goes line by line.
                const size_t want = std::min(room - pend, direct_block);
                rc = w_reserve(w, pend + want);
                if (rc != PSS_OK) break;
                uint8_t *q = w->buf + w->len + pend;
                rc = rd(q, want, &got);
                if (rc != PSS_OK || got == 0) break;
                if (memchr(q, '\r', got) == nullptr) {
                    const void *last = memrchr(q, '\n', got);
                    if (last) {
                        const size_t whole = (size_t)(static_cast<const uint8_t *>(last) - (w->buf + w->len)) + 1;
                        pend = pend + got - whole;
                        w->len += whole;
                        w->ingest_direct += whole;
                    } else {
                        pend += got;               // a line longer than the block: it goes on
                    }
                    continue;
                }
                aside.assign(w->buf + w->len, q + got);      // the tail in place and the block behind it
                pend = 0;
                w->ingest_copied += aside.size();
                rc = process(aside.data(), aside.size());
            } else {
                pend_to_line();
                rc = rd(block.data(), block.size(), &got);
                if (rc != PSS_OK || got == 0) break;
                w->ingest_copied += got;
                rc = process(block.data(), got);
            }
            if (rc != PSS_OK) break;
        }
        pend_to_line();
        if (rc == PSS_OK && !line.empty()) rc = deliver(line.data(), line.size(), false);
        return rc;
    });
}

extern "C" int pss_writer_dump(pss_writer *w)
{
    return guarded([&]() -> int { return w ? w_dump(w) : PSS_EINVAL; });
}

extern "C" int pss_writer_finalize(pss_writer *w)
{
    return guarded([&]() -> int {
        if (!w) return PSS_EINVAL;
        if (w->len) PSS_TRY(w_dump(w));   // lib.rs:129-131
        PSS_TRY(io_wait(w));              // the record in flight reaches the file before the flush
        // (lib.rs:132 flushes the BufWriter: here every record went to the file with pwrite, nothing is buffered)
        return PSS_OK;
    });
}

extern "C" int pss_writer_close(pss_writer *w)
{
    return guarded([&]() -> int {
        if (!w) return PSS_OK;
        int rc = PSS_OK;
        if (w->len) rc = w_dump(w);   // Drop -> finalize, lib.rs:138-144
        const int rc2 = io_wait(w);
        if (rc == PSS_OK) rc = rc2;
        const int e = errno;
        const std::string msg = rc != PSS_OK ? last_error() : std::string();
        const auto tc0 = std::chrono::steady_clock::now();
        pipe_stop(w);
        const auto tc1 = std::chrono::steady_clock::now();
        errno = 0;
        if (w->map_fd >= 0) (void)close(w->map_fd);
        const int serr = w->stripes.close_all();
        int crc = close(w->fd);
        if (crc == 0 && serr != 0) {        // (a stripe file's close failed: reported like the index file's own)
            errno = serr;
            crc = -1;
        }
        if (knob("PSS_TIMING"))
            fprintf(stderr, "[pss] writer close: threads and device buffers %.1f ms, close(fd) %.1f ms\n",
                    std::chrono::duration<double, std::milli>(tc1 - tc0).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc1).count());
        if (crc != 0 && rc == PSS_OK) rc = io_error("close");
        else if (rc != PSS_OK) {
            set_error("%s", msg.c_str());
            errno = e;
        }
        text_cache_give(w->buf, w->alloc);
        delete w;
        return rc;
    });
}

extern "C" uint64_t pss_writer_chunk_limit(const pss_writer *w) { return w ? w->limit : 0; }

extern "C" int pss_writer_io_stats(pss_writer *w, pss_writer_io *out)
{
    if (!w || !out) return PSS_EINVAL;
    out->records_mapped = w->records_mapped.load();
    out->records_pwritten = w->records_pwritten.load();
    out->ingest_direct_bytes = w->ingest_direct;
    out->ingest_copied_bytes = w->ingest_copied;
    return PSS_OK;
}

// ------------------------------------------------------------------- Reader --

struct pss_reader {
    int device = 0;
    DeviceCtx *ctx = nullptr;
    std::vector<ChunkDesc> chunks;      // device pointers of resident chunks
    // Residency of chunk i.  The text always lives in HBM.  The suffix array does too while it fits;
    // past the HBM budget it stays in pinned host memory that the kernels read over PCIe (tier 2:
    // the key-sample table, kept in HBM, confines every query to a few dozen such reads).
    struct Mem {
        void *text = nullptr;
        void *sa = nullptr;        // hipMalloc or (sa_host) hipHostMalloc
        void *skeys = nullptr;     // own hipMalloc when the suffix array is on the host, else inside `sa`
        bool sa_host = false;
        uint64_t hbm_bytes = 0, host_bytes = 0;
    };
    std::vector<Mem> mem;
    ChunkDesc *d_descs = nullptr;
    size_t d_descs_cap = 0;
    bool dirty = true;
    bool low_latency = false;            // single queries through the resident kernel (pss_reader_set_low_latency)
    // entries of one chunk in the reference's order (suffix-array order of their first hit, src/lib.rs:262-276) instead of
    // the order of their leftmost match: pss_reader_set_result_order, PSS_RESULT_ORDER=sa
    bool order_sa = knob("PSS_RESULT_ORDER") != nullptr && strcmp(knob("PSS_RESULT_ORDER"), "sa") == 0;
    // Residency manager ("LRU when index > HBM").  A reader with suffix arrays on the host tier keeps,
    // per chunk, a decayed count of the hits its batches found there and the number of the last batch that touched it;
    // between batches the hottest host-tier suffix array changes places with the coldest one in HBM when it is more than
    // twice as hot (one exchange per batch; PSS_READER_AUTO_RESIDENCY=0: never -- evict / promote stay as overrides).
    std::vector<uint64_t> heat, last_touch, batch_hits;
    std::vector<uint8_t> manual;         // chunks the caller placed by hand (evict / promote): the manager leaves them alone
    uint64_t batch_seq = 0, auto_moves = 0;
    bool auto_residency = knob("PSS_READER_AUTO_RESIDENCY") == nullptr || atoi(knob("PSS_READER_AUTO_RESIDENCY")) != 0;
    pss_search_stats last{};
    // A reader over several devices (pss_reader_open_multi) is a front for one reader per device -- part k holds the
    // chunks c with c % G == k on devices[k] -- each with a worker thread that answers the batch for its chunks; the
    // calling thread takes part 0 itself and merges (reference: rayon fans one search over all chunks inside the
    // process, src/lib.rs:207, 280-284).
    struct Part;
    std::vector<Part *> parts;
    std::mutex multi_mu;                 // one batch at a time through the workers
};

struct pss_reader::Part {
    pss_reader *reader = nullptr;        // plain single-device reader of this part's chunks
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    // mailbox: the caller fills the job and raises `pending`; the worker clears it when `rc` / `res` / `err` are set
    bool pending = false, quit = false;
    const uint8_t *qbytes = nullptr;
    const uint64_t *qoffsets = nullptr;
    uint32_t nq = 0;
    int mode = 0;                        // SEARCH_FULL / SEARCH_COUNTS
    int rc = 0;
    HostResult res;
    std::string err;
};

struct pss_result {
    HostResult r;
};

namespace {

int reader_sync_descs(pss_reader *r);

// HBM the reader may still take for suffix arrays: PSS_READER_HBM_BUDGET (bytes, over all chunks of
// this reader; tests use it to force the host tier), else whatever hipMalloc grants while
// kHbmReserve stays free for the search and build workspaces.
constexpr size_t kHbmReserve = (size_t)2 << 30;

void reader_free_mem(pss_reader::Mem &m)
{
    if (m.text) (void)hipFree(m.text);
    if (m.sa) (void)(m.sa_host ? hipHostFree(m.sa) : hipFree(m.sa));
    if (m.skeys) (void)hipFree(m.skeys);
    m = pss_reader::Mem{};
}

// Text (zero padded) and suffix array of one chunk; the key-sample table (search.h) lives behind
// the suffix array in the same allocation (or on its own in HBM when the suffix array is on the host).
uint64_t *reader_hits_buffer(pss_reader *r);
void reader_note_batch(pss_reader *r);

int reader_alloc_chunk(pss_reader *r, uint32_t n, ChunkDesc *out, pss_reader::Mem *mem)
{
    PSS_HIP(hipSetDevice(r->device));
    pss_reader::Mem m;
    const size_t sa_bytes = round_up((size_t)n * 4 + 16, 8);
    const bool samples = knob("PSS_NO_KEY_SAMPLES") == nullptr;
    uint32_t shift = kSampleShift;
    if (const char *ev = knob("PSS_SAMPLE_SHIFT")) {      // tests: dense tables on small chunks
        const int v = atoi(ev);
        if (v >= 0 && v <= 20) shift = (uint32_t)v;
    }
    const size_t sk_bytes = samples ? sample_count(n, shift) * 8 : 0;
    hipError_t e = hipMalloc(&m.text, (size_t)n + 128);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc of the text of a %u-byte chunk failed: %s", n, hipGetErrorString(e));
        return PSS_ENOMEM;
    }
    m.hbm_bytes = (size_t)n + 128;
    // tier 1: suffix array (+ samples) in HBM
    uint64_t used = 0;
    for (const auto &x : r->mem) used += x.hbm_bytes;
    bool want_hbm = true;
    if (const char *ev = knob("PSS_READER_HBM_BUDGET"))
        want_hbm = used + m.hbm_bytes + sa_bytes + sk_bytes <= strtoull(ev, nullptr, 0);
    // (second attempt: the grow-only workspace of the builder on this device -- up to 80 bytes per byte of the largest
    // chunk it has built, the sample sort's element buffers alone 32 -- goes back before a suffix array settles for the
    // host tier; the next build allocates what it needs again)
    for (int attempt = 0; want_hbm && attempt < 2 && !m.sa; ++attempt) {
        if (attempt == 1) {
            {
                std::lock_guard<std::recursive_mutex> lk(r->ctx->mu);
                r->ctx->stop_resident();
                for (auto &sl : r->ctx->slot) sl.release();
                r->ctx->small_hdr_ready = nullptr;
            }
            DeviceCtx *bctx = nullptr;                 // (lock order: reader side, then builder side -- nothing takes them the other way round)
            if (get_build_ctx(r->device, &bctx) == PSS_OK) {
                std::lock_guard<std::recursive_mutex> lk(bctx->mu);
                for (auto &sl : bctx->slot) sl.release();
                if (bctx->helper)
                    for (auto &sl : bctx->helper->slot) sl.release();
            }
        }
        e = hipMalloc(&m.sa, sa_bytes + sk_bytes);
        if (e == hipSuccess) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < kHbmReserve && !knob("PSS_READER_HBM_BUDGET")) {
                (void)hipFree(m.sa);                  // it fits, but would starve the workspaces
                m.sa = nullptr;
            }
        } else {
            (void)hipGetLastError();
            m.sa = nullptr;
        }
    }
    if (m.sa) {
        m.hbm_bytes += sa_bytes + sk_bytes;
        out->skeys = samples ? reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(m.sa) + sa_bytes) : nullptr;
    } else {
        // tier 2: suffix array in pinned host memory, samples in HBM
        e = hipHostMalloc(&m.sa, sa_bytes, hipHostMallocPortable);
        if (e == hipSuccess && sk_bytes) e = hipMalloc(&m.skeys, sk_bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            m.sa_host = m.sa != nullptr;
            reader_free_mem(m);
            set_error("no room for the suffix array of a %u-byte chunk in HBM or pinned host memory: %s", n,
                      hipGetErrorString(e));
            return PSS_ENOMEM;
        }
        m.sa_host = true;
        m.host_bytes = sa_bytes;
        m.hbm_bytes += sk_bytes;
        out->skeys = static_cast<uint64_t *>(m.skeys);
    }
    PSS_HIP(hipMemsetAsync(static_cast<uint8_t *>(m.text) + n, 0, 128, r->ctx->stream));
    out->text = static_cast<uint8_t *>(m.text);
    if (m.sa_host) {
        void *dp = nullptr;
        PSS_HIP(hipHostGetDevicePointer(&dp, m.sa, 0));
        out->sa = static_cast<uint32_t *>(dp);
    } else {
        out->sa = static_cast<uint32_t *>(m.sa);
    }
    out->n = n;
    out->shift = shift;
    *mem = m;
    return PSS_OK;
}

// (Re)builds the key samples of a chunk whose text and suffix array are in place (stream-ordered).
int reader_sample_chunk(pss_reader *r, const ChunkDesc &c)
{
    if (!c.skeys) return PSS_OK;
    return build_key_samples(r->ctx, c.text, c.sa, c.n, c.shift, const_cast<uint64_t *>(c.skeys));
}

void reader_free(pss_reader *r);

void part_run(pss_reader::Part *p)      // the job in p's mailbox, on p's reader
{
    pss_reader *r = p->reader;
    std::lock_guard<std::recursive_mutex> lk(r->ctx->mu);
    p->res.release();
    p->err.clear();
    int rc = PSS_OK;
    if (hipSetDevice(r->device) != hipSuccess) {
        set_error("hipSetDevice(%d) failed", r->device);
        rc = PSS_EDEVICE;
    }
    if (rc == PSS_OK) rc = reader_sync_descs(r);
    if (rc == PSS_OK) {
        uint64_t *hits = reader_hits_buffer(r);
        rc = search_batch_device(r->ctx, r->d_descs, (uint32_t)r->chunks.size(), p->qbytes, p->qoffsets, p->nq, &p->res, &r->last,
                                 (SearchMode)p->mode, false, hits, r->order_sa);
        if (rc == PSS_OK && hits) reader_note_batch(r);
    }
    if (rc != PSS_OK) p->err = last_error();
    p->rc = rc;
}

void part_worker(pss_reader::Part *p)
{
    std::unique_lock<std::mutex> lk(p->mu);
    for (;;) {
        p->cv.wait(lk, [&] { return p->pending || p->quit; });
        if (p->quit) return;
        part_run(p);
        p->pending = false;
        p->cv.notify_all();
    }
}

void reader_free(pss_reader *r)
{
    if (!r) return;
    for (pss_reader::Part *p : r->parts) {
        if (p->worker.joinable()) {
            {
                std::lock_guard<std::mutex> lk(p->mu);
                p->quit = true;
            }
            p->cv.notify_all();
            p->worker.join();
        }
        p->res.release();
        reader_free(p->reader);
        delete p;
    }
    r->parts.clear();
    if (r->ctx) (void)hipSetDevice(r->device);
    if (r->ctx) {
        std::lock_guard<std::recursive_mutex> lk(r->ctx->mu);
        if (r->ctx->resident.running && r->ctx->resident.chunks == r->d_descs) r->ctx->stop_resident();   // it reads r's chunk table
    }
    for (auto &m : r->mem) reader_free_mem(m);
    if (r->d_descs) (void)hipFree(r->d_descs);
    delete r;
}

// Reads `bytes` from fp's current position into host memory with the I/O pool (pieces of 16 MiB, several threads).
// (stripes: the bytes are units unit_base, unit_base + 1, .. of the striped layout's files instead of fp's next bytes)
int read_file_parallel(FILE *fp, void *dst, size_t bytes, const Stripes *stripes = nullptr, uint64_t unit_base = 0)
{
    const int fd = fileno(fp);
    const int64_t base = (int64_t)ftello(fp);
    const size_t piece = DeviceCtx::kIoPiece;
    IoPool::Batch batch;
    for (size_t o = 0; o < bytes; o += piece) {
        if (stripes) {
            const uint64_t u = unit_base + o / piece;
            const uint64_t S = (uint64_t)stripes->S();
            IoPool::get().submit(&batch, stripes->fd[(size_t)(u % S)], false, static_cast<uint8_t *>(dst) + o, std::min(piece, bytes - o),
                                 (int64_t)((u / S) * piece));
        } else
            IoPool::get().submit(&batch, fd, false, static_cast<uint8_t *>(dst) + o, std::min(piece, bytes - o), base + (int64_t)o);
    }
    const int err = IoPool::wait_all(&batch);
    if (err) {
        set_error("failed to fill whole buffer (truncated index file)");   // UnexpectedEof
        return PSS_EFORMAT;
    }
    if (!stripes && fseeko(fp, (off_t)(base + (int64_t)bytes), SEEK_SET) != 0) return io_error("seek");
    return PSS_OK;
}

// Reads `bytes` from fp's current position into device memory: the threads of the I/O pool pread pieces into a ring of
// pinned buffers (up to kIoPieces reads in flight), the copy stream uploads every piece as soon as it has arrived --
// reading, uploading and the page-cache copies of several pieces overlap (round 3: one thread's fread, then the copy).
int upload_from_file(pss_reader *r, FILE *fp, void *dst, size_t bytes, const Stripes *stripes = nullptr, uint64_t unit_base = 0)
{
    DeviceCtx *ctx = r->ctx;
    PSS_TRY(ctx->ensure_io_ring());
    const int fd = fileno(fp);
    const int64_t base = (int64_t)ftello(fp);
    const size_t piece = DeviceCtx::kIoPiece;
    constexpr int S = DeviceCtx::kIoPieces;
    const size_t pieces = (bytes + piece - 1) / piece;
    IoPool::Batch batch;
    IoPool &pool = IoPool::get();
    std::atomic<int> done[S];
    for (auto &x : done) x.store(1);
    size_t next = 0;
    bool short_read = false;
    auto body = [&]() -> int {
        for (size_t i = 0; i < pieces; ++i) {
            while (next < pieces && next < i + (size_t)S) {
                const int slot = (int)(next % S);
                if (next >= (size_t)S) PSS_HIP(hipEventSynchronize(ctx->io_ev[slot]));   // the upload of piece next - S is through
                const size_t o = next * piece;
                if (stripes) {
                    const uint64_t u = unit_base + next, SS = (uint64_t)stripes->S();
                    pool.submit(&batch, stripes->fd[(size_t)(u % SS)], false, ctx->io_ring[slot], std::min(piece, bytes - o),
                                (int64_t)((u / SS) * piece), &done[slot]);
                } else
                    pool.submit(&batch, fd, false, ctx->io_ring[slot], std::min(piece, bytes - o), base + (int64_t)o, &done[slot]);
                ++next;
            }
            const int slot = (int)(i % S);
            IoPool::wait_flag(&batch, &done[slot]);
            {
                std::lock_guard<std::mutex> lk(batch.mu);
                if (batch.err) { short_read = true; return PSS_OK; }
            }
            const size_t o = i * piece, k = std::min(piece, bytes - o);
            PSS_HIP(hipMemcpyAsync(static_cast<uint8_t *>(dst) + o, ctx->io_ring[slot], k, hipMemcpyHostToDevice, ctx->copy_stream));
            PSS_HIP(hipEventRecord(ctx->io_ev[slot], ctx->copy_stream));
        }
        return PSS_OK;
    };
    const int rc = body();
    const int err = IoPool::wait_all(&batch);          // always: the pool's pieces point at `done` and at the ring
    const hipError_t he = hipStreamSynchronize(ctx->copy_stream);
    if (rc != PSS_OK) return rc;
    if (err || short_read) {
        set_error("failed to fill whole buffer (truncated index file)");   // UnexpectedEof
        return PSS_EFORMAT;
    }
    PSS_HIP(he);
    if (!stripes && fseeko(fp, (off_t)(base + (int64_t)bytes), SEEK_SET) != 0) return io_error("seek");
    return PSS_OK;
}

}  // namespace

namespace {
// Device copy of the chunk descriptor array (re-uploaded whenever chunks change).
int reader_sync_descs(pss_reader *r)
{
    const uint32_t nc = (uint32_t)r->chunks.size();
    if (!r->dirty || nc == 0) return PSS_OK;
    r->ctx->stop_resident();             // (a resident search kernel keeps reading the table it was started with)
    if (r->d_descs_cap < nc) {
        if (r->d_descs) (void)hipFree(r->d_descs);
        r->d_descs = nullptr;
        const size_t cap = nc < 16 ? 16 : (size_t)nc * 2;
        PSS_HIP(hipMalloc(reinterpret_cast<void **>(&r->d_descs), sizeof(ChunkDesc) * cap));
        r->d_descs_cap = cap;
    }
    PSS_HIP(hipMemcpy(r->d_descs, r->chunks.data(), sizeof(ChunkDesc) * nc, hipMemcpyHostToDevice));
    r->dirty = false;
    return PSS_OK;
}
}  // namespace

extern "C" int pss_reader_create(int32_t device, pss_reader **out)
{
    return guarded([&]() -> int {
        if (!out) return PSS_EINVAL;
        DeviceCtx *ctx;
        PSS_TRY(get_ctx(device, &ctx));
        pss_reader *r = new pss_reader();
        r->device = device;
        r->ctx = ctx;
        *out = r;
        return PSS_OK;
    });
}

extern "C" int pss_reader_open(const char *path, int32_t device, int32_t shard_index, int32_t shard_count,
                               pss_reader **out)
{
    return guarded([&]() -> int {
        if (!path || !out || shard_count < 1 || shard_index < 0 || shard_index >= shard_count) {
            set_error("pss_reader_open: bad arguments");
            return PSS_EINVAL;
        }
        if (device == -1) {             // the default list (PSS_DEVICES / a launcher's pin / every visible device)
            int32_t defaults[64];
            const int32_t k = pss_default_devices(defaults, 64);
            if (k < 1) return PSS_EINVAL;               // (a PSS_DEVICES that does not parse: the message is set)
            if (k > 1 && shard_count == 1) return pss_reader_open_multi(path, defaults, k, out);
            device = defaults[0];       // (a shard is one process's share: one device)
        }
        errno = 0;
        FILE *fp = fopen(path, "rb");   // File::open, lib.rs:165 (NotFound -> FileNotFoundError)
        if (!fp) return io_error(path);
        struct Closer {
            FILE *f;
            ~Closer() { fclose(f); }
        } closer{fp};
        if (fseeko(fp, 0, SEEK_END) != 0) return io_error(path);
        const uint64_t flen = (uint64_t)ftello(fp);   // fs::metadata().len(), lib.rs:168-169
        fseeko(fp, 0, SEEK_SET);
        DeviceCtx *ctx;
        PSS_TRY(get_ctx(device, &ctx));
        // The device context (staging buffers, streams) is shared with every other handle on the device: it is
        // held chunk by chunk, around the uploads only, so searches of other readers and Writer builds
        // interleave with a long load instead of waiting for the whole file.
        std::unique_lock<std::recursive_mutex> lk(ctx->mu, std::defer_lock);
        pss_reader *r = new pss_reader();
        r->device = device;
        r->ctx = ctx;
        uint64_t bytes_read = 0;
        int64_t index = 0;
        int rc = PSS_OK;
        // format 2 announces itself (a reference file starts with the u32 length of its first chunk,
        // < 2^30, which these bytes are not): 64-bit lengths, otherwise the same records
        bool v2 = false;
        Stripes stripes;                       // striped layout: the suffix arrays' files
        struct CloseStripes {
            Stripes &s;
            ~CloseStripes() { s.close_all(); }
        } close_stripes{stripes};
        if (flen >= kHeaderV2) {
            uint8_t fh[kHeaderV2];
            if (fread(fh, 1, kHeaderV2, fp) == kHeaderV2 && memcmp(fh, kMagicV2, 8) == 0) {
                v2 = true;
                bytes_read = kHeaderV2;
                const uint32_t fl = (uint32_t)fh[8] | ((uint32_t)fh[9] << 8) | ((uint32_t)fh[10] << 16) | ((uint32_t)fh[11] << 24);
                if (fl & kStripedFlag) {
                    const int S = (int)((fl >> 8) & 0xffu), ul = (int)((fl >> 16) & 0xffu);
                    if (S < 1 || S > 64 || ul != kStripeUnitLog || (fl & ~0x00ffff01u)) {
                        set_error("striped index: unknown header flags %#x", fl);
                        return PSS_EFORMAT;
                    }
                    for (int j = 0; j < S; ++j) {
                        errno = 0;
                        const int sf = open(Stripes::name(path, j).c_str(), O_RDONLY | O_CLOEXEC);
                        if (sf < 0) return io_error(Stripes::name(path, j).c_str());
                        stripes.fd.push_back(sf);
                    }
                } else if (fl) {
                    set_error("index file: unknown header flags %#x", fl);
                    return PSS_EFORMAT;
                }
            } else {
                fseeko(fp, 0, SEEK_SET);
            }
        }
        const bool striped = stripes.S() != 0;
        const size_t hl = v2 ? 8 : 4;
        auto get_len = [&](uint64_t *out_len) -> bool {
            uint8_t hdr[8];
            if (fread(hdr, 1, hl, fp) != hl) return false;
            uint64_t v = 0;
            for (size_t i = 0; i < hl; ++i) v |= (uint64_t)hdr[i] << (8 * i);
            *out_len = v;
            return true;
        };
        const char *kTrunc = "failed to fill whole buffer (truncated index file)";
        while (bytes_read < flen) {   // lib.rs:174
            if (lk.owns_lock()) lk.unlock();
            uint64_t dlen64 = 0, slen = 0;
            if (!get_len(&dlen64)) { set_error("%s", kTrunc); rc = PSS_EFORMAT; break; }
            if (dlen64 > (uint64_t)INT32_MAX || bytes_read + 2 * hl + dlen64 > flen) {
                if (dlen64 > (uint64_t)INT32_MAX && bytes_read + 2 * hl + dlen64 <= flen)
                    set_error("chunk %lld: %llu bytes of text exceed the 32-bit suffix array", (long long)index, (unsigned long long)dlen64);
                else
                    set_error("%s", kTrunc);
                rc = PSS_EFORMAT;
                break;
            }
            const uint32_t dlen = (uint32_t)dlen64;
            const bool mine = (index % shard_count) == shard_index;
            ChunkDesc cd{};
            pss_reader::Mem cm;
            if (mine && dlen) {
                lk.lock();
                rc = reader_alloc_chunk(r, dlen, &cd, &cm);
                if (rc) break;
                r->chunks.push_back(cd);
                r->mem.push_back(cm);
                rc = upload_from_file(r, fp, const_cast<uint8_t *>(cd.text), dlen);
                if (rc) break;
            } else if (fseeko(fp, dlen, SEEK_CUR) != 0) { rc = io_error(path); break; }
            if (!get_len(&slen)) { set_error("%s", kTrunc); rc = PSS_EFORMAT; break; }
            // the reference format stores (4n) as u32, which wraps from 2^30 bytes of text on (lib.rs:116)
            const uint64_t want = v2 ? (uint64_t)dlen * 4 : (uint64_t)(uint32_t)((uint64_t)dlen * 4);
            if (slen != want) {
                set_error("chunk %lld: suffix array of %llu bytes does not match %u bytes of text", (long long)index,
                          (unsigned long long)slen, dlen);
                rc = PSS_EFORMAT;
                break;
            }
            const uint64_t sa_bytes_all = (uint64_t)dlen * 4;
            const uint64_t sa_bytes_file = striped ? 0 : sa_bytes_all;        // (striped: the array is not in this file)
            if (bytes_read + 2 * hl + dlen + sa_bytes_file > flen) { set_error("%s", kTrunc); rc = PSS_EFORMAT; break; }
            const uint64_t unit_base = stripes.next_unit;
            if (striped) stripes.next_unit += (sa_bytes_all + DeviceCtx::kIoPiece - 1) / DeviceCtx::kIoPiece;
            if (mine && dlen) {
                if (cm.sa_host) {      // host tier: the file is read straight into the pinned buffer
                    rc = read_file_parallel(fp, cm.sa, (size_t)sa_bytes_all, striped ? &stripes : nullptr, unit_base);
                } else {
                    rc = upload_from_file(r, fp, cm.sa, (size_t)sa_bytes_all, striped ? &stripes : nullptr, unit_base);
                }
                if (rc) break;
                rc = reader_sample_chunk(r, cd);     // uploads are complete (copy stream synchronised)
                if (rc) break;
            } else if (sa_bytes_file && fseeko(fp, (off_t)sa_bytes_file, SEEK_CUR) != 0) { rc = io_error(path); break; }
            bytes_read += 2 * hl + (uint64_t)dlen + sa_bytes_file;   // lib.rs:184
            ++index;
        }
        if (!lk.owns_lock()) lk.lock();
        if (rc == PSS_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) {
            set_error("key samples: %s", hipGetErrorString(hipGetLastError()));
            rc = PSS_EDEVICE;
        }
        if (rc == PSS_OK) rc = reader_sync_descs(r);
        if (rc != PSS_OK) {
            reader_free(r);
            return rc;
        }
        *out = r;
        return PSS_OK;
    });
}

extern "C" int pss_reader_open_multi(const char *path, const int32_t *devices, int32_t n_devices, pss_reader **out)
{
    return guarded([&]() -> int {
        if (!path || !out || !devices || n_devices < 1 || n_devices > 64) {
            set_error("pss_reader_open_multi: bad arguments");
            return PSS_EINVAL;
        }
        if (n_devices == 1) return pss_reader_open(path, devices[0], 0, 1, out);
        // every part reads the file for its own chunks (seeking over the others'), all parts at once: the uploads of
        // different devices overlap, parts sharing a device take turns on its staging buffers
        const int G = n_devices;
        std::vector<pss_reader *> rd(G, nullptr);
        std::vector<int> rcs(G, PSS_OK);
        std::vector<std::string> errs(G);
        std::vector<int> errnos(G, 0);
        std::vector<std::thread> th;
        for (int k = 0; k < G; ++k)
            th.emplace_back([&, k] {
                rcs[k] = pss_reader_open(path, devices[k], k, G, &rd[k]);
                if (rcs[k] != PSS_OK) {
                    errs[k] = last_error();
                    errnos[k] = errno;
                }
            });
        for (auto &t : th) t.join();
        for (int k = 0; k < G; ++k) {
            if (rcs[k] == PSS_OK) continue;
            set_error("%s", errs[k].c_str());
            const int rc = rcs[k], en = errnos[k];
            for (pss_reader *x : rd) reader_free(x);
            errno = en;       // (PSS_EIO: the binding turns errno into the OSError subclass the reference raises)
            return rc;
        }
        pss_reader *r = new pss_reader();
        r->device = devices[0];
        r->ctx = rd[0]->ctx;
        for (int k = 0; k < G; ++k) {
            pss_reader::Part *p = new pss_reader::Part();
            p->reader = rd[k];
            r->parts.push_back(p);
        }
        for (int k = 1; k < G; ++k) r->parts[k]->worker = std::thread(part_worker, r->parts[k]);      // part 0 runs on the caller
        *out = r;
        return PSS_OK;
    });
}

namespace {

// One batch over the parts of a multi-device reader: every worker answers for its chunks, the caller for part 0;
// then the per-part results are merged query-major, part-major inside a query (pss_merge_packed's order).
int multi_batch(pss_reader *r, const uint8_t *qbytes, const uint64_t *qoffsets, uint32_t nq, int mode, HostResult *out)
{
    std::lock_guard<std::mutex> batch(r->multi_mu);
    const auto t0 = std::chrono::steady_clock::now();
    const size_t G = r->parts.size();
    for (size_t k = 0; k < G; ++k) {
        pss_reader::Part *p = r->parts[k];
        std::lock_guard<std::mutex> lk(p->mu);
        p->qbytes = qbytes;
        p->qoffsets = qoffsets;
        p->nq = nq;
        p->mode = mode;
        if (k) p->pending = true;
    }
    for (size_t k = 1; k < G; ++k) r->parts[k]->cv.notify_all();
    part_run(r->parts[0]);
    int rc = r->parts[0]->rc;
    std::string err = r->parts[0]->err;
    for (size_t k = 1; k < G; ++k) {
        pss_reader::Part *p = r->parts[k];
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv.wait(lk, [&] { return !p->pending; });
        if (p->rc != PSS_OK && rc == PSS_OK) {
            rc = p->rc;
            err = p->err;
        }
    }
    if (rc != PSS_OK) {
        set_error("%s", err.c_str());
        return rc;
    }
    pss_search_stats st{};
    st.queries = nq;
    uint64_t E = 0, B = 0;
    for (pss_reader::Part *p : r->parts) {
        const pss_search_stats &ps = p->reader->last;
        st.hits += ps.hits;
        st.entries += ps.entries;
        st.result_bytes += ps.result_bytes;
        st.ms_device = std::max(st.ms_device, ps.ms_device);
        st.ms_interval = std::max(st.ms_interval, ps.ms_interval);
        E += p->res.n_entries;
        B += p->res.n_bytes;
    }
    out->nq = nq;
    out->qcount = static_cast<uint64_t *>(calloc(nq ? nq : 1, sizeof(uint64_t)));
    if (!out->qcount) return PSS_ENOMEM;
    if (mode == SEARCH_COUNTS) {
        for (pss_reader::Part *p : r->parts)
            for (uint32_t q = 0; q < nq; ++q) out->qcount[q] += p->res.qcount[q];
    } else {
        // The merged result lives where a single-device result would: a block of the pinned pool when it is large (reused
        // from batch to batch -- a fresh malloc of hundreds of megabytes is page faults on every first touch), else malloc.
        PSS_TRY(alloc_host_result(out, E, B, !search_knobs().no_pinned_results));
        // Query-major, part-major inside a query.  Round 6: by several threads -- one pass over the counts finds where
        // every RANGE of queries starts (output entry, output byte, every part's cursor), then the ranges are merged
        // side by side (one thread took 0.2 s for the 14.7 M entries / 0.6 GB of the 15-chunk `lines` batch: six times the
        // search itself; tests/tools/multi_merge_perf.py).
        struct RangeStart {
            uint32_t q0;
            uint64_t e_out, b_out;
            std::vector<uint64_t> cursor;
        };
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const uint32_t want = (E >= (1u << 18) || B >= ((uint64_t)32 << 20)) ? std::min<uint32_t>(16u, std::max(1u, hw / 2)) : 1u;
        const uint32_t R = std::max<uint32_t>(1u, std::min<uint32_t>(want, nq ? nq : 1u));
        std::vector<RangeStart> starts(R);
        {
            std::vector<uint64_t> cursor(G, 0);
            uint64_t e_out = 0, b_out = 0;
            uint32_t next = 0;
            for (uint32_t q = 0; q <= nq; ++q) {
                while (next < R && q == (uint32_t)((uint64_t)nq * next / R)) {
                    starts[next] = RangeStart{q, e_out, b_out, cursor};
                    ++next;
                }
                if (q == nq) break;
                for (size_t k = 0; k < G; ++k) {
                    const HostResult &pr = r->parts[k]->res;
                    const uint64_t c = pr.qcount[q];
                    if (!c) continue;
                    const uint64_t e0 = cursor[k], e1 = e0 + c;
                    e_out += c;
                    b_out += pr.offsets[e1] - pr.offsets[e0];
                    cursor[k] = e1;
                }
            }
            out->offsets[e_out] = b_out;      // (= E, B)
        }
        auto merge_range = [&](uint32_t i) {
            const uint32_t q0 = starts[i].q0, q1 = i + 1 < R ? starts[i + 1].q0 : nq;
            std::vector<uint64_t> cursor = starts[i].cursor;
            uint64_t e_out = starts[i].e_out, b_out = starts[i].b_out;
            for (uint32_t q = q0; q < q1; ++q) {
                for (size_t k = 0; k < G; ++k) {
                    const HostResult &pr = r->parts[k]->res;
                    const uint64_t c = pr.qcount[q];
                    if (!c) continue;
                    const uint64_t e0 = cursor[k], e1 = e0 + c;
                    const uint64_t b0 = pr.offsets[e0], b1 = pr.offsets[e1];
                    for (uint64_t e = e0; e < e1; ++e) out->offsets[e_out++] = b_out + (pr.offsets[e] - b0);
                    memcpy(out->bytes + b_out, pr.bytes + b0, (size_t)(b1 - b0));
                    b_out += b1 - b0;
                    cursor[k] = e1;
                    out->qcount[q] += c;
                }
            }
        };
        if (R == 1) {
            merge_range(0);
        } else {
            std::vector<std::thread> th;
            for (uint32_t i = 1; i < R; ++i) th.emplace_back(merge_range, i);
            merge_range(0);
            for (auto &t : th) t.join();
        }
        out->n_entries = E;
        out->n_bytes = B;
    }
    for (pss_reader::Part *p : r->parts) p->res.release();
    st.ms_host = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    r->last = st;
    return PSS_OK;
}

}  // namespace

extern "C" int pss_reader_add_chunk_device(pss_reader *r, const void *d_text, const void *d_sa, uint32_t n)
{
    return guarded([&]() -> int {
        if (!r || (n && (!d_text || !d_sa))) return PSS_EINVAL;
        if (!r->parts.empty()) {
            set_error("pss_reader_add_chunk_device: not on a multi-device reader");
            return PSS_EINVAL;
        }
        if (n == 0) return PSS_OK;
        std::lock_guard<std::recursive_mutex> lk(r->ctx->mu);
        ChunkDesc cd{};
        pss_reader::Mem cm;
        PSS_TRY(reader_alloc_chunk(r, n, &cd, &cm));
        r->chunks.push_back(cd);
        r->mem.push_back(cm);
        PSS_HIP(hipMemcpyAsync(cm.text, d_text, n, hipMemcpyDeviceToDevice, r->ctx->stream));
        PSS_HIP(hipMemcpyAsync(cm.sa, d_sa, (size_t)n * 4, cm.sa_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                               r->ctx->stream));
        PSS_TRY(reader_sample_chunk(r, cd));
        PSS_HIP(hipStreamSynchronize(r->ctx->stream));
        r->dirty = true;
        return reader_sync_descs(r);
    });
}

extern "C" int pss_reader_set_chunk_device(pss_reader *r, uint64_t index, const void *d_text, const void *d_sa,
                                           uint32_t n)
{
    return guarded([&]() -> int {
        if (!r || !d_text || !d_sa || n == 0 || index > r->chunks.size() || !r->parts.empty()) {
            set_error("pss_reader_set_chunk_device: bad arguments");
            return PSS_EINVAL;
        }
        std::lock_guard<std::recursive_mutex> lk(r->ctx->mu);
        if (index == r->chunks.size()) return pss_reader_add_chunk_device(r, d_text, d_sa, n);
        ChunkDesc &c = r->chunks[index];
        if (c.n != n) {   // different size: fresh allocation
            ChunkDesc fresh{};
            pss_reader::Mem fm;
            reader_free_mem(r->mem[index]);            // first: its HBM may be what the new one needs
            c = ChunkDesc{};                           // (an empty chunk if the allocation below fails)
            r->dirty = true;
            PSS_TRY(reader_alloc_chunk(r, n, &fresh, &fm));
            c = fresh;
            r->mem[index] = fm;
            r->dirty = true;
        }
        const pss_reader::Mem &cm = r->mem[index];
        PSS_HIP(hipMemcpyAsync(cm.text, d_text, n, hipMemcpyDeviceToDevice, r->ctx->stream));
        PSS_HIP(hipMemcpyAsync(cm.sa, d_sa, (size_t)n * 4, cm.sa_host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                               r->ctx->stream));
        PSS_TRY(reader_sample_chunk(r, c));
        PSS_HIP(hipStreamSynchronize(r->ctx->stream));
        return reader_sync_descs(r);
    });
}

namespace {

// Moves the suffix array of resident chunk `index` between the two tiers (HBM <-> pinned host memory the kernels read
// over PCIe); the key samples stay in HBM either way.  `to_host` = evict, else promote.
int reader_move_sa(pss_reader *r, uint64_t index, bool to_host)
{
    if (index >= r->chunks.size()) {
        set_error("chunk %llu of %zu", (unsigned long long)index, r->chunks.size());
        return PSS_EINVAL;
    }
    std::lock_guard<std::recursive_mutex> lk(r->ctx->mu);
    PSS_HIP(hipSetDevice(r->device));
    ChunkDesc &c = r->chunks[index];
    pss_reader::Mem &m = r->mem[index];
    if (m.sa_host == to_host || c.n == 0) return PSS_OK;
    const size_t sa_bytes = round_up((size_t)c.n * 4 + 16, 8);
    const size_t sk_bytes = c.skeys ? sample_count(c.n, c.shift) * 8 : 0;
    hipStream_t s = r->ctx->stream;
    if (to_host) {
        void *host = nullptr, *sk = nullptr;
        hipError_t e = hipHostMalloc(&host, sa_bytes, hipHostMallocPortable);
        if (e == hipSuccess && sk_bytes) e = hipMalloc(&sk, sk_bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (host) (void)hipHostFree(host);
            set_error("evict: no pinned host memory for the suffix array of chunk %llu: %s", (unsigned long long)index, hipGetErrorString(e));
            return PSS_ENOMEM;
        }
        PSS_HIP(hipMemcpyAsync(host, m.sa, (size_t)c.n * 4, hipMemcpyDeviceToHost, s));
        if (sk_bytes) PSS_HIP(hipMemcpyAsync(sk, c.skeys, sk_bytes, hipMemcpyDeviceToDevice, s));
        PSS_HIP(hipStreamSynchronize(s));
        (void)hipFree(m.sa);
        m.sa = host;
        m.skeys = sk;
        m.sa_host = true;
        m.hbm_bytes -= sa_bytes;
        m.host_bytes = sa_bytes;
        void *dp = nullptr;
        PSS_HIP(hipHostGetDevicePointer(&dp, host, 0));
        c.sa = static_cast<uint32_t *>(dp);
        c.skeys = static_cast<uint64_t *>(sk);
    } else {
        void *dev = nullptr;
        const hipError_t e = hipMalloc(&dev, sa_bytes + sk_bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error("promote: no HBM for the suffix array of chunk %llu: %s", (unsigned long long)index, hipGetErrorString(e));
            return PSS_ENOMEM;
        }
        uint64_t *sk = sk_bytes ? reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(dev) + sa_bytes) : nullptr;
        PSS_HIP(hipMemcpyAsync(dev, m.sa, (size_t)c.n * 4, hipMemcpyHostToDevice, s));
        if (sk_bytes) PSS_HIP(hipMemcpyAsync(sk, c.skeys, sk_bytes, hipMemcpyDeviceToDevice, s));
        PSS_HIP(hipStreamSynchronize(s));
        (void)hipHostFree(m.sa);
        if (m.skeys) (void)hipFree(m.skeys);
        m.sa = dev;
        m.skeys = nullptr;
        m.sa_host = false;
        m.hbm_bytes += sa_bytes;
        m.host_bytes = 0;
        c.sa = static_cast<uint32_t *>(dev);
        c.skeys = sk;
    }
    r->dirty = true;
    return reader_sync_descs(r);
}

// ---- residency manager ----------------------------------------------------------------------------------------
bool reader_hbm_room(pss_reader *r, size_t bytes)
{
    if (const char *ev = knob("PSS_READER_HBM_BUDGET")) {
        uint64_t used = 0;
        for (const auto &x : r->mem) used += x.hbm_bytes;
        return used + bytes <= strtoull(ev, nullptr, 0);
    }
    size_t free_b = 0, total_b = 0;
    return hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= bytes + kHbmReserve;
}

uint64_t *reader_hits_buffer(pss_reader *r)
{
    if (!r->auto_residency) return nullptr;
    bool host = false;
    for (size_t c = 0; c < r->mem.size(); ++c) host = host || (r->mem[c].sa_host && !(c < r->manual.size() && r->manual[c]));
    if (!host) return nullptr;          // everything (the manager may move) lives in HBM: nothing to decide, nothing to measure
    r->batch_hits.assign(r->chunks.size(), 0);
    return r->batch_hits.data();
}

// After a batch whose per-chunk hits are in r->batch_hits: decay, then at most one exchange.  Failures to move are not
// failures of the search: the tiers stay as they are.
void reader_note_batch(pss_reader *r)
{
    const size_t nc = r->chunks.size();
    if (r->batch_hits.size() != nc || nc == 0) return;
    r->heat.resize(nc, 0);
    r->last_touch.resize(nc, 0);
    r->batch_seq += 1;
    for (size_t c = 0; c < nc; ++c) {
        r->heat[c] = r->heat[c] / 2 + r->batch_hits[c];
        if (r->batch_hits[c]) r->last_touch[c] = r->batch_seq;
    }
    r->batch_hits.clear();
    size_t hot = nc, cold = nc;
    for (size_t c = 0; c < nc; ++c) {
        if (r->chunks[c].n == 0 || (c < r->manual.size() && r->manual[c])) continue;
        if (r->mem[c].sa_host) {
            if (hot == nc || r->heat[c] > r->heat[hot]) hot = c;
        } else if (cold == nc || r->heat[c] < r->heat[cold] ||
                   (r->heat[c] == r->heat[cold] && r->last_touch[c] < r->last_touch[cold])) {
            cold = c;
        }
    }
    if (hot == nc || r->heat[hot] < 16) return;
    const size_t need = round_up((size_t)r->chunks[hot].n * 4 + 16, 8) +
                        (r->chunks[hot].skeys ? sample_count(r->chunks[hot].n, r->chunks[hot].shift) * 8 : 0);
    const std::string keep = last_error();
    if (reader_hbm_room(r, need)) {
        if (reader_move_sa(r, hot, false) == PSS_OK) r->auto_moves += 1;
    } else if (cold != nc && r->heat[hot] > 2 * r->heat[cold]) {
        if (reader_move_sa(r, cold, true) == PSS_OK) {
            if (reader_move_sa(r, hot, false) == PSS_OK) r->auto_moves += 1;
            else (void)reader_move_sa(r, cold, false);       // no room after all: back as it was
        }
    }
    set_error("%s", keep.c_str());
}

int reader_move_any(pss_reader *r, uint64_t index, bool to_host)
{
    if (!r) return PSS_EINVAL;
    pss_reader *x = r;
    uint64_t at = index;
    if (!r->parts.empty()) {
        const uint64_t G = r->parts.size();  // chunk c of the file lives in part c % G at position c / G
        x = r->parts[index % G]->reader;
        at = index / G;
    }
    PSS_TRY(reader_move_sa(x, at, to_host));
    std::lock_guard<std::recursive_mutex> lk(x->ctx->mu);
    x->manual.resize(x->chunks.size(), 0);
    x->manual[at] = 1;                       // placed by hand: the residency manager leaves it where it is
    return PSS_OK;
}

}  // namespace

extern "C" int pss_reader_evict_chunk(pss_reader *r, uint64_t index)
{
    return guarded([&]() -> int { return reader_move_any(r, index, true); });
}
extern "C" int pss_reader_promote_chunk(pss_reader *r, uint64_t index)
{
    return guarded([&]() -> int { return reader_move_any(r, index, false); });
}

extern "C" int pss_reader_set_auto_residency(pss_reader *r, int32_t on)
{
    if (!r) return PSS_EINVAL;
    r->auto_residency = on != 0;
    r->manual.clear();                       // (switching the manager on again hands every chunk back to it)
    for (pss_reader::Part *p : r->parts) {
        p->reader->auto_residency = on != 0;
        p->reader->manual.clear();
    }
    return PSS_OK;
}

extern "C" int pss_reader_chunk_tiers(const pss_reader *r, uint8_t *tiers, uint64_t cap, uint64_t *auto_moves)
{
    if (!r) return PSS_EINVAL;
    uint64_t moves = r->auto_moves;
    if (r->parts.empty()) {
        for (size_t c = 0; c < r->mem.size() && c < cap; ++c)
            if (tiers) tiers[c] = r->mem[c].sa_host ? 1 : 0;
    } else {
        const uint64_t G = r->parts.size();      // chunk c of the file lives in part c % G at position c / G
        for (uint64_t g = 0; g < G; ++g) {
            const pss_reader *x = r->parts[g]->reader;
            moves += x->auto_moves;
            for (size_t k = 0; k < x->mem.size(); ++k) {
                const uint64_t c = (uint64_t)k * G + g;
                if (tiers && c < cap) tiers[c] = x->mem[k].sa_host ? 1 : 0;
            }
        }
    }
    if (auto_moves) *auto_moves = moves;
    return PSS_OK;
}

extern "C" uint64_t pss_reader_part_chunks(const pss_reader *r, uint64_t *counts, uint64_t cap)
{
    if (!r) return 0;
    if (r->parts.empty()) {
        if (counts && cap) counts[0] = r->chunks.size();
        return 1;
    }
    for (size_t g = 0; g < r->parts.size() && g < cap; ++g)
        if (counts) counts[g] = r->parts[g]->reader->chunks.size();
    return r->parts.size();
}

extern "C" uint64_t pss_reader_num_chunks(const pss_reader *r)
{
    if (!r) return 0;
    uint64_t nc = r->chunks.size();
    for (const pss_reader::Part *p : r->parts) nc += p->reader->chunks.size();
    return nc;
}

extern "C" int pss_reader_residency(const pss_reader *r, uint64_t *hbm_bytes, uint64_t *host_bytes, uint64_t *host_chunks)
{
    if (!r) return PSS_EINVAL;
    uint64_t hb = 0, pb = 0, hc = 0;
    auto add = [&](const pss_reader *x) {
        for (const auto &m : x->mem) {
            hb += m.hbm_bytes;
            pb += m.host_bytes;
            hc += m.sa_host ? 1 : 0;
        }
    };
    add(r);
    for (const pss_reader::Part *p : r->parts) add(p->reader);
    if (hbm_bytes) *hbm_bytes = hb;
    if (host_bytes) *host_bytes = pb;
    if (host_chunks) *host_chunks = hc;
    return PSS_OK;
}

extern "C" int pss_reader_search_batch(pss_reader *r, const uint8_t *qbytes, const uint64_t *qoffsets, uint32_t nq,
                                       pss_result **out)
{
    return guarded([&]() -> int {
        if (!r || !out || (nq && !qoffsets)) {
            set_error("pss_reader_search_batch: bad arguments");
            return PSS_EINVAL;
        }
        if (!r->parts.empty()) {
            pss_result *res = new pss_result();
            const int rc = multi_batch(r, qbytes, qoffsets, nq, SEARCH_FULL, &res->r);
            if (rc != PSS_OK) {
                pss_result_free(res);
                return rc;
            }
            *out = res;
            return PSS_OK;
        }
        std::lock_guard<std::recursive_mutex> lk(r->ctx->mu);
        PSS_HIP(hipSetDevice(r->device));
        const uint32_t nc = (uint32_t)r->chunks.size();
        PSS_TRY(reader_sync_descs(r));
        pss_result *res = new pss_result();
        uint64_t *hits = reader_hits_buffer(r);
        const int rc = search_batch_device(r->ctx, r->d_descs, nc, qbytes, qoffsets, nq, &res->r, &r->last, SEARCH_FULL,
                                           r->low_latency, hits, r->order_sa);
        if (rc == PSS_OK && hits) reader_note_batch(r);
        if (rc != PSS_OK) {
            pss_result_free(res);
            return rc;
        }
        *out = res;
        return PSS_OK;
    });
}

extern "C" int pss_reader_set_low_latency(pss_reader *r, int32_t on)
{
    return guarded([&]() -> int {
        if (!r || !r->parts.empty()) {
            set_error("pss_reader_set_low_latency: a single-device reader is required");
            return PSS_EINVAL;
        }
        std::lock_guard<std::recursive_mutex> lk(r->ctx->mu);
        PSS_HIP(hipSetDevice(r->device));
        r->low_latency = on != 0;
        if (!on) r->ctx->stop_resident();
        return PSS_OK;
    });
}

extern "C" int pss_reader_set_result_order(pss_reader *r, int32_t order)
{
    return guarded([&]() -> int {
        if (!r || (order != PSS_ORDER_TEXT && order != PSS_ORDER_SA)) {
            set_error("pss_reader_set_result_order: bad arguments");
            return PSS_EINVAL;
        }
        r->order_sa = order == PSS_ORDER_SA;
        for (auto *p : r->parts) p->reader->order_sa = r->order_sa;
        return PSS_OK;
    });
}

extern "C" int32_t pss_reader_result_order(const pss_reader *r) { return (r && r->order_sa) ? PSS_ORDER_SA : PSS_ORDER_TEXT; }

extern "C" int pss_reader_low_latency_stats(const pss_reader *r, uint64_t *launches, uint64_t *served)
{
    if (!r || !r->ctx) return PSS_EINVAL;
    std::lock_guard<std::recursive_mutex> lk(r->ctx->mu);
    if (launches) *launches = r->ctx->resident.launches;
    if (served) *served = r->ctx->resident.served;
    return PSS_OK;
}

extern "C" int pss_reader_count_batch(pss_reader *r, const uint8_t *qbytes, const uint64_t *qoffsets, uint32_t nq,
                                      uint64_t *counts)
{
    return guarded([&]() -> int {
        if (!r || (nq && (!qoffsets || !counts))) {
            set_error("pss_reader_count_batch: bad arguments");
            return PSS_EINVAL;
        }
        if (!r->parts.empty()) {
            pss_result res;
            const int rc = multi_batch(r, qbytes, qoffsets, nq, SEARCH_COUNTS, &res.r);
            if (rc == PSS_OK && nq) memcpy(counts, res.r.qcount, (size_t)nq * sizeof(uint64_t));
            res.r.release();
            return rc;
        }
        std::lock_guard<std::recursive_mutex> lk(r->ctx->mu);
        PSS_HIP(hipSetDevice(r->device));
        const uint32_t nc = (uint32_t)r->chunks.size();
        PSS_TRY(reader_sync_descs(r));
        pss_result res;
        uint64_t *hits = reader_hits_buffer(r);
        const int rc = search_batch_device(r->ctx, r->d_descs, nc, qbytes, qoffsets, nq, &res.r, &r->last, SEARCH_COUNTS, false, hits);
        if (rc == PSS_OK && hits) reader_note_batch(r);
        if (rc == PSS_OK && nq) memcpy(counts, res.r.qcount, (size_t)nq * sizeof(uint64_t));
        res.r.release();
        return rc;
    });
}

extern "C" int pss_reader_search_batch_device(pss_reader *r, const uint8_t *qbytes, const uint64_t *qoffsets, uint32_t nq,
                                              pss_device_result *out)
{
    return guarded([&]() -> int {
        if (!r || !out || (nq && !qoffsets)) {
            set_error("pss_reader_search_batch_device: bad arguments");
            return PSS_EINVAL;
        }
        if (!r->parts.empty()) {
            set_error("pss_reader_search_batch_device: a multi-device reader has no single device to leave the result on");
            return PSS_EINVAL;
        }
        std::lock_guard<std::recursive_mutex> lk(r->ctx->mu);
        PSS_HIP(hipSetDevice(r->device));
        const uint32_t nc = (uint32_t)r->chunks.size();
        PSS_TRY(reader_sync_descs(r));
        HostResult hr;
        const int rc = search_batch_device(r->ctx, r->d_descs, nc, qbytes, qoffsets, nq, &hr, &r->last, SEARCH_DEVICE, false, nullptr,
                                           r->order_sa);
        if (rc == PSS_OK) {
            out->num_queries = nq;
            out->num_entries = hr.n_entries;
            out->num_bytes = hr.n_bytes;
            out->d_counts = hr.d_qcount;
            out->d_offsets = hr.d_offsets;
            out->d_bytes = hr.d_bytes;
            out->device = r->device;
        }
        hr.release();
        return rc;
    });
}

// Host merge of per-rank packed results into one, query-major (reference: every chunk task extends one
// Mutex<Vec>, src/lib.rs:280-284; here the ranks' results are concatenated per query, rank-major inside
// a query).  One memcpy per (query, rank) segment -- a rank's entries of one query are contiguous.
extern "C" int pss_merge_packed(uint32_t world, uint64_t nq, const uint64_t *const *counts, const uint64_t *const *offsets,
                                const uint8_t *const *bytes, const uint64_t *num_entries, const uint64_t *num_bytes,
                                uint64_t *out_counts, uint64_t *out_offsets, uint8_t *out_bytes)
{
    return guarded([&]() -> int {
        if (!world || !counts || !offsets || !bytes || !num_entries || !num_bytes || !out_counts || !out_offsets) {
            set_error("pss_merge_packed: bad arguments");
            return PSS_EINVAL;
        }
        std::vector<uint64_t> cursor(world, 0);      // next entry of each rank
        uint64_t e_out = 0, b_out = 0;
        for (uint64_t q = 0; q < nq; ++q) {
            uint64_t tot = 0;
            for (uint32_t r = 0; r < world; ++r) {
                const uint64_t k = counts[r][q];
                if (!k) continue;
                const uint64_t e0 = cursor[r], e1 = e0 + k;
                if (e1 > num_entries[r]) {
                    set_error("pss_merge_packed: rank %u counts exceed its %llu entries", r, (unsigned long long)num_entries[r]);
                    return PSS_EINVAL;
                }
                const uint64_t b0 = offsets[r][e0];
                const uint64_t b1 = e1 < num_entries[r] ? offsets[r][e1] : num_bytes[r];
                if (b0 > b1 || b1 > num_bytes[r]) {      // offsets must grow and stay inside the rank's bytes
                    set_error("pss_merge_packed: rank %u offsets are not monotonic or exceed its %llu bytes", r,
                              (unsigned long long)num_bytes[r]);
                    return PSS_EINVAL;
                }
                for (uint64_t e = e0; e < e1; ++e) {
                    if (offsets[r][e] < b0 || offsets[r][e] > b1) {
                        set_error("pss_merge_packed: rank %u offsets are not monotonic", r);
                        return PSS_EINVAL;
                    }
                    out_offsets[e_out++] = b_out + (offsets[r][e] - b0);
                }
                if (b1 > b0) memcpy(out_bytes + b_out, bytes[r] + b0, (size_t)(b1 - b0));
                b_out += b1 - b0;
                cursor[r] = e1;
                tot += k;
            }
            out_counts[q] = tot;
        }
        out_offsets[e_out] = b_out;
        return PSS_OK;
    });
}

extern "C" int pss_merge_packed_device(int32_t device, uint32_t world, uint64_t nq, const void *const *d_counts,
                                       const void *const *d_starts, const void *const *d_bytes, const uint64_t *num_entries,
                                       const uint64_t *num_bytes, void *d_out_counts, void *d_out_offsets, void *d_out_bytes)
{
    return guarded([&]() -> int {
        if (!world || !d_counts || !d_starts || !d_bytes || !num_entries || !num_bytes || !d_out_counts || !d_out_offsets) {
            set_error("pss_merge_packed_device: bad arguments");
            return PSS_EINVAL;
        }
        DeviceCtx *ctx;
        PSS_TRY(get_ctx(device, &ctx));
        std::lock_guard<std::recursive_mutex> lk(ctx->mu);
        return merge_packed_device(ctx, world, nq, d_counts, d_starts, d_bytes, num_entries, num_bytes, d_out_counts, d_out_offsets,
                                   d_out_bytes);
    });
}

// ---- gather of per-rank packed results over RCCL, inside the C ABI (round 4; hardened in round 5) -----------------
// One process per GPU (north_star: "RCCL over xGMI only to gather / dedupe result strings"): every rank has answered the
// batch for its own chunks and holds a pss_device_result in HBM; the collecting rank receives the others' three buffers
// device to device (one grouped ncclSend / ncclRecv batch, exact sizes, nothing padded), merges them on its GPU
// (merge_packed_device) and brings ONE result down.  No torch: the RCCL entry points are looked up in whatever librccl
// the process has loaded (dlopen: the library is not a link-time dependency of libpss.so) or handed in as a table
// (pss_rccl_inject), the communicator is built from a 128-byte id that the caller ships to every rank by any means
// (a file, MPI, a torch.distributed broadcast) or adopted from the caller (pss_comm_adopt).
//
// What a dead or slow peer may cost (round 5): every wait on the communicator's stream is bounded
// (PSS_RCCL_TIMEOUT_MS, pss_comm_set_timeout_ms; RCCL's own asynchronous error is polled meanwhile); past the bound the
// communicator is ABORTED (ncclCommAbort), the call returns PSS_EDEVICE and so does every later call on that
// communicator -- the process goes on, other communicators and every reader keep working.  A group that was opened is
// always closed (GroupGuard).  The outcome is collective: after the sizes are known every rank contributes a go / no-go
// word, so a collecting rank that cannot reserve its buffers makes every rank return the error instead of leaving the
// others inside ncclSend.  The device context is locked only while buffers of the search workspace are read (the
// rank's own result is first copied into the communicator's buffer) and for the merge -- never while the call waits
// for a peer; the communicator has a stream, an event, a pinned scratch and a device buffer of its own.
namespace {

typedef pss_rccl_unique_id pss_nccl_id;                  // ncclUniqueId
typedef void *pss_nccl_comm;
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(pss_nccl_id *) = nullptr;
    int (*CommInitRank)(pss_nccl_comm *, int, pss_nccl_id, int) = nullptr;
    int (*CommDestroy)(pss_nccl_comm) = nullptr;
    int (*CommAbort)(pss_nccl_comm) = nullptr;                          // optional
    int (*CommGetAsyncError)(pss_nccl_comm, int *) = nullptr;           // optional
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, pss_nccl_comm, void *) = nullptr;
    int (*Recv)(void *, size_t, int, int, pss_nccl_comm, void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, pss_nccl_comm, void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
constexpr int kNcclUint8 = 1, kNcclUint64 = 5;            // ncclUint8, ncclUint64 (rccl.h)

std::mutex g_rccl_mu;
RcclApi g_injected;                                       // pss_rccl_inject
bool g_have_injected = false;

RcclApi &rccl_lookup()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {knob("PSS_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            api.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);               // the copy the process already has (torch's), if any
            if (!api.lib) api.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        auto sym = [&](const char *n) { return dlsym(api.lib, n); };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(sym("ncclCommAbort"));
        api.CommGetAsyncError = reinterpret_cast<decltype(api.CommGetAsyncError)>(sym("ncclCommGetAsyncError"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
        api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Send && api.Recv &&
                 api.AllGather;
    });
    return api;
}

// the table new communicators are made with: the injected one, else the lookup (a communicator keeps a copy of its own)
RcclApi rccl()
{
    {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        if (g_have_injected) return g_injected;
    }
    return rccl_lookup();
}

int rccl_check(const RcclApi &a, int rc, const char *what)
{
    if (rc == 0) return PSS_OK;
    set_error("%s failed: %s", what, a.GetErrorString ? a.GetErrorString(rc) : "RCCL error");
    return PSS_EDEVICE;
}
#define PSS_RCCL(expr) PSS_TRY(rccl_check(a, (expr), #expr))

uint32_t rccl_default_timeout_ms()
{
    if (const char *e = knob("PSS_RCCL_TIMEOUT_MS")) {
        const long v = atol(e);
        if (v > 0) return (uint32_t)std::min<long>(v, 3600 * 1000L);
    }
    return 60000;
}

}  // namespace

struct pss_comm {
    RcclApi api;
    pss_nccl_comm comm = nullptr;
    int32_t world = 0, rank = 0, device = 0;
    bool adopted = false;                // the caller's communicator: aborted on a timeout, never destroyed here
    bool dead = false;                   // aborted: every later call fails
    uint32_t timeout_ms = 60000;
    std::mutex mu;                       // one collective call at a time
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    uint64_t *pinned = nullptr;          // 4 KiB: sizes and go / no-go words
    DevBuf buf;                          // send copies, receive buffers, merge outputs
    uint64_t gathers = 0, aborts = 0;
};

namespace {

// Closes an open group on every path out of the scope (an error between GroupStart and GroupEnd used to leave the
// communicator inside the group).
struct GroupGuard {
    const RcclApi &a;
    bool open = false;
    explicit GroupGuard(const RcclApi &api) : a(api) {}
    int start()
    {
        PSS_TRY(rccl_check(a, a.GroupStart(), "ncclGroupStart"));
        open = true;
        return PSS_OK;
    }
    int end()
    {
        open = false;
        return rccl_check(a, a.GroupEnd(), "ncclGroupEnd");
    }
    ~GroupGuard()
    {
        if (open) (void)a.GroupEnd();
    }
};

void comm_abort(pss_comm *c, const char *why)
{
    if (c->dead) return;
    c->dead = true;
    ++c->aborts;
    if (c->comm) {
        // Without ncclCommAbort in the table the communicator is marked dead and LEFT: ncclCommDestroy waits for outstanding
        // work, and the work of a communicator that is being aborted is exactly what does not finish --
        // a leak in a path that runs once per broken peer, against a call that may never return.
        if (c->api.CommAbort) (void)c->api.CommAbort(c->comm);
        c->comm = nullptr;
    }
    // what the abort releases drains now; a stream that still does not (no ncclCommAbort in this library) is left alone
    const auto t0 = std::chrono::steady_clock::now();
    while (hipStreamQuery(c->stream) == hipErrorNotReady &&
           std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(2000))
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    (void)hipGetLastError();
    set_error("pss_gather_packed_rccl: %s; the communicator was aborted (rank %d of %d)", why, c->rank, c->world);
}

// Waits for everything enqueued on the communicator's stream, for at most its timeout; RCCL's asynchronous error is polled
// on the way.  Never holds a device context.
int comm_wait(pss_comm *c, const char *what)
{
    PSS_HIP(hipEventRecord(c->ev, c->stream));
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(c->timeout_ms);
    for (uint32_t spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(c->ev);
        if (q == hipSuccess) return PSS_OK;
        if (q != hipErrorNotReady) {
            (void)hipGetLastError();
            char why[160];
            snprintf(why, sizeof why, "%s: %s", what, hipGetErrorString(q));
            comm_abort(c, why);
            return PSS_EDEVICE;
        }
        (void)hipGetLastError();
        if (c->api.CommGetAsyncError && c->comm && (spins & 63u) == 63u) {
            int aerr = 0;
            if (c->api.CommGetAsyncError(c->comm, &aerr) == 0 && aerr != 0) {
                char why[200];
                snprintf(why, sizeof why, "%s: asynchronous RCCL error: %s", what,
                         c->api.GetErrorString ? c->api.GetErrorString(aerr) : "?");
                comm_abort(c, why);
                return PSS_EDEVICE;
            }
        }
        if (std::chrono::steady_clock::now() >= deadline) {
            char why[160];
            snprintf(why, sizeof why, "%s: no answer from the peers within %u ms (PSS_RCCL_TIMEOUT_MS)", what, c->timeout_ms);
            comm_abort(c, why);
            return PSS_EDEVICE;
        }
        if (spins < 2000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

int comm_finish_init(pss_comm *c)
{
    PSS_HIP(hipSetDevice(c->device));
    PSS_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    PSS_HIP(hipEventCreateWithFlags(&c->ev, hipEventDisableTiming));
    void *p = nullptr;
    PSS_HIP(hipHostMalloc(&p, 4096, hipHostMallocDefault));
    c->pinned = static_cast<uint64_t *>(p);
    c->timeout_ms = rccl_default_timeout_ms();
    return PSS_OK;
}

void comm_free(pss_comm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm && !c->adopted) (void)c->api.CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->ev) (void)hipEventDestroy(c->ev);
    if (c->pinned) (void)hipHostFree(c->pinned);
    c->buf.release();
    delete c;
}

}  // namespace

extern "C" int pss_rccl_inject(const pss_rccl_api *t)
{
    return guarded([&]() -> int {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        if (!t) {
            g_have_injected = false;
            g_injected = RcclApi{};
            return PSS_OK;
        }
        if (!t->group_start || !t->group_end || !t->send || !t->recv || !t->all_gather) {
            set_error("pss_rccl_inject: group_start, group_end, send, recv and all_gather are required");
            return PSS_EINVAL;
        }
        RcclApi a;
        a.GetUniqueId = t->get_unique_id;
        a.CommInitRank = t->comm_init_rank;
        a.CommDestroy = t->comm_destroy;
        a.CommAbort = t->comm_abort;
        a.CommGetAsyncError = t->comm_get_async_error;
        a.GroupStart = t->group_start;
        a.GroupEnd = t->group_end;
        a.Send = t->send;
        a.Recv = t->recv;
        a.AllGather = t->all_gather;
        a.GetErrorString = t->get_error_string;
        a.ok = true;
        g_injected = a;
        g_have_injected = true;
        return PSS_OK;
    });
}

extern "C" int pss_comm_unique_id(uint8_t *id128)
{
    return guarded([&]() -> int {
        if (!id128) return PSS_EINVAL;
        const RcclApi a = rccl();
        if (!a.ok || !a.GetUniqueId) {
            set_error("no usable librccl in this process (PSS_RCCL_LIB names one)");
            return PSS_EDEVICE;
        }
        pss_nccl_id id;
        PSS_RCCL(a.GetUniqueId(&id));
        memcpy(id128, id.internal, 128);
        return PSS_OK;
    });
}

extern "C" int pss_comm_init(const uint8_t *id128, int32_t world, int32_t rank, int32_t device, pss_comm **out)
{
    return guarded([&]() -> int {
        if (!id128 || !out || world < 1 || world > 16 || rank < 0 || rank >= world) {
            set_error("pss_comm_init: bad arguments (1 .. 16 ranks)");
            return PSS_EINVAL;
        }
        const RcclApi a = rccl();
        if (!a.ok || !a.CommInitRank || !a.CommDestroy) {
            set_error("no usable librccl in this process (PSS_RCCL_LIB names one)");
            return PSS_EDEVICE;
        }
        DeviceCtx *ctx;
        PSS_TRY(get_ctx(device, &ctx));
        PSS_HIP(hipSetDevice(device));
        pss_nccl_id id;
        memcpy(id.internal, id128, 128);
        pss_comm *c = new pss_comm();
        c->api = a;
        c->world = world;
        c->rank = rank;
        c->device = device;
        int rc = rccl_check(a, a.CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");
        if (rc != PSS_OK) c->comm = nullptr;
        if (rc == PSS_OK) rc = comm_finish_init(c);
        if (rc != PSS_OK) {
            comm_free(c);
            return rc;
        }
        *out = c;
        return PSS_OK;
    });
}

extern "C" int pss_comm_adopt(void *nccl_comm, int32_t world, int32_t rank, int32_t device, pss_comm **out)
{
    return guarded([&]() -> int {
        if (!nccl_comm || !out || world < 1 || world > 16 || rank < 0 || rank >= world) {
            set_error("pss_comm_adopt: bad arguments (1 .. 16 ranks)");
            return PSS_EINVAL;
        }
        const RcclApi a = rccl();
        if (!a.ok) {
            set_error("no usable librccl in this process (PSS_RCCL_LIB names one, pss_rccl_inject hands one in)");
            return PSS_EDEVICE;
        }
        DeviceCtx *ctx;
        PSS_TRY(get_ctx(device, &ctx));
        pss_comm *c = new pss_comm();
        c->api = a;
        c->comm = nccl_comm;
        c->adopted = true;
        c->world = world;
        c->rank = rank;
        c->device = device;
        const int rc = comm_finish_init(c);
        if (rc != PSS_OK) {
            comm_free(c);
            return rc;
        }
        *out = c;
        return PSS_OK;
    });
}

extern "C" int pss_comm_set_timeout_ms(pss_comm *c, uint32_t ms)
{
    if (!c || ms == 0) return PSS_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    c->timeout_ms = ms;
    return PSS_OK;
}

extern "C" int pss_comm_status(pss_comm *c, uint64_t *gathers, uint64_t *aborts)
{
    if (!c) return PSS_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (gathers) *gathers = c->gathers;
    if (aborts) *aborts = c->aborts;
    return c->dead ? PSS_EDEVICE : PSS_OK;
}

extern "C" int pss_comm_destroy(pss_comm *c)
{
    return guarded([&]() -> int {
        comm_free(c);
        return PSS_OK;
    });
}

extern "C" int pss_gather_packed_rccl(pss_comm *c, const pss_device_result *mine, int32_t dst, pss_result **out)
{
    return guarded([&]() -> int {
        if (!c || !mine || dst < 0 || dst >= c->world || (c->rank == dst && !out)) {
            set_error("pss_gather_packed_rccl: bad arguments");
            return PSS_EINVAL;
        }
        if (out) *out = nullptr;
        std::lock_guard<std::mutex> call(c->mu);
        if (c->dead) {
            set_error("pss_gather_packed_rccl: the communicator was aborted by an earlier failure");
            return PSS_EDEVICE;
        }
        const RcclApi &a = c->api;
        DeviceCtx *ctx;
        PSS_TRY(get_ctx(c->device, &ctx));
        PSS_HIP(hipSetDevice(c->device));
        hipStream_t s = c->stream;
        const uint32_t W = (uint32_t)c->world;
        const bool collector = (uint32_t)c->rank == (uint32_t)dst;
        const uint64_t nq = mine->num_queries;
        const uint64_t myE = mine->num_entries, myB = mine->num_bytes;
        // 0. my own result out of the search workspace into the communicator's buffer (device to device, under the
        //    context's lock: the workspace belongs to the next search from then on), sizes next to it
        const size_t hdr = 4096 + (size_t)W * 128;
        const size_t m_c = hdr, m_s = m_c + round_up((size_t)nq * 8 + 8, 256), m_b = m_s + round_up((size_t)myE * 8 + 8, 256);
        size_t need = m_b + round_up((size_t)myB + 8, 256);
        uint64_t *h = c->pinned;
        {
            std::lock_guard<std::recursive_mutex> lk(ctx->mu);
            PSS_TRY(c->buf.reserve(need));
            uint8_t *base0 = c->buf.as<uint8_t>();
            if (nq) PSS_HIP(hipMemcpyAsync(base0 + m_c, mine->d_counts, (size_t)nq * 8, hipMemcpyDeviceToDevice, ctx->stream));
            if (myE) PSS_HIP(hipMemcpyAsync(base0 + m_s, mine->d_offsets, (size_t)myE * 8, hipMemcpyDeviceToDevice, ctx->stream));
            if (myB) PSS_HIP(hipMemcpyAsync(base0 + m_b, mine->d_bytes, (size_t)myB, hipMemcpyDeviceToDevice, ctx->stream));
            PSS_HIP(hipStreamSynchronize(ctx->stream));
        }
        // 1. who sends how much: (entries, bytes, queries) of every rank
        uint64_t *d_sz = c->buf.as<uint64_t>();
        h[0] = myE;
        h[1] = myB;
        h[2] = nq;
        h[3] = 0;
        PSS_HIP(hipMemcpyAsync(d_sz, h, 32, hipMemcpyHostToDevice, s));
        PSS_RCCL(a.AllGather(d_sz, d_sz + 8, 4, kNcclUint64, c->comm, s));
        PSS_HIP(hipMemcpyAsync(h + 8, d_sz + 8, (size_t)W * 32, hipMemcpyDeviceToHost, s));
        PSS_TRY(comm_wait(c, "exchange of the result sizes"));
        std::vector<uint64_t> E(W), B(W);
        uint64_t Et = 0, Bt = 0;
        int verdict = PSS_OK;
        for (uint32_t r = 0; r < W; ++r) {
            E[r] = h[8 + 4 * r];
            B[r] = h[8 + 4 * r + 1];
            if (h[8 + 4 * r + 2] != nq && verdict == PSS_OK) {
                set_error("pss_gather_packed_rccl: rank %u answered %llu queries, this rank %llu", r,
                          (unsigned long long)h[8 + 4 * r + 2], (unsigned long long)nq);
                verdict = PSS_EINVAL;          // (every rank sees the same table and reaches the same verdict)
            }
            Et += E[r];
            Bt += B[r];
        }
        // 2. the collecting rank reserves: receive buffers of the exact sizes, then the merge outputs.  Its own copy
        //    (step 0) sits at the front and moves with a reallocation.
        std::vector<size_t> off_c(W), off_s(W), off_b(W);
        size_t o_cnt = 0, o_off = 0, o_byt = 0;
        pss_result *res = nullptr;
        if (collector && verdict == PSS_OK) {
            for (uint32_t r = 0; r < W; ++r) {
                if (r == (uint32_t)dst) continue;
                off_c[r] = need; need += round_up((size_t)nq * 8 + 8, 256);
                off_s[r] = need; need += round_up((size_t)E[r] * 8 + 8, 256);
                off_b[r] = need; need += round_up((size_t)B[r] + 8, 256);
            }
            o_cnt = need; need += round_up((size_t)nq * 8 + 8, 256);
            o_off = need; need += round_up((size_t)(Et + 1) * 8, 256);
            o_byt = need; need += round_up((size_t)Bt + 8, 256);
            if (need > c->buf.cap) {
                // grow-only buffers do not keep their contents: a second one, the front copied over, the first released
                DevBuf bigger;
                verdict = bigger.reserve(need);
                if (verdict == PSS_OK) {
                    if (hipMemcpyAsync(bigger.p, c->buf.p, m_b + round_up((size_t)myB + 8, 256), hipMemcpyDeviceToDevice, s) != hipSuccess ||
                        hipStreamSynchronize(s) != hipSuccess) {
                        (void)hipGetLastError();
                        set_error("pss_gather_packed_rccl: device copy failed");
                        bigger.release();
                        verdict = PSS_EDEVICE;
                    } else {
                        c->buf.release();
                        c->buf = bigger;
                    }
                }
            }
            if (verdict == PSS_OK) {
                res = new pss_result();
                res->r.nq = nq;
                res->r.qcount = static_cast<uint64_t *>(calloc(nq ? nq : 1, 8));
                verdict = res->r.qcount ? alloc_host_result(&res->r, Et, Bt, true) : PSS_ENOMEM;
                if (verdict != PSS_OK) set_error("host allocation of the gathered result failed");
            }
        }
        struct ResGuard {                 // the result is the caller's only when the call succeeds
            pss_result *&r;
            ~ResGuard() { if (r) pss_result_free(r); }
        } res_guard{res};
        // 3. go / no-go, collectively: nobody sends before the collecting rank holds its buffers
        d_sz = c->buf.as<uint64_t>();
        h[0] = (uint64_t)(uint32_t)(-verdict);
        PSS_HIP(hipMemcpyAsync(d_sz, h, 8, hipMemcpyHostToDevice, s));
        PSS_RCCL(a.AllGather(d_sz, d_sz + 8, 1, kNcclUint64, c->comm, s));
        PSS_HIP(hipMemcpyAsync(h + 8, d_sz + 8, (size_t)W * 8, hipMemcpyDeviceToHost, s));
        {
            const std::string mine_err = verdict != PSS_OK ? last_error() : std::string();
            PSS_TRY(comm_wait(c, "go / no-go exchange"));
            if (verdict != PSS_OK) {
                set_error("%s", mine_err.c_str());
                return verdict;
            }
        }
        for (uint32_t r = 0; r < W; ++r)
            if (h[8 + r] != 0) {
                const int theirs = -(int)(uint32_t)h[8 + r];
                set_error("pss_gather_packed_rccl: rank %u gave up before the exchange (status %d)", r, theirs);
                return (theirs == PSS_ENOMEM || theirs == PSS_EINVAL) ? theirs : PSS_EDEVICE;
            }
        uint8_t *base = c->buf.as<uint8_t>();
        if (!collector) {
            // 4a. a contributing rank: its three buffers go to dst as they are
            GroupGuard g(a);
            PSS_TRY(g.start());
            if (nq) PSS_RCCL(a.Send(base + m_c, nq, kNcclUint64, dst, c->comm, s));
            if (myE) PSS_RCCL(a.Send(base + m_s, myE, kNcclUint64, dst, c->comm, s));
            if (myB) PSS_RCCL(a.Send(base + m_b, myB, kNcclUint8, dst, c->comm, s));
            PSS_TRY(g.end());
            PSS_TRY(comm_wait(c, "sending this rank's result"));
            ++c->gathers;
            return PSS_OK;
        }
        // 4b. the collecting rank
        std::vector<const void *> pc(W), ps(W), pb(W);
        {
            GroupGuard g(a);
            PSS_TRY(g.start());
            for (uint32_t r = 0; r < W; ++r) {
                if (r == (uint32_t)dst) {
                    pc[r] = base + m_c;
                    ps[r] = base + m_s;
                    pb[r] = base + m_b;
                    continue;
                }
                pc[r] = base + off_c[r];
                ps[r] = base + off_s[r];
                pb[r] = base + off_b[r];
                if (nq) PSS_RCCL(a.Recv(base + off_c[r], nq, kNcclUint64, (int)r, c->comm, s));
                if (E[r]) PSS_RCCL(a.Recv(base + off_s[r], E[r], kNcclUint64, (int)r, c->comm, s));
                if (B[r]) PSS_RCCL(a.Recv(base + off_b[r], B[r], kNcclUint8, (int)r, c->comm, s));
            }
            PSS_TRY(g.end());
        }
        PSS_TRY(comm_wait(c, "receiving the other ranks' results"));
        // 5. merge on the device (query-major, rank-major inside a query) -- the context's scan workspace and stream, under
        //    its lock -- and one result down, on the communicator's stream again
        {
            std::lock_guard<std::recursive_mutex> lk(ctx->mu);
            PSS_TRY(merge_packed_device(ctx, W, nq, pc.data(), ps.data(), pb.data(), E.data(), B.data(), base + o_cnt, base + o_off,
                                        base + o_byt));
        }
        res->r.n_entries = Et;
        res->r.n_bytes = Bt;
        if (nq) PSS_HIP(hipMemcpyAsync(res->r.qcount, base + o_cnt, (size_t)nq * 8, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipMemcpyAsync(res->r.offsets, base + o_off, (size_t)(Et + 1) * 8, hipMemcpyDeviceToHost, s));
        if (Bt) PSS_HIP(hipMemcpyAsync(res->r.bytes, base + o_byt, Bt, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
        ++c->gathers;
        *out = res;
        res = nullptr;
        return PSS_OK;
    });
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
