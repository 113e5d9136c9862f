// common.cpp -- error state and per-device contexts.
#include "common.h"

#include <algorithm>
#include <cerrno>
#include <unistd.h>

#include <mutex>
#include <vector>

namespace pss {

static thread_local std::string g_err;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

const std::string &last_error() { return g_err; }

int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return PSS_OK;
    if (p) {
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    size_t want = round_up(bytes, (size_t)1 << 20);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        (void)hipGetLastError();
        return PSS_ENOMEM;
    }
    cap = want;
    return PSS_OK;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

int DeviceCtx::ensure_staging()
{
    if (stage[0]) return PSS_OK;
    PSS_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        PSS_HIP(hipHostMalloc(&stage[i], kStage, hipHostMallocDefault));
        PSS_HIP(hipEventCreateWithFlags(&stage_ev[i], hipEventDisableTiming));
    }
    return PSS_OK;
}

int DeviceCtx::ensure_io_ring()
{
    PSS_TRY(ensure_staging());
    if (io_ring[0]) return PSS_OK;
    for (int i = 0; i < kIoPieces; ++i) {
        PSS_HIP(hipHostMalloc(&io_ring[i], kIoPiece, hipHostMallocPortable));
        PSS_HIP(hipEventCreateWithFlags(&io_ev[i], hipEventDisableTiming));
    }
    return PSS_OK;
}

IoPool &IoPool::get()
{
    static IoPool pool;
    return pool;
}

IoPool::IoPool()
{
    int k = 0;
    if (const char *e = getenv("PSS_IO_THREADS")) k = atoi(e);
    if (k <= 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        k = (int)std::min<unsigned>(16u, std::max<unsigned>(8u, hw / 8));
    }
    if (k > 64) k = 64;
    for (int i = 0; i < k; ++i) workers_.emplace_back([this] { run(); });
}

IoPool::~IoPool()
{
    {
        std::lock_guard<std::mutex> lk(mu_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto &t : workers_)
        if (t.joinable()) t.join();
}

void IoPool::submit(Batch *b, int fd, bool write, void *buf, size_t len, int64_t off, std::atomic<int> *done)
{
    {
        std::lock_guard<std::mutex> lk(b->mu);
        b->submitted += 1;
    }
    if (done) done->store(0, std::memory_order_relaxed);
    {
        std::lock_guard<std::mutex> lk(mu_);
        q_.push_back(Task{b, fd, write, buf, len, off, done});
    }
    cv_.notify_one();
}

void IoPool::submit_copy(Batch *b, void *dst, const void *src, size_t len, std::atomic<int> *done)
{
    submit(b, -2, true, dst, len, (int64_t)reinterpret_cast<intptr_t>(src), done);
}

void IoPool::run()
{
    for (;;) {
        Task t;
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
            if (q_.empty()) return;
            t = q_.front();
            q_.pop_front();
        }
        int err = 0;
        size_t at = 0;
        uint8_t *p = static_cast<uint8_t *>(t.buf);
        if (t.fd == -2) {
            memcpy(t.buf, reinterpret_cast<const void *>((intptr_t)t.off), t.len);
            at = t.len;
        }
        while (at < t.len) {
            const ssize_t k = t.write ? pwrite(t.fd, p + at, t.len - at, (off_t)(t.off + (int64_t)at))
                                      : pread(t.fd, p + at, t.len - at, (off_t)(t.off + (int64_t)at));
            if (k < 0) {
                if (errno == EINTR) continue;
                err = errno ? errno : EIO;
                break;
            }
            if (k == 0) {               // end of file inside the piece: a truncated index (or a full disk)
                err = t.write ? ENOSPC : ENODATA;
                break;
            }
            at += (size_t)k;
        }
        {
            std::lock_guard<std::mutex> lk(t.b->mu);
            if (err && !t.b->err) t.b->err = err;
            t.b->finished += 1;
            if (t.done) t.done->store(1, std::memory_order_release);
        }
        t.b->cv.notify_all();
    }
}

int IoPool::wait_all(Batch *b)
{
    std::unique_lock<std::mutex> lk(b->mu);
    b->cv.wait(lk, [&] { return b->finished == b->submitted; });
    return b->err;
}

void IoPool::wait_flag(Batch *b, std::atomic<int> *done)
{
    std::unique_lock<std::mutex> lk(b->mu);
    b->cv.wait(lk, [&] { return done->load(std::memory_order_acquire) != 0; });
}

int DeviceCtx::ensure_search_stage()
{
    if (search_stage) return PSS_OK;
    PSS_HIP(hipHostMalloc(&search_stage, kStageQ + kStageR, hipHostMallocDefault));
    return PSS_OK;
}

void SearchKnobs::load()
{
    *this = SearchKnobs{};
    no_small_path = getenv("PSS_NO_SMALL_PATH") != nullptr;
    no_block_path = getenv("PSS_NO_BLOCK_PATH") != nullptr;
    no_search_stage = getenv("PSS_NO_SEARCH_STAGE") != nullptr;
    wave_search = getenv("PSS_WAVE_SEARCH") != nullptr;
    no_group_search = getenv("PSS_NO_GROUP_SEARCH") != nullptr;
    no_mid_pipeline = getenv("PSS_NO_MID_PIPELINE") != nullptr;
    no_pinned_results = getenv("PSS_NO_PINNED_RESULTS") != nullptr;
    small_path_events = getenv("PSS_SEARCH_EVENTS") != nullptr;
    if (const char *e = getenv("PSS_LANE_SEARCH_MIN")) lane_search_min = strtoull(e, nullptr, 0);
    if (const char *e = getenv("PSS_RESIDENT_IDLE_US")) resident_idle_us = (uint32_t)strtoul(e, nullptr, 0);
    if (const char *e = getenv("PSS_RESIDENT_LIFE_US")) resident_life_us = (uint32_t)strtoul(e, nullptr, 0);
}

static SearchKnobs g_knobs;
static std::once_flag g_knobs_once;
const SearchKnobs &search_knobs()
{
    std::call_once(g_knobs_once, []() { g_knobs.load(); });
    return g_knobs;
}
void reload_search_knobs()
{
    (void)search_knobs();
    g_knobs.load();
}

// ---- pinned result blocks -------------------------------------------------------------------
namespace {
struct PinnedBlock {
    void *p;
    size_t bytes;
};
std::mutex g_pool_mu;
std::vector<PinnedBlock> g_pool;            // free blocks
constexpr size_t kPoolMaxBlocks = 4;
constexpr size_t kPoolMaxBytes = (size_t)12 << 30;
}  // namespace

void *pinned_pool_alloc(size_t bytes, size_t *granted)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        int best = -1;
        for (int i = 0; i < (int)g_pool.size(); ++i)
            if (g_pool[i].bytes >= bytes && (best < 0 || g_pool[i].bytes < g_pool[best].bytes)) best = i;
        if (best >= 0 && g_pool[best].bytes <= 2 * bytes + ((size_t)64 << 20)) {
            PinnedBlock b = g_pool[best];
            g_pool.erase(g_pool.begin() + best);
            *granted = b.bytes;
            return b.p;
        }
    }
    void *p = nullptr;
    const size_t want = round_up(bytes + bytes / 8, (size_t)2 << 20);   // headroom: the next batch is rarely the same size
    if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) {      // the pool is shared by every device of the process
        (void)hipGetLastError();
        return nullptr;
    }
    *granted = want;
    return p;
}

void pinned_pool_free(void *p, size_t granted)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t held = 0;
        for (const auto &b : g_pool) held += b.bytes;
        if (g_pool.size() < kPoolMaxBlocks && held + granted <= kPoolMaxBytes) {
            g_pool.push_back({p, granted});
            return;
        }
    }
    (void)hipHostFree(p);
}

void pinned_pool_trim()
{
    std::vector<PinnedBlock> drop;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        drop.swap(g_pool);
    }
    for (auto &b : drop) (void)hipHostFree(b.p);
}

static constexpr int kMaxDevices = 64;
static DeviceCtx g_ctx[kMaxDevices];        // reader side
static DeviceCtx g_bctx[kMaxDevices];       // builder side
static std::mutex g_ctx_mu;

static int init_ctx(DeviceCtx &c, int device)
{
    PSS_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    PSS_HIP(hipGetDeviceProperties(&prop, device));
    c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c.pinned_cap = DeviceCtx::kPinnedBytes;
    PSS_HIP(hipHostMalloc(&c.pinned, c.pinned_cap, hipHostMallocDefault));
    PSS_HIP(hipHostGetDevicePointer(&c.pinned_dev, c.pinned, 0));
    for (hipEvent_t &e : c.search_ev) PSS_HIP(hipEventCreate(&e));
    c.device = device;
    return PSS_OK;
}

static int get_ctx_of(DeviceCtx *table, int device, DeviceCtx **out)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        set_error("no usable HIP device (%s); libpss has no CPU fallback",
                  e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return PSS_EDEVICE;
    }
    if (device < 0 || device >= count || device >= kMaxDevices) {
        set_error("device %d out of range (have %d)", device, count);
        return PSS_EINVAL;
    }
    PSS_HIP(hipSetDevice(device));
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    DeviceCtx &c = table[device];
    if (c.device < 0) PSS_TRY(init_ctx(c, device));
    *out = &c;
    return PSS_OK;
}

int get_helper_ctx(DeviceCtx *parent, DeviceCtx **out)
{
    if (!parent->helper) {
        DeviceCtx *h = new DeviceCtx();
        const int rc = init_ctx(*h, parent->device);
        if (rc != PSS_OK) {
            delete h;
            return rc;
        }
        parent->helper = h;
    }
    *out = parent->helper;
    return PSS_OK;
}

int get_ctx(int device, DeviceCtx **out) { return get_ctx_of(g_ctx, device, out); }
int get_build_ctx(int device, DeviceCtx **out) { return get_ctx_of(g_bctx, device, out); }

int DeviceCtx::ensure_resident()
{
    if (resident.arena) return PSS_OK;
    PSS_HIP(hipStreamCreateWithFlags(&resident.stream, hipStreamNonBlocking));
    // fine-grained: the kernel sees the host's writes, and the host the kernel's, while the kernel runs
    PSS_HIP(hipHostMalloc(&resident.arena, kPinnedBytes, hipHostMallocCoherent | hipHostMallocMapped));
    PSS_HIP(hipHostGetDevicePointer(&resident.arena_dev, resident.arena, 0));
    memset(resident.arena, 0, kPinnedBytes);
    return PSS_OK;
}

void DeviceCtx::Resident::post(const uint8_t *q, uint32_t plen)
{
    ResidentMailbox *mb = reinterpret_cast<ResidentMailbox *>(static_cast<uint8_t *>(arena) + kResidentMailboxOff);
    if (plen != kResidentStop) {
        if (plen <= sizeof mb->post.bytes) {
            memcpy(mb->post.bytes, q, plen);
        } else {
            memcpy(mb->query, q, plen);
            memset(mb->query + plen, 0, 32);
        }
    }
    mb->post.plen = plen;
    ++seq;
    __atomic_store_n(&mb->post.seq_b, seq, __ATOMIC_RELEASE);
    __atomic_store_n(&mb->post.seq_a, seq, __ATOMIC_RELEASE);
}

void DeviceCtx::stop_resident()
{
    if (!resident.running) return;
    resident.post(nullptr, kResidentStop);
    (void)hipStreamSynchronize(resident.stream);
    resident.running = false;
}

// HBM the grow-only workspaces of one device hold right now: the builder's context (and its helper line's), and the
// search scratch of the reader side's context -- resident indexes are not workspace (pss_reader_residency reports those).
uint64_t workspace_bytes(int device)
{
    if (device < 0 || device >= kMaxDevices) return 0;
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    uint64_t total = 0;
    for (DeviceCtx *table : {g_ctx, g_bctx}) {
        DeviceCtx &c = table[device];
        if (c.device < 0) continue;
        for (auto &s : c.slot) total += s.cap;
        if (c.helper)
            for (auto &s : c.helper->slot) total += s.cap;
    }
    return total;
}

void trim_all()
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (DeviceCtx *table : {g_ctx, g_bctx})
        for (int d = 0; d < kMaxDevices; ++d) {
            DeviceCtx &c = table[d];
            if (c.device < 0) continue;
            std::lock_guard<std::recursive_mutex> lk2(c.mu);
            (void)hipSetDevice(c.device);
            c.stop_resident();               // (it works in one of the slots)
            for (auto &s : c.slot) s.release();
            if (c.helper)
                for (auto &s : c.helper->slot) s.release();
            // the fused small-batch path keeps its cursors in one of the slots and only zeroes them when the
            // arena's address changes: a fresh allocation may come back at the old address with garbage in it
            c.small_hdr_ready = nullptr;
        }
    pinned_pool_trim();
}

}  // namespace pss
