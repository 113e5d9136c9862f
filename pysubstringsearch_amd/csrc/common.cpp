// common.cpp -- error state and per-device contexts.
#include "common.h"

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
static DeviceCtx g_ctx[kMaxDevices];
static std::mutex g_ctx_mu;

int get_ctx(int device, DeviceCtx **out)
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
    DeviceCtx &c = g_ctx[device];
    if (c.device < 0) {
        PSS_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        hipDeviceProp_t prop;
        PSS_HIP(hipGetDeviceProperties(&prop, device));
        c.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        c.pinned_cap = DeviceCtx::kPinnedBytes;
        PSS_HIP(hipHostMalloc(&c.pinned, c.pinned_cap, hipHostMallocDefault));
        PSS_HIP(hipHostGetDevicePointer(&c.pinned_dev, c.pinned, 0));
        for (hipEvent_t &e : c.search_ev) PSS_HIP(hipEventCreate(&e));
        c.device = device;
    }
    *out = &c;
    return PSS_OK;
}

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

void trim_all()
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (auto &c : g_ctx) {
        if (c.device < 0) continue;
        std::lock_guard<std::recursive_mutex> lk2(c.mu);
        (void)hipSetDevice(c.device);
        c.stop_resident();               // (it works in one of the slots)
        for (auto &s : c.slot) s.release();
        // the fused small-batch path keeps its cursors in one of the slots and only zeroes them when the
        // arena's address changes: a fresh allocation may come back at the old address with garbage in it
        c.small_hdr_ready = nullptr;
    }
    pinned_pool_trim();
}

}  // namespace pss
