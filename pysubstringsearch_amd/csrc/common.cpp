// common.cpp -- error state and per-device contexts.
#include "common.h"

#include <mutex>

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

void trim_all()
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (auto &c : g_ctx) {
        if (c.device < 0) continue;
        std::lock_guard<std::recursive_mutex> lk2(c.mu);
        (void)hipSetDevice(c.device);
        for (auto &s : c.slot) s.release();
    }
}

}  // namespace pss
