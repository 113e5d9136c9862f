// The gather of per-rank packed results over RCCL inside the C ABI (pss_comm_*, pss_gather_packed_rccl, pss_rccl_inject).
// Part of capi.cpp: included there, in this order, into the one translation unit (the pieces share its
// anonymous-namespace helpers); not a header for anybody else.

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
