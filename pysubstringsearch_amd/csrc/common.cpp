// common.cpp -- error state and per-device contexts.
#include "common.h"
#include "knobs.h"

#include <atomic>
#include <cstring>
#include <cstdio>
#include <cstdlib>
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
    if (const char *e = knob("PSS_IO_THREADS")) k = atoi(e);
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
            // The notification goes out UNDER the batch's lock: a Batch lives on its waiter's stack (write_record,
            // the reader's load paths), and a waiter that another worker's notification (or a spurious wake-up) lets
            // see finished == submitted returns and ends the Batch.  Notifying after the unlock touched a condition
            // variable in a dead stack frame -- once in some thousands of Writers it found other data there and
            // rewrote it (fuzz campaign of round 6: "stack smashing detected" in the record thread, a hang at exit).
            // With the lock held the waiter cannot leave wait() before this worker is done with the Batch.
            std::lock_guard<std::mutex> lk(t.b->mu);
            if (err && !t.b->err) t.b->err = err;
            t.b->finished += 1;
            if (t.done) t.done->store(1, std::memory_order_release);
            t.b->cv.notify_all();
        }
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
    no_small_path = knob("PSS_NO_SMALL_PATH") != nullptr;
    no_block_path = knob("PSS_NO_BLOCK_PATH") != nullptr;
    no_search_stage = knob("PSS_NO_SEARCH_STAGE") != nullptr;
    wave_search = knob("PSS_WAVE_SEARCH") != nullptr;
    no_group_search = knob("PSS_NO_GROUP_SEARCH") != nullptr;
    no_mid_pipeline = knob("PSS_NO_MID_PIPELINE") != nullptr;
    no_pinned_results = knob("PSS_NO_PINNED_RESULTS") != nullptr;
    small_path_events = knob("PSS_SEARCH_EVENTS") != nullptr;
    if (const char *e = knob("PSS_LANE_SEARCH_MIN")) lane_search_min = strtoull(e, nullptr, 0);
    if (const char *e = knob("PSS_RESIDENT_IDLE_US")) resident_idle_us = (uint32_t)strtoul(e, nullptr, 0);
    if (const char *e = knob("PSS_RESIDENT_LIFE_US")) resident_life_us = (uint32_t)strtoul(e, nullptr, 0);
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
constexpr size_t kPoolMaxBlocks = 24;      // (a reader over eight devices returns eight blocks and a merged one per batch)
// Pinned memory the pool keeps between batches.  Round 6: 12 -> 40 GiB (PSS_PINNED_POOL_BYTES): a batch on natural text
// returns 18 GB of entries (100 000 queries of 16..32 bytes on 15 chunks of `words`, 259 M entries), and a block the pool
// would not keep was pinned and unpinned on EVERY batch -- 1.8 s + 1 s around 0.06 s of kernels and 0.36 s of PCIe.
size_t pool_max_bytes()
{
    static const size_t v = [] {
        const char *e = knob("PSS_PINNED_POOL_BYTES");
        return e ? (size_t)strtoull(e, nullptr, 0) : ((size_t)40 << 30);
    }();
    return v;
}
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
    std::vector<PinnedBlock> drop;
    bool kept = false;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        size_t held = 0;
        for (const auto &b : g_pool) held += b.bytes;
        if (granted <= pool_max_bytes()) {
            // the block just used says most about the next batch: smaller ones make room for it
            while (!g_pool.empty() && (g_pool.size() >= kPoolMaxBlocks || held + granted > pool_max_bytes())) {
                size_t k = 0;
                for (size_t i = 1; i < g_pool.size(); ++i)
                    if (g_pool[i].bytes < g_pool[k].bytes) k = i;
                held -= g_pool[k].bytes;
                drop.push_back(g_pool[k]);
                g_pool.erase(g_pool.begin() + (long)k);
            }
            g_pool.push_back({p, granted});
            kept = true;
        }
    }
    for (auto &b : drop) (void)hipHostFree(b.p);
    if (!kept) (void)hipHostFree(p);
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

// ---- environment switches: the registry (knobs.h) -------------------------------------------------------------------
const KnobDef kKnobs[] = {
    {"PSS_KEY_CHARS", "0 (chosen)", "1|2|3|5|8|12|16", "builder: force the symbols packed into the initial sort key"},
    {"PSS_KEY_DROP", "-1 (chosen)", "0|1|3|7", "builder: force the low bits of the last key symbol that are left out"},
    {"PSS_NO_SAMPLE", "unset", "1", "builder: size the initial key from symbol counts even for large texts (no sizing sample)"},
    {"PSS_NO_TIES_PASS", "unset", "1", "builder: plain 8-byte-key LSD passes and key comparison in the rerank (no tie flags)"},
    {"PSS_MODE", "chosen", "dense|sparse|text", "builder: how ties are resolved -- doubling over an inverse suffix array, sparse (hash), text rounds"},
    {"PSS_TEXT_ROUNDS", "5", "0|1|2|3", "builder: text rounds at most before the rank rounds"},
    {"PSS_MSD", "-1 (screened)", "0|1|1", "initial sort: never / whenever the key fits the hybrid MSD radix sort (msd_sort.hip)"},
    {"PSS_MSD_LSD", "1", "0|1", "MSD sort: 0 = digits in MSD order with a second histogram pass (rounds 2-5); else LSD order, second pass by look-back"},
    {"PSS_MSD_KEY_CAP", "48", "", "MSD sort: largest key width tried (bits; the element format bounds it further)"},
    {"PSS_MSD_PARTIAL_SYMBOL", "unset", "1", "MSD sort: fill the element with the high bits of one more symbol than fits whole (fewer ties; measured slower on lines)"},
    {"PSS_MSD_NO_FUSE", "unset", "1", "MSD sort: ties flagged in the suffix array instead of emitted from the local sort"},
    {"PSS_MSD_SLOW_LOCAL", "unset", "1", "MSD sort: the general (ballot LSD) local-sort kernel for every tile"},
    {"PSS_MSD_SCATTER", "unset", "1", "MSD sort in MSD order: the 8192-element partition kernels of round 2"},
    {"PSS_SS", "-1 (n >= 2^24, MSD declined)", "0|1", "initial sort: never / whenever the text has the size for it the sample sort over 16-byte elements"},
    {"PSS_SS_SEG", "1", "0", "sample sort: 0 = every tile merge-sorted as one array instead of bucket by bucket"},
    {"PSS_SS_WINDOW_PLAN", "unset", "1", "sample sort: tiles by the window rule of the MSD sort instead of the greedy plan"},
    {"PSS_SS_DEBUG", "unset", "", "sample sort: print where a bucket beyond a tile came from (diagnostic)"},
    {"PSS_NO_PLAN_CACHE", "unset", "1", "builder: forget between chunks which initial sort and alphabet the previous chunk took"},
    {"PSS_NO_PLAN_FRONT", "unset", "1", "builder: remember the sort but not the alphabet (separate alphabet and recode passes)"},
    {"PSS_RLE", "-1 (runs average >= 8 bytes)", "0|1|1", "builder: never / always the run-length path (rle_build.hip)"},
    {"PSS_RLE_SORT", "unset", "1", "run-length path: expansion by the stable radix sort instead of the matrix walk"},
    {"PSS_PERIOD", "-1 (when the text repeats one word)", "0", "builder: 0 = never the closed form for a text that repeats one word"},
    {"PSS_ANCHOR", "-1 (texts of >= 2^20 bytes)", "0|1|1", "builder: never / whenever ties outlive the text rounds the anchor round (anchor_impl.h)"},
    {"PSS_ANCHOR_OMEGA", "0 (as wide as the common prefix allows)", "9|12|17|33", "anchor round: force the window of the minimizers"},
    {"PSS_ANCHOR_MIN_OMEGA", "11", "3|7|9", "anchor round: narrowest window it accepts by itself"},
    {"PSS_ANCHOR_SIDE", "-1 (>= 2^24 bytes, sampled ties show copies)", "0|1", "anchor round: never / always sort the anchors beside the text round (second stream)"},
    {"PSS_ANCHOR_SIDE_PCT", "8", "", "anchor round: share of sampled tied pairs that must be copies for the side line"},
    {"PSS_ANCHOR_CAP_DIV", "5", "3|4", "anchor round: declines above n / this many anchors"},
    {"PSS_NO_PROBE", "unset", "1", "builder: always a text round before the anchor round (no sampling of the ties)"},
    {"PSS_PROBE_SKIP_PCT", "50", "0|20|90", "builder: no text rounds when more than this share of the sampled tied pairs are repeats"},
    {"PSS_NO_MID_TIER", "unset", "1", "rounds: groups above 512 members all take the chained radix sorts"},
    {"PSS_NO_MID_MERGE", "unset", "1", "rounds: groups of 513 .. 4096 members with a crowded bin take the chained sorts (no LDS merge sort)"},
    {"PSS_BIG_MERGE", "-1 (by the average size of the large groups)", "0|1|2", "rounds: groups above 4096 members through the segmented merge sort (2: in text rounds only)"},
    {"PSS_NO_BIG_MERGE", "unset", "", "rounds: overrides PSS_BIG_MERGE"},
    {"PSS_COUNT_SORT", "unset", "1", "rank rounds: groups ranked by counting instead of the merge sort"},
    {"PSS_PERIODIC", "1", "0", "rank rounds: 0 = no periodic keys for the large groups"},
    {"PSS_TIMING", "unset", "", "builder / Writer: per-round and per-stage trace on stderr"},
    {"PSS_DEVICES", "all visible", "0|0,0|0,0,0|all", "devices of Reader(path) / Writer(path) without a device argument: all | comma-separated ordinals"},
    {"PSS_DEVICE", "unset", "", "default device of a process (launchers' LOCAL_RANK etc. are honoured the same way)"},
    {"PSS_IO_THREADS", "8 .. 16 by core count", "2|5", "file path: threads of the I/O pool"},
    {"PSS_INGEST_BLOCK", "32 MiB", "16|40|300", "Writer: size of the direct reads of add_entries_from_file_lines"},
    {"PSS_INGEST_MIN_ROOM", "1 MiB", "1|20", "Writer: room a chunk must have left for one direct read"},
    {"PSS_WRITER_MMAP", "tmpfs: 1, else 0", "0|1", "Writer: records through a shared mapping (1) or pwrite (0)"},
    {"PSS_WRITER_MMAP_MIN", "1 MiB", "16", "Writer: smallest record that goes through the mapping"},
    {"PSS_WRITER_HOST_BUDGET", "8 GiB", "", "multi-device Writer: host text of the chunks in flight (one chunk always goes through)"},
    {"PSS_STRIPES", "8", "", "striped container format 2: stripe files of a Writer"},
    {"PSS_EXPERIMENT_NO_FILE", "unset", "", "measurement only: the record thread copies without writing the file"},
    {"PSS_RESULT_ORDER", "text", "sa", "Reader: 'sa' = the reference's order inside a chunk (suffix-array order of every entry's first hit)"},
    {"PSS_READER_HBM_BUDGET", "whatever hipMalloc grants minus 2 GiB", "", "Reader: suffix arrays beyond this many bytes of HBM stay in pinned host memory"},
    {"PSS_READER_AUTO_RESIDENCY", "1", "", "Reader: 0 = the residency manager never moves a suffix array by itself"},
    {"PSS_NO_KEY_SAMPLES", "unset", "", "Reader: no key-sample table"},
    {"PSS_SAMPLE_SHIFT", "11", "", "Reader: one key sample per 2^shift suffixes"},
    {"PSS_NO_SMALL_PATH", "unset", "1", "search: no fused single-query kernels"},
    {"PSS_NO_BLOCK_PATH", "unset", "1", "search: small batches one wave per pair instead of one workgroup"},
    {"PSS_NO_SEARCH_STAGE", "unset", "1", "search: no pinned staging of queries and results"},
    {"PSS_WAVE_SEARCH", "unset", "1", "search: interval kernel one wave per pair at every batch size"},
    {"PSS_NO_GROUP_SEARCH", "unset", "1", "search: never 16 lanes per pair"},
    {"PSS_NO_MID_PIPELINE", "unset", "1", "search: always the general multi-kernel pipeline"},
    {"PSS_PINNED_POOL_BYTES", "40 GiB", "", "search: pinned host memory the result pool keeps between batches"},
    {"PSS_NO_PINNED_RESULTS", "unset", "1", "search: large results into pageable memory"},
    {"PSS_LANE_SEARCH_MIN", "8192", "1|100000", "search: pairs from which one lane per pair searches"},
    {"PSS_SEARCH_EVENTS", "unset", "", "single-query path: HIP events around the fused kernel (fills ms_device; ~4 us per query)"},
    {"PSS_RESIDENT_IDLE_US", "1000", "", "low-latency mode: the resident kernel leaves after this long without a query"},
    {"PSS_RESIDENT_LIFE_US", "50000", "", "low-latency mode: ... and after this long in any case"},
    {"PSS_TRACE_RESIDENT", "unset", "", "libpss_trace.so only: print the resident kernel's phase stamps"},
    {"PSS_RCCL_LIB", "librccl.so of the process", "", "gather inside the C ABI: the RCCL library to load"},
    {"PSS_RCCL_TIMEOUT_MS", "60000", "", "gather inside the C ABI: bound of every wait for the peers"},
};
const int kNumKnobs = (int)(sizeof(kKnobs) / sizeof(kKnobs[0]));

const char *knob(const char *name)
{
    for (int i = 0; i < kNumKnobs; ++i)
        if (strcmp(kKnobs[i].name, name) == 0) return getenv(name);
    static std::atomic<bool> said{false};
    if (!said.exchange(true)) fprintf(stderr, "[pss] %s is not in the switch registry (knobs.h): treated as unset\n", name);
    return nullptr;
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
    // The contexts that exist, listed under the table's lock; every one is then locked WITHOUT it: a reader that runs out
    // of HBM inside its own context's lock asks for the builder's context (get_build_ctx takes the table's lock) -- the
    // two orders must not cross.  Contexts are never destroyed, so the pointers stay good.
    std::vector<DeviceCtx *> live;
    {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        for (DeviceCtx *table : {g_ctx, g_bctx})
            for (int d = 0; d < kMaxDevices; ++d)
                if (table[d].device >= 0) live.push_back(&table[d]);
    }
    for (DeviceCtx *cp : live) {
        {
            DeviceCtx &c = *cp;
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
    }
    pinned_pool_trim();
}

}  // namespace pss
