// common.h -- host-side plumbing shared by the engine: error reporting across
// the C ABI, per-device context (stream + grow-only HBM workspace slots).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pss.h"

#include "knobs.h"

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

// Environment switches of the search path (exploration and tests).  Read ONCE, when the library first
// touches a device -- a getenv() per search call is measurable on the single-query latency path --
// and again only when a test calls pss_reload_env().  None of them changes a result.
struct SearchKnobs {
    bool no_small_path = false;     // PSS_NO_SMALL_PATH     skip the fused small-batch kernels
    bool no_block_path = false;     // PSS_NO_BLOCK_PATH     small batches: one wave per pair instead of one workgroup
    bool no_search_stage = false;   // PSS_NO_SEARCH_STAGE   no pinned staging of queries / results
    bool wave_search = false;       // PSS_WAVE_SEARCH       interval search: one wave per pair at every batch size
    bool no_group_search = false;   // PSS_NO_GROUP_SEARCH   never 16 lanes per pair
    bool no_mid_pipeline = false;   // PSS_NO_MID_PIPELINE   always the general multi-kernel pipeline
    bool no_pinned_results = false; // PSS_NO_PINNED_RESULTS large results into pageable memory
    bool small_path_events = false; // PSS_SEARCH_EVENTS     HIP events around the fused single-query kernel (fills ms_device there)
    uint64_t lane_search_min = 8192;   // PSS_LANE_SEARCH_MIN  pairs from which one lane per pair searches
    uint32_t resident_idle_us = 1000;  // PSS_RESIDENT_IDLE_US  resident search kernel: leaves after this long without a query ...
    uint32_t resident_life_us = 50000; // PSS_RESIDENT_LIFE_US  ... and after this long in any case (the next query starts another)
    void load();
};
const SearchKnobs &search_knobs();
void reload_search_knobs();

// Pinned host blocks for large results (bytes + offsets of one batch): pinning is what makes the D2H
// copy run at link speed instead of ~10 GB/s through the runtime's bounce buffers, but pinning a
// gigabyte costs more than copying it, so freed blocks are kept (a few, bounded) and handed out again.
void *pinned_pool_alloc(size_t bytes, size_t *granted);
void pinned_pool_free(void *p, size_t granted);
void pinned_pool_trim();

// Mailbox of the resident search kernel (search.hip, low-latency mode of a reader): fine-grained pinned host memory.
// The host posts a query by filling `post` -- one 64-byte line the kernel polls with ONE load: the query's bytes (when
// they fit; longer ones go to `query` and cost a second trip over PCIe), then the sequence number at BOTH ends of the
// line, so whichever 32-byte half of a (possibly split) read carries a new number also carries what was written before
// it.  plen = kResidentStop with a new sequence number tells the kernel to leave.  The kernel answers by writing the
// sequence number to done_seq after its results; `exited` is its last word: a query posted after it, or one it raced
// with, was not served.
struct ResidentMailbox {
    struct Post {
        volatile uint32_t seq_a;
        volatile uint32_t plen;
        uint8_t bytes[52];
        volatile uint32_t seq_b;
    } post;
    uint8_t query[256 + 64];         // queries of more than sizeof(Post::bytes) bytes
    // kernel -> host, ONE 8-byte system-scope store: the sequence number of the last query answered (low word) and the
    // checksum of the query bytes the workgroups worked on (high word, their sum mod 2^32) -- a torn or stale read of
    // the posted line shows there, also when every workgroup read the same wrong bytes
    volatile uint64_t done_seq_echo;
    volatile uint32_t exited;        //                 1: the kernel has left (lease over, or told to)
    volatile uint32_t closing;       //                 1: the kernel is about to leave and looks once more
    volatile uint32_t pad2[12];
};
static_assert(offsetof(ResidentMailbox, done_seq_echo) % 8 == 0, "one aligned 8-byte store");
// Checksum of a posted query (FNV-1a over the zero-padded 8-byte words and the length): what the host expects in `echo`
// from ONE workgroup; W workgroups add up to W times it (mod 2^32).
static inline uint32_t resident_query_checksum(const uint8_t *q, uint32_t plen)
{
    uint32_t h = 0x811C9DC5u ^ plen;
    for (uint32_t i = 0; i < plen; i += 8) {
        uint64_t w = 0;
        for (uint32_t k = 0; k < 8 && i + k < plen; ++k) w |= (uint64_t)q[i + k] << (8 * k);
        h = (h ^ (uint32_t)w) * 0x01000193u;
        h = (h ^ (uint32_t)(w >> 32)) * 0x01000193u;
    }
    return h;
}
static_assert(sizeof(ResidentMailbox::Post) == 64, "the posted line is one cache line");
constexpr uint32_t kResidentStop = 0xffffffffu;

// One per (process, device): a stream and named workspace slots.
struct DeviceCtx {
    // Workspace slots, staging buffers and the stream are shared by every handle on the
    // device; ctypes releases the GIL, so two Python threads can be inside the library at
    // once.  Every entry point that touches the context holds this lock for its duration.
    std::recursive_mutex mu;
    int device = -1;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    static constexpr int kSlots = 56;
    DevBuf slot[kSlots];
    static constexpr size_t kPinnedBytes = (size_t)640 << 10;   // 64 KiB of counters / small tables / query staging, 64 KiB of
                                                                // result bytes, 512 KiB of entry records (search.hip, SM_OFF_*)
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
    // Resident search kernel (low-latency mode of a reader, search.hip): its own stream, its own fine-grained pinned
    // arena (same layout as `pinned` on the small path + the mailbox), and what the running kernel was launched with.
    struct Resident {
        hipStream_t stream = nullptr;
        void *arena = nullptr, *arena_dev = nullptr;
        bool running = false;
        const void *chunks = nullptr;    // launch arguments of the running kernel: a query for anything else restarts it
        uint32_t nc = 0, spread = 0;
        void *d_arena = nullptr;
        uint32_t seq = 0;
        uint64_t launches = 0, served = 0, torn = 0;
        void post(const uint8_t *q, uint32_t plen);      // next sequence number, query (or kResidentStop) into the mailbox
    } resident;
    static constexpr size_t kResidentMailboxOff = 24576;   // inside the first 32 KiB of the arena (search.hip, SM_OFF_*)
    int ensure_resident();               // stream + arena, once
    void stop_resident();                // tells a running kernel to leave and waits for it (cheap when none runs)
    // Which initial sort the last build on this device took, and for what kind of text (the byte values present, the
    // size class): consecutive chunks of one corpus take the same one, so the next build skips the sizing sample that
    // would only say so again (sa_build.hip; a wrong guess is caught by the sorts' own exact checks and costs one restart).
    uint32_t plan_present[8] = {};
    uint32_t plan_logn = 0;
    int plan_path = 0;                   // 0 none, 1 hybrid MSD, 2 sample sort
    uint8_t plan_lut[256] = {};          // ... and its byte -> code table (the next build recodes with it inside the sort)
    uint32_t plan_sigma = 0;
    // ... 2: it took the sample sort, whose sorted sample (slot S_SSPLAN of sa_build.hip) cuts the next chunk of that
    // exact size and alphabet too
    uint64_t ss_plan_tag = 0;            // ss_geometry_tag() of the text the sample came from (0: none)
    uint32_t ss_plan_radix = 0;
    bool ss_refused_note = false;        // (the running build started over because the plan's sample was refused)
    uint32_t ss_plan_skip = 0, ss_plan_backoff = 0;      // builds that go by before the plan is tried again after a refusal
    // ... and whether its anchors were sorted beside the text round (sa_build.hip, SideAnchors): the depth that side line
    // was started for, the size class and the bits per symbol.  The next chunk of that kind starts its side line at once --
    // beside the initial sort as well -- instead of waiting for a sample of its ties to show copies.
    uint64_t side_plan_heff = 0;
    uint32_t side_plan_logn = 0;
    int side_plan_b = 0;
    double restart_ms = 0.0;             // device time of the attempts the running build gave up (sa_build.hip, start_over)
    int restart_depth = 0;
    // Two pinned staging buffers + a copy stream: file <-> HBM transfers are
    // double-buffered so the PCIe copy of piece i overlaps the file I/O of piece i+1.
    static constexpr size_t kStage = (size_t)64 << 20;
    void *stage[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    int ensure_staging();
    // Ring of pinned pieces for the parallel file path of the reader (IoPool threads pread into them, the copy stream
    // uploads them): kIoPieces x kIoPiece bytes, allocated on first use.
    static constexpr size_t kIoPiece = (size_t)16 << 20;
    static constexpr int kIoPieces = 12;
    void *io_ring[kIoPieces] = {};
    hipEvent_t io_ev[kIoPieces] = {};
    int ensure_io_ring();
    // A second context on the same device for work the builder runs BESIDE its main line (sa_build.hip: the anchors' own
    // sort while the text round over the whole list is under way): its own stream, pinned scratch and slots, no lock of
    // its own -- it belongs to whoever holds this context.  Created on first use (get_helper_ctx).
    DeviceCtx *helper = nullptr;
};
int get_helper_ctx(DeviceCtx *parent, DeviceCtx **out);

// A few threads that pread / pwrite disjoint pieces of one file (round 4).  A chunk record is 2.7 GB at the default chunk
// size: one thread copying it into (out of) the page cache moves 3 - 6 GB/s, which was the whole end-to-end time of the
// Writer and of Reader::new.  Record offsets are known in advance (8 + 5 n bytes per chunk, src/lib.rs:112-119), so the
// pieces of one record can be written in any order by any thread -- the file that results is byte-identical.
class IoPool {
public:
    struct Batch {                       // completion state of the pieces one caller has submitted
        std::mutex mu;
        std::condition_variable cv;
        size_t submitted = 0, finished = 0;
        int err = 0;                     // first errno (EIO for a short read)
    };
    static IoPool &get();                // process-wide, threads started on first use (PSS_IO_THREADS, default 8 .. 16)
    // One piece: the whole of [off, off + len) of fd to / from buf.  *done (optional) is set to 1 when the piece is through.
    void submit(Batch *b, int fd, bool write, void *buf, size_t len, int64_t off, std::atomic<int> *done = nullptr);
    // One piece of a plain copy (fd == -2): len bytes from src to dst -- dst a shared mapping of the file: page-cache
    // pages are then allocated by the page faults of several threads at once, where write(2) on one file holds the
    // inode's lock exclusively and lets one thread copy at a time.
    void submit_copy(Batch *b, void *dst, const void *src, size_t len, std::atomic<int> *done = nullptr);
    static int wait_all(Batch *b);       // every piece submitted so far: 0 or the first errno
    static void wait_flag(Batch *b, std::atomic<int> *done);
    int threads() const { return (int)workers_.size(); }
    ~IoPool();

private:
    IoPool();
    struct Task {
        Batch *b;
        int fd;
        bool write;
        void *buf;
        size_t len;
        int64_t off;
        std::atomic<int> *done;
    };
    void run();
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Task> q_;
    bool stop_ = false;
    std::vector<std::thread> workers_;
};

// Validates `device`, makes it current, returns its context (created lazily): the one the READER side works in --
// resident indexes, searches, gathers, uploads.
int get_ctx(int device, DeviceCtx **out);
// The context of the BUILDER side of the same device (pss_sa_build*, the Writer's builder threads): its own lock, stream,
// pinned scratch and workspace slots (round 5).  With one context per device a Reader waited for a whole build (9 .. 131 ms)
// whenever a Writer shared its GPU, and a gather over RCCL held up every build; the two sides never touch each other's
// slots (sa_build.hip 0-9, 26-49; search.hip 10-23, 28, 46, 47, 50-54; the Writer 24, 25), so they need not share a lock.
int get_build_ctx(int device, DeviceCtx **out);
// Frees every workspace slot of every context (memory pressure relief).
void trim_all();
// Bytes of HBM the workspace slots of one device's contexts hold right now (builder, its helper line, search scratch).
uint64_t workspace_bytes(int device);

static inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace pss
