// The Writer side of the C ABI: the three-stage pipeline (ingest, suffix-array builds, file), striping over several GPUs and the file formats.
// Part of capi.cpp: included there, in this order, into the one translation unit (the pieces share its
// anonymous-namespace helpers); not a header for anybody else.

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
