// The Reader side of the C ABI: opening an index (one GPU, a shard of it, or several GPUs in one process), the residency manager, the batch / count / device-result calls and the merge of packed results.
// Part of capi.cpp: included there, in this order, into the one translation unit (the pieces share its
// anonymous-namespace helpers); not a header for anybody else.

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
