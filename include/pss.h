/*
 * pss.h -- C ABI of libpss.so, the MI355X (gfx950) engine behind the
 * pysubstringsearch Writer/Reader API.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, returns an
 * int status (0 = ok, negative = error, see PSS_E*) and never throws or aborts.
 * Each declaration cites the reference interface (file:line under the
 * upstream tree, Intsights/PySubstringSearch v0.7.1) it replaces.  The
 * reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Handles are not thread-safe: one thread per handle at a time (the reference's
 * pyclass methods take `&mut self`, src/lib.rs:67,88,105,126,201).
 *
 * There is no CPU fallback: every compute entry point fails with PSS_EDEVICE
 * when no HIP device is usable.
 */
#ifndef PSS_H
#define PSS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSS_OK 0
#define PSS_EINVAL (-1)   /* bad arguments            (libsais.c:6599-6602 returns -1) */
#define PSS_ENOMEM (-2)   /* host/device allocation   (libsais.c:6505-6507 returns -2) */
#define PSS_EIO (-3)      /* I/O error, errno is set  (io::Error -> OSError, src/lib.rs:55,71,112-118) */
#define PSS_ETOOBIG (-4)  /* "entry is too big"       (src/lib.rs:92-94 -> ValueError) */
#define PSS_EDEVICE (-5)  /* HIP runtime error / no device */
#define PSS_EFORMAT (-6)  /* malformed or truncated .idx (UnexpectedEof in src/lib.rs:175-179) */

typedef struct pss_writer pss_writer;
typedef struct pss_reader pss_reader;
typedef struct pss_result pss_result;

/* ---- library ---------------------------------------------------------- */

/* Number of visible HIP devices (0 when none / runtime unusable). */
int pss_device_count(void);

/* The devices a handle opened with device == -1 uses -- what the reference-shaped call (`Reader(path)`, `Writer(path)`,
 * no device argument) gets; the reference fans out over every core of the machine without being asked (rayon's global
 * pool, src/lib.rs:205-207):
 *   PSS_DEVICES=all | 0,1,...   every visible device / the listed ordinals (one may be listed more than once);
 *   else (unset, or set to nothing) PSS_DEVICE=k, LOCAL_RANK=k, SLURM_LOCALID=k, OMPI_COMM_WORLD_LOCAL_RANK=k or
 *        MV2_COMM_WORLD_LOCAL_RANK=k (a launcher runs one process per GPU): device k mod count;
 *   else every visible device.
 * Writes at most cap ordinals to out and returns how many (>= 1); -1 (message in pss_last_error) when PSS_DEVICES is set
 * to something that is neither 'all' nor a list of ordinals below the device count -- handles opened without a device then
 * fail with PSS_EINVAL instead of spreading over every GPU (round 5: it used to count as unset). */
int32_t pss_default_devices(int32_t *out, int32_t cap);

/* Copies the calling thread's last error message into buf (NUL-terminated,
 * truncated to cap); returns the untruncated length. */
size_t pss_last_error(char *buf, size_t cap);

/* Test hook: re-reads the PSS_* environment switches of the search path, which the library reads
 * once (when it first needs them) rather than on every call. */
int pss_reload_env(void);

/* Frees the grow-only HBM workspace of every device (the builder keeps ~45 n
 * bytes around for reuse; resident Reader chunks are not touched). */
int pss_release_workspace(void);

/* HBM (bytes) the grow-only workspaces of `device` hold right now: the suffix-array builder's buffers (~45 n bytes after a
 * build of an n-byte chunk, more for natural text: DESIGN.md 3), its helper line's, and the scratch of the search path.
 * Resident indexes are not counted (pss_reader_residency).  A Writer that shares a GPU with a resident Reader can see
 * what one chunk in flight costs and give it back (pss_release_workspace).  No reference counterpart. */
uint64_t pss_workspace_bytes(int32_t device);

/* The environment switches the library reads (csrc/knobs.h: one registry, every read goes through it): how many there
 * are, and for switch i its name, what an unset switch means, the values a fuzzer may set (separated by '|'; empty: not
 * drawn) and what it does.  The strings are static.  None of the switches changes a result.  No reference counterpart. */
int32_t pss_knob_count(void);
int pss_knob_info(int32_t i, const char **name, const char **dflt, const char **fuzz, const char **what);

/* sizeof(pss_sa_stats) / sizeof(pss_search_stats) as the library was built: a binding that declares the structs itself
 * (ctypes, cgo, JNI) compares them with its own before the first call that fills one. */
uint64_t pss_sa_stats_size(void);
uint64_t pss_search_stats_size(void);

/* ---- suffix-array builder seam ----------------------------------------- */

/* Per-build statistics (filled when `stats` is non-NULL). */
typedef struct pss_sa_stats {
    uint32_t sigma;            /* distinct byte values in the text */
    uint32_t code_bits;        /* bits per recoded symbol (b) */
    uint32_t key_chars;        /* symbols packed into the initial 64-bit key (h0) */
    uint32_t initial_passes;   /* radix passes of the initial key sort */
    uint32_t rounds;           /* prefix-doubling rounds executed after it */
    uint32_t round_passes;     /* radix passes summed over all rounds */
    uint64_t sum_active;       /* sum over rounds of suffixes still unresolved */
    uint64_t sort_elems;       /* sum over every radix pass of elements moved */
    double ms_total;           /* device time of the whole build (HIP events) */
    double ms_sort;            /* device time inside radix passes (profile mode only) */
    uint64_t sort_launches;    /* radix-pass (scatter kernel) launches */
    /* profile mode: rs_scatter_kernel<false>, the generic (u64 key, u32 value) pass of the rounds
     * and of the sizing sample (reads 8 B key + 4 B value, writes the same: 24 B per element) */
    double ms_pairs;           /* summed duration of its launches */
    uint64_t pairs_launches;
    uint64_t pairs_elems;      /* elements summed over those launches */
    double ms_text;            /* first pass, rs_scatter_kernel<true> (1 B in, 12 B out) */
    uint64_t text_launches;
    uint64_t mode;             /* how ties were resolved: 0 doubling over an inverse SA (dense), 1 sparse
                                  (hash + key search, no ISA), 2 text rounds only, 3 text rounds then dense */
    uint64_t text_rounds;      /* rounds that extended ties with symbols packed from the text */
    uint64_t big_elems;        /* members of groups > 512 handled by the chained radix sorts, summed */
    uint64_t key_bits;         /* bits of the initial sort key: key_chars * code_bits minus the low bits of the
                                  last symbol that were left out to save a pass */
    /* profile mode: passes of the initial sort, fs_scatter_kernel<KIN, KOUT> (key plane bytes in / out;
     * KIN 0 = packed from the text, KOUT 0 = last pass; values are 4 B in and out, 4 B only out of the
     * text pass).  Index = (KIN / 4) * 3 + KOUT / 4. */
    double fs_ms[9];
    uint64_t fs_launches[9];
    uint64_t fs_elems[9];
    /* initial sort taken by the hybrid MSD path (two global 10-bit partition passes over 8-byte
     * [key | index] elements, then every joint bucket sorted in LDS): 1 when it ran */
    uint64_t msd;
    uint64_t msd_buckets;      /* non-empty joint (20-bit) buckets; filled whenever the path was tried */
    uint64_t msd_max_bucket;   /* largest of them (the path declines above 4088) */
    uint64_t msd_tiles;        /* local-sort workgroups */
    double msd_ms_g1;          /* profile mode: partition scatter from the text (1 B in, 8 B out per suffix) */
    double msd_ms_g2;          /* ... second partition scatter (8 B in, 8 B out) */
    double msd_ms_local;       /* ... local sort (8 B in, 4 B out, sequential) */
    uint64_t msd_slow_tiles;   /* tiles with crowded bins that took the general local-sort kernel */
    /* run-length path (texts made of long runs of equal bytes): the run heads are sorted as a string of one
     * symbol per run, the other suffixes by one short radix sort.  `rounds` etc. then describe the sort of
     * the reduced string. */
    uint64_t runs;             /* maximal runs of equal bytes in the text (always filled) */
    uint64_t rle;              /* 0: not taken; 1: taken, expansion by a stable radix sort; 2: taken, expansion by the
                                  matrix walk (no sort) */
    uint64_t rle_id_bits;      /* key bits of the expansion sort */
    double rle_ms_table;       /* profile mode: run table and symbols */
    double rle_ms_reduced;     /* ... suffix sort of the reduced string */
    double rle_ms_expand;      /* ... expansion and its radix sort */
    /* initial sort taken by the sample sort over 16-byte [key | index] elements (natural text: texts the radix
     * partition above cannot take because some 20-bit prefix holds far more suffixes than a tile): 1 when it ran */
    uint64_t ss;
    uint64_t ss_buckets;       /* non-empty joint buckets */
    uint64_t ss_max_bucket;    /* largest of them (the path declines above 4088: a sampling accident) */
    uint64_t ss_tiles;         /* local-sort workgroups */
    uint64_t ss_samples;       /* sample members the splitters were taken from */
    double ss_ms_sample;       /* profile mode: drawing and sorting the sample */
    double ss_ms_g1;           /* ... first partition (digits from the text, scatter: 3 B in, 2 + 16 B out per suffix) */
    double ss_ms_g2;           /* ... second partition (digits 16 in / 2 out, scatter 18 in / 16 out) */
    double ss_ms_local;        /* ... local merge sort (16 B in, 4 B out) */
    double ms_initial;         /* device time from the start of the build to the end of the initial sort (always filled) */
    uint64_t period;           /* word length p when the head of the text repeats one word of 2 .. 1024 bytes (else 0) ... */
    uint64_t period_extent;    /* ... how far that repetition reaches from the start of the text ... */
    uint64_t period_path;      /* ... 1: it covers the text (at most 1024 bytes behind it) and the suffix array was written
                                  in closed form (rle_build.h): rotation blocks + a host sort of the last few suffixes */
    uint64_t plan_hint;        /* 0: the sizing sample chose the initial sort; 1: the previous build on this device sorted
                                  the same kind of text (same byte values, same size class) with the MSD sort and this
                                  build went straight to it (PSS_NO_PLAN_CACHE=1: never); 2: ... and recoded the text with
                                  the remembered alphabet inside the sort's first pass, without an alphabet pass of its
                                  own (the pass checks that every byte has a code; PSS_NO_PLAN_FRONT=1: never); 3: no
                                  plan, but the chunk's own symbol counts are close to uniform (sum p^2 < 0.035: log lines,
                                  identifiers -- not natural language) and it went to the MSD sort the same way, without
                                  recode pass and sizing sample */
    /* anchor round (ties that outlive the text rounds -- duplicated stretches of text -- are resolved by ONE round
     * keyed by the ranks of a content-defined sample of positions, the anchors, instead of log2(repeat length) rank
     * rounds over the whole text; anchor_impl.h) */
    uint64_t anchor;           /* 1 when it ran */
    uint64_t anchor_omega;     /* window of the minimizers (every window of this many positions holds an anchor) */
    uint64_t anchor_w;         /* bytes hashed per position */
    uint64_t anchor_count;     /* anchors (also filled when the path declined: more than n / 5) */
    uint64_t anchor_active;    /* suffixes still tied when the round ran */
    uint64_t anchor_depth;     /* symbols every tied group shared at that point */
    uint64_t anchor_text_rounds; /* text rounds of the anchors' own sort (names of 2 omega + w - 1 symbols) */
    uint64_t anchor_rounds;    /* rank rounds over the string of anchor names */
    uint64_t anchor_sum_active;  /* anchors still tied, summed over those rounds */
    uint64_t anchor_left;      /* suffixes the round left tied (always 0; a non-zero value is an internal error that the
                                  rank rounds then repair) */
    uint64_t anchor_levels;    /* anchor levels stacked: 1 = anchors of the text; 2 = anchors of the string of their names, ... */
    uint64_t probe_pairs;      /* before the first text round: neighbours of the active list sampled from one group ... */
    uint64_t probe_same;       /* ... and those that share 48 more symbols (most of them: the ties are repeats, no text round) */
    uint64_t periodic_rounds;  /* rank rounds (of any anchor level) in which large groups took the periodic key: members of a run of
                                  period p <= h ordered by how far the period goes on and how it breaks, not by ISA[i + h] */
    uint64_t periodic_members; /* ... and the members of those groups, summed over the rounds */
    uint64_t ss_planned;       /* 1: the sample sort cut this chunk with the sorted sample of the previous chunk of the same
                                  size and alphabet (no sample of its own, no sizing sample: plan_hint = 1 with ss = 1) */
    uint64_t dup_screen;       /* first chunk with log-like symbol counts: places among 8192 sampled ones whose 16 bytes occur at an
                                  earlier sampled place too (>= 8: copies, no shortcut to the MSD sort) */
    uint64_t ss_plan_refused;  /* 1: the previous chunk's sample left a bucket beyond a tile; the build started over
                                  without the plan (the attempt is part of ms_total, ms_restarts), later chunks wait before they try */
    uint64_t ss_declined_nomem; /* 1: no room in HBM for the sample sort's two 16 n-byte element buffers -- the build went on
                                  with the LSD passes instead of failing */
    uint64_t anchor_side;      /* 1: the anchors were selected and sorted BESIDE the text round that precedes the anchor round (second
                                  stream, second host thread: sa_build.hip, SideAnchors); 2: started so, and thrown away */
    double anchor_ms;          /* device time of the anchors' selection and sort */
    double ms_restarts;        /* part of ms_total: attempts given up (a remembered plan, or the shortcut of a first chunk, that
                                  this text did not fit -- the build started over without it) */
    uint64_t msd_lookback;     /* 1: the MSD sort took its two digits in LSD order and partitioned the second pass in one sweep
                                  (decoupled look-back, no second histogram pass: round 6; texts below 2^30 bytes) */
} pss_sa_stats;

/*
 * Drop-in for `libsais(T, SA, n, 0, NULL)` as called by
 * construct_suffix_array (src/lib.rs:24-40; contract src/libsais/libsais.h:57-65,
 * src/libsais/libsais.c:6597-6610): SA[0..n) becomes the permutation of 0..n-1
 * that orders the suffixes of T by unsigned bytes, a proper prefix first.
 * n == 0 writes nothing, n == 1 writes SA[0] = 0.  T and SA are HOST pointers;
 * the text is uploaded, sorted on `device` and the result copied back.
 * Returns 0, PSS_EINVAL (NULL pointers, n < 0), PSS_ENOMEM or PSS_EDEVICE.
 */
int32_t pss_sa_build(const uint8_t *T, int32_t *SA, int32_t n, int32_t device);

/* Same, with T and SA already resident in the HBM of `device` (T must be
 * readable for n bytes, SA writable for n int32).  `flags` bit 0 = profile
 * mode (per-pass HIP events, fills ms_sort); bit 1 = build without any plan
 * shortcut (internal: restarts); bit 2 = no shortcut from the symbol counts of a
 * first chunk (internal: restarts); bit 3 = forget what earlier builds on this
 * device left behind first (a COLD build: what the first chunk of a corpus sees;
 * bench.py's value_cold).  This is the timed region of bench.py: inputs resident,
 * no PCIe. */
int32_t pss_sa_build_device(const void *d_T, void *d_SA, int32_t n, int32_t device,
                            uint32_t flags, pss_sa_stats *stats);

/* The builder's device radix sort on its own (test / measurement utility, no
 * reference counterpart): stable LSD sort of n (u64 key, u32 value) pairs by the
 * key bits [0, key_bits), in place, all pointers resident on `device`.
 * *ms_scatter (optional) receives the summed HIP-event time of the scatter kernel. */
int32_t pss_sort_pairs_device(void *d_keys, void *d_vals, uint32_t n, int32_t key_bits, int32_t device,
                              double *ms_scatter);

/* ---- Writer (src/lib.rs:42-144; pysubstringsearch/__init__.py:6-41) ----- */

/* Writer::new, src/lib.rs:50-65.  Creates/truncates `path`.  max_chunk_len < 0
 * means None (512 MiB, src/lib.rs:57).  Suffix arrays are built on `device`. */
int pss_writer_open(const char *path, int64_t max_chunk_len, int32_t device, pss_writer **out);
/* The same with the container format chosen (no reference counterpart).
 * 1 = the reference's records, u32le len | text | u32le 4n | n x i32le (src/lib.rs:112-119), whose
 *     u32 at lib.rs:116 limits a chunk to < 1 GiB of text;
 * 2 = "PSSIDX\x02\x00" | u32le flags (0) | u32le reserved (0), then records with 64-bit lengths,
 *     u64le n | text | u64le 4n | n x i32le: chunks of up to 2^31 - 1 bytes (max_chunk_len beyond
 *     that is PSS_EINVAL).  pss_reader_open recognises either format by the magic.
 * 2 | PSS_FORMAT_STRIPED (round 5) = format 2 with header flags bit 0 set (bits 8..15: S, bits 16..23: log2 of the
 *     unit, 24): the records hold no suffix arrays (u64le n | text | u64le 4n), the arrays live in S files
 *     `<path>.sa0` .. `<path>.sa<S-1>` (S = 8; PSS_STRIPES) -- every chunk's array starts a new 16 MiB unit, unit u
 *     sits in file u mod S at offset (u / S) * 16 MiB -- and are written and read by several threads, a file each:
 *     one file in the page cache takes 11 - 14 GB/s on the test box whatever the number of writers, a file per
 *     writer 47 - 97.  The stripe files travel with the index file; pss_reader_open reads the flag. */
#define PSS_FORMAT_STRIPED 0x100
int pss_writer_open_format(const char *path, int64_t max_chunk_len, int32_t device, int32_t format_version,
                           pss_writer **out);
/* The same over several devices ("each GPU builds its chunks; host writes records in chunk
 * order"): chunk k of the file is built on devices[k % n_devices], up to 2 n_devices chunks are in flight
 * (G being built, the records of the earlier ones streaming to the file in chunk order), and the file
 * is byte-identical to the single-device one.  The reference builds one chunk at a time inside
 * dump_data (src/lib.rs:105-124).  The same device may be listed more than once. */
int pss_writer_open_multi(const char *path, int64_t max_chunk_len, const int32_t *devices, int32_t n_devices,
                          int32_t format_version, pss_writer **out);
/* Writer::add_entry, src/lib.rs:88-103.  PSS_ETOOBIG when len > limit. */
int pss_writer_add_entry(pss_writer *w, const uint8_t *text, uint64_t len);
/* Writer::add_entries_from_file_lines, src/lib.rs:67-86 (bstr for_byte_line rule). */
int pss_writer_add_file_lines(pss_writer *w, const char *path);
/* Writer::dump_data, src/lib.rs:105-124: emits u32le len | data | u32le 4n | n x i32le. */
int pss_writer_dump(pss_writer *w);
/* Writer::finalize, src/lib.rs:126-135. */
int pss_writer_finalize(pss_writer *w);
/* Drop for Writer, src/lib.rs:138-144: finalize, close the file, free. */
int pss_writer_close(pss_writer *w);
/* Current chunk limit (Vec capacity in the reference, src/lib.rs:62,75,92,96). */
uint64_t pss_writer_chunk_limit(const pss_writer *w);
/* Which way the bytes went (round 5; diagnostics, nothing a caller must look at): records written through a shared
 * mapping of the index file (tmpfs; PSS_WRITER_MMAP=0|1 overrides, PSS_WRITER_MMAP_MIN = smallest such record, default
 * 1 MiB) or pwritten; bytes of pss_writer_add_file_lines input read straight into the chunk or through a block buffer
 * (lines with a '\r', lines that do not fit the chunk any more). */
typedef struct pss_writer_io {
    uint64_t records_mapped;
    uint64_t records_pwritten;
    uint64_t ingest_direct_bytes;
    uint64_t ingest_copied_bytes;
} pss_writer_io;
int pss_writer_io_stats(pss_writer *w, pss_writer_io *out);

/* ---- Reader (src/lib.rs:146-288; pysubstringsearch/__init__.py:44-73) --- */

/* Reader::new, src/lib.rs:162-199.  Parses the chunk records of `path` (the reference
 * container, or format 2 -- recognised by its magic, see pss_writer_open_format) and
 * makes the text AND suffix array of every chunk c with
 * c % shard_count == shard_index resident on `device`: in HBM, the suffix arrays
 * beyond the HBM budget in pinned host memory (pss_reader_residency)
 * (shard_index 0, shard_count 1 = whole file).  Every resident chunk also gets a
 * table of key samples (first 8 bytes of every 2048th suffix) that the searches
 * consult before the suffix array. */
int pss_reader_open(const char *path, int32_t device, int32_t shard_index, int32_t shard_count,
                    pss_reader **out);
/* The same over several devices in ONE process, no launcher, no torch (the reference fans a search over all chunks
 * inside the process too: rayon, src/lib.rs:207, 280-284): chunk c of the file becomes resident on
 * devices[c % n_devices]; every device answers a batch for its chunks on a thread of its own and the results are
 * merged on the host (query-major, device-major inside a query).  The same ordinal may be listed more than once
 * ("virtual devices": the parts then take turns on that GPU).  pss_reader_search_batch / count_batch / residency /
 * evict / promote work on such a reader; the device-resident result and the chunk hand-off calls do not. */
int pss_reader_open_multi(const char *path, const int32_t *devices, int32_t n_devices, pss_reader **out);
/* Observer of the placement above: the number of parts of the reader (1 for a single-device reader) and, in counts[0 .. cap),
 * how many chunks each part holds -- chunk c of the file lives in part c % G, so 15 chunks over 8 devices read
 * 2,2,2,2,2,2,2,1 ("report this imbalance").  No reference counterpart. */
uint64_t pss_reader_part_chunks(const pss_reader *r, uint64_t *counts, uint64_t cap);

/* Residency manager ("LRU when index > HBM"; the reference keeps every suffix array on disk,
 * src/lib.rs:179-189).  A reader whose suffix arrays do not all fit the HBM budget keeps a decayed per-chunk count of the
 * hits its batches find and, between batches, lets the hottest suffix array of the host tier change places with the
 * coldest one in HBM when it is more than twice as hot (at most one exchange per batch).  On by default
 * (PSS_READER_AUTO_RESIDENCY=0 or on = 0: off); pss_reader_evict_chunk / promote_chunk remain as explicit overrides.
 * pss_reader_chunk_tiers: tiers[c] = 0 (suffix array of chunk c in HBM) or 1 (pinned host memory), for the first `cap`
 * chunks; *auto_moves = exchanges / promotions the manager has made so far. */
int pss_reader_set_auto_residency(pss_reader *r, int32_t on);
int pss_reader_chunk_tiers(const pss_reader *r, uint8_t *tiers, uint64_t cap, uint64_t *auto_moves);
/* Residency control: move the suffix array of resident chunk `index` (file order) out of HBM into
 * pinned host memory, where the kernels read it over PCIe (evict), or back (promote; PSS_ENOMEM when HBM has no
 * room).  Text and key samples stay in HBM.  No-ops when the chunk already is where it is asked to be. */
int pss_reader_evict_chunk(pss_reader *r, uint64_t index);
int pss_reader_promote_chunk(pss_reader *r, uint64_t index);
/* Low-latency mode for single queries (off by default; single-device readers).  Launching a kernel and waiting for its
 * completion are about half of what one query costs (~10 of ~22 us); with this mode on, the first single query of a
 * burst starts a RESIDENT search kernel -- one workgroup per chunk that waits for queries in a mailbox in
 * pinned host memory -- and the following ones are posted there and answered without a launch (results are the same
 * bytes either way).  The kernel holds a lease, not the GPU: it leaves by itself after PSS_RESIDENT_IDLE_US (1000)
 * without a query and after PSS_RESIDENT_LIFE_US (50000) in any case, so a crashed host or a device-wide
 * synchronisation elsewhere in the process waits no longer than that; the next single query starts another.  Batches,
 * counts, device-resident results, queries with more hits than the path holds, and every call that changes the
 * reader's chunks go through the ordinary path (and stop the kernel where they must).  While it waits, the kernel
 * occupies one workgroup slot per chunk (a chunk with more than 1024 hits sends the query to the ordinary path) and nothing
 * else.  pss_reader_low_latency_stats: kernels started / queries answered by one, per device. */
int pss_reader_set_low_latency(pss_reader *r, int32_t on);
int pss_reader_low_latency_stats(const pss_reader *r, uint64_t *launches, uint64_t *served);
/* Order of the entries INSIDE one chunk's share of a query (round 5).  The reference walks the hits of a chunk in
 * suffix-array order and pushes an entry when its line start is first seen (src/lib.rs:262-276): a chunk's entries come
 * out in suffix-array order of their FIRST hit.  The default here (PSS_ORDER_TEXT) keeps the leftmost match of every
 * entry and lists the entries in suffix-array order of THAT match -- the same multiset (what the reference's tests
 * compare, tests/test_pysubstringsearch.py:32-37), possibly another order when an entry holds the pattern twice.
 * PSS_ORDER_SA reproduces the reference's list element by element (chunks in index order; the reference's order between
 * chunks is whatever its thread pool makes of it, src/lib.rs:207,280).  It takes the general pipeline and one more
 * sort of the hits: batches pay a few per cent, a single query loses the fused path.  PSS_RESULT_ORDER=sa in the
 * environment makes it the default of every reader opened afterwards. */
#define PSS_ORDER_TEXT 0
#define PSS_ORDER_SA 1
int pss_reader_set_result_order(pss_reader *r, int32_t order);
int32_t pss_reader_result_order(const pss_reader *r);
/* An empty reader on `device`, to be filled with pss_reader_add_chunk_device
 * (Writer -> Reader hand-off through HBM, no file). */
int pss_reader_create(int32_t device, pss_reader **out);
/* Adopts COPIES of a device-resident chunk (text n bytes, SA n int32). */
int pss_reader_add_chunk_device(pss_reader *r, const void *d_text, const void *d_sa, uint32_t n);
/* Replaces resident chunk `index` (index == num_chunks appends) by a copy of
 * the device-resident chunk; reuses the HBM allocation when the size matches
 * (re-indexing the same shard repeatedly without allocator traffic). */
int pss_reader_set_chunk_device(pss_reader *r, uint64_t index, const void *d_text, const void *d_sa, uint32_t n);
/* Chunks resident in this reader. */
uint64_t pss_reader_num_chunks(const pss_reader *r);
/* Where the resident index lives.  The text of every chunk is in HBM; a suffix array is too while it
 * fits (PSS_READER_HBM_BUDGET bytes when set, else until 2 GiB of HBM are left), otherwise it stays in
 * pinned host memory and the kernels read it over PCIe (the key samples in HBM confine a query to a few
 * dozen such reads).  host_chunks = chunks whose suffix array is on the host.  No reference counterpart:
 * Reader::new keeps the text in RAM and every suffix array on disk (src/lib.rs:176-189). */
int pss_reader_residency(const pss_reader *r, uint64_t *hbm_bytes, uint64_t *host_bytes, uint64_t *host_chunks);

/* Per-batch statistics of the last pss_reader_search_batch call. */
typedef struct pss_search_stats {
    uint64_t queries;
    uint64_t hits;          /* suffix-array hits before per-chunk dedupe */
    uint64_t entries;       /* entries returned */
    uint64_t result_bytes;
    double ms_device;       /* HIP-event time of the kernels of the batch */
    double ms_interval;     /* ... of the interval-search kernel alone */
    double ms_host;         /* wall time of the whole batch inside the library (host clock) */
} pss_search_stats;

/*
 * Reader::search (src/lib.rs:201-287) for a whole batch in one call, i.e.
 * Reader.search_multiple (pysubstringsearch/__init__.py:61-73): query q is
 * qbytes[qoffsets[q] .. qoffsets[q+1]).  For every query and every resident
 * chunk: the maximal suffix-array interval whose suffixes start with the
 * query (src/lib.rs:209-252), the entry around each hit (newline scan,
 * src/lib.rs:266-273), deduplicated per (query, chunk) on the entry's start
 * offset (src/lib.rs:262,274).  Entries come back query-major (all entries of
 * query 0, then query 1, ...), inside a query chunk-major.
 */
int pss_reader_search_batch(pss_reader *r, const uint8_t *qbytes, const uint64_t *qoffsets,
                            uint32_t nq, pss_result **out);
/*
 * Beyond the reference surface: counts[q] = the number of
 * entries pss_reader_search_batch would return for query q (after the per-chunk
 * dedupe), without materialising a single entry -- the interval search, the
 * dedupe flags and two scans; nothing but nq counters crosses PCIe.  For
 * high-hit queries this is the part of Reader::search (src/lib.rs:254-278) that
 * dominates on the CPU (README.md:57).
 */
int pss_reader_count_batch(pss_reader *r, const uint8_t *qbytes, const uint64_t *qoffsets,
                           uint32_t nq, uint64_t *counts);
/*
 * The same search with the packed result LEFT ON THE DEVICE: the multi-GPU gather (RCCL send / recv of
 * device buffers to the collecting rank, pysubstringsearch_amd/dist.py) takes it from there, so a
 * contributing rank never moves an entry through its host.  Replaces, across GPUs, what
 * `results.lock().extend(local_results)` does across chunk threads in src/lib.rs:280-284.  The pointers
 * are workspace of the device context: valid until the next search or build on that device.
 */
typedef struct pss_device_result {
    uint64_t num_queries;
    uint64_t num_entries;
    uint64_t num_bytes;
    const void *d_counts;    /* u64 [num_queries]  entries per query */
    const void *d_offsets;   /* u64 [num_entries]  start of every entry in d_bytes (entry e ends where e + 1 starts,
                                the last one at num_bytes) */
    const void *d_bytes;     /* u8  [num_bytes]    entries back to back, query-major, chunk-major inside a query */
    int32_t device;
} pss_device_result;
int pss_reader_search_batch_device(pss_reader *r, const uint8_t *qbytes, const uint64_t *qoffsets,
                                   uint32_t nq, pss_device_result *out);
/*
 * Gather of per-rank packed results over RCCL, torch-free (one process per GPU; src/lib.rs:205-207, 280-284 is what it
 * replaces: rayon tasks appending to one Vec under a mutex).  pss_comm_unique_id: 128 bytes (an ncclUniqueId) made by ONE
 * rank and shipped to the others by any means; pss_comm_init: every rank, same id, its rank and its device (collective).
 * pss_gather_packed_rccl: every rank of the communicator calls it with the pss_device_result of ITS chunks for the same
 * queries (pss_reader_search_batch_device); rank dst receives the others' buffers device to device (one grouped
 * ncclSend / ncclRecv batch), merges on its GPU (query-major, rank-major inside a query) and gets the packed result in
 * *out (pss_result_*); the other ranks get *out = NULL.  The RCCL entry points are looked up at run time in the librccl
 * the process has loaded (PSS_RCCL_LIB names another): libpss.so does not link it.
 */
typedef struct pss_comm pss_comm;
int pss_comm_unique_id(uint8_t *id128);
int pss_comm_init(const uint8_t *id128, int32_t world, int32_t rank, int32_t device, pss_comm **out);
int pss_comm_destroy(pss_comm *c);
int pss_gather_packed_rccl(pss_comm *c, const pss_device_result *mine, int32_t dst, pss_result **out);
/*
 * Failure behaviour of the gather (round 5).  Every wait for the peers is bounded: PSS_RCCL_TIMEOUT_MS in the
 * environment (default 60000), pss_comm_set_timeout_ms per communicator; RCCL's asynchronous error state is polled
 * meanwhile.  When the bound passes, or RCCL reports an error, the communicator is ABORTED (ncclCommAbort), the call
 * returns PSS_EDEVICE with the reason in pss_last_error, and every later call on that communicator returns PSS_EDEVICE
 * at once -- readers, builders and other communicators of the process keep working; make a new communicator to go on.
 * The outcome is collective where it can be: a rank that cannot go through with the exchange (the collecting rank out
 * of memory, ranks that answered different numbers of queries) says so in a go / no-go word before anybody sends, and
 * every rank returns an error.  No device context is held while the call waits for a peer.
 * pss_comm_status: PSS_OK while the communicator is usable, PSS_EDEVICE once it was aborted; counters of completed
 * gathers and of aborts (either pointer may be NULL).
 */
int pss_comm_set_timeout_ms(pss_comm *c, uint32_t ms);
int pss_comm_status(pss_comm *c, uint64_t *gathers, uint64_t *aborts);
/*
 * The collectives library as a table (round 5).  By default libpss looks RCCL up in the process (dlopen of librccl.so,
 * PSS_RCCL_LIB).  An application that links RCCL itself hands its entry points in -- signatures as in rccl.h, streams
 * and communicators as void pointers -- and may adopt a communicator it already has (pss_comm_adopt: never destroyed
 * by pss_comm_destroy, but aborted when a gather times out).  The test-suite uses the same seam to make Send / Recv /
 * GroupEnd fail and a Recv never complete.  comm_abort, comm_get_async_error, get_error_string, get_unique_id,
 * comm_init_rank and comm_destroy may be NULL (the last three: pss_comm_adopt only).  NULL restores the lookup;
 * communicators keep the table they were made with.
 */
typedef struct pss_rccl_unique_id { char internal[128]; } pss_rccl_unique_id;      /* ncclUniqueId */
typedef struct pss_rccl_api {
    int (*get_unique_id)(pss_rccl_unique_id *id);
    int (*comm_init_rank)(void **comm, int nranks, pss_rccl_unique_id id, int rank);
    int (*comm_destroy)(void *comm);
    int (*comm_abort)(void *comm);
    int (*comm_get_async_error)(void *comm, int *async_error);
    int (*group_start)(void);
    int (*group_end)(void);
    int (*send)(const void *buf, size_t count, int datatype, int peer, void *comm, void *stream);
    int (*recv)(void *buf, size_t count, int datatype, int peer, void *comm, void *stream);
    int (*all_gather)(const void *sendbuf, void *recvbuf, size_t sendcount, int datatype, void *comm, void *stream);
    const char *(*get_error_string)(int result);
} pss_rccl_api;
int pss_rccl_inject(const pss_rccl_api *api);
int pss_comm_adopt(void *nccl_comm, int32_t world, int32_t rank, int32_t device, pss_comm **out);

/*
 * Host merge of `world` packed results of the same nq queries (one per rank, each query-major) into one:
 * query-major, rank-major inside a query -- the cross-chunk concatenation of src/lib.rs:280-286 across
 * ranks.  counts[r] = u64[nq], offsets[r] = u64[num_entries[r]] entry starts, bytes[r] = num_bytes[r]
 * bytes.  out_counts[nq], out_offsets[sum(num_entries) + 1], out_bytes[sum(num_bytes)] are the caller's.
 */
int pss_merge_packed(uint32_t world, uint64_t nq, const uint64_t *const *counts, const uint64_t *const *offsets,
                     const uint8_t *const *bytes, const uint64_t *num_entries, const uint64_t *num_bytes,
                     uint64_t *out_counts, uint64_t *out_offsets, uint8_t *out_bytes);
/* The same merge with every buffer resident in the HBM of `device` (the collecting rank of a multi-GPU gather holds
 * its own result and the ones RCCL delivered there): one merged result to bring down instead of `world`.
 * d_starts[r] = u64[num_entries[r]] entry starts (pss_device_result.d_offsets); d_out_offsets receives
 * sum(num_entries) + 1 entries, d_out_counts nq, d_out_bytes sum(num_bytes).  world <= 16. */
int pss_merge_packed_device(int32_t device, uint32_t world, uint64_t nq, const void *const *d_counts,
                            const void *const *d_starts, const void *const *d_bytes, const uint64_t *num_entries,
                            const uint64_t *num_bytes, void *d_out_counts, void *d_out_offsets, void *d_out_bytes);
int pss_reader_last_stats(const pss_reader *r, pss_search_stats *stats);
/* Drops the chunks and closes the reader. */
int pss_reader_close(pss_reader *r);

/* Result accessors; memory stays owned by the result until pss_result_free. */
uint64_t pss_result_num_queries(const pss_result *res);
uint64_t pss_result_num_entries(const pss_result *res);
const uint64_t *pss_result_query_counts(const pss_result *res);  /* [num_queries] entries per query */
const uint64_t *pss_result_offsets(const pss_result *res);       /* [num_entries + 1] into bytes */
const uint8_t *pss_result_bytes(const pss_result *res);
void pss_result_free(pss_result *res);

/* ---- synthetic corpora (used by bench.py and the tests) -- */

#define PSS_CORPUS_LINES 0     /* 38-symbol alphabet, '\n' with p = 1/40 */
#define PSS_CORPUS_WORDS 1     /* 65536-word vocabulary, skewed, realistic LCP */
#define PSS_CORPUS_RUNS 2      /* lines of one repeated symbol from {a,b}, run <= 8192 */
#define PSS_CORPUS_PERIODIC 3  /* "a"*4095 + "\n" repeated */
#define PSS_CORPUS_REPEAT_LINE 4 /* one 40-byte line repeated (period 40, no runs of equal bytes) */
#define PSS_CORPUS_DUP_BLOCKS 5  /* a 1 MiB block of `lines` text repeated, 16 single-byte edits per copy */
#define PSS_CORPUS_MIXED 6       /* `words` whose middle third is one 60-byte line repeated, then 64 KiB blocks copied from the first third */
#define PSS_CORPUS_SOURCE 7      /* source-like text: 200-odd byte values, indentation, licence headers, stretches of lines copied from earlier in the chunk (round 5) */
/* Fills out[0..n) on the host; deterministic in (kind, n, chunk_index). */
int pss_gen_corpus(int kind, uint8_t *out, uint64_t n, uint64_t chunk_index);

#ifdef __cplusplus
}
#endif
#endif /* PSS_H */
