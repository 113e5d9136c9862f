// sa_plan_impl.h -- the plan of a build: sizing the initial key from a sample, the switches of the builder (Knobs), timers.
// Included by sa_build.hip (inside namespace pss, after the alphabet kernels): one translation unit, split by route.

// ---- sizing the initial sort from a sample ---------------------------------
// The initial sort costs one pass per 8 key bits; every suffix it leaves tied
// costs about ten times a pass's per-element price in the rounds.  How many
// suffixes W key bits leave tied depends on the data, not only on the symbol
// frequencies (natural text repeats far more than i.i.d. symbols do), so it is
// measured: S stratified random suffixes, their full-width keys sorted, and for
// every W = 8 P the sample members that share their top W bits with a sorted
// neighbour counted.  A member collides inside the sample with probability
// (group size - 1) * S / n, hence tied fraction ~= count / S * n / S (an
// overestimate when groups are large -- the safe direction).
__global__ __launch_bounds__(256) void sample_keys_kernel(const u8 *codes, u32 n, u32 S, int b, int kmax, int plus_one,
                                                            u64 *keys, u32 *vals)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S) return;
    const u32 stride = n / S;
    u64 x = ((u64)t + 1) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32;
    const u32 pos = t * stride + (u32)(x % stride);
    const int bits = kmax * b;
    keys[t] = text_key_at(codes, pos, b, kmax, plus_one, n) << (64 - bits);
    vals[t] = t;
}

// tied[8]: sample members whose 48th successor still shares their top 20 key bits -- a joint bucket of the
// MSD path (msd_sort.hip) with >= 49 of the S sample members holds about 49 n / S suffixes, far beyond
// what a workgroup sorts in LDS: any such member rules that path out before it starts.
// tied[9]: distinct 20-bit prefixes in the sample; n / distinct estimates the average non-empty bucket
// (exact when there are far fewer buckets than sample members), and a path whose AVERAGE bucket is close
// to the tile limit will not pass the exact check either (`lines` at n = 2^30: 6 000 per bucket).
constexpr u32 kMsdScreenRun = 48;
__global__ __launch_bounds__(256) void sample_ties_kernel(const u64 *keys, u32 S, u32 *tied /* [8]: W = 8, 16, .. 64; [8] screen */)
{
    __shared__ u32 s_c[8];
    if (threadIdx.x < 8) s_c[threadIdx.x] = 0;
    __syncthreads();
    u32 c[8] = {};
    u32 crowded = 0, distinct = 0;
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < S; t += gridDim.x * blockDim.x) {
        const u64 k = keys[t];
        const u64 dp = t > 0 ? (keys[t - 1] ^ k) : ~0ull, dn = t + 1 < S ? (keys[t + 1] ^ k) : ~0ull;
        // equal top W bits with a neighbour <=> its xor has at least W leading zeros
        const int lz = max(dp ? __builtin_clzll(dp) : 64, dn ? __builtin_clzll(dn) : 64);
#pragma unroll
        for (int w = 0; w < 8; ++w) c[w] += lz >= 8 * (w + 1) ? 1u : 0u;
        if (t + kMsdScreenRun < S && ((keys[t + kMsdScreenRun] ^ k) >> 44) == 0) ++crowded;
        if (t == 0 || (dp >> 44) != 0) ++distinct;          // first sample member of its 20-bit prefix
    }
    if (crowded) atomicAdd(&tied[8], crowded);
    if (distinct) atomicAdd(&tied[9], distinct);
#pragma unroll
    for (int w = 0; w < 8; ++w)
        if (c[w]) atomicAdd(&s_c[w], c[w]);
    __syncthreads();
    if (threadIdx.x < 8 && s_c[threadIdx.x]) atomicAdd(&tied[threadIdx.x], s_c[threadIdx.x]);
}

// -------------------------------------------------------------------- host --

static void rerank_geometry(u32 m, RerankArgs &a)
{
    a.m = m;
    a.num_tiles = (u32)(((u64)m + RR_TILE - 1) / RR_TILE);
    a.tiles_per_range = (a.num_tiles + RR_MAX_RANGES - 1) / RR_MAX_RANGES;
    if (a.tiles_per_range == 0) a.tiles_per_range = 1;
    a.num_ranges = (a.num_tiles + a.tiles_per_range - 1) / a.tiles_per_range;
}

enum Slot { S_CODES = 0, S_K0, S_K1, S_V0, S_V1, S_ISA, S_P0, S_P1, S_GRP, S_WORK, S_GRP2 = 26, S_BIG = 27, S_SCR = 29, S_SSA = 30, S_SSB = 31, S_ANC = 32 /* .. 38: one per level */, S_X0 = 39 /* .. 45 */, S_SSPLAN = 48, S_PER = 49, S_BGT = 54, S_ANCW = 55 };      // (10 .. 23, 28: search.hip; 24, 25: capi.cpp)

// Initial key width.  Model the text as i.i.d. with per-symbol collision
// probability c = sum p_i^2 (from the sampled counts): two suffixes agree on k
// symbols with probability c^k, so about n * c^k of the suffixes stay tied.
// Pick the smallest k that leaves <= 1/4096 of them tied (the sparse path
// finishes those almost for free), then widen k to fill the last radix pass.
// A wrong guess costs speed only: whatever stays tied goes to the doubling rounds.
static int choose_key_chars(const u32 *counts, u32 n, int b, int kmax)
{
    double tot = 0, c = 0;
    for (int i = 0; i < 256; ++i) tot += counts[i];
    if (tot <= 0) return kmax;
    for (int i = 0; i < 256; ++i) {
        const double p = counts[i] / tot;
        c += p * p;
    }
    if (c >= 0.999999) return kmax;
    const double need = (12.0 + log2((double)n)) / -log2(c);
    int k = (int)ceil(need);
    if (k < 1) k = 1;
    if (k > kmax) k = kmax;
    const int passes = (k * b + 7) / 8;
    k = (passes * 8) / b;
    if (k > kmax) k = kmax;
    // Within one pass of the full 64 bits the model is not trusted to save that pass: real text
    // repeats far more than i.i.d. symbols do, and then every extra initial symbol pays
    // (measured on `words`: 12 symbols / 8 passes beats 11 / 7 by 4 %).
    if ((kmax * b + 7) / 8 - passes <= 1) k = kmax;
    return k;
}

// Environment switches of the builder (exploration and tests; read on every call so a test can
// flip them between builds).  None of them changes the result.
struct Knobs {
    int key_chars = 0;          // PSS_KEY_CHARS  force the symbols packed into the initial key (0 = choose)
    int key_drop = -1;          // PSS_KEY_DROP   force the low bits of the last symbol left out (-1 = choose)
    bool no_sample = false;     // PSS_NO_SAMPLE  size the initial key from symbol counts even for large n
    bool no_flags = false;      // PSS_NO_TIES_PASS  plain 8-byte-key passes + key comparison in the rerank
    int mode = -1;              // PSS_MODE       dense / sparse / text tie resolution (-1 = choose)
    int text_rounds_max = 5;    // PSS_TEXT_ROUNDS
    int msd = -1;               // PSS_MSD        0: never the MSD initial sort, 1: whenever the key fits, unset: screened
    bool no_plan = false;       // PSS_NO_PLAN_CACHE  always take the sizing sample (never reuse the previous build's choice of sort)
    bool no_front = false;      // PSS_NO_PLAN_FRONT  reuse the choice of sort, but not the alphabet (separate alphabet and recode passes)
    int ss = -1;                // PSS_SS         0: never the sample sort over 16-byte elements, 1: whenever the text has the size for it,
                                //                unset: n >= 2^24 and the MSD sort did not take the text
    bool no_msd_fuse = false;   // PSS_MSD_NO_FUSE  MSD sort flags ties in the suffix array; the rerank kernels read them
    bool no_mid_tier = false;   // PSS_NO_MID_TIER  groups above 512 members all take the chained radix sorts
    int big_merge = -1;         // PSS_BIG_MERGE  unset: by the average size of the large groups (sa_refine_impl.h); 0: never;
                                //                1: groups above 4096 members through the segmented merge sort (bg_*_kernel) instead of the
                                //                chained radix sorts, 2: in text rounds only.  Measured at 2^29 and left OFF: real files
                                //                111.8 / 112.7 vs 112.6 / 113.5 ms, `source` 166.6 vs 171.5, `mixed` 90.7 vs 87.2 (its
                                //                groups of millions take twelve merge passes where the radix sorts take seven)
    bool no_mid_merge = false;  // PSS_NO_MID_MERGE  groups of 513 .. 4096 members with a crowded bin take the chained sorts (no LDS merge sort)
    int period = -1;            // PSS_PERIOD     0: never the closed form for texts that repeat one word (rle_build.h)
    int rle = -1;               // PSS_RLE        0: never the run-length path, 1: always, unset: when runs average >= 8 bytes
    int anchor = -1;            // PSS_ANCHOR     0: never the anchor round for ties that outlive the text rounds (rank rounds over the
                                //                whole text instead), 1: whenever ties outlive them, unset: texts of >= 2^20 bytes
    int anchor_omega = 0;       // PSS_ANCHOR_OMEGA  force the window of the minimizers (0 = as wide as the known common prefix allows)
    bool no_probe = false;      // PSS_NO_PROBE   always a text round before the anchor round (no sampling of the ties)
    int anchor_min_omega = 11;  // PSS_ANCHOR_MIN_OMEGA  narrowest window the anchor round accepts by itself
    int probe_skip_pct = 50;    // PSS_PROBE_SKIP_PCT  no text rounds when more than this share of the sampled tied pairs are repeats
    int side = -1;              // PSS_ANCHOR_SIDE  0: never sort the anchors beside the text round, 1: whenever a text round precedes the
                                //                anchor round, unset: texts of >= 2^24 bytes whose sampled ties show copies
    int side_pct = 8;           // PSS_ANCHOR_SIDE_PCT  ... at least this share of the sampled tied pairs
    int anchor_cap_div = 5;     // PSS_ANCHOR_CAP_DIV  the anchor round declines when the windows choose more than n / this many anchors
    bool count_sort = false;    // PSS_COUNT_SORT  rank rounds: groups ranked by counting (group_sort_kernel) instead of the merge sort
    bool no_periodic = false;   // PSS_PERIODIC=0  rank rounds: no periodic keys for the large groups (per_*_kernel)
    bool timing = false;        // PSS_TIMING     per-round trace on stderr
    static Knobs read()
    {
        Knobs k;
        if (const char *e = knob("PSS_KEY_CHARS")) k.key_chars = atoi(e);
        if (const char *e = knob("PSS_KEY_DROP")) k.key_drop = atoi(e);
        k.no_sample = knob("PSS_NO_SAMPLE") != nullptr;
        k.no_flags = knob("PSS_NO_TIES_PASS") != nullptr;
        if (const char *e = knob("PSS_MODE")) {
            if (!strcmp(e, "dense")) k.mode = 0;
            else if (!strcmp(e, "sparse")) k.mode = 1;
            else if (!strcmp(e, "text")) k.mode = 2;
        }
        if (const char *e = knob("PSS_TEXT_ROUNDS")) k.text_rounds_max = atoi(e);
        if (const char *e = knob("PSS_MSD")) k.msd = atoi(e);
        if (const char *e = knob("PSS_SS")) k.ss = atoi(e);
        k.no_plan = knob("PSS_NO_PLAN_CACHE") != nullptr;
        k.no_front = knob("PSS_NO_PLAN_FRONT") != nullptr;
        k.no_msd_fuse = knob("PSS_MSD_NO_FUSE") != nullptr;
        k.no_mid_tier = knob("PSS_NO_MID_TIER") != nullptr;
        k.no_mid_merge = knob("PSS_NO_MID_MERGE") != nullptr;
        if (const char *e = knob("PSS_BIG_MERGE")) k.big_merge = atoi(e);
        if (knob("PSS_NO_BIG_MERGE")) k.big_merge = 0;
        if (const char *e = knob("PSS_RLE")) k.rle = atoi(e);
        if (const char *e = knob("PSS_PERIOD")) k.period = atoi(e);
        if (const char *e = knob("PSS_ANCHOR")) k.anchor = atoi(e);
        if (const char *e = knob("PSS_ANCHOR_OMEGA")) k.anchor_omega = atoi(e);
        k.no_probe = knob("PSS_NO_PROBE") != nullptr;
        if (const char *e = knob("PSS_PROBE_SKIP_PCT")) k.probe_skip_pct = atoi(e);
        if (const char *e = knob("PSS_ANCHOR_MIN_OMEGA")) k.anchor_min_omega = std::max(2, atoi(e));
        if (const char *e = knob("PSS_ANCHOR_SIDE")) k.side = atoi(e);
        if (const char *e = knob("PSS_ANCHOR_SIDE_PCT")) k.side_pct = atoi(e);
        if (const char *e = knob("PSS_ANCHOR_CAP_DIV")) k.anchor_cap_div = std::min(5, std::max(3, atoi(e)));
        k.count_sort = knob("PSS_COUNT_SORT") != nullptr;
        { const char *e = knob("PSS_PERIODIC"); k.no_periodic = e && atoi(e) == 0; }
        k.timing = knob("PSS_TIMING") != nullptr;
        return k;
    }
};

// start / stop events of one build, destroyed on every exit path
struct BuildTimer {
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mid = nullptr;
    ~BuildTimer()
    {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (ev_mid) (void)hipEventDestroy(ev_mid);
    }
};

// Sizing of the initial sort from a sorted sample (see sample_keys_kernel): the fewest passes
// that leave <= 2 % of the suffixes tied, else the full kmax symbols.  K / V are free scratch.
static int size_initial_key(DeviceCtx *ctx, const u8 *codes, u32 n, int b, int kmax, int plus_one, u64 *K[2], u32 *V[2],
                            void *work, u32 *d_tied, u32 *h_small, bool profile, SortStats *ss, int *key_chars,
                            int *key_drop, bool *msd_screen_ok)
{
    hipStream_t s = ctx->stream;
    const u32 S = 1u << 21;
    const int bits_max = kmax * b, pmax = (bits_max + 7) / 8;
    PSS_HIP(hipMemsetAsync(d_tied, 0, 64, s));
    hipLaunchKernelGGL(sample_keys_kernel, dim3(S / 256), dim3(256), 0, s, codes, n, S, b, kmax, plus_one, K[0], V[0]);
    u32 mask = 0;
    for (int p = 0; p < 8; ++p)
        if (8 * (p + 1) > 64 - 8 * (pmax - 1)) mask |= 1u << p;   // only the top 8 (pmax - 1) bits are ever compared
    int sd = 0;
    const u64 launches = ss->launches, elems = ss->elems;
    PSS_TRY(radix_sort_pairs(ctx, K, V, S, 64, mask, nullptr, 0, work, &sd, profile, ss));
    ss->launches = launches;   // not passes of the suffix sort (their profile figures stay in: same kernel, same stream)
    ss->elems = elems;
    hipLaunchKernelGGL(sample_ties_kernel, dim3(256), dim3(256), 0, s, K[sd], S, d_tied);
    PSS_HIP(hipMemcpyAsync(h_small, d_tied, 64, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipStreamSynchronize(s));
    *msd_screen_ok = h_small[8] == 0 && h_small[9] != 0 && (double)n / (double)h_small[9] <= 3400.0;
    *key_chars = kmax;
    *key_drop = 0;
    for (int p = 2; p < pmax; ++p) {
        const double est = (double)h_small[p - 1] / S * ((double)n / S);
        if (est <= 0.02) {
            *key_chars = (8 * p + b - 1) / b;
            *key_drop = *key_chars * b - 8 * p;
            break;
        }
    }
    return PSS_OK;
}
