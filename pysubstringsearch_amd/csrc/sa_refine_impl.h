// sa_refine_impl.h -- host side of everything after the initial sort: the rounds (refine_rounds), the anchor round's driver (anchor_rank_keys) and its side line.
// Included by sa_build.hip (inside namespace pss, after the alphabet kernels): one translation unit, split by route.

// Everything after the initial sort: rerank + compaction of the tied suffixes, then rounds until no
// ties are left.  Its own function since round 2: the run-length path (rle_build below) sorts the
// suffixes of an INTEGER string (one symbol per run of the text) with the same rounds -- there is no
// text to pack keys from then (codes == nullptr): rank rounds only, starting from h0 = 1.
struct RoundsIO {
    u32 n;
    u32 *SA;
    u64 *K[2];
    u32 *V[2];
    u32 *ISA;
    u32 *P[2];
    u32 *GRP;
    const u8 *codes;            // recoded text (nullptr: rank rounds only)
    int b, plus_one, key_chars, key_drop;
    u64 h0;                     // symbols every group of the initial sort is known to share
    int cur;                    // K[cur] / V[cur]: sorted keys / suffixes of the initial sort
    int final_buf;              // V[final_buf] was redirected to SA for the initial sort (-1: not)
    u32 *v_scratch;             // ... and this is the buffer it stands for
    bool ties;                  // V[cur] carries tie flags in bit 31 (no keys)
    bool msd_fused;             // the MSD sort already produced the first active list
    u32 msd_active;
    bool no_sparse;             // the initial key does not fit 64 bits (sample sort): the sparse mode's key search cannot be used
    u8 *work;
    u32 *d_agg_head, *d_agg_cnt;
    u64 *d_red;
    u32 *d_counters;
    u32 *h_small;
    bool profile;
    // Subset sort (anchors, anchor_impl.h): the n elements are the suffixes at text positions sub_pos[0 .. n) of a text of
    // text_n symbols; element values are the ordinals.  Text rounds run until every group shares stop_text_h symbols,
    // whatever they resolve; then the group ranks are the symbols of an integer string (element v is followed by v + 1)
    // whose suffixes the rank rounds sort, starting over from h = 1.
    const u32 *sub_pos = nullptr;
    u32 text_n = 0;
    u64 stop_text_h = 0;
    u32 *grp2 = nullptr;        // second group-rank buffer (nullptr: slot S_GRP2 of the context)
    int level = 0;              // 0: the text (or the run-length path's reduced string); k: the names of level k - 1's anchors
    struct SideAnchors *side = nullptr;      // the side line of the build these rounds belong to (it may be under way already)
};

static int anchor_rank_keys(DeviceCtx *ctx, const Knobs &knobs, const RoundsIO &outer, u64 h, const u32 *syms,
                            const u32 *cur_ranks, u32 *akey, pss_sa_stats &st, bool *ok);

// The anchors' own sort BESIDE the text round (round 5).  Which positions are anchors, their names and the order of the
// anchor suffixes depend on the text and on the window only -- not on the active list the text round is busy with.  On
// text with copies in it (source code: 412 MB of real files spent 34 ms in the text round and 50 ms in the anchors'
// sort, one after the other) the anchors are therefore sorted by a second host thread on a second stream, in a context
// of its own (DeviceCtx::helper: its stream, pinned scratch and slots), while the main line runs the text round that
// raises the depth to what the window needs.  Half of the anchors' sort is a long row of small launches (rank rounds over
// lists of 10^4 .. 10^6 elements on three levels) that leave the device all but empty: they fill the gaps of the other
// stream instead of standing in line.  The window is fixed in advance (the depth the text round WILL reach); a text round
// that gives up half-way leaves the depth where it was, and the result of the side line is then thrown away.
struct SideAnchors {
    std::thread th;
    bool started = false, joined = false, ok = false;
    int rc = PSS_OK;
    std::string err;
    u64 h_eff = 0;
    pss_sa_stats st;
    u32 *akey = nullptr;
    void join()
    {
        if (started && !joined) {
            if (th.joinable()) th.join();
            joined = true;
        }
    }
    ~SideAnchors() { join(); }
};

// Starts the side line: anchors of the text `codes` for windows that fit depth h_eff, sorted in ctx's helper context.
// after (optional): an event on the main stream that the helper's stream waits for first (the codes are being written).
// No room for the buffers is not an error: the anchors then wait their turn on the main line as before.
static int side_start(DeviceCtx *ctx, const Knobs &knobs, SideAnchors &side, u32 n, const u8 *codes, int b, int plus_one, u64 h_eff,
                      hipEvent_t after)
{
    if (side.started) return PSS_OK;
    DeviceCtx *hc = nullptr;
    const size_t sort_ws = radix_sort_workspace_bytes();
    int rs = get_helper_ctx(ctx, &hc);
    if (rs == PSS_OK) rs = hc->slot[S_K0].reserve((size_t)n * 8);
    if (rs == PSS_OK) rs = hc->slot[S_K1].reserve((size_t)n * 8);
    if (rs == PSS_OK) rs = hc->slot[S_ISA].reserve((size_t)n * 4 + 64);
    if (rs == PSS_OK) rs = hc->slot[S_WORK].reserve(sort_ws + 65536);
    if (rs != PSS_OK) {
        (void)hipGetLastError();
        set_error("%s", "");
        return PSS_OK;
    }
    if (after) PSS_HIP(hipStreamWaitEvent(hc->stream, after, 0));
    u8 *hw = hc->slot[S_WORK].as<u8>();
    u8 *hsmall = hw + sort_ws;
    RoundsIO o2;
    memset(&o2, 0, sizeof o2);
    o2.n = n;
    o2.K[0] = hc->slot[S_K0].as<u64>();
    o2.K[1] = hc->slot[S_K1].as<u64>();
    o2.codes = codes;
    o2.b = b;
    o2.plus_one = plus_one;
    o2.work = hw;
    o2.d_agg_head = reinterpret_cast<u32 *>(hsmall + 4096);
    o2.d_agg_cnt = reinterpret_cast<u32 *>(hsmall + 8192);
    o2.d_red = reinterpret_cast<u64 *>(hsmall + 12288);
    o2.d_counters = reinterpret_cast<u32 *>(hsmall + 12288 + 64);
    o2.h_small = static_cast<u32 *>(hc->pinned);
    o2.level = 0;
    side.h_eff = h_eff;
    side.akey = hc->slot[S_ISA].as<u32>();
    memset(&side.st, 0, sizeof side.st);
    const int dev = ctx->device;
    SideAnchors *sp = &side;
    // (std::thread's constructor may throw -- no more threads to be had: the flag goes up only once the thread exists, so
    // that the destructor never joins what was never started; the exception travels to the C ABI's catch-all)
    side.th = std::thread([hc, knobs, o2, h_eff, sp, dev]() {
        if (hipSetDevice(dev) != hipSuccess) {
            sp->rc = PSS_EDEVICE;
            sp->err = "hipSetDevice failed in the anchors' side line";
            return;
        }
        try {
            sp->rc = anchor_rank_keys(hc, knobs, o2, h_eff, nullptr, nullptr, sp->akey, sp->st, &sp->ok);
            if (sp->rc != PSS_OK) sp->err = last_error();
        } catch (const std::bad_alloc &) {
            sp->rc = PSS_ENOMEM;
            sp->err = "host allocation failed in the anchors' side line";
        } catch (...) {                      // (nothing may leave a thread's function)
            sp->rc = PSS_EDEVICE;
            sp->err = "internal error in the anchors' side line";
        }
    });
    side.started = true;
    return PSS_OK;
}

static int refine_rounds(DeviceCtx *ctx, const Knobs &knobs, RoundsIO &io, SortStats &ss, pss_sa_stats &st)
{
    hipStream_t s = ctx->stream;
    const u32 n = io.n;
    u32 *SA = io.SA;
    u64 *K[2] = {io.K[0], io.K[1]};
    u32 *V[2] = {io.V[0], io.V[1]};
    u32 *ISA = io.ISA;
    u32 *P[2] = {io.P[0], io.P[1]};
    u32 *GRP = io.GRP;
    const u8 *codes = io.codes;
    const bool rank_only = codes == nullptr;
    const bool subset = io.sub_pos != nullptr;
    const u32 text_n = subset ? io.text_n : n;
    bool anchored = false;       // the anchor round has run: nothing may be left tied
    const int b = io.b, plus_one = io.plus_one, key_chars = io.key_chars;
    const bool profile = io.profile;
    u8 *work = io.work;
    u32 *d_agg_head = io.d_agg_head, *d_agg_cnt = io.d_agg_cnt, *d_counters = io.d_counters, *h_small = io.h_small;
    u64 *d_red = io.d_red;
    const int grid_stream = ctx->num_cus * 8;
    int cur = io.cur;
    const int final_buf = io.final_buf;
    u32 *const v_scratch = io.v_scratch;
    const bool ties = io.ties, msd_fused = io.msd_fused;
    const u32 msd_active = io.msd_active;
    const bool sa_in_place = (final_buf >= 0 && cur == final_buf);

    int rank_bits = 1;
    while ((1ull << rank_bits) <= (u64)n) ++rank_bits;      // ranks 0..n
    RerankArgs ra;
    memset(&ra, 0, sizeof ra);
    ra.agg_head = d_agg_head;
    ra.agg_cnt = d_agg_cnt;
    ra.SA = SA;
    ra.ISA = ISA;
    ra.counters = d_counters;
    ra.ht = reinterpret_cast<u64 *>(ISA);     // the two modes never coexist
    ra.rank_bits = rank_bits;
    u32 m = n;
    int pcur = 0;                // P[pcur] holds the SA positions of the active list (after round 0)
    int gcur = 0;                // G[gcur] holds its group ranks
    u32 *G[2] = {GRP, nullptr};
    bool identity_pos = true;
    enum Mode { M_DENSE = 0, M_SPARSE = 1, M_TEXT = 2 };
    Mode mode = M_DENSE;
    bool was_text = false;
    int text_rounds = 0;
    u32 m_text_prev = 0;
    bool keyed_grp = true;       // the sorted keys of the previous round carry the group rank in their high half
    u32 global_above = 0;        // rank rounds use ONE global (group, rank) sort while m stays above this
    double last_big_frac = 0.0;
    const int text_rounds_max = knobs.text_rounds_max;
    int kt = 64 / b;             // symbols per text-round key
    if (kt > 16) kt = 16;
    const int k0buf = cur;       // K[k0buf] = sorted initial keys (kept intact in sparse mode)
    u64 *SK[2] = {nullptr, nullptr};   // sparse mode: small ping-pong key buffers
    u64 **Kr = K;
    u64 h = io.h0;               // symbols every group is known to share
    const u32 grid_all = (u32)grid_stream;
    // Integer strings (rank rounds only): the ranks the first round leaves ARE the string, up to renaming -- kept for the
    // minimizers of an anchor level on top of this one (anchor_impl.h), should the rounds reach depth 32 with much left tied.
    SideAnchors own_side;
    SideAnchors &side = io.side ? *io.side : own_side;
    u32 *X0 = nullptr;
    const u32 *Xsym = nullptr;   // the same snapshot for the periodic keys of the rank rounds (per_*_kernel), never given up
    bool last_per = false;       // the last rank round met periodic runs among its large groups, or chains that will be
    u64 per_wait_h = 0;          // ... and the depth from which their period can be seen
    auto snapshot_symbols = [&]() -> int {
        const bool for_levels = !(knobs.anchor == 0 || io.level >= 6 || n < (knobs.anchor == 1 ? 64u : (1u << 20)));
        if (!for_levels && (knobs.no_periodic || io.level >= 6 || n <= 3072u)) return PSS_OK;
        PSS_TRY(ctx->slot[S_X0 + io.level].reserve((size_t)n * 4));
        u32 *snap = ctx->slot[S_X0 + io.level].as<u32>();
        PSS_HIP(hipMemcpyAsync(snap, ISA, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        if (for_levels) X0 = snap;
        Xsym = snap;
        return PSS_OK;
    };
    for (int round = 0;; ++round) {
        if (round > 96) {
            set_error("sa_build: no convergence after 96 rounds (internal error)");
            return PSS_EDEVICE;
        }
        rerank_geometry(m, ra);
        ra.keys = Kr[cur];
        ra.idx = V[cur];
        ra.SA = (round == 0 && sa_in_place) ? nullptr : SA;      // round 0: the sort already wrote SA
        ra.tied_sa = (round == 0 && ties && !msd_fused) ? V[cur] : nullptr;
        ra.pos = identity_pos ? nullptr : P[pcur];
        ra.grp = (round > 0 && !keyed_grp) ? G[gcur] : nullptr;   // group-local rounds: keys do not carry the group
        ra.pos_out = P[pcur ^ 1];
        ra.idx_out = V[cur ^ 1];
        if (round == 0) {
            if (io.grp2) G[1] = io.grp2;
            else {
                PSS_TRY(ctx->slot[S_GRP2].reserve((size_t)n * 4));
                G[1] = ctx->slot[S_GRP2].as<u32>();
            }
        }
        ra.grp_out = G[gcur ^ 1];
        const bool fused0 = round == 0 && msd_fused;     // the MSD local sort already produced this round's active list
        if (!fused0) {
            if (ra.tied_sa) hipLaunchKernelGGL(rr_reduce_tied_kernel, dim3(ra.num_ranges), dim3(RR_BLOCK), 0, s, ra);
            else hipLaunchKernelGGL(rr_reduce_kernel, dim3(ra.num_ranges), dim3(RR_BLOCK), 0, s, ra);
            hipLaunchKernelGGL(rr_scan_kernel, dim3(1), dim3(1024), 0, s, d_agg_head, d_agg_cnt, ra.num_ranges, d_counters);
            PSS_HIP(hipMemcpyAsync(h_small, d_counters, 4, hipMemcpyDeviceToHost, s));
            PSS_HIP(hipStreamSynchronize(s));
        }
        const u32 m_next = fused0 ? msd_active : h_small[0];
        if (round == 0) {
            // few ties: sparse (hash + key search); otherwise extend the ties from the text first
            mode = ((u64)m_next * 1024 <= (u64)n) ? M_SPARSE : M_TEXT;
            if (knobs.mode >= 0) mode = (Mode)knobs.mode;
            if (mode == M_SPARSE && (u64)m_next * 16 > (u64)n) mode = M_TEXT;   // hash table must fit the ISA buffer
            if (mode == M_SPARSE && io.no_sparse && m_next) mode = M_TEXT;
            if (m_next == 0) mode = M_SPARSE;                                   // nothing left: no ISA at all
            if (mode == M_TEXT && text_rounds_max <= 0) mode = M_DENSE;
            if (rank_only && m_next) mode = M_DENSE;                            // no text to pack keys from
            if (subset && m_next) mode = h < io.stop_text_h ? M_TEXT : M_DENSE;
            was_text = mode == M_TEXT;
        }
        if (fused0) {
            if (mode == M_DENSE && m_next) {      // rank rounds from the start: they need the inverse suffix array
                hipLaunchKernelGGL(isa_from_sa_kernel, dim3(grid_all), dim3(256), 0, s, SA, n, ISA);
                hipLaunchKernelGGL(isa_active_kernel, dim3(grid_all), dim3(256), 0, s, ra.idx_out, ra.grp_out, m_next, ISA);
            }
        } else if (mode == M_DENSE) hipLaunchKernelGGL(rr_apply_kernel<MODE_ISA>, dim3(ra.num_ranges), dim3(RR_BLOCK), 0, s, ra);
        else if (mode == M_SPARSE && round > 0) hipLaunchKernelGGL(rr_apply_kernel<MODE_HT>, dim3(ra.num_ranges), dim3(RR_BLOCK), 0, s, ra);
        else if (ra.tied_sa && ra.SA == nullptr && ra.pos == nullptr)
            hipLaunchKernelGGL(rr_apply_tied_kernel, dim3(ra.num_ranges), dim3(RR_BLOCK), 0, s, ra);
        else hipLaunchKernelGGL(rr_apply_kernel<MODE_NONE>, dim3(ra.num_ranges), dim3(RR_BLOCK), 0, s, ra);
        PSS_HIP(hipGetLastError());
        if (round == 0 && final_buf >= 0) V[final_buf] = v_scratch;   // later rounds must not scribble over SA
        if (m_next == 0) break;
        if (anchored && mode == M_DENSE) {
            // cannot happen: the anchor round leaves no ties (and its keys took the place of the inverse array)
            st.anchor_left += m_next;
            set_error("sa_build: %u elements tied after the anchor round of level %d (internal error)", m_next, io.level);
            return PSS_EDEVICE;
        }
        if (round == 0 && mode == M_DENSE && subset) h = 1;           // (no text round was needed: the elements are symbols already)
        if (round == 0 && mode == M_DENSE && (subset || rank_only)) PSS_TRY(snapshot_symbols());
        // (subset mode counts h in SYMBOLS of the text while its text rounds run, in elements afterwards)
        if (h >= (subset && mode == M_TEXT ? (u64)text_n : (u64)n)) {
            set_error("sa_build: %u suffixes unresolved at h=%llu >= n (internal error)", m_next,
                      (unsigned long long)h);
            return PSS_EDEVICE;
        }
        m = m_next;
        pcur ^= 1;
        gcur ^= 1;
        identity_pos = false;
        const int src = cur ^ 1;             // V[src] = compacted suffix indices, G[gcur] their groups, P[pcur] their slots
        const u32 grid = (u32)std::min<u64>((u64)grid_stream, ((u64)m + 255) / 256);

        // ------------------------------------------------------ group-local round --
        // One round over the active list: a 64-bit key per suffix, every group sorted by it.
        //   text round : key = next symbols packed from the text at offset h      (h += kt)
        //   rank round : key = rank of suffix i+h from the inverse suffix array    (h *= 2)
        // Groups of <= GS_CAP members are ranked in LDS (group_sort); members of larger groups
        // go through two chained stable radix sorts (key, then dense group number).
        // Text rounds advance h linearly; they pay off while each round resolves most ties
        // (natural-language LCPs).  When a round leaves more than 60 % of its list tied, or large
        // groups dominate, the data is repetitive: rank rounds, logarithmic in the LCP, take over.
        auto local_round = [&](bool use_text, bool *bail, const u32 *key_of_suffix = nullptr) -> int {
            *bail = false;
            const u32 nblk = (m + GS_T - 1) / GS_T;
            PSS_TRY(ctx->slot[S_SCR].reserve((size_t)m + (size_t)nblk * 24 + (SC_MAX_BLOCKS + 8) * 8 + 4096 +
                                             ((size_t)m / 2 + 16) * sizeof(MidGroup) + ((size_t)m / 512 + 16) * sizeof(MidGroup) + 512));
            u8 *scr = ctx->slot[S_SCR].as<u8>();
            size_t o = 0;
            auto carve = [&](size_t bytes) { u8 *p = scr + o; o = round_up(o + bytes, 64); return p; };
            u8 *d_big = carve(m);
            u32 *d_blk_big = reinterpret_cast<u32 *>(carve((size_t)nblk * 4));
            u32 *d_blk_heads = reinterpret_cast<u32 *>(carve((size_t)nblk * 4));
            u64 *d_off_big = reinterpret_cast<u64 *>(carve(((size_t)nblk + 1) * 8));
            u64 *d_off_heads = reinterpret_cast<u64 *>(carve(((size_t)nblk + 1) * 8));
            u64 *d_partial = reinterpret_cast<u64 *>(carve((SC_MAX_BLOCKS + 2) * 8));
            u64 *d_total = d_partial + SC_MAX_BLOCKS;
            const u32 h32 = (u32)std::min<u64>(h, 0xffffffffull);
            if (use_text)
                hipLaunchKernelGGL(text_keys_kernel, dim3(grid), dim3(256), 0, s, V[src], m, text_n, h32, codes, b, kt, plus_one,
                                   K[src], io.sub_pos);
            else if (key_of_suffix)      // anchor round: the key of suffix i is key_of_suffix[i] (the rank of the anchor its window chose)
                hipLaunchKernelGGL(rank_keys_kernel, dim3(grid), dim3(256), 0, s, V[src], m, n, 0u, key_of_suffix, K[src]);
            else
                hipLaunchKernelGGL(rank_keys_kernel, dim3(grid), dim3(256), 0, s, V[src], m, n, h32, ISA, K[src]);
            if (use_text)
                hipLaunchKernelGGL(group_sort_kernel<false>, dim3(nblk), dim3(256), 0, s, K[src], V[src], G[gcur], m, K[src ^ 1],
                                   V[src ^ 1], d_big, d_blk_big, d_blk_heads);
            else if (knobs.count_sort)
                hipLaunchKernelGGL(group_sort_kernel<true>, dim3(nblk), dim3(256), 0, s, K[src], V[src], G[gcur], m, K[src ^ 1],
                                   V[src ^ 1], d_big, d_blk_big, d_blk_heads);
            else
                hipLaunchKernelGGL(group_msort32_kernel, dim3(nblk), dim3(GM_BLOCK), 0, s, K[src], V[src], G[gcur], m, K[src ^ 1],
                                   V[src ^ 1], d_big, d_blk_big, d_blk_heads);
            if (!knobs.no_mid_tier) {
                // groups of up to MID_CAP members: one workgroup each, in LDS (no host round trip: the list and its
                // length stay on the device, the workgroups persist and walk over it)
                MidGroup *d_mid = reinterpret_cast<MidGroup *>(carve(((size_t)m / 2 + 16) * sizeof(MidGroup)));
                MidGroup *d_mid_fail = reinterpret_cast<MidGroup *>(carve(((size_t)m / 512 + 16) * sizeof(MidGroup)));   // (groups of > 512 members)
                u32 *d_mid_count = reinterpret_cast<u32 *>(carve(64));
                PSS_HIP(hipMemsetAsync(d_mid_count, 0, 8, s));
                hipLaunchKernelGGL(mid_collect_kernel, dim3(grid), dim3(256), 0, s, d_big, G[gcur], m, d_mid, d_mid_count);
                hipLaunchKernelGGL(mid_sort_kernel, dim3((u32)ctx->num_cus * 4), dim3(MID_BLOCK), 0, s, K[src], V[src], d_mid,
                                   (const u32 *)d_mid_count, use_text ? kt * b : rank_bits, K[src ^ 1], V[src ^ 1], d_big, d_blk_big,
                                   d_blk_heads, knobs.no_mid_merge ? (MidGroup *)nullptr : d_mid_fail, d_mid_count + 1);
                if (!knobs.no_mid_merge)
                    hipLaunchKernelGGL(mid_msort_kernel, dim3((u32)ctx->num_cus * 3), dim3(MID_BLOCK), 0, s, K[src], V[src], d_mid_fail,
                                       (const u32 *)(d_mid_count + 1), K[src ^ 1], V[src ^ 1], d_big, d_blk_big, d_blk_heads);
            }
            PSS_TRY(device_excl_scan(ctx, InU32{d_blk_big}, nblk, d_partial, d_total, d_off_big));
            PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
            PSS_HIP(hipStreamSynchronize(s));
            const u32 nbig = h_small[0];
            if (knobs.timing)
                fprintf(stderr, "[pss] %s round: h=%llu m=%u large-group members=%u (%.1f%%)\n", use_text ? "text" : "rank",
                        (unsigned long long)h, m, nbig, 100.0 * nbig / m);
            last_big_frac = (double)nbig / (double)m;
            if (nbig == 0) return PSS_OK;
            if (use_text && !subset && (u64)nbig * 2 > (u64)m && (text_rounds > 0 || (u64)nbig * 4 > (u64)m * 3)) {
                *bail = true;
                return PSS_OK;
            }
            PSS_TRY(ctx->slot[S_BIG].reserve((size_t)nbig * (4 + 4 + 16 + 8) + 1024));
            u8 *bscr = ctx->slot[S_BIG].as<u8>();
            size_t bo = 0;
            auto bcarve = [&](size_t bytes) { u8 *p = bscr + bo; bo = round_up(bo + bytes, 64); return p; };
            u32 *d_bt = reinterpret_cast<u32 *>(bcarve((size_t)nbig * 4));
            u32 *d_bgid = reinterpret_cast<u32 *>(bcarve((size_t)nbig * 4));
            u64 *BK[2] = {reinterpret_cast<u64 *>(bcarve((size_t)nbig * 8)), reinterpret_cast<u64 *>(bcarve((size_t)nbig * 8))};
            u32 *BV[2] = {reinterpret_cast<u32 *>(bcarve((size_t)nbig * 4)), reinterpret_cast<u32 *>(bcarve((size_t)nbig * 4))};
            PSS_TRY(device_excl_scan(ctx, InU32{d_blk_heads}, nblk, d_partial, d_total, d_off_heads));
            PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
            hipLaunchKernelGGL(big_compact_kernel, dim3(nblk), dim3(256), 0, s, d_big, G[gcur], K[src], m, d_off_big,
                               d_off_heads, d_bt, BK[0], d_bgid);
            PSS_HIP(hipStreamSynchronize(s));
            const u32 nbig_groups = h_small[0];
            int gid_bits = 1;
            while ((1ull << gid_bits) < (u64)nbig_groups) ++gid_bits;
            int big_key_bits = use_text ? kt * b : rank_bits;
            last_per = false;
            const bool per_syms = (subset || rank_only) ? Xsym != nullptr : codes != nullptr;
            if (per_wait_h > h && nbig >= 2) last_per = true;      // (chains seen, their period still ahead of the depth)
            if (!use_text && !key_of_suffix && !knobs.no_periodic && per_syms && nbig >= 2 && h < (1ull << 31) && h >= per_wait_h) {
                // periodic runs among the large groups: their members get the key that orders them at once (per_*_kernel)
                int idx_bits = 1;
                while ((1ull << idx_bits) < (u64)n) ++idx_bits;
                const size_t g4 = round_up((size_t)nbig_groups * 4, 64), e4 = round_up((size_t)nbig * 4, 64), e8 = round_up((size_t)nbig * 8 + 8, 64);
                if (ctx->slot[S_PER].reserve(e8 + e4 + e4 + e4 + e8 + e4 + e4 + e8 + 5 * g4 + 256) == PSS_OK) {
                    u8 *pb = ctx->slot[S_PER].as<u8>();
                    size_t po = 0;
                    auto pcarve = [&](size_t bytes) { u8 *q = pb + po; po += bytes; return q; };
                    u64 *PK[2] = {reinterpret_cast<u64 *>(pcarve(e8)), BK[1]};
                    u32 *PV[2] = {reinterpret_cast<u32 *>(pcarve(e4)), BV[1]};
                    u32 *d_step = reinterpret_cast<u32 *>(pcarve(e4)), *d_flag = reinterpret_cast<u32 *>(pcarve(e4));
                    u64 *d_c = reinterpret_cast<u64 *>(pcarve(e8));
                    u32 *d_epos = reinterpret_cast<u32 *>(pcarve(e4)), *d_eell = reinterpret_cast<u32 *>(pcarve(e4));
                    u64 *d_etail = reinterpret_cast<u64 *>(pcarve(e8));
                    u32 *d_pmin = reinterpret_cast<u32 *>(pcarve(g4));
                    u32 *d_gsize = reinterpret_cast<u32 *>(pcarve(4 * g4 + 256));        // gsize, links, bad, pg, out: zeroed together
                    u32 *d_links = d_gsize + g4 / 4, *d_bad = d_links + g4 / 4, *d_pg = d_bad + g4 / 4, *d_out = d_pg + g4 / 4;
                    const u32 pgrid = std::min<u32>((nbig + 255) / 256, (u32)ctx->num_cus * 16);
                    const u32 wgrid = std::min<u32>((nbig + 255) / 256, (u32)ctx->num_cus * 4);      // (waves that walk contiguous pieces)
                    PSS_HIP(hipMemsetAsync(d_pmin, 0xff, g4, s));
                    PSS_HIP(hipMemsetAsync(d_gsize, 0, 4 * g4 + 256, s));
                    PSS_HIP(hipMemsetAsync(d_out + 2, 0xff, 4, s));
                    hipLaunchKernelGGL(per_pack_kernel, dim3(pgrid), dim3(256), 0, s, d_bt, d_bgid, V[src], nbig, idx_bits, PK[0], PV[0]);
                    int dp = 0;
                    SortStats sp;
                    PSS_TRY(radix_sort_pairs(ctx, PK, PV, nbig, gid_bits + idx_bits, 0xffffffffu, nullptr, 0, work, &dp, false, &sp));
                    hipLaunchKernelGGL(per_steps_kernel, dim3(wgrid), dim3(256), 0, s, PK[dp], nbig, idx_bits, h32, d_step, d_pmin, d_gsize);
                    hipLaunchKernelGGL(per_check_kernel, dim3(wgrid), dim3(256), 0, s, PK[dp], d_step, nbig, idx_bits, h32, d_pmin,
                                       d_links, d_bad, d_out);
                    hipLaunchKernelGGL(per_decide_kernel, dim3((nbig_groups + 255) / 256), dim3(256), 0, s, d_pmin, d_gsize, d_links,
                                       d_bad, nbig_groups, d_pg, d_out);
                    PSS_HIP(hipMemcpyAsync(h_small, d_out, 12, hipMemcpyDeviceToHost, s));
                    PSS_HIP(hipStreamSynchronize(s));
                    const u32 per_members = h_small[0], arith = h_small[1];
                    // most of the list in chains whose step the depth has not reached: nothing to find before it has
                    per_wait_h = (per_members == 0 && (u64)arith * 2 >= (u64)nbig && h_small[2] != 0xffffffffu) ? h_small[2] : 0;
                    if (knobs.timing)
                        fprintf(stderr, "[pss] rank round: h=%llu large-group members=%u in periodic groups=%u, equal steps=%u\n",
                                (unsigned long long)h, nbig, per_members, arith);
                    last_per = per_members != 0 || (u64)arith * 4 >= (u64)nbig;
                    if (per_members) {
                        const PerSyms y{(subset || rank_only) ? Xsym : nullptr, (subset || rank_only) ? nullptr : codes, n};
                        hipLaunchKernelGGL(per_flag_kernel, dim3(pgrid), dim3(256), 0, s, PK[dp], d_step, nbig, idx_bits, d_pg, d_flag);
                        PSS_TRY(device_excl_scan(ctx, InU32{d_flag}, nbig, d_partial, d_total, d_c));
                        hipLaunchKernelGGL(per_ends_kernel, dim3(pgrid), dim3(256), 0, s, PK[dp], d_flag, d_c, nbig, idx_bits, d_pg, h32,
                                           y, ISA, d_epos, d_eell, d_etail);
                        hipLaunchKernelGGL(per_keys_kernel, dim3(pgrid), dim3(256), 0, s, PK[dp], PV[dp], d_c, nbig, idx_bits, d_pg, d_epos,
                                           d_eell, d_etail, d_bt, BK[0], K[src]);
                        big_key_bits = 64;
                        st.periodic_rounds += 1;
                        st.periodic_members += per_members;
                    }
                    st.round_passes += (u32)sp.launches;
                } else {
                    (void)hipGetLastError();
                    set_error("%s", "");
                }
            }
            // Default (round 6): the merge sort where the large groups are small on average (<= 2^16 members: `words`,
            // source-like text -- a handful of tile sorts and merge passes inside every group against 8 + 3 chained radix
            // passes over all of them), the chained sorts where a few groups hold millions (`mixed`: twelve merge passes
            // against seven).  Measured: words 35.5-36.0 -> 35.2-35.4 ms, `source` 3-6 ms on one box and within the noise on
            // another, mixed / dup_blocks / real files unchanged (forced on everywhere: mixed 88 -> 91).
            const bool merge_auto = knobs.big_merge < 0 && (u64)nbig <= ((u64)nbig_groups << 16);
            if ((knobs.big_merge == 1 || merge_auto || (knobs.big_merge == 2 && use_text && big_key_bits > 32)) && nbig < 0x7fffffffu) {
                // segmented merge sort of the large groups (bg_*_kernel): tiles in LDS, then merge passes inside every group
                const u32 bound = nbig / BG_TILE + nbig_groups + 2;
                const size_t g4 = round_up(((size_t)nbig_groups + 2) * 4, 64), g8 = round_up(((size_t)nbig_groups + 2) * 8, 64);
                PSS_TRY(ctx->slot[S_BGT].reserve(g4 + g8 + round_up((size_t)bound * sizeof(BigTile), 64) + 256));
                u8 *tb = ctx->slot[S_BGT].as<u8>();
                u32 *d_gstart = reinterpret_cast<u32 *>(tb);
                u64 *d_toff = reinterpret_cast<u64 *>(tb + g4);
                BigTile *d_tiles = reinterpret_cast<BigTile *>(tb + g4 + g8);
                const u32 eg = std::min<u32>((nbig + 255) / 256, (u32)ctx->num_cus * 16);
                hipLaunchKernelGGL(bg_gstart_kernel, dim3(eg), dim3(256), 0, s, d_bgid, nbig, nbig_groups, d_gstart);
                PSS_TRY(device_excl_scan(ctx, InTileCount{d_gstart}, nbig_groups, d_partial, d_total, d_toff));
                PSS_HIP(hipMemsetAsync(d_tiles, 0, (size_t)bound * sizeof(BigTile), s));
                hipLaunchKernelGGL(bg_tiles_kernel, dim3(std::min<u32>((nbig_groups + 3) / 4, (u32)ctx->num_cus * 8)), dim3(256), 0, s, d_gstart,
                                   d_toff, nbig_groups, d_tiles);
                hipLaunchKernelGGL(bg_gather_kernel, dim3(eg), dim3(256), 0, s, d_bt, V[src], nbig, BV[0]);
                const u32 wg = std::min<u32>(bound, (u32)ctx->num_cus * 2);
                hipLaunchKernelGGL(bg_tile_sort_kernel, dim3(wg), dim3(BG_BLOCK), 0, s, BK[0], BV[0], d_tiles, bound, BK[1], BV[1]);
                int from = 1;
                for (u64 L = BG_TILE; L < (u64)nbig; L <<= 1) {
                    hipLaunchKernelGGL(bg_merge_kernel, dim3(wg), dim3(BG_BLOCK), 0, s, BK[from], BV[from], BK[from ^ 1], BV[from ^ 1], d_tiles,
                                       bound, (u32)L);
                    from ^= 1;
                }
                hipLaunchKernelGGL(bg_writeback_kernel, dim3(std::min<u32>(bound, (u32)ctx->num_cus * 8)), dim3(256), 0, s, BK[0], BV[0], BK[1], BV[1],
                                   d_tiles, bound, d_bt, K[src ^ 1], V[src ^ 1]);
                st.big_elems += nbig;
                return PSS_OK;
            }
            hipLaunchKernelGGL(iota_kernel, dim3((nbig + 255) / 256), dim3(256), 0, s, BV[0], nbig);
            SortStats s1, s2;
            int d1 = 0, d2 = 0;
            PSS_TRY(radix_sort_pairs(ctx, BK, BV, nbig, big_key_bits, 0xffffffffu, nullptr, 0, work, &d1, profile, &s1));
            hipLaunchKernelGGL(gather_gid_kernel, dim3((nbig + 255) / 256), dim3(256), 0, s, BV[d1], d_bgid, nbig,
                               nbig <= 4096u, BK[d1]);
            if (nbig_groups > 1)
                PSS_TRY(radix_sort_pairs(ctx, BK, BV, nbig, gid_bits, 0xffffffffu, nullptr, d1, work, &d2, profile, &s2));
            else d2 = d1;
            hipLaunchKernelGGL(big_writeback_kernel, dim3((nbig + 255) / 256), dim3(256), 0, s, BV[d2], d_bt, K[src],
                               V[src], nbig, K[src ^ 1], V[src ^ 1]);
            st.round_passes += (u32)(s1.launches + s2.launches);
            ss.launches += s1.launches + s2.launches;
            ss.elems += s1.elems + s2.elems;
            ss.ms += s1.ms + s2.ms;
            ss.ms_pairs += s1.ms_pairs + s2.ms_pairs;
            ss.pairs_launches += s1.pairs_launches + s2.pairs_launches;
            ss.pairs_elems += s1.pairs_elems + s2.pairs_elems;
            st.big_elems += nbig;
            return PSS_OK;
        };
        if (mode == M_TEXT) {
            const bool text_progress = text_rounds == 0 || (u64)m * 10 <= (u64)m_text_prev * 6;
            bool bail = true;
            const bool anchors_on = !subset && knobs.anchor != 0 && (knobs.anchor == 1 || n >= (1u << 20));
            bool skip_text = false;
            const bool probe_skips = h >= (u64)knobs.anchor_min_omega + 3;       // deep enough for the anchor round to run at once
            const bool side_on = knobs.side != 0 && !rank_only && (knobs.side == 1 || n >= (1u << 24)) && io.level == 0;
            bool side_wanted = false;
            if (anchors_on && !anchored && text_rounds == 0 && !knobs.no_probe && m >= 4096 && (u64)m * 16 >= (u64)n && (probe_skips || side_on)) {
                // are these ties repeats (see probe_repeats_kernel)?  Then no text round will resolve them.
                u32 *d_probe = d_counters + 48;
                PSS_HIP(hipMemsetAsync(d_probe, 0, 8, s));
                const u32 samples = 8192;
                hipLaunchKernelGGL(probe_repeats_kernel, dim3(samples / 256), dim3(256), 0, s, V[src], G[gcur], m, samples, n,
                                   (u32)std::min<u64>(h, 0xffffffffull), codes, d_probe);
                PSS_HIP(hipMemcpyAsync(h_small, d_probe, 8, hipMemcpyDeviceToHost, s));
                PSS_HIP(hipStreamSynchronize(s));
                st.probe_pairs = h_small[0];
                st.probe_same = h_small[1];
                skip_text = probe_skips && h_small[0] >= 64 && (u64)h_small[1] * 100 > (u64)h_small[0] * (u64)knobs.probe_skip_pct;
                side_wanted = side_on && !skip_text && h_small[0] >= 64 && (u64)h_small[1] * 100 >= (u64)h_small[0] * (u64)knobs.side_pct;
                if (knobs.timing) fprintf(stderr, "[pss] probe: %u of %u sampled pairs share 48 more symbols%s\n", h_small[1], h_small[0], skip_text ? ": no text rounds" : (side_wanted ? ": anchors beside the text round" : ""));
            }
            if (knobs.side == 1 && side_on && anchors_on && !anchored && text_rounds == 0 && !skip_text) side_wanted = true;
            if (side_wanted && !side.started && text_rounds < text_rounds_max)
                // the depth the coming text round will reach decides the window; the anchors' sort starts now, on the side
                PSS_TRY(side_start(ctx, knobs, side, n, codes, b, plus_one, h + (u64)kt, nullptr));
            if (anchored) {
                // cannot happen: the anchor round leaves no ties.  Counted (tests assert zero) and resolved by rank rounds.
                st.anchor_left += m;
            } else if (skip_text) {
            } else if (subset) {
                if (h < io.stop_text_h) PSS_TRY(local_round(true, &bail));
            } else if (text_rounds < text_rounds_max && text_progress) {
                m_text_prev = m;
                PSS_TRY(local_round(true, &bail));
            }
            if (bail && !anchored && anchors_on) {
                // Ties that outlive the text rounds are repeats: one round keyed by the ranks of the anchors (anchor_impl.h)
                // instead of log2(length of the repeat) rank rounds over the whole text.  The key array takes the place
                // of the inverse suffix array, which this path never builds.
                bool ok = false;
                const u32 *akey = ISA;
                if (side.started) {
                    side.join();
                    if (side.rc == PSS_ENOMEM) {
                        // no room for the side line's buffers is not an error of the build (side_start says so about its own
                        // reservations; the same holds for what the helper context grows later): give its slots back and let
                        // the anchors wait their turn on the main line
                        if (ctx->helper)
                            for (auto &sl : ctx->helper->slot) sl.release();
                        (void)hipGetLastError();
                        set_error("%s", "");
                        side.ok = false;
                    } else if (side.rc != PSS_OK) {
                        set_error("%s", side.err.c_str());
                        return side.rc;
                    }
                    if (side.ok && h >= side.h_eff) {
                        // sorted beside the text round: its keys are valid for every depth from h_eff on
                        const pss_sa_stats &t = side.st;
                        st.anchor = 1;
                        st.anchor_count = t.anchor_count;
                        st.anchor_omega = t.anchor_omega;
                        st.anchor_w = t.anchor_w;
                        st.anchor_ms += t.anchor_ms;
                        st.anchor_depth = h;
                        st.anchor_text_rounds += t.anchor_text_rounds;
                        st.anchor_rounds += t.anchor_rounds;
                        st.anchor_sum_active += t.anchor_sum_active;
                        st.anchor_left += t.anchor_left;
                        st.periodic_rounds += t.periodic_rounds;
                        st.periodic_members += t.periodic_members;
                        st.anchor_levels = std::max<uint64_t>(st.anchor_levels, t.anchor_levels);
                        st.anchor_side = 1;
                        akey = side.akey;
                        ok = true;
                    } else {
                        st.anchor_side = 2;      // thrown away: the text round gave up before it reached the window's depth, or the round declined
                    }
                }
                if (!ok) PSS_TRY(anchor_rank_keys(ctx, knobs, io, h, nullptr, nullptr, ISA, st, &ok));
                if (ok) {
                    st.anchor_active = m;
                    bool b2 = false;
                    PSS_TRY(local_round(false, &b2, akey));
                    anchored = true;
                    keyed_grp = false;
                    cur = src ^ 1;
                    st.rounds += 1;
                    st.sum_active += m;
                    PSS_HIP(hipGetLastError());
                    continue;
                }
            }
            if (!bail) {
                keyed_grp = false;
                cur = src ^ 1;
                st.rounds += 1;
                st.text_rounds += 1;
                st.sum_active += m;
                text_rounds += 1;
                h += (u64)kt;
                PSS_HIP(hipGetLastError());
                continue;
            }
            // ties outlived the text rounds: build the inverse suffix array once, continue with rank rounds
            hipLaunchKernelGGL(isa_from_sa_kernel, dim3(grid_all), dim3(256), 0, s, SA, n, ISA);
            hipLaunchKernelGGL(isa_active_kernel, dim3(grid), dim3(256), 0, s, V[src], G[gcur], m, ISA);
            mode = M_DENSE;
            if (subset) {
                h = 1;               // from here on the elements are the symbols of an integer string
                PSS_TRY(snapshot_symbols());
            }
        }
        if (mode == M_DENSE && X0 && !anchored && h >= 32 && (u64)m * 16 >= (u64)n && m >= (knobs.anchor == 1 ? 64u : (1u << 19))) {
            // a level up: minimizers of this string of symbols, named by the ranks the rounds have reached
            bool ok = false;
            PSS_TRY(anchor_rank_keys(ctx, knobs, io, h, X0, ISA, ISA, st, &ok));
            if (ok) {
                bool b2 = false;
                PSS_TRY(local_round(false, &b2, ISA));
                anchored = true;
                keyed_grp = false;
                cur = src ^ 1;
                st.rounds += 1;
                st.sum_active += m;
                PSS_HIP(hipGetLastError());
                continue;
            }
            X0 = nullptr;            // declined: plain rounds to the end
        }
        // Rank rounds: group-local unless large groups dominate (repetitive data) -- then one
        // global radix sort on (group rank, rank) with constant digits skipped is cheaper than
        // ranking in LDS + compaction + two chained sorts over nearly everything.
        if (mode == M_DENSE && m <= global_above) global_above = 0;
        if (mode == M_DENSE && global_above == 0) {
            bool bail = false;
            PSS_TRY(local_round(false, &bail));
            if (last_big_frac > 0.5 && !last_per) global_above = m / 2;      // (periodic runs: the local rounds know a shortcut)
            keyed_grp = false;
            cur = src ^ 1;
            st.rounds += 1;
            st.sum_active += m;
            h *= 2;
            PSS_HIP(hipGetLastError());
            continue;
        }

        // ------------- global doubling round: sparse (hash table + key search) or dense (ISA) --
        keyed_grp = true;
        if (mode == M_SPARSE && round == 0) {
            // hash table over the initially-active suffixes, in the (unused) ISA buffer
            u32 cap = 1024;
            while (cap < 4u * m) cap <<= 1;
            ra.ht_mask = cap - 1;
            PSS_HIP(hipMemsetAsync(ra.ht, 0, (size_t)cap * 8, s));
            hipLaunchKernelGGL(ht_insert_kernel, dim3(grid), dim3(256), 0, s, ra.ht, ra.ht_mask, V[src], G[gcur], m);
            // small key buffers carved out of the free big key buffer
            SK[0] = K[k0buf ^ 1];
            SK[1] = K[k0buf ^ 1] + (size_t)m;
            Kr = SK;
        }
        h_small[0] = 0; h_small[1] = 0; h_small[2] = 0xffffffffu; h_small[3] = 0xffffffffu;
        PSS_HIP(hipMemcpyAsync(d_red, h_small, 16, hipMemcpyHostToDevice, s));
        KeyArgs ka;
        ka.idx = V[src];
        ka.grp = G[gcur];
        ka.ISA = ISA;
        ka.ht = ra.ht;
        ka.ht_mask = ra.ht_mask;
        ka.sa = SA;
        ka.codes = codes;
        ka.code_bits = b;
        ka.key_chars = key_chars;
        ka.plus_one = plus_one;
        ka.m = m;
        ka.n = n;
        ka.h = (u32)std::min<u64>(h, 0xffffffffull);
        ka.rank_bits = rank_bits;
        ka.keys = Kr[src];
        ka.red = d_red;
        const u32 grid_keys = std::max(1u, std::min(grid, (m + 2047u) / 2048u));   // >= 8 keys per thread: fewer atomics on the two words
        if (mode == M_SPARSE) hipLaunchKernelGGL(build_keys_kernel<true>, dim3(grid), dim3(256), 0, s, ka);   // latency-bound key searches: every wave helps
        else hipLaunchKernelGGL(build_keys_kernel<false>, dim3(grid_keys), dim3(256), 0, s, ka);
        PSS_HIP(hipMemcpyAsync(h_small, d_red, 16, hipMemcpyDeviceToHost, s));
        PSS_HIP(hipStreamSynchronize(s));
        const u64 vor = (u64)h_small[0] | ((u64)h_small[1] << 32);
        const u64 vand = (u64)h_small[2] | ((u64)h_small[3] << 32);
        const u64 varying = vor & ~vand;
        const int key_bits = 2 * rank_bits;
        u32 mask = 0;
        for (int p = 0; p < (key_bits + 7) / 8; ++p)
            if ((varying >> (8 * p)) & 0xffull) mask |= 1u << p;
        SortStats rs;
        PSS_TRY(radix_sort_pairs(ctx, Kr, V, m, key_bits, mask, nullptr, src, work, &cur, profile, &rs));
        st.rounds += 1;
        st.round_passes += (u32)rs.launches;
        st.sum_active += m;
        ss.launches += rs.launches;
        ss.elems += rs.elems;
        ss.ms += rs.ms;
        ss.ms_pairs += rs.ms_pairs;
        ss.pairs_launches += rs.pairs_launches;
        ss.pairs_elems += rs.pairs_elems;
        h *= 2;
    }
    st.mode = was_text ? (mode == M_TEXT ? 2u : 3u) : (u64)mode;
    return PSS_OK;
}

// The key of the anchor round (anchor_impl.h): akey[i] = rank, among the anchor suffixes, of the anchor the window at i
// chose -- for every position i of the string.  `h`: symbols every tied group of the caller's active list shares.
//   syms == nullptr  the string is the text (outer.codes): w = 4 or 8 bytes hashed per position, omega + w - 1 <= h; the
//                    anchors are named by a sort of their own (text rounds to 2 omega + w - 1 symbols), then their names
//                    are a string of 32-bit symbols whose suffixes the rank rounds sort;
//   syms != nullptr  the string is that array of 32-bit symbols (a level up: the names of a coarser level's anchors):
//                    w = 1, omega = h / 2, and the anchors' names are the ranks the caller's rounds have reached
//                    (cur_ranks, depth h >= 2 omega).
// The caller's key buffers K[0], K[1] (8 n bytes each, scratch between two rounds) hold the anchors' own sort; `akey`:
// 4 n bytes (may be cur_ranks).  *ok = false: declined (window too narrow, too many anchors) -- nothing is lost but the
// time of the selection pass.
__global__ __launch_bounds__(256) void gather_names_kernel(const u32 *pos, u32 m, const u32 *ranks, u64 *keys, u32 *vals)
{
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < m; t += gridDim.x * blockDim.x) {
        keys[t] = ranks[pos[t]];
        vals[t] = t;
    }
}

static int anchor_rank_keys(DeviceCtx *ctx, const Knobs &knobs, const RoundsIO &outer, u64 h, const u32 *syms,
                            const u32 *cur_ranks, u32 *akey, pss_sa_stats &st, bool *ok)
{
    *ok = false;
    hipStream_t s = ctx->stream;
    const u32 n = outer.n;
    const u8 *codes = outer.codes;
    const bool forced = knobs.anchor == 1;
    int w;
    u64 omega64;
    if (syms) {
        w = 1;
        omega64 = h / 2;
    } else {
        w = h >= 28 ? 8 : 4;
        omega64 = h >= (u64)w ? h - (u64)w + 1 : 0;
    }
    if (knobs.anchor_omega > 0) omega64 = std::min<u64>(omega64, (u64)knobs.anchor_omega);
    const u32 omega = (u32)std::min<u64>(omega64, 64);       // wider windows: fewer anchors, but names of 2 omega + w - 1 symbols
    if (omega < (forced ? 2u : (u32)knobs.anchor_min_omega) || n < 64 || outer.level >= 6) return PSS_OK;
    const u32 num_tiles = (n + ANC_TILE - 1) / ANC_TILE;
    const size_t n16 = round_up((size_t)n, 16) + 16;
    const u32 cap_div = outer.level == 0 ? (u32)knobs.anchor_cap_div : 5u;
    const u32 m_cap = n / cap_div + 64;
    DevBuf &slot = ctx->slot[S_ANC + outer.level];
    PSS_TRY(slot.reserve(n16 + round_up((size_t)num_tiles * 4, 64) + ((size_t)num_tiles + 2) * 8 + (SC_MAX_BLOCKS + 8) * 8 +
                         (size_t)m_cap * 4 + 1024));
    u8 *base = slot.as<u8>();
    size_t o = 0;
    auto carve = [&](size_t bytes) { u8 *p = base + o; o = round_up(o + bytes, 64); return p; };
    u8 *d_dist = carve(n16);
    u32 *d_tile_cnt = reinterpret_cast<u32 *>(carve((size_t)num_tiles * 4));
    u64 *d_tile_off = reinterpret_cast<u64 *>(carve(((size_t)num_tiles + 2) * 8));
    u64 *d_partial = reinterpret_cast<u64 *>(carve((SC_MAX_BLOCKS + 8) * 8));
    u64 *d_total = d_partial + SC_MAX_BLOCKS;
    u32 *d_Q = reinterpret_cast<u32 *>(carve((size_t)m_cap * 4));
    BuildTimer tm;
    PSS_HIP(hipEventCreate(&tm.ev0));
    PSS_HIP(hipEventCreate(&tm.ev1));
    PSS_HIP(hipEventRecord(tm.ev0, s));
    const u32 n_read = (u32)(round_up((size_t)n, 16) + 64);      // the recoded text's padding (zero)
    const u32 grid = std::min<u32>(num_tiles, (u32)ctx->num_cus * 4);
    if (syms)
        hipLaunchKernelGGL(anc_select_kernel<true>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const u8 *>(syms), n, n, omega, w,
                           d_dist, d_tile_cnt, num_tiles);
    else
        hipLaunchKernelGGL(anc_select_kernel<false>, dim3(grid), dim3(256), 0, s, codes, n, n_read, omega, w, d_dist, d_tile_cnt,
                           num_tiles);
    PSS_TRY(device_excl_scan(ctx, InU32{d_tile_cnt}, num_tiles, d_partial, d_total, d_tile_off));
    u32 *h_small = outer.h_small;
    PSS_HIP(hipMemcpyAsync(h_small, d_total, 8, hipMemcpyDeviceToHost, s));
    PSS_HIP(hipStreamSynchronize(s));
    const u32 m = h_small[0];
    if (outer.level == 0) {
        st.anchor_count = m;
        st.anchor_omega = omega;
        st.anchor_w = (u64)w;
    }
    if (knobs.timing)
        fprintf(stderr, "[pss] anchors (level %d): n=%u h=%llu omega=%u w=%d anchors=%u (n / %.1f)\n", outer.level, n,
                (unsigned long long)h, omega, w, m, (double)n / std::max(1u, m));
    if (m == 0 || m > n / cap_div) return PSS_OK;
    hipLaunchKernelGGL(anc_walk_kernel<false>, dim3(grid), dim3(256), 0, s, d_dist, n, d_tile_off, num_tiles, d_Q,
                       (const u32 *)nullptr, (u32 *)nullptr);
    // the anchors' own sort, in the caller's two key buffers
    u8 *b0 = reinterpret_cast<u8 *>(outer.K[0]), *b1 = reinterpret_cast<u8 *>(outer.K[1]);
    const size_t m8 = round_up((size_t)m * 8, 256), m4 = round_up((size_t)m * 4 + 64, 256);
    u64 *AK[2] = {reinterpret_cast<u64 *>(b0), reinterpret_cast<u64 *>(b0 + m8)};
    if (outer.level == 0 && 9 * m4 > (size_t)n * 8 && 2 * m8 <= (size_t)n * 8) {
        // more anchors than the caller's second key buffer holds nine arrays of: a slot of their own
        PSS_TRY(ctx->slot[S_ANCW].reserve(9 * m4));
        b1 = ctx->slot[S_ANCW].as<u8>();
    }
    u32 *AV[2] = {reinterpret_cast<u32 *>(b1), reinterpret_cast<u32 *>(b1 + m4)};
    u32 *A_isa = reinterpret_cast<u32 *>(b1 + 2 * m4);
    u32 *AP[2] = {reinterpret_cast<u32 *>(b1 + 3 * m4), reinterpret_cast<u32 *>(b1 + 4 * m4)};
    u32 *A_grp = reinterpret_cast<u32 *>(b1 + 5 * m4);
    u32 *A_grp2 = reinterpret_cast<u32 *>(b1 + 6 * m4);
    u32 *A_sa = reinterpret_cast<u32 *>(b1 + 7 * m4);
    u32 *A_rank = reinterpret_cast<u32 *>(b1 + 8 * m4);
    if (2 * m8 > (size_t)n * 8 || (b1 == reinterpret_cast<u8 *>(outer.K[1]) && 9 * m4 > (size_t)n * 8)) return PSS_OK;      // (tiny strings)
    int kt = 64 / outer.b;
    if (kt > 16) kt = 16;
    const u32 gk = (u32)std::min<u64>((u64)ctx->num_cus * 8, ((u64)m + 255) / 256);
    int cur = 0;
    SortStats ss;
    RoundsIO io;
    if (syms) {
        hipLaunchKernelGGL(gather_names_kernel, dim3(gk), dim3(256), 0, s, d_Q, m, cur_ranks, AK[0], AV[0]);
        int bits = 1;
        while ((1ull << bits) <= (u64)n) ++bits;
        PSS_TRY(radix_sort_pairs(ctx, AK, AV, m, bits, 0xffffffffu, nullptr, 0, outer.work, &cur, false, &ss));
        io.codes = nullptr;
        io.b = 8; io.plus_one = 0; io.key_chars = 1;
        io.h0 = 1;
    } else {
        hipLaunchKernelGGL(subset_keys_kernel, dim3(gk), dim3(256), 0, s, d_Q, m, n, codes, outer.b, kt, outer.plus_one, AK[0], AV[0]);
        PSS_TRY(radix_sort_pairs(ctx, AK, AV, m, kt * outer.b, 0xffffffffu, nullptr, 0, outer.work, &cur, false, &ss));
        io.codes = codes;
        io.b = outer.b; io.plus_one = outer.plus_one; io.key_chars = kt;
        io.h0 = (u64)kt;
        io.sub_pos = d_Q;
        io.text_n = n;
        io.stop_text_h = 2ull * omega + (u64)w - 1;
    }
    io.n = m;
    io.SA = A_sa;
    io.K[0] = AK[0]; io.K[1] = AK[1];
    io.V[0] = AV[0]; io.V[1] = AV[1];
    io.ISA = A_isa;
    io.P[0] = AP[0]; io.P[1] = AP[1];
    io.GRP = A_grp;
    io.grp2 = A_grp2;
    io.key_drop = 0;
    io.cur = cur;
    io.final_buf = -1;
    io.v_scratch = nullptr;
    io.ties = false;
    io.msd_fused = false;
    io.msd_active = 0;
    io.no_sparse = true;
    io.work = outer.work;
    io.d_agg_head = outer.d_agg_head; io.d_agg_cnt = outer.d_agg_cnt; io.d_red = outer.d_red; io.d_counters = outer.d_counters;
    io.h_small = outer.h_small;
    io.profile = false;
    io.level = outer.level + 1;
    pss_sa_stats sub;
    memset(&sub, 0, sizeof sub);
    SortStats ss2;
    PSS_TRY(refine_rounds(ctx, knobs, io, ss2, sub));
    st.anchor_text_rounds += sub.text_rounds;
    st.anchor_rounds += sub.rounds - sub.text_rounds + sub.anchor_rounds;
    st.anchor_sum_active += sub.sum_active + sub.anchor_sum_active;
    st.anchor_left += sub.anchor_left;
    st.periodic_rounds += sub.periodic_rounds;
    st.periodic_members += sub.periodic_members;
    st.anchor_levels = std::max<uint64_t>(st.anchor_levels, 1 + sub.anchor_levels);
    hipLaunchKernelGGL(isa_from_sa_kernel, dim3(gk), dim3(256), 0, s, A_sa, m, A_rank);
    hipLaunchKernelGGL(anc_walk_kernel<true>, dim3(grid), dim3(256), 0, s, d_dist, n, d_tile_off, num_tiles, (u32 *)nullptr,
                       (const u32 *)A_rank, akey);
    PSS_HIP(hipEventRecord(tm.ev1, s));
    PSS_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    PSS_HIP(hipEventElapsedTime(&ms, tm.ev0, tm.ev1));
    if (outer.level == 0) {
        st.anchor_ms += ms;
        st.anchor_depth = h;
    }
    st.anchor = 1;
    *ok = true;
    return PSS_OK;
}
