// ym_host_plan.hpp -- host runtime: the planner of one call -- sizes and correlate decomposition, point cache, job lists, descriptor, raster coverage
// Part of yagmatch.hip (included inside its anonymous namespace); not a header of its own.
// ---------------------------------------------------------------- launch one call
// sizes, lattices (ScanMatcher::MatchScan), the device window, the correlate decomposition, device buffers
int plan_sizes(ym_matcher *m, Slot &slot, CallPlan &P) {
    Call &call = slot.call;
    P.B = (int)call.items.size();
    P.nscans = (int)call.scans.size();
    if (P.B <= 0) return set_err(YM_ERR_INVALID, "empty call");
    const int B = P.B;
    YmGeom &g = P.g;
    g = m->geom;
    double rq = 0;
    for (const CallItem &it : call.items) {
        P.max_base = std::max(P.max_base, it.base_count);
        rq = std::max(rq, call.scans[it.query].max_valid);
    }
    for (const CallScan &s : call.scans) P.max_n = std::max(P.max_n, s.n);
    const int max_n = P.max_n, max_base = P.max_base;
    // Karto sizes its grid from the MATCHER's range threshold; a query reading beyond it (scans carry their own threshold:
    // /root/reference/yag_slam/models.py:110-116) points outside that grid, where GetResponse's linear-index test wraps
    // around Karto's own row pitch.  Such a call is answered exactly as Karto would: window = Karto's whole storage, every
    // linear index formed with Karto's pitch, per-cell paths only (ym_k_common.hpp, cell_value).
    const bool wrap = g.semantics == YM_SEM_KARTO && rq > m->cfg.range_threshold;
    g.kpitch = wrap ? (g.storage_w + 7) / 8 * 8 : 0;
    if (max_n > YM_MAX_BEAMS) return set_err(YM_ERR_UNSUPPORTED, "scan has %d readings; limit is %d", max_n, YM_MAX_BEAMS);

    const bool yag = P.yag = g.semantics == YM_SEM_YAGPY;
    P.chain_step = call.chain_step;
    const double coarse_off = yag ? 0.5 * m->cfg.search_size : 0.5 * (g.side - 1) * g.res;
    const double coarse_step = 2 * g.res;
    YmLattice &lc = P.lc, &lf = P.lf;
    if (yag) { // lattices are built on the device from np.arange; the Karto tables stay empty
        std::memset(&lc, 0, sizeof lc);
        std::memset(&lf, 0, sizeof lf);
        lc.step_x = lc.step_y = coarse_step;
        lc.angle_res = m->cfg.coarse_angle_resolution;
        lf.fine = 1;
        if (m->yag_fast) {
            // the lattice the production correlate kernels are launched on: len(np.arange(-s + c, s + c, step)) = ceil(((s + c) - (-s + c)) / step)
            // is floor(2 s / step) + 1 or -- where 2 s / step is an integer and the subtraction rounds down -- one less
            // (/root/reference/yag_slam/helpers.py:177-179); an item whose own lengths exceed it is scored by yag_score_kernel
            lc.nx = lc.ny = (int)std::floor(m->cfg.search_size / coarse_step + 1e-6) + 1;
            lc.nt = (int)std::floor(m->cfg.coarse_search_angle_offset / m->cfg.coarse_angle_resolution + 1e-6) + 1;
            lc.off_x = lc.off_y = coarse_off;
            lc.angle_off = 0.5 * m->cfg.coarse_search_angle_offset;
            if (lc.nx > YM_YAG_MAX_DIM || lc.nt > YM_MAX_COARSE_NT) lc.nx = lc.ny = lc.nt = 0; // (the bounds check below refuses such a matcher anyway)
        }
    } else {
        lc = make_lattice(g, coarse_off, coarse_step, call.coarse_angle_off, m->cfg.coarse_angle_resolution, 0,
                          call.penalize);
        lf = make_lattice(g, coarse_step * 0.5, g.res, 0.5 * m->cfg.coarse_angle_resolution,
                          m->cfg.fine_search_angle_resolution, 1, call.penalize);
    }
    slot.coarse = lc;
    slot.fine = lf;

    // ---- device window: the central part of Karto's storage the query endpoints can reach
    const int centre = g.border + (g.roi_w - 1) / 2;
    const double reach = rq + coarse_off + (yag ? 3 : 1) * g.res; // yagpy's fine pass reaches 2 cells past the coarse box (and the launch
                                                                  // lattice of its coarse pass one step = 2 cells past the last real hypothesis)
    int wh = (int)std::ceil(reach / g.res) + 3;
    // (in steps of 64 cells: the window -- and with it the "this tile is zero" knowledge about its memory -- then stays
    //  the same from match to match while the queries' longest readings differ by less)
    wh = (wh + 63) / 64 * 64;
    if (m->last_wh >= wh && m->last_wh - wh <= 256) wh = m->last_wh; // (and not smaller again at once: a window up to 256 cells too wide stays)
    m->last_wh = wh;
    wh = wrap ? centre : std::min(wh, centre);
    g.win_origin = centre - wh;
    g.win_w = std::min(2 * wh + 1 + (yag ? 1 : 0), g.storage_w - g.win_origin); // even yagpy grids have no centre cell
    if (wrap) { g.win_origin = 0; g.win_w = g.storage_w; }
    // tall tiles where the raster is throughput-bound and the window large (measured: 4096 items of the default config
    // gain 12 % of the raster, a single match loses 6 us, the loop config's 5 cm windows lose 3 %)
    // (what a matcher knows of its windows' memory is kept per tile: a change of tile height drops it all.  A matcher that serves
    //  single matches BETWEEN large batches therefore keeps the batches' tall tiles for its next 64 small calls -- 6 us per single
    //  match against a full raster of every window of the next batch, 2 ms per 4096 items: bench.py, cfg2x_alternating)
    {
        // (round 6: only a matcher that has SHOWN the pattern -- a small call between two large batches -- keeps the tall tiles for its small
        //  calls: the first small call after a large batch pays one reset of the window knowledge, and a matcher that never runs a large
        //  batch again is not taxed 6 us per single match for its next 64 calls)
        bool tall = B >= 512 && g.win_w >= m->tall_tiles_min_window;
        if (tall) {
            if (m->tall_pattern == 2) m->tall_alternates = true; // large, small, large: it alternates
            m->tall_pattern = 1;
            m->sticky_tall_left = m->tall_alternates ? 64 : 0;
        } else if (g.win_w >= m->tall_tiles_min_window && !call.chain_step) {
            if (m->tall_pattern == 1) m->tall_pattern = 2;
            if (m->sticky_tall_left > 0) { tall = true; m->sticky_tall_left--; }
        }
        P.tile_h = m->tile_h_forced ? m->tile_h_forced : tall ? YM_TILE_H_TALL : YM_TILE_H;
    }
    P.tiles_x = (g.win_w + YM_TILE_W - 1) / YM_TILE_W;
    P.tiles_y = (g.win_w + P.tile_h - 1) / P.tile_h;
    g.pitch = P.tiles_x * YM_TILE_W + 64;
    P.grid_stride = align_up((size_t)g.pitch * g.win_w + 64, 256);
    if ((double)g.pitch * g.win_w > 2.0e9) return set_err(YM_ERR_UNSUPPORTED, "correlation window too large");
    if (lf.nx > 64 || lf.ny > 64 || lf.nt > YM_MAX_FINE_NT || (int64_t)lf.nx * lf.ny * lf.nt > YM_MAX_FINE_HYP)
        return set_err(YM_ERR_UNSUPPORTED, "fine lattice %dx%dx%d exceeds the built-in limit", lf.nx, lf.ny, lf.nt);
    if (lc.nt > YM_MAX_COARSE_NT)
        return set_err(YM_ERR_UNSUPPORTED, "%d coarse angles exceed the built-in limit of %d", lc.nt, YM_MAX_COARSE_NT);

    // ---- coarse correlate decomposition
    P.sx = yag ? 2 : (int)kt_round_h(lc.step_x * g.scale);
    if (P.sx != 1 && P.sx != 2) return set_err(YM_ERR_UNSUPPORTED, "coarse lattice step of %d cells", P.sx);
    // yagpy lattice bounds (np.arange lengths are fixed on the device; these only size the buffers)
    P.ymaxd = yag ? std::max(8, (int)std::ceil(m->cfg.search_size / coarse_step) + 2) : 0;
    P.ymaxt = yag ? std::max(13, (int)std::ceil(m->cfg.coarse_search_angle_offset / m->cfg.coarse_angle_resolution) + 2) : 0;
    if (yag && (P.ymaxd > YM_YAG_MAX_DIM || P.ymaxt > YM_YAG_MAX_NT))
        return set_err(YM_ERR_UNSUPPORTED, "yagpy lattice %d x %d x %d exceeds the built-in limit", P.ymaxd, P.ymaxd, P.ymaxt);
    P.yvol = (size_t)P.ymaxt * P.ymaxd * P.ymaxd;
    const int G = 16;
    P.ngx = (lc.nx + G - 1) / G;
    P.nx_pad = P.ngx * G;
    const int njobs = P.njobs = lc.ny * P.ngx;
    // (measured on MI355X: sharing a block between adjacent angles does not help -- the kernel is bound by
    //  L1 tag lookups per lane, not by line reuse -- so one angle per block)
    P.tpb = 1;
    P.ktiles = lc.nt;
    // a block = 4 waves = jw job-waves x cw chunk-waves: lattices with one (two) waves of lane jobs put four (two)
    // consecutive beam chunks into one block and add them up before the partial sum is written
    P.cw = m->corr_cw > 0 ? m->corr_cw : njobs <= 64 ? 4 : njobs <= 128 ? 2 : 1;
    P.job_blocks = (njobs + (4 / P.cw) * 64 - 1) / ((4 / P.cw) * 64);
    // split the beams so that roughly >= 2048 waves are in flight, chunks of 32..512 beams; a small lattice (one
    // working wave per block) does best with blocks of 64 beams even when the batch alone fills the chip
    // (measured on MI355X, cfg2 x 256, whole step: 3 chunks 1.09 ms, 8 chunks 0.95 ms, 17 chunks 0.91 ms, 23 chunks
    // 0.97 ms; the partial sums are 16-bit)
    const double waves_one_chunk = (double)((njobs + 63) / 64) * lc.nt * B;
    int n_chunks = (int)std::ceil(2048.0 / std::max(1.0, waves_one_chunk));
    n_chunks = std::max(1, std::min(n_chunks, (max_n + 31) / 32));
    n_chunks = std::max(n_chunks, (max_n + 511) / 512);
    if (njobs <= 128) n_chunks = std::max(n_chunks, (max_n + 63) / 64);
    if (m->corr_chunks > 0) n_chunks = std::max(m->corr_chunks, (max_n + 511) / 512);
    int chunk = (max_n + n_chunks - 1) / n_chunks;
    // beams in flight per lane: 32 for the latency-bound single match (one 32-beam chunk per wave), else 16
    // (with the items pinned to XCDs 16 beats 32 on the batch: 618 vs 664 us; 48 spills)
    P.corr_u = m->corr_u > 0 ? m->corr_u : (chunk <= 32 && njobs <= 128 ? 32 : 16);
    chunk = (chunk + P.corr_u - 1) / P.corr_u * P.corr_u;
    while (P.cw > 1 && P.cw * chunk > 640) P.cw /= 2; // a group's 16-bit sums must hold cw * chunk beams of 100
    if (P.corr_u == 48) P.cw = 1;                     // (development variant: one instantiation only)
    P.job_blocks = (njobs + (4 / P.cw) * 64 - 1) / ((4 / P.cw) * 64);
    P.chunk = chunk;
    P.n_chunks = (max_n + chunk - 1) / chunk;
    P.n_groups = (P.n_chunks + P.cw - 1) / P.cw;
    // coarse grids (loop closure: 5 cm cells, neighbouring end points ~1.3 cm apart) see runs of beams in one cell
    {
        const double spacing = call.scans[call.items[0].query].beam_spacing;
        const bool likely = spacing > 0 && spacing < 0.6 * g.res;
        P.dedup = (!wrap && P.sx == 2 && chunk == 64 && (m->corr_dedup ? m->corr_dedup == 1 : likely)) ? 1 : 0;
    }

    // Batches on lattices of at most 26 x 32 (a lattice row = two lanes of 13 hypotheses) without merged offsets: the
    // patches are gathered from LDS, region by region (ym_k_region.hpp).  Single matches keep the direct kernel: a region
    // walk is one long chain.
    {
        const int half_w = (g.win_w + 1) / 2;
        // (measured, 21 angles: three blocks of 8 waves per CU beat blocks of 7 although the third block of an item idles 3 waves)
        P.rg_nw = m->corr_region_nw > 0 ? std::min(m->corr_region_nw, 16) : lc.nt <= 8 ? lc.nt : 8;
        if (P.rg_nw < 4 || P.rg_nw == 9 || (P.rg_nw > 11 && P.rg_nw != 16)) P.rg_nw = lc.nt <= 4 ? 4 : 8;
        P.rg_parts = (lc.nt + P.rg_nw - 1) / P.rg_nw;
        int rg_h = YM_RG_H;
#ifdef YM_EXPERIMENTAL // (the forms that lost to correlate_region_kernel: scripts/exp/forms, `make experimental`)
        // the wave-specialised form: blocks of 8, 11 or 12 gather waves (21 angles = 11 + 10) + 4 loader waves, regions of one class image
        P.rg_ws = m->corr_region_nw == 0 && m->corr_region_form == 2; // (measured slower than the first form: opt-in, option 32 = 2)
        if (P.rg_ws) { // (8 gather waves + 8 loader waves per block, two blocks per CU)
            P.rg_nw = YM_WS_NG;
            P.rg_parts = (lc.nt + YM_WS_NG - 1) / YM_WS_NG;
        }
        // round 5: large batches on up to 24 angles -- sixteen waves per block, two or three per angle (ym_k_region2.hpp)
        // (below two blocks per CU the regions of the first form are dealt out to more blocks instead: rsplit)
        P.rg2 = !P.rg_ws && m->corr_region_nw == 0 && !m->keep_planes && lc.nt > 0 &&
                (m->corr_region_form == 5 || (m->corr_region_form == 0 && B >= m->rg2_min_batch && 3 * B >= 2 * m->n_cus));
        P.rg2_h = m->rg2_h == 80 ? 80 : m->rg2_h == 100 ? 100 : 128;
        if (P.rg2) { P.rg_nw = lc.nt <= 8 ? lc.nt : 8; P.rg_parts = (lc.nt + P.rg_nw - 1) / P.rg_nw; }
        rg_h = P.rg_ws ? YM_WS_H : P.rg2 ? P.rg2_h : YM_RG_H;
#endif
        P.rg_nrx = (half_w + YM_RG_W - 1) / YM_RG_W;
        P.rg_nry = (half_w + rg_h - 1) / rg_h;
        P.rg_nregions = P.rg_nrx * P.rg_nry;
        // sets of 16-bit sums per (item, angle): room for the padding of the entry lists (an item that needs more is scored
        // by the per-cell path)
        P.rg_ng = ((max_n * 23 + 19) / 20 + YM_RG_FLUSH - 1) / YM_RG_FLUSH;
        P.rg_nbins = P.rg_nregions * lc.nt;
        P.region26 = !wrap && (!yag || lc.nx > 0) && !P.dedup && !call.slice && P.sx == 2 && B >= m->rg_min_batch && m->corr_region != 1 && m->corr_region != 4 && lc.nx <= 2 * YM_RG_G &&
                     lc.ny <= 32 && P.rg_ng <= 8 && (int64_t)lc.nt * max_n <= YM_RG_MAX_ENTRIES && P.rg_nbins < YM_RG_MAX_BINS && max_n < 2048 &&
                     half_w <= 4096; // (region_entry divides a class row below 4096 by the region height as a multiplication: a taller window takes the gather or the direct path)
        if (P.region26) {
            // fewer blocks than three per CU: deal every (item, angle block)'s regions out to several blocks (64 chains: the
            // kernel 121 -> 65 us with four, the enqueue 236 -> 205 us; 128 chains 317 -> 295 with two; scripts/dev/rsplit_time.py)
            P.rg_rsplit = 1;
            if (!P.rg_ws && !P.rg2 && m->corr_region_rsplit != 1 && m->corr_region_form != 3 && m->corr_region_form != 4) {
                const int blocks = B * P.rg_parts;
                P.rg_rsplit = m->corr_region_rsplit > 1 ? m->corr_region_rsplit : std::max(1, std::min(8, (3 * m->n_cus) / std::max(1, blocks)));
            }
            P.n_groups = P.rg_ng * P.rg_rsplit;
#ifdef YM_EXPERIMENTAL
            // batches that fill the chip with one block per item: correlate_item_kernel (option 32: 1 = never, 3 = always)
            P.rg_item = !P.rg_ws && !P.rg2 && P.rg_rsplit == 1 && lc.nt <= YM_IT_MAX_NT && m->corr_region_form != 1 &&
                        (m->corr_region_form == 3 || B >= m->item_min_batch);
            P.n_groups = P.rg_ng * P.rg_rsplit * (P.rg2 ? YM_R2_MAX_WPA : 1); // (rg2: every slice of an angle writes its own sets)
            // the pooled form (option 32 = 4): large batches of at most 22 angles
            P.rg_pool = !P.rg_ws && !P.rg2 && !P.rg_item && P.rg_rsplit == 1 && m->corr_region_form == 4 && lc.nt <= 2 * YM_PL_MAX_NK && m->corr_region_nw == 0;
            if (P.rg_pool) {
                P.rg_nw = lc.nt <= YM_PL_MAX_NK ? lc.nt : (lc.nt + 1) / 2;
                P.rg_parts = (lc.nt + P.rg_nw - 1) / P.rg_nw;
            }
#endif
            // the default form at eight waves stages from the window: the raster of such a call writes no planes (option 39 = 1: keeps them)
            P.win_only = !P.rg_ws && !P.rg_item && (P.rg_nw == 8 || P.rg_pool || P.rg2) && !m->keep_planes;
            // (+ the padding of the bins that hold work; a query whose list still does not fit takes the per-cell path)
            // 10 % over the pairs themselves (measured on the bench scans: 5 %)
            // (the wave-specialised form's bins are a third more and hold less each: 20 %)
            // the lists: one part per angle block of the correlate (the experimental forms read one list of all angles)
            const bool whole = P.rg_ws || P.rg2 || P.rg_item || P.rg_pool;
            P.rg_lnw = whole ? lc.nt : P.rg_nw;
            P.rg_lparts = whole ? 1 : P.rg_parts;
            P.rg_nbins = P.rg_nregions * P.rg_lnw;
            P.rg_entries_pstride = std::min((size_t)YM_RG_MAX_ENTRIES, ((size_t)P.rg_lnw * max_n * 11 / 10 + 63) / 64 * 64);
            P.rg_entries_stride = P.rg_entries_pstride * P.rg_lparts; // (positions are 16-bit in the correlate: checked on the device per part)
            P.rg_starts_stride = ((size_t)P.rg_lparts * (P.rg_nbins + 1) + P.rg_lparts + 15) / 16 * 16;
        }
    }
    // Other batches on lattices of at most 48 x 64: the general form (ym_k_gather.hpp).
    P.region = !wrap && !P.region26 && (!yag || lc.nx > 0) && !call.slice && P.sx == 2 && B >= m->lds_min_batch && m->corr_region != 1 && lc.nx <= 16 * YM_GA_MAX_SEG && lc.ny <= 64;
    if (P.region) {
        // lanes: a lane owns 16 x-adjacent hypotheses of one lattice row; a group of 32 lanes = up to 32 rows of one
        // segment (conflict-free LDS reads), or the rows past 32 of several segments; a wave = two groups
        std::vector<std::vector<uint32_t>> groups;
        P.ga_nseg = (lc.nx + YM_GA_G - 1) / YM_GA_G;
        for (int sgm = 0; sgm < P.ga_nseg; sgm++) {
            groups.emplace_back();
            for (int r = 0; r < std::min(lc.ny, 32); r++) groups.back().push_back((uint32_t)r | (uint32_t)sgm << 8 | 1u << 16);
        }
        if (lc.ny > 32) {
            const int tn = lc.ny - 32, per = 32 / tn;
            for (int sgm = 0; sgm < P.ga_nseg; sgm++) {
                if (sgm % per == 0) groups.emplace_back();
                for (int r = 32; r < lc.ny; r++) groups.back().push_back((uint32_t)r | (uint32_t)sgm << 8 | 1u << 16);
            }
        }
        P.ga_np = ((int)groups.size() + 1) / 2;
        std::vector<uint32_t> tab((size_t)P.ga_np * 64);
        for (int w = 0; w < P.ga_np; w++)
            for (int l = 0; l < 64; l++) {
                const size_t gi = (size_t)2 * w + l / 32;
                // idle lanes read what a working lane of their group (or wave) reads: a broadcast, no bank conflict
                uint32_t v = groups[2 * w][0] & 0xffffu;
                if (gi < groups.size()) v = (size_t)(l % 32) < groups[gi].size() ? groups[gi][l % 32] : (groups[gi][0] & 0xffffu);
                tab[(size_t)w * 64 + l] = v;
            }
        if (P.ga_np > YM_GA_MAX_NP) P.region = false;
        if (P.region && tab != m->ga_lane_job_host) {
            int rc2 = m->ga_lane_job.ensure(tab.size());
            if (rc2) return rc2;
            HIP_TRY(hipStreamSynchronize(m->stream)); // (calls in flight read the old table)
            HIP_TRY(hipMemcpy(m->ga_lane_job.p, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            m->ga_lane_job_host = tab;
        }
    }
    if (P.region) {
        // blocks per item and angles per wave (their sums live in registers: NA x NP x 8).  Measured on MI355X (4096 items,
        // profiles/r03_gather_sweep.md): one angle per wave and about eight angles per block -- the copy of a region costs a
        // block one memory round trip per work item, which only more resident blocks hide -- beat fewer, larger blocks
        // although every block of an item stages the item's regions again
        int parts = m->corr_region_parts > 0 ? m->corr_region_parts : (lc.nt + 7) / 8;
        parts = std::min(parts, lc.nt);
        const int na_max = P.ga_np == 1 ? 4 : P.ga_np == 2 ? 2 : 1;
        for (;; parts++) {
            P.ga_kpp = (lc.nt + parts - 1) / parts;
            P.ga_na = m->corr_region_na > 0 ? std::min(na_max, m->corr_region_na) : 1;
            if ((P.ga_kpp + P.ga_na - 1) / P.ga_na > 16 && m->corr_region_na <= 0) P.ga_na = na_max;
            P.ga_nwv = (P.ga_kpp + P.ga_na - 1) / P.ga_na;
            if (P.ga_nwv <= 16) break;
        }
        if (P.ga_nwv == 7) P.ga_nwv = 8; // (an eighth wave shares the copy work)
        P.ga_parts = (lc.nt + P.ga_kpp - 1) / P.ga_kpp;
        P.ga_cap = std::min(64 * P.ga_nwv, m->corr_region_cap > 0 ? (m->corr_region_cap + 63) / 64 * 64 : 512); // (one unit per thread and copy)
        // regions: a thread copies PER 16-byte chunks of the class image of a region (+ the patch margin) per work item;
        // the fewest staged bytes win
        const int per = YM_GA_PER;
        const int blocks_per_cu = std::max(1, std::min(3, 32 / P.ga_nwv));
        const size_t budget = m->corr_region_lds > 0 ? (size_t)m->corr_region_lds : (size_t)(160 * 1024) / blocks_per_cu - 512;
        const int tasks = per * 64 * P.ga_nwv;
        const int half_w = (g.win_w + 1) / 2;
        double best = 1e300;
        for (int nrx = 1; nrx <= 256; nrx++) {
            const int W = (half_w + nrx - 1) / nrx;
            if (nrx > 1 && (half_w + nrx - 2) / (nrx - 1) == W) continue;
            const int cpr = (W + YM_GA_G * P.ga_nseg + 3 + 15) / 16, Pp = 16 * cpr + 8;
            for (int nry = 1; nry <= 256; nry++) {
                const int H = (half_w + nry - 1) / nry, rows = H + lc.ny;
                const int rows_lds = (tasks + cpr - 1) / cpr + 1; // what the block's copy tasks cover
                if (cpr * rows > tasks || (size_t)Pp * rows > 65528 || YM_GA_LDS_BYTES(Pp, rows_lds, P.ga_cap, P.ga_kpp) > budget) continue;
                const double cost = (double)nrx * nry * Pp * rows;
                if (cost < best) { best = cost; P.ga_W = W; P.ga_H = H; P.ga_P = Pp; P.ga_rows = rows_lds; P.ga_nrx = nrx; P.ga_nry = nry; }
                break; // (more rows of regions only add margins)
            }
        }
        if (best == 1e300) P.region = false;
    }
    if (P.region) {
        P.ga_lds = YM_GA_LDS_BYTES(P.ga_P, P.ga_rows, P.ga_cap, P.ga_kpp);
        // the epilogue's per-cell maxima, distance penalties and block maxima live in the same LDS
        const size_t epi = (size_t)lc.nx * lc.ny * 16 + (size_t)P.ga_kpp * ((lc.nx * lc.ny + YM_SCORE_THREADS - 1) / YM_SCORE_THREADS) * 8;
        P.ga_lds = std::max(P.ga_lds, epi + 64);
        P.ga_ng = max_n / (YM_GA_FLUSH - 16) + 1; // sets of 16-bit sums a wave may have to write out per job
        P.ga_nbins2 = P.ga_nrx * P.ga_nry * 4 * lc.nt * 2;
        P.ga_units_stride = ((size_t)lc.nt * max_n + 128 + 63) / 64 * 64;
        P.ga_starts_stride = ((size_t)2 * P.ga_nbins2 + 1 + 64 + 15) / 16 * 16;
        P.ga_work_stride = 1 + 3 * ((size_t)P.ga_nrx * P.ga_nry * 4 + P.ga_units_stride / P.ga_cap + 1);
        P.n_groups = P.ga_ng;
    }
    {
        static const bool debug_plan = getenv("YM_DEBUG_PLAN") != nullptr; // development aid: which correlate a call takes
        if (debug_plan)
            fprintf(stderr, "[ym] B %d region %d gather %d W %d H %d P %d rows %d nrx %d nry %d nseg %d np %d parts %d kpp %d na %d nwv %d lds %zu max_n %d nx %d ny %d nt %d\n",
                    B, (int)P.region26, (int)P.region, P.ga_W, P.ga_H, P.ga_P, P.ga_rows, P.ga_nrx, P.ga_nry, P.ga_nseg, P.ga_np, P.ga_parts, P.ga_kpp, P.ga_na,
                    P.ga_nwv, P.ga_lds, max_n, lc.nx, lc.ny, lc.nt);
    }

    P.nt_stride = lc.nt;
    P.dim_stride = std::max(lc.nx, lc.ny);
    P.sums_c = (size_t)lc.nt * lc.ny * lc.nx;
    P.sums_f = (size_t)lf.nt * lf.ny * lf.nx;
    P.partial_stride = P.region26 ? (size_t)P.n_groups * lc.nt * 64 * 16 : P.region ? (size_t)P.ga_ng * lc.nt * P.ga_np * 64 * 16 : (size_t)P.n_groups * lc.nt * lc.ny * P.nx_pad;
    P.cell_blocks = (lc.nx * lc.ny + YM_SCORE_THREADS - 1) / YM_SCORE_THREADS;
    P.score_blocks = P.cell_blocks * lc.nt; // block maxima per (angle, block of cells)

    int rc;
    if ((rc = m->states.ensure(B))) return rc;
    if ((rc = m->qlocal.ensure((size_t)B * max_n))) return rc;
    if ((rc = m->qnp.ensure(B))) return rc;
    if ((rc = m->cells.ensure((size_t)B * max_base * max_n))) return rc;
    if ((rc = m->bbox.ensure((size_t)B * max_base * YM_N_BOXES(max_n)))) return rc;
    size_t window_slack = YM_RG_WINDOW_SLACK(g.pitch), planes_slack = YM_RG_PLANES_SLACK(g.pitch / 2);
#ifdef YM_EXPERIMENTAL
    window_slack = std::max(window_slack, YM_R2_WINDOW_SLACK(g.pitch, 128));
    planes_slack = std::max(planes_slack, YM_WS_PLANES_SLACK(g.pitch / 2));
#endif
    if ((rc = m->grid.ensure((size_t)B * P.grid_stride + window_slack))) return rc;
    if ((rc = m->planes.ensure((size_t)B * P.grid_stride + std::max(planes_slack, YM_GA_PLANES_SLACK(g.pitch / 2, std::max(P.ga_rows, P.ga_nry + P.ga_H), lc.ny, P.ga_P))))) return rc;
    if ((rc = m->ctrig.ensure((size_t)B * P.nt_stride))) return rc;
    if ((rc = m->foffsets.ensure((size_t)B * lf.nt * max_n))) return rc;
    if ((rc = m->hypcell.ensure((size_t)B * 2 * P.dim_stride))) return rc;
    if ((rc = m->partial.ensure((size_t)B * P.partial_stride + 16))) return rc;
    if ((rc = m->sums.ensure((size_t)B * std::max(P.sums_c + P.sums_f, 2 * P.yvol + (yag ? P.sums_c : 0))))) return rc; // (yagpy: [pass 0][pass 1][launch lattice])
    if ((rc = m->resp.ensure((size_t)B * std::max(P.sums_c, P.yvol)))) return rc;
    if (yag) {
        if (!m->yag_counters.p) {
            if ((rc = m->yag_counters.ensure(8))) return rc;
            HIP_TRY(hipMemsetAsync(m->yag_counters.p, 0, m->yag_counters.cap * sizeof(unsigned long long), m->stream));
        }
        if ((rc = m->yaxes.ensure((size_t)B * 3 * YM_YAG_MAX_DIM))) return rc;
    }
    if ((rc = m->blockmax.ensure((size_t)B * P.score_blocks))) return rc;
    if ((rc = m->probs.ensure((size_t)B * lc.nx * lc.ny))) return rc;
    P.resp = call.ext_resp ? call.ext_resp : m->resp.p;
    P.probs = call.ext_probs ? call.ext_probs : m->probs.p;
    P.fuse_score = P.region26 && !P.rg_ws && P.rg_rsplit == 1 && !m->keep_sums && m->corr_fuse_score != 2 && !yag; // (yagpy scores the integer sums its own way)
    P.k_begin = call.slice ? std::max(0, call.k_begin) : 0;
    P.k_end = call.slice ? std::min(lc.nt, call.k_end) : lc.nt;
    if (call.slice && (yag || B != 1)) return set_err(YM_ERR_UNSUPPORTED, "angle-sliced matches are single Karto matches");
    P.stamps = m->stamps_on ? m->stamps.p : nullptr;
    return YM_OK;
}

// the point cache: give every resident base scan of the call (and, on batches, every resident query) its slot and
// decide whether the slot is current.  Key = scan id * 2 + role (0 base: world points + trigger chain; 1 query:
// sensor-frame points).
int plan_cache(ym_matcher *m, Slot &slot, const CallPlan &P) {
    Call &call = slot.call;
    const int n = (int)call.scans.size();
    for (CallScan &s : call.scans) {
        s.cache = s.qcache = nullptr;
        s.stale = s.qstale = 0;
        // the scan's creation-time structure holds at this pose (ym_k_prepare.hpp, structure_kernel)
        s.direct = m->use_scan_structure && s.gov && s.cidx && std::fabs(s.pose[0]) < YM_CHAIN_POSE_LIMIT && std::fabs(s.pose[1]) < YM_CHAIN_POSE_LIMIT &&
                   std::fabs(s.pose[2]) < YM_CHAIN_HEADING_LIMIT;
    }
    if (m->cache_off) return YM_OK;
    const uint64_t this_call = ++m->call_counter;
    // roles of every scan in this call
    std::vector<unsigned char> role(n, 0); // bit 0: base of some item, bit 1: query of some item (batches only)
    for (const CallItem &it : call.items) {
        for (int j = 0; j < it.base_count; j++) role[it.base_begin + j] |= 1;
        // a few items: the query is projected by the item's own block.  A query on its FIRST use in a batch is projected into the call's
        // own buffer and gets no slot of the point cache: a node that matches every incoming scan once and drops it (bench.py,
        // cfg2x_fresh_scans: 4096 new scans per enqueue) would otherwise fill the cache with 70 MB of dead entries per enqueue, and every
        // doubling of the arena costs a device synchronisation and the re-projection of every resident scan
        if (P.B >= 8 && call.scans[it.query].query_uses > 0) role[it.query] |= 2;
    }
    struct Want { int scan, kind; };
    std::vector<Want> wants;
    for (int i = 0; i < n; i++) {
        if (call.scans[i].id == 0 || call.scans[i].n <= 0) continue;
        // a few items: a scan with a trusted structure is projected by its own block faster than its cache slot is read
        // (one round of loads instead of three), so it gets none
        if ((role[i] & 1) && !(P.B < 8 && call.scans[i].direct)) wants.push_back(Want{i, 0});
        if (role[i] & 2) wants.push_back(Want{i, 1});
    }
    auto bytes_of = [](const CallScan &s, int kind) { return align_up(kind ? YM_QCACHE_BYTES(s.n) : YM_CACHE_BYTES(s.n), 16); };
    for (int attempt = 0; attempt < 2; attempt++) {
        // look every scan up; count what the new ones need
        size_t need = 0;
        std::vector<int> found(wants.size(), -1);
        for (size_t w = 0; w < wants.size(); w++) {
            CallScan &s = call.scans[wants[w].scan];
            const uint64_t key = s.id * 2 + wants[w].kind;
            const int hint = wants[w].kind ? s.qcache_hint : s.cache_hint;
            int e = -1;
            if (hint >= 0 && (size_t)hint < m->cache_entries.size() && m->cache_entries[hint].id == key)
                e = hint;
            else {
                auto it = m->cache_index.find(key);
                if (it != m->cache_index.end()) e = it->second;
            }
            if (e >= 0 && m->cache_entries[e].n != s.n) e = -1; // cannot happen (ranges are immutable); be safe
            found[w] = e;
            if (e < 0) need += bytes_of(s, wants[w].kind);
        }
        if (m->cache_used + need > m->cache_arena.cap) {
            if (attempt == 0 && need <= m->cache_limit) {
                // grow (or, at the limit, start over): the arena's contents go, every entry with them
                size_t want = std::max(m->cache_used + need, 2 * m->cache_arena.cap);
                if (want > m->cache_limit) want = std::max(need, std::min(m->cache_limit, 2 * need));
                m->cache_entries.clear();
                m->cache_index.clear();
                m->cache_used = 0;
                m->cache_gen++;
                if (want > m->cache_arena.cap) {
                    HIP_TRY(hipStreamSynchronize(m->stream)); // calls in flight still read the old arena
                    int rc = m->cache_arena.ensure(want);
                    if (rc) return rc;
                }
                continue; // look everything up again: all new now
            }
            // does not fit even alone: cache what fits, project the rest per call
        }
        for (size_t w = 0; w < wants.size(); w++) {
            CallScan &s = call.scans[wants[w].scan];
            const int kind = wants[w].kind;
            const uint64_t key = s.id * 2 + kind;
            int e = found[w];
            int stale = 0;
            if (e < 0) {
                auto it = m->cache_index.find(key); // the same scan may appear in several chains of one call
                if (it != m->cache_index.end()) e = it->second;
            }
            if (e < 0) {
                const size_t bytes = bytes_of(s, kind);
                if (m->cache_used + bytes > m->cache_arena.cap) continue; // uncached
                e = (int)m->cache_entries.size();
                m->cache_entries.push_back(ym_matcher::CacheEntry{key, m->cache_used, s.n, {s.pose[0], s.pose[1], s.pose[2]}, this_call});
                m->cache_index.emplace(key, e);
                m->cache_used += bytes;
                stale = 1;
                m->cache_misses++;
            } else {
                ym_matcher::CacheEntry &ce = m->cache_entries[e];
                if (ce.stale_in_call == this_call) {
                    stale = 1; // (re)computed by this very call: every block that sees the scan computes it
                } else if (ce.pose[0] != s.pose[0] || ce.pose[1] != s.pose[1] || ce.pose[2] != s.pose[2]) {
                    ce.pose[0] = s.pose[0]; ce.pose[1] = s.pose[1]; ce.pose[2] = s.pose[2];
                    ce.stale_in_call = this_call;
                    stale = 1;
                    m->cache_misses++;
                } else {
                    m->cache_hits++;
                }
            }
            unsigned char *p = m->cache_arena.p + m->cache_entries[e].off;
            if (kind) { s.qcache = p; s.qstale = stale; s.qcache_hint = e; }
            else { s.cache = p; s.stale = stale; s.cache_hint = e; }
            if (stale) m->cache_gen++;
        }
        break;
    }
    return YM_OK;
}

// Batches: the work list of points_kernel -- every distinct query once (into its query slot, which the items then
// share) and every base scan whose cache slot this call has to fill once.  Base scans the point cache cannot hold get a
// slot in a per-call scratch arena, so that cells_kernel reads all of them the same way.
int plan_jobs(ym_matcher *m, Slot &slot, CallPlan &P, bool replay = false) {
    Call &call = slot.call;
    P.split_prepare = P.B >= 8;
    if (!P.split_prepare) return YM_OK;
    const int n = (int)call.scans.size();
    auto ensure_lists = [&](int n_q) {
        int rc;
        if (P.region26) { // the region correlate's lists: one per query slot
            if ((rc = m->rg_entries.ensure((size_t)n_q * P.rg_entries_stride))) return rc;
            if ((rc = m->rg_starts.ensure((size_t)n_q * P.rg_starts_stride))) return rc;
            if ((rc = m->rg_rbox.ensure((size_t)n_q * P.rg_nregions * P.rg_parts))) return rc;
#ifdef YM_EXPERIMENTAL
            if (P.rg_ws && (rc = m->rg_walk.ensure((size_t)n_q * P.rg_parts * YM_WS_WALK_WORDS))) return rc;
#endif
        }
        if (P.region) { // the gather correlate's lists: one set per query slot
            if ((rc = m->ga_units.ensure((size_t)n_q * P.ga_units_stride))) return rc;
            if ((rc = m->ga_starts.ensure((size_t)n_q * P.ga_starts_stride))) return rc;
            if ((rc = m->ga_work.ensure((size_t)n_q * P.ga_parts * P.ga_work_stride))) return rc;
            if ((rc = m->ga_counters.ensure((size_t)n_q * 4 * P.ga_nbins2 * YM_GA_CLS))) return rc;
        }
        return (int)YM_OK;
    };
    if (replay) { // (the same call planned the same way: see launch_call_body)
        P.jobs = call.plan_jobs; P.job_slot = call.plan_job_slot; P.qrep = call.plan_qrep;
        P.n_jobs = (int)P.jobs.size();
        P.n_qslots = (int)P.qrep.size();
        return ensure_lists(P.n_qslots);
    }
    std::vector<int> base_used(n, 0), qslot_of(n, -1);
    for (const CallItem &it : call.items)
        for (int j = 0; j < it.base_count; j++) base_used[it.base_begin + j] = 1;
    size_t tmp_need = 0;
    for (int i = 0; i < n; i++)
        if (base_used[i] && !call.scans[i].cache) tmp_need += align_up(YM_CACHE_BYTES(std::max(1, call.scans[i].n)), 16);
    if (tmp_need) {
        int rc = m->tmp_cache.ensure(tmp_need);
        if (rc) return rc;
        size_t at = 0;
        for (int i = 0; i < n; i++)
            if (base_used[i] && !call.scans[i].cache) {
                call.scans[i].cache = m->tmp_cache.p + at;
                call.scans[i].stale = 1;
                at += align_up(YM_CACHE_BYTES(std::max(1, call.scans[i].n)), 16);
            }
    }
    std::vector<int32_t> &jobs = P.jobs, &job_slot = P.job_slot;
    int n_q = 0;
    for (CallItem &it : call.items) {
        if (qslot_of[it.query] < 0) {
            qslot_of[it.query] = n_q++;
            P.qrep.push_back((int32_t)(&it - call.items.data()));
            const CallScan &q = call.scans[it.query];
            if (!q.qcache || q.qstale) { // not already in the point cache at this pose
                jobs.push_back((int32_t)(0x80000000u | (unsigned)it.query));
                job_slot.push_back(qslot_of[it.query]);
            }
        }
        it.qslot = qslot_of[it.query];
    }
    for (int i = 0; i < n; i++)
        if (base_used[i] && call.scans[i].stale) { jobs.push_back(i); job_slot.push_back(0); }
    P.n_jobs = (int)jobs.size();
    P.n_qslots = n_q;
    if (call.batch_uid) { call.plan_jobs = jobs; call.plan_job_slot = job_slot; call.plan_qrep = P.qrep; }
    return ensure_lists(n_q);
}

// the call descriptor: written into pinned host memory; a single match carries it in the kernel arguments, a batch
// gets it by one async H2D copy (hundreds of blocks reading pinned host memory directly is slower)
int plan_descriptor(ym_matcher *m, Slot &slot, CallPlan &P, bool replay = false) {
    const Call &call = slot.call;
    int rc;
    P.scans_bytes = align_up(sizeof(YmScanRef) * P.nscans, 16);
    const size_t items_bytes = align_up(sizeof(YmItem) * P.B, 16);
    P.desc_bytes = P.scans_bytes + items_bytes + sizeof(int32_t) * (2 * (size_t)P.n_jobs + P.qrep.size());
    const unsigned char *pinned_before = slot.desc.p;
    if ((rc = slot.desc.ensure(P.desc_bytes))) return rc;
    if (slot.desc.p != pinned_before) slot.desc_live_bytes = 0;
    if ((rc = slot.result.ensure(align_up(sizeof(YmItemState) * P.B, 64) + 64))) return rc; // (+ the completion word of single matches)
    P.inline_desc = (P.B == 1 && P.nscans <= YM_INLINE_SCANS && !P.split_prepare);
    // The slot's pinned buffer still holds its previous call's descriptor, and the device copy equals it (desc_live_bytes:
    // the slot's previous call is complete -- a slot is handed out again only after it was collected -- so both are free
    // to be rewritten).  A batch that is enqueued again differs in a few records at most (a re-posed scan, a moved cache
    // slot): every record is built in registers and WRITTEN ONLY IF IT DIFFERS, and the 5 MB copy to the device is skipped
    // when none did (round 2 filled the buffer, compared it with a shadow copy and refreshed the shadow: three passes
    // over 5 MB per enqueue of 4096 chains, 1.9 ms of host time).
    const bool live = !P.inline_desc && slot.desc_dev.p && slot.desc_live_bytes == P.desc_bytes;
    bool changed = !live;
    YmScanRef *hs = P.hs = reinterpret_cast<YmScanRef *>(slot.desc.p);
    YmItem *hi = P.hi = reinterpret_cast<YmItem *>(slot.desc.p + P.scans_bytes);
    const bool untouched = replay && live; // (the same Call, planned the same way: every record is what it was)
    for (int i = 0; i < (untouched ? 0 : P.nscans); i++) {
        const CallScan &s = call.scans[i];
        YmScanRef r;
        std::memset(&r, 0, sizeof r);
        r.ranges = s.d_ranges;
        r.n = s.n;
        r.stale = s.stale;
        r.cache = s.cache;
        r.qcache = s.qcache;
        r.qstale = s.qstale;
        r.min_angle = s.min_angle;
        r.angle_inc = s.angle_inc;
        r.min_range = s.min_range;
        r.range_threshold = s.range_threshold;
        r.pose[0] = s.pose[0]; r.pose[1] = s.pose[1]; r.pose[2] = s.pose[2];
        r.pose_dev = s.pose_dev;
        r.gov = s.direct ? s.gov : nullptr;
        r.cidx = s.direct ? s.cidx : nullptr;
        r.cnp = s.direct ? s.cnp : 0;
        if (!live || std::memcmp(&hs[i], &r, sizeof r) != 0) {
            hs[i] = r;
            changed = true;
        }
    }
    for (int i = 0; i < (untouched ? 0 : P.B); i++) {
        const YmItem it = {call.items[i].query, call.items[i].base_begin, call.items[i].base_count, call.items[i].qslot};
        if (!live || std::memcmp(&hi[i], &it, sizeof it) != 0) {
            hi[i] = it;
            changed = true;
        }
    }
    int32_t *hj = reinterpret_cast<int32_t *>(slot.desc.p + P.scans_bytes + items_bytes);
    auto put = [&](int32_t *dst, const int32_t *src, size_t count) {
        if (count && (!live || std::memcmp(dst, src, sizeof(int32_t) * count) != 0)) {
            std::memcpy(dst, src, sizeof(int32_t) * count);
            changed = true;
        }
    };
    put(hj, P.jobs.data(), (size_t)P.n_jobs);
    put(hj + P.n_jobs, P.job_slot.data(), (size_t)P.n_jobs);
    put(hj + 2 * (size_t)P.n_jobs, P.qrep.data(), P.qrep.size());
    if (!P.inline_desc) {
        if (changed) {
            slot.desc_live_bytes = 0;
            if ((rc = slot.desc_dev.ensure(P.desc_bytes))) return rc;
            HIP_TRY(hipMemcpyAsync(slot.desc_dev.p, slot.desc.p, P.desc_bytes, hipMemcpyHostToDevice, m->stream));
            slot.desc_live_bytes = P.desc_bytes;
        }
        P.d_scans = reinterpret_cast<const YmScanRef *>(slot.desc_dev.p);
        P.d_items = reinterpret_cast<const YmItem *>(slot.desc_dev.p + P.scans_bytes);
        P.d_jobs = reinterpret_cast<const int32_t *>(slot.desc_dev.p + P.scans_bytes + items_bytes);
        P.d_job_slot = P.d_jobs + P.n_jobs;
        P.d_qrep = P.d_job_slot + P.n_jobs;
    } else {
        slot.desc_live_bytes = 0; // (the descriptor travels in the kernel arguments; the device copy is not maintained)
    }
    return YM_OK;
}

// which tiles of the window the raster covers in this call (host side; on batches the device builds the work list
// inside that rectangle)
int plan_raster(ym_matcher *m, Slot &slot, CallPlan &P) {
    const Call &call = slot.call;
    const YmGeom &g = P.g;
    const int B = P.B, tiles_x = P.tiles_x, tiles_y = P.tiles_y;
    int rc;
    // the "tile is already zero" flags describe window MEMORY: they survive from call to call while the buffers and
    // the tiling stay the same, otherwise they are cleared
    const size_t per_item = (size_t)tiles_x * tiles_y, ntiles = (size_t)B * per_item;
    // (+ whether the planes are written: after calls that left them out they are stale, and the knowledge below covers both copies)
    const size_t sig[6] = {(size_t)m->grid.p, (size_t)m->planes.p, P.grid_stride, (size_t)g.pitch, (size_t)g.win_w, (size_t)P.tile_h};
    const bool tz_grow = ntiles > m->tile_zero.cap || ntiles * 8 > m->sub_zero.cap; // (the two grow at different sizes)
    if ((rc = m->tile_zero.ensure(ntiles))) return rc;
    if ((rc = m->sub_zero.ensure(ntiles * 8))) return rc;
    if (tz_grow || std::memcmp(sig, m->tz_sig, sizeof sig) != 0) {
        std::memcpy(m->tz_sig, sig, sizeof sig);
        m->tz_covered = 0;
        m->planes_stale.clear();
    }
    if ((int)m->planes_stale.size() < B) m->planes_stale.resize(B, 0);
    if (P.win_only) {
        for (int i = 0; i < B; i++) m->planes_stale[i] = 1; // (their planes are not written by this call)
    } else {
        // this call reads (or at least writes) the planes: an item whose planes lag behind forgets what it knows -- every tile of
        // it is written once, window and planes alike -- in runs of consecutive items
        for (int i = 0; i < std::min(B, m->tz_covered);) {
            if (!m->planes_stale[i]) { i++; continue; }
            int j = i;
            while (j < std::min(B, m->tz_covered) && m->planes_stale[j]) j++;
            HIP_TRY(hipMemsetAsync(m->tile_zero.p + (size_t)i * per_item, 0, (size_t)(j - i) * per_item, m->stream));
            HIP_TRY(hipMemsetAsync(m->sub_zero.p + (size_t)i * per_item * 8, 0, (size_t)(j - i) * per_item * 8, m->stream));
            for (int t = i; t < j; t++) m->item_dirty[t] = {0, 0, tiles_x - 1, tiles_y - 1};
            i = j;
        }
        for (int i = 0; i < B; i++) m->planes_stale[i] = 0;
    }
    if (B > m->tz_covered) { // items this geometry has not seen yet: unknown memory, every tile is launched once
        HIP_TRY(hipMemsetAsync(m->tile_zero.p + (size_t)m->tz_covered * per_item, 0, (size_t)(B - m->tz_covered) * per_item, m->stream));
        HIP_TRY(hipMemsetAsync(m->sub_zero.p + (size_t)m->tz_covered * per_item * 8, 0, (size_t)(B - m->tz_covered) * per_item * 8, m->stream));
        m->item_dirty.resize(B);
        for (int i = m->tz_covered; i < B; i++) m->item_dirty[i] = {0, 0, tiles_x - 1, tiles_y - 1};
        m->tz_covered = B;
    }
    // Tiles a base point can stamp: rotate every base scan's sensor-frame box into the world, take the union over
    // the call, convert to window tiles (+ smear halo, + 1 tile of hysteresis).  Only that sub-grid is launched,
    // extended to the rectangles that may still hold old non-zero bytes in any of this call's items.
    int want[4] = {tiles_x, tiles_y, -1, -1};
    // (a replayed plan of a resident batch: the same poses, the same rectangle -- kept with the call)
    const bool want_known = call.plan_want_valid && call.plan_clean && call.pose_epoch == g_pose_epoch.load(std::memory_order_relaxed) &&
                            call.batch_uid != 0 && call.plan_want_geom[0] == g.win_origin && call.plan_want_geom[1] == g.win_w && call.plan_want_geom[2] == P.tile_h;
    if (want_known) for (int k = 0; k < 4; k++) want[k] = call.plan_want[k];
    else
    for (const CallItem &it : call.items) {
        double wx0 = 1e300, wy0 = 1e300, wx1 = -1e300, wy1 = -1e300; // the chain's boxes joined (kept with the scans' poses)
        for (int j = 0; j < it.base_count; j++) {
            const CallScan &bs = call.scans[it.base_begin + j];
            wx0 = std::min(wx0, bs.wbox[0]); wy0 = std::min(wy0, bs.wbox[1]);
            wx1 = std::max(wx1, bs.wbox[2]); wy1 = std::max(wy1, bs.wbox[3]);
        }
        if (wx0 > wx1) continue; // no usable reading
        const CallScan &q = call.scans[it.query];
        const double offx = q.pose[0] - (0.5 * (g.roi_w - 1) * g.res), offy = q.pose[1] - (0.5 * (g.roi_w - 1) * g.res);
        const double pad = g.half_kernel + 3; // smear reach + rounding slack, in cells
        const double cx0 = (wx0 - offx) / g.res + g.border - g.win_origin - pad, cx1 = (wx1 - offx) / g.res + g.border - g.win_origin + pad;
        const double cy0 = (wy0 - offy) / g.res + g.border - g.win_origin - pad, cy1 = (wy1 - offy) / g.res + g.border - g.win_origin + pad;
        want[0] = std::min(want[0], (int)std::floor(cx0 / YM_TILE_W) - 1); want[2] = std::max(want[2], (int)std::floor(cx1 / YM_TILE_W) + 1);
        want[1] = std::min(want[1], (int)std::floor(cy0 / P.tile_h) - 1); want[3] = std::max(want[3], (int)std::floor(cy1 / P.tile_h) + 1);
    }
    if (call.batch_uid != 0 && !want_known) {
        Call &wc = slot.call;
        for (int k = 0; k < 4; k++) wc.plan_want[k] = want[k];
        wc.plan_want_geom[0] = g.win_origin; wc.plan_want_geom[1] = g.win_w; wc.plan_want_geom[2] = P.tile_h;
        wc.plan_want_valid = true;
    }
    if (call.chain_step) { // (predicted poses: 64 cells more each way; a negative margin, debug option 25, provokes faults)
        const int mg = m->chain_margin;
        want[0] -= mg; want[1] -= 2 * mg; want[2] += mg; want[3] += 2 * mg;
    }
    want[0] = std::max(want[0], 0); want[1] = std::max(want[1], 0);
    want[2] = std::min(want[2], tiles_x - 1); want[3] = std::min(want[3], tiles_y - 1);
    if (want[2] < want[0] || want[3] < want[1]) { want[0] = tiles_x; want[1] = tiles_y; want[2] = want[3] = -1; } // nothing can be stamped
    int *launch = P.launch;
    for (int k = 0; k < 4; k++) launch[k] = want[k];
    for (int i = 0; i < B; i++) {
        const std::array<int, 4> &d = m->item_dirty[i];
        if (d[2] < d[0] || d[3] < d[1]) continue;
        launch[0] = std::min(launch[0], d[0]); launch[1] = std::min(launch[1], d[1]);
        launch[2] = std::max(launch[2], d[2]); launch[3] = std::max(launch[3], d[3]);
    }
    // after this launch only `want` can hold non-zero bytes in the items it covered
    for (int i = 0; i < B; i++) m->item_dirty[i] = {want[0], want[1], want[2], want[3]};
    if (m->full_raster) { launch[0] = launch[1] = 0; launch[2] = tiles_x - 1; launch[3] = tiles_y - 1; }
    P.ltx = std::max(0, launch[2] - launch[0] + 1);
    P.lty = std::max(0, launch[3] - launch[1] + 1);
    if (call.chain_step) {
        // what prepare_kernel checks every kept cell against: a cell whose smear would reach a tile outside `want` (the
        // window's own edge is no limit) is a fault of the step.  `want`, not `launch`: the launch also covers the tiles that
        // may still hold an earlier call's stamps (and, on a window's first call, every tile), but item_dirty above says that
        // after this call only `want` can hold non-zero bytes -- a stamp in launch \ want would never be cleared again.
        const int h = g.half_kernel;
        P.cell_box[0] = want[0] <= 0 ? INT32_MIN : want[0] * YM_TILE_W + h;
        P.cell_box[1] = want[1] <= 0 ? INT32_MIN : want[1] * P.tile_h + h;
        P.cell_box[2] = want[2] >= tiles_x - 1 ? INT32_MAX : (want[2] + 1) * YM_TILE_W - 1 - h;
        P.cell_box[3] = want[3] >= tiles_y - 1 ? INT32_MAX : (want[3] + 1) * P.tile_h - 1 - h;
        if (want[2] < want[0] || want[3] < want[1]) { P.cell_box[0] = P.cell_box[1] = INT32_MAX; P.cell_box[2] = P.cell_box[3] = INT32_MIN; } // nothing may be stamped
    }
    P.tile_cap = std::max(1, P.ltx * P.lty);
    // a work list pays for its extra launch once the items are many
    // (from 48 items on: 256 items 210 -> 159 us, but 8 items 93 us per enqueue with the list against 80 without, 32 items
    //  148 / 144, 64 items 241 / 247)
    P.use_tile_list = B >= m->tile_list_min_batch && P.ltx * P.lty > 0 && tiles_x * tiles_y < 32768;
    if (P.use_tile_list) {
        if ((rc = m->tile_list.ensure((size_t)B * P.tile_cap))) return rc;
        if ((rc = m->tile_count.ensure(B))) return rc;
        if ((rc = m->tile_max.ensure(1))) return rc;
        // hit slots per list entry (YM_TILE_HITS = 64 is four times what the bench scans need of a tall tile; the block of a
        // tile more chunks reach walks the item's boxes itself): a chunk is named by its first cell's index, 16 bits
        P.use_tile_hits = (long long)P.max_base * P.max_n < 65536 && P.tile_cap <= 8192;
        if (m->raster_hits_per_tile < 0) P.use_tile_hits = false;
        if (P.use_tile_hits) {
            if ((rc = m->tile_hits.ensure((size_t)B * P.tile_cap * YM_TILE_HITS))) return rc;
        }
        if (!m->tile_max_host) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&m->tile_max_host), sizeof(int32_t), hipHostMallocDefault));
            *m->tile_max_host = 0;
        }
    }
    return YM_OK;
}
