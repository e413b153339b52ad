// ym_host_enqueue.hpp -- host runtime: the launches of one call, stage by stage
// Part of yagmatch.hip (included inside its anonymous namespace); not a header of its own.
// ---- K1 prepare
void enqueue_prepare(ym_matcher *m, const CallPlan &P) {
    ym::PrepareArgs a;
    a.scans = P.d_scans; a.items = P.d_items; a.g = P.g; a.lat = P.lc; a.states = m->states.p; a.qlocal = m->qlocal.p;
    a.cells = m->cells.p; a.bbox = m->bbox.p; a.ctrig = m->ctrig.p; a.hypcell = m->hypcell.p; a.probs = P.probs;
    a.max_n = P.max_n; a.max_base = P.max_base; a.nt_stride = P.nt_stride; a.dim_stride = P.dim_stride; a.stamps = P.stamps;
    a.use_inline = P.inline_desc ? 1 : 0;
    a.pad0 = 0;
    std::memset(&a.inl, 0, sizeof a.inl);
    if (a.use_inline) { // descriptor travels in the kernel arguments: no host-memory reads on the device
        a.inl.item = P.hi[0];
        for (int i = 0; i < P.nscans; i++) a.inl.scans[i] = P.hs[i];
    }
    a.qnp = m->qnp.p; a.jobs = P.d_jobs; a.job_slot = P.d_job_slot;
    a.fault = nullptr; a.step = 0; a.pad1 = 0;
    for (int k = 0; k < 4; k++) a.cell_box[k] = P.cell_box[k];
    if (P.chain_step) { a.fault = m->seq_fault.p; a.step = P.chain_step; }
    a.tile_max_zero = P.use_tile_list ? m->tile_max.p : nullptr;
    const size_t lds = YM_PREP_LDS_BYTES(P.max_n);
    if (P.split_prepare) {
        if (P.n_jobs > 0) hipLaunchKernelGGL(ym::points_kernel, dim3(P.n_jobs), dim3(YM_POINTS_THREADS), lds, m->stream, a);
        hipLaunchKernelGGL(ym::cells_kernel, dim3(P.max_base + 1, P.B), dim3(256), 0, m->stream, a);
    } else {
        // (query, base scans, item).  One item: 1024 threads per scan -- a 1081-beam scan is then one pass of every phase
        // plus a tail instead of three passes, and the blocks have the chip to themselves
        if (P.B == 1 && m->prepare_threads != 512)
            hipLaunchKernelGGL(ym::prepare_kernel<1024>, dim3(P.max_base + 2, P.B), dim3(1024), lds, m->stream, a);
        else
            hipLaunchKernelGGL(ym::prepare_kernel<512>, dim3(P.max_base + 2, P.B), dim3(512), lds, m->stream, a);
    }
}

// ---- K1b select: Karto's order-dependent "value already set" rule (only when the kernel has 100-valued taps off-centre)
int enqueue_select(ym_matcher *m, const CallPlan &P) {
    if (P.g.zone_count <= 1) return YM_OK;
    const size_t pts = (size_t)P.max_base * P.max_n;
    int log2cap = 10;
    while (((size_t)3 << log2cap) < 4 * pts) log2cap++; // load factor <= 0.75
    if (log2cap > 17 || P.g.storage_w >= 32768)
        return set_err(YM_ERR_UNSUPPORTED, "order-dependent smear (smear_deviation/resolution = %g): chains of more than 98304 readings are not supported",
                       m->cfg.smear_deviation / m->cfg.resolution);
    if (log2cap > 14 || m->select_global) { // too long for one CU's LDS: the same rule with its tables in global memory
        const size_t cap = (size_t)1 << log2cap;
        const int nb = 5; // (z2max <= 1 always: build_geometry)
        int rc = m->sel_scratch.ensure((size_t)P.B * cap * (3 + (nb - 1)));
        if (rc) return rc;
        ym::SelectGlobalArgs g;
        g.cells = m->cells.p; g.max_n = P.max_n; g.max_base = P.max_base; g.z2max = m->z2max; g.log2cap = log2cap;
        g.keys = m->sel_scratch.p; g.status = g.keys + (size_t)P.B * cap; g.minidx = g.status + (size_t)P.B * cap; g.nbr = g.minidx + (size_t)P.B * cap;
        HIP_TRY(hipMemsetAsync(g.keys, 0, (size_t)2 * P.B * cap * sizeof(unsigned), m->stream));
        HIP_TRY(hipMemsetAsync(g.minidx, 0xff, (size_t)P.B * cap * sizeof(unsigned), m->stream));
        const size_t lds = cap; // one byte per slot: the threads' lists of undecided slots
        hipLaunchKernelGGL(ym::select_global_kernel<5>, dim3(P.B), dim3(1024), lds, m->stream, g);
        return YM_OK;
    }
    if (m->z2max <= 1 && P.B <= m->select_split_max) {
        // a few items: the parallel steps (hash, earlier neighbours) as launches over all points, the chain of decisions in one
        // block per item (ym_k_prepare.hpp, select_relax_kernel)
        const size_t cap = (size_t)1 << log2cap;
        const size_t had = m->sel_tables.cap;
        int rc = m->sel_tables.ensure((size_t)2 * P.B * cap);
        if (rc) return rc;
        if (m->sel_tables.cap != had) HIP_TRY(hipMemsetAsync(m->sel_tables.p, 0, m->sel_tables.cap * sizeof(unsigned), m->stream));
        if ((rc = m->sel_rec.ensure((size_t)P.B * 12 * 1024))) return rc;
        if ((rc = m->sel_slot.ensure((size_t)P.B * pts))) return rc;
        ym::SelectSplitArgs s;
        s.cells = m->cells.p; s.max_n = P.max_n; s.max_base = P.max_base; s.log2cap = log2cap; s.pad = 0;
        s.keys = m->sel_tables.p; s.mx = s.keys + (size_t)P.B * cap; s.rec = m->sel_rec.p; s.slot_of = m->sel_slot.p; s.stamps = P.stamps;
        const dim3 grid((unsigned)((pts + YM_SELECT_SPLIT_THREADS - 1) / YM_SELECT_SPLIT_THREADS), P.B);
        hipLaunchKernelGGL(ym::select_hash_kernel, grid, dim3(YM_SELECT_SPLIT_THREADS), 0, m->stream, s);
        hipLaunchKernelGGL(ym::select_neighbours_kernel, grid, dim3(YM_SELECT_SPLIT_THREADS), 0, m->stream, s);
        hipLaunchKernelGGL(ym::select_relax_kernel, dim3(P.B), dim3(1024), cap, m->stream, s);
        return YM_OK;
    }
    ym::SelectArgs a;
    a.cells = m->cells.p; a.max_n = P.max_n; a.max_base = P.max_base; a.z2max = m->z2max; a.log2cap = log2cap; a.stamps = P.stamps;
    const size_t lds = (size_t)9 << log2cap;
    hipLaunchKernelGGL(ym::select_kernel<5>, dim3(P.B), dim3(1024), lds, m->stream, a);
    return YM_OK;
}

// ---- K1c tiles (batches; after select: it reads the boxes only) and K2 raster
int enqueue_raster(ym_matcher *m, const CallPlan &P) {
    hipStream_t st = m->stream;
    const YmGeom &g = P.g;
    if (P.use_tile_list) {
        ym::TilesArgs t;
        t.bbox = m->bbox.p; t.tile_list = m->tile_list.p; t.tile_count = m->tile_count.p; t.tile_zero = m->tile_zero.p;
        t.tile_max = m->tile_max.p;
        t.hits = P.use_tile_hits ? m->tile_hits.p : nullptr; t.tile_h = P.tile_h;
        t.hit_limit = m->raster_hits_per_tile > 0 ? std::min(m->raster_hits_per_tile, YM_TILE_HITS) : YM_TILE_HITS;
        t.max_n = P.max_n; t.max_base = P.max_base; t.half_kernel = g.half_kernel;
        t.tiles_x = P.tiles_x; t.tiles_y = P.tiles_y; t.tile_cap = P.tile_cap;
        for (int k = 0; k < 4; k++) t.launch[k] = P.launch[k];
        hipLaunchKernelGGL(ym::tiles_kernel, dim3(P.B), dim3(YM_TILES_THREADS),
                           (size_t)4 * ((P.tiles_x * P.tiles_y + 31) / 32) + (P.use_tile_hits ? (size_t)8 * P.tile_cap : 0), st, t);
    }
    ym::RasterArgs a;
    a.tiles_x = P.tiles_x; a.tiles_y = P.tiles_y; a.tile_x0 = P.launch[0]; a.tile_y0 = P.launch[1]; a.ltx = P.ltx;
    a.tile_list = P.use_tile_list ? m->tile_list.p : nullptr; a.tile_count = m->tile_count.p; a.tile_cap = P.tile_cap;
    a.cells = m->cells.p; a.bbox = m->bbox.p; a.states = m->states.p; a.g = g; a.grid = m->grid.p;
    a.grid_stride = P.grid_stride; a.planes = m->planes.p; a.lut = m->ktab.p; a.max_n = P.max_n; a.max_base = P.max_base; a.stamps = P.stamps;
    a.tile_zero = m->tile_zero.p; a.sub_zero = m->sub_zero.p; a.planes_only = m->raster_planes_only;
    a.n_rowtab = m->raster_no_rowtab ? 0 : m->n_rowtab; a.rowtab = reinterpret_cast<const uint2 *>(m->rowtab.p); a.rowtab_shift = m->rowtab_shift; a.no_planes = P.win_only ? 1 : 0;
    const size_t rlds = YM_RASTER_LDS_BYTES(P.tile_h, g.half_kernel, a.n_rowtab);
    a.tile_max = m->tile_max.p; a.tile_max_host = P.use_tile_list ? m->tile_max_host : nullptr;
    a.hits = (P.use_tile_list && P.use_tile_hits) ? m->tile_hits.p : nullptr; a.lty = P.lty; a.pad0 = 0;
    int rc;
    hipEvent_t ev_k = nullptr;
    if ((rc = prof_begin(m, 1, &ev_k))) return rc;
    if (P.ltx > 0 && P.lty > 0) {
        // blocks per item: the longest work list an earlier call of this matcher reported (+ 1/8), at most one per tile of
        // the sub-grid; the blocks stride over the list, so a stale or missing number only costs time
        const int hint = m->tile_max_host ? *reinterpret_cast<volatile int32_t *>(m->tile_max_host) : 0;
        const int gx = m->raster_gx > 0 ? std::min(P.ltx * P.lty, m->raster_gx)
                                        : hint > 0 ? std::min(P.ltx * P.lty, hint + hint / 8 + 2) : P.ltx * P.lty;
        a.first_overflow = gx;
        if (P.use_tile_list) {
            if (P.tile_h == YM_TILE_H_TALL) {
                hipLaunchKernelGGL((ym::raster_kernel<128, false, YM_TILE_H_TALL, true>), dim3(gx, P.B), dim3(128), rlds, st, a);
                if (gx < P.ltx * P.lty) hipLaunchKernelGGL((ym::raster_kernel<128, true, YM_TILE_H_TALL, true>), dim3(4, P.B), dim3(128), rlds, st, a);
            } else {
                hipLaunchKernelGGL((ym::raster_kernel<128, false, YM_TILE_H, true>), dim3(gx, P.B), dim3(128), rlds, st, a);
                if (gx < P.ltx * P.lty) hipLaunchKernelGGL((ym::raster_kernel<128, true, YM_TILE_H, true>), dim3(4, P.B), dim3(128), rlds, st, a);
            }
        } else {
            if (P.tile_h == YM_TILE_H_TALL) hipLaunchKernelGGL((ym::raster_kernel<256, false, YM_TILE_H_TALL, false>), dim3(P.ltx * P.lty, P.B), dim3(256), rlds, st, a);
            else hipLaunchKernelGGL((ym::raster_kernel<256, false, YM_TILE_H, false>), dim3(P.ltx * P.lty, P.B), dim3(256), rlds, st, a);
        }
    }
    return prof_end(m, ev_k);
}

int enqueue_correlate(ym_matcher *m, const CallPlan &P);
void enqueue_score(ym_matcher *m, Slot &slot, const CallPlan &P);

// ---- the Python matcher's two find_best_pose passes (scan_matching.py:204-214)
int enqueue_yagpy_passes(ym_matcher *m, Slot &slot, const CallPlan &P) {
    const Call &call = slot.call;
    const YmGeom &g = P.g;
    m->sums_pass_offset[0] = 0;
    m->sums_pass_offset[1] = (size_t)P.B * P.yvol;
    for (int pass = 0; pass < (call.refine ? 2 : 1); pass++) {
        ym::YagArgs a;
        std::memset(&a, 0, sizeof a);
        a.g = g; a.pass = pass; a.penalize = call.penalize; a.refine = call.refine;
        a.last = (pass == 1 || !call.refine) ? 1 : 0;
        if (pass == 0) {
            a.search_xy = m->cfg.search_size * 0.5; a.step_xy = g.res * 2;
            a.search_t = m->cfg.coarse_search_angle_offset * 0.5; a.step_t = m->cfg.coarse_angle_resolution;
        } else {
            a.search_xy = g.res * 2; a.step_xy = g.res; a.search_t = 0.0349 * 0.5; a.step_t = 0.00349;
        }
        a.coarse_angle_res = m->cfg.coarse_angle_resolution;
        a.states = m->states.p; a.host_out = reinterpret_cast<YmItemState *>(slot.result.dp);
        a.qlocal = m->qlocal.p; a.axes = m->yaxes.p; a.rot = nullptr; // (the kernels rotate the points themselves)
        a.sums = m->sums.p + m->sums_pass_offset[pass]; a.out = m->resp.p;
        a.grid = m->grid.p; a.grid_stride = P.grid_stride; a.vol_stride = P.yvol;
        a.max_n = P.max_n; a.maxd = P.ymaxd; a.maxt = P.ymaxt;
        hipLaunchKernelGGL(ym::yag_setup_kernel, dim3(1, P.B), dim3(256), 0, m->stream, a);
        if (pass == 0 && P.lc.nx > 0) {
            // the coarse pass's integer sums from the production correlate kernels (the one the batch size and the lattice select, as in
            // Karto semantics) for every item whose roundings yag_lattice_kernel proves to form a lattice; yag_score_kernel then
            // scores those sums the Python way and computes the other items' itself
            a.lat_nx = P.lc.nx; a.lat_ny = P.lc.ny; a.lat_nt = P.lc.nt; a.step_cells = P.sx;
            a.nt_stride = P.nt_stride; a.dim_stride = P.dim_stride; a.ctrig = m->ctrig.p; a.hypcell = m->hypcell.p;
            a.lsums = m->sums.p + (size_t)2 * P.B * P.yvol; a.lsums_stride = P.sums_c; a.counters = m->yag_counters.p;
            hipLaunchKernelGGL(ym::yag_lattice_kernel, dim3(P.B), dim3(256), 0, m->stream, a);
            int rc = enqueue_correlate(m, P);
            if (rc) return rc;
            enqueue_score(m, slot, P);
        }
        if (pass == 1 && m->yag_fast) {
            // the fine pass by rows (ym_k_yagpy.hpp, yag_fine_kernel); yag_score_kernel then only takes the items it left (a wider fine lattice: none)
            a.fine_rows = m->yag_fast; // (2: tests, the rows byte by byte)
            a.n_items = P.B;
            if (P.B >= 64) hipLaunchKernelGGL(ym::yag_fine_kernel<64>, dim3(8 * P.ymaxt * ((P.B + 7) / 8)), dim3(64), 0, m->stream, a);
            else hipLaunchKernelGGL(ym::yag_fine_kernel<256>, dim3(8 * P.ymaxt * ((P.B + 7) / 8)), dim3(256), 0, m->stream, a);
        }
        hipLaunchKernelGGL(ym::yag_score_kernel, dim3((P.ymaxd * P.ymaxd + 255) / 256, P.ymaxt, P.B), dim3(256), 0, m->stream, a);
        if (P.B >= 16) hipLaunchKernelGGL(ym::yag_reduce_kernel<256>, dim3(P.B), dim3(256), 0, m->stream, a);
        else hipLaunchKernelGGL(ym::yag_reduce_kernel<1024>, dim3(P.B), dim3(1024), 0, m->stream, a);
    }
    return YM_OK;
}

// ---- K4 coarse correlate
ym::RegionArgs region_args(ym_matcher *m, const CallPlan &P) {
    ym::RegionArgs r;
    r.g = P.g; r.lat = P.lc; r.grid = m->grid.p; r.planes = m->planes.p; r.grid_stride = P.grid_stride; r.ctrig = m->ctrig.p;
    r.hypcell = m->hypcell.p; r.states = m->states.p; r.qrep = P.d_qrep; r.entries = m->rg_entries.p; r.entries_stride = P.rg_entries_stride;
    r.starts = m->rg_starts.p; r.starts_stride = P.rg_starts_stride; r.partial = m->partial.p; r.partial_stride = P.partial_stride;
    r.rbox = m->rg_rbox.p; r.rbox_stride = (size_t)P.rg_nregions * P.rg_parts; r.nw = P.rg_nw; r.parts = P.rg_parts;
    r.nt_stride = P.nt_stride; r.dim_stride = P.dim_stride; r.nrx = P.rg_nrx; r.nry = P.rg_nry; r.ng = P.rg_ng; r.nbins = P.rg_nbins;
    r.force_irregular = (m->corr_region == 2 || m->corr_region == 3) ? m->corr_region - 1 : 0; r.pad = m->corr_region_dbg; r.stamps = P.stamps;
    r.fuse_score = P.fuse_score ? 1 : 0; r.resp = P.resp; r.sums_stride = P.sums_c; r.blockmax = m->blockmax.p;
    r.probs = P.probs; r.probs_stride = (size_t)P.lc.nx * P.lc.ny; r.n_blocks = P.score_blocks;
    r.rg_h = YM_RG_H; r.rg_cls = YM_RG_CLS; r.rg_zero = YM_RG_ZERO; r.pad2 = 0;
#ifdef YM_EXPERIMENTAL
    r.rg_h = P.rg_ws ? YM_WS_H : YM_RG_H; r.rg_cls = P.rg_ws ? YM_WS_CLS : P.rg_item ? YM_IT_CLS : YM_RG_CLS;
    r.rg_zero = P.rg_ws ? YM_WS_ZERO : P.rg_item ? YM_IT_ZERO : YM_RG_ZERO;
#endif
    if (P.rg2) { r.rg_h = P.rg2_h; r.rg_cls = YM_RG_PITCH * (P.rg2_h + 26); r.rg_zero = 4 * r.rg_cls; }
    r.pad2 = ((1 << 21) + r.rg_h - 1) / r.rg_h; // bin_kernel: class row / region height as a multiplication (region_entry)
    r.rg_w = 0; r.rg_pitch = YM_RG_PITCH; r.nregions = P.rg_nregions; r.pad3 = 0;
    r.walk = m->rg_walk.p; r.nitems = P.B; r.rsplit = P.rg_rsplit; r.pad4 = 0;
    r.lnw = P.rg_lnw; r.lparts = P.rg_lparts; r.entries_pstride = P.rg_entries_pstride;
    // teams of `parts` blocks per XCD: two blocks per CU, no more teams than the XCD gets items
    r.gpx = std::max(1, std::min((2 * std::max(m->n_cus, 8) / 8) / std::max(1, P.rg_parts), (P.B + 7) / 8));
    return r;
}

// bin_kernel, once per query slot of the call: after the prepare stage (item states, hypothesis cells, angle tables)
int enqueue_region_lists(ym_matcher *m, const CallPlan &P, hipStream_t st) {
    const ym::RegionArgs r = region_args(m, P);
#ifdef YM_EXPERIMENTAL
    if (P.rg_lparts == 1 && P.rg_lnw == P.lc.nt && (P.rg_ws || P.rg2 || P.rg_item || P.rg_pool)) { // the round-5 layout: one list of all angles
        const size_t bin_lds = YM_BIN_LDS_BYTES(P.rg_nbins, P.rg_entries_stride, P.rg_nregions * P.rg_parts);
        if (bin_lds > m->bin_lds_limit) { // (more than the default 64 KB of dynamic LDS has to be asked for)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(ym::bin_whole_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bin_lds));
            m->bin_lds_limit = bin_lds;
        }
        hipLaunchKernelGGL(ym::bin_whole_kernel<false>, dim3(P.n_qslots), dim3(YM_BIN_THREADS), bin_lds, st, r);
        if (P.rg_ws) hipLaunchKernelGGL(ym::region_walk_kernel, dim3(P.rg_parts, P.n_qslots), dim3(64), 0, st, r);
        return YM_OK;
    }
#endif
    const size_t bin_lds = YM_BINP_LDS_BYTES(P.rg_nbins, P.rg_entries_pstride, P.rg_nregions);
    if (bin_lds > m->binp_lds_limit) { // (more than the default 64 KB of dynamic LDS has to be asked for)
#define YM_BINP_ATTR(Y, M) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(ym::bin_kernel<Y, M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bin_lds))
        YM_BINP_ATTR(false, 18); YM_BINP_ATTR(false, 32); YM_BINP_ATTR(false, 64); YM_BINP_ATTR(true, 18); YM_BINP_ATTR(true, 32); YM_BINP_ATTR(true, 64);
#undef YM_BINP_ATTR
        m->binp_lds_limit = bin_lds;
    }
    // pairs per thread: the instantiation with registers for them (18: scans of up to 1152 readings at eight angles per part)
    const int per_thread = (P.rg_lnw * P.max_n + YM_BINP_THREADS - 1) / YM_BINP_THREADS;
    const dim3 bgrid(P.rg_lparts, P.n_qslots);
#define YM_BINP_LAUNCH(Y, M) hipLaunchKernelGGL((ym::bin_kernel<Y, M>), bgrid, dim3(YM_BINP_THREADS), bin_lds, st, r)
    if (P.yag) { if (per_thread <= 18) YM_BINP_LAUNCH(true, 18); else if (per_thread <= 32) YM_BINP_LAUNCH(true, 32); else YM_BINP_LAUNCH(true, 64); }
    else { if (per_thread <= 18) YM_BINP_LAUNCH(false, 18); else if (per_thread <= 32) YM_BINP_LAUNCH(false, 32); else YM_BINP_LAUNCH(false, 64); }
#undef YM_BINP_LAUNCH
    return YM_OK;
}

// the lists on the matcher's second stream: fork here (the prepare stage is enqueued), join in enqueue_correlate
int enqueue_region_lists_aside(ym_matcher *m, CallPlan &P) {
    if (!m->side_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking));
        pool_register_stream(m->device, m->side_stream, true);
        HIP_TRY(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(m->ev_fork, m->stream)); // (after the prepare stage; the lists themselves are enqueued by
    return YM_OK;                                   //  enqueue_region_lists_joined, once the raster's launches are out)
}
int enqueue_region_lists_joined(ym_matcher *m, CallPlan &P) {
    HIP_TRY(hipStreamWaitEvent(m->side_stream, m->ev_fork, 0));
    int rc = enqueue_region_lists(m, P, m->side_stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(m->ev_join, m->side_stream));
    P.lists_on_side_stream = true;
    return YM_OK;
}

int enqueue_correlate(ym_matcher *m, const CallPlan &P) {
    hipStream_t st = m->stream;
    ym::CorrArgs a;
    a.g = P.g; a.lat = P.lc; a.grid = m->grid.p; a.grid_stride = P.grid_stride; a.planes = m->planes.p; a.ctrig = m->ctrig.p;
    a.qlocal = m->qlocal.p; a.hypcell = m->hypcell.p; a.states = m->states.p; a.partial = m->partial.p; a.partial_stride = P.partial_stride;
    a.max_n = P.max_n; a.nt_stride = P.nt_stride; a.dim_stride = P.dim_stride; a.chunk = P.chunk; a.n_chunks = P.n_chunks;
    a.ngx = P.ngx; a.nx_pad = P.nx_pad; a.sx = P.sx; a.stamps = P.stamps; a.tpb = P.tpb; a.cw = P.cw;
    a.k_begin = P.k_begin; a.nk = std::max(0, P.k_end - P.k_begin);
    a.dedup = P.dedup; a.pad2 = 0;
    if (a.nk == 0) return YM_OK; // an empty angle slice
    int rc;
    hipEvent_t ev_k = nullptr;
    m->last_corr_form = P.region26 ? 1 : P.region ? 2 : 0;
    if (P.region26) {
        const ym::RegionArgs r = region_args(m, P);
        if (P.lists_on_side_stream) HIP_TRY(hipStreamWaitEvent(st, m->ev_join, 0));
        else if (!P.lists_cached && (rc = enqueue_region_lists(m, P, st))) return rc;
        if ((rc = prof_begin(m, 0, &ev_k))) return rc;
        const dim3 rgrid(P.rg_parts * P.rg_rsplit, P.B);
#ifdef YM_EXPERIMENTAL
        if (P.rg2) {
            const size_t lds = (size_t)r.rg_zero + 26 * YM_RG_PITCH + 32 + (size_t)m->corr_region_pad_lds;
            if (lds > m->rg2_lds_limit) { // (more than the default 64 KB of dynamic LDS has to be asked for)
                const int want = 144 * 1024; // (the 160 KB of a CU less the kernel's static 15 KB)
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(ym::correlate_region2_kernel<80>), hipFuncAttributeMaxDynamicSharedMemorySize, want));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(ym::correlate_region2_kernel<100>), hipFuncAttributeMaxDynamicSharedMemorySize, want));
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(ym::correlate_region2_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, want));
                m->rg2_lds_limit = (size_t)want;
            }
            if (P.rg2_h == 80) hipLaunchKernelGGL(ym::correlate_region2_kernel<80>, rgrid, dim3(64 * YM_R2_NW), lds, st, r);
            else if (P.rg2_h == 100) hipLaunchKernelGGL(ym::correlate_region2_kernel<100>, rgrid, dim3(64 * YM_R2_NW), lds, st, r);
            else hipLaunchKernelGGL(ym::correlate_region2_kernel<128>, rgrid, dim3(64 * YM_R2_NW), lds, st, r);
            return prof_end(m, ev_k);
        }
#endif
#ifdef YM_EXPERIMENTAL // (the three forms that lost to correlate_region_kernel: profiles/r04_region_study.md; option 32 refuses them otherwise)
        if (P.rg_item) {
            const size_t lds = YM_IT_ACC_BYTES(P.lc.nt);
            if (!m->item_lds_set) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(ym::correlate_item_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)YM_IT_ACC_BYTES(YM_IT_MAX_NT)));
                m->item_lds_set = true;
            }
            hipLaunchKernelGGL(ym::correlate_item_kernel, dim3(P.B), dim3(64 * YM_IT_NW), lds, st, r);
            return prof_end(m, ev_k);
        }
        if (P.rg_pool) {
            if (P.win_only) hipLaunchKernelGGL(ym::correlate_pool_kernel<true>, rgrid, dim3(64 * YM_PL_NW), 0, st, r);
            else hipLaunchKernelGGL(ym::correlate_pool_kernel<false>, rgrid, dim3(64 * YM_PL_NW), 0, st, r);
            return prof_end(m, ev_k);
        }
        if (P.rg_ws) {
            hipLaunchKernelGGL(ym::correlate_region_ws_kernel, dim3(8 * r.gpx * P.rg_parts), dim3(64 * (YM_WS_NG + YM_WS_NL)), 0, st, r);
            hipLaunchKernelGGL(ym::region_percell_kernel, rgrid, dim3(64 * YM_WS_NG), 0, st, r);
            return prof_end(m, ev_k);
        }
#endif
        switch (P.rg_nw) {
        case 4: hipLaunchKernelGGL(ym::correlate_region_kernel<4>, rgrid, dim3(256), 0, st, r); break;
        case 5: hipLaunchKernelGGL(ym::correlate_region_kernel<5>, rgrid, dim3(320), 0, st, r); break;
        case 6: hipLaunchKernelGGL(ym::correlate_region_kernel<6>, rgrid, dim3(384), 0, st, r); break;
        case 7: hipLaunchKernelGGL(ym::correlate_region_kernel<7>, rgrid, dim3(448), 0, st, r); break;
        case 10: hipLaunchKernelGGL(ym::correlate_region_kernel<10>, rgrid, dim3(640), 0, st, r); break;
        case 11: hipLaunchKernelGGL(ym::correlate_region_kernel<11>, rgrid, dim3(704), 0, st, r); break;
        case 16: hipLaunchKernelGGL(ym::correlate_region_kernel<16>, rgrid, dim3(1024), 0, st, r); break;
        default:
            if (P.win_only) hipLaunchKernelGGL((ym::correlate_region_kernel<8, true>), rgrid, dim3(512), (size_t)m->corr_region_pad_lds, st, r);
            else hipLaunchKernelGGL(ym::correlate_region_kernel<8>, rgrid, dim3(512), (size_t)m->corr_region_pad_lds, st, r);
            break;
        }
        return prof_end(m, ev_k);
    }
    if (P.region) {
        ym::GatherArgs r;
        std::memset(&r, 0, sizeof r);
        r.g = P.g; r.lat = P.lc; r.grid = m->grid.p; r.planes = m->planes.p; r.grid_stride = P.grid_stride; r.ctrig = m->ctrig.p;
        r.hypcell = m->hypcell.p; r.states = m->states.p; r.qrep = P.d_qrep;
        r.units = m->ga_units.p; r.units_stride = P.ga_units_stride; r.starts = m->ga_starts.p; r.starts_stride = P.ga_starts_stride;
        r.work = m->ga_work.p; r.work_stride = P.ga_work_stride; r.counters = m->ga_counters.p; r.lane_job = m->ga_lane_job.p;
        r.partial = m->partial.p; r.partial_stride = P.partial_stride; r.nt_stride = P.nt_stride; r.dim_stride = P.dim_stride;
        r.W = P.ga_W; r.H = P.ga_H; r.P = P.ga_P; r.rows = P.ga_rows; r.nrx = P.ga_nrx; r.nry = P.ga_nry; r.nseg = P.ga_nseg; r.NP = P.ga_np;
        r.ng = P.ga_ng; r.parts = P.ga_parts; r.kpp = P.ga_kpp; r.unit_cap = P.ga_cap;
        r.force_irregular = (m->corr_region == 2 || m->corr_region == 3) ? m->corr_region - 1 : 0; r.stamps = P.stamps;
        r.sums = P.yag ? m->sums.p + (size_t)2 * P.B * P.yvol : m->keep_sums ? m->sums.p : nullptr; r.resp = P.resp; r.sums_stride = P.sums_c; r.blockmax = m->blockmax.p;
        r.probs = P.probs; r.probs_stride = (size_t)P.lc.nx * P.lc.ny; r.n_blocks = P.score_blocks;
        // the lists: once per query slot of the call (they depend on the query alone)
        HIP_TRY(hipMemsetAsync(m->ga_counters.p, 0, (size_t)P.n_qslots * 4 * P.ga_nbins2 * YM_GA_CLS * sizeof(uint32_t), st));
        const dim3 pgrid(((unsigned)P.lc.nt * P.max_n + YM_GBIN_THREADS - 1) / YM_GBIN_THREADS, P.n_qslots);
        hipLaunchKernelGGL(ym::gbin_pieces_kernel<false>, pgrid, dim3(YM_GBIN_THREADS), 0, st, r);
        hipLaunchKernelGGL(ym::gbin_scan_kernel, dim3(P.n_qslots), dim3(1024), 0, st, r);
        hipLaunchKernelGGL(ym::gbin_pieces_kernel<true>, pgrid, dim3(YM_GBIN_THREADS), 0, st, r);
        if (P.ga_lds > m->ga_lds_limit) { // (more than the default 64 KB of dynamic LDS has to be asked for)
            const int want = (int)std::min<size_t>(160 * 1024, P.ga_lds);
            // (the instantiations for three blocks of eight waves per CU -- 80 VGPRs -- park the staging registers in scratch: 11 GB of
            //  scratch traffic per launch of 4096 loop-lattice items for 4 % of the kernel's time; only in builds with -DYM_EXPERIMENTAL)
#ifdef YM_EXPERIMENTAL
#define YM_GA_512(NA, NP) reinterpret_cast<const void *>(ym::gather_kernel<NA, NP, YM_GA_PER, 512>)
#else
#define YM_GA_512(NA, NP) reinterpret_cast<const void *>(ym::gather_kernel<NA, NP, YM_GA_PER>)
#endif
            const void *kernels[21] = {
#define YM_GA_BOTH(NA, NP) reinterpret_cast<const void *>(ym::gather_kernel<NA, NP, YM_GA_PER>), YM_GA_512(NA, NP), reinterpret_cast<const void *>(ym::gather_percell_kernel<NA, NP>)
                YM_GA_BOTH(1, 1), YM_GA_BOTH(2, 1), YM_GA_BOTH(3, 1), YM_GA_BOTH(4, 1), YM_GA_BOTH(1, 2), YM_GA_BOTH(2, 2), YM_GA_BOTH(1, 3)};
#undef YM_GA_BOTH
#undef YM_GA_512
            for (const void *k : kernels) HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, want));
            m->ga_lds_limit = P.ga_lds;
        }
        if ((rc = prof_begin(m, 0, &ev_k))) return rc;
        const dim3 rgrid(P.ga_parts, P.B), rblock(64 * P.ga_nwv);
        // gather_kernel takes the items whose lists exist and whose hypothesis cells form a lattice (all of them, but for fp
        // rounding accidents and oversized lists), gather_percell_kernel the others: each returns at once from the other's
        // items
#ifdef YM_EXPERIMENTAL
#define YM_GA_LAUNCH_512(NA, NP) if (P.ga_nwv <= 8 && m->corr_region_form == 6) hipLaunchKernelGGL((ym::gather_kernel<NA, NP, YM_GA_PER, 512>), rgrid, rblock, P.ga_lds, st, r); else
#else
#define YM_GA_LAUNCH_512(NA, NP)
#endif
#define YM_GA_LAUNCH(NA, NP)                                                                                               \
    do {                                                                                                                   \
        YM_GA_LAUNCH_512(NA, NP)                                                                                           \
        hipLaunchKernelGGL((ym::gather_kernel<NA, NP, YM_GA_PER>), rgrid, rblock, P.ga_lds, st, r);                        \
        hipLaunchKernelGGL((ym::gather_percell_kernel<NA, NP>), rgrid, rblock, P.ga_lds, st, r);                           \
    } while (0)
        if (P.ga_np == 1) {
            if (P.ga_na == 1) YM_GA_LAUNCH(1, 1);
            else if (P.ga_na == 2) YM_GA_LAUNCH(2, 1);
            else if (P.ga_na == 3) YM_GA_LAUNCH(3, 1);
            else YM_GA_LAUNCH(4, 1);
        } else if (P.ga_np == 2) {
            if (P.ga_na == 1) YM_GA_LAUNCH(1, 2);
            else YM_GA_LAUNCH(2, 2);
        } else YM_GA_LAUNCH(1, 3);
#undef YM_GA_LAUNCH
#undef YM_GA_LAUNCH_512
        return prof_end(m, ev_k);
    }
    if ((rc = prof_begin(m, 0, &ev_k))) return rc;
    const dim3 grid_dim(P.job_blocks, a.nk * P.n_groups, P.B);
    const size_t pad_lds = (size_t)m->corr_pad_lds;
    {
#define YM_CORR_LAUNCH(SX, U, CW) hipLaunchKernelGGL((ym::correlate_kernel<SX, U, CW>), grid_dim, dim3(YM_CORR_THREADS), pad_lds, st, a)
#define YM_CORR_BY_CW(SX, U)                                  \
    do {                                                      \
        if (P.cw == 4) YM_CORR_LAUNCH(SX, U, 4);              \
        else if (P.cw == 2) YM_CORR_LAUNCH(SX, U, 2);         \
        else YM_CORR_LAUNCH(SX, U, 1);                        \
    } while (0)
        if (P.sx == 2 && P.corr_u == 16) YM_CORR_BY_CW(2, 16);
        else if (P.sx == 2 && P.corr_u == 32) YM_CORR_BY_CW(2, 32);
        else if (P.sx == 2) YM_CORR_LAUNCH(2, 48, 1);
        else YM_CORR_BY_CW(1, 16);
#undef YM_CORR_BY_CW
#undef YM_CORR_LAUNCH
    }
    return prof_end(m, ev_k);
}

// ---- K5 score, then the finish stage: fine_kernel (coarse arg-max/mean + 3x3 fine lattice, one block per fine angle)
// + final_kernel (covariances, fine arg-max/mean) for a few items, the one-block finish_kernel on batches; results
// land in pinned host memory
void enqueue_score(ym_matcher *m, Slot &slot, const CallPlan &P) {
    hipStream_t st = m->stream;
    const YmLattice &lc = P.lc;
    if (!P.yag) {
        m->sums_pass_offset[0] = 0;
        m->sums_pass_offset[1] = (size_t)P.B * P.sums_c;
    }
    ym::ScoreArgs a;
    a.g = P.g; a.lat = lc; a.partial = m->partial.p; a.partial_stride = P.partial_stride; a.states = m->states.p;
    // (the integer sums are kept for ym_debug_sums on a few items of an ordinary lattice; on configs[4]'s 1.86 million hypotheses
    //  they are a sixth of this stage's writes: debug option 12 keeps them there too)
    a.sums = (m->keep_sums || (P.B < 8 && P.sums_c <= 65536)) ? m->sums.p : nullptr;
    if (P.yag) a.sums = m->sums.p + (size_t)2 * P.B * P.yvol; // (the launch lattice's sums: what yag_score_kernel scores the Python way)
    a.sums_stride = P.sums_c; a.resp = P.resp; a.blockmax = m->blockmax.p;
    a.n_chunks = P.n_groups; a.nx_pad = P.nx_pad; a.n_blocks = P.score_blocks; a.stamps = P.stamps;
    a.probs = P.probs; a.probs_stride = (size_t)lc.nx * lc.ny;
    a.k_begin = P.k_begin; a.k_end = P.k_end; a.lane_layout = P.region26 ? 1 : 0;
    a.write_blockmax = slot.call.slice ? 0 : 1; // a slice's maxima are recomputed once the volume is whole
    if (P.region || P.fuse_score) return; // the LDS correlates score their sums themselves
    // (a thread of score_kernel walks all angles of its cell: fine when the batch fills the chip, 36 us on 8 items, where
    //  one thread per hypothesis takes 5)
    //  (a region correlate whose regions were dealt out to several blocks -- a small batch -- leaves its sets to this stage too)
    if (P.B >= 256 || (P.B >= 64 && !P.region26)) hipLaunchKernelGGL(ym::score_kernel, dim3(P.cell_blocks, P.B), dim3(YM_SCORE_THREADS), 0, st, a);
    else if (P.k_end > P.k_begin)
        hipLaunchKernelGGL(ym::score_hyp_kernel, dim3(P.cell_blocks, P.k_end - P.k_begin, P.B), dim3(YM_SCORE_THREADS), 0, st, a);
}

void enqueue_finish(ym_matcher *m, Slot &slot, const CallPlan &P) {
    hipStream_t st = m->stream;
    const Call &call = slot.call;
    const YmLattice &lc = P.lc, &lf = P.lf;
    ym::FinishArgs a;
    a.g = P.g; a.lc = lc; a.lf = lf; a.refine = call.refine; a.max_n = P.max_n; a.nt_stride = lf.nt;
    a.n_blocks = P.score_blocks; a.states = m->states.p;
    a.host_out = reinterpret_cast<YmItemState *>(slot.result.dp);
    a.resp = P.resp; a.sums_stride = P.sums_c; a.blockmax = m->blockmax.p; a.probs = P.probs;
    a.probs_stride = (size_t)lc.nx * lc.ny; a.grid = m->grid.p; a.grid_stride = P.grid_stride;
    a.qlocal = m->qlocal.p; a.foffsets = m->foffsets.p; a.fsums = m->sums.p + m->sums_pass_offset[1];
    a.fsums_stride = P.sums_f; a.stamps = P.stamps;
    a.host_flag = nullptr; a.serial = 0; a.pad1 = 0;
    a.seq_pose = nullptr; a.seq_prior = nullptr; a.fault = nullptr; a.next_diff[0] = a.next_diff[1] = a.next_diff[2] = 0.0; a.step = 0; a.expansion = 0;
    if (call.chain_step) {
        a.host_out = call.chain_out;
        a.seq_pose = call.chain_pose_out; a.seq_prior = m->seq_pose.p; a.fault = m->seq_fault.p; a.step = call.chain_step;
        for (int k = 0; k < 3; k++) a.next_diff[k] = call.chain_next_diff[k];
        a.expansion = (m->cfg.semantics == YM_SEM_KARTO && m->cfg.use_response_expansion) ? 1 : 0;
    }
    slot.poll_serial = 0;
    // (one block per item from 128 items on; below, a block per fine angle and item is faster: 8 items 124 against 142 us per
    //  enqueue, 64 items 241 against 254, 128 equal, 256 items 506 against 482)
    if ((P.B >= 128 && m->finish_form != 1) || m->finish_form == 2) {
        const size_t lds = YM_FINISH_LDS_BYTES(call.refine ? (size_t)lf.nx * lf.ny * lf.nt : 0);
        const bool small_blocks = m->finish_threads ? m->finish_threads == 256 : P.B >= 512;
        if (small_blocks) hipLaunchKernelGGL(ym::finish_kernel<256>, dim3(P.B), dim3(256), lds, st, a);
        else hipLaunchKernelGGL(ym::finish_kernel<1024>, dim3(P.B), dim3(1024), lds, st, a);
    } else {
        if (lc.nx * lc.ny > 8 * YM_CANON) hipLaunchKernelGGL(ym::fine_kernel<true>, dim3(call.refine ? lf.nt + 1 : 1, P.B), dim3(YM_FINE_THREADS), 0, st, a);
        else hipLaunchKernelGGL(ym::fine_kernel<false>, dim3(call.refine ? lf.nt + 1 : 1, P.B), dim3(YM_FINE_THREADS), 0, st, a);
        if (P.B == 1 && m->poll_completion && !call.chain_step) { // the caller polls a word final_kernel writes after the result (no stream event to wait for)
            if (++slot.serial_counter == 0) slot.serial_counter = 1;
            slot.poll_serial = a.serial = slot.serial_counter;
            a.host_flag = reinterpret_cast<uint32_t *>(slot.result.dp + align_up(sizeof(YmItemState) * P.B, 64));
        }
        hipLaunchKernelGGL(ym::final_kernel, dim3(P.B), dim3(YM_FINISH_THREADS), 0, st, a);
    }
}
