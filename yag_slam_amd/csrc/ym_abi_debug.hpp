// ym_abi_debug.hpp -- C ABI: introspection for the parity tests, development options, profiling, counters
// Part of yagmatch.hip (included inside its extern "C" block); not a header of its own.
// ---- debug getters
int ym_debug_grid_info(ym_matcher *m, int item, ym_grid_info *info) {
    if (!m || !info) return set_err(YM_ERR_INVALID, "null argument");
    if (!m->last_valid || item < 0 || item >= m->last_B) return set_err(YM_ERR_INVALID, "no such item in the last call");
    const YmGeom &g = m->last_geom;
    info->width = g.win_w; info->height = g.win_w; info->pitch = g.pitch;
    info->origin_x = g.win_origin; info->origin_y = g.win_origin;
    info->storage_w = g.storage_w; info->storage_h = g.storage_w;
    info->roi_x = g.border; info->roi_y = g.border; info->roi_w = g.roi_w; info->roi_h = g.roi_w;
    YmItemState s;
    DEV_GUARD(m->device);
    HIP_TRY(hipStreamSynchronize(m->stream));
    HIP_TRY(hipMemcpy(&s, m->states.p + item, sizeof s, hipMemcpyDeviceToHost));
    info->offset_x = s.off_x; info->offset_y = s.off_y;
    return YM_OK;
}

int ym_debug_grid(ym_matcher *m, int item, uint8_t *out, int64_t out_bytes) {
    if (!m || !out) return set_err(YM_ERR_INVALID, "null argument");
    if (!m->last_valid || item < 0 || item >= m->last_B) return set_err(YM_ERR_INVALID, "no such item in the last call");
    const int64_t need = (int64_t)m->last_geom.pitch * m->last_geom.win_w;
    if (out_bytes < need) return set_err(YM_ERR_INVALID, "grid buffer too small: need %lld bytes", (long long)need);
    DEV_GUARD(m->device);
    HIP_TRY(hipStreamSynchronize(m->stream));
    HIP_TRY(hipMemcpy(out, m->grid.p + (size_t)item * m->last_grid_stride, (size_t)need, hipMemcpyDeviceToHost));
    return YM_OK;
}

int ym_debug_sums(ym_matcher *m, int item, int pass, uint32_t *out, int64_t out_count) {
    if (!m || !out || pass < 0 || pass > 1) return set_err(YM_ERR_INVALID, "bad argument");
    if (!m->last_valid || item < 0 || item >= m->last_B) return set_err(YM_ERR_INVALID, "no such item in the last call");
    const size_t n = m->last_sums_stride[pass];
    if (n == 0) return set_err(YM_ERR_INVALID, "pass %d did not run", pass);
    const size_t ncopy = std::min(n, (size_t)out_count); // a pass's volume is stored dense from the start of its slot
    if (m->cfg.semantics == YM_SEM_KARTO && (size_t)out_count < n)
        return set_err(YM_ERR_INVALID, "sums buffer too small: need %zu entries", n);
    DEV_GUARD(m->device);
    HIP_TRY(hipStreamSynchronize(m->stream));
    HIP_TRY(hipMemcpy(out, m->sums.p + m->sums_pass_offset[pass] + (size_t)item * n, ncopy * sizeof(uint32_t),
                      hipMemcpyDeviceToHost));
    return YM_OK;
}

int ym_debug_query_local(ym_matcher *m, int item, double *out_xy, int32_t cap, int32_t *n) {
    if (!m || !out_xy || !n) return set_err(YM_ERR_INVALID, "null argument");
    if (!m->last_valid || item < 0 || item >= m->last_B) return set_err(YM_ERR_INVALID, "no such item in the last call");
    YmItemState s;
    DEV_GUARD(m->device);
    HIP_TRY(hipStreamSynchronize(m->stream));
    HIP_TRY(hipMemcpy(&s, m->states.p + item, sizeof s, hipMemcpyDeviceToHost));
    *n = s.nq;
    if (cap < s.nq) return set_err(YM_ERR_INVALID, "buffer too small: need %d points", s.nq);
    if (s.nq > 0)
        HIP_TRY(hipMemcpy(out_xy, s.ql, sizeof(double2) * s.nq, hipMemcpyDeviceToHost));
    return YM_OK;
}

int ym_debug_cells(ym_matcher *m, int item, int32_t *out, int64_t out_count, int32_t *max_n) {
    if (!m || !max_n) return set_err(YM_ERR_INVALID, "null argument");
    if (!m->last_valid || item < 0 || item >= m->last_B) return set_err(YM_ERR_INVALID, "no such item in the last call");
    *max_n = m->last_max_n;
    const size_t per = (size_t)m->last_max_base * m->last_max_n;
    if (!out) return YM_OK;
    if ((size_t)out_count < per * 2) return set_err(YM_ERR_INVALID, "buffer too small: need %zu ints", per * 2);
    DEV_GUARD(m->device);
    HIP_TRY(hipStreamSynchronize(m->stream));
    HIP_TRY(hipMemcpy(out, m->cells.p + (size_t)item * per, sizeof(int2) * per, hipMemcpyDeviceToHost));
    return YM_OK;
}

// The options the parity tests use are tabulated in include/yagmatch.h.  The others -- development and timing switches:
//    2  rasterise every tile of the window                      3  beams in flight per lane in the direct correlate (16 / 32 / 48)
//    4  extra dynamic LDS bytes per direct-correlate block        5  beam chunks per angle of the direct correlate
//    9  chunk-waves per direct-correlate block (1, 2, 4)        23  0 = single matches wait for a stream event instead of polling
//   26  512 = the single-item prepare kernel with 512 threads   29  0 = the region path's pair lists on the call's own stream
//   33  experimental forms: loader / gather waves idle (timing) 34  blocks that share an (item, angle block)'s regions (0 = by batch size)
//   35  batch size from which experimental form 3 is default    36  the raster does not write the row-major window (timing only)
//   38  unused dynamic LDS bytes per region-correlate block     40  batch size from which batches get raster work lists (48)
//   42  the region correlate's threshold alone (0 = 48)         44  batch size from which experimental form 5 is default
int ym_debug_option(ym_matcher *m, int option, int value) {
    if (m) { m->cache_gen++; m->list_key_valid = false; } // (whatever the option changes, no earlier plan or pair list is reused)
    if (!m) return set_err(YM_ERR_INVALID, "null matcher");
    if (option == 0) return set_err(YM_ERR_INVALID, "debug option 0 (an experimental correlate form) no longer exists");
    else if (option == 2) m->full_raster = value;
    else if (option == 3) m->corr_u = value;
    else if (option == 4) m->corr_pad_lds = value;
    else if (option == 5) m->corr_chunks = value;
    else if (option == 6) m->finish_form = value;
    else if (option == 9) m->corr_cw = value;
    else if (option == 10) m->select_global = value;
    else if (option == 41) m->select_split_max = value;
    else if (option == 11) m->finish_threads = value;
    else if (option == 12) m->keep_sums = value;
    else if (option == 13) m->corr_dedup = value;
    else if (option == 14) m->corr_region = value;
    else if (option == 15) m->corr_region_na = m->corr_region_nw = value;
    else if (option == 23) m->poll_completion = value != 0;
    else if (option == 24) m->use_scan_structure = value != 0;
    else if (option == 25) m->chain_margin = value;
    else if (option == 26) m->prepare_threads = value;
    else if (option == 28) { // both LDS correlates from `value` items on (at least 8); 0 = the defaults again (gather 64, region 48)
        if (value == 0) { m->lds_min_batch = 64; m->rg_min_batch = 48; }
        else m->lds_min_batch = m->rg_min_batch = std::max(8, value);
    }
    else if (option == 42) m->rg_min_batch = value == 0 ? 48 : std::max(8, value); // the region correlate's threshold alone
    else if (option == 29) m->overlap_lists = value != 0;
    else if (option == 31) m->staged_queries = value != 0;
    else if (option == 30) m->tile_h_forced = value == YM_TILE_H || value == YM_TILE_H_TALL ? value : 0;
    else if (option == 16) m->raster_gx = value;
    else if (option == 17) m->corr_region_parts = value;
    else if (option == 21) m->corr_fuse_score = value;
    else if (option == 45) { m->list_cache_on = value != 0; m->list_key_valid = false; }
    else if (option == 46) m->yag_fast = value < 0 ? 0 : value > 2 ? 1 : value;
    else if (option == 43) m->rg2_h = value;
    else if (option == 44) {
#ifndef YM_EXPERIMENTAL
        return set_err(YM_ERR_UNSUPPORTED, "correlate_region2_kernel is compiled only into builds made with -DYM_EXPERIMENTAL");
#endif
        m->rg2_min_batch = value > 0 ? value : 1 << 30;
    }
    else if (option == 32) {
#ifndef YM_EXPERIMENTAL
        if (value >= 2) return set_err(YM_ERR_UNSUPPORTED, "correlate form %d is compiled only into builds made with -DYM_EXPERIMENTAL", value);
#endif
        m->corr_region_form = value;
    }
    else if (option == 33) m->corr_region_dbg = value;
    else if (option == 34) m->corr_region_rsplit = value;
    else if (option == 35) {
#ifndef YM_EXPERIMENTAL
        return set_err(YM_ERR_UNSUPPORTED, "correlate_item_kernel is compiled only into builds made with -DYM_EXPERIMENTAL");
#endif
        m->item_min_batch = value;
    }
    else if (option == 36) m->raster_planes_only = value;
    else if (option == 37) m->raster_no_rowtab = value;
    else if (option == 38) m->corr_region_pad_lds = value;
    else if (option == 39) m->keep_planes = value;
    else if (option == 40) m->tile_list_min_batch = value;
    else if (option == 19) m->corr_region_cap = value;
    else if (option == 20) m->corr_region_lds = value;
    else if (option == 18) m->raster_hits_per_tile = value;
    else if (option == 7) { // point cache: 0 = on (default), 1 = off, 2 = drop every entry now
        m->cache_off = value == 1;
        m->cache_entries.clear();
        m->cache_index.clear();
        m->cache_used = 0;
    }
    else if (option == 8) { // point cache limit in KiB (development / tests: force the start-over path)
        m->cache_limit = (size_t)std::max(1, value) << 10;
        HIP_TRY(hipStreamSynchronize(m->stream));
        m->cache_entries.clear();
        m->cache_index.clear();
        m->cache_used = 0;
        m->cache_arena.release();
    }
    else return set_err(YM_ERR_INVALID, "unknown option %d", option);
    return YM_OK;
}

int ym_debug_stamps(ym_matcher *m, int enable, uint64_t *out, int32_t count) {
    if (!m) return set_err(YM_ERR_INVALID, "null matcher");
    DEV_GUARD(m->device);
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (out && count > 0)
        HIP_TRY(hipMemcpy(out, m->stamps.p, sizeof(uint64_t) * std::min(count, 32), hipMemcpyDeviceToHost));
    if (enable && !m->stamps_on) HIP_TRY(hipMemset(m->stamps.p, 0, 32 * sizeof(unsigned long long))); // (some slots are counters)
    m->stamps_on = enable != 0;
    return YM_OK;
}

// ---- profiling
int ym_profile_enable(ym_matcher *m, int on) {
    if (!m) return set_err(YM_ERR_INVALID, "null matcher");
    m->profiling = on != 0;
    return YM_OK;
}

int ym_profile_read(ym_matcher *m, int which, double *ms_total, int64_t *launches, int reset) {
    if (!m || which < 0 || which > 2) return set_err(YM_ERR_INVALID, "bad argument");
    DEV_GUARD(m->device);
    HIP_TRY(hipStreamSynchronize(m->stream));
    int rc = prof_collect(m);
    if (rc) return rc;
    if (ms_total) *ms_total = m->prof[which].ms;
    if (launches) *launches = m->prof[which].launches;
    if (reset) { m->prof[which].ms = 0; m->prof[which].launches = 0; }
    return YM_OK;
}

int ym_sequence_stats(const ym_matcher *m, int64_t *segments, int64_t *faults, int64_t *sync_steps) {
    if (!m || !segments || !faults || !sync_steps) return set_err(YM_ERR_INVALID, "null argument");
    *segments = m->seq_segments; *faults = m->seq_faults; *sync_steps = m->seq_sync_steps;
    return YM_OK;
}

int ym_debug_counters(ym_matcher *m, int64_t *out, int32_t count) {
    if (!m || !out || count < 0) return set_err(YM_ERR_INVALID, "bad argument");
    int64_t v[YM_DEBUG_COUNTERS];
    std::memset(v, 0, sizeof v);
    if (m->yag_counters.p) {
        DEV_GUARD(m->device);
        unsigned long long c[4];
        HIP_TRY(hipStreamSynchronize(m->stream));
        HIP_TRY(hipMemcpy(c, m->yag_counters.p, sizeof c, hipMemcpyDeviceToHost));
        for (int i = 0; i < 4; i++) v[i] = (int64_t)c[i];
    }
    v[4] = m->list_cache_hits;
    v[5] = m->last_corr_form;
    for (int i = 0; i < std::min<int>(count, YM_DEBUG_COUNTERS); i++) out[i] = v[i];
    return YM_OK;
}

int ym_cache_stats(const ym_matcher *m, int64_t *hits, int64_t *misses) {
    if (!m) return set_err(YM_ERR_INVALID, "null matcher");
    if (hits) *hits = m->cache_hits;
    if (misses) *misses = m->cache_misses;
    return YM_OK;
}
