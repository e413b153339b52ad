// ym_abi_matcher.hpp -- C ABI: library, matcher
// Part of yagmatch.hip (included inside its extern "C" block); not a header of its own.

int ym_version(void) { return YM_VERSION; }

#ifndef YM_BUILD_ID
#define YM_BUILD_ID "unknown"
#endif
const char *ym_build_id(void) { return YM_BUILD_ID; }

int ym_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *ym_last_error(void) { return g_err.c_str(); }

ym_matcher *ym_create(const ym_config *cfg, int device) {
    if (!cfg) { set_err(YM_ERR_INVALID, "null config"); return nullptr; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_err(YM_ERR_NO_DEVICE, "no HIP device available (libyagmatch has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= n) { set_err(YM_ERR_NO_DEVICE, "device %d out of range [0, %d)", device, n); return nullptr; }
    ym_matcher *m = new ym_matcher();
    m->cfg = *cfg;
    m->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); cus = 0; }
        m->n_cus = cus > 0 ? cus : 256;
    }
    m->own_stream = nullptr;
    if (build_geometry(m) != YM_OK) { delete m; return nullptr; }
    DevGuard guard(device);
    if (!guard.ok || hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess) {
        set_err(YM_ERR_HIP, "cannot create a stream on device %d", device);
        delete m;
        return nullptr;
    }
    m->stream = m->own_stream;
    pool_register_stream(device, m->own_stream, true);
    if (upload_lut(m) != YM_OK) { ym_destroy(m); return nullptr; }
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ym::select_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 16384);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ym::select_relax_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 16384);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ym::select_global_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 1 << 17);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ym::prepare_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)YM_PREP_LDS_BYTES(YM_MAX_BEAMS));
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(ym::prepare_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)YM_PREP_LDS_BYTES(YM_MAX_BEAMS)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(ym::points_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)YM_PREP_LDS_BYTES(YM_MAX_BEAMS)) != hipSuccess) {
        set_err(YM_ERR_HIP, "cannot raise the dynamic LDS limit of prepare_kernel");
        ym_destroy(m);
        return nullptr;
    }
    if (m->stamps.ensure(32) != YM_OK) { ym_destroy(m); return nullptr; }
    (void)hipMemset(m->stamps.p, 0, 32 * sizeof(unsigned long long));
    return m;
}

void ym_destroy(ym_matcher *m) {
    if (!m) return;
    DevGuard guard(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    if (m->side_stream) (void)hipStreamSynchronize(m->side_stream);
    pool_register_stream(m->device, m->stream, false);
    pool_register_stream(m->device, m->own_stream, false);
    pool_register_stream(m->device, m->side_stream, false);
    m->ktab.release(); m->rowtab.release(); m->desc_dev.release(); m->states.release(); m->qlocal.release(); m->qnp.release(); m->tmp_cache.release(); m->cells.release(); m->bbox.release(); m->grid.release(); m->planes.release(); m->tile_zero.release(); m->sub_zero.release(); m->tile_list.release(); m->tile_count.release(); m->tile_max.release(); m->tile_hits.release(); m->sel_scratch.release(); m->sel_tables.release(); m->sel_rec.release(); m->sel_slot.release();
    m->rg_entries.release(); m->rg_starts.release(); m->rg_rbox.release(); m->rg_walk.release(); m->ga_units.release(); m->ga_starts.release(); m->ga_work.release(); m->ga_counters.release(); m->ga_lane_job.release();
    if (m->tile_max_host) { (void)hipHostFree(m->tile_max_host); m->tile_max_host = nullptr; }
    m->ctrig.release(); m->foffsets.release(); m->hypcell.release(); m->partial.release(); m->sums.release();
    m->resp.release(); m->blockmax.release(); m->probs.release(); m->tmp_ranges.release();
    m->tmp_ranges_host.release(); m->kernel_f_dev.release(); m->map_pts.release(); m->yag_counters.release(); m->cache_arena.release(); m->stamps.release(); m->yaxes.release(); m->yrot.release();
    for (Slot &s : m->slots) {
        s.desc.release();
        s.desc_dev.release();
        s.result.release();
        if (s.done) (void)hipEventDestroy(s.done);
    }
    for (auto &p : m->prof)
        for (auto &e : p.pairs) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
    if (m->side_stream) (void)hipStreamDestroy(m->side_stream);
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    delete m;
}

int ym_get_config(const ym_matcher *m, ym_config *out) {
    if (!m || !out) return set_err(YM_ERR_INVALID, "null argument");
    *out = m->cfg;
    return YM_OK;
}

int ym_set_stream(ym_matcher *m, void *hip_stream) {
    if (!m) return set_err(YM_ERR_INVALID, "null matcher");
    DEV_GUARD(m->device);
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (m->stream != m->own_stream) pool_register_stream(m->device, m->stream, false); // (everything on it has completed)
    m->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : m->own_stream;
    pool_register_stream(m->device, m->stream, true);
    return YM_OK;
}

int ym_synchronize(ym_matcher *m) {
    if (!m) return set_err(YM_ERR_INVALID, "null matcher");
    DEV_GUARD(m->device);
    HIP_TRY(hipStreamSynchronize(m->stream));
    return YM_OK;
}
