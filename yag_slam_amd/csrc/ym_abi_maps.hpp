// ym_abi_maps.hpp -- C ABI: prebuilt maps (match against a map), occupancy-grid rendering
// Part of yagmatch.hip (included inside its extern "C" block); not a header of its own.
// ---- prebuilt maps: the "match against a map" entry of the reference's Python matcher
static ym_map *map_alloc(ym_matcher *m, int width, int height) {
    if (!m) { set_err(YM_ERR_INVALID, "null matcher"); return nullptr; }
    if (m->cfg.semantics != YM_SEM_YAGPY) {
        set_err(YM_ERR_UNSUPPORTED, "maps exist only in the reference's Python matcher: create the matcher with YM_SEM_YAGPY");
        return nullptr;
    }
    if (width <= 0 || height <= 0 || (double)width * height > 1.0e9) { set_err(YM_ERR_INVALID, "bad map size %d x %d", width, height); return nullptr; }
    ym_map *mp = new ym_map();
    mp->device = m->device;
    mp->width = width; mp->height = height;
    mp->d_cgrid = nullptr; mp->d_g8 = nullptr;
    const size_t n = (size_t)width * height;
    if (hipMalloc(reinterpret_cast<void **>(&mp->d_cgrid), n * sizeof(double)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&mp->d_g8), n + 64) != hipSuccess) {
        set_err(YM_ERR_HIP, "cannot allocate a %d x %d map", width, height);
        if (mp->d_cgrid) (void)hipFree(mp->d_cgrid);
        delete mp;
        return nullptr;
    }
    return mp;
}

ym_map *ym_map_from_occupancy(ym_matcher *m, const uint8_t *image, int width, int height, int pitch, int occupied_value) {
    if (!image || pitch < width) { set_err(YM_ERR_INVALID, "bad occupancy image"); return nullptr; }
    DevGuard guard(m ? m->device : 0);
    ym_map *mp = map_alloc(m, width, height);
    if (!mp) return nullptr;
    const int ks = 2 * m->geom.half_kernel + 1;
    uint8_t *d_img = nullptr;
    bool ok = hipMalloc(reinterpret_cast<void **>(&d_img), (size_t)pitch * height) == hipSuccess &&
              hipMemcpyAsync(d_img, image, (size_t)pitch * height, hipMemcpyHostToDevice, m->stream) == hipSuccess &&
              m->kernel_f_dev.ensure(m->kernel_f.size()) == YM_OK &&
              hipMemcpyAsync(m->kernel_f_dev.p, m->kernel_f.data(), m->kernel_f.size() * sizeof(double), hipMemcpyHostToDevice, m->stream) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(ym::map_from_occupancy_kernel, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, m->stream, d_img, width,
                           height, pitch, occupied_value, m->kernel_f_dev.p, ks, mp->d_cgrid, mp->d_g8);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(m->stream) == hipSuccess;
    }
    if (d_img) (void)hipFree(d_img);
    if (!ok) { set_err(YM_ERR_HIP, "building the map failed"); ym_map_destroy(mp); return nullptr; }
    return mp;
}

ym_map *ym_map_from_grid(ym_matcher *m, const double *cgrid, int width, int height) {
    if (!cgrid) { set_err(YM_ERR_INVALID, "null grid"); return nullptr; }
    DevGuard guard(m ? m->device : 0);
    ym_map *mp = map_alloc(m, width, height);
    if (!mp) return nullptr;
    const size_t n = (size_t)width * height;
    bool ok = hipMemcpyAsync(mp->d_cgrid, cgrid, n * sizeof(double), hipMemcpyHostToDevice, m->stream) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(ym::map_from_grid_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->stream, mp->d_cgrid, n, mp->d_g8);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(m->stream) == hipSuccess;
    }
    if (!ok) { set_err(YM_ERR_HIP, "uploading the map failed"); ym_map_destroy(mp); return nullptr; }
    return mp;
}

int ym_map_size(const ym_map *mp, int *width, int *height) {
    if (!mp) return set_err(YM_ERR_INVALID, "null map");
    if (width) *width = mp->width;
    if (height) *height = mp->height;
    return YM_OK;
}

int ym_map_read(const ym_map *mp, double *out, int64_t out_count) {
    if (!mp || !out) return set_err(YM_ERR_INVALID, "null argument");
    const size_t n = (size_t)mp->width * mp->height;
    if ((size_t)out_count < n) return set_err(YM_ERR_INVALID, "buffer too small: need %zu entries", n);
    DEV_GUARD(mp->device);
    HIP_TRY(hipMemcpy(out, mp->d_cgrid, n * sizeof(double), hipMemcpyDeviceToHost));
    return YM_OK;
}

void ym_map_destroy(ym_map *mp) {
    if (!mp) return;
    DevGuard guard(mp->device);
    if (mp->d_cgrid) (void)hipFree(mp->d_cgrid);
    if (mp->d_g8) (void)hipFree(mp->d_g8);
    delete mp;
}

int ym_match_map(ym_matcher *m, const ym_map *mp, double ox, double oy, const ym_scan *const *queries, int n_queries,
                 int penalize, int refine, const ym_map_search *coarse, ym_result *out) {
    if (!m || !mp || !queries || !out) return set_err(YM_ERR_INVALID, "null argument");
    if (m->cfg.semantics != YM_SEM_YAGPY) return set_err(YM_ERR_UNSUPPORTED, "ym_match_map needs a YM_SEM_YAGPY matcher");
    if (n_queries <= 0 || n_queries > 64) return set_err(YM_ERR_INVALID, "n_queries must be in [1, 64]");
    if (mp->device != m->device) return set_err(YM_ERR_INVALID, "map lives on another device");
    DEV_GUARD(m->device);
    // scan_matching.py:136-139: the search centre is the mean of the query poses (Python's left-to-right sum), heading 0
    double sx = 0, sy = 0;
    int total = 0, max_n = 1;
    for (int i = 0; i < n_queries; i++) {
        if (!queries[i] || queries[i]->device != m->device) return set_err(YM_ERR_INVALID, "query %d is null or lives on another device", i);
        sx = i == 0 ? queries[i]->pose[0] : sx + queries[i]->pose[0];
        sy = i == 0 ? queries[i]->pose[1] : sy + queries[i]->pose[1];
        total += queries[i]->n;
        max_n = std::max(max_n, queries[i]->n);
    }
    const double ox_real = sx / (double)n_queries, oy_real = sy / (double)n_queries;
    total = std::max(total, 1);
    // the reference's hard-coded coarse pass (scan_matching.py:152-153) unless the caller overrides it
    ym_map_search cs;
    if (coarse) cs = *coarse;
    else { cs.xy_search = 0.25; cs.xy_step = 0.01; cs.angle_search = 0.1; cs.angle_step = 0.01; cs.grid_resolution = 0.05; cs.penalize = 0; cs.reserved = 0; }
    if (!(cs.xy_step > 0) || !(cs.angle_step > 0) || !(cs.grid_resolution > 0) || !(cs.xy_search > 0) || !(cs.angle_search > 0))
        return set_err(YM_ERR_INVALID, "bad coarse search parameters");
    const double res = m->cfg.resolution;
    const int maxd = std::max({8, (int)std::ceil(2 * cs.xy_search / cs.xy_step) + 2, (int)std::ceil(4 * res / res) + 2});
    const int maxt = std::max({13, (int)std::ceil(2 * cs.angle_search / cs.angle_step) + 2});
    if (maxd > YM_YAG_MAX_DIM || maxt > YM_YAG_MAX_NT)
        return set_err(YM_ERR_UNSUPPORTED, "map search lattice %d x %d x %d exceeds the built-in limit", maxd, maxd, maxt);
    const size_t vol = (size_t)maxt * maxd * maxd;
    int rc;
    Slot &slot = m->slots[kAsyncSlots];
    if (slot.in_flight) return set_err(YM_ERR_BUSY, "the synchronous slot is in flight");
    if ((rc = m->states.ensure(1))) return rc;
    if ((rc = m->map_pts.ensure((size_t)total))) return rc;
    if ((rc = m->yaxes.ensure((size_t)3 * YM_YAG_MAX_DIM))) return rc;
    if ((rc = m->yrot.ensure((size_t)maxt * total))) return rc;
    if ((rc = m->sums.ensure(2 * vol))) return rc;
    if ((rc = m->resp.ensure(vol))) return rc;
    const size_t scans_bytes = align_up(sizeof(YmScanRef) * n_queries, 16);
    if ((rc = slot.desc.ensure(scans_bytes + sizeof(YmItemState)))) return rc;
    if ((rc = slot.result.ensure(sizeof(YmItemState)))) return rc;
    if ((rc = m->desc_dev.ensure(scans_bytes))) return rc;
    slot.desc_live_bytes = 0; // (the slot's pinned descriptor buffer is rewritten here)
    YmScanRef *hs = reinterpret_cast<YmScanRef *>(slot.desc.p);
    std::memset(hs, 0, scans_bytes);
    for (int i = 0; i < n_queries; i++) {
        const ym_scan *q = queries[i];
        scan_resolve(q);
        hs[i].ranges = q->d_ranges; hs[i].n = q->n;
        hs[i].min_angle = q->min_angle; hs[i].angle_inc = q->angle_inc; hs[i].min_range = q->min_range;
        hs[i].range_threshold = q->range_threshold;
        hs[i].pose[0] = q->pose[0]; hs[i].pose[1] = q->pose[1]; hs[i].pose[2] = q->pose[2];
    }
    YmItemState *st0 = reinterpret_cast<YmItemState *>(slot.desc.p + scans_bytes);
    std::memset(st0, 0, sizeof *st0);
    st0->pose[0] = st0->center[0] = ox_real; st0->pose[1] = st0->center[1] = oy_real;
    st0->off_x = ox; st0->off_y = oy;
    st0->ql = m->map_pts.p;
    hipStream_t st = m->stream;
    HIP_TRY(hipMemcpyAsync(m->desc_dev.p, hs, scans_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->states.p, st0, sizeof *st0, hipMemcpyHostToDevice, st));
    ym::MapPointsArgs pa;
    pa.scans = reinterpret_cast<const YmScanRef *>(m->desc_dev.p); pa.n_scans = n_queries; pa.max_n = max_n;
    pa.ox_real = ox_real; pa.oy_real = oy_real; pa.out = m->map_pts.p; pa.state = m->states.p;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ym::map_points_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)YM_PREP_LDS_BYTES(YM_MAX_BEAMS));
    hipLaunchKernelGGL(ym::map_points_kernel, dim3(1), dim3(1024), YM_PREP_LDS_BYTES(max_n), st, pa);
    m->sums_pass_offset[0] = 0;
    m->sums_pass_offset[1] = vol;
    for (int pass = 0; pass < (refine ? 2 : 1); pass++) {
        ym::YagArgs a;
        std::memset(&a, 0, sizeof a);
        a.g = m->geom; a.pass = pass; a.refine = refine ? 1 : 0;
        a.last = (pass == 1 || !refine) ? 1 : 0;
        if (pass == 0) {
            a.search_xy = cs.xy_search; a.step_xy = cs.xy_step; a.search_t = cs.angle_search; a.step_t = cs.angle_step;
            a.map_res = cs.grid_resolution; a.penalize = cs.penalize ? 1 : 0;
        } else { // scan_matching.py:155-157
            a.search_xy = res * 2; a.step_xy = res; a.search_t = 0.0349 * 0.5; a.step_t = 0.00349;
            a.map_res = res; a.penalize = penalize ? 1 : 0;
        }
        a.coarse_angle_res = m->cfg.coarse_angle_resolution;
        a.states = m->states.p; a.host_out = reinterpret_cast<YmItemState *>(slot.result.dp);
        a.axes = m->yaxes.p; a.rot = m->yrot.p;
        a.sums = m->sums.p + m->sums_pass_offset[pass]; a.out = m->resp.p;
        a.grid = mp->d_g8; a.grid_stride = 0; a.vol_stride = vol;
        a.max_n = total; a.maxd = maxd; a.maxt = maxt;
        a.map_w = mp->width; a.map_h = mp->height; a.map_ox = ox; a.map_oy = oy;
        hipLaunchKernelGGL(ym::yag_setup_kernel, dim3(maxt, 1), dim3(256), 0, st, a);
        hipLaunchKernelGGL(ym::yag_score_kernel, dim3((maxd * maxd + 255) / 256, maxt, 1), dim3(256), 0, st, a);
        hipLaunchKernelGGL(ym::yag_reduce_kernel<1024>, dim3(1), dim3(1024), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    const YmItemState &r = *reinterpret_cast<const YmItemState *>(slot.result.p);
    std::memset(out, 0, sizeof *out);
    out->response = r.response;
    for (int i = 0; i < 3; i++) out->pose[i] = r.mean[i];
    for (int i = 0; i < 9; i++) out->cov[i] = r.cov[i];
    out->coarse_response = r.ybest[0][0];
    for (int i = 0; i < 3; i++) { out->coarse_dims[i] = r.ydims[0][i]; out->fine_dims[i] = refine ? r.ydims[1][i] : 0; }
    out->hypotheses = (int64_t)r.ydims[0][0] * r.ydims[0][1] * r.ydims[0][2] +
                      (refine ? (int64_t)r.ydims[1][0] * r.ydims[1][1] * r.ydims[1][2] : 0);
    out->n_query_points = r.nq;
    out->status = r.status;
    m->last_valid = false; // the debug getters describe match_scan calls
    return YM_OK;
}

// ---- occupancy-grid rendering (karto_scanmatcher.create_occupancy_grid)
ym_occupancy *ym_occupancy_create(const ym_scan *const *scans, int n_scans, double resolution, double range_threshold) {
    if (!scans || n_scans <= 0) { set_err(YM_ERR_INVALID, "no scans"); return nullptr; }
    if (!(resolution > 0) || !(range_threshold > 0)) { set_err(YM_ERR_INVALID, "resolution and range_threshold must be > 0"); return nullptr; }
    const int device = scans[0] ? scans[0]->device : -1;
    int max_n = 1;
    for (int i = 0; i < n_scans; i++) {
        if (!scans[i] || scans[i]->device != device) { set_err(YM_ERR_INVALID, "scan %d is null or lives on another device", i); return nullptr; }
        max_n = std::max(max_n, scans[i]->n);
    }
    DevGuard guard(device);
    if (!guard.ok) { set_err(YM_ERR_HIP, "cannot make device %d current", device); return nullptr; }
    std::vector<YmScanRef> hs(n_scans);
    std::memset(hs.data(), 0, sizeof(YmScanRef) * n_scans);
    for (int i = 0; i < n_scans; i++) {
        const ym_scan *q = scans[i];
        scan_resolve(q);
        hs[i].ranges = q->d_ranges; hs[i].n = q->n;
        hs[i].min_angle = q->min_angle; hs[i].angle_inc = q->angle_inc; hs[i].min_range = q->min_range;
        hs[i].range_threshold = q->max_range; // the laser's MAXIMUM range travels in this field (see occ_trace_kernel)
        hs[i].pose[0] = q->pose[0]; hs[i].pose[1] = q->pose[1]; hs[i].pose[2] = q->pose[2];
    }
    YmScanRef *d_scans = nullptr;
    double *d_boxes = nullptr;
    unsigned *d_cnt = nullptr;
    uint8_t *d_img = nullptr;
    ym_occupancy *og = nullptr;
    bool said = false; // this call has set its own error message (the thread's last message may be an older one)
    bool ok = hipMalloc(reinterpret_cast<void **>(&d_scans), sizeof(YmScanRef) * n_scans) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&d_boxes), sizeof(double) * 4 * n_scans) == hipSuccess &&
              hipMemcpy(d_scans, hs.data(), sizeof(YmScanRef) * n_scans, hipMemcpyHostToDevice) == hipSuccess;
    ym::OccArgs a;
    std::memset(&a, 0, sizeof a);
    if (ok) {
        a.scans = d_scans; a.n_scans = n_scans; a.max_n = max_n; a.range_threshold = range_threshold; a.boxes = d_boxes;
        hipLaunchKernelGGL(ym::occ_bbox_kernel, dim3(n_scans), dim3(256), 0, nullptr, a);
        std::vector<double> boxes((size_t)4 * n_scans);
        ok = hipGetLastError() == hipSuccess && hipMemcpy(boxes.data(), d_boxes, sizeof(double) * boxes.size(), hipMemcpyDeviceToHost) == hipSuccess;
        if (ok) {
            // OccupancyGrid::ComputeDimensions: the scans' bounding boxes joined, width = Round(size * scale)
            double x0 = 1e300, y0 = 1e300, x1 = -1e300, y1 = -1e300;
            for (int i = 0; i < n_scans; i++) {
                x0 = std::min(x0, boxes[4 * i]); y0 = std::min(y0, boxes[4 * i + 1]);
                x1 = std::max(x1, boxes[4 * i + 2]); y1 = std::max(y1, boxes[4 * i + 3]);
            }
            const double scale = 1.0 / resolution;
            const int width = (int)kt_round_h((x1 - x0) * scale), height = (int)kt_round_h((y1 - y0) * scale);
            if (width <= 0 || height <= 0 || (double)width * height > 2.0e9) {
                set_err(YM_ERR_UNSUPPORTED, "occupancy grid of %d x %d cells", width, height);
                said = true;
                ok = false;
            } else {
                const size_t n = (size_t)width * height;
                ok = hipMalloc(reinterpret_cast<void **>(&d_cnt), 2 * n * sizeof(unsigned)) == hipSuccess &&
                     hipMalloc(reinterpret_cast<void **>(&d_img), n) == hipSuccess &&
                     hipMemset(d_cnt, 0, 2 * n * sizeof(unsigned)) == hipSuccess;
                if (ok) {
                    a.scale = scale; a.off_x = x0; a.off_y = y0; a.width = width; a.height = height;
                    a.pass = d_cnt; a.hits = d_cnt + n; a.image = d_img;
                    hipLaunchKernelGGL(ym::occ_trace_kernel, dim3((max_n + 255) / 256, n_scans), dim3(256), 0, nullptr, a);
                    ok = hipGetLastError() == hipSuccess;
                    hipLaunchKernelGGL(ym::occ_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, a);
                    og = new ym_occupancy();
                    og->device = device;
                    og->info.width = width; og->info.height = height;
                    og->info.offset_x = x0; og->info.offset_y = y0; og->info.resolution = resolution;
                    og->image.resize(n);
                    ok = ok && hipGetLastError() == hipSuccess && hipMemcpy(og->image.data(), d_img, n, hipMemcpyDeviceToHost) == hipSuccess;
                }
            }
        }
    }
    if (d_scans) (void)hipFree(d_scans);
    if (d_boxes) (void)hipFree(d_boxes);
    if (d_cnt) (void)hipFree(d_cnt);
    if (d_img) (void)hipFree(d_img);
    if (!ok) {
        if (!said) set_err(YM_ERR_HIP, "rendering the occupancy grid failed: %s", hipGetErrorString(hipGetLastError()));
        delete og;
        return nullptr;
    }
    return og;
}

int ym_occupancy_get_info(const ym_occupancy *og, ym_occupancy_info *info) {
    if (!og || !info) return set_err(YM_ERR_INVALID, "null argument");
    *info = og->info;
    return YM_OK;
}

int ym_occupancy_read(const ym_occupancy *og, uint8_t *image, int64_t image_bytes) {
    if (!og || !image) return set_err(YM_ERR_INVALID, "null argument");
    if ((size_t)image_bytes < og->image.size()) return set_err(YM_ERR_INVALID, "buffer too small: need %zu bytes", og->image.size());
    std::memcpy(image, og->image.data(), og->image.size());
    return YM_OK;
}

void ym_occupancy_destroy(ym_occupancy *og) { delete og; }
