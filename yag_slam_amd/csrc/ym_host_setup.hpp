// ym_host_setup.hpp -- host runtime: config -> geometry, the smear kernel's tables, lattices, profiling events
// Part of yagmatch.hip (included inside its anonymous namespace); not a header of its own.

// ---------------------------------------------------------------- config -> geometry
int build_geometry(ym_matcher *m) {
    const ym_config &c = m->cfg;
    if (!(c.resolution > 0) || !(c.search_size > 0) || c.smear_deviation < 0 || !(c.range_threshold > 0))
        return set_err(YM_ERR_INVALID, "invalid matcher parameters (resolution, search_size, range_threshold must be > 0)");
    if (!(0.5 * c.resolution <= c.smear_deviation && c.smear_deviation <= 10 * c.resolution))
        return set_err(YM_ERR_INVALID, "Smear deviation must be between %g and %g", 0.5 * c.resolution,
                       10 * c.resolution);
    if (!(c.coarse_angle_resolution > 0) || !(c.fine_search_angle_resolution > 0) ||
        !(c.coarse_search_angle_offset > 0))
        return set_err(YM_ERR_INVALID, "angle offsets/resolutions must be > 0");
    if (c.semantics != YM_SEM_KARTO && c.semantics != YM_SEM_YAGPY)
        return set_err(YM_ERR_INVALID, "unknown semantics %d", c.semantics);
    YmGeom &g = m->geom;
    std::memset(&g, 0, sizeof g);
    if (c.semantics == YM_SEM_YAGPY) {
        // Scan2DMatcherPy.match_scan (/root/reference/yag_slam/scan_matching.py:183-190) and
        // calculate_kernel (/root/reference/yag_slam/helpers.py:86-97)
        g.res = c.resolution;
        g.scale = 1.0 / c.resolution;
        const int G = (int)(c.search_size / c.resolution + 1 + 2 * c.range_threshold / c.resolution);
        if (G <= 0) return set_err(YM_ERR_INVALID, "bad grid size %d", G);
        g.side = 0;
        g.roi_w = G;
        g.border = 0;
        g.storage_w = G;
        const int ks = (int)(4 * std::rint(c.smear_deviation / c.resolution) + 1);
        g.half_kernel = ks / 2;
        if (g.half_kernel > YM_MAX_KERNEL_HALF || g.half_kernel < 1)
            return set_err(YM_ERR_INVALID, "kernel half size %d out of range", g.half_kernel);
        g.semantics = c.semantics;
        g.zone_count = 1; // the Python path re-stamps occupied cells: order-independent
        const int h = g.half_kernel;
        m->kernel.assign((size_t)ks * ks, 0);
        m->kernel_f.assign((size_t)ks * ks, 0.0);
        for (int i_ = 0; i_ < ks; i_++)
            for (int j_ = 0; j_ < ks; j_++) {
                const int i = i_ - h, j = j_ - h;
                const double a = i * c.resolution, b = j * c.resolution;
                const double sqdist = a * a + b * b;
                const double v = std::exp(-0.5 * sqdist / (c.smear_deviation * c.smear_deviation));
                m->kernel_f[(size_t)i_ * ks + j_] = v;
                m->kernel[(size_t)i_ * ks + j_] = (uint8_t)(int)(100 * v); // score: int(100 * cell), helpers.py:142-145
            }
        return YM_OK;
    }
    // ScanMatcher::Create + CorrelationGrid::CreateGrid
    g.scale = 1.0 / c.resolution;
    g.res = 1.0 / g.scale;
    g.side = (int)(kt_round_h(c.search_size / c.resolution) + 1);
    const int margin = (int)std::ceil(c.range_threshold / c.resolution);
    g.roi_w = g.side + 2 * margin;
    g.half_kernel = (int)kt_round_h(2.0 * c.smear_deviation / c.resolution);
    if (g.half_kernel > YM_MAX_KERNEL_HALF || g.half_kernel < 1)
        return set_err(YM_ERR_INVALID, "kernel half size %d out of range", g.half_kernel);
    g.border = g.half_kernel + 1;
    g.storage_w = g.roi_w + 2 * g.border;
    {
        // Karto asserts an odd grid size; with an even one the last coarse lattice column falls
        // outside m_pSearchSpaceProbs and MatchScan throws "Index out of range in probability search".
        const double coff = 0.5 * (g.side - 1) * g.res, cstep = 2 * g.res;
        const int nx = (int)(kt_round_h(coff * 2.0 / cstep) + 1);
        const int last = (int)kt_round_h(((nx - 1) * cstep) * g.scale);
        if (last >= g.side)
            return set_err(YM_ERR_INVALID,
                           "search_size / resolution = %g must be an even integer (Karto: index out of range in "
                           "probability search)", c.search_size / c.resolution);
    }
    g.semantics = c.semantics;
    g.dist_var = c.distance_variance_penalty;
    g.ang_var = c.angle_variance_penalty;
    g.min_dist_pen = c.minimum_distance_penalty;
    g.min_ang_pen = c.minimum_angle_penalty;
    // CorrelationGrid::CalculateKernel
    const int h = g.half_kernel, ks = 2 * h + 1;
    m->kernel.assign((size_t)ks * ks, 0);
    int zone = 0;
    for (int i = -h; i <= h; i++)
        for (int j = -h; j <= h; j++) {
            const double d = std::hypot(i * g.res, j * g.res);
            const double z = std::exp(-0.5 * std::pow(d / c.smear_deviation, 2));
            const unsigned v = (unsigned)kt_round_h(z * YM_OCCUPIED);
            m->kernel[(size_t)(j + h) + (size_t)ks * (i + h)] = (uint8_t)v;
            zone += (v == YM_OCCUPIED);
        }
    g.zone_count = zone;
    return YM_OK;
}

// smear kernel as a function of the squared cell distance, with the proof obligation the raster
// kernel relies on: inside the (2h+1)^2 window the kernel value depends only on dx^2+dy^2 and
// never increases with it.
int upload_lut(ym_matcher *m) {
    const int h = m->geom.half_kernel, ks = 2 * h + 1;
    const int n = 2 * h * h + 1;
    std::vector<int> lut(n, -1);
    for (int dy = 0; dy <= h; dy++)
        for (int dx = 0; dx <= h; dx++) {
            const int v = m->kernel[(size_t)(dx + h) + (size_t)ks * (dy + h)];
            int &e = lut[dx * dx + dy * dy];
            if (e >= 0 && e != v)
                return set_err(YM_ERR_UNSUPPORTED, "smear kernel is not a function of squared distance at d2=%d", dx * dx + dy * dy);
            e = v;
        }
    int prev = 255;
    m->z2max = 0;
    for (int i = 0; i < n; i++)
        if (lut[i] == YM_OCCUPIED) m->z2max = i;
    // (smear_deviation <= 10 * resolution, the reference's own assertion checked above, keeps the kernel below 100 from squared
    //  distance 2 on: 100 * exp(-0.5 * 2 / 100) rounds to 99.  The nine-neighbour form of the select rule is therefore never
    //  needed and no longer instantiated -- select_kernel<9> spilled 240 bytes per lane.)
    if (m->z2max > 1) return set_err(YM_ERR_UNSUPPORTED, "smear kernel holds 100 out to squared distance %d", m->z2max);
    std::vector<uint8_t> q(n + 8, 0);
    for (int i = 0; i < n; i++) {
        if (lut[i] < 0) { q[i] = (uint8_t)prev; continue; } // unreachable distance: never looked up
        if (lut[i] > prev) return set_err(YM_ERR_UNSUPPORTED, "smear kernel is not monotone at d2=%d", i);
        prev = lut[i];
        q[i] = (uint8_t)lut[i];
    }
    int rc = m->ktab.ensure(q.size());
    if (rc) return rc;
    HIP_TRY(hipMemcpy(m->ktab.p, q.data(), q.size(), hipMemcpyHostToDevice));
    // the raster's row pass (ym_k_raster.hpp): an 8-cell group sees the 8 + 2h bitmap bits [0, 8 + 2h) of its row, cell q
    // sits at bit q + h.  Table j, indexed by the seven bits 7j .. 7j + 6, holds for every cell the distance to the nearest
    // of THOSE bits that is set and at most h away (127: none); the group's distances are the byte-wise minimum over j.
    m->n_rowtab = 0;
    m->rowtab_shift = -1;
    if (2 * h + 8 <= 32) {
        // h <= 10: the mirrored form -- the window shifted into the middle of 28 bits, two tables stored (groups 0 and 1; groups 3
        // and 2 are their mirror images).  h = 11, 12: one table per group of seven bits.
        const bool mirror = h <= 10;
        const int shift = mirror ? (28 - (2 * h + 8)) / 2 : 0;
        const int nt = mirror ? 2 : (2 * h + 8 + 6) / 7;
        std::vector<uint8_t> t((size_t)nt * 128 * 8);
        for (int j = 0; j < nt; j++)
            for (int v = 0; v < 128; v++)
                for (int c = 0; c < 8; c++) {
                    int best = 127;
                    for (int i = 0; i < 7; i++)
                        if ((v >> i) & 1) {
                            const int d = std::abs(7 * j + i - (c + h + shift));
                            if (d <= h && d < best) best = d;
                        }
                    t[((size_t)j * 128 + v) * 8 + c] = (uint8_t)best;
                }
        if (mirror) m->rowtab_shift = shift;
        if ((rc = m->rowtab.ensure(t.size()))) return rc;
        HIP_TRY(hipMemcpy(m->rowtab.p, t.data(), t.size(), hipMemcpyHostToDevice));
        m->n_rowtab = nt;
    }
    return YM_OK;
}

YmLattice make_lattice(const YmGeom &g, double off, double step, double angle_off, double angle_res, int fine,
                       int penalize) {
    YmLattice l;
    std::memset(&l, 0, sizeof l);
    l.off_x = l.off_y = off;
    l.step_x = l.step_y = step;
    l.angle_off = angle_off;
    l.angle_res = angle_res;
    l.nx = (int)(kt_round_h(off * 2.0 / step) + 1);
    l.ny = l.nx;
    l.nt = (int)(kt_round_h(angle_off * 2.0 / angle_res) + 1);
    l.fine = fine;
    l.penalize = penalize;
    (void)g;
    return l;
}

// ---------------------------------------------------------------- profiling helpers
int prof_begin(ym_matcher *m, int which, hipEvent_t *stop_out) {
    *stop_out = nullptr;
    if (!m->profiling) return YM_OK;
    ProfEvents &p = m->prof[which];
    if (p.used == p.pairs.size()) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        p.pairs.emplace_back(a, b);
    }
    HIP_TRY(hipEventRecord(p.pairs[p.used].first, m->stream));
    *stop_out = p.pairs[p.used].second;
    p.used++;
    return YM_OK;
}
int prof_end(ym_matcher *m, hipEvent_t stop) {
    if (stop) HIP_TRY(hipEventRecord(stop, m->stream));
    return YM_OK;
}
int prof_collect(ym_matcher *m) {
    for (auto &p : m->prof) {
        for (size_t i = 0; i < p.used; i++) {
            float ms = 0;
            HIP_TRY(hipEventSynchronize(p.pairs[i].second));
            HIP_TRY(hipEventElapsedTime(&ms, p.pairs[i].first, p.pairs[i].second));
            p.ms += ms;
            p.launches++;
        }
        p.used = 0;
    }
    return YM_OK;
}
