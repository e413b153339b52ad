// ym_host_call.hpp -- host runtime: launching a call, collecting it (response expansion), scans -> call descriptors
// Part of yagmatch.hip (included inside its anonymous namespace); not a header of its own.
int launch_call_body(ym_matcher *m, Slot &slot) {
    static const bool debug_host = getenv("YM_DEBUG_HOST") != nullptr; // development aid: host time of a call's phases
    timespec t_[8];
    auto mark = [&](int i) { if (debug_host) clock_gettime(CLOCK_MONOTONIC, &t_[i]); };
    mark(7);
    DEV_GUARD(m->device);
    CallPlan P;
    int rc;
    mark(0);
    m->last_corr_form = -1;
    if ((rc = plan_sizes(m, slot, P))) return rc;
    mark(1);
    // a resident batch again, no scan moved, no cache slot changed hands, nothing was left to fill: last time's plan holds
    Call &pc = slot.call;
    const bool replay = pc.batch_uid != 0 && pc.plan_clean && pc.plan_gen == m->cache_gen && pc.pose_epoch == g_pose_epoch.load(std::memory_order_relaxed);
    if (!replay && (rc = plan_cache(m, slot, P))) return rc;
    mark(2);
    if ((rc = plan_jobs(m, slot, P, replay))) return rc;
    mark(3);
    if ((rc = plan_descriptor(m, slot, P, replay))) return rc;
    if (pc.batch_uid != 0 && !replay) {
        bool clean = !m->cache_off;
        for (const CallScan &cs : pc.scans) clean = clean && !cs.stale && !cs.qstale;
        pc.plan_clean = clean;
        pc.plan_gen = m->cache_gen;
    }
    mark(4);
    hipStream_t st = m->stream;
    hipEvent_t ev_call = nullptr;
    if ((rc = prof_begin(m, 2, &ev_call))) return rc;
    if ((rc = plan_raster(m, slot, P))) return rc;
    mark(5);

    timespec e_[8];
    auto emark = [&](int i) { if (debug_host) clock_gettime(CLOCK_MONOTONIC, &e_[i]); };
    emark(0);
    enqueue_prepare(m, P);
    emark(1);
    // the region correlate's pair lists are built next to the raster on the matcher's second stream: the fork right behind the
    // prepare stage, the launches of that stream after the raster's -- while the host made them first, the device sat idle
    // between the prepare stage and the raster's first kernel (17 us of a 64-item enqueue's 214)
    if (P.region26 && !P.yag) {
        // single-query calls (a loop closure: one query, many chains): are the lists of this very query, at this pose, in this
        // window and lattice, still in the buffers?
        ym_matcher::ListKey key;
        std::memset(&key, 0, sizeof key);
        const CallScan *q0 = P.n_qslots == 1 ? &slot.call.scans[slot.call.items[0].query] : nullptr;
        bool keyed = q0 && q0->id != 0 && m->list_cache_on && !P.stamps && !P.rg_ws;
        if (keyed) {
            key.qid = q0->id; key.pose[0] = q0->pose[0]; key.pose[1] = q0->pose[1]; key.pose[2] = q0->pose[2];
            key.g = P.g; key.lc = P.lc; key.nw = P.rg_nw; key.parts = P.rg_parts; key.nrx = P.rg_nrx; key.nry = P.rg_nry;
            key.rg_h = P.rg2 ? P.rg2_h : YM_RG_H; key.force = m->corr_region; key.nregions = P.rg_nregions; key.ng = P.rg_ng;
            key.es = P.rg_entries_stride; key.ss = P.rg_starts_stride;
            key.pe = m->rg_entries.p; key.ps = m->rg_starts.p; key.pb = m->rg_rbox.p;
            P.lists_cached = m->list_key_valid && std::memcmp(&key, &m->list_key, sizeof key) == 0;
        }
        if (P.lists_cached) m->list_cache_hits++;
        else { m->list_key = key; m->list_key_valid = keyed; } // (the build is enqueued below; a failed call drops the key: launch_call)
    }
    const bool lists_aside = P.region26 && !P.yag && m->overlap_lists && !P.stamps && P.k_end > P.k_begin && !P.lists_cached;
    if (lists_aside && (rc = enqueue_region_lists_aside(m, P))) return rc;
    if ((rc = enqueue_select(m, P))) return rc;
    if ((rc = enqueue_raster(m, P))) return rc;
    if (lists_aside && (rc = enqueue_region_lists_joined(m, P))) return rc;
    emark(2);
    if (P.yag) {
        if ((rc = enqueue_yagpy_passes(m, slot, P))) return rc;
        emark(3); emark(4); emark(5);
    } else {
        if ((rc = enqueue_correlate(m, P))) return rc;
        emark(3);
        enqueue_score(m, slot, P);
        emark(4);
        if (!slot.call.slice) enqueue_finish(m, slot, P);
        else slot.plan = P; // ym_match_slice_finish picks up here
        emark(5);
    }
    if (slot.dev_best_out)
        hipLaunchKernelGGL(ym::argbest_kernel, dim3(1), dim3(256), 0, st, m->states.p, P.B, (long long)slot.chain_id_base,
                           reinterpret_cast<double *>(slot.dev_best_out));
    HIP_TRY(hipGetLastError());
    if ((rc = prof_end(m, ev_call))) return rc;
    if (!slot.done) HIP_TRY(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    if (!slot.call.chain_step) HIP_TRY(hipEventRecord(slot.done, st)); // (a chained segment is collected with one stream synchronisation)
    slot.in_flight = true;
    slot.n_items = P.B;
    HIP_TRY(hipGetLastError());
    mark(6);
    if (debug_host) {
        auto us = [&](int a, int b) { return (t_[b].tv_sec - t_[a].tv_sec) * 1e6 + (t_[b].tv_nsec - t_[a].tv_nsec) * 1e-3; };
        fprintf(stderr, "[ym] host us: guard %.1f sizes %.0f cache %.0f jobs %.0f descriptor %.0f raster-plan %.0f enqueue %.0f (B %d, %d scans)\n", us(7, 0), us(0, 1), us(1, 2),
                us(2, 3), us(3, 4), us(4, 5), us(5, 6), P.B, P.nscans);
        auto eus = [&](int a, int b) { return (e_[b].tv_sec - e_[a].tv_sec) * 1e6 + (e_[b].tv_nsec - e_[a].tv_nsec) * 1e-3; };
        fprintf(stderr, "[ym] enqueue us: prepare %.1f raster %.1f correlate %.1f score %.1f finish %.1f\n", eus(0, 1), eus(1, 2), eus(2, 3), eus(3, 4), eus(4, 5));
    }

    m->last_geom = P.g;
    m->last_lat[0] = P.lc;
    m->last_lat[1] = P.lf;
    m->last_B = P.B; m->last_max_n = P.max_n; m->last_max_base = P.max_base;
    m->last_nt_stride = P.nt_stride; m->last_dim_stride = P.dim_stride;
    m->last_grid_stride = P.grid_stride;
    m->last_sums_stride[0] = P.yag ? P.yvol : (m->keep_sums || (P.B < 8 && P.sums_c <= 65536)) ? P.sums_c : 0;
    m->last_sums_stride[1] = slot.call.refine ? (P.yag ? P.yvol : P.sums_f) : 0;
    m->last_valid = true;
    return YM_OK;
}

// The point cache is updated by plan_cache BEFORE the kernels that fill its new or re-posed entries are enqueued.  If
// anything after that fails (typically an allocation for a large batch), those entries would stay "current" without
// ever having been written, and a later, smaller call would read garbage from them: every entry this call touched is
// made stale again (a pose no scan can have), and the slot's descriptor shadow is dropped (it may name stale = 0).
int launch_call(ym_matcher *m, Slot &slot) {
    const uint64_t before = m->call_counter;
    const int rc = launch_call_body(m, slot);
    if (rc != YM_OK) {
        // a call abandoned after its pair lists were forked onto the second stream: whatever is queued there (bin_kernel
        // writes the shared list buffers) is ordered before the next call's work on the main stream
        if (m->side_stream && m->ev_join && hipEventRecord(m->ev_join, m->side_stream) == hipSuccess)
            (void)hipStreamWaitEvent(m->stream, m->ev_join, 0);
        if (m->call_counter != before)
            for (ym_matcher::CacheEntry &ce : m->cache_entries)
                if (ce.stale_in_call == m->call_counter) ce.pose[0] = ce.pose[1] = ce.pose[2] = std::nan("");
        slot.desc_live_bytes = 0;
        slot.call.plan_clean = false;
        slot.in_flight = false;
        m->list_key_valid = false; // (the lists may never have been built)
    }
    return rc;
}

void state_to_result(const ym_matcher *m, const Slot &slot, const YmItemState &s, int expansions, int64_t prior_hyp,
                     ym_result *r) {
    std::memset(r, 0, sizeof *r);
    r->response = s.response;
    for (int i = 0; i < 3; i++) r->pose[i] = s.mean[i];
    for (int i = 0; i < 9; i++) r->cov[i] = s.cov[i];
    r->coarse_response = s.coarse_response;
    r->coarse_dims[0] = slot.coarse.nx; r->coarse_dims[1] = slot.coarse.ny; r->coarse_dims[2] = slot.coarse.nt;
    int64_t hyp = (int64_t)slot.coarse.nx * slot.coarse.ny * slot.coarse.nt;
    if (slot.call.refine) {
        r->fine_dims[0] = slot.fine.nx; r->fine_dims[1] = slot.fine.ny; r->fine_dims[2] = slot.fine.nt;
        hyp += (int64_t)slot.fine.nx * slot.fine.ny * slot.fine.nt;
    }
    if (m->cfg.semantics == YM_SEM_YAGPY) {
        for (int i = 0; i < 3; i++) { r->coarse_dims[i] = s.ydims[0][i]; r->fine_dims[i] = slot.call.refine ? s.ydims[1][i] : 0; }
        hyp = (int64_t)s.ydims[0][0] * s.ydims[0][1] * s.ydims[0][2];
        if (slot.call.refine) hyp += (int64_t)s.ydims[1][0] * s.ydims[1][1] * s.ydims[1][2];
    } else if (s.nq == 0) { // MatchScan returns before any correlation
        hyp = 0;
        std::memset(r->coarse_dims, 0, sizeof r->coarse_dims);
        std::memset(r->fine_dims, 0, sizeof r->fine_dims);
    }
    r->hypotheses = prior_hyp + hyp;
    r->n_query_points = s.nq;
    r->expansions = expansions;
    r->status = s.status;
    (void)m;
}

// wait for a slot; handle Karto's response expansion (re-run with a wider coarse angle range)
int finish_call(ym_matcher *m, Slot &slot, ym_result *out /* n_items entries */) {
    if (!slot.in_flight) return set_err(YM_ERR_BUSY, "slot has no call in flight");
    bool seen = false;
    if (slot.poll_serial) {
        // single match: spin on the word final_kernel writes behind the result states (a stream event is signalled
        // several microseconds after the kernel has ended); after 2 ms fall back to the event (a faulted kernel never writes)
        const volatile uint32_t *flag = reinterpret_cast<const volatile uint32_t *>(slot.result.p + align_up(sizeof(YmItemState) * slot.n_items, 64));
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (;;) {
            for (int spin = 0; spin < 2048 && !seen; spin++) {
                seen = *flag == slot.poll_serial;
                if (!seen) __builtin_ia32_pause();
            }
            if (seen) break;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 > 2.0) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        slot.poll_serial = 0;
    }
    if (!seen) HIP_TRY(hipEventSynchronize(slot.done));
    slot.in_flight = false;
    const int B = slot.n_items;
    const YmItemState *hs = reinterpret_cast<const YmItemState *>(slot.result.p);
    std::vector<int> redo;
    std::vector<int64_t> prior(B, 0);
    for (int i = 0; i < B; i++) {
        state_to_result(m, slot, hs[i], 0, 0, &out[i]);
        if (m->cfg.semantics == YM_SEM_KARTO && m->cfg.use_response_expansion &&
            kt_double_equal_h(hs[i].coarse_response, 0.0))
            redo.push_back(i);
    }
    // up to three retries, +20 degrees each (ScanMatcher::MatchScan).  A retry re-runs the whole
    // pipeline for the affected items with the wider coarse angle range (rare path).
    const Call base_call = slot.call;
    const int64_t nxy = (int64_t)slot.coarse.nx * slot.coarse.ny;
    const bool any_redo = !redo.empty();
    double off = m->cfg.coarse_search_angle_offset;
    for (int attempt = 1; attempt <= 3 && !redo.empty(); attempt++) {
        const int64_t prev_hyp = nxy * (int64_t)(kt_round_h(off * 2.0 / m->cfg.coarse_angle_resolution) + 1);
        off += 20.0 * YM_KT_PI / 180.0;
        Slot &s2 = m->slots[kAsyncSlots];
        if (&s2 != &slot && s2.in_flight && s2.call.slice)
            return set_err(YM_ERR_BUSY, "an angle-sliced match is in flight on this matcher: finish it (ym_match_slice_finish) first");
        Call sub;
        sub.scans = base_call.scans;
        sub.penalize = base_call.penalize;
        sub.refine = base_call.refine;
        sub.coarse_angle_off = off;
        for (int i : redo) {
            prior[i] += prev_hyp;
            sub.items.push_back(base_call.items[i]);
        }
        s2.call = sub;
        int rc = launch_call(m, s2);
        if (rc) return rc;
        HIP_TRY(hipEventSynchronize(s2.done));
        s2.in_flight = false;
        const YmItemState *h2 = reinterpret_cast<const YmItemState *>(s2.result.p);
        std::vector<int> still;
        for (size_t j = 0; j < redo.size(); j++) {
            const int i = redo[j];
            state_to_result(m, s2, h2[j], attempt, prior[i], &out[i]);
            if (kt_double_equal_h(h2[j].coarse_response, 0.0)) still.push_back(i);
        }
        redo.swap(still);
    }
    if (any_redo && slot.dev_best_user) {
        // the record argbest_kernel left on the device predates the expansion: rewrite it from the final results
        int bi = 0;
        for (int i = 1; i < B; i++)
            if (out[i].response > out[bi].response) bi = i;
        const ym_result &r = out[bi];
        const double rec[8] = {r.response, (double)(slot.chain_id_base + bi), r.pose[0], r.pose[1], r.pose[2], r.cov[0], r.cov[4], r.cov[8]};
        DEV_GUARD(m->device);
        HIP_TRY(hipMemcpyAsync(slot.dev_best_user, rec, sizeof rec, hipMemcpyHostToDevice, m->stream));
        HIP_TRY(hipStreamSynchronize(m->stream));
    }
    slot.dev_best_user = nullptr;
    return YM_OK;
}

// A scan matched right after its creation (a node that receives its scans one by one) need not wait for its creation
// launch: as the QUERY of a synchronous match it needs its readings only, and those are in its pinned staging slot --
// the prepare kernel reads them from there while structure_kernel is still at work on the pool's stream.  The slot is
// pinned for the duration of the call (readers).  Returns the slot, or null when the launch has completed (the usual
// device copy and the scan's structure serve) or the scan was never staged.
ScanStage *staged_query(const ym_scan *s) {
    if (!s->stage) return nullptr;
    ScanPool &p = scan_pool(s->device);
    std::lock_guard<std::mutex> lk(p.mu);
    ScanStage *st = s->stage;
    if (!st) return nullptr;
    const volatile uint32_t *done = reinterpret_cast<const volatile uint32_t *>(st->host + kStageInfoOffset + 16);
    if (done[0] == st->serial && done[1] == st->serial) { // already there: take the info, no wait
        DevGuard guard(s->device);
        stage_wait(p, *st);
        return nullptr;
    }
    st->readers.fetch_add(1, std::memory_order_acq_rel);
    return st;
}
void release_staged(Call &call) {
    for (CallScan &cs : call.scans)
        if (cs.staged) { cs.staged->readers.fetch_sub(1, std::memory_order_acq_rel); cs.staged = nullptr; }
}

int scan_to_call(const ym_scan *s, int semantics, CallScan *o, bool staged_ok = false) {
    if (!s) return set_err(YM_ERR_INVALID, "null scan");
    o->staged = staged_ok ? staged_query(s) : nullptr;
    o->d_ranges = o->staged ? reinterpret_cast<const double *>(o->staged->dev) : s->d_ranges;
    o->n = s->n;
    o->min_angle = s->min_angle;
    o->angle_inc = s->angle_inc;
    o->min_range = s->min_range;
    o->range_threshold = s->range_threshold;
    o->pose[0] = s->pose[0]; o->pose[1] = s->pose[1]; o->pose[2] = s->pose[2];
    o->max_valid = semantics == YM_SEM_YAGPY ? s->max_valid_yagpy : s->max_valid_karto;
    for (int i = 0; i < 4; i++) { o->lbox[i] = s->lbox[i]; o->wbox[i] = s->wbox[i]; }
    o->id = s->id;
    o->beam_spacing = s->beam_spacing;
    o->cache_hint = o->qcache_hint = -1;
    const int sem = semantics == YM_SEM_YAGPY ? 1 : 0;
    if (o->staged) { o->gov = o->cidx = nullptr; o->cnp = 0; return YM_OK; } // (the points are counted and compacted by the call)
    scan_resolve(s);
    o->gov = s->gov_ok[sem] ? s->d_gov[sem] : nullptr;
    o->cidx = s->gov_ok[sem] ? s->d_cidx[sem] : nullptr;
    o->cnp = s->cnp[sem];
    return YM_OK;
}

// sensor-frame bounding box of all readings either semantics can turn into a point (r <= rt, not NaN)
void local_bbox(const double *r, int n, double min_angle, double inc, double rt, double box[4]) {
    box[0] = box[1] = 1e300;
    box[2] = box[3] = -1e300;
    // the beams' directions: one table per sensor geometry (a node's scans all come from the same sensor)
    struct Directions { double min_angle = 0, inc = 0; std::vector<double> c, s; };
    static thread_local Directions dir;
    if ((int)dir.c.size() < n || dir.min_angle != min_angle || dir.inc != inc) {
        dir.min_angle = min_angle; dir.inc = inc;
        dir.c.resize(n); dir.s.resize(n);
        for (int i = 0; i < n; i++) { const double a = min_angle + i * inc; dir.c[i] = std::cos(a); dir.s[i] = std::sin(a); }
    }
    for (int i = 0; i < n; i++) {
        const double v = r[i];
        if (v > rt || std::isnan(v)) continue;
        const double x = v * dir.c[i], y = v * dir.s[i];
        box[0] = std::min(box[0], x); box[1] = std::min(box[1], y);
        box[2] = std::max(box[2], x); box[3] = std::max(box[3], y);
    }
}

// the sensor-frame box at a pose: world bounding box of its four corners (an empty box stays empty)
void world_bbox(const double lbox[4], const double pose[3], double wbox[4]) {
    wbox[0] = wbox[1] = 1e300;
    wbox[2] = wbox[3] = -1e300;
    if (lbox[0] > lbox[2]) return;
    const double c = std::cos(pose[2]), sn = std::sin(pose[2]);
    for (int k = 0; k < 4; k++) {
        const double lx = lbox[(k & 1) ? 2 : 0], ly = lbox[(k & 2) ? 3 : 1];
        const double x = pose[0] + c * lx - sn * ly, y = pose[1] + sn * lx + c * ly;
        wbox[0] = std::min(wbox[0], x); wbox[2] = std::max(wbox[2], x);
        wbox[1] = std::min(wbox[1], y); wbox[3] = std::max(wbox[3], y);
    }
}

void max_valid_ranges(const double *r, int n, double min_range, double rt, double *karto, double *yagpy) {
    double k = 0, y = 0;
    for (int i = 0; i < n; i++) {
        const double v = r[i];
        if (v >= min_range && v <= rt) k = std::max(k, v);
        if (!(v > rt || std::isnan(v))) y = std::max(y, std::fabs(v));
    }
    *karto = k;
    *yagpy = y;
}

double median_beam_spacing(const double *r, int n, double min_range, double rt, double inc) {
    std::vector<double> v;
    v.reserve(n);
    for (int i = 0; i < n; i++)
        if (r[i] >= min_range && r[i] <= rt) v.push_back(r[i]);
    if (v.empty()) return 0.0;
    std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
    return v[v.size() / 2] * std::fabs(inc);
}

int check_desc(const ym_scan_desc *d) {
    if (!d) return set_err(YM_ERR_INVALID, "null scan descriptor");
    if (d->n < 0 || (d->n > 0 && !d->ranges)) return set_err(YM_ERR_INVALID, "scan has n=%d but no ranges", d->n);
    if (d->n > YM_MAX_BEAMS) return set_err(YM_ERR_UNSUPPORTED, "scan has %d readings; limit is %d", d->n, YM_MAX_BEAMS);
    return YM_OK;
}

// staged_query_ok: the caller waits for the call before it returns (and calls release_staged)
int build_single_call(ym_matcher *m, const ym_scan *query, const ym_scan *const *base, int n_base, int penalize,
                      int refine, Call *call, bool staged_query_ok = false) {
    if (!m || !query) return set_err(YM_ERR_INVALID, "null argument");
    if (n_base < 0 || (n_base > 0 && !base)) return set_err(YM_ERR_INVALID, "bad base scan list");
    call->scans.resize(1 + n_base);
    int rc = scan_to_call(query, m->cfg.semantics, &call->scans[0], staged_query_ok && m->staged_queries);
    if (rc) return rc;
    if (query->device != m->device) return set_err(YM_ERR_INVALID, "query scan lives on another device");
    for (int i = 0; i < n_base; i++) {
        if ((rc = scan_to_call(base[i], m->cfg.semantics, &call->scans[1 + i]))) return rc;
        if (base[i]->device != m->device) return set_err(YM_ERR_INVALID, "base scan lives on another device");
    }
    call->items.assign(1, CallItem{0, 1, n_base});
    call->penalize = penalize ? 1 : 0;
    call->refine = refine ? 1 : 0;
    call->coarse_angle_off = m->cfg.coarse_search_angle_offset;
    return YM_OK;
}
