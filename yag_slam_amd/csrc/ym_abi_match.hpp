// ym_abi_match.hpp -- C ABI: the hot path -- single matches, sequences, batches and pairs, angle-sliced matches
// Part of yagmatch.hip (included inside its extern "C" block); not a header of its own.
// ---- hot path
int ym_match_scans(ym_matcher *m, const ym_scan *query, const ym_scan *const *base, int n_base, int penalize,
                   int refine, ym_result *out) {
    if (!out) return set_err(YM_ERR_INVALID, "null result");
    if (!m) return set_err(YM_ERR_INVALID, "null matcher");
    Slot &slot = m->slots[kAsyncSlots];
    if (slot.in_flight && slot.call.slice)
        return set_err(YM_ERR_BUSY, "an angle-sliced match is in flight on this matcher: finish it (ym_match_slice_finish) first");
    Call call;
    int rc = build_single_call(m, query, base, n_base, penalize, refine, &call, true);
    if (rc) { release_staged(call); return rc; }
    slot.call = call;
    if ((rc = launch_call(m, slot)) == YM_OK) rc = finish_call(m, slot, out);
    else if (slot.in_flight) (void)hipStreamSynchronize(m->stream);
    release_staged(slot.call);
    return rc;
}

int ym_match(ym_matcher *m, const ym_scan_desc *query, const ym_scan_desc *base, int n_base, int penalize, int refine,
             ym_result *out) {
    if (!m || !out) return set_err(YM_ERR_INVALID, "null argument");
    if (n_base < 0 || (n_base > 0 && !base)) return set_err(YM_ERR_INVALID, "bad base scan list");
    int rc;
    if ((rc = check_desc(query))) return rc;
    size_t total = (size_t)query->n;
    for (int i = 0; i < n_base; i++) {
        if ((rc = check_desc(&base[i]))) return rc;
        total += (size_t)base[i].n;
    }
    if (m->slots[kAsyncSlots].in_flight && m->slots[kAsyncSlots].call.slice)
        return set_err(YM_ERR_BUSY, "an angle-sliced match is in flight on this matcher: finish it (ym_match_slice_finish) first");
    DEV_GUARD(m->device);
    // the staging buffers may still feed an earlier async copy on this stream
    HIP_TRY(hipStreamSynchronize(m->stream));
    if ((rc = m->tmp_ranges.ensure(total + 1))) return rc;
    if ((rc = m->tmp_ranges_host.ensure(sizeof(double) * (total + 1)))) return rc;
    double *hr = reinterpret_cast<double *>(m->tmp_ranges_host.p);
    Call call;
    call.scans.resize(1 + n_base);
    size_t at = 0;
    for (int i = 0; i <= n_base; i++) {
        const ym_scan_desc &d = i == 0 ? *query : base[i - 1];
        if (d.n > 0) std::memcpy(hr + at, d.ranges, sizeof(double) * d.n);
        CallScan &c = call.scans[i];
        c.d_ranges = m->tmp_ranges.p + at;
        c.n = d.n;
        c.min_angle = d.min_angle; c.angle_inc = d.angle_increment; c.min_range = d.min_range;
        c.range_threshold = d.range_threshold;
        c.pose[0] = d.pose[0]; c.pose[1] = d.pose[1]; c.pose[2] = d.pose[2];
        double k, y;
        max_valid_ranges(d.ranges, d.n, d.min_range, d.range_threshold, &k, &y);
        c.max_valid = m->cfg.semantics == YM_SEM_YAGPY ? y : k;
        local_bbox(d.ranges, d.n, d.min_angle, d.angle_increment, d.range_threshold, c.lbox);
        world_bbox(c.lbox, c.pose, c.wbox);
        c.beam_spacing = median_beam_spacing(d.ranges, d.n, d.min_range, d.range_threshold, d.angle_increment);
        at += (size_t)d.n;
    }
    if (total > 0)
        HIP_TRY(hipMemcpyAsync(m->tmp_ranges.p, hr, sizeof(double) * total, hipMemcpyHostToDevice, m->stream));
    call.items.assign(1, CallItem{0, 1, n_base});
    call.penalize = penalize ? 1 : 0;
    call.refine = refine ? 1 : 0;
    call.coarse_angle_off = m->cfg.coarse_search_angle_offset;
    Slot &slot = m->slots[kAsyncSlots];
    slot.call = call;
    if ((rc = launch_call(m, slot))) return rc;
    return finish_call(m, slot, out);
}

// tiny_tf's planar Transform arithmetic as yag_slam_amd/transform.py spells it (same operations in the same order: the
// priors must be the bits the per-scan Python path produces)
static void tf_compose(const double a[3], const double b[3], double out[3]) { // a + b
    const double c = std::cos(a[2]), s = std::sin(a[2]);
    const double x = a[0] + c * b[0] - s * b[1], y = a[1] + s * b[0] + c * b[1];
    out[0] = x; out[1] = y; out[2] = a[2] + b[2];
}
static void tf_inverse(const double a[3], double out[3]) {
    const double c = std::cos(a[2]), s = std::sin(a[2]);
    const double x = -(c * a[0] + s * a[1]), y = -(-s * a[0] + c * a[1]);
    out[0] = x; out[1] = y; out[2] = -a[2];
}

// One synchronous step of ym_map_sequence: prior from the previous scan's pose, match, pose := result.
static int sequence_step_sync(ym_matcher *m, ym_scan *const *scans, const double *odom, int i, int buffer_len, int penalize,
                              int refine, ym_result *result) {
    double inv[3], diff[3], prior[3];
    tf_inverse(odom + 3 * (size_t)(i - 1), inv);              // query.odom_pose - last.odom_pose
    tf_compose(inv, odom + 3 * (size_t)i, diff);
    tf_compose(scans[i - 1]->pose, diff, prior);                // last.corrected_pose + that
    int rc = ym_scan_set_pose(scans[i], prior[0], prior[1], prior[2]);
    if (rc) return rc;
    const int first = std::max(0, i - buffer_len);
    Slot &slot = m->slots[kAsyncSlots];
    if (slot.in_flight && slot.call.slice)
        return set_err(YM_ERR_BUSY, "an angle-sliced match is in flight on this matcher: finish it (ym_match_slice_finish) first");
    slot.call = Call();
    if ((rc = build_single_call(m, scans[i], scans + first, i - first, penalize, refine, &slot.call, true))) { release_staged(slot.call); return rc; }
    if ((rc = launch_call(m, slot)) == YM_OK) rc = finish_call(m, slot, result);
    else if (slot.in_flight) (void)hipStreamSynchronize(m->stream);
    release_staged(slot.call);
    if (rc) return rc;
    if (result->status != 0) return YM_OK;
    return ym_scan_set_pose(scans[i], result->pose[0], result->pose[1], result->pose[2]);
}

// GraphSlam.process_scan's matcher work for ONE scan (graph_slam.py:320-337), for callers that get their scans one at a
// time: prior = chain[n_chain - 1]'s pose (+) (odom_query (-) odom_last), match against the chain, pose := result.
extern "C" int ym_process_scan(ym_matcher *m, ym_scan *query, ym_scan *const *chain, int n_chain, const double *odom_last,
                               const double *odom_query, int penalize, int refine, ym_result *result) {
    if (!m || !query || !chain || !odom_last || !odom_query || !result) return set_err(YM_ERR_INVALID, "null argument");
    if (n_chain < 1) return set_err(YM_ERR_INVALID, "process_scan needs at least one scan to match against");
    for (int i = 0; i < n_chain; i++)
        if (!chain[i]) return set_err(YM_ERR_INVALID, "null scan %d", i);
    double inv[3], diff[3], prior[3];
    tf_inverse(odom_last, inv);
    tf_compose(inv, odom_query, diff);
    tf_compose(chain[n_chain - 1]->pose, diff, prior);
    int rc = ym_scan_set_pose(query, prior[0], prior[1], prior[2]);
    if (rc) return rc;
    if ((rc = ym_match_scans(m, query, chain, n_chain, penalize, refine, result))) return rc;
    if (result->status != 0) return YM_OK;
    return ym_scan_set_pose(query, result->pose[0], result->pose[1], result->pose[2]);
}

// Steps [lo, hi) of ym_map_sequence enqueued back to back, no host round trip between them: step i's final_kernel leaves
// scan i's pose and scan i + 1's odometry prior on the device (seq_pose), the kernels of step i + 1 read them from there
// (YmScanRef::pose_dev), and the host -- which plans step i + 1 before step i has run -- sizes the raster from poses it
// dead-reckons with the odometry alone.  A step whose cells leave that prediction, that Karto would abort, or that needs
// a response expansion makes the device skip the rest (seq_fault); the caller repeats it synchronously.
// Returns the number of steps completed in *done (results and poses of [lo, lo + *done) are final).
static int sequence_segment_chained(ym_matcher *m, ym_scan *const *scans, const double *odom, int lo, int hi, int buffer_len,
                                    int penalize, int refine, ym_result *results, int *done) {
    *done = 0;
    DEV_GUARD(m->device);
    int rc;
    const int n_seg = hi - lo;
    if ((rc = m->seq_pose.ensure(4 + 3 * (size_t)n_seg))) return rc; // [0..2] the next step's prior, [4 + 3k ..] the pose of step lo + k
    if ((rc = m->seq_fault.ensure(1))) return rc;
    if ((rc = m->seq_results.ensure(sizeof(YmItemState) * (size_t)n_seg))) return rc;
    HIP_TRY(hipMemsetAsync(m->seq_fault.p, 0, sizeof(int32_t), m->stream));
    Slot &slot = m->slots[kAsyncSlots];
    if (slot.in_flight) return set_err(YM_ERR_BUSY, "the matcher's synchronous slot holds a call in flight");
    // the poses the caller set: the dead-reckoned priors below overwrite them, and the scans a fault (or an error) leaves
    // unmatched get them back -- only matched scans are touched, as the header says
    std::vector<double> caller_pose(3 * (size_t)n_seg);
    for (int k = 0; k < n_seg; k++)
        for (int c = 0; c < 3; c++) caller_pose[3 * (size_t)k + c] = scans[lo + k]->pose[c];
    int enqueued = 0, posed = 0;
    for (int i = lo; i < hi; i++, enqueued++) {
        double inv[3], diff[3], prior[3], next_diff[3] = {0, 0, 0};
        tf_inverse(odom + 3 * (size_t)(i - 1), inv);
        tf_compose(inv, odom + 3 * (size_t)i, diff);
        tf_compose(scans[i - 1]->pose, diff, prior); // (scan i - 1: its true pose for i == lo, else what the odometry predicts)
        if ((rc = ym_scan_set_pose(scans[i], prior[0], prior[1], prior[2]))) break;
        posed = i - lo + 1;
        if (i + 1 < hi) {
            tf_inverse(odom + 3 * (size_t)i, inv);
            tf_compose(inv, odom + 3 * (size_t)(i + 1), next_diff);
        }
        const int first = std::max(0, i - buffer_len);
        slot.call = Call();
        if ((rc = build_single_call(m, scans[i], scans + first, i - first, penalize, refine, &slot.call))) break;
        Call &call = slot.call;
        call.chain_step = i; // (>= 1)
        for (int k = 0; k < 3; k++) call.chain_next_diff[k] = next_diff[k];
        call.chain_out = reinterpret_cast<YmItemState *>(m->seq_results.dp) + (i - lo);
        call.chain_pose_out = m->seq_pose.p + 4 + 3 * (size_t)(i - lo);
        if (i > lo) call.scans[0].pose_dev = m->seq_pose.p;              // the query's prior
        for (int j = std::max(first, lo); j < i; j++)                    // base scans matched earlier in this segment: their results
            call.scans[(size_t)(1 + j - first)].pose_dev = m->seq_pose.p + 4 + 3 * (size_t)(j - lo);
        if ((rc = launch_call(m, slot))) break;
        slot.in_flight = false; // (collected below, from seq_results)
        bool all_direct = true;
        for (const CallScan &cs : slot.call.scans) all_direct = all_direct && cs.direct;
        if (!all_direct) { // (cannot happen: the caller admits scans with a trusted structure only) -- be safe:
            enqueued++;
            rc = set_err(YM_ERR_UNSUPPORTED, "device-chained step over a scan without a trusted structure");
            break;
        }
    }
    auto restore_from = [&](int k0) {
        for (int k = k0; k < posed; k++)
            (void)ym_scan_set_pose(scans[lo + k], caller_pose[3 * (size_t)k], caller_pose[3 * (size_t)k + 1], caller_pose[3 * (size_t)k + 2]);
    };
    int32_t fault = 0;
    hipError_t herr = hipMemcpyAsync(&fault, m->seq_fault.p, sizeof fault, hipMemcpyDeviceToHost, m->stream);
    if (herr == hipSuccess) herr = hipStreamSynchronize(m->stream);
    if (herr != hipSuccess) { restore_from(0); return set_err(YM_ERR_HIP, "%s", hipGetErrorString(herr)); }
    const int good = std::min(enqueued, fault > 0 ? fault - lo : enqueued);
    restore_from(good);
    const YmItemState *hs = reinterpret_cast<const YmItemState *>(m->seq_results.p);
    for (int k = 0; k < good; k++) {
        const int i = lo + k;
        state_to_result(m, slot, hs[k], 0, 0, &results[i]);
        (void)ym_scan_set_pose(scans[i], results[i].pose[0], results[i].pose[1], results[i].pose[2]);
    }
    *done = good;
    return rc;
}

int ym_map_sequence(ym_matcher *m, ym_scan *const *scans, const double *odom, int n, int start, int buffer_len,
                    int penalize, int refine, int device_chain, ym_result *results, int32_t *n_done) {
    if (!m || !scans || !odom || !results || !n_done) return set_err(YM_ERR_INVALID, "null argument");
    if (n < 0 || start < 0 || buffer_len < 1) return set_err(YM_ERR_INVALID, "bad trajectory length, start or chain length");
    *n_done = 0;
    for (int i = 0; i < n; i++)
        if (!scans[i]) return set_err(YM_ERR_INVALID, "null scan %d", i);
    const int begin = std::min(n, std::max(start, 1));
    for (int i = 0; i < begin; i++) std::memset(&results[i], 0, sizeof results[i]);
    *n_done = begin;
    static const bool debug_host = getenv("YM_DEBUG_HOST") != nullptr;
    // chained segments need Karto semantics (one pass structure), resident scans with a trusted structure, an inline
    // descriptor (chain + query <= YM_INLINE_SCANS) and the two-kernel finish
    const bool can_chain = device_chain && m->cfg.semantics == YM_SEM_KARTO && buffer_len + 1 <= YM_INLINE_SCANS &&
                           m->finish_form != 2;
    // segment length: a fault (the odometry drifted away from the matches, a response expansion) costs the rest of its
    // segment, so the length halves after one and doubles again after a segment that went through
    int seg_len = 128;
    int i = begin, n_sync = 0, n_segments = 0;
    while (i < n) {
        int rc;
        bool chained = false;
        if (can_chain && n - i >= 2) {
            // every scan a chained step touches must carry a trusted structure (no point-cache slot then, whose pose the
            // host would not know): the segment ends before the first step that meets another kind
            const int sem = 0;
            auto trusted = [&](int j) {
                scan_resolve(scans[j]);
                return scans[j]->id != 0 && scans[j]->n > 0 && scans[j]->gov_ok[sem] && m->use_scan_structure;
            };
            int hi = std::min(n, i + seg_len);
            // (the structure is trusted within YM_CHAIN_POSE_LIMIT of the origin: stay well inside with predicted poses)
            bool chain_ok = std::fabs(scans[i - 1]->pose[0]) < 0.9 * YM_CHAIN_POSE_LIMIT && std::fabs(scans[i - 1]->pose[1]) < 0.9 * YM_CHAIN_POSE_LIMIT &&
                            std::fabs(scans[i - 1]->pose[2]) < 0.9 * YM_CHAIN_HEADING_LIMIT;
            for (int j = std::max(0, i - buffer_len); j < i; j++) chain_ok = chain_ok && trusted(j);
            for (int j = i; j < hi; j++)
                if (!trusted(j)) { hi = j; break; }
            if (chain_ok && hi - i >= 2) {
                int done = 0;
                rc = sequence_segment_chained(m, scans, odom, i, hi, buffer_len, penalize, refine, results, &done);
                if (rc) { *n_done = i + done; return rc; }
                i += done;
                *n_done = i;
                n_segments++;
                m->seq_segments++;
                chained = done == hi - (i - done);
                seg_len = chained ? std::min(128, seg_len * 2) : std::max(8, seg_len / 2);
                if (chained) continue; // (else: scan i faulted -- repeat it the ordinary way)
                m->seq_faults++;
            }
        }
        if (i >= n) break;
        if ((rc = sequence_step_sync(m, scans, odom, i, buffer_len, penalize, refine, &results[i]))) return rc;
        n_sync++;
        m->seq_sync_steps++;
        if (results[i].status != 0) return YM_OK;
        *n_done = ++i;
    }
    if (debug_host) fprintf(stderr, "[ym] map_sequence: %d scans, %d chained segments, %d synchronous steps\n", n - begin, n_segments, n_sync);
    return YM_OK;
}

int ym_async_slots(const ym_matcher *m) { return m ? kAsyncSlots : YM_ERR_INVALID; }

int ym_match_scans_async(ym_matcher *m, const ym_scan *query, const ym_scan *const *base, int n_base, int penalize,
                         int refine, int slot_idx) {
    if (!m) return set_err(YM_ERR_INVALID, "null matcher");
    if (slot_idx < 0 || slot_idx >= kAsyncSlots) return set_err(YM_ERR_INVALID, "slot %d out of range", slot_idx);
    Slot &slot = m->slots[slot_idx];
    if (slot.in_flight) return set_err(YM_ERR_BUSY, "slot %d still holds an uncollected call", slot_idx);
    Call call;
    int rc = build_single_call(m, query, base, n_base, penalize, refine, &call);
    if (rc) return rc;
    slot.call = call;
    return launch_call(m, slot);
}

int ym_wait(ym_matcher *m, int slot_idx, ym_result *out) {
    if (!m || !out) return set_err(YM_ERR_INVALID, "null argument");
    if (slot_idx < 0 || slot_idx >= kAsyncSlots) return set_err(YM_ERR_INVALID, "slot %d out of range", slot_idx);
    Slot &slot = m->slots[slot_idx];
    std::vector<ym_result> tmp(std::max(1, slot.n_items));
    int rc = finish_call(m, slot, tmp.data());
    if (rc) return rc;
    *out = tmp[0];
    return YM_OK;
}

// items of a batch: item c = queries[item_query[c]] against scans[chain_offsets[c] .. chain_offsets[c + 1])
static ym_batch *batch_new(ym_matcher *m, const ym_scan *const *queries, int n_queries, bool per_item, const ym_scan *const *scans,
                           const int32_t *chain_offsets, int n_chains) {
    if (!m || !queries || !chain_offsets) { set_err(YM_ERR_INVALID, "null argument"); return nullptr; }
    if (n_chains <= 0) { set_err(YM_ERR_INVALID, "n_chains must be > 0"); return nullptr; }
    const int n_scans = chain_offsets[n_chains];
    if (chain_offsets[0] != 0 || n_scans < 0 || (n_scans > 0 && !scans)) { set_err(YM_ERR_INVALID, "bad scan list"); return nullptr; }
    for (int c = 0; c < n_chains; c++)
        if (chain_offsets[c + 1] < chain_offsets[c]) { set_err(YM_ERR_INVALID, "chain_offsets must be non-decreasing"); return nullptr; }
    for (int i = 0; i < n_queries; i++) {
        if (!queries[i]) { set_err(YM_ERR_INVALID, "query %d is null", i); return nullptr; }
        if (queries[i]->device != m->device) { set_err(YM_ERR_INVALID, "query scan %d lives on another device", i); return nullptr; }
    }
    for (int i = 0; i < n_scans; i++)
        if (!scans[i] || scans[i]->device != m->device) { set_err(YM_ERR_INVALID, "scan %d is null or lives on another device", i); return nullptr; }
    static std::atomic<uint64_t> next_uid{1};
    ym_batch *b = new ym_batch();
    b->uid = next_uid.fetch_add(1);
    b->item_query.resize(n_chains, 0);
    if (per_item) { // a query object that serves several items is projected, and its pair lists are built, once
        std::unordered_map<const ym_scan *, int32_t> seen;
        for (int c = 0; c < n_chains; c++) {
            auto it = seen.find(queries[c]);
            if (it == seen.end()) {
                it = seen.emplace(queries[c], (int32_t)b->queries.size()).first;
                b->queries.push_back(queries[c]);
            }
            b->item_query[c] = it->second;
        }
    } else {
        b->queries.push_back(queries[0]);
    }
    b->query_hints.assign(b->queries.size(), -1);
    b->scans.assign(scans, scans + n_scans);
    b->offsets.assign(chain_offsets, chain_offsets + n_chains + 1);
    return b;
}

ym_batch *ym_batch_create(ym_matcher *m, const ym_scan *query, const ym_scan *const *scans, const int32_t *chain_offsets,
                          int n_chains) {
    if (!query) { set_err(YM_ERR_INVALID, "null argument"); return nullptr; }
    return batch_new(m, &query, 1, false, scans, chain_offsets, n_chains);
}

ym_batch *ym_pairs_create(ym_matcher *m, const ym_scan *const *queries, const ym_scan *const *scans, const int32_t *chain_offsets,
                          int n_items) {
    return batch_new(m, queries, n_items, true, scans, chain_offsets, n_items);
}

void ym_batch_destroy(ym_batch *b) { delete b; }

int ym_batch_size(const ym_batch *b) { return b ? (int)b->offsets.size() - 1 : YM_ERR_INVALID; }

int ym_batch_run_async(ym_matcher *m, const ym_batch *b, int penalize, int refine, int slot_idx, int64_t chain_id_base,
                       void *dev_best_out) {
    if (!m || !b) return set_err(YM_ERR_INVALID, "null argument");
    if (slot_idx < 0 || slot_idx >= kAsyncSlots) return set_err(YM_ERR_INVALID, "slot %d out of range", slot_idx);
    Slot &slot = m->slots[slot_idx];
    if (slot.in_flight) return set_err(YM_ERR_BUSY, "slot %d still holds an uncollected call", slot_idx);
    const int n_chains = (int)b->offsets.size() - 1, n_scans = (int)b->scans.size();
    Call &call = slot.call;
    const uint64_t epoch = g_pose_epoch.load(std::memory_order_relaxed);
    // the slot still holds this batch's Call and no scan anywhere has moved since it was built: nothing to rebuild
    // (40 961 scattered ym_scan objects are not even looked at; 1.7 ms per enqueue of 4096 chains otherwise)
    const int nq = (int)b->queries.size();
    const bool same = call.batch_uid == b->uid && call.pose_epoch == epoch && call.scans.size() == (size_t)nq + (size_t)n_scans &&
                      call.penalize == (penalize ? 1 : 0) && call.refine == (refine ? 1 : 0) && !call.slice && !call.chain_step;
    int rc;
    if (!same) {
        call = Call();
        call.scans.resize((size_t)nq + (size_t)n_scans);
        b->cache_hints.resize(n_scans, -1);
        b->query_hints.resize(nq, -1);
        // (40 960 scattered ym_scan objects: ask for the ones ahead while this one is copied -- the loop was 3.3 ms of cache misses)
        auto touch = [](const ym_scan *s) {
            if (!s) return;
            const char *p = reinterpret_cast<const char *>(s);
            __builtin_prefetch(p, 0, 1);
            __builtin_prefetch(p + 64, 0, 1);
            __builtin_prefetch(p + 128, 0, 1);
            __builtin_prefetch(p + 192, 0, 1);
        };
        auto fill = [&](int lo, int hi) -> int {
            for (int i = lo; i < hi && i < lo + 16; i++) touch(b->scans[i]);
            for (int i = lo; i < hi; i++) {
                if (i + 16 < hi) touch(b->scans[i + 16]);
                const int r_ = scan_to_call(b->scans[i], m->cfg.semantics, &call.scans[nq + i]);
                if (r_) return r_;
                call.scans[nq + i].cache_hint = b->cache_hints[i];
            }
            return YM_OK;
        };
        for (int i = 0; i < nq; i++) {
            if (i + 8 < nq) touch(b->queries[i + 8]);
            if ((rc = scan_to_call(b->queries[i], m->cfg.semantics, &call.scans[i]))) return rc;
            call.scans[i].qcache_hint = b->query_hints[i];
            call.scans[i].query_uses = b->queries[i]->query_uses.fetch_add(1, std::memory_order_relaxed);
        }
        // (tried: four threads, a quarter each -- 3.9 -> 4.4 ms, and the caller's next ym_scans_set_poses 0.75 -> 2.7 ms: the
        //  scans' cache lines then live in other cores' caches)
        if ((rc = fill(0, n_scans))) return rc;
        call.items.resize(n_chains);
        for (int c = 0; c < n_chains; c++) call.items[c] = CallItem{b->item_query[c], nq + b->offsets[c], b->offsets[c + 1] - b->offsets[c]};
        call.penalize = penalize ? 1 : 0;
        call.refine = refine ? 1 : 0;
        call.batch_uid = b->uid;
        call.pose_epoch = epoch;
    }
    if (same) {
        // the batch is run AGAIN through a Call that was built on its queries' first use (no slot of the point cache then, see plan_cache):
        // this is their second use -- they get their slots now, which takes one more planning of the call instead of a replay
        bool first_use_plan = false;
        for (int i = 0; i < nq; i++)
            if (call.scans[i].query_uses == 0) { call.scans[i].query_uses = 1; first_use_plan = true; }
        if (first_use_plan) call.plan_clean = false;
    }
    call.coarse_angle_off = m->cfg.coarse_search_angle_offset;
    slot.chain_id_base = chain_id_base;
    slot.dev_best_out = dev_best_out;
    slot.dev_best_user = dev_best_out;
    rc = launch_call(m, slot);
    slot.dev_best_out = nullptr; // a response-expansion re-run must not overwrite the caller's buffer
    if (!same) {
        for (int i = 0; i < n_scans; i++) b->cache_hints[i] = call.scans[nq + i].cache_hint;
        for (int i = 0; i < nq; i++) b->query_hints[i] = call.scans[i].qcache_hint;
    }
    return rc;
}

int ym_batch_wait(ym_matcher *m, int slot_idx, ym_result *per_chain, ym_result *best, int32_t *best_chain) {
    if (!m) return set_err(YM_ERR_INVALID, "null matcher");
    if (slot_idx < 0 || slot_idx >= kAsyncSlots) return set_err(YM_ERR_INVALID, "slot %d out of range", slot_idx);
    Slot &slot = m->slots[slot_idx];
    std::vector<ym_result> res(std::max(1, slot.n_items));
    int rc = finish_call(m, slot, res.data());
    if (rc) return rc;
    const int n = slot.n_items;
    int bi = 0;
    for (int c = 1; c < n; c++)
        if (res[c].response > res[bi].response) bi = c;
    if (per_chain) std::memcpy(per_chain, res.data(), sizeof(ym_result) * n);
    if (best) *best = res[bi];
    if (best_chain) *best_chain = bi;
    return YM_OK;
}

static int batch_run_once(ym_matcher *m, ym_batch *b, int penalize, int refine, ym_result *per_chain, ym_result *best, int32_t *best_chain) {
    if (!b) return YM_ERR_INVALID;
    // use the last async slot that is free
    int slot_idx = -1;
    for (int i = kAsyncSlots - 1; i >= 0; i--)
        if (!m->slots[i].in_flight) { slot_idx = i; break; }
    int rc = slot_idx < 0 ? set_err(YM_ERR_BUSY, "all async slots are in flight")
                          : ym_batch_run_async(m, b, penalize, refine, slot_idx, 0, nullptr);
    if (rc == YM_OK) rc = ym_batch_wait(m, slot_idx, per_chain, best, best_chain);
    ym_batch_destroy(b);
    return rc;
}

int ym_match_batch(ym_matcher *m, const ym_scan *query, const ym_scan *const *scans, const int32_t *chain_offsets,
                   int n_chains, int penalize, int refine, ym_result *per_chain, ym_result *best, int32_t *best_chain) {
    return batch_run_once(m, ym_batch_create(m, query, scans, chain_offsets, n_chains), penalize, refine, per_chain, best, best_chain);
}

int ym_match_pairs(ym_matcher *m, const ym_scan *const *queries, const ym_scan *const *scans, const int32_t *chain_offsets,
                   int n_items, int penalize, int refine, ym_result *per_item) {
    return batch_run_once(m, ym_pairs_create(m, queries, scans, chain_offsets, n_items), penalize, refine, per_item, nullptr, nullptr);
}

// ---- one match split by coarse angle over several matchers (one per GPU): BASELINE configs[4] on 8 GPUs
int ym_coarse_dims(const ym_matcher *m, int32_t dims[3]) {
    if (!m || !dims) return set_err(YM_ERR_INVALID, "null argument");
    if (m->cfg.semantics != YM_SEM_KARTO) return set_err(YM_ERR_UNSUPPORTED, "the Karto lattice only");
    const YmGeom &g = m->geom;
    const YmLattice l = make_lattice(g, 0.5 * (g.side - 1) * g.res, 2 * g.res, m->cfg.coarse_search_angle_offset,
                                     m->cfg.coarse_angle_resolution, 0, 0);
    dims[0] = l.nx; dims[1] = l.ny; dims[2] = l.nt;
    return YM_OK;
}

int ym_match_slice_begin(ym_matcher *m, const ym_scan *query, const ym_scan *const *base, int n_base, int penalize,
                         int refine, int k_begin, int k_end, double *dev_resp, double *dev_probs) {
    if (!m || !dev_resp || !dev_probs) return set_err(YM_ERR_INVALID, "null argument");
    if (k_begin < 0 || k_end < k_begin) return set_err(YM_ERR_INVALID, "bad angle slice [%d, %d)", k_begin, k_end);
    Slot &slot = m->slots[kAsyncSlots];
    if (slot.in_flight && !slot.call.slice) return set_err(YM_ERR_BUSY, "the synchronous slot is in flight");
    if (slot.in_flight) { // another slice of the same volume scored by this matcher (tests; a rank owning two blocks)
        HIP_TRY(hipEventSynchronize(slot.done));
        slot.in_flight = false;
    }
    Call call;
    int rc = build_single_call(m, query, base, n_base, penalize, refine, &call);
    if (rc) return rc;
    call.slice = true;
    call.k_begin = k_begin; call.k_end = k_end;
    call.ext_resp = dev_resp; call.ext_probs = dev_probs;
    slot.call = call;
    return launch_call(m, slot); // stops after the score stage; stream-ordered, no host wait
}

int ym_match_slice_finish(ym_matcher *m, ym_result *out) {
    if (!m || !out) return set_err(YM_ERR_INVALID, "null argument");
    Slot &slot = m->slots[kAsyncSlots];
    if (!slot.in_flight || !slot.call.slice) return set_err(YM_ERR_BUSY, "no angle-sliced match in flight");
    DEV_GUARD(m->device);
    const CallPlan &P = slot.plan;
    // the caller has completed the response volume (all slices gathered) and the per-(x, y) maxima (max over all
    // slices) on this stream: block maxima of the whole volume, then the ordinary finish stage
    hipLaunchKernelGGL(ym::blockmax_kernel, dim3(P.cell_blocks, P.lc.nt), dim3(YM_SCORE_THREADS), 0, m->stream, P.resp, P.lc.nx * P.lc.ny, m->blockmax.p);
    enqueue_finish(m, slot, P);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(slot.done, m->stream));
    const Call sliced = slot.call;
    slot.call.slice = false;
    slot.call.ext_resp = slot.call.ext_probs = nullptr;
    slot.call.k_begin = 0; slot.call.k_end = -1;
    ym_result r;
    // Karto's response expansion re-runs the match with a wider angle range: done by finish_call on this matcher alone,
    // over the whole (wider) lattice -- every rank does the same and gets the same result
    int rc = finish_call(m, slot, &r);
    (void)sliced;
    if (rc) return rc;
    *out = r;
    return YM_OK;
}
