// ym_abi_scans.hpp -- C ABI: resident scans (single and bulk creation, poses, destruction)
// Part of yagmatch.hip (included inside its extern "C" block); not a header of its own.
// ---- scans
// everything of a new scan the host computes from its descriptor (no device work)
static ym_scan *scan_host_side(int device, const ym_scan_desc *d) {
    static std::atomic<uint64_t> next_id{1};
    ym_scan *s = new ym_scan();
    s->id = next_id.fetch_add(1);
    s->device = device;
    s->n = d->n;
    s->min_angle = d->min_angle; s->max_angle = d->max_angle; s->angle_inc = d->angle_increment;
    s->min_range = d->min_range; s->max_range = d->max_range; s->range_threshold = d->range_threshold;
    s->pose[0] = d->pose[0]; s->pose[1] = d->pose[1]; s->pose[2] = d->pose[2];
    s->d_ranges = nullptr;
    max_valid_ranges(d->ranges, d->n, d->min_range, d->range_threshold, &s->max_valid_karto, &s->max_valid_yagpy);
    local_bbox(d->ranges, d->n, d->min_angle, d->angle_increment, d->range_threshold, s->lbox);
    world_bbox(s->lbox, s->pose, s->wbox);
    s->beam_spacing = median_beam_spacing(d->ranges, d->n, d->min_range, d->range_threshold, d->angle_increment);
    return s;
}
// a scan's block of device memory: ranges[n], the chain structure per semantics ([2][n][2] + [2][n] ints), 16 spare bytes
struct ScanLayout {
    size_t ranges_bytes, gov_bytes, cidx_bytes, total;
    explicit ScanLayout(int n) {
        const size_t n1 = (size_t)std::max(1, n);
        ranges_bytes = align_up(sizeof(double) * n1, 16);
        gov_bytes = align_up(sizeof(int32_t) * 2 * n1, 16);
        cidx_bytes = align_up(sizeof(int32_t) * n1, 16);
        total = ranges_bytes + 2 * gov_bytes + 2 * cidx_bytes + 16;
    }
};

ym_scan *ym_scan_create(int device, const ym_scan_desc *d) {
    if (check_desc(d) != YM_OK) return nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_err(YM_ERR_NO_DEVICE, "no HIP device available"); return nullptr; }
    if (device < 0 || device >= n) { set_err(YM_ERR_NO_DEVICE, "device %d out of range [0, %d)", device, n); return nullptr; }
    ym_scan *s = scan_host_side(device, d);
    DevGuard guard(device);
    const ScanLayout L(d->n);
    const size_t ranges_bytes = L.ranges_bytes, gov_bytes = L.gov_bytes, cidx_bytes = L.cidx_bytes, total = L.total;
    const bool structured = d->n > 0 && d->n <= YM_MAX_BEAMS;
    if (!guard.ok) { set_err(YM_ERR_HIP, "cannot select device %d", device); delete s; return nullptr; }
    if (structured) {
        unsigned char *base = nullptr;
        {
            ScanPool &p = scan_pool(device);
            std::lock_guard<std::mutex> lk(p.mu);
            if (pool_create_scan(p, s, d->ranges, total, &base) != YM_OK) { delete s; return nullptr; }
        }
        s->d_gov[0] = reinterpret_cast<int32_t *>(base + ranges_bytes);
        s->d_gov[1] = reinterpret_cast<int32_t *>(base + ranges_bytes + gov_bytes);
        s->d_cidx[0] = reinterpret_cast<int32_t *>(base + ranges_bytes + 2 * gov_bytes);
        s->d_cidx[1] = reinterpret_cast<int32_t *>(base + ranges_bytes + 2 * gov_bytes + cidx_bytes);
        return s;
    }
    // no readings, or more than the kernels stage at once (such a scan is refused by the matchers): a plain allocation
    if (hipMalloc(reinterpret_cast<void **>(&s->d_ranges), total) != hipSuccess) {
        set_err(YM_ERR_HIP, "cannot allocate device ranges");
        delete s;
        return nullptr;
    }
    if (d->n > 0 && hipMemcpy(s->d_ranges, d->ranges, sizeof(double) * d->n, hipMemcpyHostToDevice) != hipSuccess) {
        set_err(YM_ERR_HIP, "cannot upload ranges");
        (void)hipFree(s->d_ranges);
        delete s;
        return nullptr;
    }
    return s;
}

int ym_scan_set_pose(ym_scan *s, double x, double y, double heading) {
    if (!s) return set_err(YM_ERR_INVALID, "null scan");
    g_pose_epoch.fetch_add(1, std::memory_order_relaxed);
    s->pose[0] = x; s->pose[1] = y; s->pose[2] = heading;
    world_bbox(s->lbox, s->pose, s->wbox);
    return YM_OK;
}

int ym_scans_set_poses(ym_scan *const *scans, const double *xyz, int n) {
    if (n < 0 || (n > 0 && (!scans || !xyz))) return set_err(YM_ERR_INVALID, "null argument");
    for (int i = 0; i < n; i++)
        if (!scans[i]) return set_err(YM_ERR_INVALID, "null scan %d", i); // (nothing is written unless every scan can be)
    if (n > 0) g_pose_epoch.fetch_add(1, std::memory_order_relaxed);
    // (tens of thousands of scattered heap objects: the loop is a chain of cache misses unless the next ones are asked for early)
    auto touch = [](const ym_scan *s) {
        const char *p = reinterpret_cast<const char *>(s);
        __builtin_prefetch(p + offsetof(ym_scan, pose), 1, 1);
        __builtin_prefetch(p + offsetof(ym_scan, wbox), 1, 1);
    };
    for (int i = 0; i < n && i < 16; i++) touch(scans[i]);
    for (int i = 0; i < n; i++) {
        if (i + 16 < n) touch(scans[i + 16]);
        ym_scan *s = scans[i];
        s->pose[0] = xyz[3 * (size_t)i]; s->pose[1] = xyz[3 * (size_t)i + 1]; s->pose[2] = xyz[3 * (size_t)i + 2];
        world_bbox(s->lbox, s->pose, s->wbox);
    }
    return YM_OK;
}

int ym_scan_get_pose(const ym_scan *s, double pose[3]) {
    if (!s || !pose) return set_err(YM_ERR_INVALID, "null argument");
    pose[0] = s->pose[0]; pose[1] = s->pose[1]; pose[2] = s->pose[2];
    return YM_OK;
}

int ym_scan_size(const ym_scan *s) { return s ? s->n : YM_ERR_INVALID; }

int ym_scan_structure_trusted(const ym_scan *s, int semantics) {
    if (!s) return set_err(YM_ERR_INVALID, "null scan");
    scan_resolve(s);
    return s->gov_ok[semantics == YM_SEM_YAGPY ? 1 : 0] ? 1 : 0;
}

void ym_scan_destroy(ym_scan *s) {
    if (!s) return;
    DevGuard guard(s->device);
    scan_resolve(s); // (its creation launch writes into the block)
    if (s->block_bytes) {
        ScanPool &p = scan_pool(s->device);
        std::lock_guard<std::mutex> lk(p.mu);
        p.parked.push_back({s->d_ranges, s->block_bytes});
    } else if (s->d_ranges) {
        (void)hipFree(s->d_ranges);
    }
    delete s;
}

// n scans at once.  The host side of every scan (bounding box, longest reading, median beam spacing, the copy of its readings into
// pinned memory) is the same code as ym_scan_create's, spread over a few threads; the device side is ONE pool transaction, ONE upload
// and ONE launch of structure_many_kernel per chunk of kBulkChunk scans, two chunks in flight.  The scans come back resolved (no
// staging slot, nothing left to wait for).
int ym_scans_create(int device, const ym_scan_desc *descs, int n, ym_scan **out) {
    if (n < 0 || (n > 0 && (!descs || !out))) return set_err(YM_ERR_INVALID, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_err(YM_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return set_err(YM_ERR_NO_DEVICE, "device %d out of range [0, %d)", device, ndev);
    for (int i = 0; i < n; i++) {
        int rc = check_desc(&descs[i]);
        if (rc) return rc;
        out[i] = nullptr;
    }
    DEV_GUARD(device);
    ScanPool &p = scan_pool(device);
    constexpr int kBulkChunk = 2048;
    auto fail = [&](int rc) { // all or nothing
        std::string msg = g_err;
        (void)hipDeviceSynchronize();
        for (int i = 0; i < n; i++)
            if (out[i]) { out[i]->stage = nullptr; ym_scan_destroy(out[i]); out[i] = nullptr; }
        g_err = msg;
        return rc;
    };
    struct Pending { int lo = 0, hi = 0, buf = -1; size_t info_off = 0; };
    Pending pending[2];
    auto release_buf = [&](int b) {
        std::lock_guard<std::mutex> lk(p.mu);
        p.bulk[b].busy = false;
    };
    auto finish = [&](Pending &pd) -> int { // the chunk's launch is complete: its scans' info words
        if (pd.buf < 0) return YM_OK;
        ScanPool::Bulk &bk = p.bulk[pd.buf];
        const hipError_t he = hipEventSynchronize(bk.done);
        if (he != hipSuccess) { release_buf(pd.buf); pd.buf = -1; return set_err(YM_ERR_HIP, "scan creation failed on the device: %s", hipGetErrorString(he)); }
        const int32_t *info = reinterpret_cast<const int32_t *>(bk.host + pd.info_off);
        int k = 0;
        for (int i = pd.lo; i < pd.hi; i++) {
            ym_scan *s = out[i];
            if (!s->block_bytes) continue; // (an odd one: created the single way)
            s->cnp[0] = info[4 * k]; s->gov_ok[0] = info[4 * k + 1] == 0;
            s->cnp[1] = info[4 * k + 2]; s->gov_ok[1] = info[4 * k + 3] == 0;
            k++;
        }
        release_buf(pd.buf);
        pd.buf = -1;
        return YM_OK;
    };
    auto fail_all = [&](int rc) {
        for (Pending &pd : pending)
            if (pd.buf >= 0) { (void)hipEventSynchronize(p.bulk[pd.buf].done); release_buf(pd.buf); pd.buf = -1; }
        return fail(rc);
    };
    for (int lo = 0, chunk = 0; lo < n; lo += kBulkChunk, chunk++) {
        const int hi = std::min(n, lo + kBulkChunk), m = hi - lo, slot = chunk & 1;
        int rc;
        if ((rc = finish(pending[slot]))) return fail_all(rc); // (the chunk before last)
        int buf = -1;
        // layout of the chunk's staging buffer: [StructureArgs x m][info int32[4] x m][readings, 16-byte aligned per scan]
        std::vector<size_t> roff(m + 1);
        const size_t table_bytes = align_up(sizeof(ym::StructureArgs) * m, 256), info_off = table_bytes, info_bytes = align_up(sizeof(int32_t) * 4 * m, 256);
        size_t at = table_bytes + info_bytes;
        int max_n = 1;
        for (int i = 0; i < m; i++) {
            roff[i] = at;
            const int ni = descs[lo + i].n;
            if (ni > 0 && ni <= YM_MAX_BEAMS) { at += align_up(sizeof(double) * ni, 16); max_n = std::max(max_n, ni); }
        }
        roff[m] = at;
        for (int tries = 0; buf < 0; tries++) { // a staging buffer nobody holds (twelve: six creating threads at two chunks each)
            {
                std::lock_guard<std::mutex> lk(p.mu);
                if ((rc = pool_init(p, device))) return fail_all(rc);
                for (int b = 0; b < ScanPool::kBulkBuffers && buf < 0; b++)
                    if (!p.bulk[b].busy && (p.bulk[b].cap >= at || tries > 0)) { p.bulk[b].busy = true; buf = b; } // (first one that is large enough already)
            }
            if (buf < 0 && tries > 0) std::this_thread::yield();
        }
        {
            std::lock_guard<std::mutex> lk(p.mu);
            ScanPool::Bulk &bk = p.bulk[buf];
            if (at > bk.cap) {
                if (bk.host) (void)hipHostFree(bk.host);
                if (bk.dev) (void)hipFree(bk.dev);
                bk.host = bk.dev = nullptr; bk.cap = 0;
                const size_t want = align_up(at + at / 4, 4096);
                if (hipHostMalloc(reinterpret_cast<void **>(&bk.host), want, hipHostMallocDefault) != hipSuccess ||
                    hipMalloc(reinterpret_cast<void **>(&bk.dev), want) != hipSuccess) {
                    (void)hipGetLastError();
                    bk.busy = false;
                    return fail_all(set_err(YM_ERR_HIP, "cannot allocate %zu bytes of staging memory for %d scans", want, m));
                }
                bk.cap = want;
            }
            if (!bk.done && hipEventCreateWithFlags(&bk.done, hipEventDisableTiming) != hipSuccess) { bk.busy = false; return fail_all(set_err(YM_ERR_HIP, "cannot create an event")); }
            if (!p.bulk_streams[0]) {
                int lo_p = 0, hi_p = 0;
                (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p); // (numerically lowest = highest priority)
                for (hipStream_t &bs : p.bulk_streams)
                    if (hipStreamCreateWithPriority(&bs, hipStreamNonBlocking, hi_p) != hipSuccess) { (void)hipGetLastError(); bs = p.streams[0]; }
            }
        }
        ScanPool::Bulk &bk = p.bulk[buf];
        // the host side of every scan, and its readings into the pinned buffer: a few threads, a contiguous share each
        {
            const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
            const int nthreads = (int)std::max(1u, std::min({hw, 16u, (unsigned)(m / 128 + 1)}));
            auto work = [&](int t) {
                const int a0 = (int)((int64_t)m * t / nthreads), a1 = (int)((int64_t)m * (t + 1) / nthreads);
                for (int i = a0; i < a1; i++) {
                    const ym_scan_desc &d = descs[lo + i];
                    if (!(d.n > 0 && d.n <= YM_MAX_BEAMS)) continue; // (created the single way below)
                    out[lo + i] = scan_host_side(device, &d);
                    std::memcpy(bk.host + roff[i], d.ranges, sizeof(double) * d.n);
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nthreads; t++) th.emplace_back(work, t);
            work(0);
            for (auto &t : th) t.join();
        }
        // one pool transaction: a block per scan, its argument record
        ym::StructureArgs *table = reinterpret_cast<ym::StructureArgs *>(bk.host);
        int k = 0;
        {
            std::lock_guard<std::mutex> lk(p.mu);
            if (p.parked.size() >= kRecycleAt) pool_seal(p);
            pool_reap(p);
            {
                std::unordered_map<size_t, size_t> need; // blocks per size (one size, normally)
                for (int i = 0; i < m; i++)
                    if (out[lo + i]) need[align_up(ScanLayout(out[lo + i]->n).total, 1024)]++;
                for (auto &kv : need) pool_reserve(p, kv.first, kv.second);
            }
            for (int i = 0; i < m; i++) {
                ym_scan *s = out[lo + i];
                if (!s) continue;
                const ScanLayout L(s->n);
                const size_t bytes = align_up(L.total, 1024);
                unsigned char *base = static_cast<unsigned char *>(pool_block(p, bytes, false));
                if (!base) { bk.busy = false; return fail_all(set_err(YM_ERR_HIP, "cannot allocate device ranges")); }
                s->d_ranges = reinterpret_cast<double *>(base);
                s->block_bytes = bytes;
                s->d_gov[0] = reinterpret_cast<int32_t *>(base + L.ranges_bytes);
                s->d_gov[1] = reinterpret_cast<int32_t *>(base + L.ranges_bytes + L.gov_bytes);
                s->d_cidx[0] = reinterpret_cast<int32_t *>(base + L.ranges_bytes + 2 * L.gov_bytes);
                s->d_cidx[1] = reinterpret_cast<int32_t *>(base + L.ranges_bytes + 2 * L.gov_bytes + L.cidx_bytes);
                ym::StructureArgs &sa = table[k];
                std::memset(&sa, 0, sizeof sa);
                sa.sr.ranges = reinterpret_cast<const double *>(bk.dev + roff[i]); sa.sr.n = s->n; sa.sr.min_angle = s->min_angle; sa.sr.angle_inc = s->angle_inc;
                sa.sr.min_range = s->min_range; sa.sr.range_threshold = s->range_threshold;
                sa.gov[0] = s->d_gov[0]; sa.gov[1] = s->d_gov[1]; sa.cidx[0] = s->d_cidx[0]; sa.cidx[1] = s->d_cidx[1];
                sa.info = reinterpret_cast<int32_t *>(bk.dev + info_off) + 4 * k;
                sa.ranges_out = s->d_ranges;
                k++;
            }
            if (k > 0) {
                hipStream_t st = p.bulk_streams[slot];
                bool ok = hipMemcpyAsync(bk.dev, bk.host, at, hipMemcpyHostToDevice, st) == hipSuccess;
                if (ok) {
                    hipLaunchKernelGGL(ym::structure_many_kernel<512>, dim3(2, k), dim3(512), YM_PREP_LDS_BYTES(max_n), st, reinterpret_cast<const ym::StructureArgs *>(bk.dev));
                    ok = hipGetLastError() == hipSuccess &&
                         hipMemcpyAsync(bk.host + info_off, bk.dev + info_off, sizeof(int32_t) * 4 * k, hipMemcpyDeviceToHost, st) == hipSuccess &&
                         hipEventRecord(bk.done, st) == hipSuccess;
                }
                if (!ok) { bk.busy = false; return fail_all(set_err(YM_ERR_HIP, "uploading %d scans failed: %s", k, hipGetErrorString(hipGetLastError()))); }
                pending[slot].lo = lo; pending[slot].hi = hi; pending[slot].buf = buf; pending[slot].info_off = info_off;
            } else {
                bk.busy = false;
            }
        }
        // scans without readings, or with more than the kernels stage at once: the single way (a plain allocation)
        for (int i = 0; i < m; i++)
            if (!out[lo + i]) {
                out[lo + i] = ym_scan_create(device, &descs[lo + i]);
                if (!out[lo + i]) return fail_all(YM_ERR_HIP);
            }
    }
    for (Pending &pd : pending) {
        int rc = finish(pd);
        if (rc) return fail_all(rc);
    }
    return YM_OK;
}

void ym_scans_destroy(ym_scan *const *scans, int n) {
    if (!scans || n <= 0) return;
    // by device (normally one): one pool transaction for all of a device's scans
    for (int i = 0; i < n;) {
        if (!scans[i]) { i++; continue; }
        const int device = scans[i]->device;
        DevGuard guard(device);
        ScanPool &p = scan_pool(device);
        std::vector<void *> plain;
        {
            std::lock_guard<std::mutex> lk(p.mu);
            int j = i;
            for (; j < n && (!scans[j] || scans[j]->device == device); j++) {
                ym_scan *s = scans[j];
                if (!s) continue;
                if (s->stage) stage_wait(p, *s->stage); // (its creation launch writes into the block)
                if (s->block_bytes) p.parked.push_back({s->d_ranges, s->block_bytes});
                else if (s->d_ranges) plain.push_back(s->d_ranges);
                delete s;
            }
            i = j;
        }
        for (void *q : plain) (void)hipFree(q);
    }
}
