// ym_host_pool.hpp -- host runtime: ym_scan and the per-device scan pool (staged creation, bulk creation's staging buffers, event-based recycling)
// Part of yagmatch.hip (included at file scope); not a header of its own.
struct ym_scan {
    uint64_t id; // unique per created scan: the key of the matchers' point caches
    int device;
    double *d_ranges;
    int32_t *d_gov[2] = {nullptr, nullptr}; // trigger-chain structure per semantics (structure_kernel), inside d_ranges' allocation;
    int32_t *d_cidx[2] = {nullptr, nullptr}; // ... the compaction (beam -> point reading) that goes with it,
    // ... the number of point readings, and whether the structure holds at every pose (no distance test near the threshold):
    // written by structure_kernel into the scan's staging slot and read when the scan is first used (scan_resolve)
    mutable int32_t cnp[2] = {0, 0};
    mutable bool gov_ok[2] = {false, false};
    mutable std::atomic<struct ScanStage *> stage{nullptr}; // != null: the upload + structure launch of ym_scan_create is not known to be complete yet
    size_t block_bytes = 0;                 // != 0: d_ranges is a block of this size of the device's scan pool (0: its own hipMalloc)
    int n;
    double min_angle, max_angle, angle_inc, min_range, max_range, range_threshold;
    double pose[3];
    double max_valid_karto, max_valid_yagpy;
    double beam_spacing; // median valid reading x angular resolution
    mutable std::atomic<uint32_t> query_uses{0}; // batches this scan has been a query of (a matcher caches a query's projection from its second use on)
    double lbox[4]; // sensor-frame bounding box (xmin, ymin, xmax, ymax) of every reading that can become a point
    double wbox[4]; // the box at the current pose, in the world: kept with the pose so that a call need not rotate 40 000 boxes
};

// ---- the scans' device memory and upload.
// ym_scan_create costs one kernel launch and no synchronisation: the readings are copied into a pinned staging slot,
// structure_kernel reads them from there (that IS the upload), writes them and the scan's chain structure into a block
// of the device's scan pool and finally its info words and a serial number into the slot.  Whoever first needs the scan
// (a matcher building a call, ym_scan_structure_trusted, ym_scan_destroy) waits for the serial number -- normally long
// there.  Blocks of destroyed scans are parked (hipFree would synchronise at every destroy); once kRecycleAt are parked they are
// SEALED: an event is recorded on every stream a kernel that reads scan blocks can run on (the matchers register theirs), and when all
// of a generation's events have completed its blocks serve new scans -- no device-wide synchronisation, so a node that creates and
// destroys thousands of scans per step (bench.py: cfg2x_fresh_scans) never stalls the lanes that are matching.  (Round 5 synchronised the
// device instead: every 64th destroyed scan's successor waited for everything in flight.)
// The pool keeps its memory for the life of the process (35 KB per 1081-beam scan ever alive or parked at the same time).
struct ScanStage {
    const ym_scan *owner = nullptr; // the scan whose launch last used the slot and has not been waited for
    uint32_t serial = 0;
    unsigned char *host = nullptr, *dev = nullptr; // [ranges: YM_MAX_BEAMS doubles][info int32[4]][done uint32[2]]
    std::atomic<int> readers{0};    // synchronous matches in flight that read the staged readings themselves (staged_query)
};
namespace {
constexpr int kScanStages = 64;
constexpr size_t kStageInfoOffset = sizeof(double) * YM_MAX_BEAMS;
constexpr size_t kStageBytes = kStageInfoOffset + 64;
constexpr size_t kRecycleAt = 64;
constexpr int kPoolStreams = 4;
constexpr size_t kSlabBytes = 4u << 20;

struct ScanPool {
    std::mutex mu;
    int device = -1;
    bool ready = false;
    hipStream_t streams[kPoolStreams] = {}; // creation launches go round them: structure_kernel is two blocks, several run side by side
    unsigned char *stage_host = nullptr;
    ScanStage stages[kScanStages];
    uint32_t next_stage = 0, serial = 0;
    std::unordered_map<size_t, std::vector<void *>> free_blocks; // by block size
    std::vector<std::pair<void *, size_t>> parked;               // of destroyed scans; a kernel in flight may still read them
    // ym_scans_create: staging buffers (pinned host + device), each [argument records][info words][readings] of one chunk of scans; a call
    // holds one per chunk in flight (two), several threads may create scans at once
    struct Bulk { unsigned char *host = nullptr, *dev = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool busy = false; };
    static constexpr int kBulkBuffers = 12;
    Bulk bulk[kBulkBuffers];
    hipStream_t bulk_streams[2] = {}; // high priority: a creation must not queue behind a lane's 3 ms correlate
    // recycling without a device-wide synchronisation (see above)
    struct Sealed { std::vector<hipEvent_t> events; std::vector<std::pair<void *, size_t>> blocks; };
    std::deque<Sealed> sealed;
    std::vector<hipStream_t> reader_streams; // the matchers' streams of this device (ym_create, ym_set_stream, the second stream)
    std::vector<hipEvent_t> event_pool;
};

ScanPool &scan_pool(int device) {
    static ScanPool pools[64];
    return pools[device & 63];
}

// (p.mu held, p's device current)
void stage_wait(ScanPool &p, ScanStage &st) {
    const ym_scan *s = st.owner;
    if (!s) return;
    const volatile uint32_t *done = reinterpret_cast<const volatile uint32_t *>(st.host + kStageInfoOffset + 16);
    bool seen = false;
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        for (int spin = 0; spin < 2048 && !seen; spin++) {
            seen = done[0] == st.serial && done[1] == st.serial;
            if (!seen) __builtin_ia32_pause();
        }
        if (seen) break;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 > 5.0) break;
    }
    if (!seen) { // slow or failed launch: ask the stream
        (void)hipStreamSynchronize(p.streams[(&st - p.stages) % kPoolStreams]);
        seen = done[0] == st.serial && done[1] == st.serial;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (seen) {
        const int32_t *info = reinterpret_cast<const int32_t *>(st.host + kStageInfoOffset);
        s->cnp[0] = info[0]; s->gov_ok[0] = info[1] == 0;
        s->cnp[1] = info[2]; s->gov_ok[1] = info[3] == 0;
    } else { // the launch never ran: upload the readings the plain way; the matchers compute the chain per pose
        (void)hipGetLastError();
        (void)hipMemcpy(s->d_ranges, st.host, sizeof(double) * s->n, hipMemcpyHostToDevice);
        s->gov_ok[0] = s->gov_ok[1] = false;
    }
    s->stage = nullptr;
    st.owner = nullptr;
}

// (p.mu held)  sealed generations whose events have all completed: their blocks are free
void pool_reap(ScanPool &p) {
    while (!p.sealed.empty()) {
        ScanPool::Sealed &g = p.sealed.front();
        for (hipEvent_t e : g.events) {
            const hipError_t q = hipEventQuery(e);
            if (q == hipErrorNotReady) return;
            if (q != hipSuccess) (void)hipGetLastError(); // (a stream that died: its work is over)
        }
        for (auto &b : g.blocks) p.free_blocks[b.second].push_back(b.first);
        for (hipEvent_t e : g.events) p.event_pool.push_back(e);
        p.sealed.pop_front();
    }
}
// (p.mu held, p's device current)  everything parked so far becomes a generation: free once every stream that may still read it has passed
void pool_seal(ScanPool &p) {
    ScanPool::Sealed g;
    bool ok = true;
    auto mark = [&](hipStream_t st) {
        hipEvent_t e = nullptr;
        if (!p.event_pool.empty()) { e = p.event_pool.back(); p.event_pool.pop_back(); }
        else if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ok = false; return; }
        if (hipEventRecord(e, st) != hipSuccess) { (void)hipGetLastError(); p.event_pool.push_back(e); ok = false; return; }
        g.events.push_back(e);
    };
    for (hipStream_t st : p.reader_streams) mark(st);
    for (hipStream_t st : p.streams) if (st) mark(st);
    for (hipStream_t st : p.bulk_streams) if (st) mark(st);
    mark(nullptr); // the null stream (occupancy rendering, debug copies)
    if (!ok && hipDeviceSynchronize() != hipSuccess) { // a stream the pool cannot mark: the blunt way -- and if even that fails, keep them parked
        (void)hipGetLastError();
        for (hipEvent_t e : g.events) p.event_pool.push_back(e);
        return;
    }
    g.blocks.swap(p.parked);
    p.sealed.push_back(std::move(g));
}

// (p.mu held, p's device current)  at least `count` free blocks of `bytes`: what is missing comes as ONE slab (a bulk creation that
// found the free list short asked hipMalloc for a 4 MB slab per 117 scans: 36 calls per 4096 scans, each a millisecond under load)
void pool_reserve(ScanPool &p, size_t bytes, size_t count) {
    std::vector<void *> &f = p.free_blocks[bytes];
    if (f.size() >= count) return;
    const size_t missing = std::max(count - f.size(), std::max<size_t>(1, kSlabBytes / bytes));
    unsigned char *slab = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&slab), missing * bytes) != hipSuccess) { (void)hipGetLastError(); return; } // (pool_block tries again, slab by slab)
    for (size_t i = missing; i-- > 0;) f.push_back(slab + i * bytes);
}

void pool_register_stream(int device, hipStream_t st, bool add) {
    if (!st) return;
    ScanPool &p = scan_pool(device);
    std::lock_guard<std::mutex> lk(p.mu);
    auto it = std::find(p.reader_streams.begin(), p.reader_streams.end(), st);
    if (add && it == p.reader_streams.end()) p.reader_streams.push_back(st);
    if (!add && it != p.reader_streams.end()) p.reader_streams.erase(it);
}

// (p.mu held, p's device current)  look = false: the caller has just looked for completed generations itself (a bulk creation asks once
// for all its blocks: an event query per block would cost more than the block)
void *pool_block(ScanPool &p, size_t bytes, bool look = true) {
    std::vector<void *> &f = p.free_blocks[bytes];
    if (f.empty() && look) {
        if (p.parked.size() >= kRecycleAt) pool_seal(p);
        pool_reap(p);
    }
    if (f.empty()) {
        const size_t count = std::max<size_t>(1, kSlabBytes / bytes);
        unsigned char *slab = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&slab), count * bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        for (size_t i = count; i-- > 0;) f.push_back(slab + i * bytes);
    }
    void *b = f.back();
    f.pop_back();
    return b;
}

// (p.mu held, p's device current)  block + staging slot + the one launch
int pool_init(ScanPool &p, int device) {
    if (p.ready) return YM_OK;
    for (int i = 0; i < kPoolStreams; i++) HIP_TRY(hipStreamCreateWithFlags(&p.streams[i], hipStreamNonBlocking));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&p.stage_host), kStageBytes * kScanStages, hipHostMallocMapped));
    unsigned char *dev = nullptr;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), p.stage_host, 0));
    std::memset(p.stage_host, 0, kStageBytes * kScanStages);
    for (int i = 0; i < kScanStages; i++) { p.stages[i].host = p.stage_host + kStageBytes * i; p.stages[i].dev = dev + kStageBytes * i; }
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ym::structure_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)YM_PREP_LDS_BYTES(YM_MAX_BEAMS));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(ym::structure_many_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)YM_PREP_LDS_BYTES(YM_MAX_BEAMS));
    p.device = device;
    p.ready = true;
    return YM_OK;
}

int pool_create_scan(ScanPool &p, ym_scan *s, const double *ranges, size_t total, unsigned char **base_out) {
    int rc0 = pool_init(p, s->device);
    if (rc0) return rc0;
    const size_t bytes = align_up(total, 1024);
    unsigned char *base = static_cast<unsigned char *>(pool_block(p, bytes));
    if (!base) return set_err(YM_ERR_HIP, "cannot allocate device ranges");
    // a staging slot no synchronous match of another thread is reading (staged_query; readers change under p.mu only upwards,
    // so a slot seen free here stays free): never WAIT for a reader with the mutex held -- its thread may need the mutex
    // (scan_resolve) before it lets go
    ScanStage *free_stage = nullptr;
    for (int tries = 0; tries < kScanStages && !free_stage; tries++) {
        ScanStage &c = p.stages[p.next_stage++ % kScanStages];
        if (c.readers.load(std::memory_order_acquire) == 0) free_stage = &c;
    }
    if (!free_stage) { // (64 matches in flight on freshly created scans: upload the plain way, the matchers compute the chain per pose)
        if (hipMemcpy(base, ranges, sizeof(double) * s->n, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            p.free_blocks[bytes].push_back(base);
            return set_err(YM_ERR_HIP, "cannot upload ranges");
        }
        s->d_ranges = reinterpret_cast<double *>(base);
        s->block_bytes = bytes;
        s->gov_ok[0] = s->gov_ok[1] = false;
        *base_out = base;
        return YM_OK;
    }
    ScanStage &st = *free_stage;
    stage_wait(p, st); // (the slot's previous user, 64 creations ago)
    std::memcpy(st.host, ranges, sizeof(double) * s->n);
    st.serial = ++p.serial ? p.serial : ++p.serial;
    s->d_ranges = reinterpret_cast<double *>(base);
    s->block_bytes = bytes;
    const size_t n1 = (size_t)s->n;
    const size_t ranges_bytes = align_up(sizeof(double) * n1, 16), gov_bytes = align_up(sizeof(int32_t) * 2 * n1, 16);
    const size_t cidx_bytes = align_up(sizeof(int32_t) * n1, 16);
    ym::StructureArgs sa;
    std::memset(&sa, 0, sizeof sa);
    sa.sr.ranges = reinterpret_cast<const double *>(st.dev); sa.sr.n = s->n; sa.sr.min_angle = s->min_angle; sa.sr.angle_inc = s->angle_inc;
    sa.sr.min_range = s->min_range; sa.sr.range_threshold = s->range_threshold;
    sa.gov[0] = reinterpret_cast<int32_t *>(base + ranges_bytes);
    sa.gov[1] = reinterpret_cast<int32_t *>(base + ranges_bytes + gov_bytes);
    sa.cidx[0] = reinterpret_cast<int32_t *>(base + ranges_bytes + 2 * gov_bytes);
    sa.cidx[1] = reinterpret_cast<int32_t *>(base + ranges_bytes + 2 * gov_bytes + cidx_bytes);
    sa.info = reinterpret_cast<int32_t *>(st.dev + kStageInfoOffset);
    sa.ranges_out = s->d_ranges;
    sa.done = reinterpret_cast<uint32_t *>(st.dev + kStageInfoOffset + 16);
    sa.serial = st.serial;
    hipLaunchKernelGGL(ym::structure_kernel<512>, dim3(2), dim3(512), YM_PREP_LDS_BYTES(s->n), p.streams[(&st - p.stages) % kPoolStreams], sa);
    if (hipGetLastError() != hipSuccess) { // the plain way
        if (hipMemcpy(s->d_ranges, ranges, sizeof(double) * s->n, hipMemcpyHostToDevice) != hipSuccess) {
            p.free_blocks[bytes].push_back(base);
            s->d_ranges = nullptr;
            return set_err(YM_ERR_HIP, "cannot upload ranges");
        }
    } else {
        st.owner = s;
        s->stage = &st;
    }
    *base_out = base;
    return YM_OK;
}

// the scan's creation launch has completed and its info words are in the ym_scan
inline void scan_resolve(const ym_scan *s) {
    if (!s->stage) return;
    ScanPool &p = scan_pool(s->device);
    std::lock_guard<std::mutex> lk(p.mu);
    if (!s->stage) return;
    DevGuard guard(s->device);
    stage_wait(p, *s->stage);
}
}  // namespace
