// yagmatch.hip -- host runtime + C ABI of libyagmatch.so (see include/yagmatch.h).
//
// One ym_matcher = one HIP stream + one device workspace.  A call (B items) is: one H2D copy of
// the call descriptor, a fixed sequence of kernel launches (ym_kernels.hpp), one D2H copy of the
// per-item result states.  Everything between stays in HBM.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <ctime>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/yagmatch.h"
#include "ym_kernels.hpp"

struct ScanStage; // (the pinned staging slot of a scan, see the scan pool)

namespace {

thread_local std::string g_err;

std::atomic<uint64_t> g_pose_epoch{1}; // bumped by every ym_scan_set_pose: "no scan moved since" is one comparison

int set_err(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return set_err(YM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                           __FILE__, __LINE__);                                                \
    } while (0)

double kt_round_h(double v) { return v >= 0.0 ? std::floor(v + 0.5) : std::ceil(v - 0.5); }
bool kt_double_equal_h(double a, double b) {
    double d = a - b;
    return d < 0.0 ? d >= -YM_KT_TOLERANCE : d <= YM_KT_TOLERANCE;
}
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr int kAsyncSlots = 64;

// Makes `device` current for the lifetime of the guard and puts the caller's device back afterwards (the caller's
// thread may be torch code with another current device).
struct DevGuard {
    int prev = -1, dev;
    bool ok = true;
    explicit DevGuard(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DevGuard() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
    DevGuard(const DevGuard &) = delete;
    DevGuard &operator=(const DevGuard &) = delete;
};
#define DEV_GUARD(d)                                                                     \
    DevGuard dev_guard_(d);                                                              \
    if (!dev_guard_.ok) return set_err(YM_ERR_HIP, "cannot make device %d current", (d))

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0; // elements
    int ensure(size_t n) {
        if (n <= cap) return YM_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 64;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T)));
        cap = want;
        return YM_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct PinnedBuf {
    unsigned char *p = nullptr;   // host address
    unsigned char *dp = nullptr;  // the same memory as the device sees it
    size_t cap = 0;
    int ensure(size_t n) {
        if (n <= cap) return YM_OK;
        if (p) (void)hipHostFree(p);
        p = dp = nullptr;
        cap = 0;
        size_t want = align_up(n + n / 4 + 256, 256);
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&p), want, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&dp), p, 0));
        cap = want;
        return YM_OK;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = dp = nullptr;
        cap = 0;
    }
};

// a scan as a call sees it: device ranges + metadata + pose
struct CallScan {
    const double *d_ranges;
    int n;
    double min_angle, angle_inc, min_range, range_threshold;
    double pose[3];
    double max_valid; // largest reading that survives range gating (bounds the query's reach)
    double lbox[4];   // sensor-frame bounding box of the points (bounds where a base scan can stamp)
    double wbox[4];   // the same box at the scan's pose, in the world (xmin, ymin, xmax, ymax; empty: xmin > xmax)
    double beam_spacing = 0; // median valid reading x angular resolution: how far apart neighbouring end points are
    uint64_t id = 0;  // resident scan identity (0: ranges uploaded for this call only, never cached)
    int cache_hint = -1;            // entry of the matcher's point cache this scan used last time (ym_batch remembers it)
    unsigned char *cache = nullptr; // this call's cache slot (device), or null
    int stale = 0;                  // the slot must be (re)computed by this call
    ScanStage *staged = nullptr;    // the call reads the readings from the scan's staging slot (see staged_query)
    int qcache_hint = -1;           // the same three for the scan as the QUERY of a batch
    unsigned char *qcache = nullptr;
    int qstale = 0;
    uint32_t query_uses = 1;          // batches: how often the scan has been the query of a batch before this call (0: never -- see plan_cache)
    const int32_t *gov = nullptr;     // the scan's pose-independent chain structure (trusted scans) ...
    const int32_t *cidx = nullptr;    // ... its compaction ...
    int cnp = 0;                      // ... and its number of point readings
    bool direct = false;              // this call uses them (plan_cache): the scan needs no slot in the point cache
    const double *pose_dev = nullptr; // device-chained sequence: where the device finds the pose the host only predicts
};

struct CallItem {
    int query;
    int base_begin, base_count;
    int qslot = 0; // batches: query slot (distinct queries of a call are projected once)
};

struct Call {
    std::vector<CallScan> scans;
    std::vector<CallItem> items;
    int penalize = 1, refine = 1;
    double coarse_angle_off = 0; // response expansion widens this
    int expansions = 0;
    // one match split over several matchers by coarse angle (ym_match_slice_*): this matcher scores angles [k_begin,
    // k_end) into caller-owned device buffers and stops after the score stage
    int k_begin = 0, k_end = -1;
    double *ext_resp = nullptr, *ext_probs = nullptr;
    bool slice = false;
    // a step of a device-chained sequence (ym_map_sequence): the poses of the scans whose matches are still in flight come
    // from the device (CallScan::pose_dev; the host's are dead-reckoned predictions that only size the raster), the
    // result state lands in chain_out, and final_kernel leaves this step's pose and the next step's prior on the device
    // a resident batch enqueued again (ym_batch_run_async): the batch this Call was built from, the pose epoch it was built
    // in (no ym_scan_set_pose since: every field is still right), and the point-cache generation its cache / stale fields
    // were planned in without any slot left to fill -- while all three hold the host plans nothing per scan
    uint64_t batch_uid = 0, pose_epoch = 0, plan_gen = 0;
    bool plan_clean = false;
    std::vector<int32_t> plan_jobs, plan_job_slot, plan_qrep; // what plan_jobs produced for that plan ...
    int plan_want[4] = {0, 0, -1, -1};                        // ... and the tile rectangle plan_raster found the chains' boxes in,
    int plan_want_geom[3] = {0, 0, 0};                        // ... for this window (origin, width) and tile height
    bool plan_want_valid = false;
    int chain_step = 0;               // 0: an ordinary call
    double *chain_pose_out = nullptr; // DEVICE: this step's row of the segment's pose table
    double chain_next_diff[3] = {0, 0, 0};
    YmItemState *chain_out = nullptr; // DEVICE view of the pinned state this step's result goes to
};

// Everything one call's launches share: sizes, lattices, the device window, how the coarse correlate is cut up,
// strides, the descriptor, and which tiles the raster covers.  Filled in by the plan_* functions below.
struct CallPlan {
    int B = 0, nscans = 0, max_n = 1, max_base = 1;
    int tile_h = YM_TILE_H;            // rows per raster tile in this call
    bool lists_cached = false;         // the matcher's list buffers already hold this call's pair lists (ym_matcher::list_key)
    bool lists_on_side_stream = false; // the region path's bin_kernel went to the matcher's second stream (join before the region kernel)
    bool yag = false;
    YmGeom g;
    YmLattice lc, lf;
    // device window
    int tiles_x = 0, tiles_y = 0;
    size_t grid_stride = 0;
    // coarse correlate decomposition
    int sx = 2, ngx = 0, nx_pad = 0, njobs = 0, tpb = 1, job_blocks = 0, ktiles = 0, n_chunks = 1, chunk = 0, corr_u = 16;
    int dedup = 0;            // merge consecutive beams with equal lookup offsets (coarse grids)
    int cw = 1, n_groups = 1; // chunk-waves per correlate block, chunk groups (= partial sums per hypothesis)
    // batches on the default-sized lattices (up to 26 x 32): the region-staged correlate (ym_k_region.hpp)
    bool region26 = false;
    bool fuse_score = false;  // ... also scores (no score_kernel launch)
    int rg_nrx = 0, rg_nry = 0, rg_ng = 1, rg_nbins = 0, rg_nw = 7, rg_parts = 1, rg_nregions = 0;
    int rg_rsplit = 1;  // blocks that share the regions of an (item, angle block) on small batches
    bool rg_item = false; // correlate_item_kernel: one block of 16 waves per item, the item's sums in LDS
    bool rg_pool = false; // correlate_pool_kernel: two blocks of 12 waves per item, a region's patches dealt evenly, 16-bit sums in LDS
    bool win_only = false; // the region correlate stages from the row-major window and the raster does not write the planes
    bool rg_ws = false; // the wave-specialised region correlate (gather waves + loader waves, regions of YM_WS_H rows)
    bool rg2 = false;   // correlate_region2_kernel (round 5): sixteen waves per block, several waves per angle, regions rg2_h rows high
    int rg2_h = 0;
    size_t rg_entries_stride = 0, rg_starts_stride = 0, rg_entries_pstride = 0;
    int rg_lnw = 0, rg_lparts = 1; // the pair lists are built per block of rg_lnw angles (ym_k_region.hpp, bin_kernel); the experimental forms: one part of all
    // batches on other lattices up to 48 x 64, or with merged offsets: the LDS gather correlate (ym_k_gather.hpp), which
    // always scores its sums
    bool region = false;
    int ga_W = 0, ga_H = 0, ga_P = 0, ga_rows = 0, ga_nrx = 0, ga_nry = 0, ga_nseg = 1, ga_np = 1, ga_ng = 1, ga_parts = 1, ga_kpp = 1;
    int ga_na = 1, ga_nwv = 1, ga_cap = 512, ga_nbins2 = 0, n_qslots = 1;
    size_t ga_units_stride = 0, ga_starts_stride = 0, ga_work_stride = 0, ga_lds = 0;
    std::vector<int32_t> qrep; // an item of every query slot; travels at the end of the call descriptor
    const int32_t *d_qrep = nullptr;
    // yagpy lattice bounds
    int ymaxd = 0, ymaxt = 0;
    size_t yvol = 0;
    // strides
    int nt_stride = 0, dim_stride = 0, score_blocks = 0, cell_blocks = 0;
    size_t sums_c = 0, sums_f = 0, partial_stride = 0;
    // call descriptor
    size_t scans_bytes = 0, desc_bytes = 0;
    bool inline_desc = false;
    YmScanRef *hs = nullptr;
    YmItem *hi = nullptr;
    const YmScanRef *d_scans = nullptr;
    const YmItem *d_items = nullptr;
    // raster coverage
    int launch[4] = {0, 0, -1, -1}, ltx = 0, lty = 0, tile_cap = 1;
    int chain_step = 0;
    int cell_box[4] = {INT32_MIN, INT32_MIN, INT32_MAX, INT32_MAX}; // chained steps: window cells whose smear stays inside the launched tiles
    bool use_tile_list = false;
    bool use_tile_hits = false;
    unsigned long long *stamps = nullptr;
    // batches: heavy work once per distinct scan (points_kernel), then the light cells_kernel
    bool split_prepare = false;
    int n_jobs = 0;
    std::vector<int32_t> jobs, job_slot; // travel at the end of the call descriptor
    double *resp = nullptr, *probs = nullptr; // the matcher's buffers, or the caller's on an angle-sliced match
    int k_begin = 0, k_end = 0;
    const int32_t *d_jobs = nullptr, *d_job_slot = nullptr;
};

struct Slot {
    PinnedBuf desc;    // YmScanRef[] + YmItem[] staged for the H2D copy
    DevBuf<unsigned char> desc_dev;         // the slot's descriptor on the device ...
    size_t desc_live_bytes = 0;             // != 0: the pinned buffer AND the device copy hold the slot's last descriptor, of this size
    PinnedBuf result;  // YmItemState[] landed by the D2H copy
    hipEvent_t done = nullptr;
    bool in_flight = false;
    Call call;         // kept for response-expansion re-runs and result assembly
    YmLattice coarse{}, fine{};
    int n_items = 0;
    int64_t chain_id_base = 0;
    CallPlan plan;     // angle-sliced match: kept between ym_match_slice_begin and _finish
    void *dev_best_out = nullptr; // optional device buffer (8 doubles) for the cross-rank arg-max
    void *dev_best_user = nullptr; // the same pointer, kept until the slot is collected (rewritten after a response expansion)
    uint32_t poll_serial = 0;      // != 0: final_kernel writes this number into the word after the result states when they are complete
    uint32_t serial_counter = 0;
};

struct ProfEvents {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;
    size_t used = 0;
    double ms = 0;
    int64_t launches = 0;
};

}  // namespace

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

struct ym_map {
    int device;
    int width, height;
    double *d_cgrid;  // the float correlation grid as the reference holds it
    uint8_t *d_g8;    // int(100 * cell): what scoring reads
};

struct ym_occupancy {
    int device;
    ym_occupancy_info info;
    std::vector<uint8_t> image; // [height][width], row 0 = lowest y
};

struct ym_batch {
    std::vector<const ym_scan *> queries; // the distinct query scans (one for ym_batch_create; ym_pairs_create: up to one per item)
    std::vector<int32_t> item_query;      // per item: its query's index in `queries`
    std::vector<const ym_scan *> scans;
    std::vector<int32_t> offsets;
    mutable std::vector<int> cache_hints; // per scan: its entry in the owning matcher's point cache (validated on use)
    mutable std::vector<int> query_hints; // the same per query
    uint64_t uid = 0; // unique per created batch
};

struct ym_matcher {
    ym_config cfg;
    int device;
    hipStream_t own_stream, stream;
    // the region path's pair lists (bin_kernel: ONE block per query of the call, ~85 us) need nothing of the raster: they are
    // built on a second stream next to tiles + raster and joined before the region kernel
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlap_lists = true;
    bool staged_queries = true;       // a synchronous match reads a just-created query scan from its staging slot instead of waiting
    int tile_h_forced = 0;            // tests: 32 or 64 rows per raster tile whatever the call
    int sticky_tall_left = 0;         // small calls that still take the tall tiles of the last large batch (plan_sizes)
    int tall_tiles_min_window = 768;  // window width (cells) from which a batch of 512+ items gets 64-row tiles (256 items: 108 against 111 us of raster)
    YmGeom geom;                 // config part filled at create; window part per call
    std::vector<uint8_t> kernel; // Karto smear kernel (ksize x ksize)
    std::vector<double> kernel_f; // yagpy: the float kernel (helpers.py:86-97), for maps built from occupancy images
    DevBuf<double> kernel_f_dev;
    DevBuf<double2> map_pts;      // match against a map: the query point set
    int z2max = 0;               // largest squared cell distance whose kernel value is 100
    DevBuf<uint8_t> ktab;
    DevBuf<uint8_t> rowtab;   // the raster's row-pass tables (upload_lut)
    int n_rowtab = 0, rowtab_shift = -1;
    // workspace
    DevBuf<unsigned char> desc_dev; // batch call descriptors (single calls travel in the kernel arguments)
    DevBuf<YmItemState> states;
    DevBuf<double2> qlocal;    // [query slots][max_n] sensor-frame query points
    DevBuf<int32_t> qnp;       // [query slots]
    DevBuf<unsigned char> tmp_cache; // batches: per-call cache slots of base scans the point cache cannot hold
    DevBuf<int2> cells;
    DevBuf<int4> bbox;
    DevBuf<uint8_t> grid;
    DevBuf<uint8_t> planes;    // even/odd column planes of every window
    DevBuf<uint8_t> tile_zero; // per raster tile: window memory known to be zero (skips rewriting empty tiles)
    DevBuf<uint8_t> sub_zero;  // per raster tile that is not: which of its 8 x 8 sub-blocks are (8 bytes per tile)
    size_t tz_sig[6] = {0, 0, 0, 0, 0, 0}; // memory/geometry the flags are valid for
    // per workspace item: tile rectangle (x0, y0, x1, y1) outside which the item's window memory is known to be zero.
    // Items [0, tz_covered) have valid flags and rectangles; a call only rasterises (and cleans) items [0, B), so the
    // state of the items past B must survive it.
    std::vector<std::array<int, 4>> item_dirty;
    int tz_covered = 0;
    // Calls whose correlate stages from the row-major window write the window only (CallPlan::win_only); the column planes of the
    // items they touch then lag behind.  The knowledge above describes WINDOW memory and stays valid through such calls; what a later
    // call that reads the planes needs is every tile of ITS items written once more -- item by item, not the whole matcher (round 4
    // kept the mode in the signature: one single match between two batches of 4096 cost the second a full raster of all 4096 windows)
    std::vector<unsigned char> planes_stale;
    DevBuf<double2> ctrig;     // (cos, sin) per coarse angle
    DevBuf<int32_t> foffsets;  // fine lookup tables
    DevBuf<int32_t> hypcell;
    DevBuf<uint16_t> partial;  // per beam-chunk partial sums of the coarse lattice
    DevBuf<uint16_t> rg_entries; // region correlate: per query slot of a call the (beam, angle) pairs sorted by region
    DevBuf<int32_t> rg_starts;
    DevBuf<uint32_t> rg_rbox;    // per query slot, region and angle block: the box its patches read of the region
    DevBuf<uint32_t> rg_walk;    // per query slot and angle block: the walk of the wave-specialised region correlate (region_walk_kernel)
    int n_cus = 0;               // compute units of the device
    size_t bin_lds_limit = 64 * 1024; // dynamic LDS bin_whole_kernel may use so far (experimental builds)
    size_t binp_lds_limit = 64 * 1024; // ... and bin_kernel
    int chain_margin = 1;        // tiles (64 cells) added around the predicted raster rectangle of a chained step
    uint64_t cache_gen = 1;      // bumped whenever the point cache changes (entries created, re-posed, dropped) or an option is set
    int64_t seq_segments = 0, seq_faults = 0, seq_sync_steps = 0; // ym_map_sequence: chained segments, those cut short, synchronous steps
    // Batches below this size take the direct correlate kernel: a block of either LDS correlate walks all regions of its item,
    // ~190 us whatever the batch, while the direct kernel's time grows with the batch from ~15 us (measured, both lattices:
    // 8 chains 142 / 109 us against 220 / 225 per enqueue, 64 chains equal, 256 chains 675 / 759 against 485 / 452)
    int lds_min_batch = 64;
    int rg_min_batch = 48;  // the region correlate from this many items on (round 4: 48 items 180 -> 169 us, 56 items 196 -> 184; below 44 the direct kernel wins)
    int prepare_threads = 0;     // development: 512 = the single-item prepare kernel with 512 threads per scan too
    int last_wh = 0;             // half width of the previous call's device window (cells, before clamping)
    bool use_scan_structure = true; // base scans' trigger chains come from ym_scan_create's structure_kernel where that is exact
    bool poll_completion = true; // single matches: the host polls a pinned word instead of waiting for the stream event
    int corr_region_nw = 0;  // development: waves (= angles) per region-correlate block
    int item_min_batch = 1 << 30; // batches from this many items on take correlate_item_kernel
    bool item_lds_set = false;
    int raster_planes_only = 0; // timing experiment (option 36): the raster does not write the row-major window
    int raster_no_rowtab = 0;   // tests (option 37): the raster's row pass by bit scans instead of its tables
    int corr_region_rsplit = 0; // 0 = by batch size, 1 = never split an item's regions over blocks, n = always n blocks
    int tile_list_min_batch = 48; // option 40: batches from this size on get raster work lists (tiles_kernel)
    int keep_planes = 0;         // option 39: 1 = every call writes the column planes and the region correlate stages from them
    int corr_region_pad_lds = 0; // development (option 38): dynamic LDS bytes the region correlate is launched with and does not use (fewer blocks per CU)
    int corr_region_dbg = 0;  // development (timing only): 1 = the loader waves move nothing, 2 = the gather waves gather nothing
    int corr_region_form = 0; // 2 = the wave-specialised region correlate (gather waves + loader waves) instead of correlate_region_kernel
    // the pair lists of the last single-query call that built them: what they were built from.  A call with the same key finds them
    // in the list buffers and builds nothing (no bin_kernel, no second stream, no join) -- they depend on the query's readings and
    // pose, the window and the lattice alone, like the projected points the point cache keeps (round 5; option 45 = 0: off)
    struct ListKey {
        uint64_t qid; double pose[3]; YmGeom g; YmLattice lc;
        int32_t nw, parts, nrx, nry, rg_h, force, nregions, ng; size_t es, ss; const void *pe, *ps, *pb;
    };
    ListKey list_key;
    bool list_key_valid = false, list_cache_on = true;
    int64_t list_cache_hits = 0;
    int rg2_min_batch = 1 << 30; // batches from this many items on take correlate_region2_kernel (option 32 = 5: always where it can)
    int rg2_h = 128;          // option 43: class rows a region of correlate_region2_kernel owns (80, 100 or 128)
    size_t rg2_lds_limit = 0;
    int corr_fuse_score = 0; // tests: 2 = the region correlate never scores itself (score_kernel does)
    // gather correlate: per query slot of a call the (beam, angle) units sorted by region, the bin table, the work
    // lists and the counters they are built with; the lane -> (row, segment) table of the lattice
    DevBuf<uint32_t> ga_units;
    DevBuf<int32_t> ga_starts;
    DevBuf<int32_t> ga_work;
    DevBuf<uint32_t> ga_counters;
    DevBuf<uint32_t> ga_lane_job;
    std::vector<uint32_t> ga_lane_job_host; // what ga_lane_job holds
    size_t ga_lds_limit = 64 * 1024;        // dynamic LDS gather_kernel may use so far
    DevBuf<uint32_t> sums;     // coarse sums, then fine sums
    DevBuf<double> resp;
    DevBuf<double> blockmax;
    DevBuf<double> probs;
    // yagpy: the coarse pass's integer sums come from the production correlate kernels where the item's roundings provably form a
    // lattice (ym_k_yagpy.hpp, yag_lattice_kernel); option 46 = 0: every item through yag_score_kernel, the rule as written
    int yag_fast = 1;
    int last_corr_form = -1; // which coarse correlate the last call launched: 0 correlate_kernel, 1 correlate_region_kernel, 2 gather_kernel, -1 none
    DevBuf<unsigned long long> yag_counters; // [0] items through the production kernels, [1] fallbacks, [2] pairs checked exhaustively, [3] pairs that failed
    DevBuf<double> yaxes;      // yagpy: xvals, yvals, tvals per item
    DevBuf<double2> yrot;      // yagpy: points rotated per angle
    DevBuf<double> seq_pose;   // device-chained sequences: [0..2] the next step's odometry prior, [4 + 3k ..] the pose step k of the segment found
    DevBuf<int32_t> seq_fault; // ... and the first step the host has to repeat (0: none)
    PinnedBuf seq_results;     // ... and the result state of every step of a segment
    DevBuf<unsigned long long> stamps; // phase time stamps (development aid)
    bool stamps_on = false;
    int corr_u = 0;      // development: force the number of beams in flight per lane (16, 32, 48)
    int full_raster = 0; // development: launch every raster tile
    // point cache: world point readings + trigger chain of resident base scans, per (scan id, pose) -- what Karto's
    // LocalizedRangeScan keeps in m_PointReadings until the pose is set again.  One arena, bump-allocated; everything
    // that touches it runs on this matcher's stream, so recomputing a slot in place is ordered after its readers.
    struct CacheEntry { uint64_t id; size_t off; int n; double pose[3]; uint64_t stale_in_call; };
    uint64_t call_counter = 0;
    std::vector<CacheEntry> cache_entries;
    std::unordered_map<uint64_t, int> cache_index;
    DevBuf<unsigned char> cache_arena;
    size_t cache_used = 0;
    size_t cache_limit = (size_t)16 << 30; // bytes; beyond it the cache starts over
    int cache_off = 0;                     // development: 1 = never cache (every call projects every scan)
    int64_t cache_hits = 0, cache_misses = 0;
    DevBuf<unsigned> sel_scratch; // select on long chains: hash, states and neighbour lists in global memory
    DevBuf<unsigned> sel_tables;  // select on a few items (split form): hash keys and earliest-point table, zero between calls
    DevBuf<uint4> sel_rec;        // ... and the record per point
    DevBuf<unsigned> sel_slot;    // ... and the point's slot (between the hash and the neighbour launch)
    DevBuf<uint32_t> tile_list; // raster work list per item
    DevBuf<int32_t> tile_count;
    DevBuf<int32_t> tile_max;        // [1] longest raster work list of the call
    DevBuf<uint16_t> tile_hits;      // per entry of the work list: the chunks that reach its tile (YM_TILE_HITS slots)
    int32_t *tile_max_host = nullptr; // pinned: the raster kernel leaves that number here, the next call sizes its grid by it
    int finish_form = 0; // development: 1 = fine_kernel + final_kernel even on batches, 2 = finish_kernel always
    int corr_chunks = 0; // development: force the number of beam chunks of the correlate kernel
    int corr_pad_lds = 0; // development: extra dynamic LDS per correlate block (limits blocks per CU)
    int corr_cw = 0;      // development: force the chunk-waves per correlate block (1, 2, 4)
    int corr_dedup = 0;     // development / tests: 1 = always merge equal consecutive lookup offsets, 2 = never
    int corr_region = 0;    // tests: 1 = neither LDS correlate, 2 = their per-cell path, 3 = their "lists do not fit" path, 4 = the gather
                            // correlate also where the region correlate would run
    int corr_region_na = 0; // development: jobs (angle, lattice part) per wave of the gather correlate (1..4)
    int corr_region_parts = 0; // development / tests: blocks per item of the gather correlate (each takes a share of the angles)
    int corr_region_cap = 0;   // tests: units per LDS buffer (a multiple of 64; small values force chunked regions)
    int corr_region_lds = 0;   // development / tests: LDS bytes a gather block may use (small values force many regions)
    int raster_gx = 0;      // tests: raster blocks per item (0 = by the previous call's longest work list)
    int raster_hits_per_tile = 0; // tests: hit slots per entry of the raster's work list (0 = YM_TILE_HITS, -1 = no hit lists)
    int keep_sums = 0;      // development: keep the coarse integer sums of batches too (ym_debug_sums)
    int finish_threads = 0; // development: force the finish kernel's block size (256 / 1024)
    int select_global = 0; // development / tests: always evaluate the order-dependent smear rule with the global-memory kernel
    int select_split_max = 8; // items up to which the rule runs in its split form (tests: 0 = the one-block kernel always)
    DevBuf<double> tmp_ranges;   // device copy of ranges for the descriptor-based entry
    PinnedBuf tmp_ranges_host;
    Slot slots[kAsyncSlots + 1]; // last one serves the synchronous entry points
    // geometry of the last launched call (debug getters)
    YmGeom last_geom;
    YmLattice last_lat[2];
    int last_B = 0, last_max_n = 0, last_max_base = 0, last_nt_stride = 0, last_dim_stride = 0;
    size_t last_grid_stride = 0, last_sums_stride[2] = {0, 0};
    size_t sums_pass_offset[2] = {0, 0};
    bool last_valid = false;
    // profiling
    bool profiling = false;
    ProfEvents prof[3];
};

namespace {

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
        bool tall = B >= 512 && g.win_w >= m->tall_tiles_min_window;
        if (tall) m->sticky_tall_left = 64;
        else if (m->sticky_tall_left > 0 && g.win_w >= m->tall_tiles_min_window && !call.chain_step) { tall = true; m->sticky_tall_left--; }
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
        const int rg_h = P.rg_ws ? YM_WS_H : P.rg2 ? P.rg2_h : YM_RG_H;
        P.rg_nrx = (half_w + YM_RG_W - 1) / YM_RG_W;
        P.rg_nry = (half_w + rg_h - 1) / rg_h;
        P.rg_nregions = P.rg_nrx * P.rg_nry;
        // sets of 16-bit sums per (item, angle): room for the padding of the entry lists (an item that needs more is scored
        // by the per-cell path)
        P.rg_ng = ((max_n * 23 + 19) / 20 + YM_RG_FLUSH - 1) / YM_RG_FLUSH;
        P.rg_nbins = P.rg_nregions * lc.nt;
        P.region26 = !wrap && (!yag || lc.nx > 0) && !P.dedup && !call.slice && P.sx == 2 && B >= m->rg_min_batch && m->corr_region != 1 && m->corr_region != 4 && lc.nx <= 2 * YM_RG_G &&
                     lc.ny <= 32 && P.rg_ng <= 8 && (int64_t)lc.nt * max_n <= YM_RG_MAX_ENTRIES && P.rg_nbins < YM_RG_MAX_BINS && max_n < 2048;
        if (P.region26) {
            // fewer blocks than three per CU: deal every (item, angle block)'s regions out to several blocks (64 chains: the
            // kernel 121 -> 65 us with four, the enqueue 236 -> 205 us; 128 chains 317 -> 295 with two; scripts/dev/rsplit_time.py)
            P.rg_rsplit = 1;
            if (!P.rg_ws && !P.rg2 && m->corr_region_rsplit != 1 && m->corr_region_form != 3 && m->corr_region_form != 4) {
                const int blocks = B * P.rg_parts;
                P.rg_rsplit = m->corr_region_rsplit > 1 ? m->corr_region_rsplit : std::max(1, std::min(8, (3 * m->n_cus) / std::max(1, blocks)));
            }
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
    P.partial_stride = P.region26 ? (size_t)P.rg_ng * P.rg_rsplit * (P.rg2 ? YM_R2_MAX_WPA : 1) * lc.nt * 64 * 16 : P.region ? (size_t)P.ga_ng * lc.nt * P.ga_np * 64 * 16 : (size_t)P.n_groups * lc.nt * lc.ny * P.nx_pad;
    P.cell_blocks = (lc.nx * lc.ny + YM_SCORE_THREADS - 1) / YM_SCORE_THREADS;
    P.score_blocks = P.cell_blocks * lc.nt; // block maxima per (angle, block of cells)

    int rc;
    if ((rc = m->states.ensure(B))) return rc;
    if ((rc = m->qlocal.ensure((size_t)B * max_n))) return rc;
    if ((rc = m->qnp.ensure(B))) return rc;
    if ((rc = m->cells.ensure((size_t)B * max_base * max_n))) return rc;
    if ((rc = m->bbox.ensure((size_t)B * max_base * YM_N_BOXES(max_n)))) return rc;
    if ((rc = m->grid.ensure((size_t)B * P.grid_stride + std::max(YM_RG_WINDOW_SLACK(g.pitch), YM_R2_WINDOW_SLACK(g.pitch, 128))))) return rc;
    if ((rc = m->planes.ensure((size_t)B * P.grid_stride + std::max(std::max(YM_RG_PLANES_SLACK(g.pitch / 2), YM_WS_PLANES_SLACK(g.pitch / 2)), YM_GA_PLANES_SLACK(g.pitch / 2, std::max(P.ga_rows, P.ga_nry + P.ga_H), lc.ny, P.ga_P))))) return rc;
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
        if ((rc = m->yrot.ensure((size_t)B * P.ymaxt * max_n))) return rc;
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
            if (P.rg_ws && (rc = m->rg_walk.ensure((size_t)n_q * P.rg_parts * YM_WS_WALK_WORDS))) return rc;
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
        a.qlocal = m->qlocal.p; a.axes = m->yaxes.p; a.rot = m->yrot.p;
        a.sums = m->sums.p + m->sums_pass_offset[pass]; a.out = m->resp.p;
        a.grid = m->grid.p; a.grid_stride = P.grid_stride; a.vol_stride = P.yvol;
        a.max_n = P.max_n; a.maxd = P.ymaxd; a.maxt = P.ymaxt;
        hipLaunchKernelGGL(ym::yag_setup_kernel, dim3(P.ymaxt, P.B), dim3(256), 0, m->stream, a);
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
        hipLaunchKernelGGL(ym::yag_score_kernel, dim3((P.ymaxd * P.ymaxd + 255) / 256, P.ymaxt, P.B), dim3(256), 0, m->stream, a);
        hipLaunchKernelGGL(ym::yag_reduce_kernel, dim3(P.B), dim3(1024), 0, m->stream, a);
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
    r.rg_h = P.rg_ws ? YM_WS_H : YM_RG_H; r.rg_cls = P.rg_ws ? YM_WS_CLS : P.rg_item ? YM_IT_CLS : YM_RG_CLS;
    r.rg_zero = P.rg_ws ? YM_WS_ZERO : P.rg_item ? YM_IT_ZERO : YM_RG_ZERO; r.pad2 = 0;
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

}  // namespace

// =================================================================== C ABI
extern "C" {

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
        hipLaunchKernelGGL(ym::yag_reduce_kernel, dim3(1), dim3(1024), 0, st, a);
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
    else if (option == 46) m->yag_fast = value != 0;
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

}  // extern "C"
