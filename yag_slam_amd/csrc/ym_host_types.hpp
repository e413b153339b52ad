// ym_host_types.hpp -- host runtime: error text, device guard, device / pinned buffers, a call as the host sees it (CallScan, Call, CallPlan, Slot)
// Part of yagmatch.hip (included at file scope); not a header of its own.
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
