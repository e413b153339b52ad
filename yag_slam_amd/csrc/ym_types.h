// ym_types.h -- plain structs shared by host code and gfx950 kernels.
#pragma once
#include <stdint.h>

#define YM_OCCUPIED 100
#define YM_KT_TOLERANCE 1e-06
#define YM_MAX_VARIANCE 500.0
#define YM_PENALTY_GAIN 0.2
#define YM_KT_PI 3.14159265358979323846
#define YM_KT_2PI 6.28318530717958647692
#define YM_CELL_NONE INT32_MIN
#define YM_MAX_BEAMS 6000          // per scan; bounds the LDS staging of one scan (25 B per reading)
#define YM_MAX_COARSE_NT 256         // coarse angles per pass (incl. response expansion)
#define YM_MAX_FINE_NT 256           // fine angles
#define YM_YAG_MAX_DIM 512            // yagpy lattice points per axis (np.arange lengths)
#define YM_YAG_MAX_NT 256
#define YM_MAX_KERNEL_HALF 20      // sigma <= 10*res  ->  half = Round(2*sigma/res) <= 20

// Grid geometry of one matcher configuration + the device window chosen for one call.
// Karto: ScanMatcher::Create / CorrelationGrid::CreateGrid.  The device never allocates Karto's
// full (search + 2*range_threshold)/res square: it keeps the central window that the query's
// endpoints can reach (all other cells are provably never read).
struct YmGeom {
    double scale;        // 1 / resolution      (CoordinateConverter::m_Scale)
    double res;          // 1 / scale           (Grid::GetResolution())
    int32_t side;        // search-space side   Round(S/res) + 1
    int32_t roi_w;       // ROI width = height  side + 2*ceil(rt/res)
    int32_t border;      // half_kernel + 1     (ROI origin inside the storage)
    int32_t storage_w;   // roi_w + 2*border
    int32_t half_kernel; // Round(2*sigma/res)
    int32_t zone_count;  // kernel taps equal to 100 (1 = only the centre)
    // device window, in Karto storage coordinates
    int32_t win_origin;  // storage coordinate of window cell (0,0), same for x and y
    int32_t win_w;       // window width = height in cells
    int32_t pitch;       // bytes per window row (multiple of 64, >= win_w + 64)
    int32_t semantics;
    int32_t kpitch;      // 0, or Karto's own row pitch (storage width rounded up to 8): the call answers through GetResponse's
    int32_t pad;         // linear-index test over Karto's WHOLE storage (a query reading beyond the matcher's range threshold)
    // penalties (Karto CorrelateScan)
    double dist_var, ang_var, min_dist_pen, min_ang_pen;
};

// one search lattice (coarse or fine pass)
struct YmLattice {
    int32_t nx, ny, nt;
    int32_t fine;      // 0 coarse, 1 fine
    int32_t penalize;
    int32_t pad;
    double off_x, off_y;     // half extents (metres)
    double step_x, step_y;   // lattice step (metres)
    double angle_off, angle_res;
};

// device-visible scan descriptor (one per scan referenced by a call)
struct YmScanRef {
    const double *ranges; // DEVICE pointer
    int32_t n;
    int32_t stale;        // 1: the cache slot (if any) must be (re)computed by this call
    double min_angle, angle_inc, min_range, range_threshold;
    double pose[3];
    unsigned char *cache; // DEVICE pointer to this scan's slot in the matcher's point cache, or null (see YM_CACHE_*)
    unsigned char *qcache; // the same for the scan in the role of a QUERY of a batch: [int32 np][pad][double2 local[n]]
    int32_t qstale;
    int32_t pad;
    const int32_t *gov;     // DEVICE: the scan's trigger-chain structure [n][2] computed once in the sensor frame (ym_scan_create),
                            // valid at any pose (see structure_kernel); null: compute it from the projected points
    const int32_t *cidx;    // DEVICE, with gov: beam -> index of its point reading among the compacted ones, or -1 (no reading)
    int32_t cnp;            // ... and how many point readings there are
    int32_t pad2;
    const double *pose_dev; // DEVICE pointer to the pose (x, y, heading) when the host does not know it yet -- a scan of a
                            // device-chained sequence whose match is still in flight (ym_map_sequence) --, else null
};

// A matcher caches, per resident base scan and pose, what LocalizedRangeScan::Update and the viewpoint-independent half
// of FindValidPoints produce (Karto keeps m_PointReadings the same way until the pose is set again):
//   [int32 np][12 bytes pad][double2 pts[n]: world point readings, compacted, beam order]
//   [int2 gov[n]: for point i the trigger-chain node s that decides its run and t = nxt[s] (np = the run never closes)]
#define YM_CACHE_HEADER 16
#define YM_CACHE_BYTES(n) ((size_t)YM_CACHE_HEADER + (size_t)(n) * 24)
#define YM_QCACHE_BYTES(n) ((size_t)YM_CACHE_HEADER + (size_t)(n) * 16)

// one batch item = one (query, chain) problem
struct YmItem {
    int32_t query;      // index into the scan-ref table
    int32_t base_begin; // first base scan in the scan-ref table
    int32_t base_count;
    int32_t pad;        // batches: the item's query slot (distinct queries of a call are projected once)
};

// per-item device state handed from kernel to kernel, and finally copied back
struct YmItemState {
    double pose[3];       // query pose (search centre of the coarse pass)
    double off_x, off_y;  // world coordinate of ROI cell (0,0)
    double center[3];     // centre of the NEXT pass (coarse: pose, fine: coarse mean)
    double mean[3];       // result of the last pass
    double response;      // clamped best response of the last pass
    double coarse_response;
    double cov[9];
    int32_t nq;           // number of query point readings (normaliser)
    int32_t status;
    int32_t regular[2];   // per pass: hypothesis cells form an exact lattice (fast path legal)
    int32_t base_count;   // chain length of this item (copied from the call descriptor)
    int32_t qslot;        // which slot of the query-point buffer holds this item's query (when not in the point cache)
    const void *ql;       // the item's query points in the sensor frame (double2[nq])
    // "yagpy" semantics only: per pass lattice sizes and find_best_pose's return tuple
    int32_t ydims[2][3];  // nx, ny, nt
    int32_t ypad[2];
    double ybest[2][8];   // response, x, y, t, xx, yy, xy, th
    // ... and, when the coarse pass goes through the production correlate kernels (yag_lattice_kernel): xvals[0], yvals[0] of the
    // coarse lattice -- the lookup cell of a (point, angle) pair is the cell hypothesis (0, 0) reads (ym_k_common.hpp, lookup_cell_sem)
    double ylat[2];
};
