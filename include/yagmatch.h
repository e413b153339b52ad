/*
 * yagmatch.h -- C ABI of libyagmatch.so, the MI355X-native correlative scan matcher.
 *
 * Drop-in boundary for yag-slam's match_scan path.  The reference crosses into native code through
 * pybind11 (karto_scanmatcher==1.0.0, /root/reference/setup.py:46), not through a C API, so each
 * entry point below names the pybind11 object/method it replaces:
 *
 *   ym_create / ym_destroy        karto_scanmatcher.Wrapper(ScanMatcherConfig)
 *                                 /root/reference/yag_slam/scan_matching.py:33-38, /root/reference/test.py:24-25
 *   ym_scan_create / _set_pose    karto_scanmatcher.LocalizedRangeScan(LaserScanConfig, ranges, Pose2, Pose2, num, time)
 *                                 and its .corrected_pose setter
 *                                 /root/reference/yag_slam/models.py:37-39,67-75, /root/reference/test.py:27-36
 *   ym_match_scans                Wrapper.match_scan(query._scan, [b._scan ...], penalty, do_fine)
 *                                 /root/reference/yag_slam/scan_matching.py:40-42, /root/reference/test.py:38
 *   ym_match                      same call, for callers that hold plain range arrays (no resident scan)
 *   ym_match_batch, ym_batch_*    the serial chain loop of GraphSlam.try_to_close_loop
 *                                 /root/reference/yag_slam/graph_slam.py:217-236 (one query, many chains)
 *   ym_match_pairs, ym_pairs_create  N independent Wrapper.match_scan calls (N x /root/reference/yag_slam/graph_slam.py:326:
 *                                 N robots, or N segments of a log replayed side by side) in one enqueue: item i =
 *                                 query i against chain i
 *   ym_result                     the returned object's .response / .covariance / .best_pose
 *                                 /root/reference/yag_slam/scan_matching.py:42, /root/reference/test.py:39-41
 *
 * Conventions: POD only; the caller owns every input buffer and the library copies on entry; no
 * C++ exception crosses the boundary -- functions return YM_OK (0) or a negative YM_ERR_* and
 * ym_last_error() gives the text (thread-local).  One ym_matcher owns one HIP stream and one
 * device workspace: it is NOT re-entrant (the reference's matcher is not either: it owns a mutable
 * grid and is driven from one worker thread, /root/reference/ros1/slam_node_ros1:223-255).
 * Distinct matchers are independent.  There is no CPU fallback: without a usable HIP device
 * ym_create fails with YM_ERR_NO_DEVICE.
 */
#ifndef YAGMATCH_H
#define YAGMATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YM_VERSION 1

enum {
    YM_OK = 0,
    YM_ERR_INVALID = -1,     /* bad argument / bad config */
    YM_ERR_NO_DEVICE = -2,   /* no HIP device, or device index out of range */
    YM_ERR_HIP = -3,         /* a HIP runtime call failed */
    YM_ERR_UNSUPPORTED = -4, /* valid request this build cannot serve */
    YM_ERR_RANGE = -5,       /* Karto's "index out of range" / "unable to find best position" */
    YM_ERR_BUSY = -6         /* async slot still in flight / not submitted */
};

enum { YM_SEM_KARTO = 0, YM_SEM_YAGPY = 1 };

/* The 11 keys of yag-slam's config dict (/root/reference/yag_slam/helpers.py:339-351), Karto's
 * MinimumDistancePenalty, and the semantics switch. */
typedef struct ym_config {
    double angle_variance_penalty;
    double distance_variance_penalty;
    double coarse_search_angle_offset;
    double coarse_angle_resolution;
    double fine_search_angle_resolution;
    double range_threshold;
    double minimum_angle_penalty;
    double minimum_distance_penalty;
    double search_size;
    double resolution;
    double smear_deviation;
    int32_t use_response_expansion;
    int32_t semantics; /* YM_SEM_KARTO | YM_SEM_YAGPY */
} ym_config;

/* LaserScanConfig + ranges + corrected pose (/root/reference/yag_slam/models.py:25-39) */
typedef struct ym_scan_desc {
    const double *ranges; /* host pointer, n readings */
    int32_t n;
    int32_t reserved;
    double min_angle;
    double max_angle;
    double angle_increment;
    double min_range;
    double max_range;
    double range_threshold;
    double pose[3]; /* x, y, heading */
} ym_scan_desc;

typedef struct ym_result {
    double response;
    double pose[3];  /* best_pose x, y, heading */
    double cov[9];   /* row-major 3x3 covariance */
    double coarse_response; /* best response of the (last) coarse pass */
    int64_t hypotheses;     /* lattice points scored (all passes, incl. expansions) */
    int32_t coarse_dims[3]; /* nx, ny, ntheta */
    int32_t fine_dims[3];   /* nx, ny, ntheta (0 when !refine) */
    int32_t n_query_points; /* response normaliser */
    int32_t expansions;     /* response-expansion retries taken */
    int32_t status;         /* YM_OK or YM_ERR_RANGE for this item */
    int32_t reserved;
} ym_result;

typedef struct ym_matcher ym_matcher;
typedef struct ym_scan ym_scan;

/* ---- library / device ---- */
int ym_version(void);
/* first 16 hex digits of the SHA-256 of the sources this library was built from (csrc/Makefile): ties a profile to a build */
const char *ym_build_id(void);
int ym_device_count(void);
const char *ym_last_error(void);

/* ---- matcher ---- */
ym_matcher *ym_create(const ym_config *cfg, int device);
void ym_destroy(ym_matcher *m);
int ym_get_config(const ym_matcher *m, ym_config *out);
/* run on a caller-owned hipStream_t (e.g. torch's current stream); NULL -> the matcher's own */
int ym_set_stream(ym_matcher *m, void *hip_stream);
int ym_synchronize(ym_matcher *m);

/* ---- resident scans (device twin of LocalizedRangeScan) ----
 * ym_scan_create copies the readings (desc->ranges is free again on return) and costs one kernel launch without a
 * synchronisation (~20 us): the upload and the scan's chain structure complete on the device while the caller goes on,
 * and whoever uses the scan first waits for them.  The device memory comes from a per-device pool of the library:
 * ym_scan_destroy never synchronises the device (hipFree does), the block of a destroyed scan serves a later
 * ym_scan_create, and the pool keeps what it has allocated for the life of the process (35 KB per 1081-beam scan alive
 * at the same time). */
ym_scan *ym_scan_create(int device, const ym_scan_desc *desc);
/* n scans at once: what n ym_scan_create calls leave behind (the same host code per scan, the same kernel), in one pool transaction,
 * one upload and one launch per 2048 scans -- the reference builds one C++ scan per Python scan before match_scan
 * (/root/reference/yag_slam/models.py:25-39); a node that receives the scans of N robots per step, or a map file being loaded,
 * builds thousands.  out receives n handles.  All or nothing: on failure nothing stays allocated.  The scans come back complete
 * (nothing of their creation is still in flight).  ym_scans_destroy: n ym_scan_destroy calls in one pool transaction. */
int ym_scans_create(int device, const ym_scan_desc *descs, int n, ym_scan **out);
void ym_scans_destroy(ym_scan *const *scans, int n);
int ym_scan_set_pose(ym_scan *s, double x, double y, double heading);
/* n poses (x, y, heading) written to n scans in one call -- what the reference does scan by scan after every graph
 * optimisation (/root/reference/yag_slam/graph_slam.py:263-272: `vtx.obj.corrected_pose = ...` for EVERY vertex, one pybind11
 * write each, /root/reference/yag_slam/models.py:67-75).  All or nothing: a null entry fails the call before any pose is written. */
int ym_scans_set_poses(ym_scan *const *scans, const double *xyz, int n);
int ym_scan_get_pose(const ym_scan *s, double pose[3]);
int ym_scan_size(const ym_scan *s);
/* 1: the scan's trigger-chain structure, computed once at creation in the sensor frame, holds at every pose (no distance
 * test of the valid-point filter came within 1e-9 m^2 of its threshold); 0: the matchers recompute it per pose */
int ym_scan_structure_trusted(const ym_scan *s, int semantics);
void ym_scan_destroy(ym_scan *s);

/* ---- the hot path ---- */
int ym_match(ym_matcher *m, const ym_scan_desc *query, const ym_scan_desc *base, int n_base,
             int penalize, int refine, ym_result *out);
int ym_match_scans(ym_matcher *m, const ym_scan *query, const ym_scan *const *base, int n_base,
                   int penalize, int refine, ym_result *out);

/* The matcher calls of GraphSlam.process_scan for a whole trajectory, in one call
 * (/root/reference/yag_slam/graph_slam.py:306-339; the host loop a robot log is replayed with):
 *   for i = max(start, 1) .. n-1:
 *       prior_i     = corrected_{i-1} (+) (odom_i (-) odom_{i-1})                                graph_slam.py:320-324
 *       results[i]  = match(scans[i] at prior_i, scans[max(0, i - buffer_len) .. i-1], penalize, refine)
 *       corrected_i = results[i].pose                                                             graph_slam.py:326-337
 * with tiny_tf's planar Transform arithmetic ((+) composes, a (-) b = inverse(b) (+) a), operation for operation what
 * yag_slam_amd/transform.py does, so the poses are bit-identical to the per-scan calls.  scans[0 .. start) are the
 * running chain so far and keep their poses; every later scan's pose is set to its prior and then to its result (as
 * ym_scan_set_pose would).  odom = n poses (x, y, heading).  results = n entries, those of scans that are not matched
 * are zeroed.  Stops at the first scan whose match reports YM_ERR_RANGE in its result (its pose stays at the prior);
 * *n_done = index of the first scan NOT completed (n when all are).
 * device_chain != 0: no host round trip between steps either.  The steps are enqueued back to back, 128 at a time; the
 * kernel that ends step i leaves scan i's pose and scan i+1's prior in device memory, where step i+1's kernels read them,
 * and the host -- planning ahead of the device -- sizes each step's raster from poses it dead-reckons with the odometry
 * alone.  The priors are then composed with the DEVICE's cos / sin: poses and responses agree with the synchronous form
 * to rounding (~1e-12), not bit for bit.  A step whose cells leave the predicted rectangle, that Karto would abort, or
 * that needs a response expansion is detected on the device, the rest of its segment is skipped, and the step is repeated
 * synchronously.  Karto semantics, resident scans, buffer_len < 16; anything else runs synchronously. */
int ym_map_sequence(ym_matcher *m, ym_scan *const *scans, const double *odom, int n, int start, int buffer_len,
                    int penalize, int refine, int device_chain, ym_result *results, int32_t *n_done);

/* The same for ONE scan, for callers that receive their scans one at a time (GraphSlam.process_scan itself): prior =
 * chain[n_chain-1]'s pose (+) (odom_query (-) odom_last), the query's pose is set to it, the match runs against `chain`,
 * and the query's pose becomes the result's (it stays at the prior if result->status != 0).  Bit-identical to setting the
 * prior with ym_scan_set_pose and calling ym_match_scans; it only saves the caller the arithmetic and two calls. */
int ym_process_scan(ym_matcher *m, ym_scan *query, ym_scan *const *chain, int n_chain, const double *odom_last,
                    const double *odom_query, int penalize, int refine, ym_result *result);

/* counters of ym_map_sequence since ym_create: device-chained segments enqueued, segments a fault cut short, steps run
 * synchronously (all of them without device_chain) */
int ym_sequence_stats(const ym_matcher *m, int64_t *segments, int64_t *faults, int64_t *sync_steps);

/* Pipelined form: enqueue on the matcher's stream, collect later.  `slot` in [0, ym_async_slots). */
int ym_async_slots(const ym_matcher *m);
int ym_match_scans_async(ym_matcher *m, const ym_scan *query, const ym_scan *const *base, int n_base,
                         int penalize, int refine, int slot);
int ym_wait(ym_matcher *m, int slot, ym_result *out);

/* One query against n_chains candidate chains; chain c = scans[chain_offsets[c] .. chain_offsets[c+1]).
 * per_chain (nullable) receives every chain's result; best/best_chain (nullable) the arg-max over
 * response, ties to the lowest chain index. */
int ym_match_batch(ym_matcher *m, const ym_scan *query, const ym_scan *const *scans,
                   const int32_t *chain_offsets, int n_chains, int penalize, int refine,
                   ym_result *per_chain, ym_result *best, int32_t *best_chain);

/* The same as a reusable object + pipelined run.  A batch only remembers WHICH scans form the chains;
 * poses are read from the scans at every run.  If dev_best_out (nullable, DEVICE pointer to 8
 * doubles) is given, {response, chain_id_base + best chain, x, y, heading, cov_xx, cov_yy, cov_tt} of
 * the best chain is also left on the device, stream-ordered, as the payload of a cross-rank
 * arg-max (RCCL all-gather of one such record per rank).  That record is written before Karto's
 * response expansion (a host-side re-run of the items whose coarse response is 0): when an
 * expansion took place, ym_batch_wait rewrites it from the final results, so a caller that needs
 * the post-expansion record waits for the slot before it gathers. */
typedef struct ym_batch ym_batch;
ym_batch *ym_batch_create(ym_matcher *m, const ym_scan *query, const ym_scan *const *scans,
                          const int32_t *chain_offsets, int n_chains);
void ym_batch_destroy(ym_batch *b);
int ym_batch_size(const ym_batch *b);
int ym_batch_run_async(ym_matcher *m, const ym_batch *b, int penalize, int refine, int slot,
                       int64_t chain_id_base, void *dev_best_out);
int ym_batch_wait(ym_matcher *m, int slot, ym_result *per_chain, ym_result *best, int32_t *best_chain);

/* n_items INDEPENDENT matches in one enqueue: item i = queries[i] against scans[chain_offsets[i] .. chain_offsets[i+1])
 * -- what n_items separate Wrapper.match_scan(query_i._scan, chain_i, penalty, do_fine) calls compute
 * (/root/reference/yag_slam/scan_matching.py:40-42, called once per incoming scan at /root/reference/yag_slam/graph_slam.py:326),
 * item for item bit-identical to ym_match_scans(queries[i], chain i).  A query object may serve several items (it is projected,
 * and its (beam, angle) pair lists are built, once per distinct object); ym_match_batch is the special case of ONE query.
 * per_item receives n_items results.  ym_pairs_create gives the reusable form: the object is a ym_batch -- run it with
 * ym_batch_run_async / ym_batch_wait (whose best / best_chain then name the item with the highest response), free it with
 * ym_batch_destroy. */
int ym_match_pairs(ym_matcher *m, const ym_scan *const *queries, const ym_scan *const *scans,
                   const int32_t *chain_offsets, int n_items, int penalize, int refine, ym_result *per_item);
ym_batch *ym_pairs_create(ym_matcher *m, const ym_scan *const *queries, const ym_scan *const *scans,
                          const int32_t *chain_offsets, int n_items);

/* ---- one match split over several matchers by coarse angle (BASELINE configs[4] on 8 GPUs: one matcher per GPU) ----
 * Every rank rasterises the same grid and scores the coarse angles [k_begin, k_end) only, writing their responses at
 * their place in the caller-owned device volume dev_resp[nt][ny][nx] (doubles) and the per-(x, y) maxima of its slices
 * into dev_probs[ny][nx] (doubles, cleared by the call).  The caller then completes both across ranks on the matcher's
 * stream -- all-gather of the slices, element-wise MAX of dev_probs (RCCL) -- and calls _finish, which runs the rest of
 * the match (arg-max, tie mean, covariances, the fine pass) on the whole volume: the result is bit-identical to an
 * unsplit ym_match_scans on every rank.  Karto semantics only. */
int ym_coarse_dims(const ym_matcher *m, int32_t dims[3]); /* nx, ny, ntheta of the coarse lattice */
int ym_match_slice_begin(ym_matcher *m, const ym_scan *query, const ym_scan *const *base, int n_base, int penalize,
                         int refine, int k_begin, int k_end, double *dev_resp, double *dev_probs);
int ym_match_slice_finish(ym_matcher *m, ym_result *out);

/* ---- match against a prebuilt map (reference: Scan2DMatcherPy.match_scan_sets_with_map,
 * /root/reference/yag_slam/scan_matching.py:124-173; YM_SEM_YAGPY matchers only -- Karto has no such entry) ---- */
typedef struct ym_map ym_map;
/* occupancy_grid_map_to_correlation_grid (/root/reference/yag_slam/helpers.py:24-34): every pixel of `image` equal to
 * occupied_value becomes 1.0 and is max-smeared with the matcher's kernel (resolution, smear_deviation), on the device */
ym_map *ym_map_from_occupancy(ym_matcher *m, const uint8_t *image, int width, int height, int pitch, int occupied_value);
/* a correlation grid (float64 in [0, 1], row-major [y][x]) computed elsewhere */
ym_map *ym_map_from_grid(ym_matcher *m, const double *cgrid, int width, int height);
int ym_map_size(const ym_map *map, int *width, int *height);
int ym_map_read(const ym_map *map, double *out, int64_t out_count); /* the float grid, width*height entries */
void ym_map_destroy(ym_map *map);
/* one find_best_pose_non_symmetric pass (/root/reference/yag_slam/helpers.py:434-573) */
typedef struct ym_map_search {
    double xy_search, xy_step;       /* +- metres around the centre, lattice step */
    double angle_search, angle_step; /* +- radians, step */
    double grid_resolution;          /* cell size the pass uses for indexing the map and in the penalty */
    int32_t penalize;
    int32_t reserved;
} ym_map_search;
/* The query scans' point readings (at their own poses) are matched as ONE point set against the map whose cell (0, 0)
 * is at world (ox, oy): coarse pass around the mean of the query poses (coarse == NULL: the reference's constants
 * 0.25 m / 0.01 m / 0.1 rad / 0.01 rad, cell size 0.05, no penalty -- scan_matching.py:152-153), then, if refine, the fine
 * pass +-2 cells, +-0.01745 rad at 0.00349 with the matcher's resolution and `penalize`.  out->pose is the corrected
 * mean pose (x, y, heading); the caller moves every query by its difference to the uncorrected mean. */
int ym_match_map(ym_matcher *m, const ym_map *map, double ox, double oy, const ym_scan *const *queries, int n_queries,
                 int penalize, int refine, const ym_map_search *coarse, ym_result *out);

/* ---- occupancy-grid rendering: karto_scanmatcher.create_occupancy_grid(scans, resolution, range_threshold)
 * (/root/reference/yag_slam/graph_slam.py:341-342, /root/reference/ros1/slam_node_ros1:187-202).  Every scan is ray-traced
 * from its pose (open_karto OccupancyGrid::CreateFromScans): cells count passes and end-point hits, a cell passed more
 * than twice is occupied when hits / passes > 0.1, else free.  image[y][x] uses the codes the ROS node reads: 0 occupied,
 * 200 unknown, 255 free; row 0 is the lowest y; cell (0, 0) is at world (offset_x, offset_y).  Parity unpinned (the
 * wheel's source is not in the reference tree). */
typedef struct ym_occupancy ym_occupancy;
typedef struct ym_occupancy_info {
    int32_t width, height;
    double offset_x, offset_y, resolution;
} ym_occupancy_info;
ym_occupancy *ym_occupancy_create(const ym_scan *const *scans, int n_scans, double resolution, double range_threshold);
int ym_occupancy_get_info(const ym_occupancy *og, ym_occupancy_info *info);
int ym_occupancy_read(const ym_occupancy *og, uint8_t *image, int64_t image_bytes); /* width*height bytes */
void ym_occupancy_destroy(ym_occupancy *og);

/* ---- introspection for parity tests (state of the LAST completed synchronous match) ---- */
typedef struct ym_grid_info {
    int32_t width, height, pitch; /* device window (bytes) */
    int32_t origin_x, origin_y;   /* window cell (0,0) in Karto storage coordinates (incl. border) */
    int32_t storage_w, storage_h; /* Karto's full storage size the window is cut from */
    int32_t roi_x, roi_y, roi_w, roi_h;
    double offset_x, offset_y;    /* world coordinate of ROI cell (0,0) */
} ym_grid_info;
int ym_debug_grid_info(ym_matcher *m, int item, ym_grid_info *info);
int ym_debug_grid(ym_matcher *m, int item, uint8_t *out, int64_t out_bytes); /* height*pitch bytes */
/* integer correlation sums [itheta][iy][ix] of pass 0 (coarse) / 1 (fine) */
int ym_debug_sums(ym_matcher *m, int item, int pass, uint32_t *out, int64_t out_count);
/* query points in the sensor frame (xy interleaved), returns count via *n */
int ym_debug_query_local(ym_matcher *m, int item, double *out_xy, int32_t cap, int32_t *n);
/* window cell coordinates of the rasterised base points of `item`, per base scan slot:
 * out[(slot*max_n + i)*2 + {0,1}] = wx, wy  or (INT32_MIN, INT32_MIN) for filtered points */
int ym_debug_cells(ym_matcher *m, int item, int32_t *out, int64_t out_count, int32_t *max_n);

/* Switches of the parity tests: every row names a path the tests force so that each kernel and each host decision is compared with
 * the oracle (or with the default path) bit for bit.  None changes a result.  Development and timing switches that no test uses
 * (2 - 5, 9, 23, 26, 29, 33 - 36, 38, 40, 42, 44) are described where they are implemented, yag_slam_amd/csrc/ym_abi_debug.hpp.
 *
 *   option  value                          effect
 *   ------  -----------------------------  ------------------------------------------------------------------------------------------
 *    6      0 / 1 / 2                      finish stage: by batch size / fine + final kernels / the one-block finish kernel
 *    7      0 / 1 / 2                      point cache of resident base scans: on / off / drop every entry now
 *    8      KiB                            point cache limit (small values force the start-over path)
 *   10      1                              order-dependent smear rule always through the global-memory kernel
 *   11      256 / 1024 / 0                 threads per finish block (0 = by batch size)
 *   12      1                              keep the coarse integer sums for ym_debug_sums whatever the call (default: fewer than 8
 *                                          items on a lattice of at most 65536 hypotheses)
 *   13      0 / 1 / 2                      merging of consecutive beams with equal lookup offsets in the direct correlate: by grid
 *                                          coarseness / always / never
 *   14      0 / 1 / 2 / 3 / 4              coarse correlate of batches: region correlate on lattices up to 26 x 32 and gather correlate
 *                                          on others up to 48 x 64 / always the direct kernel / the LDS correlates' per-cell path / their
 *                                          "lists do not fit" path / the gather correlate also where the region correlate would run
 *   15      n                              waves per region-correlate block / angles per wave of the gather correlate
 *   16      n                              raster blocks per item on batches (0 = sized by the previous call's longest tile list)
 *   17      n                              blocks per item of the gather correlate (each takes a share of the angles)
 *   18      n / -1                         hit slots per entry of the raster's work list / no hit lists
 *   19      n                              units per LDS buffer of the gather correlate (small values cut regions into chunks)
 *   20      bytes                          LDS a gather block may use (small values make the regions small)
 *   21      2                              the region correlate leaves the scoring of its sums to the score kernel
 *   24      0                              trigger chains of base scans recomputed at every pose (not from the creation-time structure)
 *   25      tiles                          margin around the raster rectangle a device-chained step predicts (negative: every step faults)
 *   28      n / 0                          batch size from which BOTH LDS correlates replace the direct kernel (>= 8) / the defaults
 *                                          (gather correlate 64, region correlate 48)
 *   30      32 / 64 / 0                    rows per raster tile whatever the call / the host's choice
 *   31      0                              a synchronous match waits for the creation launch of a just-created query scan
 *   32      0 / 1, 2 .. 6                  form of the region correlate: correlate_region_kernel / the forms that lost (scripts/exp/forms),
 *                                          only in builds made with `make experimental`: the product library answers YM_ERR_UNSUPPORTED
 *   37      1                              the raster's row pass by bit scans instead of its tables
 *   39      1                              every call writes the column planes and the region correlate stages from them
 *   41      n                              items up to which the order-dependent smear rule runs in its split form (8; 0 = never)
 *   43      80 / 100 / 128                 region height of experimental form 5
 *   45      0                              the pair lists of a single-query batch are built at every call (default: a call whose query,
 *                                          pose, window and lattice equal those of the last list build finds them in place)
 *   46      0 / 2                          YM_SEM_YAGPY, 0: both passes scored pair by pair, the Python rule as written (default: the coarse
 *                                          pass's integer sums come from the production correlate kernels wherever the item's roundings
 *                                          provably form a lattice: ym_debug_counters; the fine pass's are taken row by row);
 *                                          2: the default, with the fine pass reading its rows byte by byte (the path of a row whose
 *                                          columns do not fit one 8-byte read)
 */
int ym_debug_option(ym_matcher *m, int option, int value);

/* counters since ym_create, out[0 .. min(count, YM_DEBUG_COUNTERS)):
 *   [0] YM_SEM_YAGPY items whose coarse sums came from the production correlate kernels, [1] items that fell back to the pair-by-pair
 *   kernel (a rounding tie that falls differently along the lattice, a read outside the device window, an np.arange longer than the launch
 *   lattice), [2] (point, angle) pairs that needed the hypothesis-by-hypothesis check, [3] pairs that failed it;
 *   [4] single-query batches that found their pair lists in place (option 45);
 *   [5] the coarse correlate the LAST call launched: 0 correlate_kernel, 1 correlate_region_kernel, 2 gather_kernel, -1 none */
#define YM_DEBUG_COUNTERS 8
int ym_debug_counters(ym_matcher *m, int64_t *out, int32_t count);

/* development aid: 100 MHz wall-clock stamps written by block 0 of each kernel at phase boundaries.
 * Reads the stamps of the last call into out[0..count) (count <= 32), then switches stamping on/off. */
int ym_debug_stamps(ym_matcher *m, int enable, uint64_t *out, int32_t count);

/* ---- profiling: HIP-event timing of the correlate kernel on the matcher's stream ---- */
int ym_profile_enable(ym_matcher *m, int on);
/* which: 0 = correlate (coarse), 1 = raster, 2 = whole call; returns accumulated ms and launch count */
int ym_profile_read(ym_matcher *m, int which, double *ms_total, int64_t *launches, int reset);
/* point cache of the matcher (what Karto's LocalizedRangeScan keeps in m_PointReadings until the pose is set again):
 * scans found current / scans (re)projected since the matcher was created */
int ym_cache_stats(const ym_matcher *m, int64_t *hits, int64_t *misses);

#ifdef __cplusplus
}
#endif
#endif /* YAGMATCH_H */
