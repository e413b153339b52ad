// ym_k_occupancy.hpp -- occupancy-grid rendering: the other native call of yag-slam's loop,
// karto_scanmatcher.create_occupancy_grid(scans, resolution, range_threshold) (/root/reference/yag_slam/graph_slam.py:341-342,
// /root/reference/ros1/slam_node_ros1:187-202).  The wheel's source is not in /root/reference: this restates open_karto's
// OccupancyGrid::{CreateFromScans, ComputeDimensions, AddScan, RayTrace, Update, UpdateCell}, Grid<T>::TraceLine and
// LocalizedRangeScan::Update (bounding box); image codes as slam_node_ros1:199-202 reads them (0 occupied, 200 unknown,
// 255 free).  PARITY UNPINNED, like every Karto-side piece.  Part of ym_kernels.hpp (include that, not this file).
#pragma once

namespace ym {

struct OccArgs {
    const YmScanRef *scans;   // pose + device ranges of every scan
    int32_t n_scans, max_n;
    double range_threshold;   // create_occupancy_grid's third argument: the laser's range threshold for this rendering
    double scale, off_x, off_y; // CoordinateConverter of the grid (scale = 1 / resolution, offset = bounding-box minimum)
    int32_t width, height;
    double *boxes;            // [n_scans][4] xmin, ymin, xmax, ymax of every scan's bounding box
    unsigned *pass, *hits;    // [height][width]
    uint8_t *image;           // [height][width]
};

// LocalizedRangeScan::Update: the bounding box holds the sensor position and every point reading with
// minimum range <= r <= range threshold.  grid (n_scans), 256 threads
__global__ __launch_bounds__(256) void occ_bbox_kernel(OccArgs a) {
    __shared__ double scratch[16];
    const YmScanRef sr = a.scans[blockIdx.x];
    double x0 = sr.pose[0], y0 = sr.pose[1], x1 = sr.pose[0], y1 = sr.pose[1];
    for (int i = threadIdx.x; i < sr.n; i += 256) {
        const double r = sr.ranges[i];
        if (!(r >= sr.min_range && r <= a.range_threshold)) continue;
        const double angle = sr.pose[2] + sr.min_angle + i * sr.angle_inc;
        const double px = sr.pose[0] + r * cos(angle), py = sr.pose[1] + r * sin(angle);
        x0 = px < x0 ? px : x0; x1 = px > x1 ? px : x1;
        y0 = py < y0 ? py : y0; y1 = py > y1 ? py : y1;
    }
    struct OpMinD { __device__ double operator()(double p, double q) const { return p < q ? p : q; } };
    x0 = block_reduce(x0, OpMinD(), 1e300, scratch);
    y0 = block_reduce(y0, OpMinD(), 1e300, scratch);
    x1 = block_reduce(x1, OpMaxD(), -1e300, scratch);
    y1 = block_reduce(y1, OpMaxD(), -1e300, scratch);
    if (threadIdx.x == 0) {
        double *b = a.boxes + 4 * (size_t)blockIdx.x;
        b[0] = x0; b[1] = y0; b[2] = x1; b[3] = y1;
    }
}

// OccupancyGrid::AddScan + RayTrace + Grid::TraceLine: one thread per beam.  Pass and hit counts are sums, so the
// order the rays are traced in does not matter.  grid (ceil(max_n / 256), n_scans)
__global__ __launch_bounds__(256) void occ_trace_kernel(OccArgs a) {
    const YmScanRef sr = a.scans[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= sr.n) return;
    const double r = sr.ranges[i];
    // AddScan: readings at or below the minimum range, at or beyond the maximum range, or NaN are ignored
    if (r <= sr.min_range || r >= sr.range_threshold /* max_range travels in this field here */ || isnan(r)) return;
    const bool end_valid = r < (a.range_threshold - YM_KT_TOLERANCE);
    const double angle = sr.pose[2] + sr.min_angle + i * sr.angle_inc;
    double px = sr.pose[0] + r * cos(angle), py = sr.pose[1] + r * sin(angle);
    if (r >= a.range_threshold) { // trace up to the range threshold only
        const double ratio = a.range_threshold / r;
        const double dx = px - sr.pose[0], dy = py - sr.pose[1];
        px = sr.pose[0] + ratio * dx;
        py = sr.pose[1] + ratio * dy;
    }
    int x0 = world_to_grid(sr.pose[0], a.off_x, a.scale), y0 = world_to_grid(sr.pose[1], a.off_y, a.scale);
    int x1 = world_to_grid(px, a.off_x, a.scale), y1 = world_to_grid(py, a.off_y, a.scale);
    const int tx = x1, ty = y1;
    // Grid<T>::TraceLine (Bresenham, both end cells included)
    const bool steep = abs(y1 - y0) > abs(x1 - x0);
    if (steep) { int t = x0; x0 = y0; y0 = t; t = x1; x1 = y1; y1 = t; }
    if (x0 > x1) { int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
    const int delta_x = x1 - x0, delta_y = abs(y1 - y0);
    int error = 0, y = y0;
    const int ystep = y0 < y1 ? 1 : -1;
    for (int x = x0; x <= x1; x++) {
        const int cx = steep ? y : x, cy = steep ? x : y;
        error += delta_y;
        if (2 * error >= delta_x) { y += ystep; error -= delta_x; }
        if (cx >= 0 && cx < a.width && cy >= 0 && cy < a.height) atomicAdd(&a.pass[(size_t)cy * a.width + cx], 1u);
    }
    // RayTrace: a valid end point counts once more as a pass, and as a hit
    if (end_valid && tx >= 0 && tx < a.width && ty >= 0 && ty < a.height) {
        atomicAdd(&a.pass[(size_t)ty * a.width + tx], 1u);
        atomicAdd(&a.hits[(size_t)ty * a.width + tx], 1u);
    }
}

// OccupancyGrid::Update / UpdateCell (MinPassThrough 2, OccupancyThreshold 0.1) -> the image codes the ROS node reads
__global__ __launch_bounds__(256) void occ_update_kernel(OccArgs a) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)a.width * a.height) return;
    const unsigned pass = a.pass[i], hits = a.hits[i];
    uint8_t v = 200; // GridStates_Unknown
    if (pass > 2u) {
        const double ratio = (double)hits / (double)pass;
        v = ratio > 0.1 ? 0 /* GridStates_Occupied */ : 255 /* GridStates_Free */;
    }
    a.image[i] = v;
}

}  // namespace ym
