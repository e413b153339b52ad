// ym_kernels.hpp -- gfx950 kernels of the correlative scan matcher (karto semantics).
//
// Pipeline of one call (B independent items, one item = one query scan vs one chain of base scans):
//   prepare_kernel   point readings, valid-point filter, grid cells        (Karto LocalizedRangeScan::Update,
//                                                                           ScanMatcher::FindValidPoints, AddScan)
//   raster_kernel    correlation-grid window with the Gaussian max-smear   (AddScans + CorrelationGrid::SmearPoint)
//   offsets_kernel   per-angle cell-offset table + hypothesis cells        (GridIndexLookup::ComputeOffsets)
//   correlate_*      integer gather-reduce over the (x, y, theta) lattice  (CorrelateScan loops + GetResponse)
//   reduce_kernel    response, penalty, arg-max, tie mean, covariances     (CorrelateScan tail,
//                                                                           ComputePositionalCovariance,
//                                                                           ComputeAngularCovariance)
// All fp64 arithmetic is written operation-for-operation like oracle/ym_oracle.c and the library is
// compiled with -ffp-contract=off, so responses are bit-identical to the oracle's.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "ym_types.h"

namespace ym {

// ------------------------------------------------------------------ Karto math:: helpers
__device__ __forceinline__ double kt_round(double v) { return v >= 0.0 ? floor(v + 0.5) : ceil(v - 0.5); }
__device__ __forceinline__ bool kt_double_equal(double a, double b) {
    double d = a - b;
    return d < 0.0 ? d >= -YM_KT_TOLERANCE : d <= YM_KT_TOLERANCE;
}
__device__ inline double kt_normalize_angle(double angle) {
    while (angle < -YM_KT_PI) {
        if (angle < -YM_KT_2PI) angle += (double)(unsigned int)(angle / -YM_KT_2PI) * YM_KT_2PI;
        else angle += YM_KT_2PI;
    }
    while (angle > YM_KT_PI) {
        if (angle > YM_KT_2PI) angle -= (double)(unsigned int)(angle / YM_KT_2PI) * YM_KT_2PI;
        else angle -= YM_KT_2PI;
    }
    return angle;
}
__device__ inline double kt_normalize_angle_difference(double minuend, double subtrahend) {
    while (minuend - subtrahend < -YM_KT_PI) minuend += YM_KT_2PI;
    while (minuend - subtrahend > YM_KT_PI) minuend -= YM_KT_2PI;
    return minuend;
}
__device__ __forceinline__ int world_to_grid(double w, double off, double scale) {
    return (int)kt_round((w - off) * scale);
}

// ------------------------------------------------------------------ block helpers (256 or 1024 threads)
// exclusive prefix position of `flag` inside the block + block total; wave = 64 lanes
__device__ __forceinline__ int block_scan_flag(bool flag, int *total, int *wave_counts) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    unsigned long long m = __ballot(flag);
    int pre = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_counts[w] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < nw; i++) {
        int c = wave_counts[i];
        if (i < w) base += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return base + pre;
}

template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce(T v, Op op) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o));
    return v;
}
// block-wide reduce, result valid in every thread; scratch >= 16 entries of T
template <typename T, typename Op>
__device__ __forceinline__ T block_reduce(T v, Op op, T identity, T *scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_reduce(v, op);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    T r = identity;
    for (int i = 0; i < nw; i++) r = op(r, scratch[i]);
    return r;
}
struct OpMaxD { __device__ double operator()(double a, double b) const { return a > b ? a : b; } };
struct OpAddD { __device__ double operator()(double a, double b) const { return a + b; } };
struct OpAddU { __device__ unsigned operator()(unsigned a, unsigned b) const { return a + b; } };
struct OpAddI { __device__ int operator()(int a, int b) const { return a + b; } };

// ================================================================== K1 prepare
struct PrepareArgs {
    const YmScanRef *scans;
    const YmItem *items;
    YmGeom g;
    YmItemState *states;
    double2 *qlocal;   // [B][max_n]
    int2 *cells;       // [B][max_base][max_n]  window cell of every compacted base point (or NONE)
    int32_t *counts;   // [B][max_base]         compacted point count per base slot
    int32_t max_n, max_base;
};

// grid (max_base + 1, B), 256 threads, dynamic LDS = max_n * 21 bytes
// blockIdx.x == 0: the query scan; blockIdx.x == 1 + j: base scan j of the item's chain.
__global__ __launch_bounds__(256) void prepare_kernel(PrepareArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int wave_counts[4];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const YmItem it = a.items[b];
    const bool is_query = blockIdx.x == 0;
    const int slot = (int)blockIdx.x - 1;
    if (!is_query && slot >= it.base_count) {
        if (tid == 0) a.counts[(size_t)b * a.max_base + slot] = 0;
        return;
    }
    const YmScanRef sr = a.scans[is_query ? it.query : it.base_begin + slot];
    const YmScanRef qr = a.scans[it.query];
    double *sx = reinterpret_cast<double *>(lds_raw);
    double *sy = sx + a.max_n;
    int *nxt = reinterpret_cast<int *>(sy + a.max_n);
    unsigned char *chain = reinterpret_cast<unsigned char *>(nxt + a.max_n);
    const bool yag = a.g.semantics == 1;

    // ---- point readings, compacted in beam order (LocalizedRangeScan::Update / _get_point_readings)
    const double px = (is_query && yag) ? 0.0 : sr.pose[0];
    const double py = (is_query && yag) ? 0.0 : sr.pose[1];
    const double pt = (is_query && yag) ? 0.0 : sr.pose[2];
    int running = 0;
    for (int c0 = 0; c0 < sr.n; c0 += 256) {
        const int i = c0 + tid;
        double r = 0.0;
        bool ok = false;
        if (i < sr.n) {
            r = sr.ranges[i];
            ok = yag ? !(r > sr.range_threshold || isnan(r)) : (r >= sr.min_range && r <= sr.range_threshold);
        }
        int total;
        const int pos = running + block_scan_flag(ok, &total, wave_counts);
        if (ok) {
            const double angle = pt + sr.min_angle + i * sr.angle_inc;
            sx[pos] = px + r * cos(angle);
            sy[pos] = py + r * sin(angle);
        }
        running += total;
    }
    const int np = running;
    __syncthreads();

    if (is_query) {
        if (tid == 0) {
            YmItemState &st = a.states[b];
            st.pose[0] = sr.pose[0]; st.pose[1] = sr.pose[1]; st.pose[2] = sr.pose[2];
            st.center[0] = sr.pose[0]; st.center[1] = sr.pose[1]; st.center[2] = sr.pose[2];
            st.off_x = sr.pose[0] - (0.5 * (a.g.roi_w - 1) * a.g.res);
            st.off_y = sr.pose[1] - (0.5 * (a.g.roi_w - 1) * a.g.res);
            st.nq = np;
            st.status = 0;
            st.regular[0] = st.regular[1] = 0;
        }
        // sensor-frame coordinates (karto: Transform(pose).InverseTransformPose; yagpy: points_local)
        double2 *ql = a.qlocal + (size_t)b * a.max_n;
        if (yag || (sr.pose[0] == 0.0 && sr.pose[1] == 0.0 && sr.pose[2] == 0.0)) {
            for (int i = tid; i < np; i += 256) ql[i] = make_double2(sx[i], sy[i]);
        } else {
            const double cr = cos(0.0 - sr.pose[2]), sn = sin(0.0 - sr.pose[2]);
            for (int i = tid; i < np; i += 256) {
                const double dx = sx[i] - sr.pose[0], dy = sy[i] - sr.pose[1];
                ql[i] = make_double2(cr * dx + (0.0 - sn) * dy, sn * dx + cr * dy);
            }
        }
        return;
    }

    // ---- valid-point filter (ScanMatcher::FindValidPoints / validate_points), parallel form:
    // nxt[i] = first j > i farther than d from point i; the trigger chain is 0 -> nxt[0] -> ...;
    // the run that ends at chain node t = nxt[s] is kept or dropped by the sign of ss(s, t).
    const double min_sq = yag ? 0.2 * 0.2 : 0.1 * 0.1;
    const double vpx = qr.pose[0], vpy = qr.pose[1];
    for (int i = tid; i < np; i += 256) {
        const double fx = sx[i], fy = sy[i];
        int j = i + 1;
        for (; j < np; j++) {
            const double dx = fx - sx[j], dy = fy - sy[j];
            if (dx * dx + dy * dy > min_sq) break;
        }
        nxt[i] = j;
        chain[i] = 0;
    }
    __syncthreads();
    if (tid == 0) {
        for (int i = 0; i < np; i = nxt[i]) chain[i] = 1;
        a.counts[(size_t)b * a.max_base + slot] = np;
    }
    __syncthreads();
    // world offset of ROI cell (0,0): same expression the query block stores
    const double off_x = qr.pose[0] - (0.5 * (a.g.roi_w - 1) * a.g.res);
    const double off_y = qr.pose[1] - (0.5 * (a.g.roi_w - 1) * a.g.res);
    int2 *cells = a.cells + ((size_t)b * a.max_base + slot) * a.max_n;
    for (int i = tid; i < np; i += 256) {
        bool keep = false;
        int s = yag ? i - 1 : i;
        if (s >= 0) {
            while (!chain[s]) s--;
            const int t = nxt[s];
            if (t < np) {
                const double fx = sx[s], fy = sy[s], cx = sx[t], cy = sy[t];
                const double aa = vpy - fy;
                const double bb = fx - vpx;
                const double cc = fy * vpx - fx * vpy;
                const double ss = cx * aa + cy * bb + cc;
                keep = yag ? (ss > 0.0) : !(ss < 0.0);
            }
        }
        int2 c = make_int2(YM_CELL_NONE, YM_CELL_NONE);
        if (keep) {
            int gx, gy;
            if (yag) {
                gx = (int)rint((sx[i] - off_x) / a.g.res);
                gy = (int)rint((sy[i] - off_y) / a.g.res);
            } else {
                gx = world_to_grid(sx[i], off_x, a.g.scale);
                gy = world_to_grid(sy[i], off_y, a.g.scale);
            }
            if (gx >= 0 && gx < a.g.roi_w && gy >= 0 && gy < a.g.roi_w)
                c = make_int2(gx + a.g.border - a.g.win_origin, gy + a.g.border - a.g.win_origin);
        }
        cells[i] = c;
    }
}

// ================================================================== K2 raster
#define YM_TILE_W 64
#define YM_TILE_H 16
struct RasterArgs {
    const int2 *cells;
    const int32_t *counts;
    const YmItem *items;
    YmGeom g;
    uint8_t *grid;        // [B][win_w rows][pitch]
    size_t grid_stride;   // bytes per item
    const uint8_t *ktab;  // (h+1) x (h+1) quadrant of the smear kernel: ktab[|dy|*(h+1) + |dx|]
    int32_t max_n, max_base;
};

// grid (tiles_x, tiles_y, B), 256 threads.  Each block owns one 64x16 tile of the window and writes
// every byte of it exactly once (so no separate clear pass exists).  The max-stamp of Karto's
// SmearPoint over a set of occupied cells equals, per cell, the kernel value at the nearest occupied
// cell inside the (2h+1)^2 window; with a radially monotone kernel that is
//   max_dy ktab[|dy|][ min |dx| of an occupied cell in row y+dy within h ],
// computed as a row pass followed by a column pass in LDS.
__global__ __launch_bounds__(256) void raster_kernel(RasterArgs a) {
    constexpr int TW = YM_TILE_W, TH = YM_TILE_H, HM = YM_MAX_KERNEL_HALF;
    __shared__ unsigned char occ[(TH + 2 * HM) * (TW + 2 * HM)];
    __shared__ unsigned char grow[(TH + 2 * HM) * TW];
    __shared__ unsigned char kt[(HM + 1) * (HM + 1)];
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int h = a.g.half_kernel;
    const int OW = TW + 2 * h, OH = TH + 2 * h;
    const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
    for (int i = tid; i < OW * OH; i += 256) occ[i] = 0;
    for (int i = tid; i < (h + 1) * (h + 1); i += 256) kt[i] = a.ktab[i];
    __syncthreads();
    const int nb = a.items[b].base_count;
    int any = 0;
    for (int s = 0; s < nb; s++) {
        const int cnt = a.counts[(size_t)b * a.max_base + s];
        const int2 *cells = a.cells + ((size_t)b * a.max_base + s) * a.max_n;
        for (int i = tid; i < cnt; i += 256) {
            const int2 c = cells[i];
            const int lx = c.x - (tx0 - h), ly = c.y - (ty0 - h);
            if (c.x != YM_CELL_NONE && lx >= 0 && lx < OW && ly >= 0 && ly < OH) {
                occ[ly * OW + lx] = 1;
                any = 1;
            }
        }
    }
    any = __syncthreads_or(any);
    uint8_t *grid = a.grid + (size_t)b * a.grid_stride;
    const int y = tid >> 4, x4 = (tid & 15) * 4;
    const bool row_ok = (ty0 + y) < a.g.win_w && (tx0 + x4) < a.g.pitch;
    if (!any) {
        if (row_ok) *reinterpret_cast<uint32_t *>(grid + (size_t)(ty0 + y) * a.g.pitch + tx0 + x4) = 0u;
        return;
    }
    // row pass: nearest occupied |dx| <= h, 255 = none
    for (int i = tid; i < OH * TW; i += 256) {
        const int ry = i / TW, rx = i % TW;
        const unsigned char *row = occ + ry * OW + rx + h;
        int best = 255;
        for (int d = 0; d <= h; d++)
            if (row[-d] | row[d]) { best = d; break; }
        grow[i] = (unsigned char)best;
    }
    __syncthreads();
    uint32_t packed = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int x = x4 + q;
        int v = 0;
        for (int dy = -h; dy <= h; dy++) {
            const int d = grow[(y + h + dy) * TW + x];
            if (d != 255) {
                const int k = kt[(dy < 0 ? -dy : dy) * (h + 1) + d];
                v = k > v ? k : v;
            }
        }
        packed |= (uint32_t)v << (8 * q);
    }
    if (row_ok) *reinterpret_cast<uint32_t *>(grid + (size_t)(ty0 + y) * a.g.pitch + tx0 + x4) = packed;
}

// ================================================================== K3 offsets + hypothesis cells
struct OffsetsArgs {
    YmGeom g;
    YmLattice lat;
    YmItemState *states;
    const double2 *qlocal;
    int32_t *offsets;  // [B][nt_stride][max_n]  window-linear cell offsets per angle
    int32_t *hypcell;  // [B][2][dim_stride]     window x cells (ix) then window y cells (iy)
    int32_t max_n, nt_stride, dim_stride;
};

// grid (ceil(max_n / 256), nt + 1, B); blockIdx.y == nt computes the hypothesis cells.
__global__ __launch_bounds__(256) void offsets_kernel(OffsetsArgs a) {
    const int b = blockIdx.z;
    const YmItemState &st = a.states[b];
    const int k = blockIdx.y;
    if (k == a.lat.nt) {
        if (blockIdx.x != 0) return;
        // CorrelateScan: gridPoint = WorldToGrid(centre + (x, y)), per axis
        int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
        int32_t *cy = cx + a.dim_stride;
        const double start_x = -a.lat.off_x, start_y = -a.lat.off_y;
        for (int i = threadIdx.x; i < a.lat.nx; i += 256) {
            const double x = start_x + i * a.lat.step_x;
            cx[i] = world_to_grid(st.center[0] + x, st.off_x, a.g.scale) + a.g.border - a.g.win_origin;
        }
        for (int i = threadIdx.x; i < a.lat.ny; i += 256) {
            const double y = start_y + i * a.lat.step_y;
            cy[i] = world_to_grid(st.center[1] + y, st.off_y, a.g.scale) + a.g.border - a.g.win_origin;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int sx = (int)kt_round(a.lat.step_x * a.g.scale), sy = (int)kt_round(a.lat.step_y * a.g.scale);
            int reg = 1;
            for (int i = 1; i < a.lat.nx; i++) reg &= (cx[i] == cx[0] + i * sx);
            for (int i = 1; i < a.lat.ny; i++) reg &= (cy[i] == cy[0] + i * sy);
            a.states[b].regular[a.lat.fine] = reg;
        }
        return;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= st.nq) return;
    const double start_angle = st.center[2] - a.lat.angle_off;
    const double angle = start_angle + k * a.lat.angle_res;
    const double cosine = cos(angle), sine = sin(angle);
    const double2 p = a.qlocal[(size_t)b * a.max_n + i];
    const double ox = cosine * p.x - sine * p.y;
    const double oy = sine * p.x + cosine * p.y;
    const int gx = world_to_grid(ox + st.off_x, st.off_x, a.g.scale);
    const int gy = world_to_grid(oy + st.off_y, st.off_y, a.g.scale);
    a.offsets[((size_t)b * a.nt_stride + k) * a.max_n + i] = gx + gy * a.g.pitch;
}

// ================================================================== K4 correlate (generic path)
struct CorrArgs {
    YmGeom g;
    YmLattice lat;
    const uint8_t *grid;
    size_t grid_stride;
    const int32_t *offsets;
    const int32_t *hypcell;
    const YmItemState *states;
    uint32_t *sums;     // [B][nt][ny][nx]
    size_t sums_stride; // per item
    int32_t max_n, nt_stride, dim_stride;
};

// One thread per hypothesis, x fastest so a wave reads one short row segment per beam.
// grid (ceil(nx*ny / 256), nt, B).  ScanMatcher::GetResponse, integer part.
__global__ __launch_bounds__(256) void correlate_generic_kernel(CorrArgs a) {
    const int b = blockIdx.z, k = blockIdx.y;
    const int h = blockIdx.x * 256 + threadIdx.x;
    const int nxy = a.lat.nx * a.lat.ny;
    if (h >= nxy) return;
    const int iy = h / a.lat.nx, ix = h - iy * a.lat.nx;
    const int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
    const int32_t *cy = cx + a.dim_stride;
    const int base = cy[iy] * a.g.pitch + cx[ix];
    const int nq = a.states[b].nq;
    const int32_t *__restrict__ offs = a.offsets + ((size_t)b * a.nt_stride + k) * a.max_n;
    const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
    const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
    unsigned sum = 0;
    int i = 0;
    for (; i + 8 <= nq; i += 8) {
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned idx = (unsigned)(base + offs[i + u]);
            v[u] = idx < limit ? grid[idx] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) sum += v[u];
    }
    for (; i < nq; i++) {
        const unsigned idx = (unsigned)(base + offs[i]);
        sum += idx < limit ? grid[idx] : 0u;
    }
    a.sums[(size_t)b * a.sums_stride + ((size_t)k * a.lat.ny + iy) * a.lat.nx + ix] = sum;
}

// ================================================================== K5 reduce
struct ReduceArgs {
    YmGeom g;
    YmLattice lat;
    const uint32_t *sums;
    size_t sums_stride;
    YmItemState *states;
    double *probs;        // [B][ny*nx] scratch: max over theta per (x, y)   (m_pSearchSpaceProbs)
    size_t probs_stride;
    const uint8_t *grid;
    size_t grid_stride;
    const int32_t *offsets;
    int32_t max_n, nt_stride;
};

__device__ __forceinline__ double hyp_response(const ReduceArgs &a, unsigned sum, int nq, double sq_dist,
                                               double angle, double center_t) {
    double response = 0.0;
    if (nq != 0) {
        response = (double)sum;
        response /= (double)(nq * YM_OCCUPIED);
    }
    if (a.lat.penalize && !kt_double_equal(response, 0.0)) {
        double dp = 1.0 - (YM_PENALTY_GAIN * sq_dist / a.g.dist_var);
        dp = dp > a.g.min_dist_pen ? dp : a.g.min_dist_pen;
        const double sq_ang = (angle - center_t) * (angle - center_t);
        double ap = 1.0 - (YM_PENALTY_GAIN * sq_ang / a.g.ang_var);
        ap = ap > a.g.min_ang_pen ? ap : a.g.min_ang_pen;
        response *= (dp * ap);
    }
    return response;
}

// grid (B), 1024 threads.
__global__ __launch_bounds__(1024) void reduce_kernel(ReduceArgs a) {
    __shared__ double scratch[16];
    __shared__ unsigned uscratch[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    YmItemState &st = a.states[b];
    const int nx = a.lat.nx, ny = a.lat.ny, nt = a.lat.nt, nxy = nx * ny;
    const int nq = st.nq;
    const double cxw = st.center[0], cyw = st.center[1], ct = st.center[2];
    const double start_x = -a.lat.off_x, start_y = -a.lat.off_y;
    const double start_angle = ct - a.lat.angle_off;
    const uint32_t *sums = a.sums + (size_t)b * a.sums_stride;
    double *probs = a.probs + (size_t)b * a.probs_stride;

    if (nq == 0) {
        // MatchScan: "scan has no readings; cannot do scan matching" -> pose, maximum covariance, 0
        if (tid == 0 && !a.lat.fine) {
            for (int i = 0; i < 9; i++) st.cov[i] = 0.0;
            st.cov[0] = YM_MAX_VARIANCE; st.cov[4] = YM_MAX_VARIANCE;
            st.cov[8] = 4 * (a.lat.angle_res * a.lat.angle_res);
            for (int i = 0; i < 3; i++) { st.mean[i] = st.pose[i]; st.center[i] = st.pose[i]; }
            st.response = 0.0;
            st.coarse_response = 1.0; // not a "zero response" in the expansion sense: nothing to retry
        }
        return;
    }

    // pass A: best response; per-(x,y) max over theta
    double lbest = -1.0;
    for (int c = tid; c < nxy; c += 1024) {
        const int iy = c / nx, ix = c - iy * nx;
        const double x = start_x + ix * a.lat.step_x, y = start_y + iy * a.lat.step_y;
        const double sq = x * x + y * y;
        double cm = -1.0;
        for (int k = 0; k < nt; k++) {
            const double angle = start_angle + k * a.lat.angle_res;
            const double r = hyp_response(a, sums[(size_t)k * nxy + c], nq, sq, angle, ct);
            cm = r > cm ? r : cm;
        }
        probs[c] = cm > 0.0 ? cm : 0.0; // the probability grid starts cleared to 0
        lbest = cm > lbest ? cm : lbest;
    }
    const double best = block_reduce(lbest, OpMaxD(), -1.0, scratch);

    // pass B: mean of all hypotheses with DoubleEqual(response, best)
    double ax = 0, ay = 0, tx = 0, ty = 0;
    int cnt = 0;
    for (int c = tid; c < nxy; c += 1024) {
        const int iy = c / nx, ix = c - iy * nx;
        const double x = start_x + ix * a.lat.step_x, y = start_y + iy * a.lat.step_y;
        const double sq = x * x + y * y;
        for (int k = 0; k < nt; k++) {
            const double angle = start_angle + k * a.lat.angle_res;
            const double r = hyp_response(a, sums[(size_t)k * nxy + c], nq, sq, angle, ct);
            if (kt_double_equal(r, best)) {
                const double hd = kt_normalize_angle(angle);
                ax += cxw + x; ay += cyw + y;
                tx += cos(hd); ty += sin(hd);
                cnt++;
            }
        }
    }
    ax = block_reduce(ax, OpAddD(), 0.0, scratch);
    ay = block_reduce(ay, OpAddD(), 0.0, scratch);
    tx = block_reduce(tx, OpAddD(), 0.0, scratch);
    ty = block_reduce(ty, OpAddD(), 0.0, scratch);
    {
        __shared__ int iscratch[16];
        cnt = block_reduce(cnt, OpAddI(), 0, iscratch);
    }
    double mean[3] = {0, 0, 0};
    int status = st.status;
    if (cnt > 0) {
        ax /= cnt; ay /= cnt; tx /= cnt; ty /= cnt;
        mean[0] = ax; mean[1] = ay; mean[2] = atan2(ty, tx);
    } else {
        status = -5; // "Unable to find best position"
    }

    if (!a.lat.fine) {
        // ComputePositionalCovariance
        double norm = 0, axx = 0, axy = 0, ayy = 0;
        const double dx = mean[0] - cxw, dy = mean[1] - cyw;
        if (!(best < YM_KT_TOLERANCE)) {
            for (int c = tid; c < nxy; c += 1024) {
                const int iy = c / nx, ix = c - iy * nx;
                const double x = start_x + ix * a.lat.step_x, y = start_y + iy * a.lat.step_y;
                const double response = probs[c];
                if (response >= (best - 0.1)) {
                    norm += response;
                    axx += ((x - dx) * (x - dx)) * response;
                    axy += ((x - dx) * (y - dy) * response);
                    ayy += ((y - dy) * (y - dy)) * response;
                }
            }
        }
        norm = block_reduce(norm, OpAddD(), 0.0, scratch);
        axx = block_reduce(axx, OpAddD(), 0.0, scratch);
        axy = block_reduce(axy, OpAddD(), 0.0, scratch);
        ayy = block_reduce(ayy, OpAddD(), 0.0, scratch);
        if (tid == 0) {
            double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            if (best < YM_KT_TOLERANCE) {
                cov[0] = YM_MAX_VARIANCE; cov[4] = YM_MAX_VARIANCE;
                cov[8] = 4 * (a.lat.angle_res * a.lat.angle_res);
            } else {
                if (norm > YM_KT_TOLERANCE) {
                    double vxx = axx / norm, vxy = axy / norm, vyy = ayy / norm;
                    const double vthth = 4 * (a.lat.angle_res * a.lat.angle_res);
                    const double min_xx = 0.1 * (a.lat.step_x * a.lat.step_x);
                    const double min_yy = 0.1 * (a.lat.step_y * a.lat.step_y);
                    vxx = vxx > min_xx ? vxx : min_xx;
                    vyy = vyy > min_yy ? vyy : min_yy;
                    const double mult = 1.0 / best;
                    cov[0] = vxx * mult; cov[1] = vxy * mult; cov[3] = vxy * mult; cov[4] = vyy * mult;
                    cov[8] = vthth;
                }
                if (kt_double_equal(cov[0], 0.0)) cov[0] = YM_MAX_VARIANCE;
                if (kt_double_equal(cov[4], 0.0)) cov[4] = YM_MAX_VARIANCE;
            }
            for (int i = 0; i < 9; i++) st.cov[i] = cov[i];
        }
    } else {
        // ComputeAngularCovariance: re-score every fine angle at the cell of the mean pose
        const double best_angle = kt_normalize_angle_difference(mean[2], ct);
        const int gx = world_to_grid(mean[0], st.off_x, a.g.scale) + a.g.border - a.g.win_origin;
        const int gy = world_to_grid(mean[1], st.off_y, a.g.scale) + a.g.border - a.g.win_origin;
        const int base = gy * a.g.pitch + gx;
        const uint8_t *grid = a.grid + (size_t)b * a.grid_stride;
        const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
        double norm = 0.0, acc = 0.0;
        for (int k = 0; k < nt; k++) {
            const int32_t *offs = a.offsets + ((size_t)b * a.nt_stride + k) * a.max_n;
            unsigned s = 0;
            for (int i = tid; i < nq; i += 1024) {
                const unsigned idx = (unsigned)(base + offs[i]);
                s += idx < limit ? grid[idx] : 0u;
            }
            s = block_reduce(s, OpAddU(), 0u, uscratch);
            const double angle = start_angle + k * a.lat.angle_res;
            double response = 0.0;
            if (nq != 0) { response = (double)s; response /= (double)(nq * YM_OCCUPIED); }
            if (response >= (best - 0.1)) {
                norm += response;
                acc += ((angle - best_angle) * (angle - best_angle)) * response;
            }
        }
        if (tid == 0) {
            if (norm > YM_KT_TOLERANCE) {
                if (acc < YM_KT_TOLERANCE) acc = a.lat.angle_res * a.lat.angle_res;
                acc /= norm;
            } else {
                acc = 1000 * (a.lat.angle_res * a.lat.angle_res);
            }
            st.cov[8] = acc;
        }
    }
    if (tid == 0) {
        st.mean[0] = mean[0]; st.mean[1] = mean[1]; st.mean[2] = mean[2];
        st.center[0] = mean[0]; st.center[1] = mean[1]; st.center[2] = mean[2]; // centre of the fine pass
        const double clamped = best > 1.0 ? 1.0 : best;
        st.response = clamped;
        if (!a.lat.fine) st.coarse_response = clamped;
        st.status = status;
    }
}

}  // namespace ym
