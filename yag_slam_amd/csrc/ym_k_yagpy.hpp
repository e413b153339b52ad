// ym_k_yagpy.hpp -- the reference's Python matcher ("yagpy" semantics): yag_setup / yag_score / yag_reduce kernels.
// Part of ym_kernels.hpp (include that, not this file).
#pragma once

namespace ym {

// ================================================================== "yagpy" semantics
// The reference's in-tree Python matcher (/root/reference/yag_slam/helpers.py:156-295 find_best_pose,
// scan_matching.py:175-222).  Unlike Karto it rounds every (hypothesis, point) pair separately
// (helpers.py:149-153), so the gather address is a function of the pair.  yag_score_kernel recomputes it in fp64 per pair
// (the rule as written).  Round 6: the COARSE pass's integer sums come from the production correlate kernels
// (correlate_kernel, correlate_region_kernel, gather_kernel) wherever yag_lattice_kernel can PROVE that the item's roundings
// form a lattice -- every hot correlate kernel is then checked bit for bit against sum volumes the reference's own
// score_world_points_on_grid produced (tests/golden/*.npz: coarse_sums) -- and from yag_score_kernel for the items it cannot
// (counted: ym_debug_counter).  Arg-max, tie mean and covariances stay yag_reduce_kernel's: the Python path's own rules.
#define YM_YAG_FINE_DIM 5 // positions per axis of the fine pass that yag_fine_kernel keeps in registers (numpy.arange(-2 res, 2 res, res): 4 or 5)
struct YagArgs {
    YmGeom g;
    int32_t pass;        // 0 coarse, 1 fine
    int32_t penalize;
    int32_t last;        // this pass produces the final result
    int32_t refine;
    double search_xy, step_xy, search_t, step_t; // find_best_pose arguments
    double coarse_angle_res;                     // for th = 4*angle_res when the fine pass is skipped
    YmItemState *states;
    YmItemState *host_out;
    const double2 *qlocal;
    double *axes;        // [B][3][YM_YAG_MAX_DIM] xvals, yvals, tvals
    double2 *rot;        // [B][maxt][max_n] points rotated by tvals[k] (prebuilt maps); nullptr: the kernels rotate the points themselves
                         // (the same two products and one sum per coordinate: 0.9 GB less to write and read per pass of 4096 matches)
    uint32_t *sums;      // [B][maxt][maxd][maxd] -> stored dense as [k][iy][ix] with the pass's nx, ny
    double *out;         // same shape, fp64 scores
    const uint8_t *grid;
    size_t grid_stride;
    size_t vol_stride;   // entries per item in sums/out
    int32_t max_n, maxd, maxt;
    // match against a prebuilt map (scan_matching.py:124-173, find_best_pose_non_symmetric helpers.py:434-573):
    // map_w > 0 selects it.  The grid is the whole map (map_w x map_h cells, row pitch map_w) with corner (map_ox,
    // map_oy); the cell size used for indexing AND in the penalty is the pass's own argument (the reference passes 0.05
    // to the coarse pass whatever the map's resolution); the penalty is centred on the search centre.
    int32_t map_w, map_h;
    double map_ox, map_oy, map_res;
    // the coarse pass through the production correlate kernels (pass 0, lat_nx > 0): the lattice those kernels are launched on
    // (np.arange lengths vary from item to item by one: the launch lattice holds every item's, the surplus hypotheses are computed and
    // never read), its cell step, the tables yag_lattice_kernel fills for them and the sums they leave
    int32_t lat_nx, lat_ny, lat_nt, step_cells;
    int32_t nt_stride, dim_stride;
    double2 *ctrig;          // [B][nt_stride]
    int32_t *hypcell;        // [B][2][dim_stride]
    const uint32_t *lsums;   // [B][lat_nt][lat_ny][lat_nx] integer sums of the launch lattice (score_kernel / score_hyp_kernel / gather_kernel)
    size_t lsums_stride;
    int32_t fine_rows, n_items;     // pass 1: yag_fine_kernel scores the items whose fine lattice is at most YM_YAG_FINE_DIM wide (all of them)
    unsigned long long *counters; // [0] items whose coarse pass went through the production kernels, [1] items that fell back to yag_score_kernel,
                                  // [2] (point, angle) pairs that needed the exhaustive check, [3] pairs that failed it
};

// numpy.arange(start, stop, step) for float64: length and i-th value (DOUBLE_fill)
__device__ __forceinline__ int yag_arange_len(double start, double stop, double step) {
    const int n = (int)ceil((stop - start) / step);
    return n < 0 ? 0 : n;
}
__device__ __forceinline__ double yag_arange_at(double start, double step, int i) {
    if (i == 0) return start;
    const double second = start + step;
    if (i == 1) return second;
    return start + i * (second - start);
}

// helpers.py:76-78 _rotate_points
__device__ __forceinline__ double2 yag_rotate(double2 p, double c, double s) { return make_double2(p.x * c - p.y * s, p.y * c + p.x * s); }

// grid (maxt, B), 256 threads: block k rotates the points by tvals[k]; block 0 also writes the axes.  (rot == nullptr: grid (1, B), the axes only.)
__global__ __launch_bounds__(256) void yag_setup_kernel(YagArgs a) {
    const int b = blockIdx.y, k = blockIdx.x, tid = threadIdx.x;
    YmItemState &st = a.states[b];
    const double cx = a.pass ? st.ybest[0][1] : st.pose[0];
    const double cy = a.pass ? st.ybest[0][2] : st.pose[1];
    const double ct = a.pass ? st.ybest[0][3] : st.pose[2];
    const int nx = min(yag_arange_len(-a.search_xy + cx, a.search_xy + cx, a.step_xy), a.maxd);
    const int ny = min(yag_arange_len(-a.search_xy + cy, a.search_xy + cy, a.step_xy), a.maxd);
    const int nt = min(yag_arange_len(-a.search_t + ct, a.search_t + ct, a.step_t), a.maxt);
    double *ax = a.axes + (size_t)b * 3 * YM_YAG_MAX_DIM;
    if (k == 0) {
        if (tid == 0) { st.ydims[a.pass][0] = nx; st.ydims[a.pass][1] = ny; st.ydims[a.pass][2] = nt; }
        for (int i = tid; i < nx; i += 256) ax[i] = yag_arange_at(-a.search_xy + cx, a.step_xy, i);
        for (int i = tid; i < ny; i += 256) ax[YM_YAG_MAX_DIM + i] = yag_arange_at(-a.search_xy + cy, a.step_xy, i);
        for (int i = tid; i < nt; i += 256) ax[2 * YM_YAG_MAX_DIM + i] = yag_arange_at(-a.search_t + ct, a.step_t, i);
    }
    if (k >= nt || !a.rot) return;
    const double t = yag_arange_at(-a.search_t + ct, a.step_t, k);
    const double c = cos(t), s = sin(t);
    const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
    double2 *rot = a.rot + ((size_t)b * a.maxt + k) * a.max_n;
    for (int l = tid; l < st.nq; l += 256) rot[l] = yag_rotate(ql[l], c, s);
}

// When do the roundings of the coarse pass form a lattice?  (helpers.py:149-153 with x = xvals[i] + xx, helpers.py:194-196.)
// For a (point, angle) pair with rotated coordinate r, hypothesis i of an axis reads cell
//     c(i) = rint(u(i)),   u(i) = fl(fl(fl(xv[i] + r) - o) / res),   xv[i] = numpy.arange's i-th value (yag_arange_at).
// The production correlate kernels need c(i) = c(0) + i * s for every i of the lattice (s = step / res cells, an integer: 2).
// With e = 2^-53 and M >= every magnitude that occurs (|xv[i]|, |xv[i] + r|, |x - o|, the search size):
//     xv[i] = start + i * step + delta_i, |delta_i| <= (2 i + 2) e M   (yag_arange_at: second = fl(start + step), d = fl(second - start),
//                                                                       xv[i] = fl(start + fl(i * d)): four roundings, two of them i times)
//     u(i)  = (xv[i] + r - o) / res up to three more roundings: e M / res each,
// and step / res = 2 exactly (step is the doubled resolution), so |u(i) - (u(0) + i * s)| <= (2 n + 8) e M / res for every i < n.
// Hence |u(0) - rint(u(0))| < 0.5 - guard, guard >= that bound, proves the pair for all i at once (no u(i) can reach a tie).  A pair
// that fails the test is not irregular yet: it is then checked hypothesis by hypothesis (x and y are separable: nx + ny roundings).
// Only a pair that fails THAT -- a genuine tie that falls differently along the axis -- makes its item irregular, and the item is
// scored by yag_score_kernel as before.  The guard used is 8 (n + 8) e M / res + 2^-40: about four times the bound.
// M = |ox| + |oy| + 2 G res: the search box lies within G res / 2 of the query pose = ox + (G - 1) res / 2, and a read inside the
// window (tested first) has |x - o| <= G res.
// Every read must also stay inside the device window for every hypothesis of the LAUNCH lattice (the production kernels test no
// bounds; cells outside the window are provably empty, DESIGN.md section 3, but their memory is another item's).
// grid (B), 256 threads.  Runs after yag_setup_kernel (axes, ydims, rot) of pass 0.
__global__ __launch_bounds__(256) void yag_lattice_kernel(YagArgs a) {
    constexpr int NT = 256;
    const int b = blockIdx.x, tid = threadIdx.x;
    YmItemState &st = a.states[b];
    const int nx = st.ydims[0][0], ny = st.ydims[0][1], nt = st.ydims[0][2], nq = st.nq;
    const double *ax = a.axes + (size_t)b * 3 * YM_YAG_MAX_DIM;
    const double xv0 = ax[0], yv0 = ax[YM_YAG_MAX_DIM];
    const double ct = st.pose[2];
    // (cos, sin) per angle of the launch lattice: the operands the Python rule rotates the points with (tvals[k] = numpy.arange's k-th value)
    __shared__ double2 s_trig[YM_YAG_MAX_NT];
    for (int k = tid; k < a.lat_nt; k += NT) {
        const double t = yag_arange_at(-a.search_t + ct, a.step_t, k);
        const double2 cs = make_double2(cos(t), sin(t));
        a.ctrig[(size_t)b * a.nt_stride + k] = cs;
        if (k < YM_YAG_MAX_NT) s_trig[k] = cs;
    }
    __syncthreads();
    const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
    int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
    int32_t *cy = cx + a.dim_stride;
    for (int i = tid; i < a.lat_nx; i += NT) cx[i] = i * a.step_cells;
    for (int i = tid; i < a.lat_ny; i += NT) cy[i] = i * a.step_cells;
    int ok = nx > 0 && ny > 0 && nt > 0 && nx <= a.lat_nx && ny <= a.lat_ny && nt <= a.lat_nt && nt <= YM_YAG_MAX_NT;
    const double ox = st.off_x, oy = st.off_y, res = a.g.res;
    const double M = fabs(ox) + fabs(oy) + 2.0 * a.g.roi_w * res;
    const double guard = 8.0 * (max(nx, ny) + 8) * M * 1.1102230246251565e-16 / res + 9.094947017729282e-13;
    const int w0 = a.g.win_origin, ww = a.g.win_w;
    const int span_x = (a.lat_nx - 1) * a.step_cells, span_y = (a.lat_ny - 1) * a.step_cells;
    unsigned slow = 0, bad = 0;
    if (ok)
        for (int p = tid; p < nq * nt; p += NT) {
            const int k = p / nq, l = p - k * nq;
            const double2 r = a.rot ? a.rot[((size_t)b * a.maxt + k) * a.max_n + l] : yag_rotate(ql[l], s_trig[k].x, s_trig[k].y);
            const double ux = ((xv0 + r.x) - ox) / res, uy = ((yv0 + r.y) - oy) / res;
            const double gx = rint(ux), gy = rint(uy);
            // the window test in fp64 first: a cell number beyond int would be undefined behaviour below
            if (!(gx >= (double)w0 && gx + span_x < (double)(w0 + ww) && gy >= (double)w0 && gy + span_y < (double)(w0 + ww))) { ok = 0; bad++; continue; }
            const bool fast_x = fabs(ux - gx) < 0.5 - guard, fast_y = fabs(uy - gy) < 0.5 - guard;
            if (fast_x && fast_y) continue;
            slow++;
            bool good = true;
            if (!fast_x)
                for (int i = 1; i < nx; i++) good = good && rint(((ax[i] + r.x) - ox) / res) == gx + (double)(i * a.step_cells);
            if (!fast_y)
                for (int i = 1; i < ny; i++) good = good && rint(((ax[YM_YAG_MAX_DIM + i] + r.y) - oy) / res) == gy + (double)(i * a.step_cells);
            if (!good) { ok = 0; bad++; }
        }
    ok = __syncthreads_and(ok);
    if (a.counters) {
        if (slow) atomicAdd(&a.counters[2], (unsigned long long)slow);
        if (bad) atomicAdd(&a.counters[3], (unsigned long long)bad);
    }
    if (tid == 0) {
        st.regular[0] = ok;
        st.ylat[0] = xv0; st.ylat[1] = yv0;
        if (a.counters) atomicAdd(&a.counters[ok ? 0 : 1], 1ull);
    }
}

// score of one hypothesis from its integer sum (helpers.py:165-196: the mean of int(100 * cell) over the points, times the penalties), stored
__device__ __forceinline__ void yag_store_score(const YagArgs &a, const YmItemState &st, int b, int k, int iy, int ix, int nx, int ny, unsigned sum, int np,
                                                double xv, double yv, double tv, double ox, double oy, double res, int GW, bool map) {
    double penalty_val = 1.0;
    if (a.penalize) {
        const double ct = a.pass ? st.ybest[0][3] : st.pose[2];
        // helpers.py:173-174: the grid's centre; find_best_pose_non_symmetric (helpers.py:451-452): the search centre
        const double sx_ = map ? (a.pass ? st.ybest[0][1] : st.pose[0]) : ox + GW * res / 2;
        const double sy_ = map ? (a.pass ? st.ybest[0][2] : st.pose[1]) : oy + GW * res / 2;
        const double sd = (xv - sx_) * (xv - sx_) + (yv - sy_) * (yv - sy_);
        const double dist_penalty = 1.0 - 0.2 * sd / (0.5 * res);
        const double sa = (tv - ct) * (tv - ct);
        const double ang_penalty = 1.0 - 0.2 * sa / (1.0 * res);
        penalty_val = dist_penalty * ang_penalty;
    }
    const size_t at = (size_t)b * a.vol_stride + ((size_t)k * ny + iy) * nx + ix;
    a.sums[at] = sum;
    a.out[at] = (double)sum / np * penalty_val / 100.0;
}

// grid (ceil(maxd*maxd/256), maxt, B): one thread per hypothesis (ix, iy) of angle k.
// helpers.py:134-153: per point rint((p - o)/res), bounds check, int(100*cell) accumulate.
// (pass 0 with lsums: the items yag_lattice_kernel proved regular take their sums from the production correlate kernels' volume.)
__global__ __launch_bounds__(256) void yag_score_kernel(YagArgs a) {
    const int b = blockIdx.z, k = blockIdx.y;
    const YmItemState &st = a.states[b];
    const int nx = st.ydims[a.pass][0], ny = st.ydims[a.pass][1], nt = st.ydims[a.pass][2];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (k >= nt || c >= nx * ny) return;
    if (a.fine_rows && nx <= YM_YAG_FINE_DIM && ny <= YM_YAG_FINE_DIM) return; // (pass 1: yag_fine_kernel has scored this item)
    const int iy = c / nx, ix = c - iy * nx;
    const double *ax = a.axes + (size_t)b * 3 * YM_YAG_MAX_DIM;
    const double xv = ax[ix], yv = ax[YM_YAG_MAX_DIM + iy], tv = ax[2 * YM_YAG_MAX_DIM + k];
    const bool map = a.map_w > 0;
    const double ox = map ? a.map_ox : st.off_x, oy = map ? a.map_oy : st.off_y, res = map ? a.map_res : a.g.res;
    const int GW = map ? a.map_w : a.g.roi_w, GH = map ? a.map_h : a.g.roi_w;
    const int w0 = map ? 0 : a.g.win_origin, ww = map ? a.map_w : a.g.win_w, wh = map ? a.map_h : a.g.win_w;
    const int pitch = map ? a.map_w : a.g.pitch;
    const double2 *__restrict__ rot = a.rot ? a.rot + ((size_t)b * a.maxt + k) * a.max_n : nullptr;
    const double2 *__restrict__ ql = reinterpret_cast<const double2 *>(st.ql);
    const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
    const int np = st.nq;
    unsigned sum = 0;
    if (a.lsums && st.regular[0]) sum = a.lsums[(size_t)b * a.lsums_stride + ((size_t)k * a.lat_ny + iy) * a.lat_nx + ix];
    else {
    const double rc = rot ? 0.0 : cos(tv), rs = rot ? 0.0 : sin(tv);
#pragma unroll 4
    for (int l = 0; l < np; l++) {
        const double2 p = rot ? rot[l] : yag_rotate(ql[l], rc, rs);
        const double x = xv + p.x, y = yv + p.y;
        const double gx = rint((x - ox) / res), gy = rint((y - oy) / res);
        const int _x = (int)gx, _y = (int)gy;
        if (_x >= 0 && _x < GW && _y >= 0 && _y < GH) {
            const int wx = _x - w0, wy = _y - w0;
            // cells outside the device window are provably empty
            if (wx >= 0 && wx < ww && wy >= 0 && wy < wh) sum += grid[(size_t)wy * pitch + wx];
        }
    }
    }
    yag_store_score(a, st, b, k, iy, ix, nx, ny, sum, np, xv, yv, tv, ox, oy, res, GW, map);
}

// rint(d / res) without the division wherever that is provably the same integer.  With rres = fl(1 / res): q = fl(d * rres) and the
// quotient fl(d / res) each differ from the real d / res by at most two roundings resp. one, i.e. from each other by less than
// 4 * 2^-53 |q|; so whenever q is farther than 2^-50 |q| + 1e-12 from a tie, both round to the integer n = rint(q).  Otherwise (and for
// a q so large that the margin exceeds one half, or a NaN) the division decides.  An fp64 division is ~15 instructions, one of them
// (v_rcp_f64) at a quarter of the rate; yag_fine_kernel makes ten per (point, angle) pair.
__device__ __forceinline__ double yag_rint_div(double d, double res, double rres) {
    const double q = d * rres, n = rint(q);
    if (fabs(q - n) < 0.5 - (fabs(q) * 8.881784197001252e-16 + 1e-12)) return n;
    return rint(d / res);
}

// The fine pass (pass 1: search +-2 cells at a step of one cell -- numpy.arange gives 4 or 5 positions per axis), exactly and without a
// proof.  The cell hypothesis (ix, iy) reads for a point is (rint(((xv[ix] + r.x) - ox) / res), rint(((yv[iy] + r.y) - oy) / res)): its
// column depends on ix alone and its row on iy alone (helpers.py:149-153), so a (point, angle) pair costs nx + ny roundings -- the same
// operations on the same operands as yag_score_kernel's, 2 nx ny of them there -- and its nx ny reads are ny rows of nx neighbouring
// bytes: one 8-byte read per row (two aligned dwords and a shift; a row whose columns do not fit the eight bytes -- a tie that falls
// to the far side next to a misaligned start -- is read byte by byte).  yag_score_kernel, with a thread per hypothesis, runs this
// pass on 25 lanes per (item, angle): 7 of the 12 ms of an enqueue of 4096 matches.  Here: 256 threads per (item, angle); the block
// walks the points of item b at fine angle k, a thread keeps its 25 sums in registers, the block adds them up and scores them.
// Items with a wider fine lattice (none: the search is the reference's constant) are left to yag_score_kernel.
// NT = 256: a single match (ten blocks, shortest chain); NT = 64 on batches: a block's fixed part -- item state, axes, cos and sin, the
// final sums: dependent loads -- is paid once per 17 points of a lane instead of once per 4, no barrier, no idle last round (755 -> 681 us;
// the rest is the vector L1's: five scattered 8-byte reads per lane and point, a line visit each).
template <int NT>
__global__ __launch_bounds__(NT) void yag_fine_kernel(YagArgs a) {
    constexpr int D = YM_YAG_FINE_DIM;
    __shared__ unsigned s_part[NT / 64][D * D];
    // grid (8 * maxt * ceil(B / 8)): the hardware deals consecutive blocks to the eight XCDs in turn; block id = 8 * j + x goes to XCD x
    // and takes item 8 * (j / maxt) + x at angle j % maxt, so the angles of an item -- which read the same rows of its window --
    // share one L2 (with a block (k, b) grid they spread over all eight and every L2 fetched the rows for itself: 816 -> 755 us per 4096 items)
    const int xcd = blockIdx.x & 7, j_ = blockIdx.x >> 3;
    const int b = (j_ / a.maxt) * 8 + xcd, k = j_ % a.maxt, tid = threadIdx.x;
    if (b >= a.n_items) return;
    const YmItemState &st = a.states[b];
    const int nx = st.ydims[1][0], ny = st.ydims[1][1], nt = st.ydims[1][2];
    if (k >= nt || nx > D || ny > D || nx * ny == 0) return; // (block-uniform)
    const double *ax = a.axes + (size_t)b * 3 * YM_YAG_MAX_DIM;
    double xv[D], yv[D];
#pragma unroll
    for (int i = 0; i < D; i++) { xv[i] = i < nx ? ax[i] : 0.0; yv[i] = i < ny ? ax[YM_YAG_MAX_DIM + i] : 0.0; }
    const double ox = st.off_x, oy = st.off_y, res = a.g.res;
    const int GW = a.g.roi_w, w0 = a.g.win_origin, ww = a.g.win_w, pitch = a.g.pitch;
    const double2 *__restrict__ rot = a.rot ? a.rot + ((size_t)b * a.maxt + k) * a.max_n : nullptr;
    const double2 *__restrict__ ql = reinterpret_cast<const double2 *>(st.ql);
    const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
    const int np = st.nq;
    __shared__ double2 s_cs;
    if (tid == 0 && !rot) { const double t = ax[2 * YM_YAG_MAX_DIM + k]; s_cs = make_double2(cos(t), sin(t)); } // (tvals[k], written by yag_setup_kernel)
    __syncthreads(); // (one wave: no barrier instruction, the LDS write is waited for)
    const double rc = rot ? 0.0 : s_cs.x, rs = rot ? 0.0 : s_cs.y;
    const double rres = 1.0 / res;
    const unsigned gmis = (unsigned)(reinterpret_cast<uintptr_t>(grid) & 3u); // (0: an item's window starts at a multiple of 256 bytes)
    unsigned sum[D * D];
#pragma unroll
    for (int h = 0; h < D * D; h++) sum[h] = 0u;
    for (int l = tid; l < np; l += NT) {
        const double2 p = rot ? rot[l] : yag_rotate(ql[l], rc, rs);
        int wx[D], wy[D]; // window column / row of hypothesis i, -1: outside the grid or the window (such a read adds nothing)
        int first = -1, span = 0;
#pragma unroll
        for (int i = 0; i < D; i++) {
            const double x = xv[i] + p.x, y = yv[i] + p.y;
            const double gx = yag_rint_div(x - ox, res, rres), gy = yag_rint_div(y - oy, res, rres);
            const int _x = (int)gx, _y = (int)gy;
            const int cx = _x - w0, cy = _y - w0;
            wx[i] = (i < nx && _x >= 0 && _x < GW && cx >= 0 && cx < ww) ? cx : -1;
            wy[i] = (i < ny && _y >= 0 && _y < GW && cy >= 0 && cy < ww) ? cy : -1;
            if (wx[i] >= 0) { // (columns do not decrease with i: the first one inside is the lowest)
                if (first < 0) first = wx[i];
                span = wx[i] - first;
            }
        }
        if (first < 0) continue;
#pragma unroll
        for (int j = 0; j < D; j++) {
            if (wy[j] < 0) continue;
            // (32-bit arithmetic throughout: a window is at most 4096 x 4096 bytes; 64-bit shifts and products run at a quarter of the rate)
            const unsigned off = (unsigned)wy[j] * (unsigned)pitch + (unsigned)first;
            const uint8_t *row = grid + ((unsigned)wy[j] * (unsigned)pitch);
            const unsigned mis = (gmis + off) & 3u;
            if (span + (int)mis <= 7 && a.fine_rows != 2) { // (fine_rows = 2: tests, every row byte by byte)
                const uint32_t *w = reinterpret_cast<const uint32_t *>(grid + (off - mis));
                const unsigned w0_ = w[0], w1_ = w[1];
#pragma unroll
                for (int i = 0; i < D; i++)
                    if (wx[i] >= 0) {
                        const unsigned o = mis + (unsigned)(wx[i] - first); // byte o of the eight
                        sum[j * D + i] += __builtin_amdgcn_ubfe(o < 4u ? w0_ : w1_, 8u * (o & 3u), 8u);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < D; i++)
                    if (wx[i] >= 0) sum[j * D + i] += row[wx[i]];
            }
        }
    }
#pragma unroll
    for (int h = 0; h < D * D; h++) sum[h] = wave_reduce(sum[h], OpAddU());
    if ((tid & 63) == 0)
#pragma unroll
        for (int h = 0; h < D * D; h++) s_part[tid >> 6][h] = sum[h];
    __syncthreads();
    if (tid < D * D) {
        const int iy = tid / D, ix = tid - iy * D;
        if (ix < nx && iy < ny) {
            unsigned total = 0;
            for (int w = 0; w < NT / 64; w++) total += s_part[w][tid];
            yag_store_score(a, st, b, k, iy, ix, nx, ny, total, np, ax[ix], ax[YM_YAG_MAX_DIM + iy], ax[2 * YM_YAG_MAX_DIM + k], ox, oy, res, GW, false);
        }
    }
}

// grid (B), NT threads: np.argmax (first maximum in the reference's [ix][iy][k] order), mean of
// all scores >= best - 1e-8, the +-5 covariance windows (helpers.py:214-295).
// NT = 1024: shortest latency for one item; NT = 256 on batches: four blocks per CU hide each other's dependent phases (the
// floating-point sums are taken in an order that does not depend on NT: block_sum_vec).
template <int NT>
__global__ __launch_bounds__(NT) void yag_reduce_kernel(YagArgs a) {
    __shared__ double scratch[16 * 5];
    __shared__ double s_v[16];
    __shared__ int s_f[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    YmItemState &st = a.states[b];
    const int nx = st.ydims[a.pass][0], ny = st.ydims[a.pass][1], nt = st.ydims[a.pass][2];
    const int nxy = nx * ny, nh = nxy * nt;
    const double *out = a.out + (size_t)b * a.vol_stride;
    const double *ax = a.axes + (size_t)b * 3 * YM_YAG_MAX_DIM;
    int status = st.status;
    if (nh == 0 || st.nq == 0) status = -1; // the reference raises (empty lattice / division by zero points)
    // ---- arg-max, ties to the lowest index in the reference's flat order f = (ix*ny + iy)*nt + k
    double bv = -INFINITY;
    int bf = 0x7fffffff;
    for (int h = tid; h < nh; h += NT) {
        const int k = h / nxy, c = h - k * nxy, iy = c / nx, ix = c - iy * nx;
        const int f = (ix * ny + iy) * nt + k;
        const double v = out[h];
        if (v > bv || (v == bv && f < bf)) { bv = v; bf = f; }
    }
    {
        const double wv = wave_reduce(bv, OpMaxD());
        const int wf = wave_reduce(bv == wv ? bf : 0x7fffffff, OpMinI());
        __syncthreads();
        if (lane == 0) { s_v[wave] = wv; s_f[wave] = wf; }
        __syncthreads();
        bv = s_v[0]; bf = s_f[0];
        for (int w = 1; w < NT / 64; w++)
            if (s_v[w] > bv || (s_v[w] == bv && s_f[w] < bf)) { bv = s_v[w]; bf = s_f[w]; }
    }
    const double response = bv;
    int ii = 0, jj = 0, kk = 0;
    if (nh > 0 && bf != 0x7fffffff) { ii = bf / (ny * nt); jj = (bf % (ny * nt)) / nt; kk = (bf % (ny * nt)) % nt; }
    // ---- mean of the near-maximal hypotheses.  The reference adds them one by one in C order of
    // out[ix][iy][k] (helpers.py:229-244) and the fine lattice's np.arange LENGTH depends on the last
    // bits of that mean, so the additions are done in exactly that order: flags in parallel, one
    // thread walks the set bits.
    __shared__ unsigned s_bits[8192];
    __shared__ unsigned long long s_words[8192 / 64]; // per 64 words of s_bits: which of them hold a flag (the walk below skips the rest:
    __shared__ double s_mean[4];                      //  the set is one or two hypotheses, the words are 196 on a 25 x 25 x 10 lattice)
    double acc[4] = {0, 0, 0, 0};
    if (nh <= 8192 * 32) {
        const int nwords = (nh + 31) / 32;
        for (int w = tid; w < nwords; w += NT) s_bits[w] = 0u;
        __syncthreads();
        for (int f = tid; f < nh; f += NT) {
            const int ix = f / (ny * nt), iy = (f % (ny * nt)) / nt, k = f % nt;
            if (out[((size_t)k * ny + iy) * nx + ix] >= response - 0.00000001) atomicOr(&s_bits[f >> 5], 1u << (f & 31));
        }
        __syncthreads();
        for (int w0 = wave * 64; w0 < nwords; w0 += NT) { // (wave-uniform loop: wave w takes the chunks w, w + NT / 64, ...)
            const unsigned long long m = __ballot(w0 + lane < nwords && s_bits[w0 + lane] != 0u);
            if (lane == 0) s_words[w0 >> 6] = m;
        }
        __syncthreads();
        if (tid == 0) {
            for (int c = 0; c < (nwords + 63) / 64; c++) {
                unsigned long long words = s_words[c];
                while (words) {
                    const int w = c * 64 + __ffsll((long long)words) - 1;
                    words &= words - 1;
                    unsigned bits = s_bits[w];
                    while (bits) {
                        const int f = w * 32 + __ffs((int)bits) - 1;
                        bits &= bits - 1;
                        const int ix = f / (ny * nt), iy = (f % (ny * nt)) / nt, k = f % nt;
                        acc[0] += ax[ix]; acc[1] += ax[YM_YAG_MAX_DIM + iy]; acc[2] += ax[2 * YM_YAG_MAX_DIM + k]; acc[3] += 1.0;
                    }
                }
            }
            for (int j = 0; j < 4; j++) s_mean[j] = acc[j];
        }
        __syncthreads();
        for (int j = 0; j < 4; j++) acc[j] = s_mean[j];
    } else {
        for (int h = tid; h < nh; h += NT)
            if (out[h] >= response - 0.00000001) {
                const int k = h / nxy, c = h - k * nxy, iy = c / nx, ix = c - iy * nx;
                acc[0] += ax[ix]; acc[1] += ax[YM_YAG_MAX_DIM + iy]; acc[2] += ax[2 * YM_YAG_MAX_DIM + k]; acc[3] += 1.0;
            }
        block_sum_vec<4>(acc, scratch);
    }
    const double bx = acc[0] / acc[3], by = acc[1] / acc[3], bt = acc[2] / acc[3];
    // ---- +-5 windows
    const int xs = max(0, ii - 5), ys = max(0, jj - 5), xe = min(nx - 1, ii + 6), ye = min(ny - 1, jj + 6);
    const int ts = max(0, kk - 5), te = min(nt - 1, kk + 6);
    double cv[5] = {0, 0, 0, 0, 0}; // XX, YY, XY, norm ; TH handled below
    const int wxn = max(0, xe - xs), wyn = max(0, ye - ys);
    for (int w = tid; w < wxn * wyn; w += NT) {
        const int i_ = xs + w / wyn, j_ = ys + w % wyn;
        const double r_ = out[((size_t)kk * ny + j_) * nx + i_];
        const double x_ = ax[i_], y_ = ax[YM_YAG_MAX_DIM + j_];
        cv[3] += r_;
        cv[0] += r_ * ((x_ - bx) * (x_ - bx));
        cv[1] += r_ * ((y_ - by) * (y_ - by));
        cv[2] += (x_ - bx) * (y_ - by) * r_;
    }
    double tw[2] = {0, 0};
    for (int k_ = ts + tid; k_ < te; k_ += NT) {
        const double r_ = out[((size_t)k_ * ny + jj) * nx + ii];
        const double t_ = ax[2 * YM_YAG_MAX_DIM + k_];
        tw[1] += r_;
        tw[0] += r_ * ((t_ - bt) * (t_ - bt));
    }
    cv[4] = tw[0];
    block_sum_vec<5>(cv, scratch);
    double thn[1] = {tw[1]};
    block_sum_vec<1>(thn, scratch);
    if (tid == 0) {
        double *o = st.ybest[a.pass];
        o[0] = response; o[1] = bx; o[2] = by; o[3] = bt;
        o[4] = cv[0] / cv[3] / response; o[5] = cv[1] / cv[3] / response; o[6] = cv[2] / cv[3] / response;
        o[7] = cv[4] / thn[0];
        st.status = status;
        if (a.last) {
            const double *co = st.ybest[0];
            const double th = a.refine ? o[7] : 4 * a.coarse_angle_res;
            st.response = o[0];
            st.coarse_response = 1.0; // no response expansion in the Python path
            st.mean[0] = o[1]; st.mean[1] = o[2]; st.mean[2] = o[3];
            st.cov[0] = co[4]; st.cov[1] = co[6]; st.cov[2] = 0.0;
            st.cov[3] = co[6]; st.cov[4] = co[5]; st.cov[5] = 0.0;
            st.cov[6] = 0.0; st.cov[7] = 0.0; st.cov[8] = th;
            if (a.host_out) a.host_out[b] = st;
        }
    }
}

// ================================================================== prebuilt maps
// helpers.py:24-34 occupancy_grid_map_to_correlation_grid: every cell of the image equal to `occupied_value` is set to
// 1.0 and max-stamped with the float kernel, taps outside the image dropped.  As a gather: a cell's value is the
// largest kernel tap over the occupied cells within reach -- the same set of candidates, and a maximum does not depend
// on the order.  One thread per cell; `cgrid` keeps the float grid as the reference holds it, `g8` what scoring reads,
// int(100 * cell) (helpers.py:142-145).
__global__ __launch_bounds__(256) void map_from_occupancy_kernel(const uint8_t *image, int width, int height, int img_pitch,
                                                                 int occupied_value, const double *kernel, int ksize,
                                                                 double *cgrid, uint8_t *g8) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    const int half = ksize / 2;
    double v = 0.0;
    for (int sy = 0; sy < ksize; sy++) {
        const int py = y - (sy - half); // the stamped point that reaches (x, y) through tap (sx, sy)
        if (py < 0 || py >= height) continue;
        for (int sx = 0; sx < ksize; sx++) {
            const int px = x - (sx - half);
            if (px < 0 || px >= width) continue;
            if (image[(size_t)py * img_pitch + px] == occupied_value) {
                const double cand = kernel[sy * ksize + sx];
                v = cand > v ? cand : v;
            }
        }
    }
    if (image[(size_t)y * img_pitch + x] == occupied_value) v = 1.0 > v ? 1.0 : v;
    cgrid[(size_t)y * width + x] = v;
    g8[(size_t)y * width + x] = (uint8_t)(int)(100 * v);
}

// a correlation grid computed elsewhere -> what scoring reads
__global__ __launch_bounds__(256) void map_from_grid_kernel(const double *cgrid, size_t n, uint8_t *g8) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) g8[i] = (uint8_t)(int)(100 * cgrid[i]);
}

// The query point set of match_scan_sets_with_map (scan_matching.py:141-150): the world point readings of every query
// scan at its own pose, concatenated in scan order, moved by _transform_points(., -ox_real, -oy_real, 0)
// (helpers.py:71-78: a rotation by 0, then the shift).  One block; dynamic LDS = YM_PREP_LDS_BYTES(max_n).
struct MapPointsArgs {
    const YmScanRef *scans; // the query scans
    int32_t n_scans, max_n;
    double ox_real, oy_real;
    double2 *out;           // [sum of readings]
    YmItemState *state;     // nq is written here
};
__global__ __launch_bounds__(1024) void map_points_kernel(MapPointsArgs a) {
    constexpr int NT = 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int s_cnt[(YM_MAX_BEAMS / NT + 1) * (NT / 64)];
    const PrepLds l = prep_lds(lds_raw, a.max_n);
    int total = 0;
    const double c0 = cos(0.0), s0 = sin(0.0), tx = -a.ox_real, ty = -a.oy_real;
    for (int q = 0; q < a.n_scans; q++) {
        const YmScanRef sr = a.scans[q];
        const int np = project_points<NT>(sr, sr.pose[0], sr.pose[1], sr.pose[2], true, l.sx, l.sy, s_cnt);
        for (int i = threadIdx.x; i < np; i += NT) {
            const double px = l.sx[i], py = l.sy[i];
            a.out[total + i] = make_double2((px * c0 - py * s0) + tx, (py * c0 + px * s0) + ty);
        }
        total += np;
        __syncthreads();
    }
    if (threadIdx.x == 0) a.state->nq = total;
}

}  // namespace ym
