// ym_kernels.hpp -- gfx950 kernels of the correlative scan matcher (karto semantics).
//
// Pipeline of one call (B independent items; one item = one query scan vs one chain of base scans):
//   prepare_kernel    point readings, valid-point filter, grid cells, coarse lookup table
//                     (Karto LocalizedRangeScan::Update, ScanMatcher::FindValidPoints, AddScan,
//                      GridIndexLookup::ComputeOffsets)
//   raster_kernel     correlation-grid window with the Gaussian max-smear
//                     (ScanMatcher::AddScans + CorrelationGrid::SmearPoint)
//   correlate_kernel  integer gather-reduce over the coarse (x, y, theta) lattice, split over beam
//                     chunks into partial sums (CorrelateScan loops + GetResponse)
//   score_kernel      partial sums -> sums -> response (+ penalty), per-block maxima
//   finish_kernel     coarse arg-max / tie mean / positional covariance, then the whole fine pass
//                     (offsets, 3x3xN correlate, arg-max, angular covariance) in one block per item
//                     (CorrelateScan tail, ComputePositionalCovariance, ComputeAngularCovariance)
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
// (int)math::Round(v): half away from zero.  trunc(|v| + 0.5) with the sign restored is the same
// integer as floor(v + 0.5) / ceil(v - 0.5) for every |v| < 2^31 and needs no floor/ceil pair.
__device__ __forceinline__ int kt_round_int(double v) {
    const int r = (int)(fabs(v) + 0.5);
    return v < 0.0 ? -r : r;
}
__device__ __forceinline__ int world_to_grid(double w, double off, double scale) {
    return kt_round_int((w - off) * scale);
}

// ------------------------------------------------------------------ block helpers
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

// ---- wave64 reductions on DPP row operations (no LDS traffic).  After the four row steps every
// lane of a 16-lane row holds its row's result; the four row results are combined through
// v_readlane, so the combination order is fixed (bit-reproducible fp64 sums).
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = dpp_i32<CTRL>(__double2loint(v)), hi = dpp_i32<CTRL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
#define YM_DPP_QUAD_1032 0xB1
#define YM_DPP_QUAD_2301 0x4E
#define YM_DPP_ROW_ROR4 0x124
#define YM_DPP_ROW_ROR8 0x128
template <typename Op>
__device__ __forceinline__ double wave_reduce(double v, Op op) {
    v = op(v, dpp_f64<YM_DPP_QUAD_1032>(v));
    v = op(v, dpp_f64<YM_DPP_QUAD_2301>(v));
    v = op(v, dpp_f64<YM_DPP_ROW_ROR4>(v));
    v = op(v, dpp_f64<YM_DPP_ROW_ROR8>(v));
    return op(op(readlane_f64(v, 0), readlane_f64(v, 16)), op(readlane_f64(v, 32), readlane_f64(v, 48)));
}
template <typename Op>
__device__ __forceinline__ int wave_reduce(int v, Op op) {
    v = op(v, dpp_i32<YM_DPP_QUAD_1032>(v));
    v = op(v, dpp_i32<YM_DPP_QUAD_2301>(v));
    v = op(v, dpp_i32<YM_DPP_ROW_ROR4>(v));
    v = op(v, dpp_i32<YM_DPP_ROW_ROR8>(v));
    return op(op(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
              op(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
template <typename Op>
__device__ __forceinline__ unsigned wave_reduce(unsigned v, Op op) {
    struct Wrap { Op op; __device__ int operator()(int a, int b) const { return (int)op((unsigned)a, (unsigned)b); } };
    return (unsigned)wave_reduce((int)v, Wrap{op});
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
struct OpMinI { __device__ int operator()(int a, int b) const { return a < b ? a : b; } };
struct OpMaxI { __device__ int operator()(int a, int b) const { return a > b ? a : b; } };

// development aid: block (0,0,0) thread 0 records the 100 MHz wall clock at phase boundaries
#define YM_STAMP(args, idx)                                                                       \
    do {                                                                                          \
        if ((args).stamps && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) \
            (args).stamps[idx] = wall_clock64();                                                  \
    } while (0)

#define YM_STAMP_B1(args, idx)                                                                    \
    do {                                                                                          \
        if ((args).stamps && threadIdx.x == 0 && blockIdx.x == 1 && blockIdx.y == 0 && blockIdx.z == 0) \
            (args).stamps[idx] = wall_clock64();                                                  \
    } while (0)

// GridIndexLookup::ComputeOffsets for one angle and one point -> window-linear offset
__device__ __forceinline__ int lookup_offset(double2 p, double cosine, double sine, double off_x, double off_y,
                                             double scale, int pitch) {
    const double ox = cosine * p.x - sine * p.y;
    const double oy = sine * p.x + cosine * p.y;
    const int gx = world_to_grid(ox + off_x, off_x, scale);
    const int gy = world_to_grid(oy + off_y, off_y, scale);
    return gx + gy * pitch;
}

// hypothesis cells of one lattice axis: WorldToGrid(centre + (start + i*step)), window coordinates
__device__ __forceinline__ int hyp_cell(double centre, double start, int i, double step, double off, const YmGeom &g) {
    const double v = start + i * step;
    return world_to_grid(centre + v, off, g.scale) + g.border - g.win_origin;
}

// raster tile (also used by prepare_kernel, which builds the raster's work list)
#define YM_TILE_W 64
#define YM_TILE_H 32

// ================================================================== K1 prepare
#define YM_PREP_LDS_BYTES(max_n) ((size_t)(max_n) * 25 + ((size_t)(max_n) / 64 + 2) * 4 + 16)
#define YM_INLINE_SCANS 16
struct YmInlineDesc {        // call descriptor passed in the kernel arguments (single item, few scans)
    YmItem item;
    YmScanRef scans[YM_INLINE_SCANS];
};
struct PrepareArgs {
    const YmScanRef *scans;  // pinned host memory (device-mapped); unused when use_inline
    const YmItem *items;
    int32_t use_inline;
    int32_t pad0;
    YmInlineDesc inl;
    YmGeom g;
    YmLattice lat;           // coarse lattice
    YmItemState *states;
    double2 *qlocal;         // [B][max_n]
    int2 *cells;             // [B][max_base][max_n]  window cell of every base point, NONE when filtered
    int4 *bbox;              // [B][max_base][ceil(max_n/64)] window bounding box of 64 consecutive cells
    double2 *ctrig;          // [B][nt_stride] (cos, sin) of every coarse angle
    int32_t *hypcell;        // [B][2][dim_stride]
    double *probs;           // [B][ny*nx] cleared here, filled by score_kernel
    int32_t max_n, max_base, nt_stride, dim_stride;
    unsigned long long *stamps;
};

// grid (max_base + 1, B), NT threads, dynamic LDS = YM_PREP_LDS_BYTES(max_n).  NT = 512: shortest latency (single
// match); NT = 256: the kernel needs ~100 VGPRs (fp64 sincos), i.e. 16 waves per CU, and four blocks of 256 hide
// each other's barriers and loads better than two of 512 (98 -> 76 us on 256 items).
// blockIdx.x == 0: the query scan; blockIdx.x == 1 + j: base scan j of the item's chain.
template <int NT>
__global__ __launch_bounds__(NT) void prepare_kernel(PrepareArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    YM_STAMP(a, 0);
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const YmItem it = a.use_inline ? a.inl.item : a.items[b];
    const bool is_query = blockIdx.x == 0;
    const int slot = (int)blockIdx.x - 1;
    const int n_cchunks = (a.max_n + 63) / 64;
    if (!is_query && slot >= it.base_count) { // unused chain slot: no points, empty boxes
        int4 *bbox = a.bbox + ((size_t)b * a.max_base + slot) * n_cchunks;
        for (int i = threadIdx.x; i < n_cchunks; i += NT) bbox[i] = make_int4(INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN);
        return;
    }
    const int si = is_query ? it.query : it.base_begin + slot;
    const YmScanRef sr = a.use_inline ? a.inl.scans[si] : a.scans[si];
    const YmScanRef qr = a.use_inline ? a.inl.scans[it.query] : a.scans[it.query];
    double *sx = reinterpret_cast<double *>(lds_raw);
    double *sy = sx + a.max_n;
    int *nxt = reinterpret_cast<int *>(sy + a.max_n);
    int *ex = nxt + a.max_n;                     // exit of the chain walk from point i out of its segment
    int *ent = ex + a.max_n;                     // chain entry node per 64-point segment (max_n/64 + 1)
    unsigned char *chain = reinterpret_cast<unsigned char *>(ent + a.max_n / 64 + 2);
    const bool yag = a.g.semantics == 1;

    // ---- point readings, compacted in beam order (LocalizedRangeScan::Update / _get_point_readings)
    const double px = (is_query && yag) ? 0.0 : sr.pose[0];
    const double py = (is_query && yag) ? 0.0 : sr.pose[1];
    const double pt = (is_query && yag) ? 0.0 : sr.pose[2];
    // One barrier for the whole scan instead of two per NT beams: pass 1 counts the valid beams of every (chunk of NT
    // beams, wave) by ballot, pass 2 re-reads the ranges (L1) and places each valid beam after everything before it.
    constexpr int NW = NT / 64;
    __shared__ int s_cnt[(YM_MAX_BEAMS / NT + 1) * NW];
    const int lane_ = tid & 63, wave_ = tid >> 6;
    const int per = (sr.n + NT - 1) / NT; // chunks of NT beams
    auto valid = [&](int i, double &r) {
        r = 0.0;
        if (i >= sr.n) return false;
        r = sr.ranges[i];
        return yag ? !(r > sr.range_threshold || isnan(r)) : (r >= sr.min_range && r <= sr.range_threshold);
    };
    for (int k = 0; k < per; k++) {
        double r;
        const unsigned long long m = __ballot(valid(k * NT + tid, r));
        if (lane_ == 0) s_cnt[k * NW + wave_] = __popcll(m);
    }
    __syncthreads();
    int running = 0;
    for (int k = 0; k < per; k++) {
        const int i = k * NT + tid;
        double r;
        const bool ok = valid(i, r);
        const unsigned long long m = __ballot(ok);
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const int c = s_cnt[k * NW + w];
            before += w < wave_ ? c : 0;
            total += c;
        }
        if (ok) {
            const int pos = running + before + __popcll(m & ((1ull << lane_) - 1ull));
            const double angle = pt + sr.min_angle + i * sr.angle_inc;
            sx[pos] = px + r * cos(angle);
            sy[pos] = py + r * sin(angle);
        }
        running += total;
    }
    const int np = running;
    __syncthreads();
    YM_STAMP(a, 1);
    YM_STAMP_B1(a, 20);
    // world offset of ROI cell (0,0): MatchScan, "set scan pose to be center of grid"
    const double off_x = qr.pose[0] - (0.5 * (a.g.roi_w - 1) * a.g.res);
    const double off_y = qr.pose[1] - (0.5 * (a.g.roi_w - 1) * a.g.res);

    if (is_query) {
        if (tid == 0) {
            YmItemState &st = a.states[b];
            st.pose[0] = sr.pose[0]; st.pose[1] = sr.pose[1]; st.pose[2] = sr.pose[2];
            st.center[0] = sr.pose[0]; st.center[1] = sr.pose[1]; st.center[2] = sr.pose[2];
            st.off_x = off_x;
            st.off_y = off_y;
            st.nq = np;
            st.status = 0;
            st.regular[0] = st.regular[1] = 0;
            st.base_count = it.base_count;
        }
        // sensor-frame coordinates (karto: Transform(pose).InverseTransformPose; yagpy: points_local)
        double2 *ql = a.qlocal + (size_t)b * a.max_n;
        const bool identity = yag || (sr.pose[0] == 0.0 && sr.pose[1] == 0.0 && sr.pose[2] == 0.0);
        const double cr = cos(0.0 - sr.pose[2]), sn = sin(0.0 - sr.pose[2]);
        __syncthreads();
        for (int i = tid; i < np; i += NT) {
            double2 l;
            if (identity) {
                l = make_double2(sx[i], sy[i]);
            } else {
                const double dx = sx[i] - sr.pose[0], dy = sy[i] - sr.pose[1];
                l = make_double2(cr * dx + (0.0 - sn) * dy, sn * dx + cr * dy);
            }
            ql[i] = l;
        }
        // one fp64 sin/cos per coarse angle (GridIndexLookup::ComputeOffsets); the cell offsets
        // themselves are computed by the correlate blocks that consume them
        if (tid < a.lat.nt) {
            const double angle = (sr.pose[2] - a.lat.angle_off) + tid * a.lat.angle_res;
            a.ctrig[(size_t)b * a.nt_stride + tid] = make_double2(cos(angle), sin(angle));
        }
        YM_STAMP(a, 2);
        for (int i = tid; i < a.lat.nx * a.lat.ny; i += NT) a.probs[(size_t)b * a.lat.nx * a.lat.ny + i] = 0.0;
        // coarse hypothesis cells + regularity flag
        int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
        int32_t *cy = cx + a.dim_stride;
        for (int i = tid; i < a.lat.nx; i += NT) cx[i] = hyp_cell(sr.pose[0], -a.lat.off_x, i, a.lat.step_x, off_x, a.g);
        for (int i = tid; i < a.lat.ny; i += NT) cy[i] = hyp_cell(sr.pose[1], -a.lat.off_y, i, a.lat.step_y, off_y, a.g);
        __syncthreads();
        {
            const int stx = kt_round_int(a.lat.step_x * a.g.scale), sty = kt_round_int(a.lat.step_y * a.g.scale);
            int ok = 1;
            for (int i = tid; i < a.lat.nx; i += NT) ok &= (cx[i] == cx[0] + i * stx);
            for (int i = tid; i < a.lat.ny; i += NT) ok &= (cy[i] == cy[0] + i * sty);
            ok = __syncthreads_and(ok);
            if (tid == 0) a.states[b].regular[0] = ok;
        }
        YM_STAMP(a, 18);
        return;
    }

    // ---- valid-point filter (ScanMatcher::FindValidPoints / validate_points), parallel form:
    // nxt[i] = first j > i farther than d from point i; the trigger chain is 0 -> nxt[0] -> ...;
    // the run that ends at chain node t = nxt[s] is kept or dropped by the sign of ss(s, t).
    const double min_sq = yag ? 0.2 * 0.2 : 0.1 * 0.1;
    const double vpx = qr.pose[0], vpy = qr.pose[1];
    for (int i = tid; i < np; i += NT) {
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
    YM_STAMP_B1(a, 21);
    // Mark the chain 0 -> nxt[0] -> ... without one long serial walk: cut the points into segments
    // of 64; (1) every point walks to the first node past its own segment, (2) one thread hops from
    // segment to segment with those exits (<= n/64 hops), (3) one thread per entered segment marks
    // the chain nodes inside it.
    constexpr int SEG = 64;
    const int nseg = (np + SEG - 1) / SEG;
    for (int i = tid; i < np; i += NT) {
        const int seg_end = min(np, (i / SEG + 1) * SEG);
        int j = nxt[i];
        while (j < seg_end) j = nxt[j];
        ex[i] = j;
    }
    for (int i = tid; i < nseg; i += NT) ent[i] = -1;
    __syncthreads();
    if (tid == 0)
        for (int cur = 0; cur < np; cur = ex[cur]) ent[cur / SEG] = cur;
    __syncthreads();
    for (int sgi = tid; sgi < nseg; sgi += NT) {
        int c = ent[sgi];
        if (c >= 0) {
            const int seg_end = min(np, (sgi + 1) * SEG);
            for (; c < seg_end; c = nxt[c]) chain[c] = 1;
        }
    }
    __syncthreads();
    YM_STAMP_B1(a, 22);
    int2 *cells = a.cells + ((size_t)b * a.max_base + slot) * a.max_n;
    int4 *bbox = a.bbox + ((size_t)b * a.max_base + slot) * n_cchunks;
    for (int i0 = 0; i0 < n_cchunks * 64; i0 += NT) {
        const int i = i0 + tid; // a wave covers one 64-cell chunk
        int2 c = make_int2(YM_CELL_NONE, YM_CELL_NONE);
        if (i < np) {
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
        }
        if (i < a.max_n) cells[i] = c;
        // bounding box of the chunk's rasterised cells: the raster kernel reads a chunk only when
        // this box touches its tile
        const bool has = c.x != YM_CELL_NONE;
        const int x0 = wave_reduce(has ? c.x : INT32_MAX, OpMinI()), y0 = wave_reduce(has ? c.y : INT32_MAX, OpMinI());
        const int x1 = wave_reduce(has ? c.x : INT32_MIN, OpMaxI()), y1 = wave_reduce(has ? c.y : INT32_MIN, OpMaxI());
        if ((tid & 63) == 0 && i / 64 < n_cchunks) bbox[i / 64] = make_int4(x0, y0, x1, y1);
    }
    YM_STAMP_B1(a, 23);
}

// ================================================================== K1c tiles: the raster kernel's work list (batches)
// One raster block per tile of the window lets ~3 of 4 blocks find out, after a round trip through the chunk boxes
// and a barrier, that they have nothing to do; on a batch that was most of the raster's time.  For batches this
// kernel (one block per item) turns the item's chunk boxes into the list of tiles that have work: a tile bitmap in
// LDS (every box marks the few tiles its smear halo reaches), compacted together with the tiles that hold stale
// bytes from an earlier call and only need clearing.  The raster kernel then runs one block per list entry.
// (Doing this in the last prepare block to finish needs a device-scope release per block, which on this part
// writes the XCD's L2 back: measured 98 -> 771 us for the prepare kernel.  A kernel boundary is cheaper.)
#define YM_TILES_THREADS 256
struct TilesArgs {
    const int4 *bbox;        // [B][max_base][ceil(max_n/64)]
    uint16_t *tile_list;     // [B][tile_cap] tile index (tiy * tiles_x + tix), | 0x8000 = only needs clearing
    int32_t *tile_count;     // [B]
    const uint8_t *tile_zero;// [B][tiles_y][tiles_x] 1 = the tile's memory is known to hold zeros
    int32_t max_n, max_base, half_kernel;
    int32_t tiles_x, tiles_y, tile_cap;
    int32_t launch[4];       // tile rectangle (x0, y0, x1, y1) the raster covers in this call
};
// grid (B), dynamic LDS = 4 * ceil(tiles_x * tiles_y / 32) bytes
__global__ __launch_bounds__(YM_TILES_THREADS) void tiles_kernel(TilesArgs a) {
    extern __shared__ unsigned tile_bits[];
    __shared__ int s_n;
    constexpr int NT = YM_TILES_THREADS;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int ntiles = a.tiles_x * a.tiles_y, nwords = (ntiles + 31) / 32;
    for (int i = tid; i < nwords; i += NT) tile_bits[i] = 0u;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int h = a.half_kernel;
    const int lx0 = a.launch[0], ly0 = a.launch[1], lx1 = a.launch[2], ly1 = a.launch[3];
    const int n_boxes = a.max_base * ((a.max_n + 63) / 64);
    const int4 *bbox = a.bbox + (size_t)b * n_boxes;
    for (int c = tid; c < n_boxes; c += NT) {
        const int4 bb = bbox[c];
        if (bb.x > bb.z) continue;
        // tiles whose halo-extended rectangle [t*T - h, t*T + T + h - 1] meets the box (the raster kernel's own test)
        const int tx0 = max(lx0, max(bb.x - h, 0) / YM_TILE_W), tx1 = min(lx1, (bb.z + h) / YM_TILE_W);
        const int ty0 = max(ly0, max(bb.y - h, 0) / YM_TILE_H), ty1 = min(ly1, (bb.w + h) / YM_TILE_H);
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) {
                const int t = ty * a.tiles_x + tx;
                atomicOr(&tile_bits[t >> 5], 1u << (t & 31));
            }
    }
    __syncthreads();
    const int ltx = lx1 - lx0 + 1, lty = ly1 - ly0 + 1;
    const uint8_t *tz = a.tile_zero + (size_t)b * ntiles;
    uint16_t *list = a.tile_list + (size_t)b * a.tile_cap;
    for (int i = tid; i < ltx * lty; i += NT) {
        const int ty = ly0 + i / ltx, tx = lx0 + i % ltx, t = ty * a.tiles_x + tx;
        const bool hit = (tile_bits[t >> 5] >> (t & 31)) & 1u;
        if (hit || tz[t] == 0) list[atomicAdd(&s_n, 1)] = (uint16_t)(t | (hit ? 0 : 0x8000));
    }
    __syncthreads();
    if (tid == 0) a.tile_count[b] = s_n;
}

// ================================================================== K1b select (only when the smear kernel has taps == 100 off-centre)
// Karto's AddScan skips a point whose cell already holds 100 ("value already set").  With
// smear_deviation >= 9.99 * resolution the four neighbours of an occupied cell are stamped 100 as
// well, so whether a point is rasterised depends on the points before it: a point is EFFECTIVE iff
// no earlier effective point lies within squared cell distance z2max (the radius of the kernel's
// 100-valued disc), in Karto's order (base scans in order, beams in order).
//
// Parallel form of that greedy rule.  Only the earliest point of a cell can be effective (a later
// one is knocked out by it, or by whatever knocked it out).  So: (1) hash every cell to its earliest
// point index (LDS, atomicMin); (2) relax the undecided cells: a cell whose earlier neighbours are
// all decided "no" becomes effective, a cell with an effective earlier neighbour is out -- decisions
// are final, so reading a neighbour's fresh or stale state is equally safe, and the globally
// earliest undecided cell always resolves, so the loop ends; (3) erase every point
// that is not the earliest of an effective cell from `cells`.  One block per item.
struct SelectArgs {
    int2 *cells;          // [B][max_base][max_n]
    int32_t max_n, max_base;
    int32_t z2max;        // largest squared distance whose kernel value is 100
    int32_t log2cap;      // hash capacity = 1 << log2cap entries (dynamic LDS: 9 bytes per entry)
    unsigned long long *stamps;
};

__device__ __forceinline__ unsigned select_key(int x, int y) {
    return ((unsigned)(y + 32768) << 16) | ((unsigned)(x + 32768) & 0xffffu);
}
// The table is open addressing over BUCKETS of four keys (one 16-byte LDS read per probe; a plain
// linear-probing table made the slowest lane of a wave walk 20-40 slots).  Keys fill a bucket front
// to back and are never removed, so an empty last slot means "not in this bucket or any later one".
// slot of `key`, or -1.  bmask = buckets - 1, shift = 32 - log2(buckets).
__device__ __forceinline__ int select_find(const unsigned *keys, unsigned bmask, int shift, unsigned key) {
    unsigned bk = (key * 2654435761u) >> shift;
    for (;;) {
        const uint4 q = *reinterpret_cast<const uint4 *>(keys + 4 * bk);
        if (q.x == key) return (int)(4 * bk);
        if (q.y == key) return (int)(4 * bk + 1);
        if (q.z == key) return (int)(4 * bk + 2);
        if (q.w == key) return (int)(4 * bk + 3);
        if (q.w == 0u) return -1;
        bk = (bk + 1u) & bmask;
    }
}
// slot of `key`, inserting it if absent
__device__ __forceinline__ int select_insert(unsigned *keys, unsigned bmask, int shift, unsigned key) {
    unsigned bk = (key * 2654435761u) >> shift;
    for (;;) {
        const uint4 q = *reinterpret_cast<const uint4 *>(keys + 4 * bk);
        int j = (q.x == key || q.x == 0u) ? 0 : (q.y == key || q.y == 0u) ? 1 : (q.z == key || q.z == 0u) ? 2 : (q.w == key || q.w == 0u) ? 3 : 4;
        for (; j < 4; j++) {
            const unsigned prev = atomicCAS(&keys[4 * bk + j], 0u, key);
            if (prev == 0u || prev == key) return (int)(4 * bk + j);
        }
        bk = (bk + 1u) & bmask;
    }
}

// NB = 5 for z2max = 1 (plus-shaped disc), 9 for z2max = 2
template <int NB>
__global__ __launch_bounds__(1024) void select_kernel(SelectArgs a) {
    constexpr int NT = 1024, KMAX = 16, NW = (NB - 1) / 4;
    constexpr int DX[9] = {0, 1, -1, 0, 0, 1, 1, -1, -1};
    constexpr int DY[9] = {0, 0, 0, 1, -1, 1, -1, 1, -1};
    extern __shared__ unsigned sel_lds[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const unsigned cap = 1u << a.log2cap, bmask = (cap >> 2) - 1u;
    const int shift = 32 - (a.log2cap - 2);
    unsigned *keys = sel_lds;
    unsigned *minidx = sel_lds + cap;
    unsigned char *status = reinterpret_cast<unsigned char *>(sel_lds + 2 * cap); // 0 undecided, 1 effective, 2 out
    YM_STAMP(a, 24);
    for (unsigned i = tid; i < cap; i += NT) { keys[i] = 0u; minidx[i] = 0xffffffffu; status[i] = 0; }
    __syncthreads();
    int2 *cells = a.cells + (size_t)b * a.max_base * a.max_n;
    const int total = a.max_base * a.max_n;
    YM_STAMP(a, 25);
    // (1) cell -> earliest point index
    for (int e = tid; e < total; e += NT) {
        const int2 c = cells[e];
        if (c.x == YM_CELL_NONE) continue;
        const int slot = select_insert(keys, bmask, shift, select_key(c.x, c.y));
        atomicMin(&minidx[slot], (unsigned)e);
    }
    __syncthreads();
    YM_STAMP(a, 26);
    // (2a) per owned slot (tid + k*NT): the slots of the neighbour cells that hold an EARLIER point,
    // packed as 16-bit slot numbers (0xffff = none).  A cell with no earlier neighbour is effective.
    unsigned long long nb[KMAX][NW];
    unsigned und = 0;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
#pragma unroll
        for (int w = 0; w < NW; w++) nb[k][w] = ~0ull;
        const unsigned s = tid + k * NT;
        if (s >= cap) continue;
        const unsigned key = keys[s];
        if (key == 0u) continue;
        const unsigned me = minidx[s];
        const int x = (int)(key & 0xffffu) - 32768, y = (int)(key >> 16) - 32768;
        bool any = false;
#pragma unroll
        for (int n = 1; n < NB; n++) {
            const int t = select_find(keys, bmask, shift, select_key(x + DX[n], y + DY[n]));
            if (t >= 0 && minidx[t] < me) {
                any = true;
                const int j = n - 1;
                nb[k][j >> 2] &= ~(0xffffull << (16 * (j & 3)));
                nb[k][j >> 2] |= (unsigned long long)(unsigned)t << (16 * (j & 3));
            }
        }
        if (any) und |= 1u << k;
        else status[s] = 1;
    }
    YM_STAMP(a, 27);
    // (2b) asynchronous relaxation, no barriers: every wave keeps re-reading the state of the earlier
    // neighbours of its undecided cells.  A decision is final and is taken only from final states
    // (a stale "undecided" read merely delays it), and all 16 waves of the block are resident, so
    // this terminates with the sequential greedy result whatever the interleaving.
    unsigned char *vst = status;
    unsigned wmask = 0;
#pragma unroll
    for (int k = 0; k < KMAX; k++) wmask |= __ballot((und >> k) & 1u) ? (1u << k) : 0u;
    while (wmask) {
        unsigned m = wmask;
        while (m) {
            const int k = __builtin_ctz(m); // wave-uniform
            m &= m - 1;
            bool still = false;
            asm volatile("" ::: "memory"); // re-read the states every time
            if ((und >> k) & 1u) {
                bool knocked = false, pending = false;
#pragma unroll
                for (int j = 0; j < NB - 1; j++) {
                    const unsigned t = (unsigned)(nb[k][j >> 2] >> (16 * (j & 3))) & 0xffffu;
                    if (t != 0xffffu) {
                        const unsigned char stt = vst[t];
                        knocked |= stt == 1;
                        pending |= stt == 0;
                    }
                }
                const unsigned s = tid + k * NT;
                if (knocked) vst[s] = 2;
                else if (!pending) vst[s] = 1;
                else still = true;
                if (!still) und &= ~(1u << k);
            }
            if (__ballot(still) == 0ull) wmask &= ~(1u << k);
        }
    }
    __syncthreads();
    YM_STAMP(a, 28);
    // (3) keep only the earliest point of every effective cell
    for (int e = tid; e < total; e += NT) {
        const int2 c = cells[e];
        if (c.x == YM_CELL_NONE) continue;
        const int t = select_find(keys, bmask, shift, select_key(c.x, c.y));
        if (!(t >= 0 && status[t] == 1 && minidx[t] == (unsigned)e)) cells[e] = make_int2(YM_CELL_NONE, YM_CELL_NONE);
    }
    YM_STAMP(a, 29);
}

// ================================================================== K2 raster
struct RasterArgs {
    const int2 *cells;
    const int4 *bbox;     // [B][max_base][ceil(max_n/64)]
    const YmItemState *states;
    YmGeom g;
    uint8_t *grid;        // [B][win_w rows][pitch]
    size_t grid_stride;   // bytes per item
    uint8_t *planes;      // [B][2][win_w rows][pitch/2]: plane p holds columns 2*x+p of the window
    const uint8_t *lut;   // smear kernel value by squared cell distance: lut[dx*dx + dy*dy], 2*h*h + 1 entries
    int32_t max_n, max_base;
    uint8_t *tile_zero;   // [B][tiles_y][tiles_x]: 1 = this tile of the window memory is known to hold zeros
    int32_t tiles_x, tiles_y; // full tiling of the window
    int32_t tile_x0, tile_y0; // first tile of the launched sub-grid (tiles outside it are known to be zero)
    int32_t ltx;              // tile columns of the launched sub-grid
    const uint16_t *tile_list; // [B][tile_cap] work list built by tiles_kernel, or null: one block per sub-grid tile
    const int32_t *tile_count; // [B]
    int32_t tile_cap;
    unsigned long long *stamps;
};

// grid (launched tiles in x, in y, B), 256 threads.  Each block owns one 64x32 tile of the window and
// writes every byte of it exactly once (so no separate clear pass exists; a tile that is empty now and
// whose memory is known to be zero from an earlier call is skipped).  Karto's SmearPoint
// max-stamps a (2h+1)^2 kernel at every occupied cell; the kernel value depends only on the squared
// cell distance and never grows with it (checked on the host when the matcher is created), so a
// cell's final value is lut[min squared distance to an occupied cell inside the (2h+1)^2 window]:
//   row pass   g(y, x)  = min |dx| <= h with cell (y, x+dx) occupied      (bit scans on a row bitmap)
//   column pass m(y, x) = min over |dy| <= h of dy^2 + g(y+dy, x)^2        (8 cells per lane)
// NT = 256: one tile row per thread, shortest latency (single match); NT = 128: two rows per thread, twice the
// blocks per CU -- the tiles with work are latency-bound, so a batch gains (raster 160 -> 140 us on 256 items)
template <int NT>
__global__ __launch_bounds__(NT) void raster_kernel(RasterArgs a) {
    constexpr int TW = YM_TILE_W, TH = YM_TILE_H, HM = YM_MAX_KERNEL_HALF;
    constexpr int RW = (TW + 2 * HM + 63) / 64 + 1;  // 64-bit words per bitmap row, + 1 so a funnel read never leaves the row
    constexpr int LPR = TW / 8;                        // lanes per tile row (8 cells each)
    __shared__ unsigned long long occ[(TH + 2 * HM) * RW];
    __shared__ __attribute__((aligned(8))) unsigned char grow[(TH + 2 * HM) * TW];
    __shared__ unsigned char lut[2 * HM * HM + 8];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    // grid (tiles of the launched sub-grid, B).  With a work list (batches) block i takes entry i and the blocks
    // past the list's end leave at once; without one (a few items: one more launch would cost more than it saves)
    // block i is tile i of the sub-grid and finds out by itself whether any chunk box reaches it.
    const bool listed = a.tile_list != nullptr;
    unsigned entry;
    if (listed) {
        if ((int)blockIdx.x >= a.tile_count[b]) return;
        entry = a.tile_list[(size_t)b * a.tile_cap + blockIdx.x];
    } else {
        const int sy = (int)blockIdx.x / a.ltx, sx = (int)blockIdx.x - sy * a.ltx;
        // rotate the tile column by the row: a sub-grid width that is a multiple of 8 would otherwise pin every
        // tile column (i.e. every wall) to one XCD
        entry = (unsigned)((a.tile_y0 + sy) * a.tiles_x + a.tile_x0 + (sx + 3 * sy + 5 * b) % a.ltx);
    }
    const int tile = (int)(entry & 0x7fffu);
    const int h = a.g.half_kernel;
    const int OW = TW + 2 * h, OH = TH + 2 * h;
    const int tiy = tile / a.tiles_x, tix = tile - tiy * a.tiles_x;
    const int tx0 = tix * TW, ty0 = tiy * TH;
    YM_STAMP(a, 4);
    // candidate chunks: 64 consecutive cells of one base scan whose bounding box touches tile + halo
    const int n_cchunks = (a.max_n + 63) / 64;
    const int n_boxes = a.max_base * n_cchunks;
    const int4 *bbox = a.bbox + (size_t)b * n_boxes;
    const int lo_x = tx0 - h, hi_x = tx0 + TW + h - 1, lo_y = ty0 - h, hi_y = ty0 + TH + h - 1;
    uint8_t *grid = a.grid + (size_t)b * a.grid_stride;
    // thread -> 8 consecutive cells (x8 ..) of tile rows y0, y0 + NT / LPR, ...
    const int y0 = tid / LPR, x8 = (tid % LPR) * 8;
    const size_t plane_bytes = (size_t)(a.g.pitch / 2) * a.g.win_w;
    uint8_t *planes = a.planes + (size_t)b * a.grid_stride;
    // the window row-major and its even / odd column planes (v_perm_b32 byte gathers) for 8 cells of tile row y
    auto store8 = [&](int y, uint32_t p0, uint32_t p1) {
        if (ty0 + y < a.g.win_w) {
            *reinterpret_cast<uint2 *>(grid + (size_t)(ty0 + y) * a.g.pitch + tx0 + x8) = make_uint2(p0, p1);
            uint8_t *pl = planes + (size_t)(ty0 + y) * (a.g.pitch / 2) + (tx0 + x8) / 2;
            *reinterpret_cast<uint32_t *>(pl) = __builtin_amdgcn_perm(p1, p0, 0x06040200u);
            *reinterpret_cast<uint32_t *>(pl + plane_bytes) = __builtin_amdgcn_perm(p1, p0, 0x07050301u);
        }
    };
    auto zero_tile = [&]() {
        for (int y = y0; y < TH; y += NT / LPR) store8(y, 0u, 0u);
    };
    uint8_t *tz = a.tile_zero + ((size_t)b * a.tiles_y + tiy) * a.tiles_x + tix;
    __shared__ int s_hits[256];
    __shared__ int s_nhits;
    if (entry & 0x8000u) { // no chunk reaches this tile, but its memory still holds an earlier call's bytes
        zero_tile();
        if (tid == 0) *tz = 1;
        return;
    }
    if (tid == 0) s_nhits = 0;
    if (!listed) { // decide "no box at all" before touching LDS
        int my_hits = 0;
        for (int c = tid; c < n_boxes; c += NT) {
            const int4 bb = bbox[c];
            my_hits += (bb.x <= hi_x && bb.z >= lo_x && bb.y <= hi_y && bb.w >= lo_y) ? 1 : 0;
        }
        if (__syncthreads_or(my_hits) == 0) {
            // empty tile: zeros -- unless this memory is already known to be zero from an earlier call
            if (*tz == 0) {
                zero_tile();
                __syncthreads();
                if (tid == 0) *tz = 1;
            }
            return;
        }
    } else {
        __syncthreads();
    }
    // chunks whose box touches tile + halo, compacted so that the cell loads of several chunks are in flight together
    for (int c = tid; c < n_boxes; c += NT) {
        const int4 bb = bbox[c];
        if (bb.x <= hi_x && bb.z >= lo_x && bb.y <= hi_y && bb.w >= lo_y) {
            const int at = atomicAdd(&s_nhits, 1);
            if (at < 256) s_hits[at] = c;
        }
    }
    for (int i = tid; i < OH * RW; i += NT) occ[i] = 0ull;
    for (int i = tid; i <= 2 * h * h; i += NT) lut[i] = a.lut[i];
    __syncthreads();
    const int nhits = s_nhits;
    const int2 *cells = a.cells + (size_t)b * a.max_base * a.max_n;
    unsigned *occ32 = reinterpret_cast<unsigned *>(occ);
    int any = 0;
    if (nhits <= 256) {
        // work item = (hit chunk, cell of the chunk); 4 items per thread in flight
        const int nwork = nhits * 64;
        for (int w0 = 0; w0 < nwork; w0 += 4 * NT) {
            int2 cc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int w = w0 + u * NT + tid;
                cc[u] = make_int2(YM_CELL_NONE, YM_CELL_NONE);
                if (w < nwork) {
                    const int chunk = s_hits[w >> 6];
                    const int slot = chunk / n_cchunks, i = (chunk - slot * n_cchunks) * 64 + (w & 63);
                    if (i < a.max_n) cc[u] = cells[(size_t)slot * a.max_n + i];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int lx = cc[u].x - lo_x, ly = cc[u].y - lo_y;
                if (cc[u].x != YM_CELL_NONE && lx >= 0 && lx < OW && ly >= 0 && ly < OH) {
                    atomicOr(&occ32[ly * (RW * 2) + (lx >> 5)], 1u << (lx & 31));
                    any = 1;
                }
            }
        }
    } else {
        // more hit chunks than the list holds: walk every box (rare)
        for (int c0 = 0; c0 < n_boxes; c0 += NT) {
            const int c = c0 + tid;
            bool hit = false;
            if (c < n_boxes) {
                const int4 bb = bbox[c];
                hit = bb.x <= hi_x && bb.z >= lo_x && bb.y <= hi_y && bb.w >= lo_y;
            }
            unsigned long long mask = __ballot(hit);
            while (mask) {
                const int bit = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int chunk = c0 + (tid & ~63) + bit; // wave-uniform
                const int slot = chunk / n_cchunks, ci = chunk - slot * n_cchunks;
                const int i = ci * 64 + (tid & 63);
                if (i < a.max_n) {
                    const int2 c2 = cells[(size_t)slot * a.max_n + i];
                    const int lx = c2.x - lo_x, ly = c2.y - lo_y;
                    if (c2.x != YM_CELL_NONE && lx >= 0 && lx < OW && ly >= 0 && ly < OH) {
                        atomicOr(&occ32[ly * (RW * 2) + (lx >> 5)], 1u << (lx & 31));
                        any = 1;
                    }
                }
            }
        }
    }
    any = __syncthreads_or(any);
    YM_STAMP(a, 5);
    if (!any) {
        if (*tz == 0) {
            zero_tile();
            __syncthreads();
            if (tid == 0) *tz = 1;
        }
        return;
    }
    if (tid == 0) *tz = 0;
    // row pass: nearest occupied |dx| <= h, 255 = none.  Bit x+h of a bitmap row is tile column x.
    // Work item = 8 consecutive cells of one (halo) row.  Walls are thin: most 8-cell groups see no bit within
    // reach at all and leave after one test.
    const unsigned long long wmask = (1ull << (2 * h + 1)) - 1ull, lmask = (1ull << h) - 1ull;
    const unsigned long long gmask = (1ull << (2 * h + 8)) - 1ull;
    for (int i = tid; i < OH * LPR; i += NT) {
        const int ry = i / LPR, rx = (i % LPR) * 8;
        const int w = rx >> 6, sft = rx & 63;
        const unsigned long long lo = occ[ry * RW + w], hi = occ[ry * RW + w + 1];
        const unsigned long long sw = (sft ? ((lo >> sft) | (hi << (64 - sft))) : lo) & gmask; // bits rx .. rx + 7 + 2h
        uint32_t out[2] = {0xffffffffu, 0xffffffffu};
        if (sw) {
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const unsigned long long win = (sw >> q) & wmask;
                if (win) {
                    const unsigned long long right = win >> h, left = win & lmask;
                    const int dr = right ? (__ffsll((long long)right) - 1) : 255;
                    const int dl = left ? (h - 63 + __clzll((long long)left)) : 255;
                    const unsigned g = (unsigned)(dr < dl ? dr : dl);
                    out[q >> 2] = (out[q >> 2] & ~(0xffu << (8 * (q & 3)))) | (g << (8 * (q & 3)));
                }
            }
        }
        *reinterpret_cast<uint2 *>(&grow[ry * TW + rx]) = make_uint2(out[0], out[1]);
    }
    __syncthreads();
    YM_STAMP(a, 6);
    // column pass: 8 cells per lane as four pairs of 16-bit lanes (cells 0|2, 1|3, 4|6, 5|7): the candidate
    // g*g + dy*dy is at most 255^2 + h^2 < 65536, so one v_pk_mad_u16 + one v_pk_min_u16 serve two cells.
    // A row whose 8 distances are all "none" contributes nothing.
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const unsigned max_d2 = (unsigned)(2 * h * h);
    for (int y = y0; y < TH; y += NT / LPR) {
        us2 mn2[4];
#pragma unroll
        for (int q = 0; q < 4; q++) mn2[q] = (us2){0xffff, 0xffff};
        for (int dy = -h; dy <= h; dy++) {
            const uint2 gg = *reinterpret_cast<const uint2 *>(&grow[(y + h + dy) * TW + x8]);
            if ((gg.x & gg.y) == 0xffffffffu) continue;
            const unsigned short d2 = (unsigned short)(dy * dy);
            const us2 dd = (us2){d2, d2};
            const uint32_t u[4] = {gg.x & 0x00ff00ffu, (gg.x >> 8) & 0x00ff00ffu, gg.y & 0x00ff00ffu, (gg.y >> 8) & 0x00ff00ffu};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                us2 gq;
                __builtin_memcpy(&gq, &u[q], 4);
                mn2[q] = __builtin_elementwise_min(mn2[q], (us2)(gq * gq + dd)); // g = 255 (none) is larger than any real distance
            }
        }
        unsigned mn[8];
        mn[0] = mn2[0].x; mn[2] = mn2[0].y; mn[1] = mn2[1].x; mn[3] = mn2[1].y;
        mn[4] = mn2[2].x; mn[6] = mn2[2].y; mn[5] = mn2[3].x; mn[7] = mn2[3].y;
        uint32_t packed[2] = {0u, 0u};
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const unsigned v = mn[q] <= max_d2 ? lut[mn[q]] : 0u;
            packed[q >> 2] |= v << (8 * (q & 3));
        }
        store8(y, packed[0], packed[1]);
    }
    YM_STAMP(a, 7);
}

// ================================================================== K4 correlate (coarse lattice)
#define YM_CORR_THREADS 256
struct CorrArgs {
    YmGeom g;
    YmLattice lat;
    const uint8_t *grid;
    size_t grid_stride;
    const uint8_t *planes;  // even/odd column planes of the window (coarse step = 2 cells)
    const double2 *ctrig;   // [B][nt_stride] (cos, sin) per coarse angle
    const double2 *qlocal;  // [B][max_n] query points in the sensor frame
    const int32_t *hypcell;
    const YmItemState *states;
    uint16_t *partial;     // [B][n_chunks][nt][ny][nx_pad], 16-bit: a chunk sums at most 512 beams x 100
    size_t partial_stride; // per item
    int32_t max_n, nt_stride, dim_stride;
    int32_t chunk;         // beams per chunk (multiple of 16, <= 512 keeps the 16-bit lanes from overflowing)
    int32_t n_chunks;
    int32_t tpb;           // adjacent angles per block (1, 2 or 4)
    int32_t ngx;           // x groups per row = ceil(nx / G)
    int32_t nx_pad;        // ngx * G
    int32_t sx;            // cell stride between x-adjacent hypotheses (1 or 2)
    unsigned long long *stamps;
};

// Workgroups are handed to the 8 XCDs round-robin in launch order, and each XCD has its own L2.  The blocks of
// one item (all angles and beam chunks) read the same grid band, so they are renumbered to run on ONE XCD: XCD x
// works through items x, x + 8, x + 16, ...  Returns the item; `inner` = index within the item's blocks.
__device__ __forceinline__ int xcd_item_of_block(int &inner_x, int &inner_y) {
    const int per_item = gridDim.x * gridDim.y, nb = gridDim.z;
    const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int full = (nb / 8) * 8 * per_item;
    int item = blockIdx.z, inner = blockIdx.x + gridDim.x * blockIdx.y;
    if (L < full) {
        const int t = L >> 3;
        item = (L & 7) + 8 * (t / per_item);
        inner = t % per_item;
    }
    inner_y = inner / gridDim.x;
    inner_x = inner - inner_y * gridDim.x;
    return item;
}

// the same for a grid (blocks per item, items)
__device__ __forceinline__ int xcd_item_of_block_2d(int &inner) {
    const int per_item = gridDim.x, nb = gridDim.y;
    const int L = blockIdx.x + gridDim.x * blockIdx.y;
    int item = blockIdx.y;
    inner = blockIdx.x;
    if (L < (nb / 8) * 8 * per_item) {
        const int t = L >> 3;
        item = (L & 7) + 8 * (t / per_item);
        inner = t % per_item;
    }
    return item;
}

// The correlate kernels accumulate 16 hypotheses per lane in eight dwords of two 16-bit lanes each: acc[2j] holds
// hypotheses 4j and 4j + 2, acc[2j + 1] holds 4j + 1 and 4j + 3 (even / odd bytes of grid dword j).  A chunk is at
// most 512 beams of at most 100, so partial sums are stored as 16-bit values, in hypothesis order (v_perm_b32).
__device__ __forceinline__ void store_partial16(uint16_t *out, const uint32_t (&acc)[8]) {
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        w[2 * j] = __builtin_amdgcn_perm(acc[2 * j + 1], acc[2 * j], 0x05040100u);     // hyp 4j, 4j + 1
        w[2 * j + 1] = __builtin_amdgcn_perm(acc[2 * j + 1], acc[2 * j], 0x07060302u); // hyp 4j + 2, 4j + 3
    }
    *reinterpret_cast<uint4 *>(out) = make_uint4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<uint4 *>(out + 8) = make_uint4(w[4], w[5], w[6], w[7]);
}

// Lane job = 16 x-adjacent hypotheses of one lattice row for one angle.  The coarse search steps
// 2 cells (SX = 2), so its hypotheses' cells for one beam are every other byte of a row: they are
// contiguous in the even- or odd-column plane the raster kernel also writes (which plane is a
// per-beam, wave-uniform choice: parity of hypothesis column + beam offset).  For every beam of
// its chunk a lane loads the 16 plane bytes of its 16 hypotheses (row start + wave-uniform beam
// offset) and accumulates them in 16-bit lanes: no cross-lane reduction, no wasted bytes.
// Partial sums per beam chunk are added up by score_kernel.  Each block first builds the offsets
// of its own beam chunk in LDS (GridIndexLookup::ComputeOffsets for one angle: rotate the
// sensor-frame point, WorldToGrid), so no lookup table ever round-trips through HBM.  Loads are
// issued 16 beams at a time; entries past the last beam are 0 and are masked by a scalar.
// grid (ceil(ny*ngx / 256), nt * n_chunks, B).
template <int SX, int U /* beams in flight per lane */>
__global__ __launch_bounds__(YM_CORR_THREADS) void correlate_kernel(CorrArgs a) {
    constexpr int G = 16;           // hypotheses per lane
    int bx, by;
    const int b = xcd_item_of_block(bx, by);
    // a block covers jobs_pb lane jobs of each of tpb adjacent angles for one beam chunk: waves of
    // one block read overlapping grid patches (adjacent angles shift the patch by a few cells)
    const int tpb = a.tpb, jobs_pb = YM_CORR_THREADS / tpb;
    const int ktiles = (a.lat.nt + tpb - 1) / tpb;
    const int k = (by % ktiles) * tpb + threadIdx.x / jobs_pb, chunk = by / ktiles;
    const int job = bx * jobs_pb + threadIdx.x % jobs_pb;
    __shared__ int offs_all[4][512];
    int *offs = offs_all[threadIdx.x / jobs_pb];
    YM_STAMP(a, 8);
    const int njobs = a.lat.ny * a.ngx;
    const bool k_ok = k < a.lat.nt;
    const YmItemState &st = a.states[b];
    const int nq = st.nq;
    const int regular = st.regular[0];
    const int i0 = chunk * a.chunk;
    const int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
    const int32_t *cy = cx + a.dim_stride;
    const int cx0 = cx[0];
    const int half_pitch = a.g.pitch / 2;
    const int plane_bytes = half_pitch * a.g.win_w;
    if (k_ok) {
        const double2 cs = a.ctrig[(size_t)b * a.nt_stride + k];
        const double off_x = st.off_x, off_y = st.off_y;
        const double2 *ql = a.qlocal + (size_t)b * a.max_n;
        for (int c = threadIdx.x % jobs_pb; c < a.chunk; c += jobs_pb) {
            const int i = i0 + c;
            int o = i < nq ? lookup_offset(ql[i], cs.x, cs.y, off_x, off_y, a.g.scale, a.g.pitch) : 0;
            if (SX == 2 && regular) {
                // window-linear index of hypothesis column 0 for this beam -> (plane, index in plane)
                const int l = o + cx0;
                o = (l >> 1) + (l & 1) * plane_bytes;
            }
            offs[c] = o;
        }
    }
    __syncthreads();
    if (job >= njobs || !k_ok) return;
    const int iy = job / a.ngx, xg = job - iy * a.ngx;
    const int cyv = cy[iy];
    uint16_t *out = a.partial + (size_t)b * a.partial_stride +
                    (((size_t)chunk * a.lat.nt + k) * a.lat.ny + iy) * a.nx_pad + (size_t)xg * G;

    if (regular) {
        const uint8_t *__restrict__ src = SX == 2 ? a.planes + (size_t)b * a.grid_stride : a.grid + (size_t)b * a.grid_stride;
        const uint32_t lane_off = SX == 2 ? (uint32_t)(cyv * half_pitch + xg * G)
                                          : (uint32_t)(cyv * a.g.pitch + cx0 + xg * G);
        uint32_t acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0u;
        if (SX == 2) {
            // Plane loads are made dword-aligned (a byte-unaligned 16-byte load costs the vector L1 ~1.4x the
            // lookups, profiles/r01_c): the lane loads the aligned 16 bytes below its first hypothesis, takes the
            // 17th..19th byte from its right-hand neighbour lane (same row, next 16 hypotheses: DPP wave shift) and
            // funnels by the beam's byte misalignment, which is wave-uniform (v_alignbyte_b32).
            const int lane = threadIdx.x & 63;
            // lanes whose neighbour is not the next group of the same row load the extra dword themselves
            const bool extra = (lane == 63 && xg != a.ngx - 1) || (a.nx_pad - a.lat.nx < 3 && xg == a.ngx - 1);
            for (int c = 0; c < a.chunk; c += U) {
                uint4 w[U];
                uint32_t e[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t ad = lane_off + ((uint32_t)offs[c + u] & ~3u);
                    w[u] = *reinterpret_cast<const uint4 *>(__builtin_assume_aligned(src + ad, 4));
                    e[u] = 0u;
                    if (extra) e[u] = *reinterpret_cast<const uint32_t *>(src + ad + 16);
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t m = (i0 + c + u) < nq ? 0x00FF00FFu : 0u; // wave-uniform
                    const uint32_t rr = (uint32_t)offs[c + u] & 3u;
                    const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[u].x, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
                    const uint32_t w4 = extra ? e[u] : nb;
                    const uint32_t x0 = __builtin_amdgcn_alignbyte(w[u].y, w[u].x, rr), x1 = __builtin_amdgcn_alignbyte(w[u].z, w[u].y, rr);
                    const uint32_t x2 = __builtin_amdgcn_alignbyte(w[u].w, w[u].z, rr), x3 = __builtin_amdgcn_alignbyte(w4, w[u].w, rr);
                    acc[0] += x0 & m; acc[1] += (x0 >> 8) & m;
                    acc[2] += x1 & m; acc[3] += (x1 >> 8) & m;
                    acc[4] += x2 & m; acc[5] += (x2 >> 8) & m;
                    acc[6] += x3 & m; acc[7] += (x3 >> 8) & m;
                }
            }
        } else {
            for (int c = 0; c < a.chunk; c += U) {
                uint4 w[U];
#pragma unroll
                for (int u = 0; u < U; u++) __builtin_memcpy(&w[u], src + (uint32_t)(lane_off + (uint32_t)offs[c + u]), 16);
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t m = (i0 + c + u) < nq ? 0x00FF00FFu : 0u; // wave-uniform
                    acc[0] += w[u].x & m; acc[1] += (w[u].x >> 8) & m;
                    acc[2] += w[u].y & m; acc[3] += (w[u].y >> 8) & m;
                    acc[4] += w[u].z & m; acc[5] += (w[u].z >> 8) & m;
                    acc[6] += w[u].w & m; acc[7] += (w[u].w >> 8) & m;
                }
            }
        }
        store_partial16(out, acc);
        YM_STAMP(a, 9);
    } else {
        // hypothesis cells are not an exact lattice (possible only through fp rounding): per-cell path
        const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
        const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
        const int n_here = min(nq - i0, a.chunk);
        for (int j = 0; j < G; j++) {
            const int ix = xg * G + j;
            unsigned sum = 0;
            if (ix < a.lat.nx) {
                const int base = cyv * a.g.pitch + cx[ix];
                for (int i = 0; i < n_here; i++) {
                    const unsigned idx = (unsigned)(base + offs[i]);
                    sum += idx < limit ? grid[idx] : 0u;
                }
            }
            out[j] = (uint16_t)sum;
        }
    }
}

__device__ __forceinline__ int2 lookup_cell(double2 p, double cosine, double sine, double off_x, double off_y, double scale) {
    const double ox = cosine * p.x - sine * p.y;
    const double oy = sine * p.x + cosine * p.y;
    return make_int2(world_to_grid(ox + off_x, off_x, scale), world_to_grid(oy + off_y, off_y, scale));
}

// ---- LDS-staged coarse correlate.
// The global-load kernel above is bound by the vector L1 (about one lane access per clock per CU, 52+ lane
// accesses per (beam, angle), profiles/r01_b/r01_c).  Consecutive beams of a scan hit neighbouring cells, so the
// patches of a GROUP of 32 consecutive beams overlap: their bounding rectangle holds 4-6x fewer bytes than the
// group gathers.  This kernel copies that rectangle of both column planes into LDS once (aligned 16-byte loads:
// 4-6x fewer L1 accesses) and gathers from LDS, whose read path is 8x wider than the L1's.
//   block   = (item, angle, beam chunk, 64 lane jobs), 4 waves; all waves hold the same 64 lane jobs and split
//             the beams of a group (8 each), their packed sums are added at the end
//   group   = 32 consecutive beams; rectangle = rows Y0..Y0+H of plane bytes Xp0..Xp0+16*spr, LDS pitch 48/80/
//             112/144 B (2*pitch = 32 mod 64: the 26 rows x 2 lanes of a ds_read_b128 fall on distinct banks)
//   pipeline: global loads of group g+1 are in flight (registers) while group g is gathered from buffer g&1, then
//             stored into buffer (g+1)&1; one barrier per group
//   a group whose rectangle does not fit (depth discontinuity inside the group) is gathered with direct loads.
#define YM_ST_GROUP 32
#define YM_ST_PER_WAVE (YM_ST_GROUP / 4)
#define YM_ST_REGION 16384               // bytes per LDS buffer (both planes of one rectangle)
#define YM_ST_NL 4                       // staging loads per thread per plane: rows (tid >> 3) + 32 u, segment tid & 7
#define YM_ST_MAX_GROUPS 16              // chunk <= 512 beams
struct StRect { int Xp0, Y0, H, spr, pitch, use_lds, nvalid; };

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ StRect st_rect_of(const int *s_rect, int ngroups, int g) {
    StRect r;
    r.Xp0 = r.Y0 = r.H = r.spr = r.use_lds = r.nvalid = 0;
    r.pitch = 80;
    if (g < ngroups) {
        const int4 p = *reinterpret_cast<const int4 *>(s_rect + g * 8), q = *reinterpret_cast<const int4 *>(s_rect + g * 8 + 4);
        r.Xp0 = __builtin_amdgcn_readfirstlane(p.x); r.Y0 = __builtin_amdgcn_readfirstlane(p.y);
        r.H = __builtin_amdgcn_readfirstlane(p.z); r.spr = __builtin_amdgcn_readfirstlane(p.w);
        r.pitch = __builtin_amdgcn_readfirstlane(q.x); r.use_lds = __builtin_amdgcn_readfirstlane(q.y);
        r.nvalid = __builtin_amdgcn_readfirstlane(q.z);
    }
    return r;
}
// The staging loads are inline asm on purpose: hipcc otherwise sinks a load whose only use is a conditional LDS
// store down to that store and waits right there, which serialises the pipeline (seen in the ISA).  As asm they
// are issued where written (before the gather of the previous group) and waited for with the explicit
// s_waitcnt after it.  Thread (row = tid >> 3, seg = tid & 7) copies 16-byte block `seg` of rows row, row + 32, ...
// of both planes; lanes outside the rectangle issue nothing.
__device__ __forceinline__ void st_stage_load(u32x4 (&v)[2 * YM_ST_NL], const StRect &r, const uint8_t *planes, int plane_bytes,
                                              int half_pitch) {
    const int seg = threadIdx.x & 7, row0 = threadIdx.x >> 3;
    if (r.use_lds && seg < r.spr) {
        // the funnel's look-ahead block may start past the plane row: re-read the row's last full block instead
        const int xb = min(r.Xp0 + 16 * seg, half_pitch - 16);
        const uint32_t off = (uint32_t)((r.Y0 + row0) * half_pitch + xb);
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            const uint8_t *base = planes + (size_t)pl * plane_bytes;
#pragma unroll
            for (int u = 0; u < YM_ST_NL; u++)
                if (row0 + 32 * u < r.H) {
                    const uint32_t o = off + (uint32_t)(32 * u * half_pitch);
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[pl * YM_ST_NL + u]) : "v"(o), "s"(base) : "memory");
                }
        }
    }
}
// wait for the staging loads, then hand the registers to the compiler
#define YM_ST_WAIT(v)                                                                                          \
    asm volatile("s_waitcnt vmcnt(0)"                                                                          \
                 : "+v"((v)[0]), "+v"((v)[1]), "+v"((v)[2]), "+v"((v)[3]), "+v"((v)[4]), "+v"((v)[5]), "+v"((v)[6]), "+v"((v)[7]) \
                 :                                                                                             \
                 : "memory")
__device__ __forceinline__ void st_stage_store(const u32x4 (&v)[2 * YM_ST_NL], const StRect &r, unsigned char *buf) {
    const int seg = threadIdx.x & 7, row0 = threadIdx.x >> 3;
    if (r.use_lds && seg < r.spr) {
        unsigned char *p = buf + row0 * r.pitch + 16 * seg;
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int u = 0; u < YM_ST_NL; u++)
                if (row0 + 32 * u < r.H)
                    *reinterpret_cast<u32x4 *>(p + (pl * r.H + 32 * u) * r.pitch) = v[pl * YM_ST_NL + u];
    }
}

struct StCtx { // wave/lane constants of the gather
    const uint8_t *planes;
    const int2 *cells;
    int cx0, cy0, iy, xg, wave, lane, plane_bytes, half_pitch, mode;
};
__device__ __forceinline__ void st_accumulate(uint32_t (&acc)[8], uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
    const uint32_t m = 0x00FF00FFu;
    acc[0] += x0 & m; acc[1] += (x0 >> 8) & m;
    acc[2] += x1 & m; acc[3] += (x1 >> 8) & m;
    acc[4] += x2 & m; acc[5] += (x2 >> 8) & m;
    acc[6] += x3 & m; acc[7] += (x3 >> 8) & m;
}
// four beams at once: grid bytes are <= 100, so two beams add without carries as packed bytes; the two pair sums
// are then split into even/odd bytes (v_and / v_perm) and added to the 16-bit lanes with one v_add3 each
__device__ __forceinline__ void st_accumulate4(uint32_t (&acc)[8], const uint32_t (&x)[4][4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t s01 = x[0][j] + x[1][j], s23 = x[2][j] + x[3][j];
        acc[2 * j] = acc[2 * j] + (s01 & 0x00FF00FFu) + (s23 & 0x00FF00FFu);
        acc[2 * j + 1] = acc[2 * j + 1] + __builtin_amdgcn_perm(0u, s01, 0x0c030c01u) + __builtin_amdgcn_perm(0u, s23, 0x0c030c01u);
    }
}
// 16 bytes starting `sh` (0..15, wave-uniform) bytes into the 32 bytes (w0, w1)
#define YM_ST_FUNNEL(w0, w1, sh, x0, x1, x2, x3)                                                                     \
    do {                                                                                                              \
        const int rr_ = (sh) & 3;                                                                                     \
        switch ((sh) >> 2) {                                                                                          \
        case 0:                                                                                                       \
            x0 = __builtin_amdgcn_alignbyte((w0).y, (w0).x, rr_); x1 = __builtin_amdgcn_alignbyte((w0).z, (w0).y, rr_); \
            x2 = __builtin_amdgcn_alignbyte((w0).w, (w0).z, rr_); x3 = __builtin_amdgcn_alignbyte((w1).x, (w0).w, rr_); \
            break;                                                                                                    \
        case 1:                                                                                                       \
            x0 = __builtin_amdgcn_alignbyte((w0).z, (w0).y, rr_); x1 = __builtin_amdgcn_alignbyte((w0).w, (w0).z, rr_); \
            x2 = __builtin_amdgcn_alignbyte((w1).x, (w0).w, rr_); x3 = __builtin_amdgcn_alignbyte((w1).y, (w1).x, rr_); \
            break;                                                                                                    \
        case 2:                                                                                                       \
            x0 = __builtin_amdgcn_alignbyte((w0).w, (w0).z, rr_); x1 = __builtin_amdgcn_alignbyte((w1).x, (w0).w, rr_); \
            x2 = __builtin_amdgcn_alignbyte((w1).y, (w1).x, rr_); x3 = __builtin_amdgcn_alignbyte((w1).z, (w1).y, rr_); \
            break;                                                                                                    \
        default:                                                                                                      \
            x0 = __builtin_amdgcn_alignbyte((w1).x, (w0).w, rr_); x1 = __builtin_amdgcn_alignbyte((w1).y, (w1).x, rr_); \
            x2 = __builtin_amdgcn_alignbyte((w1).z, (w1).y, rr_); x3 = __builtin_amdgcn_alignbyte((w1).w, (w1).z, rr_); \
            break;                                                                                                    \
        }                                                                                                             \
    } while (0)
// gather this wave's beams of group g (beams wave*8 .. wave*8+7 of the group)
__device__ __forceinline__ void st_gather(uint32_t (&acc)[8], const StCtx &c, int g, const StRect &r, const unsigned char *buf) {
    constexpr int G = 16;
    constexpr int HALF = YM_ST_PER_WAVE / 2;
    static_assert(HALF == 4, "st_accumulate4 takes four beams");
    const int first = c.wave * YM_ST_PER_WAVE;
    // lane q (< 32) of every wave prepares beam q of the group: the wave-uniform part of its LDS (or plane) address
    // and its byte shift; the gather loop below picks them up with v_readlane
    const int2 cc = c.cells[g * YM_ST_GROUP + (c.lane & (YM_ST_GROUP - 1))];
    const int col0 = c.cx0 + cc.x;
    if (r.use_lds) {
        // two aligned 16-byte LDS reads per beam, then a wave-uniform byte funnel.  (A single ds_read_b128 at the
        // 4-byte-aligned address + ds_read_b32 is legal on gfx950 and needs no dword switch, but measured slower:
        // the misaligned read is split by the LDS, scripts/exp/lds_unaligned.hip, profiles/r01_c.)
        const int a16 = (col0 >> 1) - r.Xp0;
        const int plane_sz = r.H * r.pitch;
        const int ubase = (col0 & 1) * plane_sz + (c.cy0 + cc.y - r.Y0) * r.pitch + (a16 & ~15);
        const int ush = a16 & 15;
        const int lane_off = (2 * c.iy) * r.pitch + G * c.xg;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint4 w0[HALF], w1[HALF];
            const int bfirst = first + h * HALF;
            if (bfirst >= r.nvalid) break; // wave-uniform
#pragma unroll
            for (int q = 0; q < HALF; q++) {
                const int bb = min(bfirst + q, r.nvalid - 1); // past the end: last beam again, dropped below
                const int addr = lane_off + __builtin_amdgcn_readlane(ubase, bb);
                w0[q] = *reinterpret_cast<const uint4 *>(buf + addr);
                w1[q] = *reinterpret_cast<const uint4 *>(buf + addr + 16);
            }
            uint32_t x[HALF][4];
#pragma unroll
            for (int q = 0; q < HALF; q++) {
                const int sh = __builtin_amdgcn_readlane(ush, min(bfirst + q, r.nvalid - 1));
                YM_ST_FUNNEL(w0[q], w1[q], sh, x[q][0], x[q][1], x[q][2], x[q][3]);
            }
            if (bfirst + HALF <= r.nvalid) {
                st_accumulate4(acc, x);
            } else {
#pragma unroll
                for (int q = 0; q < HALF; q++)
                    if (bfirst + q < r.nvalid) st_accumulate(acc, x[q][0], x[q][1], x[q][2], x[q][3]);
            }
        }
    } else if (c.mode != 2) {
        // direct (byte-unaligned) plane loads for a group whose rectangle does not fit: all of this wave's beams
        // in flight at once -- these waves are bound by the vector L1 while the staged ones are bound by issue,
        // so the two kinds overlap on a CU
        const int ubase = (col0 & 1) * c.plane_bytes + (c.cy0 + cc.y) * c.half_pitch + (col0 >> 1);
        const uint32_t lane_off = (uint32_t)((2 * c.iy) * c.half_pitch + G * c.xg);
        if (first < r.nvalid) {
            uint4 w[YM_ST_PER_WAVE];
#pragma unroll
            for (int q = 0; q < YM_ST_PER_WAVE; q++) {
                const int bb = min(first + q, r.nvalid - 1);
                __builtin_memcpy(&w[q], c.planes + (lane_off + (uint32_t)__builtin_amdgcn_readlane(ubase, bb)), 16);
            }
            if (first + YM_ST_PER_WAVE <= r.nvalid) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    uint32_t x[HALF][4];
#pragma unroll
                    for (int q = 0; q < HALF; q++) { x[q][0] = w[h * HALF + q].x; x[q][1] = w[h * HALF + q].y; x[q][2] = w[h * HALF + q].z; x[q][3] = w[h * HALF + q].w; }
                    st_accumulate4(acc, x);
                }
            } else {
#pragma unroll
                for (int q = 0; q < YM_ST_PER_WAVE; q++)
                    if (first + q < r.nvalid) st_accumulate(acc, w[q].x, w[q].y, w[q].z, w[q].w);
            }
        }
    }
}

// grid (ceil(njobs / 64), nt * n_chunks, B), 256 threads, SX == 2 only
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void correlate_staged_kernel(CorrArgs a) {
    constexpr int G = 16;
    __shared__ __attribute__((aligned(16))) unsigned char region[2 * YM_ST_REGION];
    __shared__ int2 s_cells[512];
    __shared__ int s_box[YM_ST_MAX_GROUPS * 4];
    __shared__ int s_rect[YM_ST_MAX_GROUPS * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    const int b = xcd_item_of_block(bx, by);
    const int k = by % a.lat.nt, chunk = by / a.lat.nt;
    const int njobs = a.lat.ny * a.ngx;
    const int j_first = bx * 64, j_last = min(njobs, j_first + 64) - 1;
    const bool active = j_first + lane <= j_last;
    const int job = min(j_first + lane, j_last); // idle lanes shadow the last job (valid addresses, result dropped)
    const int iy = job / a.ngx, xg = job - iy * a.ngx;
    const YmItemState &st = a.states[b];
    const int nq = st.nq;
    const int i0 = chunk * a.chunk;
    const int32_t *cxp = a.hypcell + (size_t)b * 2 * a.dim_stride;
    const int cx0 = cxp[0], cy0 = cxp[a.dim_stride];
    const int half_pitch = a.g.pitch / 2;
    const int plane_bytes = half_pitch * a.g.win_w;
    const uint8_t *__restrict__ planes = a.planes + (size_t)b * a.grid_stride;
    const int iy_min = j_first / a.ngx, iy_max = j_last / a.ngx;
    const int xg_min = iy_min == iy_max ? j_first % a.ngx : 0, xg_max = iy_min == iy_max ? j_last % a.ngx : a.ngx - 1;
    const int ngroups = (a.chunk + YM_ST_GROUP - 1) / YM_ST_GROUP;

    YM_STAMP(a, 20);
    // ---- cells of the whole chunk (GridIndexLookup::ComputeOffsets for this angle), bounding box per group
    if (tid < YM_ST_MAX_GROUPS * 4) s_box[tid] = (tid & 1) ? INT32_MIN : INT32_MAX; // [g][x0, x1, y0, y1]
    __syncthreads();
    {
        const double2 cs = a.ctrig[(size_t)b * a.nt_stride + k];
        const double off_x = st.off_x, off_y = st.off_y;
        const double2 *ql = a.qlocal + (size_t)b * a.max_n;
        for (int c = tid; c < a.chunk; c += 256) {
            const int i = i0 + c;
            int2 cell = make_int2(0, 0);
            if (i < nq) {
                cell = lookup_cell(ql[i], cs.x, cs.y, off_x, off_y, a.g.scale);
                int *bx = s_box + (c / YM_ST_GROUP) * 4;
                atomicMin(bx + 0, cell.x); atomicMax(bx + 1, cell.x);
                atomicMin(bx + 2, cell.y); atomicMax(bx + 3, cell.y);
            }
            s_cells[c] = cell;
        }
    }
    __syncthreads();
    if (tid < ngroups) {
        const int g0 = tid * YM_ST_GROUP;
        const int nvalid = max(0, min(min(YM_ST_GROUP, a.chunk - g0), nq - (i0 + g0)));
        int Xp0 = 0, Y0 = 0, H = 0, spr = 0, pitch = 48, use_lds = 0;
        if (nvalid > 0) {
            const int *bx = s_box + tid * 4;
            const int X0 = cx0 + bx[0] + 2 * G * xg_min, X1 = cx0 + bx[1] + 2 * G * xg_max + 2 * (G - 1);
            Y0 = cy0 + 2 * iy_min + bx[2];
            H = cy0 + 2 * iy_max + bx[3] - Y0 + 1;
            Xp0 = (X0 >> 1) & ~15;
            spr = ((((X1 >> 1) - Xp0 + 1) + 15) >> 4) + 1; // +1: the funnel reads one block ahead
            pitch = spr <= 3 ? 48 : spr <= 5 ? 80 : spr <= 7 ? 112 : 144;
            use_lds = (spr <= 8 && H <= 32 * YM_ST_NL && 2 * H * pitch <= YM_ST_REGION && nvalid >= YM_ST_GROUP / 2) ? 1 : 0;
        }
        int *r = s_rect + tid * 8;
        r[0] = Xp0; r[1] = Y0; r[2] = H; r[3] = spr; r[4] = pitch; r[5] = use_lds; r[6] = nvalid;
        if (a.stamps && nvalid > 0) { // development statistics: groups, staged groups, staged bytes
            atomicAdd(a.stamps + 29, 1ull);
            atomicAdd(a.stamps + 30, (unsigned long long)use_lds);
            atomicAdd(a.stamps + 31, (unsigned long long)(use_lds ? 2 * H * spr * 16 : 0));
        }
    }
    __syncthreads();
#define rect_of(S) st_rect_of(s_rect, ngroups, (S))
    YM_STAMP(a, 21);

    uint32_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0u;

    StCtx ctx;
    ctx.planes = planes; ctx.cells = s_cells; ctx.cx0 = cx0; ctx.cy0 = cy0; ctx.iy = iy; ctx.xg = xg; ctx.wave = wave; ctx.lane = lane; ctx.mode = a.tpb;
    ctx.plane_bytes = plane_bytes; ctx.half_pitch = half_pitch;
    if (st.regular[0]) {
        u32x4 v[2 * YM_ST_NL]; // the next group's rectangle, in flight while the current one is gathered
        StRect rc = rect_of(0);
        st_stage_load(v, rc, planes, plane_bytes, half_pitch);
        YM_ST_WAIT(v);
        st_stage_store(v, rc, region);
        __syncthreads();
        for (int g = 0; g < ngroups && rc.nvalid > 0; g++) {
            const StRect rn = rect_of(g + 1);
            if (a.tpb != 4) st_stage_load(v, rn, planes, plane_bytes, half_pitch);
            if (a.tpb != 3) st_gather(acc, ctx, g, rc, region + (g & 1) * YM_ST_REGION);
            YM_ST_WAIT(v);
            st_stage_store(v, rn, region + ((g + 1) & 1) * YM_ST_REGION);
            __syncthreads();
            rc = rn;
        }
#undef rect_of
        YM_STAMP(a, 22);
        // ---- add the four waves' packed 16-bit sums, wave 0 writes the partials
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        uint32_t *red = reinterpret_cast<uint32_t *>(region);
#pragma unroll
        for (int j = 0; j < 8; j++) red[(wave * 8 + j) * 64 + lane] = acc[j];
        __syncthreads();
        if (wave == 0 && active) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = red[j * 64 + lane] + red[(8 + j) * 64 + lane] + red[(16 + j) * 64 + lane] + red[(24 + j) * 64 + lane];
            uint16_t *out = a.partial + (size_t)b * a.partial_stride +
                            (((size_t)chunk * a.lat.nt + k) * a.lat.ny + iy) * a.nx_pad + (size_t)xg * G;
            store_partial16(out, acc);
        }
        YM_STAMP(a, 23);
    } else {
        // hypothesis cells are not an exact lattice (possible only through fp rounding): per-cell path
        if (wave != 0 || !active) return;
        const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
        const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
        const int32_t *cyp = cxp + a.dim_stride;
        uint16_t *out = a.partial + (size_t)b * a.partial_stride +
                        (((size_t)chunk * a.lat.nt + k) * a.lat.ny + iy) * a.nx_pad + (size_t)xg * G;
        const int n_here = min(nq - i0, a.chunk);
        for (int j = 0; j < G; j++) {
            const int ix = xg * G + j;
            unsigned sum = 0;
            if (ix < a.lat.nx) {
                const int base = cyp[iy] * a.g.pitch + cxp[ix];
                for (int i = 0; i < n_here; i++) {
                    const unsigned idx = (unsigned)(base + s_cells[i].x + s_cells[i].y * a.g.pitch);
                    sum += idx < limit ? grid[idx] : 0u;
                }
            }
            out[j] = (uint16_t)sum;
        }
    }
}

// ================================================================== K5a score
#define YM_SCORE_THREADS 256
struct ScoreArgs {
    YmGeom g;
    YmLattice lat;
    const uint16_t *partial;
    size_t partial_stride;
    const YmItemState *states;
    uint32_t *sums;       // [B][nt][ny][nx]
    size_t sums_stride;
    double *resp;         // [B][nt][ny][nx]
    double *blockmax;     // [B][n_blocks]
    unsigned long long *probs; // [B][ny*nx] bit patterns of non-negative doubles, zeroed by prepare_kernel
    size_t probs_stride;
    int32_t n_chunks, nx_pad, n_blocks;
    unsigned long long *stamps;
};

__device__ __forceinline__ double hyp_response(const YmGeom &g, int penalize, unsigned sum, int nq, double sq_dist,
                                               double angle, double center_t) {
    double response = 0.0;
    if (nq != 0) {
        response = (double)sum;
        response /= (double)(nq * YM_OCCUPIED);
    }
    if (penalize && !kt_double_equal(response, 0.0)) {
        double dp = 1.0 - (YM_PENALTY_GAIN * sq_dist / g.dist_var);
        dp = dp > g.min_dist_pen ? dp : g.min_dist_pen;
        const double sq_ang = (angle - center_t) * (angle - center_t);
        double ap = 1.0 - (YM_PENALTY_GAIN * sq_ang / g.ang_var);
        ap = ap > g.min_ang_pen ? ap : g.min_ang_pen;
        response *= (dp * ap);
    }
    return response;
}

// one thread per coarse hypothesis: add the beam-chunk partials, normalise, penalise.
// grid (n_blocks, B)
__global__ __launch_bounds__(YM_SCORE_THREADS) void score_kernel(ScoreArgs a) {
    __shared__ double scratch[16];
    const int b = blockIdx.y;
    const YmItemState &st = a.states[b];
    const int nx = a.lat.nx, ny = a.lat.ny, nt = a.lat.nt, nxy = nx * ny;
    const int h = blockIdx.x * YM_SCORE_THREADS + threadIdx.x;
    double r = -1.0;
    YM_STAMP(a, 10);
    if (h < nxy * nt) {
        const int k = h / nxy, c = h - k * nxy;
        const int iy = c / nx, ix = c - iy * nx;
        const uint16_t *p = a.partial + (size_t)b * a.partial_stride + ((size_t)k * ny + iy) * a.nx_pad + ix;
        const size_t cstride = (size_t)nt * ny * a.nx_pad;
        unsigned sum = 0;
#pragma unroll 8
        for (int c2 = 0; c2 < a.n_chunks; c2++) sum += p[(size_t)c2 * cstride];
        const double x = -a.lat.off_x + ix * a.lat.step_x, y = -a.lat.off_y + iy * a.lat.step_y;
        const double ct = st.center[2];
        const double angle = (ct - a.lat.angle_off) + k * a.lat.angle_res;
        r = hyp_response(a.g, a.lat.penalize, sum, st.nq, x * x + y * y, angle, ct);
        a.sums[(size_t)b * a.sums_stride + h] = sum;
        a.resp[(size_t)b * a.sums_stride + h] = r;
        // search-space probability grid: max over theta per (x, y).  Responses are >= 0, so the
        // u64 order of the bit patterns is the numeric order and an integer atomic max is exact.
        if (r > 0.0) atomicMax(&a.probs[(size_t)b * a.probs_stride + c], (unsigned long long)__double_as_longlong(r));
    }
    const double m = block_reduce(r, OpMaxD(), -1.0, scratch);
    if (threadIdx.x == 0) a.blockmax[(size_t)b * a.n_blocks + blockIdx.x] = m;
    YM_STAMP(a, 11);
}

// ================================================================== K6 finish
#define YM_FINISH_THREADS 256
#define YM_MAX_FINE_HYP 4096
struct FinishArgs {
    YmGeom g;
    YmLattice lc, lf;
    int32_t refine;
    int32_t max_n, nt_stride, n_blocks;
    YmItemState *states;
    YmItemState *host_out;    // pinned host memory, written directly (nullable)
    const double *resp;       // coarse responses [B][nt][ny][nx]
    size_t sums_stride;
    const double *blockmax;   // [B][n_blocks] maxima of YM_SCORE_THREADS consecutive responses
    const double *probs;      // [B][ny*nx] max over theta per (x, y)  (m_pSearchSpaceProbs)
    size_t probs_stride;
    const uint8_t *grid;
    size_t grid_stride;
    const double2 *qlocal;
    int32_t *foffsets;        // [B][nt_f][max_n] fine lookup table (scratch)
    uint32_t *fsums;          // [B][nt_f*ny_f*nx_f] fine sums (kept for parity tests)
    size_t fsums_stride;
    unsigned long long *stamps;
};

// sum N doubles across the block in one round (2 barriers); result in every thread
// Floating-point sums that reach the result (tie means, covariances) are accumulated by the first YM_CANON threads
// only, element e by thread e % YM_CANON in increasing e: the kernels that share this code run with 256, 512 or 1024
// threads and must produce the same bits.  (block_sum_vec adds the waves in order; idle waves contribute exact zeros.)
#define YM_CANON 256
template <int N>
__device__ __forceinline__ void block_sum_vec(double (&v)[N], double *scratch /* >= 16*N */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int j = 0; j < N; j++) v[j] = wave_reduce(v[j], OpAddD());
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int j = 0; j < N; j++) scratch[w * N + j] = v[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < N; j++) {
        double r = 0.0;
        for (int i = 0; i < nw; i++) r += scratch[i * N + j];
        v[j] = r;
    }
}

// tie-set mean of CorrelateScan: accumulate one hypothesis
__device__ __forceinline__ void tie_accumulate(double (&acc)[5], const YmLattice &L, int h, double cxw, double cyw,
                                               double start_angle) {
    const int nxy = L.nx * L.ny;
    const int k = h / nxy, c = h - k * nxy, iy = c / L.nx, ix = c - iy * L.nx;
    const double x = -L.off_x + ix * L.step_x, y = -L.off_y + iy * L.step_y;
    const double hd = kt_normalize_angle(start_angle + k * L.angle_res);
    acc[0] += cxw + x; acc[1] += cyw + y;
    acc[2] += cos(hd); acc[3] += sin(hd);
    acc[4] += 1.0;
}

// Coarse tail of CorrelateScan for one item: best response, mean of all hypotheses with
// DoubleEqual(response, best).  Runs redundantly in every block that needs the coarse mean.
// Returns best (unclamped); mean[] and *status valid in every thread.
template <int NT>
__device__ __forceinline__ double coarse_best_and_mean(const YmLattice &L, const double *resp, const double *bm,
                                                       int n_blocks, const double pose[3], double mean[3], int *status,
                                                       double *scratch /* >= 80 */, int *s_list /* NT */, int *s_tmp /* NT */,
                                                       int *s_nlist) {
    const int tid = threadIdx.x;
    const int nh = L.nx * L.ny * L.nt;
    const double start_angle = pose[2] - L.angle_off;
    if (tid == 0) *s_nlist = 0;
    double lb = -1.0;
    for (int i0 = tid; i0 < n_blocks; i0 += 4 * NT) {
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = (i0 + u * NT) < n_blocks ? bm[i0 + u * NT] : -1.0;
#pragma unroll
        for (int u = 0; u < 4; u++) lb = v[u] > lb ? v[u] : lb;
    }
    const double best = block_reduce(lb, OpMaxD(), -1.0, scratch);
    // score blocks that can hold a hypothesis with DoubleEqual(response, best), in ASCENDING order: which thread sums
    // which hypothesis must not depend on a race.  Unordered compaction by atomics, then every entry finds its
    // rank among the (few) others.
    int overflow = 0;
    for (int i = tid; i < n_blocks; i += NT)
        if (bm[i] >= best - YM_KT_TOLERANCE) {
            const int at = atomicAdd(s_nlist, 1);
            if (at < NT) s_tmp[at] = i; else overflow = 1;
        }
    overflow = __syncthreads_or(overflow);
    if (!overflow) {
        const int n = *s_nlist;
        if (tid < n) {
            const int mine = s_tmp[tid];
            int rank = 0;
            for (int j = 0; j < n; j++) rank += s_tmp[j] < mine ? 1 : 0;
            s_list[rank] = mine;
        }
        __syncthreads();
    }
    double acc[5] = {0, 0, 0, 0, 0};
    if (tid < YM_CANON) {
        if (!overflow) {
            const int nlist = *s_nlist;
            for (int w = tid; w < nlist * YM_SCORE_THREADS; w += YM_CANON) {
                const int h = s_list[w / YM_SCORE_THREADS] * YM_SCORE_THREADS + (w % YM_SCORE_THREADS);
                if (h < nh && kt_double_equal(resp[h], best)) tie_accumulate(acc, L, h, pose[0], pose[1], start_angle);
            }
        } else {
            for (int h = tid; h < nh; h += YM_CANON)
                if (kt_double_equal(resp[h], best)) tie_accumulate(acc, L, h, pose[0], pose[1], start_angle);
        }
    }
    block_sum_vec<5>(acc, scratch);
    if (acc[4] > 0.0) {
        const double cnt = acc[4]; // exact small integer, same value as Karto's int count
        mean[0] = acc[0] / cnt; mean[1] = acc[1] / cnt;
        mean[2] = atan2(acc[3] / cnt, acc[2] / cnt);
    } else {
        mean[0] = mean[1] = mean[2] = 0.0;
        *status = -5; // "Unable to find best position"
    }
    return best;
}

// ScanMatcher::ComputePositionalCovariance over the per-(x,y) maxima of the coarse pass; cov valid in every thread
template <int NT>
__device__ __forceinline__ void positional_covariance(const FinishArgs &a, int b, const YmItemState &st, const double mean[3],
                                                      double best, double cov[9], double *scratch) {
    const int tid = threadIdx.x;
    for (int i = 0; i < 9; i++) cov[i] = (i % 4 == 0) ? 1.0 : 0.0;
    const YmLattice &L = a.lc;
    const int nx = L.nx, nxy = nx * L.ny;
    const double cxw = st.pose[0], cyw = st.pose[1];
    const double start_x = -L.off_x, start_y = -L.off_y;
    double sums[4] = {0, 0, 0, 0};
    const double dx = mean[0] - cxw, dy = mean[1] - cyw;
    if (!(best < YM_KT_TOLERANCE)) {
        const double *probs = a.probs + (size_t)b * a.probs_stride;
        // eight loads in flight; the additions stay in increasing cell order (the canonical order above)
        for (int c0 = tid; c0 < nxy && tid < YM_CANON; c0 += 8 * YM_CANON) {
            double pv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) pv[u] = (c0 + u * YM_CANON) < nxy ? probs[c0 + u * YM_CANON] : -1.0;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int c = c0 + u * YM_CANON;
                const double response = pv[u];
                if (c < nxy && response >= (best - 0.1)) {
                    const int iy = c / nx, ix = c - iy * nx;
                    const double x = start_x + ix * L.step_x, y = start_y + iy * L.step_y;
                    sums[0] += response;
                    sums[1] += ((x - dx) * (x - dx)) * response;
                    sums[2] += ((x - dx) * (y - dy) * response);
                    sums[3] += ((y - dy) * (y - dy)) * response;
                }
            }
        }
    }
    block_sum_vec<4>(sums, scratch);
    if (best < YM_KT_TOLERANCE) {
        cov[0] = YM_MAX_VARIANCE; cov[4] = YM_MAX_VARIANCE;
        cov[8] = 4 * (L.angle_res * L.angle_res);
    } else {
        const double norm = sums[0];
        if (norm > YM_KT_TOLERANCE) {
            double vxx = sums[1] / norm, vxy = sums[2] / norm, vyy = sums[3] / norm;
            const double vthth = 4 * (L.angle_res * L.angle_res);
            const double min_xx = 0.1 * (L.step_x * L.step_x);
            const double min_yy = 0.1 * (L.step_y * L.step_y);
            vxx = vxx > min_xx ? vxx : min_xx;
            vyy = vyy > min_yy ? vyy : min_yy;
            const double mult = 1.0 / best;
            cov[0] = vxx * mult; cov[1] = vxy * mult; cov[3] = vxy * mult; cov[4] = vyy * mult;
            cov[8] = vthth;
        }
        if (kt_double_equal(cov[0], 0.0)) cov[0] = YM_MAX_VARIANCE;
        if (kt_double_equal(cov[4], 0.0)) cov[4] = YM_MAX_VARIANCE;
    }
}

// ---- K6a fine: grid (nt_f + 1, B) (or (1, B) without refinement).  Block k < nt_f scores the 3x3 fine lattice for
// fine angle k; the extra block computes the coarse pass's positional covariance at the same time.
#define YM_FINE_THREADS 512
__global__ __launch_bounds__(YM_FINE_THREADS) void fine_kernel(FinishArgs a) {
    constexpr int NT = YM_FINE_THREADS;
    __shared__ double scratch[16 * 5];
    __shared__ int s_list[NT], s_tmp[NT];
    __shared__ int s_nlist;
    __shared__ double s_cs[2];
    __shared__ int s_cx[64], s_cy[64];
    __shared__ unsigned s_sum[YM_MAX_FINE_HYP];
    int k;
    const int b = xcd_item_of_block_2d(k); // the blocks of an item share its grid patch: keep them on one XCD
    const int tid = threadIdx.x, lane = tid & 63;
    YM_STAMP(a, 12);
    YmItemState &st = a.states[b];
    const int nq = st.nq;
    if (nq == 0) return;
    const double pose[3] = {st.pose[0], st.pose[1], st.pose[2]};
    const double off_x = st.off_x, off_y = st.off_y;
    double mean[3];
    int status = 0;
    const double best = coarse_best_and_mean<NT>(a.lc, a.resp + (size_t)b * a.sums_stride,
                                                 a.blockmax + (size_t)b * a.n_blocks, a.n_blocks, pose, mean, &status,
                                                 scratch, s_list, s_tmp, &s_nlist);
    if (k == (a.refine ? a.lf.nt : 0)) { // extra block: coarse result + positional covariance for final_kernel
        double cov[9];
        positional_covariance<NT>(a, b, st, mean, best, cov, scratch);
        if (tid == 0) {
            st.center[0] = mean[0]; st.center[1] = mean[1]; st.center[2] = mean[2];
            st.coarse_response = best; // unclamped
            st.status = status;
            for (int i = 0; i < 9; i++) st.cov[i] = cov[i];
        }
        return;
    }
    YM_STAMP(a, 13);

    const YmLattice &L = a.lf;
    const int nx = L.nx, ny = L.ny, nxy = nx * ny;
    const double start_x = -L.off_x, start_y = -L.off_y;
    if (tid == 0) {
        const double angle = (mean[2] - L.angle_off) + k * L.angle_res;
        s_cs[0] = cos(angle);
        s_cs[1] = sin(angle);
    }
    for (int i = tid; i < nx; i += NT) s_cx[i] = hyp_cell(mean[0], start_x, i, L.step_x, off_x, a.g);
    for (int i = tid; i < ny; i += NT) s_cy[i] = hyp_cell(mean[1], start_y, i, L.step_y, off_y, a.g);
    for (int h = tid; h < nxy; h += NT) s_sum[h] = 0u;
    __syncthreads();
    YM_STAMP(a, 14);
    const double cosine = s_cs[0], sine = s_cs[1];
    const uint8_t *grid = a.grid + (size_t)b * a.grid_stride;
    const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
    int32_t *foff = a.foffsets + ((size_t)b * a.nt_stride + k) * a.max_n;
    const double2 *ql = a.qlocal + (size_t)b * a.max_n;
    const bool block3 = nx == 3 && ny == 3 && s_cx[1] == s_cx[0] + 1 && s_cx[2] == s_cx[0] + 2 &&
                        s_cy[1] == s_cy[0] + 1 && s_cy[2] == s_cy[0] + 2;
    if (block3) {
        // Karto's fine lattice is always 3x3 cells: a lane reads the 3x3 cell block under its beam
        // as three 4-byte words.
        const uint32_t base0 = (uint32_t)(s_cy[0] * a.g.pitch + s_cx[0]);
        unsigned acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid; i < nq; i += 4 * NT) {
            uint32_t w[4][3];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int ii = i + u * NT;
                int off = 0;
                if (ii < nq) {
                    off = lookup_offset(ql[ii], cosine, sine, off_x, off_y, a.g.scale, a.g.pitch);
                    foff[ii] = off;
                }
                const uint32_t idx = base0 + (uint32_t)off;
#pragma unroll
                for (int r = 0; r < 3; r++) __builtin_memcpy(&w[u][r], grid + (uint32_t)(idx + r * a.g.pitch), 4);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t m = (i + u * NT) < nq ? 0xffu : 0u;
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    acc[3 * r] += w[u][r] & m; acc[3 * r + 1] += (w[u][r] >> 8) & m; acc[3 * r + 2] += (w[u][r] >> 16) & m;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 9; j++) acc[j] = wave_reduce(acc[j], OpAddU());
        if (lane == 0)
#pragma unroll
            for (int j = 0; j < 9; j++) atomicAdd(&s_sum[j], acc[j]);
    } else {
        // generic lattice: per beam, every (iy, ix) cell
        for (int i = tid; i < nq; i += NT) {
            const int off = lookup_offset(ql[i], cosine, sine, off_x, off_y, a.g.scale, a.g.pitch);
            foff[i] = off;
            for (int c = 0; c < nxy; c++) {
                const int iy = c / nx, ix = c - iy * nx;
                const unsigned idx = (unsigned)(s_cy[iy] * a.g.pitch + s_cx[ix] + off);
                if (idx < limit) atomicAdd(&s_sum[c], (unsigned)grid[idx]);
            }
        }
    }
    __syncthreads();
    uint32_t *fs = a.fsums + (size_t)b * a.fsums_stride + (size_t)k * nxy;
    for (int h = tid; h < nxy; h += NT) fs[h] = s_sum[h];
    YM_STAMP(a, 15);
}

// ---- K6b final: grid (B), 256 threads: fine arg-max / mean, angular covariance, result.
__global__ __launch_bounds__(YM_FINISH_THREADS) void final_kernel(FinishArgs a) {
    constexpr int NT = YM_FINISH_THREADS;
    __shared__ double scratch[16 * 5];
    __shared__ double s_fresp[YM_MAX_FINE_HYP];
    __shared__ unsigned s_asum[YM_MAX_FINE_NT];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63;
    YM_STAMP(a, 16);
    YmItemState &st = a.states[b];
    const int nq = st.nq;
    if (nq == 0) {
        // MatchScan: "scan has no readings; cannot do scan matching" -> pose, maximum covariance, 0
        if (tid == 0) {
            for (int i = 0; i < 9; i++) st.cov[i] = 0.0;
            st.cov[0] = YM_MAX_VARIANCE; st.cov[4] = YM_MAX_VARIANCE;
            st.cov[8] = 4 * (a.lc.angle_res * a.lc.angle_res);
            for (int i = 0; i < 3; i++) { st.mean[i] = st.pose[i]; st.center[i] = st.pose[i]; }
            st.response = 0.0;
            st.coarse_response = 1.0; // nothing to retry with a wider angle
            if (a.host_out) a.host_out[b] = st;
        }
        return;
    }
    const uint8_t *grid = a.grid + (size_t)b * a.grid_stride;
    const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
    const double off_x = st.off_x, off_y = st.off_y;
    double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double mean[3] = {st.center[0], st.center[1], st.center[2]}; // coarse mean (fine_kernel, block 0)
    double best = st.coarse_response;                             // coarse best, unclamped
    int status = st.status;

    // positional covariance of the coarse pass: computed by fine_kernel's extra block
    for (int i = 0; i < 9; i++) cov[i] = st.cov[i];
    const double coarse_response = best > 1.0 ? 1.0 : best;
    double response = coarse_response;

    // ------------------------------------------------------------- fine tail (CorrelateScan, doingFineMatch)
    if (a.refine) {
        const YmLattice &L = a.lf;
        const int nx = L.nx, ny = L.ny, nt = L.nt, nxy = nx * ny, nh = nxy * nt;
        const double cxw = mean[0], cyw = mean[1], ct = mean[2];
        const double start_x = -L.off_x, start_y = -L.off_y, start_angle = ct - L.angle_off;
        const uint32_t *fs = a.fsums + (size_t)b * a.fsums_stride;
        double lb = -1.0;
        for (int h = tid; h < nh; h += NT) {
            const int k = h / nxy, c = h - k * nxy, iy = c / nx, ix = c - iy * nx;
            const double x = start_x + ix * L.step_x, y = start_y + iy * L.step_y;
            const double r = hyp_response(a.g, L.penalize, fs[h], nq, x * x + y * y, start_angle + k * L.angle_res, ct);
            s_fresp[h] = r;
            lb = r > lb ? r : lb;
        }
        for (int k = tid; k < nt; k += NT) s_asum[k] = 0u;
        best = block_reduce(lb, OpMaxD(), -1.0, scratch);
        double acc[5] = {0, 0, 0, 0, 0};
        for (int h = tid; h < nh && tid < YM_CANON; h += YM_CANON)
            if (kt_double_equal(s_fresp[h], best)) tie_accumulate(acc, L, h, cxw, cyw, start_angle);
        block_sum_vec<5>(acc, scratch);
        if (acc[4] > 0.0) {
            const double cnt = acc[4];
            mean[0] = acc[0] / cnt; mean[1] = acc[1] / cnt;
            mean[2] = atan2(acc[3] / cnt, acc[2] / cnt);
        } else {
            status = -5;
        }
        YM_STAMP(a, 17);
        // ComputeAngularCovariance: re-score every fine angle at the cell of the mean pose
        const double best_angle = kt_normalize_angle_difference(mean[2], ct);
        const int gx = world_to_grid(mean[0], off_x, a.g.scale) + a.g.border - a.g.win_origin;
        const int gy = world_to_grid(mean[1], off_y, a.g.scale) + a.g.border - a.g.win_origin;
        const int base = gy * a.g.pitch + gx;
        const int32_t *foff = a.foffsets + (size_t)b * a.nt_stride * a.max_n;
        __syncthreads(); // s_asum cleared above
        // GetResponse(angle k, cell of the mean pose) uses the fine pass's own lookup offsets, so when that cell is one
        // of the fine lattice's cells (always, unless fp rounding puts the tie mean outside) the sum IS the fine
        // pass's integer sum for (k, that cell): take it instead of gathering the scan again.
        int hit_x = -1, hit_y = -1;
        for (int i = 0; i < nx; i++)
            if (hyp_cell(cxw, start_x, i, L.step_x, off_x, a.g) == gx) hit_x = i;
        for (int i = 0; i < ny; i++)
            if (hyp_cell(cyw, start_y, i, L.step_y, off_y, a.g) == gy) hit_y = i;
        if (hit_x >= 0 && hit_y >= 0) {
            for (int k = tid; k < nt; k += NT) s_asum[k] = fs[(size_t)k * nxy + hit_y * nx + hit_x];
        } else {
            // work item = (angle, beam); beams padded to whole waves so that a wave shares one angle
            const int nq_pad = (nq + 63) & ~63;
            const int total = nt * nq_pad;
            for (int w0 = 0; w0 < total; w0 += 8 * NT) {
                int kk[8];
                unsigned idx[8], v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int w = w0 + u * NT + tid;
                    kk[u] = w < total ? w / nq_pad : -1; // wave-uniform
                    const int i = w - kk[u] * nq_pad;
                    idx[u] = (kk[u] >= 0 && i < nq) ? (unsigned)(base + foff[(size_t)kk[u] * a.max_n + i]) : limit;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = idx[u] < limit ? grid[idx[u]] : 0u;
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const unsigned sum = wave_reduce(v[u], OpAddU());
                    if (lane == 0 && kk[u] >= 0) atomicAdd(&s_asum[kk[u]], sum);
                }
            }
        }
        __syncthreads();
        double norm = 0.0, accv = 0.0;
        for (int k = 0; k < nt; k++) {
            const double angle = start_angle + k * L.angle_res;
            double r = (double)s_asum[k];
            r /= (double)(nq * YM_OCCUPIED);
            if (r >= (best - 0.1)) {
                norm += r;
                accv += ((angle - best_angle) * (angle - best_angle)) * r;
            }
        }
        if (norm > YM_KT_TOLERANCE) {
            if (accv < YM_KT_TOLERANCE) accv = L.angle_res * L.angle_res;
            accv /= norm;
        } else {
            accv = 1000 * (L.angle_res * L.angle_res);
        }
        cov[8] = accv;
        response = best > 1.0 ? 1.0 : best;
    }
    if (tid == 0) {
        for (int i = 0; i < 9; i++) st.cov[i] = cov[i];
        for (int i = 0; i < 3; i++) { st.mean[i] = mean[i]; st.center[i] = mean[i]; }
        st.response = response;
        st.coarse_response = coarse_response;
        st.status = status;
        if (a.host_out) a.host_out[b] = st;
    }
    YM_STAMP(a, 19);
}

// ---- K6 finish, one block per item (batches): everything fine_kernel + final_kernel do, without the eleven-fold
// recomputation of the coarse arg-max that one-block-per-fine-angle costs.  Wave w scores the 3x3 fine lattice for
// fine angles w, w + 16, ...; the fine sums stay in LDS.  grid (B), 1024 threads.
#define YM_FINISH1_THREADS 1024
__global__ __launch_bounds__(YM_FINISH1_THREADS) void finish_kernel(FinishArgs a) {
    constexpr int NT = YM_FINISH1_THREADS, NW = NT / 64;
    __shared__ double scratch[16 * 5];
    __shared__ int s_list[NT], s_tmp[NT];
    __shared__ int s_nlist;
    __shared__ double2 s_cs[YM_MAX_FINE_NT];
    __shared__ int s_cx[64], s_cy[64];
    __shared__ unsigned s_sum[YM_MAX_FINE_HYP];
    __shared__ double s_fresp[YM_MAX_FINE_HYP];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    YmItemState &st = a.states[b];
    const int nq = st.nq;
    if (nq == 0) {
        // MatchScan: "scan has no readings; cannot do scan matching" -> pose, maximum covariance, 0
        if (tid == 0) {
            for (int i = 0; i < 9; i++) st.cov[i] = 0.0;
            st.cov[0] = YM_MAX_VARIANCE; st.cov[4] = YM_MAX_VARIANCE;
            st.cov[8] = 4 * (a.lc.angle_res * a.lc.angle_res);
            for (int i = 0; i < 3; i++) { st.mean[i] = st.pose[i]; st.center[i] = st.pose[i]; }
            st.response = 0.0;
            st.coarse_response = 1.0; // nothing to retry with a wider angle
            if (a.host_out) a.host_out[b] = st;
        }
        return;
    }
    const double pose[3] = {st.pose[0], st.pose[1], st.pose[2]};
    const double off_x = st.off_x, off_y = st.off_y;
    double mean[3], cov[9];
    int status = 0;
    double best = coarse_best_and_mean<NT>(a.lc, a.resp + (size_t)b * a.sums_stride, a.blockmax + (size_t)b * a.n_blocks,
                                           a.n_blocks, pose, mean, &status, scratch, s_list, s_tmp, &s_nlist);
    positional_covariance<NT>(a, b, st, mean, best, cov, scratch);
    const double coarse_response = best > 1.0 ? 1.0 : best;
    const double cmean[3] = {mean[0], mean[1], mean[2]};
    double response = coarse_response;
    const uint8_t *grid = a.grid + (size_t)b * a.grid_stride;
    const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);

    if (a.refine) { // ------------------------------------------------ fine pass (CorrelateScan, doingFineMatch)
        const YmLattice &L = a.lf;
        const int nx = L.nx, ny = L.ny, nt = L.nt, nxy = nx * ny, nh = nxy * nt;
        const double cxw = cmean[0], cyw = cmean[1], ct = cmean[2];
        const double start_x = -L.off_x, start_y = -L.off_y, start_angle = ct - L.angle_off;
        for (int k = tid; k < nt; k += NT) {
            const double angle = start_angle + k * L.angle_res;
            s_cs[k] = make_double2(cos(angle), sin(angle));
        }
        for (int i = tid; i < nx; i += NT) s_cx[i] = hyp_cell(cxw, start_x, i, L.step_x, off_x, a.g);
        for (int i = tid; i < ny; i += NT) s_cy[i] = hyp_cell(cyw, start_y, i, L.step_y, off_y, a.g);
        for (int h = tid; h < nh; h += NT) s_sum[h] = 0u;
        __syncthreads();
        const double2 *ql = a.qlocal + (size_t)b * a.max_n;
        const bool block3 = nx == 3 && ny == 3 && s_cx[1] == s_cx[0] + 1 && s_cx[2] == s_cx[0] + 2 &&
                            s_cy[1] == s_cy[0] + 1 && s_cy[2] == s_cy[0] + 2;
        for (int k = wave; k < nt; k += NW) { // wave-uniform
            const double cosine = s_cs[k].x, sine = s_cs[k].y;
            if (block3) {
                // Karto's fine lattice is always 3x3 cells: a lane reads the 3x3 cell block under its beam as three
                // 4-byte words
                const uint32_t base0 = (uint32_t)(s_cy[0] * a.g.pitch + s_cx[0]);
                unsigned acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int i = lane; i < nq; i += 4 * 64) {
                    uint32_t w[4][3];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int ii = i + u * 64;
                        const int off = ii < nq ? lookup_offset(ql[ii], cosine, sine, off_x, off_y, a.g.scale, a.g.pitch) : 0;
                        const uint32_t idx = base0 + (uint32_t)off;
#pragma unroll
                        for (int r = 0; r < 3; r++) __builtin_memcpy(&w[u][r], grid + (uint32_t)(idx + r * a.g.pitch), 4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t m = (i + u * 64) < nq ? 0xffu : 0u;
#pragma unroll
                        for (int r = 0; r < 3; r++) {
                            acc[3 * r] += w[u][r] & m; acc[3 * r + 1] += (w[u][r] >> 8) & m; acc[3 * r + 2] += (w[u][r] >> 16) & m;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 9; j++) acc[j] = wave_reduce(acc[j], OpAddU());
                if (lane == 0)
#pragma unroll
                    for (int j = 0; j < 9; j++) s_sum[k * 9 + j] = acc[j];
            } else {
                // generic lattice: per beam, every (iy, ix) cell
                for (int i = lane; i < nq; i += 64) {
                    const int off = lookup_offset(ql[i], cosine, sine, off_x, off_y, a.g.scale, a.g.pitch);
                    for (int c = 0; c < nxy; c++) {
                        const int iy = c / nx, ix = c - iy * nx;
                        const unsigned idx = (unsigned)(s_cy[iy] * a.g.pitch + s_cx[ix] + off);
                        if (idx < limit) atomicAdd(&s_sum[k * nxy + c], (unsigned)grid[idx]);
                    }
                }
            }
        }
        __syncthreads();
        uint32_t *fs = a.fsums + (size_t)b * a.fsums_stride; // kept for the parity tests
        double lb = -1.0;
        for (int h = tid; h < nh; h += NT) {
            const int k = h / nxy, c = h - k * nxy, iy = c / nx, ix = c - iy * nx;
            const double x = start_x + ix * L.step_x, y = start_y + iy * L.step_y;
            const double r = hyp_response(a.g, L.penalize, s_sum[h], nq, x * x + y * y, start_angle + k * L.angle_res, ct);
            s_fresp[h] = r;
            fs[h] = s_sum[h];
            lb = r > lb ? r : lb;
        }
        best = block_reduce(lb, OpMaxD(), -1.0, scratch);
        double acc[5] = {0, 0, 0, 0, 0};
        for (int h = tid; h < nh && tid < YM_CANON; h += YM_CANON)
            if (kt_double_equal(s_fresp[h], best)) tie_accumulate(acc, L, h, cxw, cyw, start_angle);
        block_sum_vec<5>(acc, scratch);
        if (acc[4] > 0.0) {
            const double cnt = acc[4];
            mean[0] = acc[0] / cnt; mean[1] = acc[1] / cnt;
            mean[2] = atan2(acc[3] / cnt, acc[2] / cnt);
        } else {
            status = -5;
        }
        // ComputeAngularCovariance: GetResponse(angle k, cell of the mean pose) with the fine pass's lookup offsets is the
        // fine pass's own sum whenever that cell is a cell of the fine lattice (always, unless fp rounding puts the
        // tie mean outside); otherwise gather again.
        const double best_angle = kt_normalize_angle_difference(mean[2], ct);
        const int gx = world_to_grid(mean[0], off_x, a.g.scale) + a.g.border - a.g.win_origin;
        const int gy = world_to_grid(mean[1], off_y, a.g.scale) + a.g.border - a.g.win_origin;
        int hit_x = -1, hit_y = -1;
        for (int i = 0; i < nx; i++) if (s_cx[i] == gx) hit_x = i;
        for (int i = 0; i < ny; i++) if (s_cy[i] == gy) hit_y = i;
        unsigned *s_asum = reinterpret_cast<unsigned *>(s_list); // free by now
        __syncthreads();
        if (hit_x >= 0 && hit_y >= 0) {
            for (int k = tid; k < nt; k += NT) s_asum[k] = s_sum[k * nxy + hit_y * nx + hit_x];
        } else {
            const int base = gy * a.g.pitch + gx;
            for (int k = wave; k < nt; k += NW) {
                const double cosine = s_cs[k].x, sine = s_cs[k].y;
                unsigned v = 0;
                for (int i = lane; i < nq; i += 64) {
                    const unsigned idx = (unsigned)(base + lookup_offset(ql[i], cosine, sine, off_x, off_y, a.g.scale, a.g.pitch));
                    v += idx < limit ? grid[idx] : 0u;
                }
                v = wave_reduce(v, OpAddU());
                if (lane == 0) s_asum[k] = v;
            }
        }
        __syncthreads();
        double norm = 0.0, accv = 0.0;
        for (int k = 0; k < nt; k++) {
            const double angle = start_angle + k * L.angle_res;
            double r = (double)s_asum[k];
            r /= (double)(nq * YM_OCCUPIED);
            if (r >= (best - 0.1)) {
                norm += r;
                accv += ((angle - best_angle) * (angle - best_angle)) * r;
            }
        }
        if (norm > YM_KT_TOLERANCE) {
            if (accv < YM_KT_TOLERANCE) accv = L.angle_res * L.angle_res;
            accv /= norm;
        } else {
            accv = 1000 * (L.angle_res * L.angle_res);
        }
        cov[8] = accv;
        response = best > 1.0 ? 1.0 : best;
    }
    if (tid == 0) {
        for (int i = 0; i < 9; i++) st.cov[i] = cov[i];
        for (int i = 0; i < 3; i++) { st.mean[i] = mean[i]; st.center[i] = mean[i]; }
        st.response = response;
        st.coarse_response = coarse_response;
        st.status = status;
        if (a.host_out) a.host_out[b] = st;
    }
}

// ================================================================== K7 arg-best over the items of a call
// One block.  out[0..8) = {response, global chain id, x, y, heading, cov_xx, cov_yy, cov_tt} of the
// item with the highest response (ties: lowest index) -- the payload of the cross-rank arg-max.
__global__ __launch_bounds__(256) void argbest_kernel(const YmItemState *states, int n_items, long long id_base,
                                                      double *out) {
    __shared__ double s_r[4];
    __shared__ int s_i[4];
    double br = -1.0;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n_items; i += 256) {
        const double r = states[i].response;
        if (r > br || (r == br && i < bi)) { br = r; bi = i; }
    }
    // wave arg-max (value, then lowest index), then across the 4 waves
    const double wr = wave_reduce(br, OpMaxD());
    const int wi = wave_reduce(br == wr ? bi : 0x7fffffff, OpMinI());
    if ((threadIdx.x & 63) == 0) { s_r[threadIdx.x >> 6] = wr; s_i[threadIdx.x >> 6] = wi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = s_r[0];
        int idx = s_i[0];
        for (int w = 1; w < 4; w++)
            if (s_r[w] > r || (s_r[w] == r && s_i[w] < idx)) { r = s_r[w]; idx = s_i[w]; }
        if (idx == 0x7fffffff || idx >= n_items) idx = 0;
        const YmItemState &st = states[idx];
        out[0] = st.response; out[1] = (double)(id_base + idx);
        out[2] = st.mean[0]; out[3] = st.mean[1]; out[4] = st.mean[2];
        out[5] = st.cov[0]; out[6] = st.cov[4]; out[7] = st.cov[8];
    }
}

// ================================================================== "yagpy" semantics
// The reference's in-tree Python matcher (/root/reference/yag_slam/helpers.py:156-295 find_best_pose,
// scan_matching.py:175-222).  Unlike Karto it rounds every (hypothesis, point) pair separately
// (helpers.py:149-153), so the gather address is recomputed in fp64 per pair; this path exists for
// parity with the reference-generated golden vectors, not for speed.
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
    double2 *rot;        // [B][maxt][max_n] points rotated by tvals[k]
    uint32_t *sums;      // [B][maxt][maxd][maxd] -> stored dense as [k][iy][ix] with the pass's nx, ny
    double *out;         // same shape, fp64 scores
    const uint8_t *grid;
    size_t grid_stride;
    size_t vol_stride;   // entries per item in sums/out
    int32_t max_n, maxd, maxt;
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

// grid (maxt, B), 256 threads: block k rotates the points by tvals[k]; block 0 also writes the axes.
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
    if (k >= nt) return;
    const double t = yag_arange_at(-a.search_t + ct, a.step_t, k);
    const double c = cos(t), s = sin(t);
    const double2 *ql = a.qlocal + (size_t)b * a.max_n;
    double2 *rot = a.rot + ((size_t)b * a.maxt + k) * a.max_n;
    for (int l = tid; l < st.nq; l += 256) { // helpers.py:76-78 _rotate_points
        const double2 p = ql[l];
        rot[l] = make_double2(p.x * c - p.y * s, p.y * c + p.x * s);
    }
}

// grid (ceil(maxd*maxd/256), maxt, B): one thread per hypothesis (ix, iy) of angle k.
// helpers.py:134-153: per point rint((p - o)/res), bounds check, int(100*cell) accumulate.
__global__ __launch_bounds__(256) void yag_score_kernel(YagArgs a) {
    const int b = blockIdx.z, k = blockIdx.y;
    const YmItemState &st = a.states[b];
    const int nx = st.ydims[a.pass][0], ny = st.ydims[a.pass][1], nt = st.ydims[a.pass][2];
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (k >= nt || c >= nx * ny) return;
    const int iy = c / nx, ix = c - iy * nx;
    const double *ax = a.axes + (size_t)b * 3 * YM_YAG_MAX_DIM;
    const double xv = ax[ix], yv = ax[YM_YAG_MAX_DIM + iy], tv = ax[2 * YM_YAG_MAX_DIM + k];
    const double ox = st.off_x, oy = st.off_y, res = a.g.res;
    const int G = a.g.roi_w, w0 = a.g.win_origin, ww = a.g.win_w, pitch = a.g.pitch;
    const double2 *__restrict__ rot = a.rot + ((size_t)b * a.maxt + k) * a.max_n;
    const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
    const int np = st.nq;
    unsigned sum = 0;
#pragma unroll 4
    for (int l = 0; l < np; l++) {
        const double2 p = rot[l];
        const double x = xv + p.x, y = yv + p.y;
        const double gx = rint((x - ox) / res), gy = rint((y - oy) / res);
        const int _x = (int)gx, _y = (int)gy;
        if (_x >= 0 && _x < G && _y >= 0 && _y < G) {
            const int wx = _x - w0, wy = _y - w0;
            // cells outside the device window are provably empty
            if (wx >= 0 && wx < ww && wy >= 0 && wy < ww) sum += grid[wy * pitch + wx];
        }
    }
    double penalty_val = 1.0;
    if (a.penalize) {
        const double ct = a.pass ? st.ybest[0][3] : st.pose[2];
        const double sx_ = ox + G * res / 2, sy_ = oy + G * res / 2;
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

// grid (B), 1024 threads: np.argmax (first maximum in the reference's [ix][iy][k] order), mean of
// all scores >= best - 1e-8, the +-5 covariance windows (helpers.py:214-295).
__global__ __launch_bounds__(1024) void yag_reduce_kernel(YagArgs a) {
    constexpr int NT = 1024;
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
    __shared__ double s_mean[4];
    double acc[4] = {0, 0, 0, 0};
    if (nh <= 8192 * 32) {
        for (int w = tid; w < (nh + 31) / 32; w += NT) s_bits[w] = 0u;
        __syncthreads();
        for (int f = tid; f < nh; f += NT) {
            const int ix = f / (ny * nt), iy = (f % (ny * nt)) / nt, k = f % nt;
            if (out[((size_t)k * ny + iy) * nx + ix] >= response - 0.00000001) atomicOr(&s_bits[f >> 5], 1u << (f & 31));
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 0; w < (nh + 31) / 32; w++) {
                unsigned bits = s_bits[w];
                while (bits) {
                    const int f = w * 32 + __ffs((int)bits) - 1;
                    bits &= bits - 1;
                    const int ix = f / (ny * nt), iy = (f % (ny * nt)) / nt, k = f % nt;
                    acc[0] += ax[ix]; acc[1] += ax[YM_YAG_MAX_DIM + iy]; acc[2] += ax[2 * YM_YAG_MAX_DIM + k]; acc[3] += 1.0;
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

}  // namespace ym
