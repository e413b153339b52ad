// ym_k_common.hpp -- shared device helpers: Karto math, block/wave reductions, stamps, tile size.
// Part of ym_kernels.hpp (include that, not this file).
#pragma once

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
// the same over each 32-lane half of the wave (every lane gets its half's result)
template <typename Op>
__device__ __forceinline__ int half_wave_reduce(int v, Op op) {
    v = op(v, dpp_i32<YM_DPP_QUAD_1032>(v));
    v = op(v, dpp_i32<YM_DPP_QUAD_2301>(v));
    v = op(v, dpp_i32<YM_DPP_ROW_ROR4>(v));
    v = op(v, dpp_i32<YM_DPP_ROW_ROR8>(v));
    const int lo = op(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16));
    const int hi = op(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48));
    return (threadIdx.x & 32) ? hi : lo;
}
// the same over each row of 16 lanes (min / max: the rotations visit every lane of the row)
template <typename Op>
__device__ __forceinline__ int row16_reduce(int v, Op op) {
    v = op(v, dpp_i32<YM_DPP_QUAD_1032>(v));
    v = op(v, dpp_i32<YM_DPP_QUAD_2301>(v));
    v = op(v, dpp_i32<YM_DPP_ROW_ROR4>(v));
    v = op(v, dpp_i32<YM_DPP_ROW_ROR8>(v));
    return v;
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

// The lookup cell of one (point, angle) pair in either semantics.
// Karto (GridIndexLookup::ComputeOffsets): the cell displacement of the rotated point, WorldToGrid(rot + gridOffset); the hypothesis
// cells (hyp_cell) are added to it.
// "yagpy": the reference's Python matcher rounds every (hypothesis, point) pair on its own -- score_world_points_on_grid,
// /root/reference/yag_slam/helpers.py:149-153: np.round(((xvals[i] + xx) - ox) / res), half to even -- so a lookup table exists only
// where those roundings form a lattice: rint(((xvals[i] + xx) - ox) / res) == rint(((xvals[0] + xx) - ox) / res) + i * step_cells for
// every i.  yag_lattice_kernel (ym_k_yagpy.hpp) PROVES that per (point, angle) pair before any production correlate kernel may treat
// the item as regular; the lookup cell is then the WINDOW cell hypothesis (0, 0) reads (add = xvals[0], yvals[0]: YmItemState::ylat)
// and the hypothesis cells are i * step_cells.  The rotation is the expression of helpers.py:76-78 (the operands of its products and
// sums commuted, which changes no bit).
__device__ __forceinline__ int2 lookup_cell_sem(const YmGeom &g, double2 p, double cosine, double sine, double off_x, double off_y,
                                                double add_x, double add_y) {
    const double ox = cosine * p.x - sine * p.y;
    const double oy = sine * p.x + cosine * p.y;
    if (g.semantics == 1 /* YM_SEM_YAGPY */)
        return make_int2((int)rint(((add_x + ox) - off_x) / g.res) - g.win_origin, (int)rint(((add_y + oy) - off_y) / g.res) - g.win_origin);
    return make_int2(world_to_grid(ox + off_x, off_x, g.scale), world_to_grid(oy + off_y, off_y, g.scale));
}
__device__ __forceinline__ int lookup_offset_sem(const YmGeom &g, double2 p, double cosine, double sine, double off_x, double off_y,
                                                 double add_x, double add_y, int pitch) {
    const int2 c = lookup_cell_sem(g, p, cosine, sine, off_x, off_y, add_x, add_y);
    return c.x + c.y * pitch;
}

// ScanMatcher::GetResponse reads pByte[offset] after the test IsUpTo(gridPositionIndex + offset, data size) on the LINEAR
// index: an offset that leaves the grid sideways wraps into a neighbouring row.  Inside the device window no offset of a
// reading within the matcher's range threshold ever leaves it, so the fast kernels need no test and the per-cell paths test
// against the window.  A query that holds a reading BEYOND that threshold (scans carry their own) is answered with the
// window = Karto's whole storage and every linear index formed with Karto's row pitch (g.kpitch), exactly as Karto would:
// lin_pitch() is the pitch of lookup offsets and hypothesis cells, cell_value() the read.
__device__ __forceinline__ int lin_pitch(const YmGeom &g) { return g.kpitch ? g.kpitch : g.pitch; }
__device__ __forceinline__ unsigned cell_value(const YmGeom &g, const uint8_t *grid, unsigned limit, unsigned idx) {
    if (!g.kpitch) return idx < limit ? grid[idx] : 0u;
    if (idx >= (unsigned)(g.kpitch * g.storage_w)) return 0u; // IsUpTo(index, data size)
    const unsigned cy = idx / (unsigned)g.kpitch, cx = idx - cy * (unsigned)g.kpitch;
    return cx < (unsigned)g.win_w ? grid[cy * (unsigned)g.pitch + cx] : 0u; // (the bytes between width and pitch are zero)
}

// hypothesis cells of one lattice axis: WorldToGrid(centre + (start + i*step)), window coordinates
__device__ __forceinline__ int hyp_cell(double centre, double start, int i, double step, double off, const YmGeom &g) {
    const double v = start + i * step;
    return world_to_grid(centre + v, off, g.scale) + g.border - g.win_origin;
}

// raster tile (also used by prepare_kernel, which builds the raster's work list)
// cells per chunk box (consecutive readings of one base scan whose window bounding box the raster tests against its tile):
// 32 = half a wave.  (64 listed a third more tiles than hold a cell in reach and loaded 11 x 64 cells per tile.)
#define YM_BOX_CELLS 16
#define YM_N_BOXES(n) (((n) + YM_BOX_CELLS - 1) / YM_BOX_CELLS)
#define YM_TILE_W 64
#define YM_TILE_H 32      // raster tile: 64 x 32 cells -- or 64 x 64 (YM_TILE_H_TALL) on large batches with large windows, chosen per call
#define YM_TILE_H_TALL 64
#define YM_RASTER_RW ((YM_TILE_W + 2 * YM_MAX_KERNEL_HALF + 63) / 64 + 1) // 64-bit words per row of the raster's occupancy bitmap
#define YM_RASTER_GP (YM_TILE_W + 8) // LDS bytes per row of the raster's row distances / finished tile (read and written with lane = row: 18 dwords apart)
// dynamic LDS of raster_kernel: row distances of the tile and its halo | bitmap + the row pass's tables, later (with 544 bytes more) the finished tile | the smear values
#define YM_RASTER_UNION(th, h, ntab) ((size_t)((th) + 2 * (h)) * YM_RASTER_RW * 8 + (size_t)(ntab) * 1024 > (size_t)(th) * YM_RASTER_GP ? (size_t)((th) + 2 * (h)) * YM_RASTER_RW * 8 + (size_t)(ntab) * 1024 : (size_t)(th) * YM_RASTER_GP)
#define YM_RASTER_LDS_BYTES(th, h, ntab) ((size_t)((th) + 2 * (h)) * YM_RASTER_GP + YM_RASTER_UNION(th, h, ntab) + (size_t)((2 * (h) * (h) + 2 + 15) / 16 * 16))
#define YM_TILE_HITS 64    // hit slots per entry of the raster's work list: the chunks that reach the entry's tile (four times what the bench scans need of a tall tile)

}  // namespace ym
