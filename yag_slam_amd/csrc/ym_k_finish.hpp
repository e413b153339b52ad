// ym_k_finish.hpp -- K5 score_kernel, K6 fine_kernel / final_kernel / finish_kernel, K7 argbest_kernel.
// Part of ym_kernels.hpp (include that, not this file).
#pragma once

namespace ym {

// ================================================================== K5a score
#define YM_SCORE_THREADS 256
struct ScoreArgs {
    YmGeom g;
    YmLattice lat;
    const uint16_t *partial;
    size_t partial_stride;
    const YmItemState *states;
    uint32_t *sums;       // [B][nt][ny][nx], or null: batches do not keep the integer sums (a third of this kernel's writes)
    size_t sums_stride;
    double *resp;         // [B][nt][ny][nx]
    double *blockmax;     // [B][n_blocks]
    double *probs;        // [B][ny*nx] max over theta per (x, y)
    size_t probs_stride;
    int32_t n_chunks, nx_pad, n_blocks;
    int32_t k_begin, k_end;   // the coarse angles this launch scores (a slice, or all)
    int32_t write_blockmax;
    int32_t lane_layout;      // 1: partial sums as correlate_region_kernel leaves them, [group][angle][lane][16] with
                              // lane = 32 * (ix / 13) + iy, slot ix % 13 (0: [group][angle][iy][nx_pad])
    unsigned long long *stamps;
};

// Karto's distance penalty of a lattice cell (CorrelateScan): depends on the cell only
__device__ __forceinline__ double dist_penalty(const YmGeom &g, double sq_dist) {
    double dp = 1.0 - (YM_PENALTY_GAIN * sq_dist / g.dist_var);
    return dp > g.min_dist_pen ? dp : g.min_dist_pen;
}
// response of one hypothesis from its integer sum, with the cell's distance penalty already known
__device__ __forceinline__ double hyp_response_dp(const YmGeom &g, int penalize, unsigned sum, int nq, double dp,
                                                  double angle, double center_t) {
    double response = 0.0;
    if (nq != 0) {
        response = (double)sum;
        response /= (double)(nq * YM_OCCUPIED);
    }
    if (penalize && !kt_double_equal(response, 0.0)) {
        const double sq_ang = (angle - center_t) * (angle - center_t);
        double ap = 1.0 - (YM_PENALTY_GAIN * sq_ang / g.ang_var);
        ap = ap > g.min_ang_pen ? ap : g.min_ang_pen;
        response *= (dp * ap);
    }
    return response;
}
__device__ __forceinline__ double hyp_response(const YmGeom &g, int penalize, unsigned sum, int nq, double sq_dist,
                                               double angle, double center_t) {
    return hyp_response_dp(g, penalize, sum, nq, dist_penalty(g, sq_dist), angle, center_t);
}

// One thread per (x, y) lattice cell, walking the coarse angles [k_begin, k_end): add the chunk-group partials,
// normalise, penalise, keep the maximum over theta (Karto's search-space probability grid: a plain store, no atomics).
// Block maxima are kept per (angle, block of YM_SCORE_THREADS cells): blockmax[k * n_cell_blocks + cb] covers the
// hypotheses h = k * nxy + cb * YM_SCORE_THREADS + t -- the finish stage looks for the arg-max and the tie set there.
// grid (n_cell_blocks, B)
__global__ __launch_bounds__(YM_SCORE_THREADS) void score_kernel(ScoreArgs a) {
    __shared__ double scratch[16];
    const int b = blockIdx.y;
    const YmItemState &st = a.states[b];
    const int nx = a.lat.nx, ny = a.lat.ny, nt = a.lat.nt, nxy = nx * ny;
    const int c = blockIdx.x * YM_SCORE_THREADS + threadIdx.x;
    const bool live = c < nxy;
    const int iy = live ? c / nx : 0, ix = live ? c - iy * nx : 0;
    const double x = -a.lat.off_x + ix * a.lat.step_x, y = -a.lat.off_y + iy * a.lat.step_y;
    const double sq_dist = x * x + y * y;
    const double ct = st.center[2];
    const int nq = st.nq;
    const size_t kstride = a.lane_layout ? (size_t)64 * 16 : (size_t)ny * a.nx_pad;
    const size_t cstride = (size_t)nt * kstride;
    const uint16_t *p0 = a.partial + (size_t)b * a.partial_stride +
                         (a.lane_layout ? (size_t)((32 * (ix / 13) + iy) * 16 + ix % 13) : (size_t)iy * a.nx_pad + ix);
    double best = 0.0;
    YM_STAMP(a, 10);
    for (int k = a.k_begin; k < a.k_end; k++) {
        double r = -1.0;
        if (live) {
            const uint16_t *p = p0 + (size_t)k * kstride;
            unsigned sum = 0;
#pragma unroll 4
            for (int c2 = 0; c2 < a.n_chunks; c2++) sum += p[(size_t)c2 * cstride];
            const double angle = (ct - a.lat.angle_off) + k * a.lat.angle_res;
            r = hyp_response(a.g, a.lat.penalize, sum, nq, sq_dist, angle, ct);
            const size_t h = (size_t)k * nxy + c;
            if (a.sums) a.sums[(size_t)b * a.sums_stride + h] = sum; // kept for the parity tests (single matches)
            a.resp[(size_t)b * a.sums_stride + h] = r;
            best = r > best ? r : best;
        }
        if (a.write_blockmax) {
            const double m = block_reduce(r, OpMaxD(), -1.0, scratch);
            if (threadIdx.x == 0) a.blockmax[(size_t)b * a.n_blocks + (size_t)k * gridDim.x + blockIdx.x] = m;
        }
    }
    if (live) a.probs[(size_t)b * a.probs_stride + c] = best; // max over theta of this launch's angles (responses are >= 0)
    YM_STAMP(a, 11);
}

// The same for a few items (a single match, an angle slice): one thread per HYPOTHESIS -- a thread walking all angles
// would be a long serial chain with nothing to hide it behind -- in the same block layout: block (cb, k) holds the cells
// cb * YM_SCORE_THREADS + t of angle k.  The per-(x, y) maximum is then an integer atomic max on the fp64 bit patterns
// (responses are >= 0, so the u64 order is the numeric order; probs is zeroed by the prepare stage).
// grid (n_cell_blocks, k_end - k_begin, B)
__global__ __launch_bounds__(YM_SCORE_THREADS) void score_hyp_kernel(ScoreArgs a) {
    __shared__ double scratch[16];
    const int b = blockIdx.z;
    const YmItemState &st = a.states[b];
    const int nx = a.lat.nx, ny = a.lat.ny, nt = a.lat.nt, nxy = nx * ny;
    const int c = blockIdx.x * YM_SCORE_THREADS + threadIdx.x, k = a.k_begin + blockIdx.y;
    double r = -1.0;
    YM_STAMP(a, 10);
    if (c < nxy) {
        const int iy = c / nx, ix = c - iy * nx;
        // (lane_layout: the region correlate's sets, [set][angle][lane = 32 * (ix / 13) + iy][ix % 13 of 16])
        const size_t kstride = a.lane_layout ? (size_t)64 * 16 : (size_t)ny * a.nx_pad;
        const uint16_t *p = a.partial + (size_t)b * a.partial_stride + (size_t)k * kstride +
                            (a.lane_layout ? (size_t)((32 * (ix / 13) + iy) * 16 + ix % 13) : (size_t)iy * a.nx_pad + ix);
        const size_t cstride = (size_t)nt * kstride;
        unsigned sum = 0;
#pragma unroll 8
        for (int c2 = 0; c2 < a.n_chunks; c2++) sum += p[(size_t)c2 * cstride];
        const double x = -a.lat.off_x + ix * a.lat.step_x, y = -a.lat.off_y + iy * a.lat.step_y;
        const double ct = st.center[2];
        const double angle = (ct - a.lat.angle_off) + k * a.lat.angle_res;
        r = hyp_response(a.g, a.lat.penalize, sum, st.nq, x * x + y * y, angle, ct);
        const size_t h = (size_t)k * nxy + c;
        if (a.sums) a.sums[(size_t)b * a.sums_stride + h] = sum;
        a.resp[(size_t)b * a.sums_stride + h] = r;
        if (r > 0.0)
            atomicMax(reinterpret_cast<unsigned long long *>(a.probs) + (size_t)b * a.probs_stride + c, (unsigned long long)__double_as_longlong(r));
    }
    if (a.write_blockmax) {
        const double m = block_reduce(r, OpMaxD(), -1.0, scratch);
        if (threadIdx.x == 0) a.blockmax[(size_t)b * a.n_blocks + (size_t)k * gridDim.x + blockIdx.x] = m;
    }
    YM_STAMP(a, 11);
}

// block maxima of a volume that was scored in angle slices (by several matchers): what score_kernel writes itself when
// it scores the whole volume.  grid (n_cell_blocks, nt)
__global__ __launch_bounds__(YM_SCORE_THREADS) void blockmax_kernel(const double *resp, int nxy, double *blockmax) {
    __shared__ double scratch[16];
    const int c = blockIdx.x * YM_SCORE_THREADS + threadIdx.x, k = blockIdx.y;
    const double m = block_reduce(c < nxy ? resp[(size_t)k * nxy + c] : -1.0, OpMaxD(), -1.0, scratch);
    if (threadIdx.x == 0) blockmax[(size_t)k * gridDim.x + blockIdx.x] = m;
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
    const double *blockmax;   // [B][n_blocks] maxima per (angle, block of YM_SCORE_THREADS lattice cells)
    const double *probs;      // [B][ny*nx] max over theta per (x, y)  (m_pSearchSpaceProbs)
    size_t probs_stride;
    const uint8_t *grid;
    size_t grid_stride;
    const double2 *qlocal;
    int32_t *foffsets;        // [B][nt_f][max_n] fine lookup table (scratch)
    uint32_t *fsums;          // [B][nt_f*ny_f*nx_f] fine sums (kept for parity tests)
    size_t fsums_stride;
    unsigned long long *stamps;
    uint32_t *host_flag;      // single matches: pinned host word that receives `serial` once host_out[0] is complete (the
    uint32_t serial, pad1;    // caller polls it instead of waiting for a stream event), or null
    // device-chained sequences (ym_map_sequence): the step's pose goes to seq_pose[0..2] (this scan's row of the segment's
    // pose table) and the next step's odometry prior = that pose (+) next_diff (tiny_tf's Transform composition) to
    // seq_prior[0..2]; a step that Karto would abort or repeat with a wider angle range (response expansion) is a fault
    // the host handles
    double *seq_pose;         // null: not a chained step
    double *seq_prior;
    int32_t *fault;
    double next_diff[3];
    int32_t step, expansion;  // expansion: the matcher's use_response_expansion
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
// The same when most threads hold exact zeros (`mine` = this thread's v is not all zero; tie sums: one thread, one tie,
// nearly always).  With at most one contributor the sum IS its values -- x + 0.0 = x whatever the order -- so they are
// handed around through LDS: two barriers, no reduction tree (1.4 -> 0.4 us on the critical path of every match).
template <int N>
__device__ __forceinline__ void block_sum_vec_sparse(double (&v)[N], double *scratch /* >= 16*N */, bool mine) {
    const int contributors = __syncthreads_count(mine ? 1 : 0);
    if (contributors > 1) { block_sum_vec<N>(v, scratch); return; }
    if (mine)
#pragma unroll
        for (int j = 0; j < N; j++) scratch[j] = v[j];
    __syncthreads();
    if (contributors == 1)
#pragma unroll
        for (int j = 0; j < N; j++) v[j] = scratch[j];
}

// tie-set mean of CorrelateScan: accumulate one hypothesis.  trig[k] = (cos, sin) of the heading of lattice angle k
// (tie_trig_table): a single lane computing an fp64 sin and cos AFTER the winner is known is a 1.3 us dependent chain
// on the critical path of every match; computed for all angles by as many lanes at the start of the phase it hides behind
// the loads of the block maxima.
__device__ __forceinline__ void tie_accumulate(double (&acc)[5], const YmLattice &L, int h, double cxw, double cyw,
                                               const double2 *trig) {
    const int nxy = L.nx * L.ny;
    const int k = h / nxy, c = h - k * nxy, iy = c / L.nx, ix = c - iy * L.nx;
    const double x = -L.off_x + ix * L.step_x, y = -L.off_y + iy * L.step_y;
    acc[0] += cxw + x; acc[1] += cyw + y;
    acc[2] += trig[k].x; acc[3] += trig[k].y;
    acc[4] += 1.0;
}
// (a barrier must separate this from the first tie_accumulate)
template <int NT>
__device__ __forceinline__ void tie_trig_table(const YmLattice &L, double start_angle, double2 *trig) {
    for (int k = threadIdx.x; k < L.nt; k += NT) {
        const double hd = kt_normalize_angle(start_angle + k * L.angle_res);
        trig[k] = make_double2(cos(hd), sin(hd));
    }
}

// Coarse tail of CorrelateScan for one item: best response, mean of all hypotheses with
// DoubleEqual(response, best).  Runs redundantly in every block that needs the coarse mean.
// Returns best (unclamped); mean[] and *status valid in every thread.
template <int NT, int KEEP = 4 /* block maxima per thread that stay in registers (all in flight together, no second read) */>
__device__ __forceinline__ double coarse_best_and_mean(const YmLattice &L, const double *resp, const double *bm,
                                                       int n_blocks, const double pose[3], double mean[3], int *status,
                                                       double *scratch /* >= 80 */, int *s_list /* NT */, int *s_tmp /* NT */,
                                                       int *s_nlist, double2 *s_trig /* L.nt */,
                                                       const double *first_round = nullptr /* [KEEP]: bm[tid + u * NT] or -1, loaded by the caller */) {
    const int tid = threadIdx.x;
    const int nh = L.nx * L.ny * L.nt;
    const double start_angle = pose[2] - L.angle_off;
    if (tid == 0) *s_nlist = 0;
    double lb = -1.0;
    double v0[KEEP]; // (the first round of block maxima is in flight while the table is computed)
#pragma unroll
    for (int u = 0; u < KEEP; u++) v0[u] = first_round ? first_round[u] : (tid + u * NT) < n_blocks ? bm[tid + u * NT] : -1.0;
    tie_trig_table<NT>(L, start_angle, s_trig);
#pragma unroll
    for (int u = 0; u < KEEP; u++) lb = v0[u] > lb ? v0[u] : lb;
    for (int i0 = tid + KEEP * NT; i0 < n_blocks; i0 += 4 * NT) {
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
#pragma unroll
    for (int u = 0; u < KEEP; u++) // (from the registers: the maxima are -1 beyond the last block)
        if (v0[u] >= best - YM_KT_TOLERANCE) {
            const int at = atomicAdd(s_nlist, 1);
            if (at < NT) s_tmp[at] = tid + u * NT; else overflow = 1;
        }
    for (int i = tid + KEEP * NT; i < n_blocks; i += NT)
        if (bm[i] >= best - YM_KT_TOLERANCE) {
            const int at = atomicAdd(s_nlist, 1);
            if (at < NT) s_tmp[at] = i; else overflow = 1;
        }
    overflow = __syncthreads_or(overflow);
    if (!overflow && *s_nlist > 1) { // (block-uniform; one candidate block, the usual case, needs no order)
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
            const int *list = nlist > 1 ? s_list : s_tmp;
            const int nxy = L.nx * L.ny, ncb = (nxy + YM_SCORE_THREADS - 1) / YM_SCORE_THREADS;
            for (int w = tid; w < nlist * YM_SCORE_THREADS; w += YM_CANON) {
                const int bid = list[w / YM_SCORE_THREADS]; // block = (angle k, block cb of cells)
                const int k = bid / ncb, c = (bid - k * ncb) * YM_SCORE_THREADS + (w % YM_SCORE_THREADS);
                const int h = k * nxy + c;
                if (c < nxy && kt_double_equal(resp[h], best)) tie_accumulate(acc, L, h, pose[0], pose[1], s_trig);
            }
        } else {
            for (int h = tid; h < nh; h += YM_CANON)
                if (kt_double_equal(resp[h], best)) tie_accumulate(acc, L, h, pose[0], pose[1], s_trig);
        }
    }
    block_sum_vec_sparse<5>(acc, scratch, acc[4] != 0.0);
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
template <int NT, bool WIDE = false /* 32 loads in flight on lattices beyond 2048 cells (fine_kernel: registers to spare) */>
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
        // U loads in flight; the additions stay in increasing cell order (the canonical order above).  A large lattice
        // (configs[4]: 201 x 201 cells) is 158 cells per canonical thread: at eight loads per round that was twenty dependent
        // round trips in the one block everything after it waits for.
        auto walk = [&](auto u_tag) {
            constexpr int U = decltype(u_tag)::value;
            for (int c0 = tid; c0 < nxy && tid < YM_CANON; c0 += U * YM_CANON) {
                double pv[U];
#pragma unroll
                for (int u = 0; u < U; u++) pv[u] = (c0 + u * YM_CANON) < nxy ? probs[c0 + u * YM_CANON] : -1.0;
#pragma unroll
                for (int u = 0; u < U; u++) {
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
        };
        if (WIDE && nxy > 8 * YM_CANON) walk(std::integral_constant<int, WIDE ? 32 : 8>());
        else walk(std::integral_constant<int, 8>());
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
template <bool WIDE /* a coarse lattice beyond 2048 cells: the covariance block keeps 32 loads in flight (131 VGPRs instead of 94) */>
__global__ __launch_bounds__(YM_FINE_THREADS) void fine_kernel(FinishArgs a) {
    constexpr int NT = YM_FINE_THREADS;
    __shared__ double scratch[16 * 5];
    __shared__ int s_list[NT], s_tmp[NT];
    __shared__ int s_nlist;
    __shared__ double s_cs[2];
    __shared__ int s_cx[64], s_cy[64];
    __shared__ unsigned s_sum[YM_MAX_FINE_HYP];
    __shared__ double2 s_trig[YM_MAX_COARSE_NT];
    int k;
    const int b = xcd_item_of_block_2d(k); // the blocks of an item share its grid patch: keep them on one XCD
    const int tid = threadIdx.x, lane = tid & 63;
    YM_STAMP(a, 12);
    // the block maxima leave together with the item's state: the "no readings" test on the state would otherwise put a
    // memory round trip of its own in front of them
    const double *bm = a.blockmax + (size_t)b * a.n_blocks;
    constexpr int KEEP = WIDE ? 16 : 4; // (configs[4]: 7268 block maxima, fourteen per thread -- one round trip, not four, and no second read)
    double bm0[KEEP];
#pragma unroll
    for (int u = 0; u < KEEP; u++) bm0[u] = (tid + u * NT) < a.n_blocks ? bm[tid + u * NT] : -1.0;
    YmItemState &st = a.states[b];
    const int nq = st.nq;
    if (nq == 0) return;
    const double pose[3] = {st.pose[0], st.pose[1], st.pose[2]};
    const double off_x = st.off_x, off_y = st.off_y;
    double mean[3];
    int status = 0;
    const double best = coarse_best_and_mean<NT, KEEP>(a.lc, a.resp + (size_t)b * a.sums_stride, bm, a.n_blocks, pose, mean, &status,
                                                       scratch, s_list, s_tmp, &s_nlist, s_trig, bm0);
    if (k == (a.refine ? a.lf.nt : 0)) { // extra block: coarse result + positional covariance for final_kernel
        double cov[9];
        positional_covariance<NT, WIDE>(a, b, st, mean, best, cov, scratch);
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
    const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
    const bool block3 = !a.g.kpitch && nx == 3 && ny == 3 && s_cx[1] == s_cx[0] + 1 && s_cx[2] == s_cx[0] + 2 &&
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
            const int off = lookup_offset(ql[i], cosine, sine, off_x, off_y, a.g.scale, lin_pitch(a.g));
            foff[i] = off;
            for (int c = 0; c < nxy; c++) {
                const int iy = c / nx, ix = c - iy * nx;
                const unsigned v = cell_value(a.g, grid, limit, (unsigned)(s_cy[iy] * lin_pitch(a.g) + s_cx[ix] + off));
                if (v) atomicAdd(&s_sum[c], v);
            }
        }
    }
    __syncthreads();
    uint32_t *fs = a.fsums + (size_t)b * a.fsums_stride + (size_t)k * nxy;
    for (int h = tid; h < nxy; h += NT) fs[h] = s_sum[h];
    YM_STAMP(a, 15);
}

// The loop over the fine angles of ComputeAngularCovariance: thread k takes angle k (its response -- an fp64 division -- and its
// weighted term; eleven divisions in a row in EVERY thread were 0.4 us at the end of every match), then the sums are taken in
// k order as the sequential loop takes them (an angle it skips contributes +0.0: x + 0.0 == x).  tmp: 2 * nt doubles that are
// free; the result in every thread.  (Block-uniform call: it holds a barrier.)
template <int NT>
__device__ __forceinline__ double angular_cov_sums(const unsigned *s_asum, int nt, int nq, double start_angle, double angle_res,
                                                   double best, double best_angle, double *tmp, double *norm_out) {
    for (int k = threadIdx.x; k < nt; k += NT) {
        const double angle = start_angle + k * angle_res;
        double r = (double)s_asum[k];
        r /= (double)(nq * YM_OCCUPIED);
        const bool in = r >= (best - 0.1);
        tmp[2 * k] = in ? r : 0.0;
        tmp[2 * k + 1] = in ? ((angle - best_angle) * (angle - best_angle)) * r : 0.0;
    }
    __syncthreads();
    double norm = 0.0, accv = 0.0;
    for (int k = 0; k < nt; k++) {
        norm += tmp[2 * k];
        accv += tmp[2 * k + 1];
    }
    *norm_out = norm;
    return accv;
}

// a chained step's pose and the next step's odometry prior (ym_map_sequence; yag_slam_amd/transform.py: a + b)
__device__ __forceinline__ void chain_next_pose(const FinishArgs &a, const double pose[3]) {
    const double c = cos(pose[2]), s = sin(pose[2]);
    a.seq_pose[0] = pose[0]; a.seq_pose[1] = pose[1]; a.seq_pose[2] = pose[2];
    a.seq_prior[0] = pose[0] + c * a.next_diff[0] - s * a.next_diff[1];
    a.seq_prior[1] = pose[1] + s * a.next_diff[0] + c * a.next_diff[1];
    a.seq_prior[2] = pose[2] + a.next_diff[2];
}

// ---- K6b final: grid (B), 256 threads: fine arg-max / mean, angular covariance, result.
__global__ __launch_bounds__(YM_FINISH_THREADS) void final_kernel(FinishArgs a) {
    constexpr int NT = YM_FINISH_THREADS;
    __shared__ double scratch[16 * 5];
    __shared__ double s_fresp[YM_MAX_FINE_HYP];
    __shared__ unsigned s_fsum[YM_MAX_FINE_HYP]; // the fine sums as loaded (the angular covariance reads one column of them)
    __shared__ unsigned s_asum[YM_MAX_FINE_NT];
    __shared__ double2 s_trig[YM_MAX_FINE_NT];
    __shared__ int s_hit[2];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63;
    YM_STAMP(a, 16);
    // (the first fine sums leave together with the item's state, not a round trip after it)
    const uint32_t f0 = (a.refine && tid < a.lf.nx * a.lf.ny * a.lf.nt) ? a.fsums[(size_t)b * a.fsums_stride + tid] : 0u;
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
            if (a.host_flag) { __threadfence_system(); *reinterpret_cast<volatile uint32_t *>(a.host_flag) = a.serial; }
            if (a.seq_pose) chain_next_pose(a, st.pose);
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
        tie_trig_table<NT>(L, start_angle, s_trig);
        for (int h = tid; h < nh; h += NT) {
            const int k = h / nxy, c = h - k * nxy, iy = c / nx, ix = c - iy * nx;
            const double x = start_x + ix * L.step_x, y = start_y + iy * L.step_y;
            const uint32_t fsum = h == tid ? f0 : fs[h];
            const double r = hyp_response(a.g, L.penalize, fsum, nq, x * x + y * y, start_angle + k * L.angle_res, ct);
            s_fresp[h] = r;
            s_fsum[h] = fsum;
            lb = r > lb ? r : lb;
        }
        for (int k = tid; k < nt; k += NT) s_asum[k] = 0u;
        if (tid < 2) s_hit[tid] = -1;
        best = block_reduce(lb, OpMaxD(), -1.0, scratch);
        double acc[5] = {0, 0, 0, 0, 0};
        for (int h = tid; h < nh && tid < YM_CANON; h += YM_CANON)
            if (kt_double_equal(s_fresp[h], best)) tie_accumulate(acc, L, h, cxw, cyw, s_trig);
        block_sum_vec_sparse<5>(acc, scratch, acc[4] != 0.0);
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
        const int base = gy * lin_pitch(a.g) + gx;
        const int32_t *foff = a.foffsets + (size_t)b * a.nt_stride * a.max_n;
        // GetResponse(angle k, cell of the mean pose) uses the fine pass's own lookup offsets, so when that cell is one
        // of the fine lattice's cells (always, unless fp rounding puts the tie mean outside) the sum IS the fine
        // pass's integer sum for (k, that cell): take it instead of gathering the scan again.  Lattice column i is looked
        // at by thread i, row i by thread 64 + i (the last match wins, as in a loop over i: the cells increase with i).
        if (tid < nx && hyp_cell(cxw, start_x, tid, L.step_x, off_x, a.g) == gx) atomicMax(&s_hit[0], tid);
        if (tid >= 64 && tid - 64 < ny && hyp_cell(cyw, start_y, tid - 64, L.step_y, off_y, a.g) == gy) atomicMax(&s_hit[1], tid - 64);
        __syncthreads(); // s_asum cleared above, the hits in
        const int hit_x = s_hit[0], hit_y = s_hit[1];
        if (hit_x >= 0 && hit_y >= 0) {
            for (int k = tid; k < nt; k += NT) s_asum[k] = s_fsum[k * nxy + hit_y * nx + hit_x];
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
                    idx[u] = (kk[u] >= 0 && i < nq) ? (unsigned)(base + foff[(size_t)kk[u] * a.max_n + i]) : 0xffffffffu; // (never a cell)
                }
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = cell_value(a.g, grid, limit, idx[u]);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const unsigned sum = wave_reduce(v[u], OpAddU());
                    if (lane == 0 && kk[u] >= 0) atomicAdd(&s_asum[kk[u]], sum);
                }
            }
        }
        __syncthreads();
        double norm = 0.0;
        double accv = angular_cov_sums<NT>(s_asum, nt, nq, start_angle, L.angle_res, best, best_angle, s_fresp, &norm);
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
        if (a.host_out) {
            // what the host reads of a result (state_to_result), straight from registers: copying `st` would first read
            // the whole state back from device memory -- one more round trip at the very end of the chain
            YmItemState *ho = a.host_out + b;
            for (int i = 0; i < 9; i++) ho->cov[i] = cov[i];
            for (int i = 0; i < 3; i++) { ho->mean[i] = mean[i]; ho->center[i] = mean[i]; }
            ho->response = response;
            ho->coarse_response = coarse_response;
            ho->nq = nq;
            ho->status = status;
        }
        if (a.host_flag) { __threadfence_system(); *reinterpret_cast<volatile uint32_t *>(a.host_flag) = a.serial; }
        if (a.seq_pose) {
            if (status != 0 || (a.expansion && kt_double_equal(coarse_response, 0.0))) atomicCAS(a.fault, 0, a.step);
            chain_next_pose(a, mean);
        }
    }
    YM_STAMP(a, 19);
}

// ---- K6 finish, one block per item (batches): everything fine_kernel + final_kernel do, without the eleven-fold
// recomputation of the coarse arg-max that one-block-per-fine-angle costs.  Wave w scores the 3x3 fine lattice for
// fine angles w, w + 16, ...; the fine sums stay in LDS.  grid (B), 1024 threads.
// NT = 1024 threads: shortest latency per item; NT = 256: the kernel is a chain of short dependent phases, so on a big
// batch several small blocks per CU hide each other's waits better than one large block after the other (dynamic LDS
// = 12 bytes per fine hypothesis: the fine sums and responses)
#define YM_FINISH1_THREADS 1024
#ifndef YM_FINISH_OCC
#define YM_FINISH_OCC 5 // (waves per SIMD the compiler leaves room for at 256 threads; 7 and 8 -- 68 bytes of scratch -- measured the same step)
#endif
#define YM_FINISH_LDS_BYTES(nh) ((size_t)(nh) * 12 + 16)
template <int NT>
__global__ __launch_bounds__(NT, NT == 256 ? YM_FINISH_OCC : 4 /* NT = 256: eight blocks of four waves per CU (64 VGPRs): the kernel is a chain of short dependent phases */) void finish_kernel(FinishArgs a) {
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char fin_lds[];
    __shared__ double scratch[16 * 5];
    __shared__ int s_list[NT], s_tmp[NT];
    __shared__ int s_nlist;
    __shared__ double2 s_cs[YM_MAX_FINE_NT];
    __shared__ double2 s_trig[YM_MAX_COARSE_NT > YM_MAX_FINE_NT ? YM_MAX_COARSE_NT : YM_MAX_FINE_NT]; // tie tables: coarse, then fine
    __shared__ int s_cx[64], s_cy[64];
    const int nh_fine = a.refine ? a.lf.nx * a.lf.ny * a.lf.nt : 0;
    double *s_fresp = reinterpret_cast<double *>(fin_lds);                      // [nh_fine]
    unsigned *s_sum = reinterpret_cast<unsigned *>(fin_lds + (size_t)nh_fine * 8); // [nh_fine]
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
                                           a.n_blocks, pose, mean, &status, scratch, s_list, s_tmp, &s_nlist, s_trig);
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
        tie_trig_table<NT>(L, start_angle, s_trig); // (the coarse table is done with: positional_covariance ends with barriers)
        for (int i = tid; i < nx; i += NT) s_cx[i] = hyp_cell(cxw, start_x, i, L.step_x, off_x, a.g);
        for (int i = tid; i < ny; i += NT) s_cy[i] = hyp_cell(cyw, start_y, i, L.step_y, off_y, a.g);
        for (int h = tid; h < nh; h += NT) s_sum[h] = 0u;
        __syncthreads();
        const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
        const bool block3 = !a.g.kpitch && nx == 3 && ny == 3 && s_cx[1] == s_cx[0] + 1 && s_cx[2] == s_cx[0] + 2 &&
                            s_cy[1] == s_cy[0] + 1 && s_cy[2] == s_cy[0] + 2;
        if (block3) {
            // Karto's fine lattice is always 3x3 cells: a lane reads the 3x3 cell block under its beam as three 4-byte words.
            // Round 5: the four lanes of a QUAD take the same beam at four consecutive fine angles -- 0.0035 rad apart, their
            // blocks lie a cell or two apart along the beam's arc, i.e. mostly in the same 128-byte line of a row, and the vector L1
            // charges a quad one clock per line it touches (this loop was the kernel: 0.63 line visits per CU and clock with
            // lane = beam, every lane its own line).  A wave takes a quarter (NW-th) of the beams through all angle groups; the
            // lanes of one angle add their sums by butterflies over the beam index, then one LDS atomic per (wave, hypothesis).
            const uint32_t base0 = (uint32_t)(s_cy[0] * a.g.pitch + s_cx[0]);
            const int sub = lane & 3, bl = lane >> 2;
            const int per_wave = (nq + NW - 1) / NW, i_lo = wave * per_wave, i_hi = min(nq, i_lo + per_wave);
            for (int k0 = 0; k0 < nt; k0 += 4) { // block-uniform
                const int k = k0 + sub;
                const bool kin = k < nt;
                const double cosine = s_cs[kin ? k : k0].x, sine = s_cs[kin ? k : k0].y;
                unsigned acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int i0 = i_lo; i0 < i_hi; i0 += 4 * 16) { // wave-uniform
                    uint32_t w[4][3];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int ii = i0 + u * 16 + bl;
                        const int off = (ii < i_hi && kin) ? lookup_offset(ql[ii], cosine, sine, off_x, off_y, a.g.scale, a.g.pitch) : 0;
                        const uint32_t idx = base0 + (uint32_t)off;
#pragma unroll
                        for (int r = 0; r < 3; r++) __builtin_memcpy(&w[u][r], grid + (uint32_t)(idx + r * a.g.pitch), 4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t m = ((i0 + u * 16 + bl) < i_hi && kin) ? 0xffu : 0u;
#pragma unroll
                        for (int r = 0; r < 3; r++) {
                            acc[3 * r] += w[u][r] & m; acc[3 * r + 1] += (w[u][r] >> 8) & m; acc[3 * r + 2] += (w[u][r] >> 16) & m;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 9; j++) {
                    unsigned v = acc[j];
                    v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
                    if (bl == 0 && kin && v) atomicAdd(&s_sum[k * 9 + j], v);
                }
            }
        } else {
            for (int k = wave; k < nt; k += NW) { // wave-uniform
                const double cosine = s_cs[k].x, sine = s_cs[k].y;
                // generic lattice: per beam, every (iy, ix) cell
                for (int i = lane; i < nq; i += 64) {
                    const int off = lookup_offset(ql[i], cosine, sine, off_x, off_y, a.g.scale, lin_pitch(a.g));
                    for (int c = 0; c < nxy; c++) {
                        const int iy = c / nx, ix = c - iy * nx;
                        const unsigned v = cell_value(a.g, grid, limit, (unsigned)(s_cy[iy] * lin_pitch(a.g) + s_cx[ix] + off));
                        if (v) atomicAdd(&s_sum[k * nxy + c], v);
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
            if (kt_double_equal(s_fresp[h], best)) tie_accumulate(acc, L, h, cxw, cyw, s_trig);
        block_sum_vec_sparse<5>(acc, scratch, acc[4] != 0.0);
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
            const int base = gy * lin_pitch(a.g) + gx;
            for (int k = wave; k < nt; k += NW) {
                const double cosine = s_cs[k].x, sine = s_cs[k].y;
                unsigned v = 0;
                for (int i = lane; i < nq; i += 64)
                    v += cell_value(a.g, grid, limit, (unsigned)(base + lookup_offset(ql[i], cosine, sine, off_x, off_y, a.g.scale, lin_pitch(a.g))));
                v = wave_reduce(v, OpAddU());
                if (lane == 0) s_asum[k] = v;
            }
        }
        __syncthreads();
        double norm = 0.0;
        double accv = angular_cov_sums<NT>(s_asum, nt, nq, start_angle, L.angle_res, best, best_angle, s_fresp, &norm);
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

}  // namespace ym
