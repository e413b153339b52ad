// ym_k_prepare.hpp -- K1 prepare_kernel, K1c tiles_kernel, K1b select_kernel.
// Part of ym_kernels.hpp (include that, not this file).
#pragma once

namespace ym {

// ================================================================== K1 prepare
#define YM_PREP_LDS_BYTES(max_n) ((size_t)(max_n) * 25 + ((size_t)(max_n) / 64 + 2) * 4 + 16)
#define YM_INLINE_SCANS 16
struct YmInlineDesc {        // call descriptor passed in the kernel arguments (single item, few scans)
    YmItem item;
    YmScanRef scans[YM_INLINE_SCANS];
};
struct PrepareArgs {
    const YmScanRef *scans;  // device copy of the call descriptor; unused when use_inline
    const YmItem *items;
    int32_t use_inline;
    int32_t pad0;
    YmInlineDesc inl;
    YmGeom g;
    YmLattice lat;           // coarse lattice
    YmItemState *states;
    double2 *qlocal;         // [query slots][max_n]
    int32_t *qnp;            // [query slots] number of point readings of the slot's query
    int2 *cells;             // [B][max_base][max_n]  window cell of every base point, NONE when filtered
    int4 *bbox;              // [B][max_base][YM_N_BOXES(max_n)] window bounding box of YM_BOX_CELLS consecutive cells
    double2 *ctrig;          // [B][nt_stride] (cos, sin) of every coarse angle
    int32_t *hypcell;        // [B][2][dim_stride]
    double *probs;           // [B][ny*nx] cleared here, filled by the score stage
    int32_t max_n, max_base, nt_stride, dim_stride;
    // batches: the heavy work (projection, trigger chain) runs once per distinct stale scan / distinct query in
    // points_kernel; jobs[j] = scan index, bit 31 set = "as a query, into query slot (j's low bits in qjob_slot)"
    const int32_t *jobs;     // [n_jobs] scan index | (query ? 0x80000000 : 0)
    const int32_t *job_slot; // [n_jobs] query slot of a query job
    unsigned long long *stamps;
    // device-chained sequences (ym_map_sequence): the host sized the raster's tile rectangle from PREDICTED poses; a kept
    // cell outside cell_box (window cells whose smear stays inside the launched tiles) or an earlier step's fault makes
    // *fault = step (first one wins) and the host repeats from there with the poses it then knows
    int32_t *fault;          // null: not a chained step
    int32_t step, pad1;
    int32_t cell_box[4];     // x0, y0, x1, y1 (inclusive)
    int32_t *tile_max_zero;  // batches with raster work lists: the word tiles_kernel raises to the longest list, zeroed here (a
                             // memset of its own was one more launch in the chain), or null
};

// LDS carve-up shared by the kernels that project a scan
struct PrepLds {
    double *sx, *sy;
    int *nxt, *ex, *ent;
    unsigned char *chain;
};
__device__ __forceinline__ PrepLds prep_lds(unsigned char *raw, int max_n) {
    PrepLds l;
    l.sx = reinterpret_cast<double *>(raw);
    l.sy = l.sx + max_n;
    l.nxt = reinterpret_cast<int *>(l.sy + max_n);
    l.ex = l.nxt + max_n;                     // exit of the chain walk from point i out of its segment
    l.ent = l.ex + max_n;                     // chain entry node per 64-point segment (max_n/64 + 1)
    l.chain = reinterpret_cast<unsigned char *>(l.ent + max_n / 64 + 2);
    return l;
}

// ---- point readings, compacted in beam order (LocalizedRangeScan::Update / _get_point_readings) -> sx, sy; returns np.
// One barrier for the whole scan instead of two per NT beams: pass 1 counts the valid beams of every (chunk of NT
// beams, wave) by ballot, pass 2 re-reads the ranges (L1) and places each valid beam after everything before it.
template <int NT>
__device__ __forceinline__ int project_points(const YmScanRef &sr, double px, double py, double pt, bool yag, double *sx, double *sy,
                                              int *s_cnt /* (YM_MAX_BEAMS / NT + 1) * NT / 64 */, int32_t *cidx_out = nullptr) {
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x;
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
        const int pos = running + before + __popcll(m & ((1ull << lane_) - 1ull));
        if (ok) {
            const double angle = pt + sr.min_angle + i * sr.angle_inc;
            sx[pos] = px + r * cos(angle);
            sy[pos] = py + r * sin(angle);
        }
        if (cidx_out && i < sr.n) cidx_out[i] = ok ? pos : -1;
        running += total;
    }
    __syncthreads();
    return running;
}

// The same with the compaction known (YmScanRef::cidx, from structure_kernel: which beams give a point reading, and where
// it goes, does not depend on the pose): one pass, one round of loads, one barrier.  gov != null: the scan's deciding
// pairs travel with it into LDS (l.nxt / l.ex are free when no chain is walked).
template <int NT>
__device__ __forceinline__ int project_points_indexed(const YmScanRef &sr, double px, double py, double pt, const PrepLds &l, bool with_gov) {
    for (int i = threadIdx.x; i < sr.n; i += NT) {
        const int pos = sr.cidx[i];
        const double r = sr.ranges[i];
        if (pos >= 0) {
            const double angle = pt + sr.min_angle + i * sr.angle_inc;
            l.sx[pos] = px + r * cos(angle);
            l.sy[pos] = py + r * sin(angle);
        }
    }
    if (with_gov) {
        const int2 *gov = reinterpret_cast<const int2 *>(sr.gov);
        for (int i = threadIdx.x; i < sr.cnp; i += NT) {
            const int2 g = gov[i];
            l.nxt[i] = g.x;
            l.ex[i] = g.y;
        }
    }
    __syncthreads();
    return sr.cnp;
}

// ---- the query-independent half of the valid-point filter (ScanMatcher::FindValidPoints / validate_points), parallel
// form: nxt[i] = first j > i farther than d from point i; the trigger chain is 0 -> nxt[0] -> ...; chain[i] = 1 for
// its nodes.  Marked without one long serial walk: cut the points into segments of 64; (1) every point walks to the
// first node past its own segment, (2) one thread hops from segment to segment with those exits (<= n/64 hops),
// (3) one thread per entered segment marks the chain nodes inside it.
// GUARD: also report (block-wide) whether any distance test came within YM_CHAIN_GUARD of the threshold.  The chain is a
// function of those tests' outcomes alone; squared distances between the same two readings computed at two different
// poses differ by rounding only (< 1e-11 m^2 for poses within 10 km and headings within 1000 rad: coordinates below 2^14 m
// carry errors below 4e-12 m, distances are at most 0.2 m), so a scan without a near test has the SAME chain at every
// such pose.
#define YM_CHAIN_GUARD 1e-9
#define YM_CHAIN_POSE_LIMIT 1.0e4
#define YM_CHAIN_HEADING_LIMIT 1.0e3 // (a heading of 1000 rad costs a beam angle 1e-13 rad of rounding: 3e-12 m at 30 m)
template <int NT, bool GUARD = false>
__device__ __forceinline__ int mark_chain(const PrepLds &l, int np, bool yag) {
    const int tid = threadIdx.x;
    const double min_sq = yag ? 0.2 * 0.2 : 0.1 * 0.1;
    int near = 0;
    for (int i = tid; i < np; i += NT) {
        const double fx = l.sx[i], fy = l.sy[i];
        int j = i + 1;
        for (; j < np; j++) {
            const double dx = fx - l.sx[j], dy = fy - l.sy[j];
            const double d2 = dx * dx + dy * dy;
            if (GUARD) near |= fabs(d2 - min_sq) <= YM_CHAIN_GUARD ? 1 : 0;
            if (d2 > min_sq) break;
        }
        l.nxt[i] = j;
        l.chain[i] = 0;
    }
    int unsafe = 0;
    if (GUARD) unsafe = __syncthreads_or(near);
    else __syncthreads();
    constexpr int SEG = 64;
    const int nseg = (np + SEG - 1) / SEG;
    for (int i = tid; i < np; i += NT) {
        const int seg_end = min(np, (i / SEG + 1) * SEG);
        int j = l.nxt[i];
        while (j < seg_end) j = l.nxt[j];
        l.ex[i] = j;
    }
    for (int i = tid; i < nseg; i += NT) l.ent[i] = -1;
    __syncthreads();
    if (tid == 0)
        for (int cur = 0; cur < np; cur = l.ex[cur]) l.ent[cur / SEG] = cur;
    __syncthreads();
    for (int sgi = tid; sgi < nseg; sgi += NT) {
        int c = l.ent[sgi];
        if (c >= 0) {
            const int seg_end = min(np, (sgi + 1) * SEG);
            for (; c < seg_end; c = l.nxt[c]) l.chain[c] = 1;
        }
    }
    __syncthreads();
    return unsafe;
}

// per point: the chain node that decides it (karto: the last node at or before i; yagpy: before i, none for point 0)
// and where that node's run ends (np = it never does)
__device__ __forceinline__ int2 gov_walk(const PrepLds &l, int i, int np, bool yag) {
    int s = yag ? i - 1 : i;
    if (s >= 0)
        while (!l.chain[s]) s--;
    return make_int2(s, s >= 0 ? l.nxt[s] : np);
}

// a projected base scan -> its slot of the matcher's point cache
template <int NT>
__device__ __forceinline__ void store_cache(const YmScanRef &sr, const PrepLds &l, int np, bool yag) {
    double2 *cpts = reinterpret_cast<double2 *>(sr.cache + YM_CACHE_HEADER);
    int2 *cgov = reinterpret_cast<int2 *>(sr.cache + YM_CACHE_HEADER + (size_t)sr.n * 16);
    const int2 *known = reinterpret_cast<const int2 *>(sr.gov);
    for (int i = threadIdx.x; i < np; i += NT) {
        cpts[i] = make_double2(l.sx[i], l.sy[i]);
        cgov[i] = known ? known[i] : gov_walk(l, i, np, yag);
    }
    if (threadIdx.x == 0) *reinterpret_cast<int *>(sr.cache) = np;
}

// sensor-frame coordinates of a projected query (karto: Transform(pose).InverseTransformPose; yagpy: points_local)
template <int NT>
__device__ __forceinline__ void store_query_local(const YmScanRef &sr, const PrepLds &l, int np, bool yag, double2 *ql) {
    const bool identity = yag || (sr.pose[0] == 0.0 && sr.pose[1] == 0.0 && sr.pose[2] == 0.0);
    const double cr = cos(0.0 - sr.pose[2]), sn = sin(0.0 - sr.pose[2]);
    for (int i = threadIdx.x; i < np; i += NT) {
        double2 v;
        if (identity) {
            v = make_double2(l.sx[i], l.sy[i]);
        } else {
            const double dx = l.sx[i] - sr.pose[0], dy = l.sy[i] - sr.pose[1];
            v = make_double2(cr * dx + (0.0 - sn) * dy, sn * dx + cr * dy);
        }
        ql[i] = v;
    }
}

// everything of an item that depends on its query's pose and the lattice, but not on the query's readings:
// state, (cos, sin) per coarse angle (GridIndexLookup::ComputeOffsets), hypothesis cells
// (np < 0: the number of point readings is written by someone else -- prepare_kernel's query block)
template <int NT>
__device__ __forceinline__ void init_item(const PrepareArgs &a, int b, const YmItem &it, const YmScanRef &sr, int np, int qslot,
                                          const double2 *ql, double off_x, double off_y) {
    const int tid = threadIdx.x;
    if (tid == 0) {
        YmItemState &st = a.states[b];
        st.pose[0] = sr.pose[0]; st.pose[1] = sr.pose[1]; st.pose[2] = sr.pose[2];
        st.center[0] = sr.pose[0]; st.center[1] = sr.pose[1]; st.center[2] = sr.pose[2];
        st.off_x = off_x;
        st.off_y = off_y;
        if (np >= 0) st.nq = np;
        st.status = 0;
        st.regular[0] = st.regular[1] = 0;
        st.base_count = it.base_count;
        st.qslot = qslot;
        st.ql = ql;
    }
    // one fp64 sin/cos per coarse angle; the cell offsets themselves are computed by the correlate blocks that
    // consume them
    if (tid < a.lat.nt) {
        const double angle = (sr.pose[2] - a.lat.angle_off) + tid * a.lat.angle_res;
        a.ctrig[(size_t)b * a.nt_stride + tid] = make_double2(cos(angle), sin(angle));
    }
    for (int i = tid; i < a.lat.nx * a.lat.ny; i += NT) a.probs[(size_t)b * a.lat.nx * a.lat.ny + i] = 0.0; // (score_hyp_kernel: atomic max)
    // coarse hypothesis cells + regularity flag
    int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
    int32_t *cy = cx + a.dim_stride;
    for (int i = tid; i < a.lat.nx; i += NT) cx[i] = hyp_cell(sr.pose[0], -a.lat.off_x, i, a.lat.step_x, off_x, a.g);
    for (int i = tid; i < a.lat.ny; i += NT) cy[i] = hyp_cell(sr.pose[1], -a.lat.off_y, i, a.lat.step_y, off_y, a.g);
    __syncthreads();
    {
        const int stx = kt_round_int(a.lat.step_x * a.g.scale), sty = kt_round_int(a.lat.step_y * a.g.scale);
        int ok = a.g.kpitch == 0; // (Karto's linear-index test over the whole storage: the per-cell paths)
        for (int i = tid; i < a.lat.nx; i += NT) ok &= (cx[i] == cx[0] + i * stx);
        for (int i = tid; i < a.lat.ny; i += NT) ok &= (cy[i] == cy[0] + i * sty);
        ok = __syncthreads_and(ok);
        if (tid == 0) a.states[b].regular[0] = ok;
    }
}

__device__ __forceinline__ void chain_check_cell(const PrepareArgs &a, int2 c) {
    if (a.fault && c.x != YM_CELL_NONE && (c.x < a.cell_box[0] || c.y < a.cell_box[1] || c.x > a.cell_box[2] || c.y > a.cell_box[3]))
        atomicCAS(a.fault, 0, a.step);
}

// The query-dependent half of FindValidPoints + AddScan's cell lookup for one base scan of one item: point i is kept
// iff its run's trigger pair (s, t) puts t on the far side of the line through the viewpoint and s; kept points
// become window cells, 64 consecutive cells share a bounding box.  PT(i) / GOV(i) read point i and its (s, t) from
// LDS (scan just projected) or from the matcher's point cache.
template <int NT, typename PT, typename GOV>
__device__ __forceinline__ void prepare_cells(const PrepareArgs &a, int b, int slot, int np, bool yag, double vpx, double vpy,
                                              double off_x, double off_y, PT pt_of, GOV gov_of) {
    const int tid = threadIdx.x;
    const int n_cchunks = YM_N_BOXES(a.max_n);
    int2 *cells = a.cells + ((size_t)b * a.max_base + slot) * a.max_n;
    int4 *bbox = a.bbox + ((size_t)b * a.max_base + slot) * n_cchunks;
    for (int i0 = 0; i0 < n_cchunks * YM_BOX_CELLS; i0 += NT) {
        const int i = i0 + tid; // a row of 16 lanes covers one chunk of YM_BOX_CELLS cells
        int2 c = make_int2(YM_CELL_NONE, YM_CELL_NONE);
        if (i < np) {
            bool keep = false;
            const int2 g = gov_of(i);
            if (g.x >= 0 && g.y < np) {
                const double2 f = pt_of(g.x), cp = pt_of(g.y);
                const double fx = f.x, fy = f.y, cx = cp.x, cy = cp.y;
                const double aa = vpy - fy;
                const double bb = fx - vpx;
                const double cc = fy * vpx - fx * vpy;
                const double ss = cx * aa + cy * bb + cc;
                keep = yag ? (ss > 0.0) : !(ss < 0.0);
            }
            if (keep) {
                const double2 p = pt_of(i);
                int gx, gy;
                if (yag) {
                    gx = (int)rint((p.x - off_x) / a.g.res);
                    gy = (int)rint((p.y - off_y) / a.g.res);
                } else {
                    gx = world_to_grid(p.x, off_x, a.g.scale);
                    gy = world_to_grid(p.y, off_y, a.g.scale);
                }
                if (gx >= 0 && gx < a.g.roi_w && gy >= 0 && gy < a.g.roi_w)
                    c = make_int2(gx + a.g.border - a.g.win_origin, gy + a.g.border - a.g.win_origin);
            }
        }
        if (i < a.max_n) cells[i] = c;
        chain_check_cell(a, c);
        // bounding box of the chunk's rasterised cells: the raster kernel reads a chunk only when
        // this box touches its tile
        const bool has = c.x != YM_CELL_NONE;
        const int x0 = row16_reduce(has ? c.x : INT32_MAX, OpMinI()), y0 = row16_reduce(has ? c.y : INT32_MAX, OpMinI());
        const int x1 = row16_reduce(has ? c.x : INT32_MIN, OpMaxI()), y1 = row16_reduce(has ? c.y : INT32_MIN, OpMaxI());
        if ((tid & (YM_BOX_CELLS - 1)) == 0 && i / YM_BOX_CELLS < n_cchunks) bbox[i / YM_BOX_CELLS] = make_int4(x0, y0, x1, y1);
    }
}

// an unused chain slot of a ragged batch: no points (select_kernel walks every slot's cells), empty boxes
template <int NT>
__device__ __forceinline__ void clear_slot(const PrepareArgs &a, int b, int slot) {
    const int n_cchunks = YM_N_BOXES(a.max_n);
    int2 *cells = a.cells + ((size_t)b * a.max_base + slot) * a.max_n;
    int4 *bbox = a.bbox + ((size_t)b * a.max_base + slot) * n_cchunks;
    for (int i = threadIdx.x; i < a.max_n; i += NT) cells[i] = make_int2(YM_CELL_NONE, YM_CELL_NONE);
    for (int i = threadIdx.x; i < n_cchunks; i += NT) bbox[i] = make_int4(INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN);
}

// one base scan of one item from the point cache: no LDS, no barrier.  The three dependent rounds of loads (who decides
// point i -> the two deciding points -> the point itself) are issued for Q chunks of the scan at a time, so a block
// waits three round trips per Q * NT points instead of three per NT.
template <int NT>
__device__ __forceinline__ void cells_from_cache(const PrepareArgs &a, int b, int slot, const YmScanRef &sr, const YmScanRef &qr, bool yag) {
    constexpr int Q = NT == 256 ? 3 : 4; // (cells_kernel, measured: Q = 4 398 us per 4096 items at 86 VGPRs, Q = 3 326 at 72, Q = 2 360 at 58)
    const double2 *__restrict__ cpts = reinterpret_cast<const double2 *>(sr.cache + YM_CACHE_HEADER);
    const int2 *__restrict__ cgov = reinterpret_cast<const int2 *>(sr.cache + YM_CACHE_HEADER + (size_t)sr.n * 16);
    const int np = *reinterpret_cast<const int *>(sr.cache);
    const double off_x = qr.pose[0] - (0.5 * (a.g.roi_w - 1) * a.g.res);
    const double off_y = qr.pose[1] - (0.5 * (a.g.roi_w - 1) * a.g.res);
    const double vpx = qr.pose[0], vpy = qr.pose[1];
    const int tid = threadIdx.x;
    const int n_cchunks = YM_N_BOXES(a.max_n);
    int2 *cells = a.cells + ((size_t)b * a.max_base + slot) * a.max_n;
    int4 *bbox = a.bbox + ((size_t)b * a.max_base + slot) * n_cchunks;
    for (int i0 = 0; i0 < n_cchunks * YM_BOX_CELLS; i0 += Q * NT) {
        int2 g[Q];
        double2 f[Q], t[Q], p[Q];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int i = i0 + q * NT + tid;
            g[q] = i < np ? cgov[i] : make_int2(-1, np);
        }
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int i = i0 + q * NT + tid;
            const bool decided = g[q].x >= 0 && g[q].y < np;
            f[q] = cpts[decided ? g[q].x : 0];
            t[q] = cpts[decided ? g[q].y : 0];
            p[q] = cpts[i < np ? i : 0];
        }
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const int i = i0 + q * NT + tid; // a wave covers one 64-cell chunk
            if (i0 + q * NT >= n_cchunks * YM_BOX_CELLS) break; // block-uniform
            int2 c = make_int2(YM_CELL_NONE, YM_CELL_NONE);
            if (i < np && g[q].x >= 0 && g[q].y < np) {
                const double fx = f[q].x, fy = f[q].y, cx = t[q].x, cy = t[q].y;
                const double aa = vpy - fy;
                const double bb = fx - vpx;
                const double cc = fy * vpx - fx * vpy;
                const double ss = cx * aa + cy * bb + cc;
                const bool keep = yag ? (ss > 0.0) : !(ss < 0.0);
                int gx, gy;
                if (yag) {
                    gx = (int)rint((p[q].x - off_x) / a.g.res);
                    gy = (int)rint((p[q].y - off_y) / a.g.res);
                } else {
                    gx = world_to_grid(p[q].x, off_x, a.g.scale);
                    gy = world_to_grid(p[q].y, off_y, a.g.scale);
                }
                if (keep && gx >= 0 && gx < a.g.roi_w && gy >= 0 && gy < a.g.roi_w)
                    c = make_int2(gx + a.g.border - a.g.win_origin, gy + a.g.border - a.g.win_origin);
            }
            if (i < a.max_n) cells[i] = c;
            chain_check_cell(a, c);
            const bool has = c.x != YM_CELL_NONE;
            const int x0 = row16_reduce(has ? c.x : INT32_MAX, OpMinI()), y0 = row16_reduce(has ? c.y : INT32_MAX, OpMinI());
            const int x1 = row16_reduce(has ? c.x : INT32_MIN, OpMaxI()), y1 = row16_reduce(has ? c.y : INT32_MIN, OpMaxI());
            if ((tid & (YM_BOX_CELLS - 1)) == 0 && i / YM_BOX_CELLS < n_cchunks) bbox[i / YM_BOX_CELLS] = make_int4(x0, y0, x1, y1);
        }
    }
}

// ---- K1, fused form (a few items): grid (max_base + 2, B), NT threads, dynamic LDS = YM_PREP_LDS_BYTES(max_n).
// blockIdx.x == 0: the query scan; blockIdx.x == 1 + j: base scan j of the item's chain; the last block: the item's
// state, coarse-angle table and hypothesis cells.  A base scan whose slot of
// the point cache is current takes the short path; otherwise it is projected here (and its slot filled).
template <int NT>
__global__ __launch_bounds__(NT) void prepare_kernel(PrepareArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int s_cnt[(YM_MAX_BEAMS / NT + 1) * (NT / 64)];
    if (a.stamps && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) a.stamps[28] = wall_clock64() - a.stamps[19]; // idle since the previous call's end
    YM_STAMP(a, 0);
    const int b = blockIdx.y;
    if (a.tile_max_zero && blockIdx.x == 0 && b == 0 && threadIdx.x == 0) *a.tile_max_zero = 0;
    const YmItem it = a.use_inline ? a.inl.item : a.items[b];
    const bool is_query = blockIdx.x == 0;
    const int slot = (int)blockIdx.x - 1;
    if ((int)blockIdx.x == a.max_base + 1) {
        // the item's block: everything that depends on the query's POSE and the lattice but not on its readings (state,
        // (cos, sin) per coarse angle -- fp64 sin / cos on a handful of lanes is a 1.3 us chain --, hypothesis cells, the
        // cleared maxima) runs beside the query's projection instead of after it
        YmScanRef qr = a.use_inline ? a.inl.scans[it.query] : a.scans[it.query];
        if (qr.pose_dev) { qr.pose[0] = qr.pose_dev[0]; qr.pose[1] = qr.pose_dev[1]; qr.pose[2] = qr.pose_dev[2]; }
        const double off_x = qr.pose[0] - (0.5 * (a.g.roi_w - 1) * a.g.res);
        const double off_y = qr.pose[1] - (0.5 * (a.g.roi_w - 1) * a.g.res);
        init_item<NT>(a, b, it, qr, -1, b, a.qlocal + (size_t)b * a.max_n, off_x, off_y);
        YM_STAMP_B1(a, 21);
        return;
    }
    if (!is_query && slot >= it.base_count) {
        clear_slot<NT>(a, b, slot);
        return;
    }
    const int si = is_query ? it.query : it.base_begin + slot;
    YmScanRef sr = a.use_inline ? a.inl.scans[si] : a.scans[si];
    YmScanRef qr = a.use_inline ? a.inl.scans[it.query] : a.scans[it.query];
    if (sr.pose_dev) { sr.pose[0] = sr.pose_dev[0]; sr.pose[1] = sr.pose_dev[1]; sr.pose[2] = sr.pose_dev[2]; }
    if (qr.pose_dev) { qr.pose[0] = qr.pose_dev[0]; qr.pose[1] = qr.pose_dev[1]; qr.pose[2] = qr.pose_dev[2]; }
    if (a.fault && *a.fault) { // a chained step after a fault: nothing to do (the host repeats it); leave an empty problem
        if (!is_query) { clear_slot<NT>(a, b, slot); return; }
        sr.n = 0;
    }
    const bool yag = a.g.semantics == 1;
    if (!is_query && sr.cache && !sr.stale) {
        cells_from_cache<NT>(a, b, slot, sr, qr, yag);
        return;
    }
    const PrepLds l = prep_lds(lds_raw, a.max_n);
    const double px = (is_query && yag) ? 0.0 : sr.pose[0];
    const double py = (is_query && yag) ? 0.0 : sr.pose[1];
    const double pt = (is_query && yag) ? 0.0 : sr.pose[2];
    const bool indexed = sr.cidx != nullptr && sr.n > 0; // (with cidx comes gov: the scan's structure is trusted at this pose)
    const int np = indexed ? project_points_indexed<NT>(sr, px, py, pt, l, !is_query) : project_points<NT>(sr, px, py, pt, yag, l.sx, l.sy, s_cnt);
    YM_STAMP(a, 1);
    YM_STAMP_B1(a, 20);
    // world offset of ROI cell (0,0): MatchScan, "set scan pose to be center of grid"
    const double off_x = qr.pose[0] - (0.5 * (a.g.roi_w - 1) * a.g.res);
    const double off_y = qr.pose[1] - (0.5 * (a.g.roi_w - 1) * a.g.res);
    if (is_query) {
        store_query_local<NT>(sr, l, np, yag, a.qlocal + (size_t)b * a.max_n);
        if (threadIdx.x == 0) { a.qnp[b] = np; a.states[b].nq = np; } // (the rest of the state: the item's block)
        YM_STAMP(a, 2);
        YM_STAMP(a, 18);
        return;
    }
    const int2 *known = reinterpret_cast<const int2 *>(sr.gov); // the chain structure, when it is known to hold at this pose
    if (!known) mark_chain<NT>(l, np, yag);
    YM_STAMP_B1(a, 22);
    if (sr.cache) store_cache<NT>(sr, l, np, yag);
    if (indexed) // (the deciding pairs came into LDS with the points)
        prepare_cells<NT>(a, b, slot, np, yag, qr.pose[0], qr.pose[1], off_x, off_y,
                          [&](int i) { return make_double2(l.sx[i], l.sy[i]); }, [&](int i) { return make_int2(l.nxt[i], l.ex[i]); });
    else if (known)
        prepare_cells<NT>(a, b, slot, np, yag, qr.pose[0], qr.pose[1], off_x, off_y,
                          [&](int i) { return make_double2(l.sx[i], l.sy[i]); }, [&](int i) { return known[i]; });
    else
        prepare_cells<NT>(a, b, slot, np, yag, qr.pose[0], qr.pose[1], off_x, off_y,
                          [&](int i) { return make_double2(l.sx[i], l.sy[i]); }, [&](int i) { return gov_walk(l, i, np, yag); });
    YM_STAMP_B1(a, 23);
}

// ---- K0 structure: grid (2), once per scan (ym_scan_create).  Block 0: Karto's rules, block 1: the Python matcher's.
// The trigger chain of the valid-point filter (which readings start a run, and where each run ends) depends on the
// readings' mutual distances only; computed here at the identity pose with a guard band around the distance threshold
// (mark_chain<GUARD>), it is what any pose within YM_CHAIN_POSE_LIMIT gives, bit for bit, unless a test was near the
// threshold -- then info says so and the matcher recomputes the chain from the projected points at every pose, as before.
struct StructureArgs {
    YmScanRef sr;       // ranges + sensor parameters
    int32_t *gov[2];    // [n][2] per semantics: for compacted point i the chain node that decides it and where its run ends
    int32_t *cidx[2];   // [n] per semantics: beam -> index of its point reading among the compacted ones, or -1
    int32_t *info;      // [2][2]: number of point readings, 1 = a distance test was within the guard band
    double *ranges_out; // null, or where block 0 copies the readings (sr.ranges is then the pinned host buffer they were staged in:
                        // the one launch is the upload)
    uint32_t *done;     // null, or [2] (pinned host memory): each block stores `serial` here when everything it writes is visible
    uint32_t serial;
    uint32_t pad;
};
template <int NT>
__device__ __forceinline__ void structure_body(const StructureArgs &a, int sem, unsigned char *lds_raw, int *s_cnt) {
    const bool yag = sem == 1;
    const PrepLds l = prep_lds(lds_raw, a.sr.n);
    // (no array of the argument record is indexed by a run-time value: that would put the record into scratch memory)
    const int np = project_points<NT>(a.sr, 0.0, 0.0, 0.0, yag, l.sx, l.sy, s_cnt, yag ? a.cidx[1] : a.cidx[0]);
    const int unsafe = mark_chain<NT, true>(l, np, yag);
    int2 *gov = reinterpret_cast<int2 *>(yag ? a.gov[1] : a.gov[0]);
    for (int i = threadIdx.x; i < np; i += NT) gov[i] = gov_walk(l, i, np, yag);
    if (a.ranges_out && sem == 0)
        for (int i = threadIdx.x; i < a.sr.n; i += NT) a.ranges_out[i] = a.sr.ranges[i];
    if (threadIdx.x == 0) { a.info[2 * sem] = np; a.info[2 * sem + 1] = unsafe; }
    if (a.done) {
        __syncthreads(); // every lane's stores are issued and waited for ...
        if (threadIdx.x == 0) {
            __threadfence_system(); // ... and written back before the host (and the kernels it launches next) can see the word
            __hip_atomic_store(a.done + sem, a.serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// one scan (ym_scan_create): grid (2) = one block per semantics
template <int NT>
__global__ __launch_bounds__(NT) void structure_kernel(StructureArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int s_cnt[(YM_MAX_BEAMS / NT + 1) * (NT / 64)];
    structure_body<NT>(a, (int)blockIdx.x, lds_raw, s_cnt);
}
// many scans (ym_scans_create): grid (2, scans), the argument records in device memory; completion by stream order (done == null)
template <int NT>
__global__ __launch_bounds__(NT) void structure_many_kernel(const StructureArgs *table) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int s_cnt[(YM_MAX_BEAMS / NT + 1) * (NT / 64)];
    const StructureArgs a = table[blockIdx.y];
    structure_body<NT>(a, (int)blockIdx.x, lds_raw, s_cnt);
}

// ---- K1 for batches, first half: the heavy, query-independent work ONCE per distinct scan of the call.
// grid (n_jobs), 256 threads, dynamic LDS = YM_PREP_LDS_BYTES(max_n).  A query job projects the query and leaves its
// sensor-frame points in its query slot; a base job projects a base scan whose cache slot is stale and fills it.
#define YM_POINTS_THREADS 256
__global__ __launch_bounds__(YM_POINTS_THREADS) void points_kernel(PrepareArgs a) {
    constexpr int NT = YM_POINTS_THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int s_cnt[(YM_MAX_BEAMS / NT + 1) * (NT / 64)];
    const int job = a.jobs[blockIdx.x];
    const bool is_query = job < 0;
    const YmScanRef sr = a.scans[job & 0x7fffffff];
    const bool yag = a.g.semantics == 1;
    const PrepLds l = prep_lds(lds_raw, a.max_n);
    const double px = (is_query && yag) ? 0.0 : sr.pose[0];
    const double py = (is_query && yag) ? 0.0 : sr.pose[1];
    const double pt = (is_query && yag) ? 0.0 : sr.pose[2];
    const int np = (sr.cidx && sr.n > 0) ? project_points_indexed<NT>(sr, px, py, pt, l, false)
                                         : project_points<NT>(sr, px, py, pt, yag, l.sx, l.sy, s_cnt);
    if (is_query) { // into the scan's query slot of the point cache, else into the call's own buffer
        const int qs = a.job_slot[blockIdx.x];
        store_query_local<NT>(sr, l, np, yag, sr.qcache ? reinterpret_cast<double2 *>(sr.qcache + YM_CACHE_HEADER) : a.qlocal + (size_t)qs * a.max_n);
        if (threadIdx.x == 0) *(sr.qcache ? reinterpret_cast<int *>(sr.qcache) : a.qnp + qs) = np;
        return;
    }
    if (!sr.gov) mark_chain<NT>(l, np, yag);
    store_cache<NT>(sr, l, np, yag);
}

// ---- K1 for batches, second half: grid (max_base + 1, B), 256 threads, no dynamic LDS.  Block 0 of an item sets the
// item up around its query's slot; block 1 + j turns base scan j's cached points into this item's cells.
__global__ __launch_bounds__(256) void cells_kernel(PrepareArgs a) {
    constexpr int NT = 256;
    const int b = blockIdx.y;
    if (a.tile_max_zero && blockIdx.x == 0 && b == 0 && threadIdx.x == 0) *a.tile_max_zero = 0;
    const YmItem it = a.items[b];
    const int slot = (int)blockIdx.x - 1;
    const YmScanRef qr = a.scans[it.query];
    if (blockIdx.x == 0) {
        const int qs = it.pad; // query slot, set by the host
        const double off_x = qr.pose[0] - (0.5 * (a.g.roi_w - 1) * a.g.res);
        const double off_y = qr.pose[1] - (0.5 * (a.g.roi_w - 1) * a.g.res);
        const double2 *ql = qr.qcache ? reinterpret_cast<const double2 *>(qr.qcache + YM_CACHE_HEADER) : a.qlocal + (size_t)qs * a.max_n;
        init_item<NT>(a, b, it, qr, qr.qcache ? *reinterpret_cast<const int *>(qr.qcache) : a.qnp[qs], qs, ql, off_x, off_y);
        return;
    }
    if (slot >= it.base_count) {
        clear_slot<NT>(a, b, slot);
        return;
    }
    cells_from_cache<NT>(a, b, slot, a.scans[it.base_begin + slot], qr, a.g.semantics == 1);
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
    const int4 *bbox;        // [B][max_base][YM_N_BOXES(max_n)]
    uint32_t *tile_list;     // [B][tile_cap] tile index (tiy * tiles_x + tix), | 0x8000 = only needs clearing, | hits << 16 (0xffff =
                             // more than YM_TILE_HITS: the raster block walks the item's boxes)
    int32_t *tile_count;     // [B]
    int32_t *tile_max;       // longest list of the call (zeroed by the host before the launch)
    const uint8_t *tile_zero;// [B][tiles_y][tiles_x] 1 = the tile's memory is known to hold zeros
    int32_t max_n, max_base, half_kernel;
    int32_t tiles_x, tiles_y, tile_cap;
    int32_t launch[4];       // tile rectangle (x0, y0, x1, y1) the raster covers in this call
    // per LIST ENTRY the chunks that reach its tile, so that a raster block neither scans every box of the item nor looks
    // anything up before it can load them (their address depends on the entry's position alone):
    uint16_t *hits;          // [B][tile_cap][YM_TILE_HITS] index of the chunk's first cell in the item's cells; null = no lists (the raster walks the boxes)
    int32_t tile_h;          // rows per tile in this call (YM_TILE_H or YM_TILE_H_TALL)
    int32_t hit_limit;       // hit slots in use, <= YM_TILE_HITS (tests lower it)
};
// grid (B), dynamic LDS = 4 * ceil(tiles_x * tiles_y / 32) bytes (+ 8 bytes per tile of the rectangle with hit lists)
__global__ __launch_bounds__(YM_TILES_THREADS) void tiles_kernel(TilesArgs a) {
    extern __shared__ unsigned tile_bits[];
    __shared__ int s_n;
    constexpr int NT = YM_TILES_THREADS;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int ntiles = a.tiles_x * a.tiles_y, nwords = (ntiles + 31) / 32;
    const int h = a.half_kernel;
    const int lx0 = a.launch[0], ly0 = a.launch[1], lx1 = a.launch[2], ly1 = a.launch[3];
    const int ltx = lx1 - lx0 + 1, lty = ly1 - ly0 + 1, nsub = ltx * lty;
    int *cnt = reinterpret_cast<int *>(tile_bits + nwords); // [nsub] hits per tile of the rectangle
    int *fill = cnt + nsub;                                 // [nsub] list position << 16 | hit slots filled
    for (int i = tid; i < nwords; i += NT) tile_bits[i] = 0u;
    if (a.hits)
        for (int i = tid; i < nsub; i += NT) cnt[i] = 0;
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int n_boxes = a.max_base * YM_N_BOXES(a.max_n);
    const int4 *bbox = a.bbox + (size_t)b * n_boxes;
    const int th_shift = a.tile_h == YM_TILE_H_TALL ? 6 : 5; // (tile heights are 32 or 64: a runtime division costs ~25 instructions)
    static_assert(YM_TILE_H == 32 && YM_TILE_H_TALL == 64, "tile heights as shifts");
    for (int c = tid; c < n_boxes; c += NT) {
        const int4 bb = bbox[c];
        if (bb.x > bb.z) continue;
        // tiles whose halo-extended rectangle [t*T - h, t*T + T + h - 1] meets the box (the raster kernel's own test)
        const int tx0 = max(lx0, max(bb.x - h, 0) / YM_TILE_W), tx1 = min(lx1, (bb.z + h) / YM_TILE_W);
        const int ty0 = max(ly0, max(bb.y - h, 0) >> th_shift), ty1 = min(ly1, (bb.w + h) >> th_shift);
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) {
                const int t = ty * a.tiles_x + tx;
                atomicOr(&tile_bits[t >> 5], 1u << (t & 31));
                if (a.hits) atomicAdd(&cnt[(ty - ly0) * ltx + (tx - lx0)], 1);
            }
    }
    __syncthreads();
    const uint8_t *tz = a.tile_zero + (size_t)b * ntiles;
    uint32_t *list = a.tile_list + (size_t)b * a.tile_cap;
    for (int i = tid; i < nsub; i += NT) {
        const int ty = ly0 + i / ltx, tx = lx0 + i % ltx, t = ty * a.tiles_x + tx;
        const bool hit = (tile_bits[t >> 5] >> (t & 31)) & 1u;
        if (hit || tz[t] == 0) {
            const int at = atomicAdd(&s_n, 1);
            const uint32_t nh = !a.hits ? 0xffffu : cnt[i] > a.hit_limit ? 0xffffu : (uint32_t)cnt[i];
            list[at] = (uint32_t)(t | (hit ? 0 : 0x8000)) | nh << 16;
            if (a.hits) fill[i] = at << 16;
        }
    }
    __syncthreads();
    if (tid == 0) {
        a.tile_count[b] = s_n;
        if (s_n > *reinterpret_cast<volatile int32_t *>(a.tile_max)) atomicMax(a.tile_max, s_n); // (few blocks raise it)
    }
    if (!a.hits) return;
    // the hit slots: every box again, to the entries of the tiles it reaches (in any order: the raster ORs bits)
    uint16_t *hits = a.hits + (size_t)b * a.tile_cap * YM_TILE_HITS;
    const int n_cchunks = YM_N_BOXES(a.max_n);
    for (int c = tid; c < n_boxes; c += NT) {
        const int4 bb = bbox[c];
        if (bb.x > bb.z) continue;
        const int slot = c / n_cchunks;
        const uint16_t first_cell = (uint16_t)(slot * a.max_n + (c - slot * n_cchunks) * YM_BOX_CELLS); // (the host: max_base * max_n < 65536)
        const int tx0 = max(lx0, max(bb.x - h, 0) / YM_TILE_W), tx1 = min(lx1, (bb.z + h) / YM_TILE_W);
        const int ty0 = max(ly0, max(bb.y - h, 0) >> th_shift), ty1 = min(ly1, (bb.w + h) >> th_shift);
        for (int ty = ty0; ty <= ty1; ty++)
            for (int tx = tx0; tx <= tx1; tx++) {
                const int i = (ty - ly0) * ltx + (tx - lx0);
                if (cnt[i] > a.hit_limit) continue;
                const int old = atomicAdd(&fill[i], 1);
                hits[(size_t)(old >> 16) * YM_TILE_HITS + (old & 0xffff)] = first_cell;
            }
    }
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
// the same search, continued at bucket bk
__device__ __forceinline__ int select_find_from(const unsigned *keys, unsigned bmask, unsigned bk, unsigned key) {
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
    constexpr int NT = 1024, NW = (NB - 1) / 4;
    constexpr int PMAX = 12; // points per thread: the host sends at most 12 288 readings here (3 * capacity >= 4 * readings, capacity <= 2^14)
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
    // The thread's points: all loaded at once and kept for step (3), with the slots step (1) finds for them.  (The one block
    // of an item is a latency chain: a load per loop trip -- which the stores of step (3) into the same array keep the
    // compiler from moving -- cost steps (1) and (3) a memory round trip per point, 14 + 58 of the kernel's 118 us.)
    int2 *cells = a.cells + (size_t)b * a.max_base * a.max_n;
    const int total = a.max_base * a.max_n;
    int2 mine[PMAX];
    unsigned short slot_of[PMAX];
    // A thread takes `per` CONSECUTIVE points and, in steps (2a) / (2b), decides the cells whose earliest point is one of
    // them, in that order: the earlier neighbours of a cell are mostly the cells of the beams just before it, so a wall's
    // chain of dependent decisions resolves inside one thread in one sweep instead of hopping from thread to thread once
    // per sweep (the relaxation took 53 of the kernel's 111 us while a thread owned hash SLOTS).
    const int per = (total + NT - 1) / NT, e0 = tid * per;
#pragma unroll
    for (int q = 0; q < PMAX; q++) {
        const int e = e0 + q;
        mine[q] = (q < per && e < total) ? cells[e] : make_int2(YM_CELL_NONE, YM_CELL_NONE);
    }
    for (unsigned i = tid; i < cap; i += NT) { keys[i] = 0u; minidx[i] = 0xffffffffu; status[i] = 0; }
    __syncthreads();
    YM_STAMP(a, 25);
    // (1) cell -> earliest point index
#pragma unroll
    for (int q = 0; q < PMAX; q++) {
        slot_of[q] = 0;
        if (mine[q].x == YM_CELL_NONE) continue;
        const int slot = select_insert(keys, bmask, shift, select_key(mine[q].x, mine[q].y));
        slot_of[q] = (unsigned short)slot;
        atomicMin(&minidx[slot], (unsigned)(e0 + q));
    }
    __syncthreads();
    YM_STAMP(a, 26);
    // (2a) per owned cell (= owned point that is the earliest of its cell): the slots of the neighbour cells that hold an
    // EARLIER point, packed as 16-bit slot numbers (0xffff = none).  A cell with no earlier neighbour is effective.
    unsigned long long nb[PMAX][NW];
    unsigned und = 0;
#pragma unroll
    for (int k = 0; k < PMAX; k++) {
#pragma unroll
        for (int w = 0; w < NW; w++) nb[k][w] = ~0ull;
        if (mine[k].x == YM_CELL_NONE) continue;
        const unsigned s = slot_of[k], me = (unsigned)(e0 + k);
        if (minidx[s] != me) continue; // a later point of its cell: never effective, nothing to decide
        const int x = mine[k].x, y = mine[k].y;
        bool any = false;
        // the first probe of every neighbour is issued before any is looked at (the probes are latency, not bandwidth)
        unsigned nkey[NB], nbk[NB];
        uint4 nq[NB];
#pragma unroll
        for (int n = 1; n < NB; n++) {
            nkey[n] = select_key(x + DX[n], y + DY[n]);
            nbk[n] = (nkey[n] * 2654435761u) >> shift;
            nq[n] = *reinterpret_cast<const uint4 *>(keys + 4 * nbk[n]);
        }
#pragma unroll
        for (int n = 1; n < NB; n++) {
            const uint4 q4 = nq[n];
            const unsigned key = nkey[n];
            int t = q4.x == key ? (int)(4 * nbk[n]) : q4.y == key ? (int)(4 * nbk[n] + 1) : q4.z == key ? (int)(4 * nbk[n] + 2)
                    : q4.w == key ? (int)(4 * nbk[n] + 3) : q4.w == 0u ? -1 : -2;
            if (t == -2) t = select_find_from(keys, bmask, (nbk[n] + 1u) & bmask, key); // (a full bucket without the key: go on)
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
    YM_STAMP(a, 3);
    // (2b) asynchronous relaxation, no barriers: every wave keeps re-reading the state of the earlier
    // neighbours of its undecided cells.  A decision is final and is taken only from final states
    // (a stale "undecided" read merely delays it), and all 16 waves of the block are resident, so
    // this terminates with the sequential greedy result whatever the interleaving.
    unsigned char *vst = status;
    unsigned wmask = 0;
#pragma unroll
    for (int k = 0; k < PMAX; k++) wmask |= __ballot((und >> k) & 1u) ? (1u << k) : 0u;
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
                const unsigned s = slot_of[k];
                if (knocked) vst[s] = 2;
                else if (!pending) vst[s] = 1;
                else still = true;
                if (!still) und &= ~(1u << k);
            }
            if (__ballot(still) == 0ull) wmask &= ~(1u << k);
        }
    }
    __syncthreads();
    YM_STAMP(a, 30);
    // (3) keep only the earliest point of every effective cell
#pragma unroll
    for (int q = 0; q < PMAX; q++) {
        if (mine[q].x == YM_CELL_NONE) continue;
        const int e = e0 + q, t = slot_of[q];
        if (!(status[t] == 1 && minidx[t] == (unsigned)e)) cells[e] = make_int2(YM_CELL_NONE, YM_CELL_NONE);
    }
    YM_STAMP(a, 31);
}

// ---- the same rule on a FEW items (z2max = 1), split by what is parallel and what is not (round 4).  Steps (1) and (2a) of
// select_kernel are parallel work that one block does alone: 45 of its 79 us are 16 waves issuing ~4000 instructions each on
// one CU.  Here they are two launches over all points (hash and earliest-point table in global memory, left zeroed for the
// next call), and the one block per item that remains only runs the chain of dependent decisions (2b) and the erase (3) from
// a 16-byte record per point (SelectSplitArgs::rec): rec.x = slot | owned << 16 | effective << 17 (0xffffffff: no point), rec.y / rec.z = the slots of
// the neighbour cells with an earlier point, 16 bits each (0xffff: none).
struct SelectSplitArgs {
    int2 *cells;          // [B][max_base][max_n]
    int32_t max_n, max_base;
    int32_t log2cap, pad;
    unsigned *keys;       // [B][cap]  zero between calls
    unsigned *mx;         // [B][cap]  zero between calls: max over the cell's points of ~index (= its earliest point)
    unsigned *slot_of;    // [B][max_base * max_n]  the point's slot (0xffffffff: no cell)
    uint4 *rec;           // [B][12 * 1024]  the record of point e = t * per + q at [q * 1024 + t]: thread t of select_relax_kernel reads its
                          //                 `per` = ceil(points / 1024) consecutive points with coalesced loads
    unsigned long long *stamps;
};
#define YM_SELECT_SPLIT_THREADS 256
// (1) grid (ceil(total / 256), B)
__global__ __launch_bounds__(YM_SELECT_SPLIT_THREADS) void select_hash_kernel(SelectSplitArgs a) {
    const int b = blockIdx.y, e = blockIdx.x * YM_SELECT_SPLIT_THREADS + threadIdx.x;
    const int total = a.max_base * a.max_n;
    if (e >= total) return;
    const unsigned cap = 1u << a.log2cap, bmask = (cap >> 2) - 1u;
    const int shift = 32 - (a.log2cap - 2);
    const int2 c = a.cells[(size_t)b * total + e];
    unsigned slot = 0xffffffffu;
    if (c.x != YM_CELL_NONE) {
        slot = (unsigned)select_insert(a.keys + (size_t)b * cap, bmask, shift, select_key(c.x, c.y));
        atomicMax(&a.mx[(size_t)b * cap + slot], ~(unsigned)e);
    }
    a.slot_of[(size_t)b * total + e] = slot;
}
// (2a) grid (ceil(total / 256), B)
__global__ __launch_bounds__(YM_SELECT_SPLIT_THREADS) void select_neighbours_kernel(SelectSplitArgs a) {
    constexpr int DX[5] = {0, 1, -1, 0, 0};
    constexpr int DY[5] = {0, 0, 0, 1, -1};
    const int b = blockIdx.y, e = blockIdx.x * YM_SELECT_SPLIT_THREADS + threadIdx.x;
    const int total = a.max_base * a.max_n;
    if (e >= total) return;
    const unsigned cap = 1u << a.log2cap, bmask = (cap >> 2) - 1u;
    const int shift = 32 - (a.log2cap - 2);
    const unsigned *keys = a.keys + (size_t)b * cap, *mx = a.mx + (size_t)b * cap;
    const int per = (total + 1023) / 1024;
    uint4 *rec = a.rec + (size_t)b * 12 * 1024 + (size_t)(e % per) * 1024 + e / per;
    const unsigned s = a.slot_of[(size_t)b * total + e];
    const int2 c = a.cells[(size_t)b * total + e]; // (a point without a cell: YM_CELL_NONE, any bucket)
    // the first probe of every neighbour leaves before the point's own entry is looked at: one round trip, not two
    unsigned nkey[5], nbk[5];
    uint4 nq[5];
#pragma unroll
    for (int n = 1; n < 5; n++) {
        nkey[n] = select_key(c.x + DX[n], c.y + DY[n]);
        nbk[n] = (nkey[n] * 2654435761u) >> shift;
        nq[n] = *reinterpret_cast<const uint4 *>(keys + 4 * nbk[n]);
    }
    if (s == 0xffffffffu) {
        *rec = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0u);
        return;
    }
    const unsigned mine = ~(unsigned)e;
    if (mx[s] != mine) { // a later point of its cell: never effective, nothing to decide
        *rec = make_uint4(s, 0xffffffffu, 0xffffffffu, 0u);
        return;
    }
    int tn[5];
#pragma unroll
    for (int n = 1; n < 5; n++) {
        const uint4 q4 = nq[n];
        const unsigned key = nkey[n];
        int t = q4.x == key ? (int)(4 * nbk[n]) : q4.y == key ? (int)(4 * nbk[n] + 1) : q4.z == key ? (int)(4 * nbk[n] + 2)
                : q4.w == key ? (int)(4 * nbk[n] + 3) : q4.w == 0u ? -1 : -2;
        if (t == -2) t = select_find_from(keys, bmask, (nbk[n] + 1u) & bmask, key); // (a full bucket without the key: go on)
        tn[n] = t;
    }
    unsigned mt[5];
#pragma unroll
    for (int n = 1; n < 5; n++) mt[n] = mx[tn[n] >= 0 ? tn[n] : (int)s]; // (the four entries together)
    unsigned nbv[4];
    bool any = false;
#pragma unroll
    for (int n = 1; n < 5; n++) {
        const bool earlier = tn[n] >= 0 && mt[n] > mine; // (~index larger = index smaller)
        nbv[n - 1] = earlier ? (unsigned)tn[n] : 0xffffu;
        any |= earlier;
    }
    *rec = make_uint4(s | 1u << 16 | (any ? 0u : 1u << 17), nbv[0] | nbv[1] << 16, nbv[2] | nbv[3] << 16, 0u);
}
// (2b) + (3): grid (B), 1024 threads, dynamic LDS = one state byte per slot
__global__ __launch_bounds__(1024) void select_relax_kernel(SelectSplitArgs a) {
    constexpr int NT = 1024, PMAX = 12;
    extern __shared__ unsigned sel_lds[];
    unsigned char *status = reinterpret_cast<unsigned char *>(sel_lds); // 0 undecided, 1 effective, 2 out
    const int b = blockIdx.x, tid = threadIdx.x;
    const unsigned cap = 1u << a.log2cap;
    const int total = a.max_base * a.max_n;
    int2 *cells = a.cells + (size_t)b * total;
    const uint4 *rec = a.rec + (size_t)b * 12 * 1024;
    // a thread takes `per` CONSECUTIVE points (see select_kernel: a wall's chain of decisions resolves inside one thread)
    const int per = (total + NT - 1) / NT, e0 = tid * per;
    YM_STAMP(a, 24);
    uint4 r[PMAX];
#pragma unroll
    for (int q = 0; q < PMAX; q++) {
        const int e = e0 + q;
        r[q] = (q < per && e < total) ? rec[q * NT + tid] : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0u);
    }
    for (unsigned i = tid; i < cap / 4; i += NT) sel_lds[i] = 0u;
    __syncthreads();
    unsigned und = 0;
#pragma unroll
    for (int q = 0; q < PMAX; q++) {
        if (r[q].x == 0xffffffffu || !(r[q].x & (1u << 16))) continue;
        if (r[q].x & (1u << 17)) status[r[q].x & 0xffffu] = 1;
        else und |= 1u << q;
    }
    YM_STAMP(a, 3);
    // asynchronous relaxation, no barriers (select_kernel, step (2b))
    unsigned char *vst = status;
    unsigned wmask = 0;
#pragma unroll
    for (int k = 0; k < PMAX; k++) wmask |= __ballot((und >> k) & 1u) ? (1u << k) : 0u;
    while (wmask) {
#pragma unroll
        for (int k = 0; k < PMAX; k++) { // (unrolled: r[k] stays in registers; a point without undecided lanes costs a scalar test)
            if (!((wmask >> k) & 1u)) continue;
            bool still = false;
            asm volatile("" ::: "memory"); // re-read the states every time
            if ((und >> k) & 1u) {
                bool knocked = false, pending = false;
                const unsigned nbs[4] = {r[k].y & 0xffffu, r[k].y >> 16, r[k].z & 0xffffu, r[k].z >> 16};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (nbs[j] != 0xffffu) {
                        const unsigned char stt = vst[nbs[j]];
                        knocked |= stt == 1;
                        pending |= stt == 0;
                    }
                }
                const unsigned s = r[k].x & 0xffffu;
                if (knocked) vst[s] = 2;
                else if (!pending) vst[s] = 1;
                else still = true;
                if (!still) und &= ~(1u << k);
            }
            if (__ballot(still) == 0ull) wmask &= ~(1u << k);
        }
    }
    __syncthreads();
    YM_STAMP(a, 30);
    // (3) keep only the earliest point of every effective cell
#pragma unroll
    for (int q = 0; q < PMAX; q++) {
        if (r[q].x == 0xffffffffu) continue;
        if (!((r[q].x & (1u << 16)) && status[r[q].x & 0xffffu] == 1)) cells[e0 + q] = make_int2(YM_CELL_NONE, YM_CELL_NONE);
    }
    // the tables as the next call expects them
    uint4 *k4 = reinterpret_cast<uint4 *>(a.keys + (size_t)b * cap), *m4 = reinterpret_cast<uint4 *>(a.mx + (size_t)b * cap);
    for (unsigned i = tid; i < cap / 4; i += NT) { k4[i] = make_uint4(0u, 0u, 0u, 0u); m4[i] = make_uint4(0u, 0u, 0u, 0u); }
    YM_STAMP(a, 31);
}

// ---- the same rule for chains too long for one CU's LDS (more than 12 288 readings): hash, earliest-point index, neighbour
// lists and states live in global memory (one 1024-thread block per item again -- the relaxation is a chain of
// dependent decisions, not a parallel job -- but its reads now cost an L2 round trip instead of an LDS one).  LDS only
// holds each thread's shrinking list of undecided slots.  Capacity 2^17 slots = 98 304 readings at load factor 0.75.
struct SelectGlobalArgs {
    int2 *cells;          // [B][max_base][max_n]
    int32_t max_n, max_base;
    int32_t z2max, log2cap;
    unsigned *keys;       // [B][cap]           zeroed by the host
    unsigned *status;     // [B][cap]           zeroed by the host: 0 undecided, 1 effective, 2 out
    unsigned *minidx;     // [B][cap]           set to 0xffffffff by the host
    unsigned *nbr;        // [B][NB - 1][cap]   slot of the neighbour cell if it holds an earlier point, else 0xffffffff
};

template <int NB>
__global__ __launch_bounds__(1024) void select_global_kernel(SelectGlobalArgs a) {
    constexpr int NT = 1024;
    constexpr int DX[9] = {0, 1, -1, 0, 0, 1, 1, -1, -1};
    constexpr int DY[9] = {0, 0, 0, 1, -1, 1, -1, 1, -1};
    extern __shared__ unsigned char und_list[]; // [cap / NT][NT]: ordinals k of the slots tid + k * NT still undecided
    const int b = blockIdx.x, tid = threadIdx.x;
    const unsigned cap = 1u << a.log2cap, bmask = (cap >> 2) - 1u;
    const int shift = 32 - (a.log2cap - 2);
    unsigned *keys = a.keys + (size_t)b * cap, *minidx = a.minidx + (size_t)b * cap, *status = a.status + (size_t)b * cap;
    unsigned *nbr = a.nbr + (size_t)b * (NB - 1) * cap;
    int2 *cells = a.cells + (size_t)b * a.max_base * a.max_n;
    const int total = a.max_base * a.max_n;
    // (1) cell -> earliest point index
    for (int e = tid; e < total; e += NT) {
        const int2 c = cells[e];
        if (c.x == YM_CELL_NONE) continue;
        const int slot = select_insert(keys, bmask, shift, select_key(c.x, c.y));
        atomicMin(&minidx[slot], (unsigned)e);
    }
    // the tables were built by atomics in L2: make them visible to this CU's plain loads (its L1 may hold lines fetched
    // while they were still being filled)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // (2a) earlier neighbours of every cell; a cell without one is effective
    int n_und = 0;
    for (unsigned s = tid, k = 0; s < cap; s += NT, k++) {
        const unsigned key = keys[s];
        if (key == 0u) continue;
        const unsigned me = minidx[s];
        const int x = (int)(key & 0xffffu) - 32768, y = (int)(key >> 16) - 32768;
        bool any = false;
#pragma unroll
        for (int n = 1; n < NB; n++) {
            const int t = select_find(keys, bmask, shift, select_key(x + DX[n], y + DY[n]));
            const bool earlier = t >= 0 && minidx[t] < me;
            nbr[(size_t)(n - 1) * cap + s] = earlier ? (unsigned)t : 0xffffffffu;
            any |= earlier;
        }
        if (any) und_list[(size_t)(n_und++) * NT + tid] = (unsigned char)k;
        else __hip_atomic_store(&status[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (2b) asynchronous relaxation (see select_kernel): decisions are final and only taken from final states, so stale
    // reads merely delay; every wave of the block is resident, so the earliest undecided cell always gets decided
    while (__ballot(n_und > 0)) {
        int kept = 0;
        for (int i = 0; i < n_und; i++) {
            const unsigned k = und_list[(size_t)i * NT + tid], s = tid + k * NT;
            bool knocked = false, pending = false;
#pragma unroll
            for (int j = 0; j < NB - 1; j++) {
                const unsigned t = nbr[(size_t)j * cap + s];
                if (t != 0xffffffffu) {
                    const unsigned stt = __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    knocked |= stt == 1u;
                    pending |= stt == 0u;
                }
            }
            if (knocked) __hip_atomic_store(&status[s], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (!pending) __hip_atomic_store(&status[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else und_list[(size_t)(kept++) * NT + tid] = (unsigned char)k;
        }
        n_und = kept;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // (3) keep only the earliest point of every effective cell
    for (int e = tid; e < total; e += NT) {
        const int2 c = cells[e];
        if (c.x == YM_CELL_NONE) continue;
        const int t = select_find(keys, bmask, shift, select_key(c.x, c.y));
        if (!(t >= 0 && __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u && minidx[t] == (unsigned)e))
            cells[e] = make_int2(YM_CELL_NONE, YM_CELL_NONE);
    }
}

}  // namespace ym
