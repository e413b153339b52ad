// ym_k_item.hpp -- K4i: the coarse correlate of large batches on lattices up to 26 x 32, ONE block per item (round 4).
// Part of ym_kernels.hpp (include that, not this file).
//
// correlate_region_kernel (ym_k_region.hpp) gives every coarse angle a wave of its own: three blocks of eight waves per item,
// each staging the item's regions for its eight angles.  What binds it (profiles/r04_region_study.md): its gather loop runs
// at the vector-issue rate, 1.9 of the kernel's 2.85 ms, and a wave spends a quarter of its life in it -- the rest at the
// barriers, where it waits for the wave whose angle has the most patches in the region (0 to 150 of them; 43 on average),
// and in the staging phases, which every region goes through three times per item.
// Here a block owns ALL angles of an item and its sixteen waves share the patches of a region EVENLY, whatever their angle:
//   * the pooled entry list of a region -- bin_kernel's bins of region R, angle 0 .. nt - 1, contiguous and each a multiple
//     of four entries -- is cut into sixteen runs of quads; a wave gathers its run angle segment by angle segment into the
//     packed 16-bit registers of rg_gather4 and, when the angle changes or the run ends, adds them to the item's 32-bit
//     sums in LDS (acc32[angle][hypothesis 0..12][lane], ds_add_u32: two waves may be at work on one angle);
//   * a region is staged once per item (a third of the first form's staging loads, stores and L2 requests), by 1024
//     threads (three 16-byte tasks each instead of six), and the two barriers of a round now close ~1100 patches instead
//     of ~430, with every wave arriving at about the same time;
//   * the sums never leave the CU: no sets of partial sums in global memory, no flush; the block scores them from LDS
//     (wave w: angles w, w + 16, ...).
// 70 KB of sums + 45 KB region + 8 KB entries: one block per CU, four waves per SIMD -- what the gather loop needs to reach
// its issue rate (scripts/exp/rg_proto.hip: 11.4 CU clocks per patch at two blocks of eight waves).
// The lists are the first form's (bin_kernel with 8 angles per box: a region's box is the union of its angle blocks' boxes).
#pragma once

namespace ym {

#define YM_IT_NW 16                          // waves per block
#define YM_IT_LPS 36                         // rows the staging threads of a class image cover at once (216 of its 256 threads copy)
#define YM_IT_IMG_ROWS (3 * YM_IT_LPS)       // rows of a class image in LDS: three bands, >= YM_RG_ROWS, so that no copy task leaves its image
#define YM_IT_CLS (YM_RG_PITCH * YM_IT_IMG_ROWS)
#define YM_IT_ZERO (4 * YM_IT_CLS)           // LDS offset of the all-zero patch
#define YM_IT_LDS_BYTES (YM_IT_ZERO + 26 * YM_RG_PITCH + 32)
#define YM_IT_MAXE 4096                      // pooled entries of a region the block holds in LDS (more: read from global memory)
#define YM_IT_KSTRIDE (YM_RG_G * 64)         // dwords of sums per angle
#define YM_IT_ACC_BYTES(nt) ((size_t)(nt) * YM_IT_KSTRIDE * 4)
#define YM_IT_MAX_NT 31                      // 31 x 3328 B of sums + the static 55 KB stay below 160 KB

// grid (B), 1024 threads, dynamic LDS = YM_IT_ACC_BYTES(nt)
__global__ __launch_bounds__(64 * YM_IT_NW, 4) void correlate_item_kernel(RegionArgs a) {
    constexpr int NW = YM_IT_NW, NT = 64 * NW;
    constexpr int TPC = NT / 4;                       // staging threads per class image
    constexpr int LPS = YM_IT_LPS;                    // rows the threads of a class cover at once
    constexpr int PER = 3;                            // copy tasks per thread
    static_assert(LPS * YM_RG_SEGS <= TPC && LPS * PER >= YM_RG_ROWS && LPS * PER == YM_IT_IMG_ROWS, "the copy tasks cover a class image and stay inside it");
    static_assert(YM_IT_LDS_BYTES < 65536, "entries are 16-bit LDS offsets");
    __shared__ __attribute__((aligned(16))) unsigned char region[YM_IT_LDS_BYTES]; // four class images + the zero patch
    __shared__ __attribute__((aligned(16))) unsigned short elist[YM_IT_MAXE];      // the pooled entries of the region being gathered
    __shared__ int rlist[YM_RG_MAX_REGIONS];
    __shared__ uint32_t rboxl[YM_RG_MAX_REGIONS];
    __shared__ int bstart[2][YM_MAX_COARSE_NT + 1];   // first entry of every angle's bin of the region being gathered / staged
    __shared__ int rcount;
    extern __shared__ __attribute__((aligned(16))) uint32_t acc32[]; // [nt][13][64]
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const YmItemState &st = a.states[b];
    const int nt = a.lat.nt, nx = a.lat.nx, ny = a.lat.ny;
    const int row = lane & 31, half = lane >> 5;
    const bool job = row < ny && half * YM_RG_G < nx;
    const int half_pitch = a.g.pitch / 2;
    const int plane_bytes = half_pitch * a.g.win_w;
    const uint8_t *__restrict__ planes = a.planes + (size_t)b * a.grid_stride;
    const int32_t *__restrict__ starts = a.starts + (size_t)st.qslot * a.starts_stride;
    const uint16_t *__restrict__ entries = a.entries + (size_t)st.qslot * a.entries_stride;
    const uint32_t lds0 = (uint32_t)(size_t)region;
    // idle lanes read what lane (row 0, same half) reads: the same address is a broadcast
    const uint32_t lane_off = lds0 + (uint32_t)((job ? row : 0) * YM_RG_PITCH + (half * YM_RG_G < nx ? half * YM_RG_G : 0));
    for (int i = tid; i < nt * YM_IT_KSTRIDE; i += NT) acc32[i] = 0u;
    // a run's packed 16-bit sums -> the item's 32-bit sums of angle k
    auto deposit = [&](uint32_t (&acc)[8], int k) {
        rg_odd(acc);
        if (job) {
            uint32_t *dst = acc32 + (size_t)k * YM_IT_KSTRIDE + lane;
#pragma unroll
            for (int j = 0; j < YM_RG_G; j++) {
                const uint32_t v = (acc[2 * (j >> 2) + (j & 1)] >> (16 * ((j >> 1) & 1))) & 0xffffu;
                if (half * YM_RG_G + j < nx) atomicAdd(&dst[j * 64], v);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0u;
    };
    const bool regular = st.regular[0] && a.force_irregular != 1 && starts[a.nbins] >= 0;
    if (regular) {
        const int nreg = a.nregions;
        for (int i = tid; i < (YM_IT_LDS_BYTES - YM_IT_ZERO) / 4; i += NT) reinterpret_cast<uint32_t *>(region + YM_IT_ZERO)[i] = 0u;
        if (wave == 0) { // the regions in which a patch of any angle starts
            int n = 0;
            for (int R0 = 0; R0 < nreg; R0 += 64) {
                const int R = R0 + lane;
                const bool has = R < nreg && starts[(size_t)R * nt] != starts[(size_t)(R + 1) * nt];
                const unsigned long long mask = __ballot(has);
                if (has) rlist[n + __popcll(mask & ((1ull << lane) - 1ull))] = R;
                n += __popcll(mask);
            }
            if (lane == 0) rcount = n;
        }
        __syncthreads();
        const int nlist = rcount;
        {
            // what the item's patches read of a listed region: the union of its angle blocks' boxes (bin_kernel keeps one per
            // block of a.nw angles), rows rmin .. rmax + ny - 1, bytes (xmin & ~3) .. xmax + 15 (+ 13 for the second half of a row)
            const uint32_t *rb = a.rbox + (size_t)st.qslot * a.rbox_stride;
            const uint32_t reach = (uint32_t)(nx > YM_RG_G ? 15 + YM_RG_G : 15);
            for (int i = tid; i < nlist; i += NT) {
                uint32_t rmin = 255u, rmax = 0u, xmin = 255u, xmax = 0u;
                for (int q = 0; q < a.parts; q++) {
                    const uint32_t v = rb[(size_t)rlist[i] * a.parts + q];
                    if ((v & 0xffu) > ((v >> 8) & 0xffu)) continue; // (an angle block without a patch here: 255 | 0)
                    rmin = min(rmin, v & 0xffu); rmax = max(rmax, (v >> 8) & 0xffu);
                    xmin = min(xmin, (v >> 16) & 0xffu); xmax = max(xmax, v >> 24);
                }
                const uint32_t r0 = rmin, r1 = min((uint32_t)(YM_RG_ROWS - 1), rmax + (uint32_t)ny - 1u);
                const uint32_t s0 = (xmin & 0xfcu) >> 4, s1 = min((uint32_t)(YM_RG_SEGS - 1), (xmax + reach) >> 4);
                rboxl[i] = r0 | r1 << 8 | s0 << 16 | s1 << 24;
            }
        }
        __syncthreads(); // (the boxes are read by every wave)
        // Copy tasks: a thread owns one 16-byte segment `seg` of the rows r0, r0 + LPS, ... of ONE class image (as in
        // correlate_region_kernel; whole bands of LPS rows are skipped by a scalar branch when the box does not reach them)
        const uint32_t cls = (uint32_t)tid / TPC, j0 = (uint32_t)tid - cls * TPC;
        const uint32_t seg = j0 % YM_RG_SEGS, r0 = j0 / YM_RG_SEGS;
        const bool copier = j0 < (uint32_t)(LPS * YM_RG_SEGS);
        const uint32_t src0 = (cls & 1u) * (uint32_t)plane_bytes + (2u * r0 + (cls >> 1)) * (uint32_t)half_pitch + 16u * seg;
        const uint32_t src_step = 2u * LPS * (uint32_t)half_pitch;
        const uint32_t dst0 = (cls * YM_IT_IMG_ROWS + r0) * YM_RG_PITCH + 16u * seg;
        uint4 v[PER];
        auto band_in = [&](int q, uint32_t bx) {
            return (uint32_t)(q * LPS) <= ((bx >> 8) & 0xffu) && (uint32_t)(q * LPS + LPS - 1) >= (bx & 0xffu);
        };
        auto stage_load = [&](int R, uint32_t bx) {
            const int RX = R % a.nrx, RY = R / a.nrx;
            const uint8_t *src = planes + ((size_t)(2 * RY * YM_RG_H) * half_pitch + (size_t)RX * YM_RG_W); // (wave-uniform)
            const bool seg_in = copier && seg >= ((bx >> 16) & 0xffu) && seg <= (bx >> 24);
#pragma unroll
            for (int q = 0; q < PER; q++)
                if (band_in(q, bx)) v[q] = *reinterpret_cast<const uint4 *>(src + (seg_in ? src0 + (uint32_t)q * src_step : 0u));
        };
        auto stage_store = [&](uint32_t bx) {
            const bool seg_in = copier && seg >= ((bx >> 16) & 0xffu) && seg <= (bx >> 24);
#pragma unroll
            for (int q = 0; q < PER; q++) {
                uint32_t *d = reinterpret_cast<uint32_t *>(region + dst0 + (uint32_t)(q * LPS * YM_RG_PITCH));
                if (band_in(q, bx) && seg_in) { d[0] = v[q].x; d[1] = v[q].y; d[2] = v[q].z; d[3] = v[q].w; }
            }
        };
        // the pooled entries of a region, four per thread (the first YM_IT_MAXE of them), and its bin starts: loaded into
        // registers while the previous region is gathered, stored between the barriers
        const uint2 *__restrict__ entries4 = reinterpret_cast<const uint2 *>(entries);
        uint2 ev = make_uint2(0u, 0u);
        int bv = 0;
        auto lists_load = [&](int R) {
            const int32_t *srow = starts + (size_t)R * nt;
            const int e0 = srow[0], e1 = srow[nt]; // (wave-uniform; the last region's end is starts[nbins])
            ev = make_uint2(0u, 0u);
            if (e0 + 4 * tid < e1) ev = entries4[(e0 >> 2) + tid];
            bv = tid <= nt ? srow[tid] : 0;
        };
        auto lists_store = [&](int which) {
            reinterpret_cast<uint2 *>(elist)[tid] = ev;
            if (tid <= nt) bstart[which][tid] = bv;
        };
        // One wave's share of the region's quads [q0, q1) (quad = four entries of one angle's bin): its angle segments in turn.
        // The entries of a quad are read one quad ahead (LDS), across the segments' borders.
        auto gather_run = [&](const int *bs, int q0, int q1) {
            if (q0 >= q1) return;
            const int base = __builtin_amdgcn_readfirstlane(bs[0]); // (entries of the region: [bs[0], bs[nt]))
            int k = 0;
            while (k + 1 < nt && (__builtin_amdgcn_readfirstlane(bs[k + 1]) - base) <= 4 * q0) k++; // the angle whose bin holds quad q0
            int kend = (__builtin_amdgcn_readfirstlane(bs[k + 1]) - base) >> 2; // first quad of the next angle's bin
            uint32_t acc[8];
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = 0u;
            const uint2 *el = reinterpret_cast<const uint2 *>(elist);
            const int q1l = min(q1, YM_IT_MAXE / 4); // (quads beyond the LDS list -- a region with more than YM_IT_MAXE padded entries -- from global memory)
            int in_run = 0;
            auto step = [&](int c, uint2 e) {
                while (c >= kend) { // the angle changes: the run's sums so far belong to angle k
                    if (in_run) deposit(acc, k);
                    in_run = 0;
                    k++;
                    kend = (__builtin_amdgcn_readfirstlane(bs[k + 1]) - base) >> 2;
                }
                rg_gather4(acc, lane_off, e);
                in_run += 4;
                if (in_run == YM_RG_FLUSH) { deposit(acc, k); in_run = 0; } // (16-bit sums; never reached by a run of <= 256 quads)
            };
            int c = q0;
            if (c < q1l) {
                uint2 e0 = el[c];
                for (; c < q1l; c++) {
                    const uint2 e = e0;
                    e0 = el[min(c + 1, q1l - 1)];
                    step(c, e);
                }
            }
            for (; c < q1; c++) step(c, entries4[(base >> 2) + c]);
            if (in_run) deposit(acc, k);
        };
        int cur = 0;
        if (nlist > 0) {
            const uint32_t bx = __builtin_amdgcn_readfirstlane(rboxl[0]);
            lists_load(rlist[0]);
            stage_load(rlist[0], bx);
            stage_store(bx);
            lists_store(0);
        }
        __syncthreads();
        for (int ri = 0; ri < nlist; ri++) {
            // two-stage pipeline: the global loads of the next region are in flight (registers) while this one is gathered
            const bool has_next = ri + 1 < nlist;
            uint32_t nbx = 0u;
            if (has_next) {
                nbx = __builtin_amdgcn_readfirstlane(rboxl[ri + 1]);
                const int Rn = __builtin_amdgcn_readfirstlane(rlist[ri + 1]);
                lists_load(Rn);
                stage_load(Rn, nbx);
            }
            {
                const int *bs = bstart[cur];
                const int nq4 = (__builtin_amdgcn_readfirstlane(bs[nt]) - __builtin_amdgcn_readfirstlane(bs[0])) >> 2;
                gather_run(bs, (int)(((long long)nq4 * wave) / NW), (int)(((long long)nq4 * (wave + 1)) / NW));
            }
            __syncthreads(); // every wave is done with region ri
            if (has_next) {
                stage_store(nbx);
                lists_store(cur ^ 1);
            }
            cur ^= 1;
            __syncthreads();
        }
    } else {
        // hypothesis cells are not an exact lattice (possible only through fp rounding), or the query's lists did not fit:
        // per-cell path over the window, wave w takes the angles w, w + NW, ...
        __syncthreads();
        const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
        const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
        const int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
        const int32_t *cy = cx + a.dim_stride;
        const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
        const int nq = st.nq;
        for (int k = wave; k < nt; k += NW) {
            const double2 cs = a.ctrig[(size_t)b * a.nt_stride + k];
            if (job)
                for (int j = 0; j < YM_RG_G; j++) {
                    const int ix = half * YM_RG_G + j;
                    if (ix >= nx) break;
                    const int base = cy[row] * lin_pitch(a.g) + cx[ix];
                    unsigned sum = 0;
                    for (int i = 0; i < nq; i++)
                        sum += cell_value(a.g, grid, limit, (unsigned)(base + lookup_offset(ql[i], cs.x, cs.y, st.off_x, st.off_y, a.g.scale, lin_pitch(a.g))));
                    acc32[(size_t)k * YM_IT_KSTRIDE + j * 64 + lane] = sum;
                }
        }
        __syncthreads();
    }
    // ---- the sums are complete (the last barrier above)
    if (!a.fuse_score) {
        // parity tests / kept integer sums: the sums as a.ng sets of 16-bit partials in the region correlate's layout (score_kernel adds them)
        for (int k = wave; k < nt; k += NW)
            for (int f = 0; f < a.ng; f++) {
                uint32_t w[8];
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    uint32_t lo = 0u, hi = 0u;
                    if (j < YM_RG_G) { const uint32_t s = acc32[(size_t)k * YM_IT_KSTRIDE + j * 64 + lane]; lo = s > 65535u * f ? min(s - 65535u * f, 65535u) : 0u; }
                    if (j + 1 < YM_RG_G) { const uint32_t s = acc32[(size_t)k * YM_IT_KSTRIDE + (j + 1) * 64 + lane]; hi = s > 65535u * f ? min(s - 65535u * f, 65535u) : 0u; }
                    w[j >> 1] = lo | hi << 16;
                }
                uint16_t *out = a.partial + (size_t)b * a.partial_stride + (((size_t)f * nt + k) * 64 + lane) * 16;
                *reinterpret_cast<uint4 *>(out) = make_uint4(w[0], w[1], w[2], w[3]);
                *reinterpret_cast<uint4 *>(out + 8) = make_uint4(w[4], w[5], w[6], w[7]);
            }
        return;
    }
    // ---- score (score_kernel's arithmetic, statement for statement): response, penalty, block maxima; the per-(x, y) maximum
    // over theta goes through LDS (the region buffer is free now) so that only one atomic per cell reaches memory
    unsigned long long *pmax = reinterpret_cast<unsigned long long *>(region); // [ny * nx] fp64 bit patterns, >= 0
    const int nxy = nx * ny;
    double *dpen = reinterpret_cast<double *>(region) + ((nxy + 1) & ~1); // [ny * nx] distance penalty of every cell
    for (int i = tid; i < nxy; i += NT) {
        pmax[i] = 0ull;
        const int iy = i / nx, ix = i - iy * nx;
        const double x = -a.lat.off_x + ix * a.lat.step_x, y = -a.lat.off_y + iy * a.lat.step_y;
        dpen[i] = dist_penalty(a.g, x * x + y * y);
    }
    __syncthreads();
    {
        const double ct = st.center[2];
        const int nq = st.nq;
        const int ncb = (nxy + YM_SCORE_THREADS - 1) / YM_SCORE_THREADS;
        const int c0 = row * nx + half * YM_RG_G, cb0 = job ? c0 / YM_SCORE_THREADS : 0;
        for (int k = wave; k < nt; k += NW) {
            const double angle = (ct - a.lat.angle_off) + k * a.lat.angle_res;
            double bmax0 = -1.0, bmax1 = -1.0; // block maxima this lane contributes to (its 13 cells span at most 2 blocks)
#pragma unroll
            for (int j = 0; j < YM_RG_G; j++) {
                const int ix = half * YM_RG_G + j;
                if (job && ix < nx) {
                    const int c = row * nx + ix;
                    const unsigned tot = acc32[(size_t)k * YM_IT_KSTRIDE + j * 64 + lane];
                    const double r = hyp_response_dp(a.g, a.lat.penalize, tot, nq, dpen[c], angle, ct);
                    a.resp[(size_t)b * a.sums_stride + (size_t)k * nxy + c] = r;
                    if (c / YM_SCORE_THREADS == cb0) bmax0 = r > bmax0 ? r : bmax0;
                    else bmax1 = r > bmax1 ? r : bmax1;
                    if (r > 0.0) atomicMax(&pmax[c], (unsigned long long)__double_as_longlong(r));
                }
            }
            for (int cb = 0; cb < ncb; cb++) {
                const double mine = !job ? -1.0 : cb == cb0 ? bmax0 : cb == cb0 + 1 ? bmax1 : -1.0;
                const double m = wave_reduce(mine, OpMaxD());
                if (lane == 0) a.blockmax[(size_t)b * a.n_blocks + (size_t)k * ncb + cb] = m;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < nxy; i += NT)
        if (pmax[i]) atomicMax(reinterpret_cast<unsigned long long *>(a.probs) + (size_t)b * a.probs_stride + i, pmax[i]);
}

} // namespace ym
