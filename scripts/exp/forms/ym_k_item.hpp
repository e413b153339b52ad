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
#define YM_IT_LPS ((YM_RG_ROWS + 2) / 3)     // rows the staging threads of a class image cover at once (three bands; 36 rows = 216 of its 256 threads copy)
#define YM_IT_IMG_ROWS (3 * YM_IT_LPS)       // rows of a class image in LDS: three bands, >= YM_RG_ROWS, so that no copy task leaves its image
#define YM_IT_CLS (YM_RG_PITCH * YM_IT_IMG_ROWS)
#define YM_IT_ZERO (4 * YM_IT_CLS)           // LDS offset of the all-zero patch
#define YM_IT_LDS_BYTES (YM_IT_ZERO + 26 * YM_RG_PITCH + 32)
#define YM_IT_MAXE 4096                      // pooled entries of a region the block holds in LDS (more: read from global memory)
#define YM_IT_KSTRIDE (YM_RG_G * 64)         // dwords of sums per angle
#define YM_IT_ACC_BYTES(nt) ((size_t)(nt) * YM_IT_KSTRIDE * 4)
#define YM_IT_MAX_NT 31                      // 31 x 3328 B of sums + the static 55 KB stay below 160 KB

#ifdef YM_EXPERIMENTAL // (both forms measured slower than correlate_region_kernel: compiled only with -DYM_EXPERIMENTAL, debug option 32 = 3 / 4)
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

// ================================================================== K4p: the pooled form at two blocks per item (round 4, second half)
// correlate_item_kernel (above) lost because its 115 KB of LDS leave one block per CU.  What it is after -- the waves of a block
// sharing a region's patches EVENLY instead of each waiting for the wave whose angle has the most (the region correlate's rounds
// cost the maximum over eight angles: 1.49 x the mean on the bench query) -- also works with HALF an item per block: eleven (ten)
// angles, twelve waves, the sums of the block's angles as packed 16-bit pairs in LDS (2 KB per angle, the registers' own format:
// a deposit is seven ds_add_u32), the region staged from the row-major window: 73 KB, two blocks = 24 waves per CU.
//   * lists, boxes, region geometry: the region correlate's (bin_kernel with nw = 11: two angle blocks);
//   * a region's pooled entries -- the bins of the block's angles, contiguous -- are dealt to the waves in quads; a wave gathers its
//     share angle part by angle part into registers and adds them to the angle's LDS sums when the angle changes;
//   * 16-bit sums hold 652 patches: angle k's padded entries are counted through the walk (cum), and its sums leave LDS as partial
//     set m exactly when the count passes 652 (m + 1) -- a bin that straddles such a boundary is gathered in two SUB-ROUNDS with the
//     flush between them (a barrier each side; ~one per angle and item) -- so the sets are the region correlate's sets;
//   * the block scores its angles from LDS (+ the sets it wrote), wave w the angles w, w + 12, ...
#endif // YM_EXPERIMENTAL
#define YM_PL_NW 12
#define YM_PL_MAX_NK 11
#define YM_PL_MAXE 2048
#ifdef YM_EXPERIMENTAL
template <bool WIN>
__global__ __launch_bounds__(64 * YM_PL_NW, 6 /* two blocks = 24 waves per CU: 80 VGPRs */) void correlate_pool_kernel(RegionArgs a) {
    constexpr int NW = YM_PL_NW, NT = 64 * NW;
    constexpr int TPC = WIN ? NT / 2 : NT / 4;
    constexpr int NSEG = WIN ? 2 * YM_RG_SEGS : YM_RG_SEGS;
    constexpr int LPS = TPC / NSEG;                   // rows the threads of a class (WIN: row parity) cover at once
    constexpr int PER = (YM_RG_ROWS + LPS - 1) / LPS; // copy tasks per thread
    __shared__ __attribute__((aligned(16))) unsigned char region[YM_RG_LDS_BYTES]; // four class images + the zero patch
    __shared__ __attribute__((aligned(16))) unsigned short elist[YM_PL_MAXE];      // the pooled entries of the region being gathered
    __shared__ uint32_t sum16[YM_PL_MAX_NK][8][64];   // per angle of the block: the lanes' packed 16-bit sums (rg_gather4's registers)
    __shared__ int rlist[YM_RG_MAX_REGIONS];
    __shared__ uint32_t rboxl[YM_RG_MAX_REGIONS];
    __shared__ int bstart[2][YM_PL_MAX_NK + 1];       // first entry of every angle's bin of the region being gathered / staged
    __shared__ int bcum[2][YM_PL_MAX_NK];             // padded entries of the angle before that region
    __shared__ int s_sets[YM_PL_MAX_NK];
    __shared__ int rcount;
    int p;
    const int b = xcd_item_of_block_2d(p);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const YmItemState &st = a.states[b];
    const int nt = a.lat.nt, nx = a.lat.nx, ny = a.lat.ny, ng = a.ng;
    const int k_lo = p * a.nw, k_hi = min(nt, k_lo + a.nw), nk = k_hi - k_lo;
    const int row = lane & 31, half = lane >> 5;
    const bool job = row < ny && half * YM_RG_G < nx;
    const int half_pitch = a.g.pitch / 2;
    const int plane_bytes = half_pitch * a.g.win_w;
    const uint8_t *__restrict__ planes = a.planes + (size_t)b * a.grid_stride;
    const uint8_t *__restrict__ window = a.grid + (size_t)b * a.grid_stride;
    const int32_t *__restrict__ starts = a.starts + (size_t)st.qslot * a.starts_stride;
    const uint16_t *__restrict__ entries = a.entries + (size_t)st.qslot * a.entries_stride;
    const uint32_t lds0 = (uint32_t)(size_t)region;
    const uint32_t lane_off = lds0 + (uint32_t)((job ? row : 0) * YM_RG_PITCH + (half * YM_RG_G < nx ? half * YM_RG_G : 0));
    uint16_t *partial = a.partial + (size_t)b * a.partial_stride;
    for (int i = tid; i < YM_PL_MAX_NK * 8 * 64; i += NT) (&sum16[0][0][0])[i] = 0u;
    // angle kk's LDS sums leave as partial set f (score_kernel's layout 1: 16 halves per lane, in hypothesis order as
    // store_partial16 writes them: the halves of a register pair (4j | 4j + 2, 4j + 1 | 4j + 3) become 4j, 4j + 1 | 4j + 2, 4j + 3)
    // and start again at zero
    auto flush_angle = [&](int kk, int f) {
        for (int i = tid; i < 4 * 64; i += NT) {
            const int jj = i >> 6, l = i & 63;
            const uint32_t lo = sum16[kk][2 * jj][l], hi = sum16[kk][2 * jj + 1][l];
            if (f < ng) {
                uint32_t *d = reinterpret_cast<uint32_t *>(partial + (((size_t)f * nt + (k_lo + kk)) * 64 + l) * 16) + 2 * jj;
                d[0] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
                d[1] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);
            }
            sum16[kk][2 * jj][l] = 0u; sum16[kk][2 * jj + 1][l] = 0u;
        }
    };
    const bool listed = st.regular[0] && a.force_irregular != 1 && starts[a.nbins] >= 0;
    const bool regular = listed && nk > 0;
    int my_cum = 0; // (thread kk < nk: the padded entries of angle k_lo + kk in the regions walked so far)
    if (regular) {
        const int nreg = a.nregions;
        for (int i = tid; i < (YM_RG_LDS_BYTES - YM_RG_ZERO) / 4; i += NT) reinterpret_cast<uint32_t *>(region + YM_RG_ZERO)[i] = 0u;
        if (wave == 0) { // the regions in which a patch of this block's angles starts
            int n = 0;
            for (int R0 = 0; R0 < nreg; R0 += 64) {
                const int R = R0 + lane;
                const bool has = R < nreg && starts[(size_t)R * nt + k_lo] != starts[(size_t)R * nt + k_hi];
                const unsigned long long mask = __ballot(has);
                if (has) rlist[n + __popcll(mask & ((1ull << lane) - 1ull))] = R;
                n += __popcll(mask);
            }
            if (lane == 0) rcount = n;
        }
        __syncthreads();
        const int nlist = rcount;
        {
            const uint32_t *rb = a.rbox + (size_t)st.qslot * a.rbox_stride;
            const uint32_t reach = (uint32_t)(nx > YM_RG_G ? 15 + YM_RG_G : 15);
            for (int i = tid; i < nlist; i += NT) {
                const uint32_t v = rb[(size_t)rlist[i] * a.parts + p];
                const uint32_t r0 = v & 0xffu, r1 = min((uint32_t)(YM_RG_ROWS - 1), ((v >> 8) & 0xffu) + (uint32_t)ny - 1u);
                const uint32_t s0 = ((v >> 16) & 0xfcu) >> 4, s1 = min((uint32_t)(YM_RG_SEGS - 1), ((v >> 24) + reach) >> 4);
                rboxl[i] = r0 | r1 << 8 | s0 << 16 | s1 << 24;
            }
        }
        __syncthreads();
        // copy tasks, as in correlate_region_kernel (WIN: a 16-byte segment of a window row = eight class bytes of each column parity)
        const uint32_t cls = WIN ? 2u * ((uint32_t)tid / TPC) : (uint32_t)tid / TPC, j0 = (uint32_t)tid % TPC;
        const uint32_t seg = j0 % NSEG, r0 = j0 / NSEG;
        const bool copier = j0 < (uint32_t)(LPS * NSEG);
        const uint32_t src0 = WIN ? (2u * r0 + (cls >> 1)) * (uint32_t)a.g.pitch + 16u * seg
                                  : (cls & 1u) * (uint32_t)plane_bytes + (2u * r0 + (cls >> 1)) * (uint32_t)half_pitch + 16u * seg;
        const uint32_t src_step = WIN ? 2u * LPS * (uint32_t)a.g.pitch : 2u * LPS * (uint32_t)half_pitch;
        const uint32_t dst0 = (cls * YM_RG_ROWS + r0) * YM_RG_PITCH + (WIN ? 8u : 16u) * seg;
        uint4 v[PER];
        auto band_in = [&](int q, uint32_t bx) {
            return (uint32_t)(q * LPS) <= ((bx >> 8) & 0xffu) && (uint32_t)(q * LPS + LPS - 1) >= (bx & 0xffu);
        };
        auto seg_inside = [&](uint32_t bx) {
            return WIN ? copier && seg >= 2u * ((bx >> 16) & 0xffu) && seg <= 2u * (bx >> 24) + 1u
                       : copier && seg >= ((bx >> 16) & 0xffu) && seg <= (bx >> 24);
        };
        auto stage_load = [&](int R, uint32_t bx) {
            const int RX = R % a.nrx, RY = R / a.nrx;
            const uint8_t *src = WIN ? window + ((size_t)(2 * RY * YM_RG_H) * a.g.pitch + (size_t)RX * (2 * YM_RG_W))
                                     : planes + ((size_t)(2 * RY * YM_RG_H) * half_pitch + (size_t)RX * YM_RG_W);
            const bool seg_in = seg_inside(bx);
#pragma unroll
            for (int q = 0; q < PER; q++)
                if (band_in(q, bx)) v[q] = *reinterpret_cast<const uint4 *>(src + (seg_in ? src0 + (uint32_t)q * src_step : 0u));
        };
        auto stage_store = [&](uint32_t bx) {
            const bool seg_in = seg_inside(bx);
#pragma unroll
            for (int q = 0; q < PER; q++) {
                uint32_t *d = reinterpret_cast<uint32_t *>(region + dst0 + (uint32_t)(q * LPS * YM_RG_PITCH));
                if (band_in(q, bx) && seg_in && r0 + (uint32_t)(q * LPS) < (uint32_t)YM_RG_ROWS) { // (the last band runs past the image)
                    if (WIN) {
                        d[0] = __builtin_amdgcn_perm(v[q].y, v[q].x, 0x06040200u); d[1] = __builtin_amdgcn_perm(v[q].w, v[q].z, 0x06040200u);
                        d[YM_RG_CLS / 4] = __builtin_amdgcn_perm(v[q].y, v[q].x, 0x07050301u); d[YM_RG_CLS / 4 + 1] = __builtin_amdgcn_perm(v[q].w, v[q].z, 0x07050301u);
                    } else { d[0] = v[q].x; d[1] = v[q].y; d[2] = v[q].z; d[3] = v[q].w; }
                }
            }
        };
        // the pooled entries of a region (the first YM_PL_MAXE of them) and its bin starts: registers while the previous region is
        // gathered, LDS between the barriers; thread kk < nk also keeps angle kk's running count
        const uint2 *__restrict__ entries4 = reinterpret_cast<const uint2 *>(entries);
        uint2 ev = make_uint2(0u, 0u);
        int bv = 0, bnext = 0;
        auto lists_load = [&](int R) {
            const int32_t *srow = starts + (size_t)R * nt + k_lo;
            const int e0 = srow[0], e1 = srow[nk]; // (wave-uniform; the last bin's end is starts[nbins])
            ev = make_uint2(0u, 0u);
            if (e0 + 4 * tid < e1 && 4 * tid < YM_PL_MAXE) ev = entries4[(e0 >> 2) + tid];
            bv = tid <= nk ? srow[tid] : 0;
            bnext = tid < nk ? srow[tid + 1] : 0;
        };
        auto lists_store = [&](int which) {
            if (4 * tid < YM_PL_MAXE) reinterpret_cast<uint2 *>(elist)[tid] = ev;
            if (tid <= nk) bstart[which][tid] = bv;
            if (tid < nk) { bcum[which][tid] = my_cum; my_cum += bnext - bv; }
        };
        auto deposit = [&](uint32_t (&acc)[8], int kk) {
#pragma unroll
            for (int j = 0; j < 7; j++) atomicAdd(&sum16[kk][j][lane], acc[j]);
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = 0u;
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
            const bool has_next = ri + 1 < nlist;
            uint32_t nbx = 0u;
            if (has_next) {
                nbx = __builtin_amdgcn_readfirstlane(rboxl[ri + 1]);
                const int Rn = __builtin_amdgcn_readfirstlane(rlist[ri + 1]);
                lists_load(Rn);
                stage_load(Rn, nbx);
            }
            {
                // Lane kk < nk of every wave holds angle kk's numbers of the region being gathered (two LDS reads per round; a
                // scalar loop over the angles with three LDS round trips each cost more than the gather itself): bin [e0, e1),
                // count c0 before it; its portion in sub-round s = the entries that belong to set c0 / FLUSH + s.
                const int *bs = bstart[cur], *cm = bcum[cur];
                const int e0 = lane < nk ? bs[lane] : 0, e1 = lane < nk ? bs[lane + 1] : 0, c0 = lane < nk ? cm[lane] : 0;
                const int base = __builtin_amdgcn_readfirstlane(e0); // (lane 0: the first entry of the region's pooled list)
                const int len = e1 - e0, set0 = c0 / YM_RG_FLUSH;
                const int mysub = len > 0 ? (c0 + len - 1) / YM_RG_FLUSH - set0 + 1 : 1;
                const int nsub = wave_reduce(mysub, OpMaxI());
                const unsigned first_flush = (unsigned)__ballot(len > 0 && c0 > 0 && c0 % YM_RG_FLUSH == 0);
                for (int s = 0; s < nsub; s++) {
                    // this lane's angle in sub-round s: entries [lo, hi), nq4 quads, `pre` pooled quads before it
                    const int lo = max(e0, e0 + (set0 + s) * YM_RG_FLUSH - c0), hi = max(lo, min(e1, e0 + (set0 + s + 1) * YM_RG_FLUSH - c0));
                    const int nq4 = (hi - lo) >> 2;
                    int incl = nq4;
#pragma unroll
                    for (int d = 1; d < 16; d <<= 1) {
                        const int t = __shfl_up(incl, d);
                        if (lane >= d) incl += t;
                    }
                    const int total = __builtin_amdgcn_readlane(incl, 15); // (nk <= 11 < 16; the lanes past nk hold zero quads)
                    const int pre = incl - nq4;
                    const unsigned fl = s == 0 ? first_flush : (unsigned)__ballot(hi > lo);
                    if (fl) { // (block-uniform: every wave computes the same numbers)
                        if (s > 0) __syncthreads(); // every deposit of the sub-round before is in
                        for (int kk = 0; kk < nk; kk++)
                            if ((fl >> kk) & 1u) flush_angle(kk, __shfl(set0, kk) + s - 1);
                        __syncthreads();
                    }
                    const int q0 = (int)(((long long)total * wave) / NW), q1 = (int)(((long long)total * (wave + 1)) / NW);
                    if (q0 < q1) {
                        uint32_t acc[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) acc[j] = 0u;
#pragma unroll
                        for (int kk = 0; kk < YM_PL_MAX_NK; kk++) {
                            if (kk < nk) {
                                const int kpre = __builtin_amdgcn_readlane(pre, kk), kn = __builtin_amdgcn_readlane(nq4, kk);
                                const int a0 = max(q0, kpre), a1 = min(q1, kpre + kn);
                                if (a0 < a1) { // (wave-uniform)
                                    const int first = ((__builtin_amdgcn_readlane(lo, kk) - base) >> 2) + (a0 - kpre), n = a1 - a0;
                                    // (two plain loads under a wave-uniform branch: written as one conditional expression hipcc selects
                                    //  between the LDS and the global POINTER and dies in its back end)
                                    auto quad = [&](int qi) {
                                        uint2 q;
                                        if (__builtin_amdgcn_readfirstlane(qi) < YM_PL_MAXE / 4) q = reinterpret_cast<const uint2 *>(elist)[qi]; // (elist by name: its address space stays known)
                                        else q = entries4[(base >> 2) + qi];
                                        return q;
                                    };
                                    uint2 e = quad(first);
                                    for (int c = 0; c < n; c++) { // (the next quad's entries are read while this one is gathered)
                                        const uint2 en = quad(first + min(c + 1, n - 1));
                                        rg_gather4(acc, lane_off, e);
                                        e = en;
                                    }
                                    deposit(acc, kk);
                                }
                            }
                        }
                    }
                }
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
        __syncthreads();
        if (nk > 0 && !listed) {
            // hypothesis cells are not an exact lattice (possible only through fp rounding), or the query's lists did not fit:
            // per-cell path over the window, wave w takes the angles k_lo + w, k_lo + w + NW, ...; sets of FLUSH beams as the
            // region correlate writes them
            const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
            const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
            const int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
            const int32_t *cy = cx + a.dim_stride;
            const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
            const int nq = st.nq;
            for (int kk = wave; kk < nk; kk += NW) {
                const double2 cs = a.ctrig[(size_t)b * a.nt_stride + k_lo + kk];
                for (int g = 0; g < ng; g++) {
                    uint32_t acc[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[j] = 0u;
                    if (job)
                        for (int j = 0; j < YM_RG_G; j++) {
                            const int ix = half * YM_RG_G + j;
                            if (ix >= nx) break;
                            const int base = cy[row] * lin_pitch(a.g) + cx[ix];
                            unsigned sum = 0;
                            const int i1 = min(nq, (g + 1) * YM_RG_FLUSH);
                            for (int i = g * YM_RG_FLUSH; i < i1; i++)
                                sum += cell_value(a.g, grid, limit, (unsigned)(base + lookup_offset(ql[i], cs.x, cs.y, st.off_x, st.off_y, a.g.scale, lin_pitch(a.g))));
                            acc[2 * (j >> 2) + (j & 1)] += sum << (16 * ((j >> 1) & 1));
                        }
                    store_partial16(partial + (((size_t)g * nt + (k_lo + kk)) * 64 + lane) * 16, acc);
                }
            }
        }
        __syncthreads();
    }
    // sets written so far per angle: (cum - 1) / FLUSH full ones have left LDS, the one being filled is there.  Without lists
    // (per-cell path) all ng sets are in memory and the LDS sums are zero.
    if (tid < nk) s_sets[tid] = regular ? (my_cum > 0 ? (my_cum - 1) / YM_RG_FLUSH : 0) : ng;
    __syncthreads();
    if (!a.fuse_score) {
        if (regular)
            for (int kk = 0; kk < nk; kk++) {
                const int f = s_sets[kk];
                flush_angle(kk, f);
                for (int f2 = f + 1; f2 < ng; f2++) // the sets this angle never reached: zeros
                    for (int i = tid; i < 8 * 64; i += NT)
                        reinterpret_cast<uint32_t *>(partial + (((size_t)f2 * nt + (k_lo + kk)) * 64 + (i & 63)) * 16)[i >> 6] = 0u;
            }
        return;
    }
    // ---- score (score_kernel's arithmetic, statement for statement), as in correlate_region_kernel
    unsigned long long *pmax = reinterpret_cast<unsigned long long *>(region); // [ny * nx] fp64 bit patterns, >= 0
    const int nxy = nx * ny;
    double *dpen = reinterpret_cast<double *>(region) + ((nxy + 1) & ~1);
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
        for (int kk = wave; kk < nk; kk += NW) {
            const int k = k_lo + kk;
            unsigned tot[YM_RG_G];
#pragma unroll
            for (int j = 0; j < YM_RG_G; j++) tot[j] = (sum16[kk][2 * (j >> 2) + (j & 1)][lane] >> (16 * ((j >> 1) & 1))) & 0xffffu;
            for (int f = 0; f < min(s_sets[kk], ng); f++) {
                const uint16_t *pp = partial + (((size_t)f * nt + k) * 64 + lane) * 16;
#pragma unroll
                for (int j = 0; j < YM_RG_G; j++) tot[j] += pp[j];
            }
            const double angle = (ct - a.lat.angle_off) + k * a.lat.angle_res;
            double bmax0 = -1.0, bmax1 = -1.0;
#pragma unroll
            for (int j = 0; j < YM_RG_G; j++) {
                const int ix = half * YM_RG_G + j;
                if (job && ix < nx) {
                    const int c = row * nx + ix;
                    const double r = hyp_response_dp(a.g, a.lat.penalize, tot[j], nq, dpen[c], angle, ct);
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

#endif // YM_EXPERIMENTAL

} // namespace ym
