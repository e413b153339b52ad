// ym_k_region2.hpp -- K4r2 (round 5): the region correlate of LARGE batches with SEVERAL WAVES PER ANGLE.
// Part of ym_kernels.hpp (include that, not this file).
//
// correlate_region_kernel (ym_k_region.hpp) gives every coarse angle of an angle block one wave; a round of its region walk
// (stage - barrier - gather - barrier) lasts as long as the wave whose angle has the most patches in the region gathers: on the
// bench query the sum of those maxima is 4222 patches per item where an even deal would give 2838 (profiles/r04_region_study.md),
// and the third block of an item (angles 16 .. 20) idles three of its eight waves.  Two forms with an even deal were built in
// round 4 (one block per item; two blocks per item with the sums in LDS): both paid for the deal with occupancy or LDS
// atomics and lost.  This form keeps everything of the first form -- lists, boxes, staging from the row-major window, the
// gather loop, sums in registers, the wave that scores -- and changes ONE thing: a block has SIXTEEN waves for its (up to)
// eight angles, and the floor(16 / angles) waves of an angle split the angle's patches of a region between them in quads:
//   * the longest gather of a round is half (a third, in the five-angle block) of what it was;
//   * the staging of a region is spread over 1024 threads: three or four 16-byte tasks per thread instead of six;
//   * two blocks = 32 waves per CU instead of three blocks = 24, and the 80 KB a block may use hold regions H class rows high
//     (template parameter; 64 x 80 before), so that an item's walk has fewer rounds;
//   * a wave's sums leave as its OWN sets of 16-bit partials (slice s of an angle writes sets s * ng ...), and when the block
//     scores, the slices of an angle are added through LDS (the region buffer is free by then).
#pragma once

namespace ym {

#define YM_R2_NW 16      // waves per block
#define YM_R2_MAX_WPA 3  // at most this many waves share an angle (sets of partial sums are kept per slice)

template <int H>
struct R2Geom {
    static constexpr int ROWS = H + 26;                  // staged rows per class image: the region + the patch height
    static constexpr int CLS = YM_RG_PITCH * ROWS;       // bytes per class image
    static constexpr int ZERO = 4 * CLS;                 // LDS offset of the all-zero patch the padding entries point at
    static constexpr int LDS_BYTES = ZERO + 26 * YM_RG_PITCH + 32;
    static_assert(H <= 255 && ZERO + 3 < 65536, "boxes hold class rows in a byte, entries are 16-bit LDS offsets");
};
// bytes the host keeps past the last item's row-major window (what the staging loop reads of the last item, see YM_RG_WINDOW_SLACK)
#define YM_R2_WINDOW_SLACK(pitch, h) ((size_t)(2 * ((h) + 26) + 2 * (h) + 2) * (size_t)(pitch) + 512)
#define YM_R2_COMB_OFFSET 16384 // where the slices' sums meet in LDS when the block scores (past the per-cell maxima and penalties)

#ifdef YM_EXPERIMENTAL // measured on 4096 cfg2 items (round 5, same run): 2986 us with regions of 100 rows, 3109 with 80, 3249 with 128, against
                       // 2730 for correlate_region_kernel<8, true> -- bit-exact, slower: kept as the study's fourth form, option 32 = 5
// grid (P, B): block (p, item) = 16 waves for the angles [p * a.nw, min(nt, (p + 1) * a.nw)), a.nw <= 8 (the host: 8).
// Wave w: angle index w % nk, slice w / nk of that angle's patches (valid while slice < wpa = min(16 / nk, YM_R2_MAX_WPA)).
// Lane = 13 x-adjacent hypotheses of one lattice row, as in correlate_region_kernel.  Dynamic LDS = R2Geom<H>::LDS_BYTES.
template <int H>
__global__ __launch_bounds__(64 * YM_R2_NW, 8 /* two blocks = 32 waves per CU: 64 VGPRs */) void correlate_region2_kernel(RegionArgs a) {
    using G = R2Geom<H>;
    constexpr int NW = YM_R2_NW, NT = 64 * NW;
    constexpr int TPC = NT / 2;                        // staging threads per row parity
    constexpr int NSEG = 2 * YM_RG_SEGS;               // 16-byte segments of a staged window row
    constexpr int LPS = TPC / NSEG;                    // rows the threads of a row parity cover at once
    constexpr int RSTEP = LPS;
    constexpr int PER = (G::ROWS + LPS - 1) / LPS;     // copy tasks per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char region[]; // four class images + the zero patch
    __shared__ int rlist[YM_RG_MAX_REGIONS];
    __shared__ uint32_t rboxl[YM_RG_MAX_REGIONS];
    __shared__ unsigned short seginfo[NW][YM_RG_MAX_REGIONS][2]; // per wave and listed region: its first entry and the end
    __shared__ uint2 elist[NW][64];                    // per wave: its first 256 entries of the region being gathered
    __shared__ int rcount;
    __shared__ int s_flushed[NW];
    int p;
    const int b = xcd_item_of_block_2d(p);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const YmItemState &st = a.states[b];
    const int nt = a.lat.nt, nx = a.lat.nx, ny = a.lat.ny, ng = a.ng;
    const int k_lo = p * a.nw, k_hi = min(nt, k_lo + a.nw), nk = k_hi - k_lo; // (nk >= 1: the host launches ceil(nt / nw) blocks per item)
    const int wpa = min(NW / nk, YM_R2_MAX_WPA);
    const int ai = wave % nk, slice = wave / nk;
    const int k = k_lo + ai;
    const bool kvalid = slice < wpa;
    const int row = lane & 31, half = lane >> 5;
    const bool job = row < ny && half * YM_RG_G < nx;
    const int32_t *__restrict__ starts = a.starts + (size_t)st.qslot * a.starts_stride;
    const uint16_t *__restrict__ entries = a.entries + (size_t)st.qslot * a.entries_stride;
    const uint32_t lds0 = (uint32_t)(size_t)region;
    const uint32_t lane_off = lds0 + (uint32_t)((job ? row : 0) * YM_RG_PITCH + (half * YM_RG_G < nx ? half * YM_RG_G : 0));
    uint16_t *partial = a.partial + (size_t)b * a.partial_stride;
    uint32_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0u;
    int in_set = 0, flushed = 0;
    auto flush = [&]() { // the wave's set number `flushed`: set slice * ng + flushed of angle k
        if (flushed < ng) store_partial16(partial + (((size_t)(slice * ng + flushed) * nt + k) * 64 + lane) * 16, acc);
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0u;
        flushed++;
        in_set = 0;
    };
    YM_STAMP(a, 8);
    const bool regular = st.regular[0] && a.force_irregular != 1 && starts[a.nbins] >= 0;
    if (regular) {
        const int nreg = a.nrx * a.nry;
        for (int i = tid; i < (G::LDS_BYTES - G::ZERO) / 4; i += NT) reinterpret_cast<uint32_t *>(region + G::ZERO)[i] = 0u;
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
                const uint32_t r0 = v & 0xffu, r1 = min((uint32_t)(G::ROWS - 1), ((v >> 8) & 0xffu) + (uint32_t)ny - 1u);
                const uint32_t s0 = ((v >> 16) & 0xfcu) >> 4, s1 = min((uint32_t)(YM_RG_SEGS - 1), ((v >> 24) + reach) >> 4);
                rboxl[i] = r0 | r1 << 8 | s0 << 16 | s1 << 24;
            }
        }
        if (kvalid)
            for (int i = lane; i < nlist; i += 64) { // this wave's share of its angle's entries of region i: quads [q * slice / wpa, q * (slice + 1) / wpa)
                const int32_t *srow = starts + (size_t)rlist[i] * nt + k;
                const int t0 = srow[0], q = (srow[1] - t0) >> 2;
                seginfo[wave][i][0] = (unsigned short)(t0 + 4 * ((q * slice) / wpa));
                seginfo[wave][i][1] = (unsigned short)(t0 + 4 * ((q * (slice + 1)) / wpa));
            }
        __syncthreads();
        // copy tasks, as in correlate_region_kernel<8, true>: a thread owns one 16-byte segment of a WINDOW row (eight class bytes of
        // each column parity) of the rows r0, r0 + RSTEP, ... of one row parity
        const uint32_t cls = 2u * ((uint32_t)tid / TPC), j0 = (uint32_t)tid % TPC;
        const uint32_t seg = j0 % NSEG, r0 = j0 / NSEG;
        const bool copier = j0 < (uint32_t)(LPS * NSEG);
        const uint32_t src0 = (2u * r0 + (cls >> 1)) * (uint32_t)a.g.pitch + 16u * seg;
        const uint32_t src_step = 2u * RSTEP * (uint32_t)a.g.pitch;
        const uint32_t dst0 = (cls * G::ROWS + r0) * YM_RG_PITCH + 8u * seg;
        uint4 v[PER];
        auto band_in = [&](int q, uint32_t bx) {
            return (uint32_t)(q * RSTEP) <= ((bx >> 8) & 0xffu) && (uint32_t)(q * RSTEP + RSTEP - 1) >= (bx & 0xffu);
        };
        const uint8_t *__restrict__ window = a.grid + (size_t)b * a.grid_stride;
        auto seg_inside = [&](uint32_t bx) { return copier && seg >= 2u * ((bx >> 16) & 0xffu) && seg <= 2u * (bx >> 24) + 1u; };
        auto stage_load = [&](int R, uint32_t bx) {
            const int RX = R % a.nrx, RY = R / a.nrx;
            const uint8_t *src = window + ((size_t)(2 * RY * H) * a.g.pitch + (size_t)RX * (2 * YM_RG_W)); // (wave-uniform)
            const bool seg_in = seg_inside(bx);
#pragma unroll
            for (int q = 0; q < PER; q++)
                if (band_in(q, bx)) v[q] = *reinterpret_cast<const uint4 *>(src + (seg_in ? src0 + (uint32_t)q * src_step : 0u));
        };
        auto stage_store = [&](uint32_t bx) {
            const bool seg_in = seg_inside(bx);
#pragma unroll
            for (int q = 0; q < PER; q++) {
                uint32_t *d = reinterpret_cast<uint32_t *>(region + dst0 + (uint32_t)(q * RSTEP * YM_RG_PITCH));
                if (band_in(q, bx) && seg_in && r0 + (uint32_t)(q * RSTEP) < (uint32_t)G::ROWS) { // (the last band may run past the image)
                    d[0] = __builtin_amdgcn_perm(v[q].y, v[q].x, 0x06040200u); d[1] = __builtin_amdgcn_perm(v[q].w, v[q].z, 0x06040200u);
                    d[G::CLS / 4] = __builtin_amdgcn_perm(v[q].y, v[q].x, 0x07050301u); d[G::CLS / 4 + 1] = __builtin_amdgcn_perm(v[q].w, v[q].z, 0x07050301u);
                }
            }
        };
        auto segment = [&](int ri, int &t0, int &t2) {
            t0 = t2 = 0;
            if (kvalid) {
                t0 = __builtin_amdgcn_readfirstlane((int)seginfo[wave][ri][0]);
                t2 = __builtin_amdgcn_readfirstlane((int)seginfo[wave][ri][1]);
            }
        };
        const uint2 *__restrict__ entries4 = reinterpret_cast<const uint2 *>(entries); // four entries per element
        uint2 ev = make_uint2(0u, 0u);
        auto entries_load = [&](int t0, int t2) {
            ev = make_uint2(0u, 0u);
            if (t0 + 4 * lane < t2) ev = entries4[(t0 >> 2) + lane];
        };
        auto gather = [&](int lo, int hi) { // entries [lo, hi)
            const int lds_hi = min(hi, lo + 256);
            if (lo < lds_hi) {
                const uint2 *el = elist[wave];
                const int n4 = (lds_hi - lo) >> 2;
                uint2 e0 = el[0], e1 = el[min(1, n4 - 1)];
                int c = 0;
                for (; c + 1 < n4; c += 2) {
                    rg_gather4(acc, lane_off, e0);
                    e0 = el[min(c + 2, n4 - 1)];
                    in_set += 4;
                    if (in_set == YM_RG_FLUSH) flush();
                    rg_gather4(acc, lane_off, e1);
                    e1 = el[min(c + 3, n4 - 1)];
                    in_set += 4;
                    if (in_set == YM_RG_FLUSH) flush();
                }
                if (c < n4) {
                    rg_gather4(acc, lane_off, e0);
                    in_set += 4;
                    if (in_set == YM_RG_FLUSH) flush();
                }
            }
            for (int c = lds_hi; c < hi; c += 4) { // (a very long segment)
                rg_gather4(acc, lane_off, entries4[c >> 2]);
                in_set += 4;
                if (in_set == YM_RG_FLUSH) flush();
            }
        };
        int s0 = 0, s2 = 0;
        if (nlist > 0) {
            const uint32_t bx = __builtin_amdgcn_readfirstlane(rboxl[0]);
            segment(0, s0, s2);
            entries_load(s0, s2);
            stage_load(rlist[0], bx);
            stage_store(bx);
            elist[wave][lane] = ev;
        }
        __syncthreads();
        for (int ri = 0; ri < nlist; ri++) {
            // two-stage pipeline: the global loads of the next region are in flight (registers) while this one is gathered
            const bool has_next = ri + 1 < nlist;
            int n0 = 0, n2 = 0;
            uint32_t nbx = 0u;
            if (has_next) {
                nbx = __builtin_amdgcn_readfirstlane(rboxl[ri + 1]);
                segment(ri + 1, n0, n2);
                entries_load(n0, n2);
                stage_load(rlist[ri + 1], nbx);
            }
            gather(s0, s2);
            __syncthreads(); // every wave is done with region ri
            if (has_next) {
                stage_store(nbx);
                elist[wave][lane] = ev;
            }
            s0 = n0; s2 = n2;
            __syncthreads();
        }
    } else if (kvalid && job && slice == 0) {
        // hypothesis cells are not an exact lattice (possible only through fp rounding), or the query's lists did not fit: per-cell
        // path over the window by the angle's first wave, set g = the beams [g * FLUSH, (g + 1) * FLUSH)
        const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
        const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
        const int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
        const int32_t *cy = cx + a.dim_stride;
        const double2 cs = a.ctrig[(size_t)b * a.nt_stride + k];
        const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
        const int nq = st.nq;
        for (int g = 0; g < ng; g++) {
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
            flush();
        }
    }
    YM_STAMP(a, 9);
    if (!a.fuse_score) {
        // every set of every slice exists in memory for score_kernel: the set being filled, then empty ones; the first wave of an
        // angle also writes the (empty) sets of the slices this block has no wave for
        if (!kvalid) return;
        while (flushed < ng) flush();
        if (slice == 0)
            for (int s = wpa; s < YM_R2_MAX_WPA; s++)
                for (int f = 0; f < ng; f++) store_partial16(partial + (((size_t)(s * ng + f) * nt + k) * 64 + lane) * 16, acc); // (acc is zero after a flush)
        return;
    }
    // ---- score (score_kernel's arithmetic, statement for statement), as in correlate_region_kernel; first the slices of an angle
    // meet in LDS: slice s > 0 leaves its registers (seven dwords per lane) and its number of written sets there
    unsigned long long *pmax = reinterpret_cast<unsigned long long *>(region); // [ny * nx] fp64 bit patterns, >= 0
    const int nxy = nx * ny;
    double *dpen = reinterpret_cast<double *>(region) + ((nxy + 1) & ~1);     // [ny * nx] distance penalty of every cell
    uint32_t *comb = reinterpret_cast<uint32_t *>(region + YM_R2_COMB_OFFSET); // [MAX_WPA - 1][8 angles][7][64]
    __syncthreads(); // every wave has left the region walk
    for (int i = tid; i < nxy; i += NT) {
        pmax[i] = 0ull;
        const int iy = i / nx, ix = i - iy * nx;
        const double x = -a.lat.off_x + ix * a.lat.step_x, y = -a.lat.off_y + iy * a.lat.step_y;
        dpen[i] = dist_penalty(a.g, x * x + y * y);
    }
    if (kvalid) {
        if (lane == 0) s_flushed[wave] = min(flushed, ng);
        if (slice > 0) {
#pragma unroll
            for (int j = 0; j < 7; j++) comb[(((slice - 1) * 8 + ai) * 7 + j) * 64 + lane] = acc[j];
        }
    }
    __syncthreads();
    if (kvalid && slice == 0) {
        unsigned tot[YM_RG_G];
#pragma unroll
        for (int j = 0; j < YM_RG_G; j++) tot[j] = (acc[2 * (j >> 2) + (j & 1)] >> (16 * ((j >> 1) & 1))) & 0xffffu;
        for (int s = 1; s < wpa; s++) {
#pragma unroll
            for (int j = 0; j < YM_RG_G; j++)
                tot[j] += (comb[(((s - 1) * 8 + ai) * 7 + 2 * (j >> 2) + (j & 1)) * 64 + lane] >> (16 * ((j >> 1) & 1))) & 0xffffu;
        }
        for (int s = 0; s < wpa; s++) // the sets the slices wrote out earlier (a slice of more than YM_RG_FLUSH patches: long queries only)
            for (int f = 0; f < s_flushed[s * nk + ai]; f++) {
                const uint16_t *pp = partial + (((size_t)(s * ng + f) * nt + k) * 64 + lane) * 16;
#pragma unroll
                for (int j = 0; j < YM_RG_G; j++) tot[j] += pp[j];
            }
        const double ct = st.center[2];
        const int nq = st.nq;
        const double angle = (ct - a.lat.angle_off) + k * a.lat.angle_res;
        const int ncb = (nxy + YM_SCORE_THREADS - 1) / YM_SCORE_THREADS;
        double bmax0 = -1.0, bmax1 = -1.0; // block maxima this lane contributes to (its 13 cells span at most 2 blocks)
        const int c0 = row * nx + half * YM_RG_G, cb0 = job ? c0 / YM_SCORE_THREADS : 0;
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
    __syncthreads();
    for (int i = tid; i < nxy; i += NT)
        if (pmax[i]) atomicMax(reinterpret_cast<unsigned long long *>(a.probs) + (size_t)b * a.probs_stride + i, pmax[i]);
}
#endif // YM_EXPERIMENTAL

} // namespace ym
