// ym_k_region_ws.hpp -- experimental: the round-5 list builder (one block per query, all angles) and the wave-specialised region correlate
// (round 4).  Both measured slower than what the product library runs (bin_kernel per angle block, correlate_region_kernel):
// profiles/r04_region_study.md, r05_region_study.md.  Part of ym_exp_forms.hpp; compiled only with -DYM_EXPERIMENTAL (`make experimental`).
#pragma once

namespace ym {

// ---- the round-5 form: ONE block per query slot sorts the pairs of all angles (lparts = 1, lnw = nt).  What the experimental correlate
// forms read; the product library builds its lists per angle block (bin_kernel above).
// grid (Q): one block per query slot of the call.  YM_BIN_THREADS threads.  Counting sort in LDS: count, scan, place (the order inside a bin is arbitrary: the
// sums are integers).  GridIndexLookup::ComputeOffsets for every coarse angle happens here.
// Inside a bin the entries are sorted by their byte misalignment (entry & 3; class images and rows are multiples of 4
// bytes), every run of equal misalignment is padded to an even length and the bin to a multiple of four with entries
// that point at the all-zero patch: the gather then adds the raw dwords of a PAIR of patches before one byte funnel, and
// never meets a ragged group.  An item whose padded list would not fit -- the buffers, one angle's share the ng sets of
// 16-bit sums the gather may fill, or more regions with work than a correlate block can list -- gets starts[nbins] = -1 and is scored by the per-cell path of correlate_region_kernel.
// YAG: the lookup cells of the reference's Python matcher (items yag_lattice_kernel proved regular; ym_k_common.hpp, lookup_cell_sem) --
// a template parameter, so that the Karto instantiation, a whole launch of the metric workload, stays the code it was.
template <bool YAG = false>
__global__ __launch_bounds__(YM_BIN_THREADS, YM_BIN_MIN_WAVES) void bin_whole_kernel(RegionArgs a) {
    constexpr int MAXP = (YM_RG_MAX_ENTRIES + YM_BIN_THREADS - 1) / YM_BIN_THREADS; // pairs per thread
    // dynamic LDS (YM_BIN_LDS_BYTES: sized by the host so that two blocks share a CU on the usual lattice):
    extern __shared__ __attribute__((aligned(16))) unsigned char bin_smem[];
    int *wave_tot = reinterpret_cast<int *>(bin_smem);                                   // [YM_BIN_THREADS / 64]
    int *angle_tot = wave_tot + YM_BIN_THREADS / 64;                                     // [YM_MAX_COARSE_NT] padded entries per coarse angle
    unsigned *region_bits = reinterpret_cast<unsigned *>(angle_tot + YM_MAX_COARSE_NT);    // [YM_RG_MAX_BINS / 32] regions that hold a patch
    int *regions_used = reinterpret_cast<int *>(region_bits + YM_RG_MAX_BINS / 32);
    unsigned (*cnt)[2] = reinterpret_cast<unsigned (*)[2]>(regions_used + 1);             // [nbins] four 16-bit counters (one per
                                                                                         // misalignment), later the runs' first positions
    unsigned short *ent = reinterpret_cast<unsigned short *>(cnt + a.nbins);             // [entries_stride]
    const int nboxes = a.nregions * a.parts;
    unsigned (*box)[4] = reinterpret_cast<unsigned (*)[4]>(bin_smem + ((reinterpret_cast<unsigned char *>(ent + a.entries_stride) - bin_smem) + 15) / 16 * 16); // [nboxes] rmin, rmax, xmin, xmax
    const int qs = blockIdx.x, b = a.qrep[qs], tid = threadIdx.x, lane = tid & 63;
    const YmItemState &st = a.states[b];
    const int nq = st.nq, nt = a.lat.nt;
    const int total = nq * nt;
    const int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
    const int cx0 = cx[0], cy0 = cx[a.dim_stride];
    const double off_x = st.off_x, off_y = st.off_y;
    const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
    const double2 *trig = a.ctrig + (size_t)b * a.nt_stride;
    int32_t *starts = a.starts + (size_t)qs * a.starts_stride;
    // The query's sensor-frame points and the angle table go through LDS: pass 1 reads point i of angle k for `per` consecutive
    // pairs per thread -- from global memory that is one dependent load per pair, 64 cache lines per wave load (the lanes are
    // `per` points apart).  They borrow the entry list's room, which nothing writes before pass 1 is over (round 5: one query per
    // ITEM of a batch, ym_pairs_create, made this kernel a whole launch of 4096 blocks instead of one block).
    double2 *qls = reinterpret_cast<double2 *>(bin_smem + ((reinterpret_cast<unsigned char *>(ent) - bin_smem) + 15) / 16 * 16);
    const bool ql_in_lds = (size_t)(nq + nt) * sizeof(double2) + 16 <= (size_t)a.entries_stride * 2;
    double2 *trigs = qls + nq;
    if (ql_in_lds) {
        for (int i = tid; i < nq; i += YM_BIN_THREADS) qls[i] = ql[i];
        if (tid < nt) trigs[tid] = trig[tid];
    }
    for (int i = tid; i < a.nbins * 2; i += YM_BIN_THREADS) (&cnt[0][0])[i] = 0u;
    if (tid < YM_MAX_COARSE_NT) angle_tot[tid] = 0;
    if (tid < YM_RG_MAX_BINS / 32) region_bits[tid] = 0u;
    if (tid == 0) *regions_used = 0;
    for (int i = tid; i < nboxes; i += YM_BIN_THREADS) { box[i][0] = 255u; box[i][1] = 0u; box[i][2] = 255u; box[i][3] = 0u; }
    __syncthreads();
    // Pass 1: bin, entry and RANK inside (bin, misalignment) of every pair, kept in registers (the rank is what the
    // counting atomic returns), so that pass 2 neither recomputes the cells nor needs a second atomic.
    // (29 bits of bin and entry + the rank's low 3 bits in one register, its other 8 bits four to a register: 35 registers
    // for 28 pairs, so that two blocks share a CU; the host keeps max_n below 2048 on this path)
    unsigned key[MAXP];          // rank & 7 << 29 | bin << 16 | entry; 0xffffffff = no pair
    unsigned rank_hi[(MAXP + 3) / 4];
#pragma unroll
    for (int q = 0; q < (MAXP + 3) / 4; q++) rank_hi[q] = 0u;
    // A thread takes `per` CONSECUTIVE pairs (beam i of angle k, stepped from one pair to the next): neighbouring beams of one
    // angle mostly share their bin and nearly always their box, so the lanes of a wave -- `per` beams apart -- spread over
    // the counters (64 lanes' atomics on one LDS word take turns: 59 of this kernel's 85 us went there while lane = beam), and a thread
    // keeps the box of its run of pairs in registers and sends it when the box changes.
    const int per_thread = (total + YM_BIN_THREADS - 1) / YM_BIN_THREADS;
    const unsigned inv_nw = 65536u / (unsigned)a.nw + 1u; // (angle block k / nw as a multiplication: exact for k < 256, nw <= 256)
    const int p0 = tid * per_thread;
    int k = p0 / max(nq, 1), i = p0 - k * nq;
    int cur = -1;                                 // the box the thread is collecting, and its extent so far
    unsigned r0 = 255u, r1 = 0u, x0 = 255u, x1 = 0u;
    auto send_box = [&]() {
        if (cur >= 0) {
            unsigned *bx = box[cur];
            atomicMin(&bx[0], r0); atomicMax(&bx[1], r1); atomicMin(&bx[2], x0); atomicMax(&bx[3], x1);
        }
    };
    // (the same loop over the points in LDS and, for a query whose points do not fit into the borrowed room, in global memory: written
    //  once with the source as a parameter -- a select between an LDS and a global POINTER makes every load a flat load)
    auto pass1 = [&](auto point_of, auto trig_of) {
#pragma unroll
        for (int q = 0; q < MAXP; q++) {
            const int p = p0 + q;
            key[q] = 0xffffffffu;
            if (q < per_thread && p < total) {
                const double2 cs = trig_of(k);
                const double2 pt = point_of(i);
                int bin, region; unsigned e, er, ex;
                const int2 lc = YAG ? lookup_cell_sem(a.g, pt, cs.x, cs.y, off_x, off_y, st.ylat[0], st.ylat[1]) : lookup_cell(pt, cs.x, cs.y, off_x, off_y, a.g.scale);
                if (region_entry(a, lc, cx0, cy0, k, bin, e, region, er, ex)) {
                    const unsigned rank = (atomicAdd(&cnt[bin][(e >> 1) & 1u], 1u << (16 * (e & 1u))) >> (16 * (e & 1u))) & 0xffffu;
                    key[q] = (rank & 7u) << 29 | (unsigned)bin << 16 | e;
                    rank_hi[q >> 2] |= ((rank >> 3) & 0xffu) << (8 * (q & 3));
                    const int bi = (int)(__umul24((unsigned)region, (unsigned)a.parts) + (__umul24((unsigned)k, inv_nw) >> 16));
                    if (bi != cur) { send_box(); cur = bi; r0 = r1 = er; x0 = x1 = ex; }
                    else { r0 = min(r0, er); r1 = max(r1, er); x0 = min(x0, ex); x1 = max(x1, ex); }
                }
            }
            if (++i >= nq) { i = 0; k++; }
        }
    };
    if (ql_in_lds) pass1([&](int i_) { return qls[i_]; }, [&](int k_) { return trigs[k_]; });
    else pass1([&](int i_) { return ql[i_]; }, [&](int k_) { return trig[k_]; });
    send_box();
    __syncthreads();
    {
        uint32_t *rb = a.rbox + (size_t)qs * a.rbox_stride;
        for (int i = tid; i < nboxes; i += YM_BIN_THREADS) rb[i] = box[i][0] | box[i][1] << 8 | box[i][2] << 16 | box[i][3] << 24;
    }
    // exclusive scan of the padded bin sizes: thread t owns the bins [t * per, (t + 1) * per)
    const int per = (a.nbins + YM_BIN_THREADS - 1) / YM_BIN_THREADS;
    const int first = tid * per;
    int padded_total;
    {
        int local = 0;
        for (int j = 0; j < per; j++)
            if (first + j < a.nbins) {
                const unsigned c01 = cnt[first + j][0], c23 = cnt[first + j][1];
                const int c[4] = {(int)(c01 & 0xffffu), (int)(c01 >> 16), (int)(c23 & 0xffffu), (int)(c23 >> 16)};
                const int padded = (((c[0] + 1) & ~1) + ((c[1] + 1) & ~1) + ((c[2] + 1) & ~1) + ((c[3] + 1) & ~1) + 3) & ~3;
                local += padded;
                if (padded) {
                    atomicAdd(&angle_tot[(first + j) % nt], padded);
                    const int R = (first + j) / nt;
                    atomicOr(&region_bits[R >> 5], 1u << (R & 31));
                }
            }
        int incl = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        if (lane == 63) wave_tot[tid >> 6] = incl;
        __syncthreads();
        if (tid < YM_RG_MAX_BINS / 32 && region_bits[tid]) atomicAdd(regions_used, __popc(region_bits[tid]));
        __syncthreads();
        int base = 0, all = 0;
        for (int w = 0; w < YM_BIN_THREADS / 64; w++) {
            if (w < (tid >> 6)) base += wave_tot[w];
            all += wave_tot[w];
        }
        padded_total = all;
        bool fits = all <= YM_RG_MAX_ENTRIES && all <= (int)a.entries_stride && a.force_irregular != 2;
        for (int k = 0; k < nt; k++) fits = fits && angle_tot[k] <= a.ng * YM_RG_FLUSH;
        fits = fits && *regions_used <= YM_RG_MAX_REGIONS; // (what a block of correlate_region_kernel can list)
        int run = base + incl - local;
        for (int j = 0; j < per; j++)
            if (first + j < a.nbins) {
                const unsigned c01 = cnt[first + j][0], c23 = cnt[first + j][1];
                const int c[4] = {(int)(c01 & 0xffffu), (int)(c01 >> 16), (int)(c23 & 0xffffu), (int)(c23 >> 16)};
                starts[first + j] = run;
                int pos = run;
                unsigned fill[4];
                for (int r = 0; r < 4; r++) { // run r: its entries from pos on (placed below), then the padding
                    fill[r] = (unsigned)pos;
                    if (fits && (c[r] & 1)) ent[pos + c[r]] = (unsigned short)(a.rg_zero + r);
                    pos += (c[r] + 1) & ~1;
                }
                if (fits && ((pos - run) & 3)) { ent[pos] = (unsigned short)a.rg_zero; ent[pos + 1] = (unsigned short)a.rg_zero; }
                pos = run + ((pos - run + 3) & ~3);
                cnt[first + j][0] = fill[0] | fill[1] << 16;
                cnt[first + j][1] = fill[2] | fill[3] << 16;
                run = pos;
            }
        if (tid == 0) { starts[a.nbins] = fits ? all : -1; if (a.stamps && qs == 0) a.stamps[26] = (unsigned long long)all; }
        if (!fits) return; // (block-uniform)
    }
    __syncthreads();
    // Pass 2: every pair to its run's first position + its rank
#pragma unroll
    for (int q = 0; q < MAXP; q++) {
        if (key[q] == 0xffffffffu) continue;
        const unsigned bin = (key[q] >> 16) & 0x1fffu, e = key[q] & 0xffffu;
        const unsigned rank = key[q] >> 29 | ((rank_hi[q >> 2] >> (8 * (q & 3))) & 0xffu) << 3;
        const unsigned f = cnt[bin][(e >> 1) & 1u];
        ent[((e & 1u) ? f >> 16 : f & 0xffffu) + rank] = (unsigned short)e;
    }
    __syncthreads();
    uint32_t *out = reinterpret_cast<uint32_t *>(a.entries + (size_t)qs * a.entries_stride);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(ent);
    for (int i = tid; i < (padded_total + 1) / 2; i += YM_BIN_THREADS) out[i] = src[i];
}


// ================================================================== the wave-specialised form (round 4)
// Phase clocks of correlate_region_kernel (YM_RG_PROF, scripts/dev/region_phases.py, 4096 items): a wave gathers for a
// quarter of its life.  The rest: 37 % in the issue of the next region's staging loads -- every global_load_dwordx4 of the
// burst a block fires after its barrier queues behind the others at the CU's one address unit, ~600 clocks each --, 21 % at
// the barrier after the gather, 15 % in the staging stores and the barrier after them, the block's set-up and scoring.
// On average 1.5 of a SIMD's six waves are gathering, and the gather loop alone (scripts/exp/rg_proto.hip) reaches its
// issue rate only from two per SIMD on.  Here the two jobs belong to different waves, and a block does not end with its item:
//   waves 0 .. NG - 1      GATHER: wave w owns coarse angle p * NG + w of the block's current item, walks the item's regions,
//                          waits until a region's buffer is FULL, gathers its patches, says DONE, scores its angle, goes on
//                          to the next item.  No staging, no barrier.
//   waves NG .. NG + 7     LOAD: two groups of four (one wave per class image); group g fills buffer g, i.e. every other
//                          region of the stream: loads in registers, waits until every gatherer is DONE with the buffer's
//                          previous region, stores, says FULL.  They are the ones that wait for memory, and they run ahead
//                          into the next item while the gatherers finish this one: the pipeline never drains.
// Two region buffers per block, counters in LDS instead of barriers (full[b] / done[b] only grow; a region's number in the
// block's stream says what to wait for).  Two blocks of 16 waves per CU, each with its own sequence of (item, angle block)
// units: the blocks that share an item run on one XCD at the same time (the item's planes in one L2).  What a unit's walk
// needs -- the regions with work, their boxes, every angle's segment of the entry list -- depends on the QUERY alone and
// comes from region_walk_kernel, once per query of the call.  Regions own YM_WS_H class rows (two buffers in 80 KB).
// (Tried and dropped: regions of ONE class image, 256 x 74 class bytes at a pitch of 324 -- rows of 288 contiguous bytes for the
//  loaders, less margin per owned byte.  A wall's bounding box fills such a region, the walk has 64 rounds of 16 patches per
//  wave instead of 27 of 43, and a round costs a wave one memory round trip whatever it gathers: 4140 us against 3281.
//  One loader group with two register sets per wave: a destination register the compiler also uses as a temporary makes it
//  wait for every load in flight.)
#define YM_WS_H 56
#define YM_WS_ROWS (YM_WS_H + 26)
#define YM_WS_CLS (YM_RG_PITCH * YM_WS_ROWS)
#define YM_WS_ZERO (4 * YM_WS_CLS)                      // every buffer is followed by its own all-zero patch
#define YM_WS_ZERO_BYTES (25 * YM_RG_PITCH + 32)        // (the last lane row reads 16 bytes from byte 13 on of patch row 25)
#define YM_WS_BUF ((YM_WS_ZERO + YM_WS_ZERO_BYTES + 15) / 16 * 16)
#define YM_WS_NG 8                                      // gather waves (= angles) per block
#define YM_WS_NL 8                                      // loader waves: two groups of one per class image; group g fills buffer g
#define YM_WS_PLANES_SLACK(half_pitch) ((size_t)(2 * YM_WS_ROWS + 2 * YM_WS_H + 22) * (size_t)(half_pitch) + 256)
// a unit's walk (region_walk_kernel): [0] regions with work, [16 + i] region, [16 + MAX + i] box, [16 + 2 MAX + w MAX + i] first
// entry | end << 16 of gather wave w
#define YM_WS_WALK_WORDS (16 + (2 + YM_WS_NG) * YM_RG_MAX_REGIONS)
static_assert(YM_WS_ZERO + YM_WS_ZERO_BYTES < 65536, "entries are 16-bit LDS offsets");

#ifdef YM_EXPERIMENTAL // (measured slower than correlate_region_kernel: compiled only into builds made with -DYM_EXPERIMENTAL, debug option 32 = 2)
// grid (parts, Q), one wave: the walk of angle block p of query slot q
__global__ __launch_bounds__(64) void region_walk_kernel(RegionArgs a) {
    __shared__ int rl[YM_RG_MAX_REGIONS];
    const int p = blockIdx.x, qs = blockIdx.y, lane = threadIdx.x;
    const int nt = a.lat.nt, nx = a.lat.nx, ny = a.lat.ny;
    const int32_t *__restrict__ starts = a.starts + (size_t)qs * a.starts_stride;
    uint32_t *wk = a.walk + ((size_t)qs * a.parts + p) * YM_WS_WALK_WORDS;
    if (starts[a.nbins] < 0) { if (lane == 0) wk[0] = 0u; return; } // (no list: the per-cell path)
    const int k_lo = p * a.nw, k_hi = min(nt, k_lo + a.nw);
    const int nreg = a.nregions;
    int n = 0;
    for (int R0 = 0; R0 < nreg; R0 += 64) {
        const int R = R0 + lane;
        const bool has = R < nreg && starts[(size_t)R * nt + k_lo] != starts[(size_t)R * nt + k_hi];
        const unsigned long long mask = __ballot(has);
        const int at = n + __popcll(mask & ((1ull << lane) - 1ull));
        if (has && at < YM_RG_MAX_REGIONS) rl[at] = R;
        n += __popcll(mask);
    }
    n = min(n, YM_RG_MAX_REGIONS); // (bin_kernel refuses a list with more regions in use)
    __syncthreads();
    if (lane == 0) wk[0] = (uint32_t)n;
    const uint32_t *rb = a.rbox + (size_t)qs * a.rbox_stride;
    const uint32_t reach = (uint32_t)(nx > YM_RG_G ? 15 + YM_RG_G : 15);
    for (int i = lane; i < n; i += 64) {
        const int R = rl[i];
        wk[16 + i] = (uint32_t)R;
        // what the block's patches read of the region: rows rmin .. rmax + ny - 1, bytes (xmin & ~3) .. xmax + 15 (+ 13 for the
        // second half of a lattice row)
        const uint32_t v = rb[(size_t)R * a.parts + p];
        const uint32_t r0 = v & 0xffu, r1 = min((uint32_t)(YM_WS_ROWS - 1), ((v >> 8) & 0xffu) + (uint32_t)ny - 1u);
        const uint32_t s0 = ((v >> 16) & 0xfcu) >> 4, s1 = min((uint32_t)(YM_RG_SEGS - 1), ((v >> 24) + reach) >> 4);
        wk[16 + YM_RG_MAX_REGIONS + i] = r0 | r1 << 8 | s0 << 16 | s1 << 24;
        for (int w = 0; w < a.nw; w++) {
            const int k = k_lo + w;
            wk[16 + (2 + w) * YM_RG_MAX_REGIONS + i] = k < nt ? ((uint32_t)starts[(size_t)R * nt + k] & 0xffffu) | (uint32_t)starts[(size_t)R * nt + k + 1] << 16 : 0u;
        }
    }
}

// grid (parts, B), YM_WS_NG waves: the items the wave-specialised kernel leaves out -- hypothesis cells that are not an exact
// lattice (possible only through fp rounding), or a query whose lists did not fit -- scored cell by cell over the window.  A
// block of any other item returns at once.
__global__ __launch_bounds__(64 * YM_WS_NG) void region_percell_kernel(RegionArgs a) {
    const int p = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const YmItemState &st = a.states[b];
    const int32_t *starts = a.starts + (size_t)st.qslot * a.starts_stride;
    if (st.regular[0] && a.force_irregular != 1 && starts[a.nbins] >= 0) return;
    const int nt = a.lat.nt, nx = a.lat.nx, ny = a.lat.ny, ng = a.ng;
    const int k = p * YM_WS_NG + wave;
    const int row = lane & 31, half = lane >> 5;
    if (k >= nt) return;
    const bool job = row < ny && half * YM_RG_G < nx;
    const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
    const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
    const int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
    const int32_t *cy = cx + a.dim_stride;
    const double2 cs = a.ctrig[(size_t)b * a.nt_stride + k];
    const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
    const int nq = st.nq;
    for (int gs = 0; gs < ng; gs++) { // set gs = the beams [gs * FLUSH, (gs + 1) * FLUSH)
        uint32_t acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0u;
        if (job)
            for (int j = 0; j < YM_RG_G; j++) {
                const int ix = half * YM_RG_G + j;
                if (ix >= nx) break;
                const int base = cy[row] * lin_pitch(a.g) + cx[ix];
                unsigned sum = 0;
                const int i1 = min(nq, (gs + 1) * YM_RG_FLUSH);
                for (int i = gs * YM_RG_FLUSH; i < i1; i++)
                    sum += cell_value(a.g, grid, limit, (unsigned)(base + lookup_offset_sem(a.g, ql[i], cs.x, cs.y, st.off_x, st.off_y, a.g.semantics == 1 ? st.ylat[0] : st.off_x, a.g.semantics == 1 ? st.ylat[1] : st.off_y, lin_pitch(a.g))));
                acc[2 * (j >> 2) + (j & 1)] += sum << (16 * ((j >> 1) & 1));
            }
        rg_odd(acc);
        store_partial16(a.partial + (size_t)b * a.partial_stride + (((size_t)gs * nt + k) * 64 + lane) * 16, acc);
    }
}

// grid (8 * gpx * parts): block j runs on XCD j % 8 (workgroups go round the XCDs in launch order); of the gpx * parts
// blocks of an XCD, `parts` neighbours form a group that works through the items x + 8 (t + gpx s), s = 0, 1, ...
__global__ __launch_bounds__(64 * (YM_WS_NG + YM_WS_NL), 8 /* two blocks of 16 waves per CU = 8 waves per SIMD */) void correlate_region_ws_kernel(RegionArgs a) {
    constexpr int NG = YM_WS_NG;
    constexpr int NT = 64 * (NG + YM_WS_NL);
    constexpr int LPS = 64 / YM_RG_SEGS;                              // rows a loader wave covers at once (60 of its 64 lanes copy)
    constexpr int PER = (YM_WS_ROWS + LPS - 1) / LPS;                 // copy tasks per loader thread
    __shared__ __attribute__((aligned(16))) unsigned char bufs[2 * YM_WS_BUF];
    __shared__ uint2 elist[NG][32];                   // per gather wave: its first 128 entries of the region it is about to gather
    __shared__ uint32_t full[2], done[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool gatherer = wave < NG;
    const int parts = a.parts, gpx = a.gpx;
    const int xcd = (int)blockIdx.x & 7, qx = (int)blockIdx.x >> 3;
    const int team = qx / parts, p = qx - team * parts;
    const int nt = a.lat.nt, nx = a.lat.nx, ny = a.lat.ny, ng = a.ng;
    const int k = p * NG + wave;
    const bool kvalid = gatherer && k < nt;
    const int row = lane & 31, half = lane >> 5;
    const bool job = row < ny && half * YM_RG_G < nx;
    const int half_pitch = a.g.pitch / 2;
    const int plane_bytes = half_pitch * a.g.win_w;
    const uint32_t lds0 = (uint32_t)(size_t)bufs;
    const uint32_t lane_part = (uint32_t)((job ? row : 0) * YM_RG_PITCH + (half * YM_RG_G < nx ? half * YM_RG_G : 0));
    // The counters: a wave that has stored (gathered) says so once, lane 0, after its own LDS operations have completed.
    // Written out by hand: the release / acquire forms of the atomics also wait for the wave's vector-memory operations
    // (vmcnt(0): the HIP memory model orders global memory too) -- for a loader that is the next region's loads in flight.
    // What is needed: the LDS executes a CU's operations in order, so "my stores have completed (lgkmcnt(0)), then the
    // add" on one side and "the load that saw the count, then my reads" on the other is enough.
    auto signal = [&](bool is_full, int which) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) {
            if (is_full) __hip_atomic_fetch_add(&full[which], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(&done[which], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto wait_for = [&](bool is_full, int which, uint32_t target) {
        for (;;) {
            // (one lane reads: sixteen waiting loader waves polling with all their lanes take a fifth of the LDS cycles)
            uint32_t seen = 0u;
            if (lane == 0)
                seen = is_full ? __hip_atomic_load(&full[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                               : __hip_atomic_load(&done[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((int32_t)((uint32_t)__builtin_amdgcn_readfirstlane(seen) - target) >= 0) break;
            if (is_full) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(8);
        }
        asm volatile("" ::: "memory");
    };
    for (int i = tid; i < 2 * YM_WS_ZERO_BYTES / 4; i += NT) {
        const int which = i >= YM_WS_ZERO_BYTES / 4;
        reinterpret_cast<uint32_t *>(bufs + which * YM_WS_BUF + YM_WS_ZERO)[i - which * (YM_WS_ZERO_BYTES / 4)] = 0u;
    }
    if (tid < 2) { full[tid] = 0u; done[tid] = 0u; }
    __syncthreads(); // (the only barrier)
    if (team >= gpx) return;
    uint32_t g = 0u; // regions this block's stream has held so far: region g of the stream lives in buffer g & 1
    for (int s = 0;; s++) {
        const int b = xcd + 8 * (team + gpx * s);
        if (b >= a.nitems) break;
        const YmItemState &st = a.states[b];
        const int qslot = __builtin_amdgcn_readfirstlane(st.qslot);
        const int32_t *__restrict__ starts = a.starts + (size_t)qslot * a.starts_stride;
        const uint32_t *__restrict__ wk = a.walk + ((size_t)qslot * parts + p) * YM_WS_WALK_WORDS;
        const bool regular = st.regular[0] && a.force_irregular != 1 && starts[a.nbins] >= 0;
        const int nlist = regular ? __builtin_amdgcn_readfirstlane((int)wk[0]) : 0;
        if (!gatherer) {
            // ---- LOAD: loader l belongs to group l >> 2 and stages class image l & 3 of the regions g + ri with (g + ri) & 1 = its group.
            // Lane = one 16-byte segment of the rows r0, r0 + LPS, ...  A loader's round is one exposed memory round trip (load,
            // wait, store); the two groups' round trips overlap, and with two blocks per CU sixteen loader waves keep 144 KB of
            // requests in flight, what the first form's 24 waves did.
            const uint32_t lw = (uint32_t)(wave - NG), cls = lw & 3u;
            const int grp = (int)(lw >> 2);
            const uint32_t seg = (uint32_t)lane % YM_RG_SEGS, r0 = (uint32_t)lane / YM_RG_SEGS;
            const bool copier = lane < LPS * YM_RG_SEGS;
            const uint32_t src0 = (cls & 1u) * (uint32_t)plane_bytes + (2u * r0 + (cls >> 1)) * (uint32_t)half_pitch + 16u * seg;
            const uint32_t src_step = 2u * LPS * (uint32_t)half_pitch;
            const uint32_t dst0 = (cls * YM_WS_ROWS + r0) * YM_RG_PITCH + 16u * seg;
            const uint8_t *__restrict__ planes = a.planes + (size_t)b * a.grid_stride;
            unsigned char *buf = bufs + grp * YM_WS_BUF;
            uint4 v[PER];
            auto band_in = [&](int q, uint32_t bx) { // (wave-uniform: a band the box does not reach issues no load and no store)
                return (uint32_t)(q * LPS) <= ((bx >> 8) & 0xffu) && (uint32_t)(q * LPS + LPS - 1) >= (bx & 0xffu);
            };
            // (the unit's walk in two registers per table, one region per lane: a round reads it with v_readlane instead of
            //  waiting for a memory round trip before it can issue its loads)
            const uint32_t rlA = lane < nlist ? wk[16 + lane] : 0u, rlB = 64 + lane < nlist ? wk[16 + 64 + lane] : 0u;
            const uint32_t bxA = lane < nlist ? wk[16 + YM_RG_MAX_REGIONS + lane] : 0u, bxB = 64 + lane < nlist ? wk[16 + YM_RG_MAX_REGIONS + 64 + lane] : 0u;
            for (int ri = ((g & 1u) == (uint32_t)grp) ? 0 : 1; ri < nlist; ri += 2) {
                const uint32_t gg = g + (uint32_t)ri;
                const uint32_t bx = ri < 64 ? __builtin_amdgcn_readlane(bxA, ri) : __builtin_amdgcn_readlane(bxB, ri - 64);
                const int R = (int)(ri < 64 ? __builtin_amdgcn_readlane(rlA, ri) : __builtin_amdgcn_readlane(rlB, ri - 64));
                const int RX = R % a.nrx, RY = R / a.nrx;
                const uint8_t *src = planes + ((size_t)(2 * RY * YM_WS_H) * half_pitch + (size_t)RX * YM_RG_W);
                const bool seg_in = copier && seg >= ((bx >> 16) & 0xffu) && seg <= (bx >> 24);
#pragma unroll
                for (int q = 0; q < PER; q++)
                    if (band_in(q, bx) && !(a.pad & 1)) v[q] = *reinterpret_cast<const uint4 *>(src + (seg_in ? src0 + (uint32_t)q * src_step : 0u));
                if (gg >= 2u) wait_for(false, grp, (uint32_t)NG * (gg >> 1)); // the buffer's previous region: every gatherer is done with it
#pragma unroll
                for (int q = 0; q < PER; q++) {
                    uint32_t *d = reinterpret_cast<uint32_t *>(buf + dst0 + (uint32_t)(q * LPS * YM_RG_PITCH));
                    // (the last band reaches past the class image: rows beyond it belong to the next image)
                    const bool row_in = (q + 1) * LPS <= YM_WS_ROWS || r0 + (uint32_t)(q * LPS) < (uint32_t)YM_WS_ROWS;
                    if (band_in(q, bx) && seg_in && row_in && !(a.pad & 1)) { d[0] = v[q].x; d[1] = v[q].y; d[2] = v[q].z; d[3] = v[q].w; }
                }
                signal(true, grp);
            }
            g += (uint32_t)nlist;
            continue;
        }
        // ---- GATHER
        uint32_t acc[8];
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0u;
        int in_set = 0, flushed = 0;
        auto flush = [&]() {
            if (flushed < ng) {
                rg_odd(acc);
                store_partial16(a.partial + (size_t)b * a.partial_stride + (((size_t)flushed * nt + k) * 64 + lane) * 16, acc);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = 0u;
            flushed++;
            in_set = 0;
        };
        if (regular) {
            const uint2 *__restrict__ entries4 = reinterpret_cast<const uint2 *>(a.entries + (size_t)qslot * a.entries_stride);
            const uint32_t *__restrict__ wseg = wk + 16 + (2 + wave) * YM_RG_MAX_REGIONS;
            auto gather = [&](uint32_t lane_off, int lo, int hi) { // entries [lo, hi)
                const int lds_hi = min(hi, lo + 128);
                if (lo < lds_hi) {
                    const uint2 *el = elist[wave];
                    const int n4 = (lds_hi - lo) >> 2;
                    // two quads per trip, each read one quad ahead into its own pair of registers
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
            // (the wave's segments of the unit's regions, one region per lane: v_readlane per round, not a memory round trip)
            const uint32_t sgA = (kvalid && lane < nlist) ? wseg[lane] : 0u, sgB = (kvalid && 64 + lane < nlist) ? wseg[64 + lane] : 0u;
            auto seg_of = [&](int ri) { return ri < 64 ? __builtin_amdgcn_readlane(sgA, ri) : __builtin_amdgcn_readlane(sgB, ri - 64); };
            int s0 = 0, s2 = 0;
            if (nlist > 0) {
                const uint32_t sg = seg_of(0);
                s0 = (int)(sg & 0xffffu); s2 = (int)(sg >> 16);
                uint2 ev = make_uint2(0u, 0u);
                if (s0 + 4 * lane < s2 && lane < 32) ev = entries4[(s0 >> 2) + lane];
                if (lane < 32) elist[wave][lane] = ev; // (the wave's own list: no other wave reads it)
            }
            for (int ri = 0; ri < nlist; ri++) {
                const uint32_t gg = g + (uint32_t)ri;
                int n0 = 0, n2 = 0;
                uint2 ev = make_uint2(0u, 0u);
                if (ri + 1 < nlist) { // the next region's entries travel while this one is gathered
                    const uint32_t sg = seg_of(ri + 1);
                    n0 = (int)(sg & 0xffffu); n2 = (int)(sg >> 16);
                    if (n0 + 4 * lane < n2 && lane < 32) ev = entries4[(n0 >> 2) + lane];
                }
                wait_for(true, (int)(gg & 1u), (uint32_t)(YM_WS_NL / 2) * ((gg >> 1) + 1u));
                if (!(a.pad & 2)) gather(lds0 + (gg & 1u) * (uint32_t)YM_WS_BUF + lane_part, s0, s2);
                if (ri + 1 < nlist && lane < 32) elist[wave][lane] = ev;
                signal(false, (int)(gg & 1u)); // (the reads of this region and the store above have completed)
                s0 = n0; s2 = n2;
            }
            g += (uint32_t)nlist;
        } else continue; // (an item without a lattice or a list: region_percell_kernel)
        if (!kvalid) continue;
        // the sums leave as sets of 16-bit partials; score_kernel turns them into responses (scoring here, in the wave that
        // holds the sums, saves that kernel's 85 us per 4096 items and costs this one 100 spilled registers)
        while (flushed < ng) flush(); // the set being filled, then empty ones
    }
}

#endif // YM_EXPERIMENTAL


}  // namespace ym
