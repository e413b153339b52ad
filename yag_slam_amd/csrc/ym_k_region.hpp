// ym_k_region.hpp -- K4r: the coarse correlate of BATCHES on lattices up to 26 x 32, staged through LDS region by region
// (round 2 form, tuned for the default 26 x 26 x 21 lattice; ym_k_gather.hpp is the general form: wider lattices, patches
// with multiplicities, any number of readings).
// Part of ym_kernels.hpp (include that, not this file).
//
// correlate_kernel (ym_k_correlate.hpp) gathers every (beam, angle) patch straight from the column planes and is bound
// by the vector L1: a quad of lanes costs one clock per 128-byte line it touches, 34 line visits per (beam, angle) wave
// load on a 26 x 26 lattice.  The LDS serves the same 16 bytes per lane in 8 clocks
// (scripts/exp/lds_gather.hip: two ds_read2_b32 per lane at a 4-byte-aligned address, rows 25 dwords apart, the two
// halves of a lattice row in the two halves of the wave: conflict-free, 128 B/clk) -- if the bytes are in LDS.
// They can be: the hypotheses of one beam are every other cell of every other row, i.e. a DENSE 26 x 26 block of bytes
// in the image of one (column parity, row parity) class of window cells, and the 22 701 patches of an item
// (1081 beams x 21 angles) lie along the walls.  So
//   bin_kernel            sorts the (beam, angle) pairs of an item by the 64 x 80-byte REGION of class space their patch
//                         starts in (key: region, angle), 16-bit entries;
//   correlate_region_kernel  walks the regions that hold work: copies the region (+ the 26-byte patch margin) of all four
//                         classes into LDS once -- 42 KB serve ~1000 patches of 676 bytes each -- and every wave, which
//                         owns one angle, gathers its patches from there into packed 16-bit sums kept in registers.
// The sums leave the registers every YM_RG_FLUSH patches of a wave, in lane order (score_kernel's layout 1); on batches
// whose integer sums nobody asked for the wave scores them itself at the end (fuse_score).
#pragma once

namespace ym {

// development build (-DYM_RG_PROF=1): every wave of correlate_region_kernel adds the shader clocks it spends per phase to
// stamps[8 + phase] (0 setup, 1 issue of the next region's loads, 2 gather, 3 barrier after the gather, 4 staging stores,
// 5 barrier after them, 6 scoring); scripts/dev/region_phases.py
#ifndef YM_RG_PROF
#define YM_RG_PROF 0
#endif
#if YM_RG_PROF
#if YM_RG_PROF == 2 // phase 1 in three parts: 7 the next region's box and segment (LDS), 8 its entry list (one global load), 1 the staging loads; 2 gather
#define YM_RG_PH(i) do { if ((i) == 7 || (i) == 8 || (i) == 1 || (i) == 2 || (i) == 5) { const uint32_t t_ = (uint32_t)__builtin_amdgcn_s_memtime(); \
    if ((i) != 5) ph[(i) == 7 ? 0 : (i) == 8 ? 1 : (i) == 1 ? 2 : 3] += t_ - pt; pt = t_; } } while (0)
#else
#define YM_RG_PH(i) do { const uint32_t t_ = (uint32_t)__builtin_amdgcn_s_memtime(); if ((i) >= 2 && (i) <= 5) ph[(i) - 2] += t_ - pt; pt = t_; } while (0)
#endif
#else
#define YM_RG_PH(i) do { } while (0)
#endif

// development build (-DYM_RG_ABLATE=bits, TIMING ONLY, wrong sums): 1 = the gather adds nothing (no LDS read, no vector work), 2 = no
// staging loads, 4 = no staging stores -- what each part of a round costs the kernel (scripts/dev/r05_ablate.sh)
#ifndef YM_RG_ABLATE
#define YM_RG_ABLATE 0
#endif
#ifndef YM_RG_PRIO
#define YM_RG_PRIO 0
#endif
#define YM_RG_W 64                            // region width and height in class bytes (128 x 160 window cells): what three
#ifndef YM_RG_H
#define YM_RG_H 80                            // blocks per CU leave room for in LDS
#endif
#define YM_RG_PITCH 100                       // LDS bytes per staged row: 25 dwords, odd -> 26 rows on 26 distinct banks
#define YM_RG_ROWS (YM_RG_H + 26)             // + the patch height
#define YM_RG_SEGS 6                          // 16-byte blocks staged per row (96 >= 64 + 26 + 3)
#define YM_RG_CLS (YM_RG_PITCH * YM_RG_ROWS)  // bytes per class image
#define YM_RG_G 13                            // hypotheses per lane: 13 bytes + 3 of misalignment = four dwords
#define YM_RG_ZERO (4 * YM_RG_CLS)               // LDS offset of an all-zero patch: what the padding entries point at
#define YM_RG_LDS_BYTES (YM_RG_ZERO + 26 * YM_RG_PITCH + 32)
// bytes the host keeps past the last item's planes: a staged region may start up to (ROWS - 1) * 2 + 1 rows and 96 bytes
// past the last cell of the second plane (never gathered, but read)
#define YM_RG_PLANES_SLACK(half_pitch) ((size_t)(2 * YM_RG_ROWS + 2 * YM_RG_H + 2) * (size_t)(half_pitch) + 256)
// the same past the last item's row-major window, for the kernel that stages from it (WIN): rows of `pitch` bytes, 2 * 16 * SEGS bytes along a row
#define YM_RG_WINDOW_SLACK(pitch) ((size_t)(2 * YM_RG_ROWS + 2 * YM_RG_H + 2) * (size_t)(pitch) + 512)
// (what the staging loop of correlate_region_kernel reads of the last item: class 3, task row ROWS - 1 of a region that
//  starts at most H - 1 class rows before the window's last one, i.e. plane row 2 * (H - 1 + ROWS - 1) + 1 past it, and
//  16 * SEGS bytes along it)
static_assert(2 * YM_RG_ROWS + 2 * YM_RG_H + 2 >= 2 * (YM_RG_H - 1 + YM_RG_ROWS - 1) + 1 + 1, "YM_RG_PLANES_SLACK does not cover the rows a staged region reads");
static_assert(256 >= 16 * YM_RG_SEGS, "YM_RG_PLANES_SLACK does not cover the bytes a staged row reads");
static_assert(16 * YM_RG_SEGS >= YM_RG_W + 2 * YM_RG_G + 3 && YM_RG_PITCH >= 16 * YM_RG_SEGS && (YM_RG_PITCH / 4) % 2 == 1,
              "a staged row must hold the region, the patch margin and the misalignment, at an odd number of dwords per row");
#define YM_RG_MAX_BINS 8192
#define YM_RG_MAX_REGIONS 96                   // regions with work a block of correlate_region_kernel can list
#define YM_RG_MAX_ENTRIES 28672
#define YM_RG_FLUSH 652                       // patches per set of 16-bit sums: 652 x 100 < 65536 (a multiple of four)
#define YM_BIN_THREADS 1024
#ifndef YM_BIN_MIN_WAVES
#define YM_BIN_MIN_WAVES 8 // waves per SIMD the compiler must leave room for: 8 = two blocks per CU (64 VGPRs)
#endif
#define YM_BIN_LDS_BYTES(nbins, entries, nboxes) ((size_t)(YM_BIN_THREADS / 64 + YM_MAX_COARSE_NT + YM_RG_MAX_BINS / 32 + 1) * 4 + (size_t)(nbins) * 8 + ((size_t)(entries) * 2 + 15) / 16 * 16 + (size_t)(nboxes) * 16)

struct RegionArgs {
    YmGeom g;
    YmLattice lat;
    const uint8_t *grid;
    const uint8_t *planes;
    size_t grid_stride;
    const double2 *ctrig;   // [B][nt_stride]
    const int32_t *hypcell; // [B][2][dim_stride]
    const YmItemState *states;
    const int32_t *qrep;    // [n_qslots] an item that uses the query slot
    uint16_t *entries;      // [Q][entries_stride]: LDS offset of the patch's first byte in the staged region, sorted by bin; one
    size_t entries_stride;  // list per QUERY SLOT of the call (the pairs depend on the query alone, not on the chain)
    int32_t *starts;        // [Q][starts_stride]: first entry of bin (region * nt + angle); [nbins] = total, or -1: no list
    size_t starts_stride;
    // [Q][rbox_stride]: per (region, block of nw angles) the box of its patches' origins inside the region, rmin | rmax << 8 |
    // xmin << 16 | xmax << 24 (class rows and bytes): a correlate block stages only the rows and 16-byte segments its own
    // patches read -- a wall crosses a region, it does not fill it
    uint32_t *rbox;
    size_t rbox_stride;
    int32_t nw, parts;      // waves (= angles) per correlate block, blocks per item
    uint16_t *partial;      // [B][ng][nt][64 lanes][16]: ng sets of 16-bit sums, each of at most YM_RG_FLUSH patches
    size_t partial_stride;
    int32_t nt_stride, dim_stride;
    int32_t nrx, nry, ng, nbins;
    int32_t force_irregular; // tests: 1 = take the per-cell path, 2 = bin_kernel reports that the padded list does not fit
    int32_t fuse_score;      // 1: the wave that holds an angle's sums also scores them (what score_kernel does otherwise)
    double *resp;            // [B][nt][ny][nx]                         (fuse_score)
    size_t sums_stride;
    double *blockmax;        // [B][n_blocks], block = (angle, YM_SCORE_THREADS cells)
    double *probs;           // [B][ny*nx] max over theta per (x, y); zeroed by the prepare stage
    size_t probs_stride;
    int32_t n_blocks, pad;
    // the region geometry the lists are built for (the two correlate kernels stage regions of different heights): class rows a
    // region owns, LDS bytes from one class image to the next, LDS offset of the all-zero patch the padding entries point at
    int32_t rg_h, rg_cls, rg_zero, pad2;
    // wave-specialised form (rg_w != 0): a region is ONE class image, rg_w class bytes x rg_h class rows, staged at a row
    // pitch of rg_pitch bytes; region index = (ry * nrx + rx) * 4 + class; nregions = 4 * nrx * nry (else nrx * nry)
    int32_t rg_w, rg_pitch, nregions, pad3;
    // wave-specialised form: the walks (region_walk_kernel: [Q][parts][YM_WS_WALK_WORDS]), items of the call, block teams per XCD
    uint32_t *walk;
    int32_t nitems, gpx;
    int32_t rsplit, pad4;    // correlate_region_kernel: blocks that share the regions of an (item, angle block); <= 1: one
    unsigned long long *stamps;
    // Round 6: the lists of a query are built PER ANGLE BLOCK (lparts blocks of lnw angles, by one bin_kernel block each: a third of the
    // pairs, 25 KB of LDS, eight waves -- such a block sits beside two region-correlate blocks of another lane, where the round-5 block
    // (all 21 angles: 65 KB, sixteen waves) pushed two of three off its CU).  starts: [Q][lparts][nbins + 1] first entry of bin
    // (region * lnw + angle - part * lnw), positions in the query's entry array (part p's entries start at p * entries_pstride);
    // then [lparts] flags: total of the part, or -1 = no list (that block's angles take the per-cell path).  nbins = regions * lnw.
    // (lparts = 1, lnw = nt is the round-5 layout: what the experimental forms read, built by bin_whole_kernel.)
    int32_t lnw, lparts;
    size_t entries_pstride;
};
// the starts row of list part p, and whether the part has a list
__device__ __forceinline__ const int32_t *rg_part_starts(const RegionArgs &a, int qslot, int p) {
    return a.starts + (size_t)qslot * a.starts_stride + (size_t)p * (a.nbins + 1);
}
__device__ __forceinline__ bool rg_part_listed(const RegionArgs &a, int qslot, int p) {
    return a.starts[(size_t)qslot * a.starts_stride + (size_t)a.lparts * (a.nbins + 1) + p] >= 0;
}

// the bin of one (beam, angle) pair and its entry = the LDS offset of the patch's first byte once the region is staged;
// false if the patch origin is outside the regions (never for a patch the window holds; kept so that nothing is ever
// written out of bounds)
// (region = its index, er / ex = the patch origin's class row and byte inside the region)
// (bin_nw: angles per region in the bin numbering -- all of them, or those of one list part; k counts from the part's first angle then)
__device__ __forceinline__ bool region_entry(const RegionArgs &a, int2 cell, int cx0, int cy0, int k, int &bin, unsigned &entry, int &region,
                                             unsigned &er, unsigned &ex, int bin_nw = -1) {
    const int X = cx0 + cell.x, Y = cy0 + cell.y;
    if (X < 0 || Y < 0) return false;
    const unsigned xc = (unsigned)X >> 1, yc = (unsigned)Y >> 1;
    const unsigned cls = (unsigned)((X & 1) | ((Y & 1) << 1));
#ifdef YM_EXPERIMENTAL
    if (a.rg_w) { // one class image per region (the wave-specialised form)
        const unsigned rx = xc / (unsigned)a.rg_w, ry = yc / (unsigned)a.rg_h;
        if ((int)rx >= a.nrx || (int)ry >= a.nry) return false;
        region = (int)((ry * (unsigned)a.nrx + rx) * 4u + cls);
        bin = region * a.lat.nt + k;
        er = yc - ry * (unsigned)a.rg_h;
        ex = xc - rx * (unsigned)a.rg_w;
        entry = er * (unsigned)a.rg_pitch + ex;
        return true;
    }
#endif
    // (round 5: every product below is of numbers below 2^24 -- v_mul_u32_u24 at the full rate instead of v_mul_lo_u32 at a quarter --
    //  and the division by the region height is a multiplication by pad2 = ceil(2^21 / rg_h), exact for class rows below 4096:
    //  the error yc (pad2 rg_h - 2^21) / (2^21 rg_h) stays below 4096 / 2^21 < 1 / rg_h for rg_h <= 255; this kernel is a whole launch of
    //  4096 blocks when every item of a batch has its own query)
    if (yc >= 4096u) return false; // (never: a window is at most 4073 cells wide)
    const unsigned rx = xc / (unsigned)YM_RG_W, ry = __umul24(yc, (unsigned)a.pad2) >> 21;
    if ((int)rx >= a.nrx || (int)ry >= a.nry) return false;
    region = (int)(__umul24(ry, (unsigned)a.nrx) + rx);
    bin = (int)__umul24((unsigned)region, (unsigned)(bin_nw < 0 ? a.lat.nt : bin_nw)) + k;
    er = yc - __umul24(ry, (unsigned)a.rg_h);
    ex = xc - rx * YM_RG_W;
    entry = __umul24(cls, (unsigned)a.rg_cls) + __umul24(er, (unsigned)YM_RG_PITCH) + ex;
    return true;
}

// ---- the pair lists, per angle block (round 6).  grid (lparts, Q): block (p, q) sorts the (beam, angle) pairs of query slot q
// whose angle lies in [p * lnw, (p + 1) * lnw) by (region, angle); YM_BINP_THREADS threads.  Counting sort in LDS: count, scan, place (the order
// inside a bin is arbitrary: the sums are integers).  GridIndexLookup::ComputeOffsets for the part's coarse angles happens here.
// Inside a bin the entries are sorted by their byte misalignment (entry & 3; class images and rows are multiples of 4
// bytes), every run of equal misalignment is padded to an even length and the bin to a multiple of four with entries
// that point at the all-zero patch: the gather then adds the raw dwords of a PAIR of patches before one byte funnel, and
// never meets a ragged group.  A part whose padded list would not fit -- its share of the buffer, one angle's share of the ng sets of
// 16-bit sums the gather may fill, or more regions with work than a correlate block can list -- gets its flag = -1 and the
// correlate block of those angles takes the per-cell path.
// YAG: the lookup cells of the reference's Python matcher (items yag_lattice_kernel proved regular; ym_k_common.hpp, lookup_cell_sem).
// MAXP: pairs per thread the instantiation has registers for (18: scans of up to 1152 readings at eight angles per part; 32: up to 2047).
#define YM_BINP_THREADS 512
#define YM_BINP_LDS_BYTES(nbins, entries, nboxes) ((size_t)(YM_BINP_THREADS / 64 + 16 + YM_RG_MAX_BINS / 32 + 1) * 4 + (size_t)(nbins) * 8 + ((size_t)(entries) * 2 + 15) / 16 * 16 + (size_t)(nboxes) * 16)
template <bool YAG, int MAXP>
__global__ __launch_bounds__(YM_BINP_THREADS, MAXP <= 18 ? 4 : 2) void bin_kernel(RegionArgs a) {
    constexpr int NT = YM_BINP_THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char bin_smem[];
    int *wave_tot = reinterpret_cast<int *>(bin_smem);                                   // [NT / 64]
    int *angle_tot = wave_tot + NT / 64;                                                 // [16] padded entries per angle of the part
    unsigned *region_bits = reinterpret_cast<unsigned *>(angle_tot + 16);                // [YM_RG_MAX_BINS / 32] regions that hold a patch
    int *regions_used = reinterpret_cast<int *>(region_bits + YM_RG_MAX_BINS / 32);
    unsigned (*cnt)[2] = reinterpret_cast<unsigned (*)[2]>(regions_used + 1);             // [nbins] four 16-bit counters (one per
                                                                                         // misalignment), later the runs' first positions
    unsigned short *ent = reinterpret_cast<unsigned short *>(cnt + a.nbins);             // [entries_pstride]
    const int nboxes = a.nregions;
    unsigned (*box)[4] = reinterpret_cast<unsigned (*)[4]>(bin_smem + ((reinterpret_cast<unsigned char *>(ent + a.entries_pstride) - bin_smem) + 15) / 16 * 16); // [nregions] rmin, rmax, xmin, xmax
    const int part = blockIdx.x, qs = blockIdx.y, b = a.qrep[qs], tid = threadIdx.x, lane = tid & 63;
    const YmItemState &st = a.states[b];
    const int nw = a.lnw, k_lo = part * nw, nk = min(a.lat.nt, k_lo + nw) - k_lo;
    const int nq = st.nq;
    const int total = nq * nk;
    const int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
    const int cx0 = cx[0], cy0 = cx[a.dim_stride];
    const double off_x = st.off_x, off_y = st.off_y;
    const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
    const double2 *trig = a.ctrig + (size_t)b * a.nt_stride + k_lo;
    int32_t *starts = a.starts + (size_t)qs * a.starts_stride + (size_t)part * (a.nbins + 1);
    int32_t *flag = a.starts + (size_t)qs * a.starts_stride + (size_t)a.lparts * (a.nbins + 1) + part;
    const int pos0 = part * (int)a.entries_pstride; // the part's first position in the query's entry array
    // the query's sensor-frame points and the part's angle table go through LDS (they borrow the entry list's room, which nothing
    // writes before pass 1 is over): pass 1 reads point i of angle k for `per` consecutive pairs per thread
    double2 *qls = reinterpret_cast<double2 *>(bin_smem + ((reinterpret_cast<unsigned char *>(ent) - bin_smem) + 15) / 16 * 16);
    const bool ql_in_lds = (size_t)(nq + nk) * sizeof(double2) + 16 <= (size_t)a.entries_pstride * 2;
    double2 *trigs = qls + nq;
    if (ql_in_lds) {
        for (int i = tid; i < nq; i += NT) qls[i] = ql[i];
        if (tid < nk) trigs[tid] = trig[tid];
    }
    for (int i = tid; i < a.nbins * 2; i += NT) (&cnt[0][0])[i] = 0u;
    if (tid < 16) angle_tot[tid] = 0;
    if (tid < YM_RG_MAX_BINS / 32) region_bits[tid] = 0u;
    if (tid == 0) *regions_used = 0;
    for (int i = tid; i < nboxes; i += NT) { box[i][0] = 255u; box[i][1] = 0u; box[i][2] = 255u; box[i][3] = 0u; }
    __syncthreads();
    // Pass 1: bin, entry and RANK inside (bin, misalignment) of every pair, kept in registers (the rank is what the
    // counting atomic returns), so that pass 2 neither recomputes the cells nor needs a second atomic.
    unsigned key[MAXP];          // rank & 7 << 29 | bin << 16 | entry; 0xffffffff = no pair
    unsigned rank_hi[(MAXP + 3) / 4];
#pragma unroll
    for (int q = 0; q < (MAXP + 3) / 4; q++) rank_hi[q] = 0u;
    // A thread takes `per` CONSECUTIVE pairs (beam i of angle k, stepped from one pair to the next): neighbouring beams of one
    // angle mostly share their bin and nearly always their box, so the lanes of a wave -- `per` beams apart -- spread over
    // the counters, and a thread keeps the box of its run of pairs in registers and sends it when the box changes.
    const int per_thread = (total + NT - 1) / NT;
    const int p0 = tid * per_thread;
    int k = p0 / max(nq, 1), i = p0 - k * nq; // (k counts from the part's first angle)
    int cur = -1;                                 // the box the thread is collecting, and its extent so far
    unsigned r0 = 255u, r1 = 0u, x0 = 255u, x1 = 0u;
    auto send_box = [&]() {
        if (cur >= 0) {
            unsigned *bx = box[cur];
            atomicMin(&bx[0], r0); atomicMax(&bx[1], r1); atomicMin(&bx[2], x0); atomicMax(&bx[3], x1);
        }
    };
    bool overflow = per_thread > MAXP; // (the host picks the instantiation by the longest scan: never)
    auto pass1 = [&](auto point_of, auto trig_of) {
#pragma unroll
        for (int q = 0; q < MAXP; q++) {
            const int pp = p0 + q;
            key[q] = 0xffffffffu;
            if (q < per_thread && pp < total) {
                const double2 cs = trig_of(k);
                const double2 pt = point_of(i);
                int bin, region; unsigned e, er, ex;
                const int2 lc = YAG ? lookup_cell_sem(a.g, pt, cs.x, cs.y, off_x, off_y, st.ylat[0], st.ylat[1]) : lookup_cell(pt, cs.x, cs.y, off_x, off_y, a.g.scale);
                if (region_entry(a, lc, cx0, cy0, k, bin, e, region, er, ex, nw)) {
                    const unsigned rank = (atomicAdd(&cnt[bin][(e >> 1) & 1u], 1u << (16 * (e & 1u))) >> (16 * (e & 1u))) & 0xffffu;
                    key[q] = (rank & 7u) << 29 | (unsigned)bin << 16 | e;
                    rank_hi[q >> 2] |= ((rank >> 3) & 0xffu) << (8 * (q & 3));
                    if (region != cur) { send_box(); cur = region; r0 = r1 = er; x0 = x1 = ex; }
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
        for (int r = tid; r < nboxes; r += NT) rb[(size_t)r * a.lparts + part] = box[r][0] | box[r][1] << 8 | box[r][2] << 16 | box[r][3] << 24;
    }
    // exclusive scan of the padded bin sizes: thread t owns the bins [t * per, (t + 1) * per)
    const int per = (a.nbins + NT - 1) / NT;
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
                    atomicAdd(&angle_tot[(first + j) % nw], padded);
                    const int R = (first + j) / nw;
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
        for (int w = 0; w < NT / 64; w++) {
            if (w < (tid >> 6)) base += wave_tot[w];
            all += wave_tot[w];
        }
        padded_total = all;
        bool fits = all <= (int)a.entries_pstride && pos0 + all <= 65532 && a.force_irregular != 2 && !overflow;
        for (int kk = 0; kk < nk; kk++) fits = fits && angle_tot[kk] <= a.ng * YM_RG_FLUSH;
        fits = fits && *regions_used <= YM_RG_MAX_REGIONS; // (what a block of correlate_region_kernel can list)
        int run = base + incl - local;
        for (int j = 0; j < per; j++)
            if (first + j < a.nbins) {
                const unsigned c01 = cnt[first + j][0], c23 = cnt[first + j][1];
                const int c[4] = {(int)(c01 & 0xffffu), (int)(c01 >> 16), (int)(c23 & 0xffffu), (int)(c23 >> 16)};
                starts[first + j] = pos0 + run;
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
        if (tid == 0) {
            starts[a.nbins] = pos0 + all;
            *flag = fits ? all : -1;
            if (a.stamps && qs == 0 && part == 0) a.stamps[26] = (unsigned long long)all;
        }
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
    uint32_t *out = reinterpret_cast<uint32_t *>(a.entries + (size_t)qs * a.entries_stride + pos0);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(ent);
    for (int i2 = tid; i2 < (padded_total + 1) / 2; i2 += NT) out[i2] = src[i2];
}

// The 16 bytes at LDS byte address `addr` (any alignment): two ds_read2_b32 at the dword below + a byte funnel.
// (A ds_read_b128 at a 4-byte-aligned address is legal on gfx950 but takes 64 clk per wave, lds_gather.hip.)
// The registers an asm statement that only ISSUES a read names as outputs are not written when the statement ends, and
// the compiler is free to copy them right there (seen: wrong sums): every such register is either waited for inside
// the issuing statement or passes through the statement that waits for it ("+v") before anything else touches it.
typedef unsigned int rg_u32x2 __attribute__((ext_vector_type(2)));
// Four patches (their LDS origins: four 16-bit entries, the same in every lane -- a broadcast read of the wave's entry list)
// into the packed 16-bit sums (acc[2j]: hypotheses 4j, 4j + 2; acc[2j + 1]: 4j + 1, 4j + 3).  Patches 2i and 2i + 1 share their misalignment (bin_kernel): grid
// bytes are at most 100, so their RAW dwords add without carries and one funnel serves both.
// Round 4: the funnel and the widening are ONE v_perm_b32 per pair of hypotheses -- selector (rr, 0x0c, rr + 2, 0x0c) picks bytes
// rr and rr + 2 of the eight bytes s[j + 1] : s[j] into the low bytes of the two 16-bit fields and zeroes the rest (0x0c = the
// constant 0), (rr + 1, 0x0c, rr + 3, 0x0c) the odd ones: 14 v_perm per pair of patches instead of 8 v_alignbyte + 8 v_and +
// 6 v_lshrrev, and the odd sums are clean 16-bit fields like the even ones.  The loop alone: 11.5 -> 10.1 CU clocks per patch
// (scripts/exp/rg_proto.hip variant 2, profiles/r04_rg_proto.txt).
__device__ __forceinline__ void rg_perm_pair(const rg_u32x2 &pa, const rg_u32x2 &qa, const rg_u32x2 &pb, const rg_u32x2 &qb, uint32_t rr,
                                             uint32_t (&e)[4], uint32_t (&o)[4]) {
    const uint32_t s0 = pa.x + pb.x, s1 = pa.y + pb.y, s2 = qa.x + qb.x, s3 = qa.y + qb.y; // raw dwords of two patches: bytes <= 200
    const uint32_t selE = rr * 0x00010001u + 0x0c020c00u, selO = selE + 0x00010001u;
    e[0] = __builtin_amdgcn_perm(s1, s0, selE); o[0] = __builtin_amdgcn_perm(s1, s0, selO);
    e[1] = __builtin_amdgcn_perm(s2, s1, selE); o[1] = __builtin_amdgcn_perm(s2, s1, selO);
    e[2] = __builtin_amdgcn_perm(s3, s2, selE); o[2] = __builtin_amdgcn_perm(s3, s2, selO);
    e[3] = __builtin_amdgcn_perm(0u, s3, selE); // byte 12 (the 13th hypothesis) is at most byte 15 of the four dwords
    o[3] = 0u;                                  // (acc[7] would hold hypotheses 13 and 15 of the lane: there are only 13)
}
// All eight reads of the four patches are issued at once; the first pair is funnelled while the second pair's reads are
// still in flight (LDS reads return in order: lgkmcnt(4) = the first four are back).
// The registers an asm statement that only ISSUES a read names as outputs are not written when the statement ends, and
// the compiler is free to copy them right there (seen: wrong sums): every such register is either waited for inside
// the issuing statement or passes through the statement that waits for it ("+v") before anything else touches it.
__device__ __forceinline__ void rg_gather4(uint32_t (&acc)[8], uint32_t lane_off, uint2 ee /* four 16-bit origins, wave-uniform */) {
    const uint32_t ad0 = lane_off + (ee.x & 0xffffu), ad1 = lane_off + (ee.x >> 16), ad2 = lane_off + (ee.y & 0xffffu), ad3 = lane_off + (ee.y >> 16);
#if YM_RG_ABLATE & 1
    acc[0] += ad0 ^ ad1 ^ ad2 ^ ad3; // (the entries are still read)
    return;
#endif
    rg_u32x2 p0, q0, p1, q1, p2, q2, p3, q3;
    asm volatile("ds_read2_b32 %0, %8 offset1:1\n\tds_read2_b32 %1, %8 offset0:2 offset1:3\n\t"
                 "ds_read2_b32 %2, %9 offset1:1\n\tds_read2_b32 %3, %9 offset0:2 offset1:3\n\t"
                 "ds_read2_b32 %4, %10 offset1:1\n\tds_read2_b32 %5, %10 offset0:2 offset1:3\n\t"
                 "ds_read2_b32 %6, %11 offset1:1\n\tds_read2_b32 %7, %11 offset0:2 offset1:3\n\t"
                 "s_waitcnt lgkmcnt(4)"
                 : "=&v"(p0), "=&v"(q0), "=&v"(p1), "=&v"(q1), "=&v"(p2), "=&v"(q2), "=&v"(p3), "=&v"(q3)
                 : "v"(ad0 & ~3u), "v"(ad1 & ~3u), "v"(ad2 & ~3u), "v"(ad3 & ~3u)
                 : "memory");
    uint32_t e[2][4], o[2][4];
    rg_perm_pair(p0, q0, p1, q1, ad0 & 3u, e[0], o[0]);
    // (the second pair's registers are written by the LDS until here: they pass through this statement and nothing else)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p2), "+v"(q2), "+v"(p3), "+v"(q3) : : "memory");
    rg_perm_pair(p2, q2, p3, q3, ad2 & 3u, e[1], o[1]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        acc[2 * j] = acc[2 * j] + e[0][j] + e[1][j];
        if (j < 3) acc[2 * j + 1] = acc[2 * j + 1] + o[0][j] + o[1][j];
    }
}

// (until round 4 the odd accumulators held the running sum of dword >> 8 and were separated here; v_perm hands them over as
//  clean 16-bit fields, so this is the identity -- kept as the one place that says what layout the sums leave the loop in:
//  acc[2j] = hypotheses 4j | 4j + 2 << 16, acc[2j + 1] = 4j + 1 | 4j + 3 << 16)
__device__ __forceinline__ void rg_odd(uint32_t (&)[8]) {}

// grid (P, B): block (p, item) = NW waves, wave w owns coarse angle p * NW + w.  Lane = 13 x-adjacent hypotheses of one
// lattice row: row = lane & 31, half = lane >> 5 (nx <= 26, ny <= 32: checked on the host).
// The walk over the regions is a two-stage pipeline: while region i is gathered, the global loads of region i + 1 are in
// flight (registers) together with the wave's first 128 entries of it; they go to LDS between the two barriers that end
// the gather.
// WIN (round 4, large batches): the regions are staged from the ROW-MAJOR WINDOW, not from its column planes -- a staged row of
// the two column classes of one row parity is 192 contiguous window bytes (16 bytes = eight class bytes of each parity, split by
// two v_perm_b32 on their way into LDS) instead of 96 + 96 bytes half a megabyte apart, and the raster of such a call does not
// write the planes at all: half its bytes (on a box whose memory takes writes slowly the raster's time is its stores).
template <int NW, bool WIN = false>
__global__ __launch_bounds__(64 * NW, NW <= 8 ? (3 * NW + 3) / 4 : 1 /* three blocks per CU: 80 VGPRs */) void correlate_region_kernel(RegionArgs a) {
    constexpr int NT = 64 * NW;
    constexpr int PER = (YM_RG_ROWS + (NT / 4) / YM_RG_SEGS - 1) / ((NT / 4) / YM_RG_SEGS); // copy tasks per thread (rows of its segment)
    __shared__ __attribute__((aligned(16))) unsigned char region[YM_RG_LDS_BYTES]; // four class images + the zero patch
    __shared__ int rlist[YM_RG_MAX_REGIONS];
    __shared__ uint32_t rboxl[YM_RG_MAX_REGIONS];     // per listed region: first row | last row << 8 | first segment << 16 | last << 24 to stage
    __shared__ unsigned short seginfo[NW][YM_RG_MAX_REGIONS][2]; // per wave and listed region: its first entry and the end
    // (until round 5 a wave's first 256 entries of the region being gathered went through LDS -- uint2 elist[NW][64], a broadcast
    //  ds_read_b64 per four patches; the compiler, seeing a wave-uniform value, put a v_readfirstlane and with it an
    //  s_waitcnt lgkmcnt(0) right behind every such read: one exposed LDS round trip per four patches, behind the queue of every
    //  other wave's gather reads.  They now stay in the register pair they arrive in -- lane i = entries 4 i .. 4 i + 3 -- and
    //  quad c comes out of it with two v_readlane)
    __shared__ int rcount;
    // Small batches (fewer blocks than the chip holds): the listed regions of an (item, angle block) are dealt out to rsplit
    // blocks, each with its own sets of partial sums -- a block's region walk is a chain of ~20 stage / gather rounds, 118 us
    // whatever the batch, and only more blocks shorten it (score_kernel adds the sets; no fused scoring then).
    int p;
    const int b = xcd_item_of_block_2d(p);
    const int rsplit = a.rsplit > 1 ? a.rsplit : 1, rsi = p / a.parts;
    p -= rsi * a.parts;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const YmItemState &st = a.states[b];
    const int nt = a.lat.nt, nx = a.lat.nx, ny = a.lat.ny, ng = a.ng;
    const int k = p * NW + wave;
    const bool kvalid = k < nt;
    const int row = lane & 31, half = lane >> 5;
    const bool job = row < ny && half * YM_RG_G < nx;
    const int half_pitch = a.g.pitch / 2;
    const int plane_bytes = half_pitch * a.g.win_w;
    const uint8_t *__restrict__ planes = a.planes + (size_t)b * a.grid_stride;
    // (the lists of this block's angles: list part p -- the host builds them with lnw = NW, lparts = parts)
    const int32_t *__restrict__ starts = rg_part_starts(a, st.qslot, p);
    const uint16_t *__restrict__ entries = a.entries + (size_t)st.qslot * a.entries_stride;
    const uint32_t lds0 = (uint32_t)(size_t)region;
    // idle lanes read what lane (row 0, same half) reads: the same address is a broadcast, any other address could share a
    // bank with a working lane
    const uint32_t lane_off = lds0 + (uint32_t)((job ? row : 0) * YM_RG_PITCH + (half * YM_RG_G < nx ? half * YM_RG_G : 0));
    uint32_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0u;
    // the 16-bit sums hold YM_RG_FLUSH patches: the wave counts what it has added and, when a set is full, writes it out as
    // partial set number `flushed` and starts the next (score_kernel adds the sets)
    int in_set = 0, flushed = 0;
    auto flush = [&]() {
        if (flushed < ng) {
            rg_odd(acc);
            store_partial16(a.partial + (size_t)b * a.partial_stride + (((size_t)(rsi * ng + flushed) * nt + k) * 64 + lane) * 16, acc);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc[j] = 0u;
        flushed++;
        in_set = 0;
    };
#if !YM_RG_PROF
    YM_STAMP(a, 8);
#endif
#if YM_RG_PROF
    uint32_t ph[4] = {0u, 0u, 0u, 0u};
    uint32_t pt = (uint32_t)__builtin_amdgcn_s_memtime();
    const uint32_t pt_begin = pt;
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime(); // (100 MHz: against the shader clocks of pt it gives the clock the wave ran at)
#endif
    const bool regular = st.regular[0] && a.force_irregular != 1 && rg_part_listed(a, st.qslot, p);
    if (regular) {
        const int k_lo = p * NW, k_hi = min(nt, k_lo + NW);
        const int nreg = a.nrx * a.nry;
        for (int i = tid; i < (YM_RG_LDS_BYTES - YM_RG_ZERO) / 4; i += NT) reinterpret_cast<uint32_t *>(region + YM_RG_ZERO)[i] = 0u;
        // the regions in which a patch of this block's angles starts
        if (wave == 0) {
            int n = 0;
            for (int R0 = 0; R0 < nreg; R0 += 64) {
                const int R = R0 + lane;
                const bool has = R < nreg && starts[(size_t)R * NW] != starts[(size_t)R * NW + (k_hi - k_lo)];
                const unsigned long long mask = __ballot(has);
                const int at = n + __popcll(mask & ((1ull << lane) - 1ull)); // (its place among the regions with work)
                if (has && at % rsplit == rsi) rlist[at / rsplit] = R;
                n += __popcll(mask);
            }
            if (lane == 0) rcount = n > rsi ? (n - rsi + rsplit - 1) / rsplit : 0;
        }
        __syncthreads();
        const int nlist = rcount;
        {
            // what this block's patches read of a listed region: rows rmin .. rmax + ny - 1, bytes (xmin & ~3) .. xmax + 15 (+ 13 for
            // the second half of a lattice row)
            const uint32_t *rb = a.rbox + (size_t)st.qslot * a.rbox_stride;
            const uint32_t reach = (uint32_t)(nx > YM_RG_G ? 15 + YM_RG_G : 15);
            for (int i = tid; i < nlist; i += NT) {
                const uint32_t v = rb[(size_t)rlist[i] * a.parts + p];
                const uint32_t r0 = v & 0xffu, r1 = min((uint32_t)(YM_RG_ROWS - 1), ((v >> 8) & 0xffu) + (uint32_t)ny - 1u);
                const uint32_t s0 = ((v >> 16) & 0xfcu) >> 4, s1 = min((uint32_t)(YM_RG_SEGS - 1), ((v >> 24) + reach) >> 4);
                rboxl[i] = r0 | r1 << 8 | s0 << 16 | s1 << 24;
            }
        }
        if (kvalid)
            for (int i = lane; i < nlist; i += 64) {
                const int32_t *srow = starts + (size_t)rlist[i] * NW + wave;
                seginfo[wave][i][0] = (unsigned short)srow[0]; seginfo[wave][i][1] = (unsigned short)srow[1];
            }
        __syncthreads();
        // Copy tasks.  A thread owns one 16-byte segment `seg` of the rows r0, r0 + RSTEP, ... of ONE class image: its source
        // and LDS offsets are a base plus a constant stride per task (two registers instead of one per task), and whether a
        // task lies inside the box a region's patches read (`bx`: first row | last row << 8 | first segment << 16 | last << 24,
        // wave-uniform, rboxl) is one segment test per region and two row compares per task.
        // Nothing is range-checked against the window: rows past it and blocks past a plane row are other bytes of the planes
        // buffer (the host allocates YM_RG_PLANES_SLACK bytes past the last item), and no patch the window holds reads them.
        // (WIN: a thread owns one 16-byte segment of a WINDOW row -- eight class bytes of each column parity -- of one row parity:
        //  `cls` = the even-column class of its row parity, the odd-column class is the next image; twice the segments per row,
        //  half the threads per row parity: the same rows at once, the same number of tasks)
        constexpr int TPC = WIN ? NT / 2 : NT / 4;        // threads per class image (WIN: per row parity)
        constexpr int NSEG = WIN ? 2 * YM_RG_SEGS : YM_RG_SEGS;
        constexpr int LPS = TPC / NSEG;                   // rows the threads of a class cover at once
        constexpr int RSTEP = LPS;
        static_assert(LPS >= 1 && LPS * PER >= YM_RG_ROWS, "the copy tasks must cover a class image");
        const uint32_t cls = WIN ? 2u * ((uint32_t)tid / TPC) : (uint32_t)tid / TPC, j = (uint32_t)tid % TPC;
        const uint32_t seg = j % NSEG, r0 = j / NSEG;
        const bool copier = j < (uint32_t)(LPS * NSEG);
        const uint32_t src0 = WIN ? (2u * r0 + (cls >> 1)) * (uint32_t)a.g.pitch + 16u * seg
                                  : (cls & 1u) * (uint32_t)plane_bytes + (2u * r0 + (cls >> 1)) * (uint32_t)half_pitch + 16u * seg;
        const uint32_t src_step = WIN ? 2u * RSTEP * (uint32_t)a.g.pitch : 2u * RSTEP * (uint32_t)half_pitch;
        const uint32_t dst0 = (cls * YM_RG_ROWS + r0) * YM_RG_PITCH + (WIN ? 8u : 16u) * seg;
        uint4 v[PER];
        auto in_box = [&](int q, uint32_t bx, bool seg_in) {
            (void)q; (void)bx;
            return seg_in; // (rows: whole bands, see band_in -- a per-lane row test costs more vector instructions than the rows it saves)
        };
        // task q of every thread is one of the rows q * RSTEP .. + RSTEP - 1: a band the box does not reach is skipped by the
        // whole block (a scalar branch: no load or store instruction is issued for it)
        auto band_in = [&](int q, uint32_t bx) {
            return (uint32_t)(q * RSTEP) <= ((bx >> 8) & 0xffu) && (uint32_t)(q * RSTEP + RSTEP - 1) >= (bx & 0xffu);
        };
        const uint8_t *__restrict__ window = a.grid + (size_t)b * a.grid_stride;
        // (the box names 16-byte segments of a class row; a window segment holds eight class bytes of each parity)
        auto seg_inside = [&](uint32_t bx) {
            return WIN ? copier && seg >= 2u * ((bx >> 16) & 0xffu) && seg <= 2u * (bx >> 24) + 1u
                       : copier && seg >= ((bx >> 16) & 0xffu) && seg <= (bx >> 24);
        };
        auto stage_load = [&](int R, uint32_t bx) {
            const int RX = R % a.nrx, RY = R / a.nrx;
            const uint8_t *src = WIN ? window + ((size_t)(2 * RY * YM_RG_H) * a.g.pitch + (size_t)RX * (2 * YM_RG_W))
                                     : planes + ((size_t)(2 * RY * YM_RG_H) * half_pitch + (size_t)RX * YM_RG_W); // (wave-uniform)
            const bool seg_in = seg_inside(bx);
            // (a task outside the box loads the region's first bytes -- one line for all of them -- and stores nothing; a load
            //  under a per-lane predicate costs the kernel 28 bytes of scratch per lane and is slower)
#pragma unroll
            for (int q = 0; q < PER; q++)
#if YM_RG_ABLATE & 2
                v[q] = make_uint4((uint32_t)(size_t)src, seg_in, bx, q);
#else
                if (band_in(q, bx)) v[q] = *reinterpret_cast<const uint4 *>(src + (in_box(q, bx, seg_in) ? src0 + (uint32_t)q * src_step : 0u));
#endif
        };
        auto stage_store = [&](uint32_t bx) {
            const bool seg_in = seg_inside(bx);
#pragma unroll
            for (int q = 0; q < PER; q++) {
                uint32_t *d = reinterpret_cast<uint32_t *>(region + dst0 + (uint32_t)(q * RSTEP * YM_RG_PITCH)); // class images are contiguous
                if (band_in(q, bx) && in_box(q, bx, seg_in) && (!(YM_RG_ABLATE & 4) || v[q].x == 0x12345u)) {
                    if (WIN) { // even window columns -> this image, odd ones -> the next
                        d[0] = __builtin_amdgcn_perm(v[q].y, v[q].x, 0x06040200u); d[1] = __builtin_amdgcn_perm(v[q].w, v[q].z, 0x06040200u);
                        d[YM_RG_CLS / 4] = __builtin_amdgcn_perm(v[q].y, v[q].x, 0x07050301u); d[YM_RG_CLS / 4 + 1] = __builtin_amdgcn_perm(v[q].w, v[q].z, 0x07050301u);
                    } else { d[0] = v[q].x; d[1] = v[q].y; d[2] = v[q].z; d[3] = v[q].w; }
                }
            }
        };
        // this wave's entries of a region: [t0, t2) (a multiple of four entries)
        auto segment = [&](int ri, int &t0, int &t2) {
            t0 = t2 = 0;
            if (kvalid) {
                t0 = __builtin_amdgcn_readfirstlane((int)seginfo[wave][ri][0]);
                t2 = __builtin_amdgcn_readfirstlane((int)seginfo[wave][ri][1]);
            }
        };
        // The first 256 entries of a wave's segment travel like the region itself: loaded into a register pair while the previous
        // region is gathered (ev), handed over between the barriers (cev), read from there lane by lane (v_readlane).
        // No vector-memory wait inside the gather: that would also wait for the staging loads in flight.
        const uint2 *__restrict__ entries4 = reinterpret_cast<const uint2 *>(entries); // four entries per element
        uint2 ev = make_uint2(0u, 0u), cev = make_uint2(0u, 0u);
        auto entries_load = [&](int t0, int t2) {
            ev = make_uint2(0u, 0u);
            if (t0 + 4 * lane < t2) ev = entries4[(t0 >> 2) + lane];
        };
        auto gather = [&](int lo, int hi) { // entries [lo, hi)
            const int lds_hi = min(hi, lo + 256);
#if YM_RG_PRIO
            // a round lasts as long as its longest gather: the wave with many patches goes first at its SIMD's issue port
            // (s_setprio 0 .. 3 by the number of patches; YM_RG_PRIO = the count that makes a level)
            {
                const int n = hi - lo;
                if (n >= 3 * YM_RG_PRIO) __builtin_amdgcn_s_setprio(3);
                else if (n >= 2 * YM_RG_PRIO) __builtin_amdgcn_s_setprio(2);
                else if (n >= YM_RG_PRIO) __builtin_amdgcn_s_setprio(1);
            }
#endif
            if (lo < lds_hi) {
                const int n4 = (lds_hi - lo) >> 2;
                for (int c = 0; c < n4; c++) { // (c is wave-uniform: the lane select of v_readlane is a scalar register)
                    rg_gather4(acc, lane_off, make_uint2(__builtin_amdgcn_readlane(cev.x, c), __builtin_amdgcn_readlane(cev.y, c)));
                    in_set += 4;
                    if (in_set == YM_RG_FLUSH) flush();
                }
            }
            for (int c = lds_hi; c < hi; c += 4) { // (a very long segment)
                rg_gather4(acc, lane_off, entries4[c >> 2]);
                in_set += 4;
                if (in_set == YM_RG_FLUSH) flush();
            }
#if YM_RG_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };
        int s0 = 0, s2 = 0;
        if (nlist > 0) {
            const uint32_t bx = __builtin_amdgcn_readfirstlane(rboxl[0]);
            segment(0, s0, s2);
            entries_load(s0, s2);
            stage_load(rlist[0], bx);
            stage_store(bx);
            cev = ev;
        }
        __syncthreads();
        YM_RG_PH(0);
        for (int ri = 0; ri < nlist; ri++) {
            // two-stage pipeline: the global loads of the next region are in flight (registers) while this one is gathered
            const bool has_next = ri + 1 < nlist;
            int n0 = 0, n2 = 0;
            uint32_t nbx = 0u;
            if (has_next) {
                nbx = __builtin_amdgcn_readfirstlane(rboxl[ri + 1]);
                segment(ri + 1, n0, n2);
                YM_RG_PH(7);
                entries_load(n0, n2);
                YM_RG_PH(8);
                stage_load(rlist[ri + 1], nbx);
            }
            YM_RG_PH(1);
            gather(s0, s2);
            YM_RG_PH(2);
            __syncthreads(); // every wave is done with region ri
            YM_RG_PH(3);
            if (has_next) {
                stage_store(nbx);
                cev = ev;
            }
            s0 = n0; s2 = n2;
            YM_RG_PH(4);
            __syncthreads();
            YM_RG_PH(5);
        }
    } else if (kvalid && job && rsi == 0) {
        // hypothesis cells are not an exact lattice (possible only through fp rounding): per-cell path over the window
        const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
        const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
        const int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
        const int32_t *cy = cx + a.dim_stride;
        const double2 cs = a.ctrig[(size_t)b * a.nt_stride + k];
        const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
        const int nq = st.nq;
        const double add_x = a.g.semantics == 1 ? st.ylat[0] : st.off_x, add_y = a.g.semantics == 1 ? st.ylat[1] : st.off_y; // (ym_k_common.hpp, lookup_cell_sem)
        for (int g = 0; g < ng; g++) { // set g = the beams [g * FLUSH, (g + 1) * FLUSH)
            for (int j = 0; j < YM_RG_G; j++) {
                const int ix = half * YM_RG_G + j;
                if (ix >= nx) break;
                const int base = cy[row] * lin_pitch(a.g) + cx[ix];
                unsigned sum = 0;
                const int i1 = min(nq, (g + 1) * YM_RG_FLUSH);
                for (int i = g * YM_RG_FLUSH; i < i1; i++)
                    sum += cell_value(a.g, grid, limit, (unsigned)(base + lookup_offset_sem(a.g, ql[i], cs.x, cs.y, st.off_x, st.off_y, add_x, add_y, lin_pitch(a.g))));
                acc[2 * (j >> 2) + (j & 1)] += sum << (16 * ((j >> 1) & 1));
            }
            flush();
        }
    }
#if !YM_RG_PROF
    YM_STAMP(a, 9);
#endif
#if YM_RG_PROF
    auto prof_out = [&]() {
        if (a.stamps && lane == 0) {
            for (int i = 0; i < 4; i++) atomicAdd(a.stamps + 10 + i, (unsigned long long)ph[i]);
            atomicAdd(a.stamps + 8, (unsigned long long)(pt - pt_begin)); // the wave's whole life
            atomicAdd(a.stamps + 9, __builtin_amdgcn_s_memrealtime() - rt_begin); // ... in 100 MHz ticks
        }
    };
#endif
    if (!a.fuse_score) {
        if (!kvalid) return;
        while (flushed < ng) flush(); // the set being filled, then empty ones
        return;
    }
    // ---- score (score_kernel's arithmetic, statement for statement): this wave holds the last set of its angle's sums in
    // registers and wrote the earlier ones itself; response, penalty, block maxima; the per-(x, y) maximum over theta goes
    // through LDS (the region buffer is free now) so that only one atomic per cell and block reaches memory
    unsigned long long *pmax = reinterpret_cast<unsigned long long *>(region); // [ny * nx] fp64 bit patterns, >= 0
    const int nxy = nx * ny;
    double *dpen = reinterpret_cast<double *>(region) + ((nxy + 1) & ~1); // [ny * nx] distance penalty of every cell: once
                                                                           // per block, not once per wave (an fp64 division)
    __syncthreads(); // every wave has left the region walk
    for (int i = tid; i < nxy; i += NT) {
        pmax[i] = 0ull;
        const int iy = i / nx, ix = i - iy * nx;
        const double x = -a.lat.off_x + ix * a.lat.step_x, y = -a.lat.off_y + iy * a.lat.step_y;
        dpen[i] = dist_penalty(a.g, x * x + y * y);
    }
    __syncthreads();
    if (kvalid) {
        unsigned tot[YM_RG_G];
        rg_odd(acc);
#pragma unroll
        for (int j = 0; j < YM_RG_G; j++) tot[j] = (acc[2 * (j >> 2) + (j & 1)] >> (16 * ((j >> 1) & 1))) & 0xffffu;
        for (int f = 0; f < min(flushed, ng); f++) { // the sets this lane wrote out earlier
            const uint16_t *pp = a.partial + (size_t)b * a.partial_stride + (((size_t)f * nt + k) * 64 + lane) * 16;
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
        // block maxima of this angle: blockmax[k * ncb + cb]
        for (int cb = 0; cb < ncb; cb++) {
            const double mine = !job ? -1.0 : cb == cb0 ? bmax0 : cb == cb0 + 1 ? bmax1 : -1.0;
            const double m = wave_reduce(mine, OpMaxD());
            if (lane == 0) a.blockmax[(size_t)b * a.n_blocks + (size_t)k * ncb + cb] = m;
        }
    }
    __syncthreads();
    for (int i = tid; i < nxy; i += NT)
        if (pmax[i]) atomicMax(reinterpret_cast<unsigned long long *>(a.probs) + (size_t)b * a.probs_stride + i, pmax[i]);
#if YM_RG_PROF
    YM_RG_PH(6);
    prof_out();
#endif
}

// (the forms that lost to correlate_region_kernel -- wave-specialised, one block per item, pooled, sixteen waves per block -- and the
//  round-5 list builder they read live in scripts/exp/forms/: profiles/r04_region_study.md, r05_region_study.md; `make experimental`)

} // namespace ym
