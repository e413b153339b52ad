// ym_k_gather.hpp -- K4g: the coarse correlate of BATCHES on any lattice up to 48 x 64, gathered from LDS region by region.
// Part of ym_kernels.hpp (include that, not this file).
//
// The hypotheses of one beam are every other cell of every other row of the window, i.e. a DENSE nx x ny block of bytes
// (a "patch") in the image of one (column parity, row parity) CLASS of window cells; sum(hypothesis) = the sum of the
// patches of all beams, byte by byte.  The (beam, angle) pairs of a QUERY depend on the query alone (its points, pose
// and lattice), not on the chain it is matched against, so one list serves every item of a batch:
//   gbin_count / gbin_scan / gbin_place   once per distinct query of a call: the (beam, angle) pairs, runs of beams in one
//                         cell merged into one patch with a multiplicity (coarse grids), sorted by the REGION of class
//                         space the patch starts in, the angle, the patch's byte alignment and its multiplicity; packed as
//                         32-bit UNITS of two patches of equal alignment and multiplicity (the odd patch of a class is a
//                         unit of its own, listed behind the pairs); a work list of regions.
//   gather_kernel         a block of an item (eight waves, one coarse angle each; ceil(nt / 8) blocks per item) walks the work
//                         list: the loads of region i + 1 (ONE class image + the region's units + its row of the bin
//                         table) are in flight in registers while region i is gathered from LDS.  A wave keeps its angle's
//                         sums in registers; a lane owns 16 x-adjacent hypotheses of one lattice row and reads their 20 bytes
//                         as three ds_read_b64 (LDS row pitch = 8 x odd: conflict-free at 256 B/clk); the raw dwords of a
//                         unit's two patches are added as packed bytes (<= 200), byte-aligned by one v_alignbyte per dword
//                         and widened into 16-bit lanes (even bytes: v_and + v_add; odd bytes: the sum of x >> 8, separated
//                         when the sums leave the registers), times the multiplicity where there is one.  Units are decoded
//                         by vector instructions on wave-uniform registers: the scalar ALU issues no faster than the vector
//                         ALU (profiles/r03_issue_rates.md) and the loop keeps it for loop control.  The block then scores
//                         its sums itself.
// On the default 26 x 26 lattice the round-2 form (ym_k_region.hpp: 13 hypotheses per lane, four dwords per patch) needs a
// third fewer vector instructions and keeps that domain; this file serves every other batch lattice up to 48 x 64, merged
// offsets and scans of any length.  What was measured on the way: profiles/r03_gather_sweep.md, r03_lds_dma_experiment.md.
#pragma once

namespace ym {

#define YM_GA_G 16          // hypotheses per lane
#define YM_GA_MAX_SEG 3     // lanes per lattice row: nx <= 48
#define YM_GA_MAX_NP 3      // waves per angle: ny * ceil(nx / 16) <= 192 lane jobs
#define YM_GA_CLS 16        // (byte shift 0..3) x (multiplicity 1..4) classes inside a bin
#define YM_GA_FLUSH 652     // weight (patches x multiplicity) a set of 16-bit sums holds: 652 x 100 < 65536
#define YM_GA_MULT 0x40000000 // flag on a run's first-unit entry: some unit of the run has a multiplicity above 1
#define YM_GBIN_THREADS 256
#ifndef YM_GA_OCC512
#define YM_GA_OCC512 6 // waves per SIMD blocks of up to 512 threads are compiled for (6: three blocks of eight waves per CU)
#endif
#ifndef YM_GA_PER
#define YM_GA_PER 4          // 16-byte chunks of a class image a thread copies per work item
#endif
// bytes the host keeps past the last item's planes: a staged region may start up to 2 * (H + ny) + 3 rows and P bytes past
// the last cell of the second plane (never gathered, but read)
#define YM_GA_PLANES_SLACK(half_pitch, H, ny, P) ((size_t)(2 * ((H) + (ny)) + 4) * (size_t)(half_pitch) + (size_t)(P) + 256)

struct GatherArgs {
    YmGeom g;
    YmLattice lat;
    const uint8_t *grid;
    const uint8_t *planes;
    size_t grid_stride;
    const double2 *ctrig;    // [B][nt_stride]
    const int32_t *hypcell;  // [B][2][dim_stride]
    const YmItemState *states;
    const int32_t *qrep;     // [n_qslots] an item that uses the query slot
    // the lists of every query slot of the call
    uint32_t *units;         // [Q][units_stride] low 16: (LDS offset of patch A & ~3) | byte shift, high 16: (offset of patch B & ~3) |
                             // multiplicity - 1 (a single: no B)
    size_t units_stride;
    int32_t *starts;         // [Q][starts_stride] first unit of run = ((region * nt + angle) * 2 + (offset bit 2)) * 2 + (0: pairs, 1:
                             // singles), | YM_GA_MULT if the run holds a multiplicity above 1; [4 * nreg * nt] = total or -1
    size_t starts_stride;
    int32_t *work;           // [Q][parts][work_stride]: count, then (region, first unit, end) triples
    size_t work_stride;
    uint32_t *counters;      // [Q][4][nbins2 * 16] class sizes (zeroed by the host), then gbin_scan's: write cursors, first pair unit, single unit
    const uint32_t *lane_job; // [NP * 64] row | seg << 8 | valid << 16 of every lane of a part's wave
    uint16_t *partial;       // [B][ng][njobs][64][16] sets of 16-bit sums a wave had to write out
    size_t partial_stride;
    int32_t nt_stride, dim_stride;
    int32_t W, H, P, rows;   // region size in class bytes, LDS row pitch, staged rows (H + ny)
    int32_t nrx, nry, nseg, NP, ng, parts, kpp; // kpp: angles per part
    int32_t unit_cap;        // units per LDS buffer
    int32_t force_irregular; // tests: 1 = the per-cell path, 2 = "the lists do not fit"
    uint32_t *sums;          // [B][nt][ny][nx] integer sums, or null
    double *resp;            // [B][nt][ny][nx]
    size_t sums_stride;
    double *blockmax;        // [B][n_blocks], block = (angle, YM_SCORE_THREADS cells)
    double *probs;           // [B][ny*nx] max over theta per (x, y); zeroed by the prepare stage
    size_t probs_stride;
    int32_t n_blocks, pad;
    unsigned long long *stamps;
};

// region and LDS offset of the patch of one (beam, angle) pair; false if the origin is outside the regions (never for a
// patch the window holds; kept so that nothing is ever written out of bounds)
__device__ __forceinline__ bool ga_entry(const GatherArgs &a, int2 cell, int cx0, int cy0, int &R, unsigned &e) {
    const int X = cx0 + cell.x, Y = cy0 + cell.y;
    const int xc = X >> 1, yc = Y >> 1;
    const int rx = xc / a.W, ry = yc / a.H;
    if (X < 0 || Y < 0 || rx >= a.nrx || ry >= a.nry) return false;
    const int cls = (X & 1) | ((Y & 1) << 1);
    R = (cls * a.nry + ry) * a.nrx + rx;
    e = (unsigned)((yc - ry * a.H) * a.P + (xc - rx * a.W));
    return true;
}

// One thread per (angle, beam): a beam whose cell differs from its predecessor's heads a run; the run's length is the
// patch's multiplicity (sum_i G[c + o_i] = sum_runs len * G[c + o_run]: integer, exact), cut into pieces of at most 4.
// PLACE = false counts the pieces per (bin2, class); PLACE = true (after gbin_scan) writes them into their units.
// GridIndexLookup::ComputeOffsets for every coarse angle happens here.   grid (ceil(nt * max_n / 256), Q)
template <bool PLACE>
__global__ __launch_bounds__(YM_GBIN_THREADS) void gbin_pieces_kernel(GatherArgs a) {
    const int q = blockIdx.y;
    const int b = a.qrep[q];
    const YmItemState &st = a.states[b];
    const int nq = st.nq, nt = a.lat.nt;
    const int p = blockIdx.x * YM_GBIN_THREADS + threadIdx.x;
    if (p >= nq * nt) return;
    const int nbins2 = a.nrx * a.nry * 4 * nt * 2;
    const int32_t *starts = a.starts + (size_t)q * a.starts_stride;
    if (PLACE && starts[2 * nbins2] < 0) return; // the lists do not fit: every item of this query takes the per-cell path
    const int k = p / nq, i = p - k * nq;
    const int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
    const int cx0 = cx[0], cy0 = cx[a.dim_stride];
    const double off_x = st.off_x, off_y = st.off_y;
    const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
    const double2 cs = a.ctrig[(size_t)b * a.nt_stride + k];
    const bool yag = a.g.semantics == 1; // ("yagpy" items proven regular: ym_k_common.hpp, lookup_cell_sem)
    const double add_x = yag ? st.ylat[0] : off_x, add_y = yag ? st.ylat[1] : off_y;
    const int2 c = lookup_cell_sem(a.g, ql[i], cs.x, cs.y, off_x, off_y, add_x, add_y);
    if (i > 0) {
        const int2 pc = lookup_cell_sem(a.g, ql[i - 1], cs.x, cs.y, off_x, off_y, add_x, add_y);
        if (pc.x == c.x && pc.y == c.y) return; // inside a run
    }
    int m = 1;
    for (int j = i + 1; j < nq; j++) {
        const int2 nc = lookup_cell_sem(a.g, ql[j], cs.x, cs.y, off_x, off_y, add_x, add_y);
        if (nc.x != c.x || nc.y != c.y) break;
        m++;
    }
    int R;
    unsigned e;
    if (!ga_entry(a, c, cx0, cy0, R, e)) return;
    const unsigned bin2 = (unsigned)((R * nt + k) * 2) + ((e >> 2) & 1u);
    const size_t ncls = (size_t)nbins2 * YM_GA_CLS, at = (size_t)bin2 * YM_GA_CLS + (e & 3u) * 4u;
    uint32_t *cnt = a.counters + (size_t)q * 4 * ncls; // [0]: sizes, [1]: cursors, [2]: first pair unit, [3]: single unit
    uint32_t *units = a.units + (size_t)q * a.units_stride;
    uint16_t *half = reinterpret_cast<uint16_t *>(units);
    while (m > 0) {
        const int mp = m < 4 ? m : 4;
        if (!PLACE) atomicAdd(&cnt[at + mp - 1], 1u);
        else {
            const unsigned pos = atomicAdd(&cnt[ncls + at + mp - 1], 1u), n = cnt[at + mp - 1];
            if (pos < (n & ~1u)) { // half of a pair: patch A carries the byte shift, patch B the multiplicity
                const unsigned u = cnt[2 * ncls + at + mp - 1] + (pos >> 1);
                half[2 * u + (pos & 1u)] = (pos & 1u) ? (uint16_t)((e & ~3u) | (unsigned)(mp - 1)) : (uint16_t)e;
            } else {
                units[cnt[3 * ncls + at + mp - 1]] = e | (unsigned)(mp - 1) << 16;
            }
        }
        m -= mp;
    }
}

// One block per query slot: the classes' sizes -> the runs of every bin2 (its pairs, then its singles), the classes' first
// units, and per part the work list of region chunks.  If anything does not fit, starts[2 * nbins2] = -1 and the items of
// this query are scored by gather_percell_kernel.
__global__ __launch_bounds__(1024) void gbin_scan_kernel(GatherArgs a) {
    __shared__ int wave_tot[16];
    __shared__ int running;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int nt = a.lat.nt, nreg = a.nrx * a.nry * 4;
    const int nbins2 = nreg * nt * 2;
    const size_t ncls = (size_t)nbins2 * YM_GA_CLS;
    uint32_t *cnt = a.counters + (size_t)q * 4 * ncls;
    int32_t *starts = a.starts + (size_t)q * a.starts_stride;
    // thread t owns the bins [t * per, (t + 1) * per)
    const int per = (nbins2 + 1023) / 1024;
    const int first = tid * per;
    int local = 0; // units
    for (int j = 0; j < per && first + j < nbins2; j++)
        for (int c = 0; c < YM_GA_CLS; c++) local += (int)((cnt[(size_t)(first + j) * YM_GA_CLS + c] + 1u) >> 1);
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) wave_tot[tid >> 6] = incl;
    __syncthreads();
    int base = 0, all = 0;
    for (int w = 0; w < 16; w++) {
        if (w < (tid >> 6)) base += wave_tot[w];
        all += wave_tot[w];
    }
    const bool fits = all <= (int)a.units_stride - 64 && a.force_irregular != 2;
    int run = base + incl - local;
    for (int j = 0; j < per && first + j < nbins2; j++) {
        const size_t at = (size_t)(first + j) * YM_GA_CLS;
        int pairs = 0, mult = 0;
        for (int c = 0; c < YM_GA_CLS; c++) {
            const int n = (int)cnt[at + c];
            pairs += n >> 1;
            if (n && (c & 3)) mult = YM_GA_MULT;
        }
        starts[2 * (first + j)] = run | mult;
        starts[2 * (first + j) + 1] = (run + pairs) | mult;
        int pu = run, su = run + pairs;
        for (int c = 0; c < YM_GA_CLS; c++) {
            const int n = (int)cnt[at + c];
            cnt[ncls + at + c] = 0u;
            cnt[2 * ncls + at + c] = (uint32_t)pu;
            cnt[3 * ncls + at + c] = (uint32_t)su;
            pu += n >> 1;
            su += n & 1;
        }
        run = su;
    }
    if (tid == 0) running = 0;
    __syncthreads(); // (starts[] of every run is written: the work lists read them)
    bool ok = fits;
    const int max_work = ((int)a.work_stride - 1) / 3;
    for (int part = 0; part < a.parts; part++) {
        int32_t *work = a.work + ((size_t)q * a.parts + part) * a.work_stride;
        const int k_lo = part * a.kpp, k_hi = min(nt, k_lo + a.kpp);
        if (tid == 0) running = 0;
        __syncthreads();
        for (int R0 = 0; R0 < nreg; R0 += 1024) {
            const int R = R0 + tid;
            int u0 = 0, u1 = 0;
            if (R < nreg && k_lo < k_hi) {
                u0 = starts[(R * nt + k_lo) * 4] & (YM_GA_MULT - 1);
                u1 = (R * nt + k_hi) * 2 < nbins2 ? starts[(R * nt + k_hi) * 4] & (YM_GA_MULT - 1) : all;
            }
            const int n = (u1 - u0 + a.unit_cap - 1) / a.unit_cap;
            int inc = n;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int v = __shfl_up(inc, d);
                if (lane >= d) inc += v;
            }
            __syncthreads();
            if (lane == 63) wave_tot[tid >> 6] = inc;
            __syncthreads();
            int wbase = running;
            for (int w = 0; w < (tid >> 6); w++) wbase += wave_tot[w];
            int at = wbase + inc - n;
            for (int c = 0; c < n; c++, at++)
                if (at < max_work) {
                    work[1 + 3 * at] = R;
                    work[2 + 3 * at] = u0 + c * a.unit_cap;
                    work[3 + 3 * at] = min(u1, u0 + (c + 1) * a.unit_cap);
                }
            __syncthreads();
            if (tid == 1023) running = wbase + inc;
            __syncthreads();
        }
        if (running > max_work) ok = false;
        if (tid == 0) work[0] = min(running, max_work);
        __syncthreads();
    }
    if (tid == 0) {
        starts[2 * nbins2] = ok ? all : -1;
        if (a.stamps && q == 0) a.stamps[26] = (unsigned long long)all;
    }
}

typedef unsigned int ga_u32x2 __attribute__((ext_vector_type(2)));

// The 20 bytes at LDS address `ad` (a multiple of 4) are read as three 8-byte-aligned ds_read_b64 (256 B/clk; a ds_read_b32
// would meet 2-way bank conflicts on a pitch of 8 x odd): Q = bit 2 of the address (the same in every lane: the lanes'
// offsets are multiples of 8) says whether the first wanted dword is the low (Q = 0) or the high one (Q = 1) of its pair.
#define YM_GA_READS(p, q, r, ad) "ds_read_b64 " p ", " ad "\n\tds_read_b64 " q ", " ad " offset:8\n\tds_read_b64 " r ", " ad " offset:16\n\t"

template <int Q>
__device__ __forceinline__ void ga_dwords(const ga_u32x2 &p, const ga_u32x2 &q, const ga_u32x2 &r, uint32_t (&d)[5]) {
    if (Q == 0) { d[0] = p.x; d[1] = p.y; d[2] = q.x; d[3] = q.y; d[4] = r.x; }
    else { d[0] = p.y; d[1] = q.x; d[2] = q.y; d[3] = r.x; d[4] = r.y; }
}

// the sums of one unit into the wave's registers: E[j] holds hypotheses 4j (low 16 bits) and 4j + 2, S[j] hypotheses 4j + 1 and
// 4j + 3.  Round 5: funnel and widening are ONE v_perm_b32 per pair of hypotheses, as in the region correlate since round 4 --
// selector (shift, 0x0c, shift + 2, 0x0c) picks bytes shift and shift + 2 of the eight bytes s[j + 1] : s[j] into the low bytes of
// the two 16-bit fields and zeroes the rest, (shift + 1, 0x0c, shift + 3, 0x0c) the odd ones: four instructions per dword
// (2 v_perm + 2 v_add, or 2 v_mad_u32_u24 with a multiplicity) instead of five (v_alignbyte, v_and, v_add, v_lshrrev, v_add), and
// the odd sums are clean fields (until then S held the running sum of dword >> 8, separated when the sums left the registers).
template <bool MULT>
__device__ __forceinline__ void ga_accumulate(uint32_t (&E)[4], uint32_t (&S)[4], const uint32_t (&s)[5], uint32_t shift, uint32_t mp) {
    const uint32_t selE = shift * 0x00010001u + 0x0c020c00u, selO = selE + 0x00010001u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t e = __builtin_amdgcn_perm(s[j + 1], s[j], selE), o = __builtin_amdgcn_perm(s[j + 1], s[j], selO);
        if (MULT) {
            E[j] = __umul24(e, mp) + E[j]; // (v_mad_u32_u24: a field is at most 200 << 16 | 200 < 2^24, times at most 4)
            S[j] = __umul24(o, mp) + S[j];
        } else {
            E[j] += e;
            S[j] += o;
        }
    }
}
__device__ __forceinline__ uint32_t ga_odd(uint32_t, uint32_t S) { return S; } // sum(4j + 1) | sum(4j + 3) << 16 (the identity since round 5)

// Two pair units (four patches) of one alignment half Q; u0 / u1 = the units' records, the same in every lane (decoded by
// vector instructions).  A unit's six reads are issued and waited for, then accumulated, then the next unit's (issuing all
// twelve at once and accumulating the first unit under the second one's reads was measured again at the end of round 3:
// 4573 against 4585 us per 4096 loop-lattice items, i.e. nothing -- the other waves of the SIMD hide the LDS latency --,
// for 12 more registers).
// The registers an asm statement that only ISSUES a read names as outputs are not written when the statement ends, and
// the compiler is free to copy them right there: every such register is either waited for inside the issuing statement
// or passes through the statement that waits for it ("+v") before anything else touches it.
template <int Q, bool MULT>
__device__ __forceinline__ void ga_gather2(uint32_t (&E)[4], uint32_t (&S)[4], uint32_t lane_off, uint32_t u0, uint32_t u1) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t u = h ? u1 : u0;
        // (the 8-byte-aligned address at or below the patch's first dword)
        const uint32_t adA = lane_off + (u & 0xfff8u), adB = lane_off + ((u >> 16) & 0xfff8u);
        ga_u32x2 pa, qa, ra, pb, qb, rb;
        asm volatile(YM_GA_READS("%0", "%1", "%2", "%6") YM_GA_READS("%3", "%4", "%5", "%7") "s_waitcnt lgkmcnt(0)"
                     : "=&v"(pa), "=&v"(qa), "=&v"(ra), "=&v"(pb), "=&v"(qb), "=&v"(rb) : "v"(adA), "v"(adB) : "memory");
        uint32_t da[5], db[5];
        ga_dwords<Q>(pa, qa, ra, da);
        ga_dwords<Q>(pb, qb, rb, db);
#pragma unroll
        for (int j = 0; j < 5; j++) da[j] += db[j]; // packed bytes, each <= 200
        ga_accumulate<MULT>(E, S, da, u & 3u, ((u >> 16) & 3u) + 1u);
    }
}
// one unit: a pair (PAIR) or a single patch
template <int Q, bool PAIR>
__device__ __forceinline__ void ga_gather1(uint32_t (&E)[4], uint32_t (&S)[4], uint32_t lane_off, uint32_t u0) {
    const uint32_t adA0 = lane_off + (u0 & 0xfff8u), adB0 = lane_off + ((u0 >> 16) & 0xfff8u);
    ga_u32x2 pa0, qa0, ra0, pb0, qb0, rb0;
    uint32_t da[5], db[5];
    if (PAIR) {
        asm volatile(YM_GA_READS("%0", "%1", "%2", "%6") YM_GA_READS("%3", "%4", "%5", "%7") "s_waitcnt lgkmcnt(0)"
                     : "=&v"(pa0), "=&v"(qa0), "=&v"(ra0), "=&v"(pb0), "=&v"(qb0), "=&v"(rb0) : "v"(adA0), "v"(adB0) : "memory");
        ga_dwords<Q>(pa0, qa0, ra0, da);
        ga_dwords<Q>(pb0, qb0, rb0, db);
#pragma unroll
        for (int j = 0; j < 5; j++) da[j] += db[j];
    } else {
        asm volatile(YM_GA_READS("%0", "%1", "%2", "%3") "s_waitcnt lgkmcnt(0)" : "=&v"(pa0), "=&v"(qa0), "=&v"(ra0) : "v"(adA0) : "memory");
        ga_dwords<Q>(pa0, qa0, ra0, da);
    }
    ga_accumulate<true>(E, S, da, u0 & 3u, ((u0 >> 16) & 3u) + 1u);
}

// dynamic LDS of gather_kernel: [image | units (+ 64 spare) | bin row]; `rows` = the staged rows incl. the ones the last copy
// tasks of a block spill into (the host rounds them up to whole tasks)
#define YM_GA_IMAGE_BYTES(P, rows) (((size_t)(P) * (rows) + 255) / 256 * 256)
#define YM_GA_ROW_INTS(kpp) (((4 * (kpp) + 1) + 63) / 64 * 64)
#define YM_GA_LDS_BYTES(P, rows, cap, kpp) (YM_GA_IMAGE_BYTES(P, rows) + ((size_t)(cap) + 64) * 4 + (size_t)YM_GA_ROW_INTS(kpp) * 4 + 64)

// What both forms of the kernel share: the block's place, its angles, the sums and how a set of them leaves the registers.
// grid (parts, B); a block = NWV waves; wave w owns the angles
// k_lo + w * NA .. + NA - 1 of the block's part [k_lo, k_hi); an angle's lattice is spread over NP waves' worth of lanes
// ("lattice parts", a.lane_job), which the wave works through one after the other with the same units.
#define YM_GA_SETUP(NWV_EXPR)                                                                                              \
    const int NWV = (NWV_EXPR);                                                                                            \
    extern __shared__ __attribute__((aligned(16))) unsigned char ga_smem[];                                               \
    int part;                                                                                                              \
    const int b = xcd_item_of_block_2d(part);                                                                              \
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);                         \
    const YmItemState &st = a.states[b];                                                                                   \
    const int q = st.qslot;                                                                                                \
    const int nt = a.lat.nt;                                                                                               \
    const int k_lo = part * a.kpp, k_hi = min(nt, k_lo + a.kpp);                                                           \
    const int nbins2 = a.nrx * a.nry * 4 * nt * 2;                                                                         \
    const int32_t *__restrict__ starts = a.starts + (size_t)q * a.starts_stride;                                          \
    unsigned char *buf0 = ga_smem;                                                                                         \
    int jk[NA];               /* angle (-1: none) */                                                                       \
    uint32_t E[NA][NP][4], S[NA][NP][4];                                                                                   \
    int in_set[NA], flushed[NA];                                                                                           \
    _Pragma("unroll") for (int n = 0; n < NA; n++) {                                                                       \
        const int k = k_lo + wave * NA + n;                                                                                \
        jk[n] = (k < k_hi && wave < NWV) ? k : -1;                                                                         \
        in_set[n] = 0;                                                                                                     \
        flushed[n] = 0;                                                                                                    \
        _Pragma("unroll") for (int p2 = 0; p2 < NP; p2++)                                                                  \
            _Pragma("unroll") for (int j2 = 0; j2 < 4; j2++) E[n][p2][j2] = S[n][p2][j2] = 0u;                            \
    }                                                                                                                      \
    auto flush = [&](int n) {                                                                                              \
        _Pragma("unroll") for (int p2 = 0; p2 < NP; p2++) {                                                                \
            if (flushed[n] < a.ng && jk[n] >= 0) {                                                                         \
                uint32_t acc[8];                                                                                           \
                _Pragma("unroll") for (int j = 0; j < 4; j++) { acc[2 * j] = E[n][p2][j]; acc[2 * j + 1] = ga_odd(E[n][p2][j], S[n][p2][j]); } \
                const size_t job = (size_t)(jk[n] * NP + p2);                                                              \
                store_partial16(a.partial + (size_t)b * a.partial_stride + ((((size_t)flushed[n] * nt * NP) + job) * 64 + lane) * 16, acc); \
            }                                                                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; j++) E[n][p2][j] = S[n][p2][j] = 0u;                                  \
        }                                                                                                                  \
        flushed[n]++;                                                                                                      \
        in_set[n] = 0;                                                                                                     \
    };                                                                                                                     \
    (void)buf0; (void)tid; (void)starts; (void)nbins2; (void)flush; (void)in_set

// ---- score (score_kernel's arithmetic, statement for statement): a wave holds the last set of its angles' sums in
// registers and wrote the earlier ones itself; response, penalty, block maxima; the per-(x, y) maximum over theta and
// the block maxima go through LDS (the buffers are free now) so that only one atomic per cell and block reaches memory.
// Called by the NT working threads of the block.
template <int NA, int NP>
__device__ __forceinline__ void ga_score(const GatherArgs &a, int b, int NT, int k_lo, int k_hi, const int (&jk)[NA], const uint32_t (&E)[NA][NP][4],
                                         const uint32_t (&S)[NA][NP][4], const int (&flushed)[NA], unsigned char *buf0) {
    const int tid = threadIdx.x, lane = tid & 63;
    const YmItemState &st = a.states[b];
    const int nt = a.lat.nt, nx = a.lat.nx, ny = a.lat.ny;
    const int nxy = nx * ny, ncb = (nxy + YM_SCORE_THREADS - 1) / YM_SCORE_THREADS;
    unsigned long long *pmax = reinterpret_cast<unsigned long long *>(buf0);   // [ny * nx] fp64 bit patterns, >= 0
    double *dpen = reinterpret_cast<double *>(buf0) + nxy;                      // [ny * nx] distance penalty of every cell
    unsigned long long *bmax = reinterpret_cast<unsigned long long *>(buf0) + 2 * (size_t)nxy; // [angles of the block][ncb]
    __syncthreads();
    for (int i = tid; i < nxy; i += NT) {
        pmax[i] = 0ull;
        const int iy = i / nx, ix = i - iy * nx;
        const double x = -a.lat.off_x + ix * a.lat.step_x, y = -a.lat.off_y + iy * a.lat.step_y;
        dpen[i] = dist_penalty(a.g, x * x + y * y);
    }
    for (int i = tid; i < (k_hi - k_lo) * ncb; i += NT) bmax[i] = 0ull;
    __syncthreads();
    const double ct = st.center[2];
    const int nq = st.nq;
#pragma unroll
    for (int n = 0; n < NA; n++) {
        if (jk[n] < 0) continue;
        const int k = jk[n];
        const double angle = (ct - a.lat.angle_off) + k * a.lat.angle_res;
#pragma unroll
        for (int p2 = 0; p2 < NP; p2++) {
            const uint32_t lj = a.lane_job[p2 * 64 + lane];
            const bool job = (lj >> 16) != 0;
            const int row = lj & 0xff, seg = (lj >> 8) & 0xff;
            unsigned tot[YM_GA_G];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t e = E[n][p2][j], o = ga_odd(E[n][p2][j], S[n][p2][j]);
                tot[4 * j] = e & 0xffffu; tot[4 * j + 1] = o & 0xffffu; tot[4 * j + 2] = e >> 16; tot[4 * j + 3] = o >> 16;
            }
            for (int f = 0; f < min(flushed[n], a.ng); f++) { // the sets this lane wrote out earlier
                const uint16_t *pp = a.partial + (size_t)b * a.partial_stride + ((((size_t)f * nt * NP) + (size_t)(k * NP + p2)) * 64 + lane) * 16;
#pragma unroll
                for (int j = 0; j < YM_GA_G; j++) tot[j] += pp[j];
            }
            double bm0 = -1.0, bm1 = -1.0; // block maxima this lane contributes to (its 16 cells span at most 2 blocks)
            const int c0 = row * nx + seg * YM_GA_G, cb0 = c0 / YM_SCORE_THREADS;
#pragma unroll
            for (int j = 0; j < YM_GA_G; j++) {
                const int ix = seg * YM_GA_G + j;
                if (job && ix < nx) {
                    const int c = row * nx + ix;
                    const double r = hyp_response_dp(a.g, a.lat.penalize, tot[j], nq, dpen[c], angle, ct);
                    const size_t h = (size_t)k * nxy + c;
                    if (a.sums) a.sums[(size_t)b * a.sums_stride + h] = tot[j];
                    a.resp[(size_t)b * a.sums_stride + h] = r;
                    if (c / YM_SCORE_THREADS == cb0) bm0 = r > bm0 ? r : bm0;
                    else bm1 = r > bm1 ? r : bm1;
                    if (r > 0.0) atomicMax(&pmax[c], (unsigned long long)__double_as_longlong(r));
                }
            }
            if (job) {
                if (bm0 > 0.0) atomicMax(&bmax[(k - k_lo) * ncb + cb0], (unsigned long long)__double_as_longlong(bm0));
                if (bm1 > 0.0) atomicMax(&bmax[(k - k_lo) * ncb + cb0 + 1], (unsigned long long)__double_as_longlong(bm1));
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < nxy; i += NT)
        if (pmax[i]) atomicMax(reinterpret_cast<unsigned long long *>(a.probs) + (size_t)b * a.probs_stride + i, pmax[i]);
    for (int i = tid; i < (k_hi - k_lo) * ncb; i += NT)
        a.blockmax[(size_t)b * a.n_blocks + (size_t)k_lo * ncb + i] = __longlong_as_double((long long)bmax[i]);
}

// The walk over the work items is a two-stage pipeline: while item i is gathered, the global loads of item i + 1 (the
// class image as PER 16-byte chunks per thread, one unit and one bin-row entry per thread) are in flight in registers;
// they go to LDS between the two barriers that end the gather.  (LDS-DMA was measured for this copy and lost: a wave
// that issues global_load_lds is held for hundreds of cycles per instruction, profiles/r03_lds_dma_experiment.md.)
// MAXT: the largest block the instantiation is launched with: blocks of up to eight waves (the usual case: one angle per wave,
// about eight angles per block) are compiled for three blocks per CU (80 VGPRs; round 5 -- until then 104 VGPRs left two).
template <int NA, int NP, int PER, int MAXT = 1024>
__global__ __launch_bounds__(MAXT, MAXT <= 512 ? YM_GA_OCC512 : 4) void gather_kernel(GatherArgs a) {
    YM_GA_SETUP((int)(blockDim.x >> 6));
    if (!(st.regular[0] && a.force_irregular != 1 && starts[2 * nbins2] >= 0)) return; // gather_percell_kernel's item
    const int NT = 64 * NWV;
    const int32_t *__restrict__ work = a.work + ((size_t)q * a.parts + part) * a.work_stride;
    const int half_pitch = a.g.pitch / 2;
    const int plane_bytes = half_pitch * a.g.win_w;
    const uint8_t *__restrict__ planes = a.planes + (size_t)b * a.grid_stride;
    const uint32_t *__restrict__ units = a.units + (size_t)q * a.units_stride;
    // LDS carve-up: [image | units | bin row]
    const size_t image_bytes = YM_GA_IMAGE_BYTES(a.P, a.rows);
    uint32_t *ul = reinterpret_cast<uint32_t *>(buf0 + image_bytes);
    int32_t *row = reinterpret_cast<int32_t *>(buf0 + image_bytes + ((size_t)a.unit_cap + 64) * 4);
    const uint32_t img = (uint32_t)(size_t)buf0;
    const int nwork = work[0];
    uint32_t loff[NP];        // row * P + 16 * seg of the lane in every lattice part
#pragma unroll
    for (int p2 = 0; p2 < NP; p2++) {
        const uint32_t lj = a.lane_job[p2 * 64 + lane];
        loff[p2] = img + (lj & 0xffu) * (uint32_t)a.P + ((lj >> 8) & 0xffu) * YM_GA_G;
    }
    YM_STAMP(a, 8);
    // copy task t = (image row, 16-byte chunk): thread tid takes t = tid, tid + NT, ... (tasks past the image are done, and
    // their bytes land past the image rows, inside the buffer: the host sizes it for PER * NT chunks).  cpr chunks per
    // staged row, pitch P = 16 * cpr + 8 bytes: an odd number of 8-byte words, so 32 consecutive rows are 32 distinct
    // banks for ds_read_b64.  Nothing is range-checked: rows past the window and chunks past a plane row are other bytes
    // of the planes buffer (the host allocates the slack) that no patch the window holds ever reads.
    const int cpr = (a.P - 8) >> 4;
    const int r_first = tid / cpr, c_first = tid - r_first * cpr;       // task tid
    const int r_step = NT / cpr, c_step = NT - r_step * cpr;            // what NT tasks further means
    uint4 v[PER];
    uint32_t vu = 0;
    int32_t vr = 0;
    int staged = -1; // the region whose image the buffer holds
    bool fresh = false;
    auto stage_load = [&](int R, int ub, int ue) {
        // (unconditional loads at clamped addresses: a register that is set by a load OR a move makes the compiler wait for
        // every load in flight before the move -- the image loads issued just before it among them)
        vu = units[ub + min(tid, max(ue - ub - 1, 0))];
        vr = starts[((size_t)R * nt + k_lo) * 4 + min(tid, 4 * (k_hi - k_lo))];
        fresh = staged != R;
        if (fresh) {
            staged = R;
            const int cls = R / (a.nrx * a.nry), rr = R - cls * (a.nrx * a.nry);
            const int RY = rr / a.nrx, RX = rr - RY * a.nrx;
            const uint8_t *src = planes + (size_t)(cls & 1) * plane_bytes + ((size_t)(2 * RY * a.H) + (cls >> 1)) * half_pitch + (size_t)RX * a.W; // (wave-uniform)
            int r = r_first, c = c_first;
#pragma unroll
            for (int i = 0; i < PER; i++) {
                v[i] = *reinterpret_cast<const uint4 *>(__builtin_assume_aligned(src + ((uint32_t)r * 2u * (uint32_t)half_pitch + 16u * (uint32_t)c), 4));
                r += r_step; c += c_step;
                if (c >= cpr) { c -= cpr; r++; }
            }
        }
    };
    auto stage_store = [&]() {
        if (fresh) {
            int r = r_first, c = c_first;
#pragma unroll
            for (int i = 0; i < PER; i++) {
                uint2 *d = reinterpret_cast<uint2 *>(buf0 + ((uint32_t)r * (uint32_t)a.P + 16u * (uint32_t)c));
                d[0] = make_uint2(v[i].x, v[i].y);
                d[1] = make_uint2(v[i].z, v[i].w);
                r += r_step; c += c_step;
                if (c >= cpr) { c -= cpr; r++; }
            }
        }
        if (tid < a.unit_cap) ul[tid] = vu;
        if (tid < 4 * (k_hi - k_lo) + 1) row[tid] = vr;
    };
    // the work list travels through registers too: item i + 2 is loaded behind the loads of item i + 1
    int cR = 0, cub = 0, cue = 0, nR = 0, nub = 0, nue = 0;
    if (nwork > 0) { cR = work[1]; cub = work[2]; cue = work[3]; }
    if (nwork > 1) { nR = work[4]; nub = work[5]; nue = work[6]; }
    cR = __builtin_amdgcn_readfirstlane(cR); cub = __builtin_amdgcn_readfirstlane(cub); cue = __builtin_amdgcn_readfirstlane(cue);
    if (nwork > 0) { stage_load(cR, cub, cue); stage_store(); }
    __syncthreads();
    // a zero the compiler cannot see through: what is added to it stays in a vector register (the unit records are read
    // through a vector-register pointer that moves by vector adds)
    uint32_t vzero = 0u;
    asm volatile("" : "+v"(vzero));
    for (int wi = 0; wi < nwork; wi++) {
        nR = __builtin_amdgcn_readfirstlane(nR); nub = __builtin_amdgcn_readfirstlane(nub); nue = __builtin_amdgcn_readfirstlane(nue);
        const bool has_next = wi + 1 < nwork;
        if (has_next) stage_load(nR, nub, nue); // in flight while this item is gathered
        int fR = 0, fub = 0, fue = 0;
        if (wi + 2 < nwork) { fR = work[1 + 3 * (wi + 2)]; fub = work[2 + 3 * (wi + 2)]; fue = work[3 + 3 * (wi + 2)]; }
        {
            // this wave's angles: per angle four runs of units -- the pairs and the singles with address bit 2 clear, then those
            // with it set -- and this chunk's share of them
            int rb[NA][5]; // run boundaries, relative to the chunk
            bool mult[NA][4];
            {
                int t[NA][5];
#pragma unroll
                for (int n = 0; n < NA; n++) {
                    const int kk = max(jk[n], k_lo) - k_lo;
#pragma unroll
                    for (int j = 0; j < 5; j++) t[n][j] = row[4 * kk + j];
                }
#pragma unroll
                for (int n = 0; n < NA; n++)
#pragma unroll
                    for (int j = 0; j < 5; j++) {
                        const int e = __builtin_amdgcn_readfirstlane(t[n][j]);
                        if (j < 4) mult[n][j] = (e & YM_GA_MULT) != 0;
                        rb[n][j] = jk[n] < 0 ? 0 : min(max(e & (YM_GA_MULT - 1), cub), cue) - cub;
                    }
            }
            // the records of the next two units are read before the current ones are gathered (broadcast reads through a
            // vector-register pointer); across the angles of a wave the units are one run of the list
            int c = rb[0][0];
            uint32_t up = (uint32_t)(size_t)ul + 4u * (uint32_t)c + vzero;
            typedef const __attribute__((address_space(3))) uint32_t *lds_u32;
            uint32_t v0 = ((lds_u32)(uintptr_t)up)[0], v1 = ((lds_u32)(uintptr_t)up)[1];
#pragma unroll
            for (int n = 0; n < NA; n++) {
                if (jk[n] < 0) continue;
                if (c != rb[n][0]) { // (chunked regions, parts)
                    c = rb[n][0];
                    up = (uint32_t)(size_t)ul + 4u * (uint32_t)c + vzero;
                    v0 = ((lds_u32)(uintptr_t)up)[0]; v1 = ((lds_u32)(uintptr_t)up)[1];
                }
                auto run = [&](auto qtag, int j) {
                    constexpr int Q = decltype(qtag)::value;
                    const int pe = rb[n][j + 1], se = rb[n][j + 2];
                    const bool mu = mult[n][j];
                    for (; c + 2 <= pe; c += 2) { // two pair units
                        if (in_set[n] > YM_GA_FLUSH - 16) flush(n);
                        const uint32_t u0 = v0, u1 = v1;
                        v0 = ((lds_u32)(uintptr_t)up)[2]; v1 = ((lds_u32)(uintptr_t)up)[3];
                        up += 8u;
#pragma unroll
                        for (int p2 = 0; p2 < NP; p2++) {
                            if (mu) ga_gather2<Q, true>(E[n][p2], S[n][p2], loff[p2], u0, u1);
                            else ga_gather2<Q, false>(E[n][p2], S[n][p2], loff[p2], u0, u1);
                        }
                        // (weight = patches x multiplicity: exact, so that a wave never fills more sets than the host provides)
                        in_set[n] += mu ? 2 * (int)__builtin_amdgcn_readfirstlane(((u0 >> 16) & 3u) + ((u1 >> 16) & 3u) + 2u) : 4;
                    }
                    for (; c < se; c++) { // the odd pair unit, then the singles
                        if (in_set[n] > YM_GA_FLUSH - 16) flush(n);
                        const uint32_t u0 = v0;
                        v0 = v1; v1 = ((lds_u32)(uintptr_t)up)[2];
                        up += 4u;
#pragma unroll
                        for (int p2 = 0; p2 < NP; p2++) {
                            if (c < pe) ga_gather1<Q, true>(E[n][p2], S[n][p2], loff[p2], u0);
                            else ga_gather1<Q, false>(E[n][p2], S[n][p2], loff[p2], u0);
                        }
                        in_set[n] += (c < pe ? 2 : 1) * (mu ? (int)__builtin_amdgcn_readfirstlane(((u0 >> 16) & 3u) + 1u) : 1);
                    }
                };
                run(std::integral_constant<int, 0>(), 0);
                run(std::integral_constant<int, 1>(), 2);
            }
        }
        __syncthreads(); // every wave is done with item wi
        if (has_next) stage_store();
        cR = nR; cub = nub; cue = nue;
        nR = fR; nub = fub; nue = fue;
        __syncthreads();
    }
    YM_STAMP(a, 9);
    ga_score<NA, NP>(a, b, NT, k_lo, k_hi, jk, E, S, flushed, buf0);
}

// The items gather_kernel leaves alone -- hypothesis cells that are not an exact lattice (possible only through fp
// rounding), or lists that did not fit -- cell by cell over the window, in the same lanes and with the same scoring.
template <int NA, int NP>
__global__ __launch_bounds__(1024) void gather_percell_kernel(GatherArgs a) {
    YM_GA_SETUP((int)(blockDim.x >> 6));
    if (st.regular[0] && a.force_irregular != 1 && starts[2 * nbins2] >= 0) return;
    const int nx = a.lat.nx;
    const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
    const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
    const int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
    const int32_t *cy = cx + a.dim_stride;
    const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
    const int nq = st.nq;
    const double add_x = a.g.semantics == 1 ? st.ylat[0] : st.off_x, add_y = a.g.semantics == 1 ? st.ylat[1] : st.off_y; // (ym_k_common.hpp, lookup_cell_sem)
#pragma unroll
    for (int n = 0; n < NA; n++) {
        if (jk[n] < 0) continue;
        const double2 cs = a.ctrig[(size_t)b * a.nt_stride + jk[n]];
        for (int i0 = 0; i0 < nq; i0 += YM_GA_FLUSH) { // one set of 16-bit sums per YM_GA_FLUSH beams
            const int i1 = min(nq, i0 + YM_GA_FLUSH);
#pragma unroll
            for (int p2 = 0; p2 < NP; p2++) {
                const uint32_t lj = a.lane_job[p2 * 64 + lane];
                if (!(lj >> 16)) continue;
                const int row = lj & 0xff, seg = (lj >> 8) & 0xff;
#pragma unroll // (E / S are indexed by j: a loop the compiler leaves rolled puts them into scratch memory)
                for (int j = 0; j < YM_GA_G; j++) {
                    const int ix = seg * YM_GA_G + j;
                    if (ix >= nx) break;
                    const int base = cy[row] * lin_pitch(a.g) + cx[ix];
                    unsigned sum = 0;
                    for (int i = i0; i < i1; i++)
                        sum += cell_value(a.g, grid, limit, (unsigned)(base + lookup_offset_sem(a.g, ql[i], cs.x, cs.y, st.off_x, st.off_y, add_x, add_y, lin_pitch(a.g))));
                    // hypothesis j of dword j >> 2: even ones in E (low / high half), odd ones in S
                    if ((j & 1) == 0) E[n][p2][j >> 2] += sum << (8 * (j & 2));
                    else S[n][p2][j >> 2] += sum << (8 * (j & 2));
                }
            }
            if (i1 < nq) flush(n);
        }
    }
    ga_score<NA, NP>(a, b, 64 * NWV, k_lo, k_hi, jk, E, S, flushed, buf0);
}

} // namespace ym
