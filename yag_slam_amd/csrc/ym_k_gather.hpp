// ym_k_gather.hpp -- K4g: the coarse correlate of BATCHES, gathered from LDS region by region (round 3 form).
// Part of ym_kernels.hpp (include that, not this file).
//
// The hypotheses of one beam are every other cell of every other row of the window, i.e. a DENSE nx x ny block of bytes
// (a "patch") in the image of one (column parity, row parity) CLASS of window cells; sum(hypothesis) = the sum of the
// patches of all beams, byte by byte.  The (beam, angle) pairs of a QUERY depend on the query alone (its points, pose
// and lattice), not on the chain it is matched against, so one list serves every item of a batch:
//   gbin_count / gbin_scan / gbin_place   once per distinct query of a call: the (beam, angle) pairs, runs of beams in one
//                         cell merged into one patch with a multiplicity (coarse grids), sorted by the REGION of class
//                         space the patch starts in, the angle, the patch's byte alignment and its multiplicity; packed as
//                         32-bit UNITS of two patches of equal alignment and multiplicity; a work list of regions.
//   gather_kernel         one block per item walks the work list: region i + 1 (ONE class image + the region's units + its
//                         row of the bin table) is copied global -> LDS by LDS-DMA (global_load_lds) into the second buffer
//                         while region i is gathered; one barrier per region.  A wave owns NA (angle, lattice part) jobs and
//                         keeps their sums in registers; a lane owns 16 x-adjacent hypotheses of one lattice row and reads
//                         their 20 bytes as ds_read_b64 + ds_read_b64 + ds_read_b32 (LDS row pitch = 8 x odd: conflict-free
//                         at 256 B/clk); the raw dwords of a unit's two patches are added as packed bytes (<= 200), byte-
//                         aligned by one v_alignbyte per dword (the shift is wave-uniform) and widened into 16-bit lanes
//                         (even bytes: v_and + v_add; odd bytes: the sum of x >> 8, separated when the sums leave the
//                         registers), times the multiplicity where there is one.  The block then scores its sums itself.
// Every region is staged ONCE per item (round 2 staged every region three times, once per block of an item's angles).
#pragma once

namespace ym {

#define YM_GA_G 16          // hypotheses per lane
#define YM_GA_MAX_SEG 3     // lanes per lattice row: nx <= 48
#define YM_GA_MAX_NP 3      // waves per angle: ny * ceil(nx / 16) <= 192 lane jobs
#define YM_GA_CLS 16        // (byte shift 0..3) x (multiplicity 1..4) classes inside a bin
#define YM_GA_FLUSH 652     // weight (patches x multiplicity) a set of 16-bit sums holds: 652 x 100 < 65536
#define YM_GA_ZERO_BYTES 32 // the all-zero bytes a unit without a second patch reads
#define YM_GBIN_THREADS 256
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
    uint32_t *units;         // [Q][units_stride] low 16: LDS offset of patch A, high 16: (offset of patch B & ~7) | no B << 2 | multiplicity - 1
    size_t units_stride;
    int32_t *starts;         // [Q][starts_stride] first unit of bin2 = (region * nt + angle) * 2 + (offset bit 2); [nbins2] = total or -1
    size_t starts_stride;
    int32_t *work;           // [Q][parts][work_stride]: count, then (region, first unit, end) triples
    size_t work_stride;
    uint32_t *counters;      // [Q][nbins2 * 16] zeroed by the host; gbin_scan turns them into write cursors
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
    if (PLACE && starts[nbins2] < 0) return; // the lists do not fit: every item of this query takes the per-cell path
    const int k = p / nq, i = p - k * nq;
    const int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
    const int cx0 = cx[0], cy0 = cx[a.dim_stride];
    const double off_x = st.off_x, off_y = st.off_y;
    const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
    const double2 cs = a.ctrig[(size_t)b * a.nt_stride + k];
    const int2 c = lookup_cell(ql[i], cs.x, cs.y, off_x, off_y, a.g.scale);
    if (i > 0) {
        const int2 pc = lookup_cell(ql[i - 1], cs.x, cs.y, off_x, off_y, a.g.scale);
        if (pc.x == c.x && pc.y == c.y) return; // inside a run
    }
    int m = 1;
    for (int j = i + 1; j < nq; j++) {
        const int2 nc = lookup_cell(ql[j], cs.x, cs.y, off_x, off_y, a.g.scale);
        if (nc.x != c.x || nc.y != c.y) break;
        m++;
    }
    int R;
    unsigned e;
    if (!ga_entry(a, c, cx0, cy0, R, e)) return;
    const unsigned bin2 = (unsigned)((R * nt + k) * 2) + ((e >> 2) & 1u);
    uint32_t *cnt = a.counters + (size_t)q * nbins2 * YM_GA_CLS + (size_t)bin2 * YM_GA_CLS + (e & 3u) * 4u;
    uint16_t *half = reinterpret_cast<uint16_t *>(a.units + (size_t)q * a.units_stride);
    while (m > 0) {
        const int mp = m < 4 ? m : 4;
        const unsigned pos = atomicAdd(&cnt[mp - 1], 1u); // PLACE: the cursor of the class, in patches
        if (PLACE) half[pos] = (pos & 1u) ? (uint16_t)((e & ~7u) | (unsigned)(mp - 1)) : (uint16_t)e;
        m -= mp;
    }
}

// One block per query slot: the classes' sizes -> first unit of every bin2 (every class padded to whole units), the write
// cursors of the classes, the "no second patch" marks, and per part the work list of region chunks.  If anything does not
// fit, starts[nbins2] = -1 and the items of this query are scored by gather_kernel's per-cell path.
__global__ __launch_bounds__(1024) void gbin_scan_kernel(GatherArgs a) {
    __shared__ int wave_tot[16];
    __shared__ int running;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int nt = a.lat.nt, nreg = a.nrx * a.nry * 4;
    const int nbins2 = nreg * nt * 2;
    uint32_t *cnt = a.counters + (size_t)q * nbins2 * YM_GA_CLS;
    int32_t *starts = a.starts + (size_t)q * a.starts_stride;
    uint32_t *units = a.units + (size_t)q * a.units_stride;
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
        starts[first + j] = run;
        for (int c = 0; c < YM_GA_CLS; c++) {
            uint32_t *pc = &cnt[(size_t)(first + j) * YM_GA_CLS + c];
            const int n = (int)*pc;
            *pc = (uint32_t)(2 * run); // cursor, in patches
            if (fits && (n & 1)) units[run + (n >> 1)] = (4u | (unsigned)(c & 3)) << 16; // the class's last unit has no second patch
            run += (n + 1) >> 1;
        }
    }
    if (tid == 0) running = 0;
    __syncthreads(); // (starts[] of every bin is written: the work lists read them)
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
                u0 = starts[(R * nt + k_lo) * 2];
                u1 = (R * nt + k_hi) * 2 < nbins2 ? starts[(R * nt + k_hi) * 2] : all;
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
        starts[nbins2] = ok ? all : -1;
        if (a.stamps && q == 0) a.stamps[26] = (unsigned long long)all;
    }
}

typedef unsigned int ga_u32x2 __attribute__((ext_vector_type(2)));

// the 20 bytes at LDS address `ad` (a multiple of 4) as five dwords d[0..4]: the two 8-byte-aligned pairs by ds_read_b64
// (256 B/clk), the odd dword by ds_read_b32.  Q = bit 2 of the address (wave-uniform: the lanes' offsets are multiples of
// 8).  Only ISSUES the reads: the registers are written later (see ga_gather2).
#define YM_GA_READS_Q0(p, q, r, ad) "ds_read_b64 " p ", " ad "\n\tds_read_b64 " q ", " ad " offset:8\n\tds_read_b32 " r ", " ad " offset:16\n\t"
#define YM_GA_READS_Q1(p, q, r, ad) "ds_read_b32 " r ", " ad "\n\tds_read_b64 " p ", " ad " offset:4\n\tds_read_b64 " q ", " ad " offset:12\n\t"

template <int Q>
__device__ __forceinline__ void ga_dwords(const ga_u32x2 &p, const ga_u32x2 &q, uint32_t r, uint32_t (&d)[5]) {
    if (Q == 0) { d[0] = p.x; d[1] = p.y; d[2] = q.x; d[3] = q.y; d[4] = r; }
    else { d[0] = r; d[1] = p.x; d[2] = p.y; d[3] = q.x; d[4] = q.y; }
}

// the sums of one unit into the wave's registers: E[j] holds hypotheses 4j (low 16 bits) and 4j + 2, S[j] the running sum
// of (dword >> 8) = sum(4j + 1) + 2^8 sum(4j + 2) + 2^16 sum(4j + 3); ga_odd() separates it when the sums leave the registers
template <bool MULT>
__device__ __forceinline__ void ga_accumulate(uint32_t (&E)[4], uint32_t (&S)[4], const uint32_t (&da)[5], const uint32_t (&db)[5],
                                              uint32_t shift, uint32_t mp) {
    uint32_t s[5];
#pragma unroll
    for (int j = 0; j < 5; j++) s[j] = da[j] + db[j]; // packed bytes, each <= 200
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t x = __builtin_amdgcn_alignbyte(s[j + 1], s[j], shift);
        if (MULT) {
            E[j] = __umul24(x & 0x00FF00FFu, mp) + E[j]; // (v_mad_u32_u24)
            S[j] = __umul24(x >> 8, mp) + S[j];
        } else {
            E[j] += x & 0x00FF00FFu;
            S[j] += x >> 8;
        }
    }
}
__device__ __forceinline__ uint32_t ga_odd(uint32_t E, uint32_t S) { return S - ((E >> 16) << 8); } // sum(4j + 1) | sum(4j + 3) << 16

// Two units (four patches) of one alignment half Q.  All twelve reads are issued at once; the first unit is accumulated
// while the second one's reads are in flight (LDS reads return in order: lgkmcnt(6) = the first six are back).
// The registers an asm statement that only ISSUES a read names as outputs are not written when the statement ends, and
// the compiler is free to copy them right there: every such register is either waited for inside the issuing statement
// or passes through the statement that waits for it ("+v") before anything else touches it.
template <int Q, bool MULT>
__device__ __forceinline__ void ga_gather2(uint32_t (&E)[4], uint32_t (&S)[4], uint32_t lane_off, uint32_t zero_off, uint32_t u0, uint32_t u1) {
    // (wave-uniform) addresses: a unit without a second patch reads the zero bytes, the same address in every lane
    const uint32_t a0 = u0 & 0xffffu, a1 = u1 & 0xffffu;
    const uint32_t nb0 = (u0 >> 18) & 1u, nb1 = (u1 >> 18) & 1u;
    const uint32_t b0 = nb0 ? zero_off + (a0 & 4u) : ((u0 >> 16) & 0xfff8u) | (a0 & 4u);
    const uint32_t b1 = nb1 ? zero_off + (a1 & 4u) : ((u1 >> 16) & 0xfff8u) | (a1 & 4u);
    const uint32_t adA0 = lane_off + (a0 & ~3u), adB0 = (nb0 ? 0u : lane_off) + b0;
    const uint32_t adA1 = lane_off + (a1 & ~3u), adB1 = (nb1 ? 0u : lane_off) + b1;
    ga_u32x2 pa0, qa0, pb0, qb0, pa1, qa1, pb1, qb1;
    uint32_t ra0, rb0, ra1, rb1;
    if (Q == 0)
        asm volatile(YM_GA_READS_Q0("%0", "%1", "%2", "%12") YM_GA_READS_Q0("%3", "%4", "%5", "%13")
                     YM_GA_READS_Q0("%6", "%7", "%8", "%14") YM_GA_READS_Q0("%9", "%10", "%11", "%15") "s_waitcnt lgkmcnt(6)"
                     : "=&v"(pa0), "=&v"(qa0), "=&v"(ra0), "=&v"(pb0), "=&v"(qb0), "=&v"(rb0), "=&v"(pa1), "=&v"(qa1), "=&v"(ra1), "=&v"(pb1),
                       "=&v"(qb1), "=&v"(rb1)
                     : "v"(adA0), "v"(adB0), "v"(adA1), "v"(adB1)
                     : "memory");
    else
        asm volatile(YM_GA_READS_Q1("%0", "%1", "%2", "%12") YM_GA_READS_Q1("%3", "%4", "%5", "%13")
                     YM_GA_READS_Q1("%6", "%7", "%8", "%14") YM_GA_READS_Q1("%9", "%10", "%11", "%15") "s_waitcnt lgkmcnt(6)"
                     : "=&v"(pa0), "=&v"(qa0), "=&v"(ra0), "=&v"(pb0), "=&v"(qb0), "=&v"(rb0), "=&v"(pa1), "=&v"(qa1), "=&v"(ra1), "=&v"(pb1),
                       "=&v"(qb1), "=&v"(rb1)
                     : "v"(adA0), "v"(adB0), "v"(adA1), "v"(adB1)
                     : "memory");
    uint32_t da[5], db[5];
    ga_dwords<Q>(pa0, qa0, ra0, da);
    ga_dwords<Q>(pb0, qb0, rb0, db);
    ga_accumulate<MULT>(E, S, da, db, a0 & 3u, (u0 >> 16 & 3u) + 1u);
    // (the second unit's registers are written by the LDS until here: they pass through this statement and nothing else)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pa1), "+v"(qa1), "+v"(ra1), "+v"(pb1), "+v"(qb1), "+v"(rb1) : : "memory");
    ga_dwords<Q>(pa1, qa1, ra1, da);
    ga_dwords<Q>(pb1, qb1, rb1, db);
    ga_accumulate<MULT>(E, S, da, db, a1 & 3u, (u1 >> 16 & 3u) + 1u);
}
// one unit (the odd one at the end of a job's list of a region)
template <int Q>
__device__ __forceinline__ void ga_gather1(uint32_t (&E)[4], uint32_t (&S)[4], uint32_t lane_off, uint32_t zero_off, uint32_t u0) {
    const uint32_t a0 = u0 & 0xffffu, nb0 = (u0 >> 18) & 1u;
    const uint32_t b0 = nb0 ? zero_off + (a0 & 4u) : ((u0 >> 16) & 0xfff8u) | (a0 & 4u);
    const uint32_t adA0 = lane_off + (a0 & ~3u), adB0 = (nb0 ? 0u : lane_off) + b0;
    ga_u32x2 pa0, qa0, pb0, qb0;
    uint32_t ra0, rb0;
    if (Q == 0)
        asm volatile(YM_GA_READS_Q0("%0", "%1", "%2", "%6") YM_GA_READS_Q0("%3", "%4", "%5", "%7") "s_waitcnt lgkmcnt(0)"
                     : "=&v"(pa0), "=&v"(qa0), "=&v"(ra0), "=&v"(pb0), "=&v"(qb0), "=&v"(rb0) : "v"(adA0), "v"(adB0) : "memory");
    else
        asm volatile(YM_GA_READS_Q1("%0", "%1", "%2", "%6") YM_GA_READS_Q1("%3", "%4", "%5", "%7") "s_waitcnt lgkmcnt(0)"
                     : "=&v"(pa0), "=&v"(qa0), "=&v"(ra0), "=&v"(pb0), "=&v"(qb0), "=&v"(rb0) : "v"(adA0), "v"(adB0) : "memory");
    uint32_t da[5], db[5];
    ga_dwords<Q>(pa0, qa0, ra0, da);
    ga_dwords<Q>(pb0, qb0, rb0, db);
    ga_accumulate<true>(E, S, da, db, a0 & 3u, (u0 >> 16 & 3u) + 1u);
}

// LDS-DMA: 64 lanes x 4 bytes from per-lane global addresses to 256 contiguous LDS bytes at `lds` (wave-uniform)
__device__ __forceinline__ void ga_dma4(const void *src, void *lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src), (__attribute__((address_space(3))) void *)(lds), 4, 0, 0);
}

// dynamic LDS of gather_kernel: [2 x (image | units | bin row)][zero bytes]
#define YM_GA_IMAGE_BYTES(P, rows) (((size_t)(P) * (rows) + 255) / 256 * 256)
#define YM_GA_ROW_INTS(kpp) (((2 * (kpp) + 1) + 63) / 64 * 64)
#define YM_GA_BUF_BYTES(P, rows, cap, kpp) (YM_GA_IMAGE_BYTES(P, rows) + (size_t)(cap) * 4 + (size_t)YM_GA_ROW_INTS(kpp) * 4)
#define YM_GA_LDS_BYTES(P, rows, cap, kpp) (2 * YM_GA_BUF_BYTES(P, rows, cap, kpp) + YM_GA_ZERO_BYTES + 64)

// grid (parts, B): block (part, item) = NWV = blockDim.x / 64 waves; wave w owns the jobs w * NA .. w * NA + NA - 1 of the part, job = (angle
// k_lo + job / NP, lattice part job % NP); the lanes of a part's wave are given by a.lane_job.
template <int NA>
__global__ __launch_bounds__(1024) void gather_kernel(GatherArgs a) {
    const int NT = blockDim.x, NWV = NT >> 6;
    extern __shared__ __attribute__((aligned(16))) unsigned char ga_smem[];
    int part;
    const int b = xcd_item_of_block_2d(part);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const YmItemState &st = a.states[b];
    const int q = st.qslot;
    const int nt = a.lat.nt, nx = a.lat.nx, ny = a.lat.ny, NP = a.NP;
    const int k_lo = part * a.kpp, k_hi = min(nt, k_lo + a.kpp);
    const int njobs = (k_hi - k_lo) * NP;            // of this block
    const int half_pitch = a.g.pitch / 2;
    const int plane_bytes = half_pitch * a.g.win_w;
    const uint8_t *__restrict__ planes = a.planes + (size_t)b * a.grid_stride;
    const int nreg = a.nrx * a.nry * 4, nbins2 = nreg * nt * 2;
    const int32_t *__restrict__ starts = a.starts + (size_t)q * a.starts_stride;
    const uint32_t *__restrict__ units = a.units + (size_t)q * a.units_stride;
    const int32_t *__restrict__ work = a.work + ((size_t)q * a.parts + part) * a.work_stride;
    // LDS carve-up
    const size_t image_bytes = YM_GA_IMAGE_BYTES(a.P, a.rows);
    const int row_ints = YM_GA_ROW_INTS(a.kpp);
    const size_t buf_bytes = image_bytes + (size_t)a.unit_cap * 4 + (size_t)row_ints * 4;
    unsigned char *buf0 = ga_smem;
    unsigned char *zero = buf0 + 2 * buf_bytes;
    const uint32_t lds0 = (uint32_t)(size_t)buf0;
    const uint32_t zero_off = (uint32_t)(size_t)zero;
    // the jobs of this wave and the lane's place in them
    int jk[NA], jp[NA];       // angle, lattice part (-1: no job)
    uint32_t loff[NA];        // row * P + 16 * seg of the lane in the job's part
    uint32_t E[NA][4], S[NA][4];
    int in_set[NA], flushed[NA];
#pragma unroll
    for (int n = 0; n < NA; n++) {
        const int j = wave * NA + n;
        jk[n] = j < njobs ? k_lo + j / NP : -1;
        jp[n] = j < njobs ? j % NP : 0;
        const uint32_t lj = a.lane_job[jp[n] * 64 + lane];
        loff[n] = (lj & 0xffu) * (uint32_t)a.P + ((lj >> 8) & 0xffu) * YM_GA_G;
        in_set[n] = 0;
        flushed[n] = 0;
#pragma unroll
        for (int j2 = 0; j2 < 4; j2++) E[n][j2] = S[n][j2] = 0u;
    }
    auto flush = [&](int n) {
        if (flushed[n] < a.ng && jk[n] >= 0) {
            uint32_t acc[8];
#pragma unroll
            for (int j = 0; j < 4; j++) { acc[2 * j] = E[n][j]; acc[2 * j + 1] = ga_odd(E[n][j], S[n][j]); }
            const size_t job = (size_t)(jk[n] * NP + jp[n]);
            store_partial16(a.partial + (size_t)b * a.partial_stride + ((((size_t)flushed[n] * nt * NP) + job) * 64 + lane) * 16, acc);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) E[n][j] = S[n][j] = 0u;
        flushed[n]++;
        in_set[n] = 0;
    };
    YM_STAMP(a, 8);
    const bool regular = st.regular[0] && a.force_irregular != 1 && starts[nbins2] >= 0;
    if (regular) {
        if (tid < YM_GA_ZERO_BYTES / 4) reinterpret_cast<uint32_t *>(zero)[tid] = 0u;
        const int nwork = work[0];
        // DMA of one work item (region R, units [ub, ue)) into buffer `bi`: the class image (unless the buffer holds it
        // already), the units and the region's row of the bin table.  Nothing is range-checked: rows past the window,
        // columns past a plane row and units past the list are other bytes of the same buffers (the host allocates the
        // slack) that no patch the window holds ever reads.
        const int dpr = a.P >> 2;                              // dwords per staged row
        const int img_dwords = a.rows * dpr;
        const int step_r = (64 * NWV) / dpr, step_c = (64 * NWV) - step_r * dpr; // what 64 * NWV dwords further means
        const int r_first = (wave * 64 + lane) / dpr, c_first = (wave * 64 + lane) - r_first * dpr;
        int staged[2] = {-1, -1}; // the region whose image a buffer holds
        auto stage = [&](int R, int ub, int ue, int bi) {
            unsigned char *buf = buf0 + (size_t)bi * buf_bytes;
            if (staged[bi] != R) {
                staged[bi] = R;
                const int cls = R / (a.nrx * a.nry), rr = R - cls * (a.nrx * a.nry);
                const int RY = rr / a.nrx, RX = rr - RY * a.nrx;
                const uint8_t *src = planes + (size_t)(cls & 1) * plane_bytes + ((size_t)(2 * RY * a.H) + (cls >> 1)) * half_pitch + (size_t)RX * a.W;
                int r = r_first, c4 = c_first;
                for (int t = wave; t * 64 < img_dwords; t += NWV) {
                    ga_dma4(src + (size_t)r * 2 * half_pitch + 4 * c4, buf + (size_t)t * 256);
                    r += step_r; c4 += step_c;
                    if (c4 >= dpr) { c4 -= dpr; r++; }
                }
            }
            unsigned char *ubuf = buf + image_bytes;
            for (int t = wave; t * 64 < ue - ub; t += NWV) ga_dma4(units + ub + t * 64 + lane, ubuf + (size_t)t * 256);
            unsigned char *rbuf = ubuf + (size_t)a.unit_cap * 4;
            const int32_t *srow = starts + ((size_t)R * nt + k_lo) * 2;
            for (int t = wave; t * 64 < 2 * (k_hi - k_lo) + 1; t += NWV) ga_dma4(srow + t * 64 + lane, rbuf + (size_t)t * 256);
        };
        // the work list travels through registers: item wi + 2 is loaded while item wi is gathered (after the DMA of item
        // wi + 1 was issued: vector-memory operations complete in order, so the wait for that DMA covers the load)
        int cR = 0, cub = 0, cue = 0, nR = 0, nub = 0, nue = 0;
        if (nwork > 0) { cR = work[1]; cub = work[2]; cue = work[3]; }
        if (nwork > 1) { nR = work[4]; nub = work[5]; nue = work[6]; }
        cR = __builtin_amdgcn_readfirstlane(cR); cub = __builtin_amdgcn_readfirstlane(cub); cue = __builtin_amdgcn_readfirstlane(cue);
        if (nwork > 0) stage(cR, cub, cue, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int wi = 0; wi < nwork; wi++) {
            const int bi = wi & 1;
            nR = __builtin_amdgcn_readfirstlane(nR); nub = __builtin_amdgcn_readfirstlane(nub); nue = __builtin_amdgcn_readfirstlane(nue);
            if (wi + 1 < nwork) stage(nR, nub, nue, bi ^ 1); // in flight while this item is gathered
            int fR = 0, fub = 0, fue = 0;
            if (wi + 2 < nwork) { fR = work[1 + 3 * (wi + 2)]; fub = work[2 + 3 * (wi + 2)]; fue = work[3 + 3 * (wi + 2)]; }
            const int ub = cub, ue = cue;
            const unsigned char *buf = buf0 + (size_t)bi * buf_bytes;
            const uint32_t *ul = reinterpret_cast<const uint32_t *>(buf + image_bytes);
            const int32_t *row = reinterpret_cast<const int32_t *>(buf + image_bytes + (size_t)a.unit_cap * 4);
            const uint32_t img = lds0 + (uint32_t)(bi * buf_bytes);
#pragma unroll
            for (int n = 0; n < NA; n++) {
                if (jk[n] < 0) continue;
                const int kk = jk[n] - k_lo;
                // this job's units of the region: [s0, s1) with address bit 2 clear, [s1, s2) with it set; this chunk's share
                int s0 = __builtin_amdgcn_readfirstlane(row[2 * kk]), s1 = __builtin_amdgcn_readfirstlane(row[2 * kk + 1]),
                    s2 = __builtin_amdgcn_readfirstlane(row[2 * kk + 2]);
                s0 = max(s0, ub); s1 = min(max(s1, ub), ue); s2 = min(s2, ue);
                const uint32_t lo = img + loff[n];
                auto run = [&](auto qtag, int lo_u, int hi_u) {
                    constexpr int Q = decltype(qtag)::value;
                    int c = lo_u;
                    for (; c + 2 <= hi_u; c += 2) {
                        if (in_set[n] > YM_GA_FLUSH - 16) flush(n);
                        const uint32_t u0 = __builtin_amdgcn_readfirstlane(ul[c - ub]), u1 = __builtin_amdgcn_readfirstlane(ul[c + 1 - ub]);
                        const uint32_t m0 = (u0 >> 16 & 3u) + 1u, m1 = (u1 >> 16 & 3u) + 1u;
                        if ((m0 | m1) == 1u) ga_gather2<Q, false>(E[n], S[n], lo, zero_off, u0, u1);
                        else ga_gather2<Q, true>(E[n], S[n], lo, zero_off, u0, u1);
                        in_set[n] += (int)(m0 * (2u - (u0 >> 18 & 1u)) + m1 * (2u - (u1 >> 18 & 1u)));
                    }
                    if (c < hi_u) {
                        if (in_set[n] > YM_GA_FLUSH - 16) flush(n);
                        const uint32_t u0 = __builtin_amdgcn_readfirstlane(ul[c - ub]);
                        ga_gather1<Q>(E[n], S[n], lo, zero_off, u0);
                        in_set[n] += (int)(((u0 >> 16 & 3u) + 1u) * (2u - (u0 >> 18 & 1u)));
                    }
                };
                run(std::integral_constant<int, 0>(), s0, s1);
                run(std::integral_constant<int, 1>(), s1, s2);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's share of the next item has landed
            __syncthreads();                                  // every wave is done with this buffer, and every share has landed
            cR = nR; cub = nub; cue = nue;
            nR = fR; nub = fub; nue = fue;
        }
    } else {
        // hypothesis cells are not an exact lattice (possible only through fp rounding), or the lists did not fit: per-cell
        // path over the window
        const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
        const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
        const int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
        const int32_t *cy = cx + a.dim_stride;
        const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
        const int nq = st.nq;
#pragma unroll
        for (int n = 0; n < NA; n++) {
            if (jk[n] < 0) continue;
            const uint32_t lj = a.lane_job[jp[n] * 64 + lane];
            if (!(lj >> 16)) continue;
            const int row = lj & 0xff, seg = (lj >> 8) & 0xff;
            const double2 cs = a.ctrig[(size_t)b * a.nt_stride + jk[n]];
            for (int i0 = 0; i0 < nq; i0 += YM_GA_FLUSH) { // one set of 16-bit sums per YM_GA_FLUSH beams
                const int i1 = min(nq, i0 + YM_GA_FLUSH);
                for (int j = 0; j < YM_GA_G; j++) {
                    const int ix = seg * YM_GA_G + j;
                    if (ix >= nx) break;
                    const int base = cy[row] * a.g.pitch + cx[ix];
                    unsigned sum = 0;
                    for (int i = i0; i < i1; i++) {
                        const unsigned idx = (unsigned)(base + lookup_offset(ql[i], cs.x, cs.y, st.off_x, st.off_y, a.g.scale, a.g.pitch));
                        sum += idx < limit ? grid[idx] : 0u;
                    }
                    // hypothesis j: 4 (j >> 2) + (j & 3); even ones in E (low / high half), odd ones through S
                    if ((j & 1) == 0) E[n][j >> 2] += sum << (8 * (j & 2));
                    S[n][j >> 2] += (j & 1) ? sum << (8 * (j & 2)) : (j & 2) ? sum << 8 : 0u; // (what the sum of x >> 8 would hold)
                }
                if (i1 < nq) flush(n);
            }
        }
    }
    YM_STAMP(a, 9);
    // ---- score (score_kernel's arithmetic, statement for statement): a wave holds the last set of its jobs' sums in
    // registers and wrote the earlier ones itself; response, penalty, block maxima; the per-(x, y) maximum over theta and
    // the block maxima go through LDS (the buffers are free now) so that only one atomic per cell and block reaches memory
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
        const uint32_t lj = a.lane_job[jp[n] * 64 + lane];
        const bool job = (lj >> 16) != 0;
        const int row = lj & 0xff, seg = (lj >> 8) & 0xff;
        unsigned tot[YM_GA_G];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t e = E[n][j], o = ga_odd(E[n][j], S[n][j]);
            tot[4 * j] = e & 0xffffu; tot[4 * j + 1] = o & 0xffffu; tot[4 * j + 2] = e >> 16; tot[4 * j + 3] = o >> 16;
        }
        for (int f = 0; f < min(flushed[n], a.ng); f++) { // the sets this lane wrote out earlier
            const uint16_t *pp = a.partial + (size_t)b * a.partial_stride + ((((size_t)f * nt * NP) + (size_t)(k * NP + jp[n])) * 64 + lane) * 16;
#pragma unroll
            for (int j = 0; j < YM_GA_G; j++) tot[j] += pp[j];
        }
        const double angle = (ct - a.lat.angle_off) + k * a.lat.angle_res;
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
    __syncthreads();
    for (int i = tid; i < nxy; i += NT)
        if (pmax[i]) atomicMax(reinterpret_cast<unsigned long long *>(a.probs) + (size_t)b * a.probs_stride + i, pmax[i]);
    for (int i = tid; i < (k_hi - k_lo) * ncb; i += NT)
        a.blockmax[(size_t)b * a.n_blocks + (size_t)k_lo * ncb + i] = __longlong_as_double((long long)bmax[i]);
}

} // namespace ym
