// ym_k_correlate.hpp -- K4 correlate_kernel and the experimental LDS-staged correlate_staged_kernel.
// Part of ym_kernels.hpp (include that, not this file).
#pragma once
#include <type_traits>

namespace ym {

// ================================================================== K4 correlate (coarse lattice)
#define YM_CORR_THREADS 256
#define YM_CORR_DEDUP_U 8
struct CorrArgs {
    YmGeom g;
    YmLattice lat;
    const uint8_t *grid;
    size_t grid_stride;
    const uint8_t *planes;  // even/odd column planes of the window (coarse step = 2 cells)
    const double2 *ctrig;   // [B][nt_stride] (cos, sin) per coarse angle
    const double2 *qlocal;  // [B][max_n] query points in the sensor frame
    const int32_t *hypcell;
    const YmItemState *states;
    uint16_t *partial;     // [B][n_groups][nt][ny][nx_pad], 16-bit: a chunk group sums at most 640 beams x 100
    size_t partial_stride; // per item
    int32_t max_n, nt_stride, dim_stride;
    int32_t chunk;         // beams per chunk (multiple of 16, <= 512 keeps the 16-bit lanes from overflowing)
    int32_t n_chunks;
    int32_t tpb;           // staged kernel: development mode switch
    int32_t cw;            // chunk-waves per block (1, 2 or 4): consecutive beam chunks summed inside a block
    int32_t dedup;         // 1: consecutive beams with the same lookup offset are merged into one entry with a multiplicity
    int32_t pad2;          // (coarse grids: several beams per cell; needs chunk == 64, one wave builds one chunk)
    int32_t k_begin, nk;   // the coarse angles this launch scores: [k_begin, k_begin + nk) (all of them unless the match
                           // is split over several matchers by angle)
    int32_t ngx;           // x groups per row = ceil(nx / G)
    int32_t nx_pad;        // ngx * G
    int32_t sx;            // cell stride between x-adjacent hypotheses (1 or 2)
    unsigned long long *stamps;
};

// Workgroups are handed to the 8 XCDs round-robin in launch order, and each XCD has its own L2.  The blocks of
// one item (all angles and beam chunks) read the same grid band, so they are renumbered to run on ONE XCD: XCD x
// works through items x, x + 8, x + 16, ...  Returns the item; `inner` = index within the item's blocks.
__device__ __forceinline__ int xcd_item_of_block(int &inner_x, int &inner_y) {
    const int per_item = gridDim.x * gridDim.y, nb = gridDim.z;
    const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int full = (nb / 8) * 8 * per_item;
    int item = blockIdx.z, inner = blockIdx.x + gridDim.x * blockIdx.y;
    if (L < full) {
        const int t = L >> 3;
        item = (L & 7) + 8 * (t / per_item);
        inner = t % per_item;
    }
    inner_y = inner / gridDim.x;
    inner_x = inner - inner_y * gridDim.x;
    return item;
}

// the same for a grid (blocks per item, items)
__device__ __forceinline__ int xcd_item_of_block_2d(int &inner) {
    const int per_item = gridDim.x, nb = gridDim.y;
    const int L = blockIdx.x + gridDim.x * blockIdx.y;
    int item = blockIdx.y;
    inner = blockIdx.x;
    if (L < (nb / 8) * 8 * per_item) {
        const int t = L >> 3;
        item = (L & 7) + 8 * (t / per_item);
        inner = t % per_item;
    }
    return item;
}

// The correlate kernels accumulate 16 hypotheses per lane in eight dwords of two 16-bit lanes each: acc[2j] holds
// hypotheses 4j and 4j + 2, acc[2j + 1] holds 4j + 1 and 4j + 3 (even / odd bytes of grid dword j).  A chunk is at
// most 512 beams of at most 100, so partial sums are stored as 16-bit values, in hypothesis order (v_perm_b32).
__device__ __forceinline__ void store_partial16(uint16_t *out, const uint32_t (&acc)[8]) {
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        w[2 * j] = __builtin_amdgcn_perm(acc[2 * j + 1], acc[2 * j], 0x05040100u);     // hyp 4j, 4j + 1
        w[2 * j + 1] = __builtin_amdgcn_perm(acc[2 * j + 1], acc[2 * j], 0x07060302u); // hyp 4j + 2, 4j + 3
    }
    *reinterpret_cast<uint4 *>(out) = make_uint4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<uint4 *>(out + 8) = make_uint4(w[4], w[5], w[6], w[7]);
}

// Lane job = 16 x-adjacent hypotheses of one lattice row for one angle.  The coarse search steps
// 2 cells (SX = 2), so its hypotheses' cells for one beam are every other byte of a row: they are
// contiguous in the even- or odd-column plane the raster kernel also writes (which plane is a
// per-beam, wave-uniform choice: parity of hypothesis column + beam offset).  For every beam of
// its chunk a lane loads the 16 plane bytes of its 16 hypotheses (row start + wave-uniform beam
// offset) and accumulates them in 16-bit lanes: no cross-lane reduction, no wasted bytes.
// A block is four waves = jw job-waves x cw chunk-waves (a.cw = 1, 2 or 4): small lattices have one wave of jobs,
// so the four waves take four consecutive beam chunks of the same jobs and add their packed sums through LDS before
// one 16-bit partial per chunk GROUP is written (a quarter of the partial-sum traffic score_kernel would otherwise
// read back).  Each block first builds the offsets of its own beam chunks in LDS (GridIndexLookup::ComputeOffsets for
// one angle: rotate the sensor-frame point, WorldToGrid), so no lookup table ever round-trips through HBM.  Loads are
// issued U beams at a time; entries past the last beam are 0 and are masked by a scalar.
// grid (ceil(ny*ngx / (64 jw)), nt * n_groups, B).
template <int SX, int U /* beams in flight per lane */, int CW /* chunk-waves per block */>
__global__ __launch_bounds__(YM_CORR_THREADS) void correlate_kernel(CorrArgs a) {
    constexpr int G = 16;           // hypotheses per lane
    int bx, by;
    const int b = xcd_item_of_block(bx, by);
    constexpr int cw = CW, jw = 4 / cw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jw_idx = wave % jw, cw_idx = wave / jw;
    const int k = a.k_begin + by % a.nk, group = by / a.nk;
    const int chunk = group * cw + cw_idx;
    const int job = (bx * jw + jw_idx) * 64 + lane;
    __shared__ int offs_all[CW][512];
    __shared__ unsigned short mult_all[CW][128]; // dedup: multiplicity of every merged entry
    __shared__ int n_entries[CW];                // dedup: entries of the chunk, padded to a multiple of U
    __shared__ uint32_t red[CW > 1 ? 4 * 8 * 64 : 1];
    const int *offs = offs_all[cw_idx];
    YM_STAMP(a, 8);
    const int njobs = a.lat.ny * a.ngx;
    const YmItemState &st = a.states[b];
    const int nq = st.nq;
    const int regular = st.regular[0];
    const int i0 = chunk * a.chunk;
    const int32_t *cx = a.hypcell + (size_t)b * 2 * a.dim_stride;
    const int32_t *cy = cx + a.dim_stride;
    const int cx0 = cx[0];
    const int half_pitch = a.g.pitch / 2;
    const int plane_bytes = half_pitch * a.g.win_w;
    {
        const double2 cs = a.ctrig[(size_t)b * a.nt_stride + k];
        const double off_x = st.off_x, off_y = st.off_y;
        const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
        const bool dedup = SX == 2 && a.dedup && regular; // (host: only with chunk == 64)
        if (dedup) {
            // On a coarse grid consecutive beams land in the same cell: every run of equal lookup offsets becomes ONE
            // entry with its length as multiplicity (sum_i G[c + o_i] = sum_runs len * G[c + o_run]: integer, exact).
            // One wave compacts one 64-beam chunk by ballot.
            for (int ci = wave; ci < cw; ci += 4) {
                const int i = (group * cw + ci) * 64 + lane;
                const bool valid = i < nq;
                int o = 0;
                if (valid) {
                    const int l = lookup_offset(ql[i], cs.x, cs.y, off_x, off_y, a.g.scale, a.g.pitch) + cx0;
                    o = (l >> 1) + (l & 1) * plane_bytes;
                }
                const int prev = __shfl_up(o, 1);
                const bool newrun = valid && (lane == 0 || o != prev);
                const unsigned long long mask = __ballot(newrun);
                const int nvalid = __popcll(__ballot(valid)), n_u = __popcll(mask);
                if (newrun) {
                    const int pos = __popcll(mask & ((1ull << lane) - 1ull));
                    const unsigned long long higher = lane == 63 ? 0ull : (mask >> (lane + 1));
                    const int next = higher ? lane + 1 + (__ffsll((long long)higher) - 1) : nvalid;
                    offs_all[ci][pos] = o;
                    mult_all[ci][pos] = (unsigned short)(next - lane);
                }
                const int padded = (n_u + YM_CORR_DEDUP_U - 1) / YM_CORR_DEDUP_U * YM_CORR_DEDUP_U;
                if (lane >= n_u && lane < padded) { offs_all[ci][lane] = 0; mult_all[ci][lane] = 0; }
                if (lane + 64 < padded) { offs_all[ci][lane + 64] = 0; mult_all[ci][lane + 64] = 0; }
                if (lane == 0) n_entries[ci] = padded;
            }
        } else
        for (int e = threadIdx.x; e < cw * a.chunk; e += YM_CORR_THREADS) {
            const int ci = e / a.chunk, c = e - ci * a.chunk;
            const int i = (group * cw + ci) * a.chunk + c;
            int o = i < nq ? lookup_offset(ql[i], cs.x, cs.y, off_x, off_y, a.g.scale, a.g.pitch) : 0;
            if (SX == 2 && regular) {
                // window-linear index of hypothesis column 0 for this beam -> (plane, index in plane)
                const int l = o + cx0;
                o = (l >> 1) + (l & 1) * plane_bytes;
            }
            offs_all[ci][c] = o;
        }
    }
    __syncthreads();
    const bool active = job < njobs && i0 < nq;
    const int iy = min(job, njobs - 1) / a.ngx, xg = min(job, njobs - 1) - iy * a.ngx;
    const int cyv = cy[iy];
    uint32_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0u;

    if (active && regular) {
        const uint8_t *__restrict__ src = SX == 2 ? a.planes + (size_t)b * a.grid_stride : a.grid + (size_t)b * a.grid_stride;
        const uint32_t lane_off = SX == 2 ? (uint32_t)(cyv * half_pitch + xg * G)
                                          : (uint32_t)(cyv * a.g.pitch + cx0 + xg * G);
        if (SX == 2) {
            // Plane loads are made dword-aligned (a byte-unaligned 16-byte load costs the vector L1 ~1.4x the
            // lookups, profiles/r01_c): the lane loads the aligned 16 bytes below its first hypothesis, takes the
            // 17th..19th byte from its right-hand neighbour lane (same row, next 16 hypotheses: DPP wave shift) and
            // funnels by the beam's byte misalignment, which is wave-uniform (v_alignbyte_b32).
            // lanes whose neighbour is not the next group of the same row load the extra dword themselves
            const bool extra = (lane == 63 && xg != a.ngx - 1) || (a.nx_pad - a.lat.nx < 3 && xg == a.ngx - 1);
            // Grid bytes are at most 100, so the bytes of TWO beams add without carries as packed u8: beams are added in
            // pairs first and the pair sum is split into the 16-bit lanes (a third fewer VALU per beam).  Only the last
            // chunk of a scan holds beams past the last reading; they are masked there (MASKED), nowhere else.
            if (a.dedup) {
                typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                const unsigned short *mult = mult_all[cw_idx];
                const int n_ent = n_entries[cw_idx];
                constexpr int DU = YM_CORR_DEDUP_U; // merged entries in flight per lane (their count is padded to a multiple)
                for (int c = 0; c < n_ent; c += DU) {
                    uint4 w[DU];
                    uint32_t e[DU];
#pragma unroll
                    for (int u = 0; u < DU; u++) {
                        const uint32_t ad = lane_off + ((uint32_t)offs[c + u] & ~3u);
                        w[u] = *reinterpret_cast<const uint4 *>(__builtin_assume_aligned(src + ad, 4));
                        e[u] = 0u;
                    }
                    if (extra) {
#pragma unroll
                        for (int u = 0; u < DU; u++) {
                            const uint32_t ad = lane_off + ((uint32_t)offs[c + u] & ~3u);
                            e[u] = *reinterpret_cast<const uint32_t *>(src + ad + 16);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < DU; u++) {
                        const uint32_t rr = (uint32_t)offs[c + u] & 3u;
                        const unsigned short mv = mult[c + u]; // wave-uniform; 0 for the padding entries
                        const us2 mm = (us2){mv, mv};
                        const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[u].x, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
                        const uint32_t w4 = extra ? e[u] : nb;
                        const uint32_t x[4] = {__builtin_amdgcn_alignbyte(w[u].y, w[u].x, rr), __builtin_amdgcn_alignbyte(w[u].z, w[u].y, rr),
                                               __builtin_amdgcn_alignbyte(w[u].w, w[u].z, rr), __builtin_amdgcn_alignbyte(w4, w[u].w, rr)};
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t ev = x[j] & 0x00FF00FFu, od = __builtin_amdgcn_perm(0u, x[j], 0x0c030c01u);
                            us2 a0, a1, e2, o2;
                            __builtin_memcpy(&a0, &acc[2 * j], 4); __builtin_memcpy(&a1, &acc[2 * j + 1], 4);
                            __builtin_memcpy(&e2, &ev, 4); __builtin_memcpy(&o2, &od, 4);
                            a0 = (us2)(e2 * mm + a0); // v_pk_mad_u16: a run is at most 64 beams x 100
                            a1 = (us2)(o2 * mm + a1);
                            __builtin_memcpy(&acc[2 * j], &a0, 4); __builtin_memcpy(&acc[2 * j + 1], &a1, 4);
                        }
                    }
                }
            } else {
            auto run = [&](auto masked_tag) {
                constexpr bool MASKED = decltype(masked_tag)::value;
                for (int c = 0; c < a.chunk; c += U) {
                    uint4 w[U];
                    uint32_t e[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const uint32_t ad = lane_off + ((uint32_t)offs[c + u] & ~3u);
                        w[u] = *reinterpret_cast<const uint4 *>(__builtin_assume_aligned(src + ad, 4));
                        e[u] = 0u;
                    }
                    if (extra) { // one divergent region for all U loads (not U of them)
#pragma unroll
                        for (int u = 0; u < U; u++) {
                            const uint32_t ad = lane_off + ((uint32_t)offs[c + u] & ~3u); // 32-bit wrap: offsets may be negative
                            e[u] = *reinterpret_cast<const uint32_t *>(src + ad + 16);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; u += 2) {
                        uint32_t x[2][4];
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const uint32_t rr = (uint32_t)offs[c + u + h] & 3u;
                            const uint32_t nb = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[u + h].x, 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
                            const uint32_t w4 = extra ? e[u + h] : nb;
                            x[h][0] = __builtin_amdgcn_alignbyte(w[u + h].y, w[u + h].x, rr);
                            x[h][1] = __builtin_amdgcn_alignbyte(w[u + h].z, w[u + h].y, rr);
                            x[h][2] = __builtin_amdgcn_alignbyte(w[u + h].w, w[u + h].z, rr);
                            x[h][3] = __builtin_amdgcn_alignbyte(w4, w[u + h].w, rr);
                            if (MASKED) {
                                const uint32_t m = (i0 + c + u + h) < nq ? 0xFFFFFFFFu : 0u; // wave-uniform
#pragma unroll
                                for (int j = 0; j < 4; j++) x[h][j] &= m;
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const uint32_t s2 = x[0][j] + x[1][j]; // packed bytes, each <= 200
                            acc[2 * j] += s2 & 0x00FF00FFu;
                            acc[2 * j + 1] += __builtin_amdgcn_perm(0u, s2, 0x0c030c01u); // bytes 1 and 3 -> 16-bit lanes
                        }
                    }
                }
            };
            if (i0 + a.chunk <= nq) run(std::false_type());
            else run(std::true_type());
            }
        } else {
            for (int c = 0; c < a.chunk; c += U) {
                uint4 w[U];
#pragma unroll
                for (int u = 0; u < U; u++) __builtin_memcpy(&w[u], src + (uint32_t)(lane_off + (uint32_t)offs[c + u]), 16);
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t m = (i0 + c + u) < nq ? 0x00FF00FFu : 0u; // wave-uniform
                    acc[0] += w[u].x & m; acc[1] += (w[u].x >> 8) & m;
                    acc[2] += w[u].y & m; acc[3] += (w[u].y >> 8) & m;
                    acc[4] += w[u].z & m; acc[5] += (w[u].z >> 8) & m;
                    acc[6] += w[u].w & m; acc[7] += (w[u].w >> 8) & m;
                }
            }
        }
        YM_STAMP(a, 9);
    } else if (active) {
        // hypothesis cells are not an exact lattice (possible only through fp rounding): per-cell path, packed like the
        // fast path's lanes (acc[2j + (h & 1)] holds hypothesis 4j + h in its low (h < 2) or high 16 bits)
        const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
        const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
        const int n_here = min(nq - i0, a.chunk);
        for (int j = 0; j < G; j++) {
            const int ix = xg * G + j;
            unsigned sum = 0;
            if (ix < a.lat.nx) {
                const int base = cyv * a.g.pitch + cx[ix];
                for (int i = 0; i < n_here; i++) {
                    const unsigned idx = (unsigned)(base + offs[i]);
                    sum += idx < limit ? grid[idx] : 0u;
                }
            }
            acc[2 * (j >> 2) + (j & 1)] += sum << (16 * ((j >> 1) & 1));
        }
    }
    if (cw > 1) { // add the chunk-waves of each job-wave (packed 16-bit lanes: a group is at most 640 beams x 100)
#pragma unroll
        for (int j = 0; j < 8; j++) red[(wave * 8 + j) * 64 + lane] = acc[j];
        __syncthreads();
        if (cw_idx != 0) return;
#pragma unroll
        for (int j = 0; j < 8; j++)
            for (int w2 = 1; w2 < cw; w2++) acc[j] += red[((w2 * jw + jw_idx) * 8 + j) * 64 + lane];
    }
    if (job >= njobs) return;
    uint16_t *out = a.partial + (size_t)b * a.partial_stride +
                    (((size_t)group * a.lat.nt + k) * a.lat.ny + iy) * a.nx_pad + (size_t)xg * G;
    store_partial16(out, acc);
}

__device__ __forceinline__ int2 lookup_cell(double2 p, double cosine, double sine, double off_x, double off_y, double scale) {
    const double ox = cosine * p.x - sine * p.y;
    const double oy = sine * p.x + cosine * p.y;
    return make_int2(world_to_grid(ox + off_x, off_x, scale), world_to_grid(oy + off_y, off_y, scale));
}

// ---- LDS-staged coarse correlate.
// The global-load kernel above is bound by the vector L1 (about one lane access per clock per CU, 52+ lane
// accesses per (beam, angle), profiles/r01_b/r01_c).  Consecutive beams of a scan hit neighbouring cells, so the
// patches of a GROUP of 32 consecutive beams overlap: their bounding rectangle holds 4-6x fewer bytes than the
// group gathers.  This kernel copies that rectangle of both column planes into LDS once (aligned 16-byte loads:
// 4-6x fewer L1 accesses) and gathers from LDS, whose read path is 8x wider than the L1's.
//   block   = (item, angle, beam chunk, 64 lane jobs), 4 waves; all waves hold the same 64 lane jobs and split
//             the beams of a group (8 each), their packed sums are added at the end
//   group   = 32 consecutive beams; rectangle = rows Y0..Y0+H of plane bytes Xp0..Xp0+16*spr, LDS pitch 48/80/
//             112/144 B (2*pitch = 32 mod 64: the 26 rows x 2 lanes of a ds_read_b128 fall on distinct banks)
//   pipeline: global loads of group g+1 are in flight (registers) while group g is gathered from buffer g&1, then
//             stored into buffer (g+1)&1; one barrier per group
//   a group whose rectangle does not fit (depth discontinuity inside the group) is gathered with direct loads.
#define YM_ST_GROUP 32
#define YM_ST_PER_WAVE (YM_ST_GROUP / 4)
#define YM_ST_REGION 16384               // bytes per LDS buffer (both planes of one rectangle)
#define YM_ST_NL 4                       // staging loads per thread per plane: rows (tid >> 3) + 32 u, segment tid & 7
#define YM_ST_MAX_GROUPS 16              // chunk <= 512 beams
struct StRect { int Xp0, Y0, H, spr, pitch, use_lds, nvalid; };

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ StRect st_rect_of(const int *s_rect, int ngroups, int g) {
    StRect r;
    r.Xp0 = r.Y0 = r.H = r.spr = r.use_lds = r.nvalid = 0;
    r.pitch = 80;
    if (g < ngroups) {
        const int4 p = *reinterpret_cast<const int4 *>(s_rect + g * 8), q = *reinterpret_cast<const int4 *>(s_rect + g * 8 + 4);
        r.Xp0 = __builtin_amdgcn_readfirstlane(p.x); r.Y0 = __builtin_amdgcn_readfirstlane(p.y);
        r.H = __builtin_amdgcn_readfirstlane(p.z); r.spr = __builtin_amdgcn_readfirstlane(p.w);
        r.pitch = __builtin_amdgcn_readfirstlane(q.x); r.use_lds = __builtin_amdgcn_readfirstlane(q.y);
        r.nvalid = __builtin_amdgcn_readfirstlane(q.z);
    }
    return r;
}
// The staging loads are inline asm on purpose: hipcc otherwise sinks a load whose only use is a conditional LDS
// store down to that store and waits right there, which serialises the pipeline (seen in the ISA).  As asm they
// are issued where written (before the gather of the previous group) and waited for with the explicit
// s_waitcnt after it.  Thread (row = tid >> 3, seg = tid & 7) copies 16-byte block `seg` of rows row, row + 32, ...
// of both planes; lanes outside the rectangle issue nothing.
__device__ __forceinline__ void st_stage_load(u32x4 (&v)[2 * YM_ST_NL], const StRect &r, const uint8_t *planes, int plane_bytes,
                                              int half_pitch) {
    const int seg = threadIdx.x & 7, row0 = threadIdx.x >> 3;
    if (r.use_lds && seg < r.spr) {
        // the funnel's look-ahead block may start past the plane row: re-read the row's last full block instead
        const int xb = min(r.Xp0 + 16 * seg, half_pitch - 16);
        const uint32_t off = (uint32_t)((r.Y0 + row0) * half_pitch + xb);
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            const uint8_t *base = planes + (size_t)pl * plane_bytes;
#pragma unroll
            for (int u = 0; u < YM_ST_NL; u++)
                if (row0 + 32 * u < r.H) {
                    const uint32_t o = off + (uint32_t)(32 * u * half_pitch);
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[pl * YM_ST_NL + u]) : "v"(o), "s"(base) : "memory");
                }
        }
    }
}
// wait for the staging loads, then hand the registers to the compiler
#define YM_ST_WAIT(v)                                                                                          \
    asm volatile("s_waitcnt vmcnt(0)"                                                                          \
                 : "+v"((v)[0]), "+v"((v)[1]), "+v"((v)[2]), "+v"((v)[3]), "+v"((v)[4]), "+v"((v)[5]), "+v"((v)[6]), "+v"((v)[7]) \
                 :                                                                                             \
                 : "memory")
__device__ __forceinline__ void st_stage_store(const u32x4 (&v)[2 * YM_ST_NL], const StRect &r, unsigned char *buf) {
    const int seg = threadIdx.x & 7, row0 = threadIdx.x >> 3;
    if (r.use_lds && seg < r.spr) {
        unsigned char *p = buf + row0 * r.pitch + 16 * seg;
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int u = 0; u < YM_ST_NL; u++)
                if (row0 + 32 * u < r.H)
                    *reinterpret_cast<u32x4 *>(p + (pl * r.H + 32 * u) * r.pitch) = v[pl * YM_ST_NL + u];
    }
}

struct StCtx { // wave/lane constants of the gather
    const uint8_t *planes;
    const int2 *cells;
    int cx0, cy0, iy, xg, wave, lane, plane_bytes, half_pitch, mode;
};
__device__ __forceinline__ void st_accumulate(uint32_t (&acc)[8], uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
    const uint32_t m = 0x00FF00FFu;
    acc[0] += x0 & m; acc[1] += (x0 >> 8) & m;
    acc[2] += x1 & m; acc[3] += (x1 >> 8) & m;
    acc[4] += x2 & m; acc[5] += (x2 >> 8) & m;
    acc[6] += x3 & m; acc[7] += (x3 >> 8) & m;
}
// four beams at once: grid bytes are <= 100, so two beams add without carries as packed bytes; the two pair sums
// are then split into even/odd bytes (v_and / v_perm) and added to the 16-bit lanes with one v_add3 each
__device__ __forceinline__ void st_accumulate4(uint32_t (&acc)[8], const uint32_t (&x)[4][4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t s01 = x[0][j] + x[1][j], s23 = x[2][j] + x[3][j];
        acc[2 * j] = acc[2 * j] + (s01 & 0x00FF00FFu) + (s23 & 0x00FF00FFu);
        acc[2 * j + 1] = acc[2 * j + 1] + __builtin_amdgcn_perm(0u, s01, 0x0c030c01u) + __builtin_amdgcn_perm(0u, s23, 0x0c030c01u);
    }
}
// 16 bytes starting `sh` (0..15, wave-uniform) bytes into the 32 bytes (w0, w1)
#define YM_ST_FUNNEL(w0, w1, sh, x0, x1, x2, x3)                                                                     \
    do {                                                                                                              \
        const int rr_ = (sh) & 3;                                                                                     \
        switch ((sh) >> 2) {                                                                                          \
        case 0:                                                                                                       \
            x0 = __builtin_amdgcn_alignbyte((w0).y, (w0).x, rr_); x1 = __builtin_amdgcn_alignbyte((w0).z, (w0).y, rr_); \
            x2 = __builtin_amdgcn_alignbyte((w0).w, (w0).z, rr_); x3 = __builtin_amdgcn_alignbyte((w1).x, (w0).w, rr_); \
            break;                                                                                                    \
        case 1:                                                                                                       \
            x0 = __builtin_amdgcn_alignbyte((w0).z, (w0).y, rr_); x1 = __builtin_amdgcn_alignbyte((w0).w, (w0).z, rr_); \
            x2 = __builtin_amdgcn_alignbyte((w1).x, (w0).w, rr_); x3 = __builtin_amdgcn_alignbyte((w1).y, (w1).x, rr_); \
            break;                                                                                                    \
        case 2:                                                                                                       \
            x0 = __builtin_amdgcn_alignbyte((w0).w, (w0).z, rr_); x1 = __builtin_amdgcn_alignbyte((w1).x, (w0).w, rr_); \
            x2 = __builtin_amdgcn_alignbyte((w1).y, (w1).x, rr_); x3 = __builtin_amdgcn_alignbyte((w1).z, (w1).y, rr_); \
            break;                                                                                                    \
        default:                                                                                                      \
            x0 = __builtin_amdgcn_alignbyte((w1).x, (w0).w, rr_); x1 = __builtin_amdgcn_alignbyte((w1).y, (w1).x, rr_); \
            x2 = __builtin_amdgcn_alignbyte((w1).z, (w1).y, rr_); x3 = __builtin_amdgcn_alignbyte((w1).w, (w1).z, rr_); \
            break;                                                                                                    \
        }                                                                                                             \
    } while (0)
// gather this wave's beams of group g (beams wave*8 .. wave*8+7 of the group)
__device__ __forceinline__ void st_gather(uint32_t (&acc)[8], const StCtx &c, int g, const StRect &r, const unsigned char *buf) {
    constexpr int G = 16;
    constexpr int HALF = YM_ST_PER_WAVE / 2;
    static_assert(HALF == 4, "st_accumulate4 takes four beams");
    const int first = c.wave * YM_ST_PER_WAVE;
    // lane q (< 32) of every wave prepares beam q of the group: the wave-uniform part of its LDS (or plane) address
    // and its byte shift; the gather loop below picks them up with v_readlane
    const int2 cc = c.cells[g * YM_ST_GROUP + (c.lane & (YM_ST_GROUP - 1))];
    const int col0 = c.cx0 + cc.x;
    if (r.use_lds) {
        // two aligned 16-byte LDS reads per beam, then a wave-uniform byte funnel.  (A single ds_read_b128 at the
        // 4-byte-aligned address + ds_read_b32 is legal on gfx950 and needs no dword switch, but measured slower:
        // the misaligned read is split by the LDS, scripts/exp/lds_unaligned.hip, profiles/r01_c.)
        const int a16 = (col0 >> 1) - r.Xp0;
        const int plane_sz = r.H * r.pitch;
        const int ubase = (col0 & 1) * plane_sz + (c.cy0 + cc.y - r.Y0) * r.pitch + (a16 & ~15);
        const int ush = a16 & 15;
        const int lane_off = (2 * c.iy) * r.pitch + G * c.xg;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            uint4 w0[HALF], w1[HALF];
            const int bfirst = first + h * HALF;
            if (bfirst >= r.nvalid) break; // wave-uniform
#pragma unroll
            for (int q = 0; q < HALF; q++) {
                const int bb = min(bfirst + q, r.nvalid - 1); // past the end: last beam again, dropped below
                const int addr = lane_off + __builtin_amdgcn_readlane(ubase, bb);
                w0[q] = *reinterpret_cast<const uint4 *>(buf + addr);
                w1[q] = *reinterpret_cast<const uint4 *>(buf + addr + 16);
            }
            uint32_t x[HALF][4];
#pragma unroll
            for (int q = 0; q < HALF; q++) {
                const int sh = __builtin_amdgcn_readlane(ush, min(bfirst + q, r.nvalid - 1));
                YM_ST_FUNNEL(w0[q], w1[q], sh, x[q][0], x[q][1], x[q][2], x[q][3]);
            }
            if (bfirst + HALF <= r.nvalid) {
                st_accumulate4(acc, x);
            } else {
#pragma unroll
                for (int q = 0; q < HALF; q++)
                    if (bfirst + q < r.nvalid) st_accumulate(acc, x[q][0], x[q][1], x[q][2], x[q][3]);
            }
        }
    } else if (c.mode != 2) {
        // direct (byte-unaligned) plane loads for a group whose rectangle does not fit: all of this wave's beams
        // in flight at once -- these waves are bound by the vector L1 while the staged ones are bound by issue,
        // so the two kinds overlap on a CU
        const int ubase = (col0 & 1) * c.plane_bytes + (c.cy0 + cc.y) * c.half_pitch + (col0 >> 1);
        const uint32_t lane_off = (uint32_t)((2 * c.iy) * c.half_pitch + G * c.xg);
        if (first < r.nvalid) {
            uint4 w[YM_ST_PER_WAVE];
#pragma unroll
            for (int q = 0; q < YM_ST_PER_WAVE; q++) {
                const int bb = min(first + q, r.nvalid - 1);
                __builtin_memcpy(&w[q], c.planes + (lane_off + (uint32_t)__builtin_amdgcn_readlane(ubase, bb)), 16);
            }
            if (first + YM_ST_PER_WAVE <= r.nvalid) {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    uint32_t x[HALF][4];
#pragma unroll
                    for (int q = 0; q < HALF; q++) { x[q][0] = w[h * HALF + q].x; x[q][1] = w[h * HALF + q].y; x[q][2] = w[h * HALF + q].z; x[q][3] = w[h * HALF + q].w; }
                    st_accumulate4(acc, x);
                }
            } else {
#pragma unroll
                for (int q = 0; q < YM_ST_PER_WAVE; q++)
                    if (first + q < r.nvalid) st_accumulate(acc, w[q].x, w[q].y, w[q].z, w[q].w);
            }
        }
    }
}

// grid (ceil(njobs / 64), nt * n_chunks, B), 256 threads, SX == 2 only
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void correlate_staged_kernel(CorrArgs a) {
    constexpr int G = 16;
    __shared__ __attribute__((aligned(16))) unsigned char region[2 * YM_ST_REGION];
    __shared__ int2 s_cells[512];
    __shared__ int s_box[YM_ST_MAX_GROUPS * 4];
    __shared__ int s_rect[YM_ST_MAX_GROUPS * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    const int b = xcd_item_of_block(bx, by);
    const int k = a.k_begin + by % a.nk, chunk = by / a.nk;
    const int njobs = a.lat.ny * a.ngx;
    const int j_first = bx * 64, j_last = min(njobs, j_first + 64) - 1;
    const bool active = j_first + lane <= j_last;
    const int job = min(j_first + lane, j_last); // idle lanes shadow the last job (valid addresses, result dropped)
    const int iy = job / a.ngx, xg = job - iy * a.ngx;
    const YmItemState &st = a.states[b];
    const int nq = st.nq;
    const int i0 = chunk * a.chunk;
    const int32_t *cxp = a.hypcell + (size_t)b * 2 * a.dim_stride;
    const int cx0 = cxp[0], cy0 = cxp[a.dim_stride];
    const int half_pitch = a.g.pitch / 2;
    const int plane_bytes = half_pitch * a.g.win_w;
    const uint8_t *__restrict__ planes = a.planes + (size_t)b * a.grid_stride;
    const int iy_min = j_first / a.ngx, iy_max = j_last / a.ngx;
    const int xg_min = iy_min == iy_max ? j_first % a.ngx : 0, xg_max = iy_min == iy_max ? j_last % a.ngx : a.ngx - 1;
    const int ngroups = (a.chunk + YM_ST_GROUP - 1) / YM_ST_GROUP;

    YM_STAMP(a, 20);
    // ---- cells of the whole chunk (GridIndexLookup::ComputeOffsets for this angle), bounding box per group
    if (tid < YM_ST_MAX_GROUPS * 4) s_box[tid] = (tid & 1) ? INT32_MIN : INT32_MAX; // [g][x0, x1, y0, y1]
    __syncthreads();
    {
        const double2 cs = a.ctrig[(size_t)b * a.nt_stride + k];
        const double off_x = st.off_x, off_y = st.off_y;
        const double2 *ql = reinterpret_cast<const double2 *>(st.ql);
        for (int c = tid; c < a.chunk; c += 256) {
            const int i = i0 + c;
            int2 cell = make_int2(0, 0);
            if (i < nq) {
                cell = lookup_cell(ql[i], cs.x, cs.y, off_x, off_y, a.g.scale);
                int *bx = s_box + (c / YM_ST_GROUP) * 4;
                atomicMin(bx + 0, cell.x); atomicMax(bx + 1, cell.x);
                atomicMin(bx + 2, cell.y); atomicMax(bx + 3, cell.y);
            }
            s_cells[c] = cell;
        }
    }
    __syncthreads();
    if (tid < ngroups) {
        const int g0 = tid * YM_ST_GROUP;
        const int nvalid = max(0, min(min(YM_ST_GROUP, a.chunk - g0), nq - (i0 + g0)));
        int Xp0 = 0, Y0 = 0, H = 0, spr = 0, pitch = 48, use_lds = 0;
        if (nvalid > 0) {
            const int *bx = s_box + tid * 4;
            const int X0 = cx0 + bx[0] + 2 * G * xg_min, X1 = cx0 + bx[1] + 2 * G * xg_max + 2 * (G - 1);
            Y0 = cy0 + 2 * iy_min + bx[2];
            H = cy0 + 2 * iy_max + bx[3] - Y0 + 1;
            Xp0 = (X0 >> 1) & ~15;
            spr = ((((X1 >> 1) - Xp0 + 1) + 15) >> 4) + 1; // +1: the funnel reads one block ahead
            pitch = spr <= 3 ? 48 : spr <= 5 ? 80 : spr <= 7 ? 112 : 144;
            use_lds = (spr <= 8 && H <= 32 * YM_ST_NL && 2 * H * pitch <= YM_ST_REGION && nvalid >= YM_ST_GROUP / 2) ? 1 : 0;
        }
        int *r = s_rect + tid * 8;
        r[0] = Xp0; r[1] = Y0; r[2] = H; r[3] = spr; r[4] = pitch; r[5] = use_lds; r[6] = nvalid;
        if (a.stamps && nvalid > 0) { // development statistics: groups, staged groups, staged bytes
            atomicAdd(a.stamps + 29, 1ull);
            atomicAdd(a.stamps + 30, (unsigned long long)use_lds);
            atomicAdd(a.stamps + 31, (unsigned long long)(use_lds ? 2 * H * spr * 16 : 0));
        }
    }
    __syncthreads();
#define rect_of(S) st_rect_of(s_rect, ngroups, (S))
    YM_STAMP(a, 21);

    uint32_t acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0u;

    StCtx ctx;
    ctx.planes = planes; ctx.cells = s_cells; ctx.cx0 = cx0; ctx.cy0 = cy0; ctx.iy = iy; ctx.xg = xg; ctx.wave = wave; ctx.lane = lane; ctx.mode = a.tpb;
    ctx.plane_bytes = plane_bytes; ctx.half_pitch = half_pitch;
    if (st.regular[0]) {
        u32x4 v[2 * YM_ST_NL]; // the next group's rectangle, in flight while the current one is gathered
        StRect rc = rect_of(0);
        st_stage_load(v, rc, planes, plane_bytes, half_pitch);
        YM_ST_WAIT(v);
        st_stage_store(v, rc, region);
        __syncthreads();
        for (int g = 0; g < ngroups && rc.nvalid > 0; g++) {
            const StRect rn = rect_of(g + 1);
            if (a.tpb != 4) st_stage_load(v, rn, planes, plane_bytes, half_pitch);
            if (a.tpb != 3) st_gather(acc, ctx, g, rc, region + (g & 1) * YM_ST_REGION);
            YM_ST_WAIT(v);
            st_stage_store(v, rn, region + ((g + 1) & 1) * YM_ST_REGION);
            __syncthreads();
            rc = rn;
        }
#undef rect_of
        YM_STAMP(a, 22);
        // ---- add the four waves' packed 16-bit sums, wave 0 writes the partials
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        uint32_t *red = reinterpret_cast<uint32_t *>(region);
#pragma unroll
        for (int j = 0; j < 8; j++) red[(wave * 8 + j) * 64 + lane] = acc[j];
        __syncthreads();
        if (wave == 0 && active) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] = red[j * 64 + lane] + red[(8 + j) * 64 + lane] + red[(16 + j) * 64 + lane] + red[(24 + j) * 64 + lane];
            uint16_t *out = a.partial + (size_t)b * a.partial_stride +
                            (((size_t)chunk * a.lat.nt + k) * a.lat.ny + iy) * a.nx_pad + (size_t)xg * G;
            store_partial16(out, acc);
        }
        YM_STAMP(a, 23);
    } else {
        // hypothesis cells are not an exact lattice (possible only through fp rounding): per-cell path
        if (wave != 0 || !active) return;
        const uint8_t *__restrict__ grid = a.grid + (size_t)b * a.grid_stride;
        const unsigned limit = (unsigned)(a.g.pitch * a.g.win_w);
        const int32_t *cyp = cxp + a.dim_stride;
        uint16_t *out = a.partial + (size_t)b * a.partial_stride +
                        (((size_t)chunk * a.lat.nt + k) * a.lat.ny + iy) * a.nx_pad + (size_t)xg * G;
        const int n_here = min(nq - i0, a.chunk);
        for (int j = 0; j < G; j++) {
            const int ix = xg * G + j;
            unsigned sum = 0;
            if (ix < a.lat.nx) {
                const int base = cyp[iy] * a.g.pitch + cxp[ix];
                for (int i = 0; i < n_here; i++) {
                    const unsigned idx = (unsigned)(base + s_cells[i].x + s_cells[i].y * a.g.pitch);
                    sum += idx < limit ? grid[idx] : 0u;
                }
            }
            out[j] = (uint16_t)sum;
        }
    }
}

}  // namespace ym
