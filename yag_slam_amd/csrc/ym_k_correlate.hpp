// ym_k_correlate.hpp -- K4 correlate_kernel: the direct (global-load) coarse correlate.
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
    int32_t tpb;           // (unused)
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
        // ("yagpy" items proven regular: the lookup cell is the window cell hypothesis (0, 0) reads, ym_k_common.hpp)
        const bool yag = a.g.semantics == 1;
        const double add_x = yag ? st.ylat[0] : off_x, add_y = yag ? st.ylat[1] : off_y;
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
                    const int l = lookup_offset_sem(a.g, ql[i], cs.x, cs.y, off_x, off_y, add_x, add_y, a.g.pitch) + cx0;
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
            int o = i < nq ? lookup_offset_sem(a.g, ql[i], cs.x, cs.y, off_x, off_y, add_x, add_y, regular ? a.g.pitch : lin_pitch(a.g)) : 0;
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
                const int base = cyv * lin_pitch(a.g) + cx[ix];
                for (int i = 0; i < n_here; i++) sum += cell_value(a.g, grid, limit, (unsigned)(base + offs[i]));
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

}  // namespace ym
