// ym_k_raster.hpp -- K2 raster_kernel.
// Part of ym_kernels.hpp (include that, not this file).
#pragma once

namespace ym {

// ================================================================== K2 raster
struct RasterArgs {
    const int2 *cells;
    const int4 *bbox;     // [B][max_base][YM_N_BOXES(max_n)]
    const YmItemState *states;
    YmGeom g;
    uint8_t *grid;        // [B][win_w rows][pitch]
    size_t grid_stride;   // bytes per item
    uint8_t *planes;      // [B][2][win_w rows][pitch/2]: plane p holds columns 2*x+p of the window
    const uint8_t *lut;   // smear kernel value by squared cell distance: lut[dx*dx + dy*dy], 2*h*h + 1 entries
    int32_t max_n, max_base;
    uint8_t *tile_zero;   // [B][tiles_y][tiles_x]: 1 = this tile of the window memory is known to hold zeros
    uint8_t *sub_zero;    // [B][tiles_y][tiles_x][8]: of a tile that is NOT known to be zero, byte b bit g = the 8 x 8 cells of rows
                          // 8b .. 8b + 7, columns 8g .. 8g + 7 are known to hold zeros (written by the block that last rastered the tile)
    int32_t tiles_x, tiles_y; // full tiling of the window
    int32_t tile_x0, tile_y0; // first tile of the launched sub-grid (tiles outside it are known to be zero)
    int32_t ltx;              // tile columns of the launched sub-grid
    const uint32_t *tile_list; // [B][tile_cap] work list built by tiles_kernel (tile | 0x8000 = clear only | hits << 16, 0xffff = more
                               // than the tile's hit slots hold), or null: one block per sub-grid tile
    const int32_t *tile_count; // [B]
    int32_t tile_cap;
    int32_t first_overflow;    // OVERFLOW launch: the first list entry the main launch's grid did not reach
    const uint16_t *hits;      // [B][tile_cap][YM_TILE_HITS] per LIST ENTRY the chunks that reach its tile, as the index of the chunk's
                               // first cell in the item's cells (tiles_kernel), or null: the blocks walk the item's boxes
    int32_t lty, pad0;
    const int32_t *tile_max;   // longest work list of the call (tiles_kernel)
    int32_t *tile_max_host;    // pinned host word block (0, 0) copies it to: the next call sizes its grid by it
    int32_t planes_only, n_rowtab; // planes_only: timing experiment, 1 = the row-major window is not written; n_rowtab: tables STORED
    int32_t rowtab_shift, no_planes; // no_planes: 1 = the column planes are not written (a call whose correlate stages from the window)
                                   // rowtab_shift >= 0: the mirrored form (h <= 10) -- the 8 + 2h bits, shifted left by this much, sit in the middle
                                   // of 28 = four groups of seven; tables 0 and 1 serve groups 0 and 1 directly and groups 3 and 2
                                   // through the mirror image (index bit-reversed, the eight distances in reverse order): 2 KB of LDS
                                   // instead of 4.  -1: table j serves bits 7j .. 7j + 6
    // the row pass's tables (host, upload_lut; n_rowtab = ceil((8 + 2h) / 7) of them, 0 = none: h > 12): table j, indexed by the
    // seven bitmap bits 7j .. 7j + 6 of the 8 + 2h an 8-cell group sees, holds for each of the eight cells its distance to the
    // nearest of those bits that is set and within reach (127: none)
    const uint2 *rowtab;
    unsigned long long *stamps;
};

// grid (launched tiles in x, in y, B), 256 threads.  Each block owns one 64x32 tile of the window and
// writes every byte of it exactly once (so no separate clear pass exists; a tile that is empty now and
// whose memory is known to be zero from an earlier call is skipped).  Karto's SmearPoint
// max-stamps a (2h+1)^2 kernel at every occupied cell; the kernel value depends only on the squared
// cell distance and never grows with it (checked on the host when the matcher is created), so a
// cell's final value is lut[min squared distance to an occupied cell inside the (2h+1)^2 window]:
//   row pass   g(y, x)  = min |dx| <= h with cell (y, x+dx) occupied      (bit scans on a row bitmap)
//   column pass m(y, x) = min over |dy| <= h of dy^2 + g(y+dy, x)^2        (8 cells per lane)
// NT = 256: one tile row per thread, shortest latency (single match); NT = 128: two rows per thread, twice the
// blocks per CU -- the tiles with work are latency-bound, so a batch gains (raster 160 -> 140 us on 256 items)
// TH = 32 rows per tile, or 64: half the blocks, a sixth less halo (4096 items: raster 1.63 -> 1.44 ms), but longer
// blocks -- a single match loses 6 us, the loop lattice's small windows 3 % --, so the host picks per call.
// LISTED (batches, tiles_kernel ran): block x of an item takes entry x of the item's work list.  What a block needs to start --
// its entry and the first 32 of the tile's hit slots -- sits at addresses that depend on (item, x) alone, so ONE memory round
// trip brings it all, the second one the cells (round 4; before: list entry -> the tile's hit_start pair -> its hits -> LDS ->
// cells, four dependent round trips, and the raster was bound by exactly that chain times the 11 blocks a CU held: its two
// passes are 9 % of its time).  No hit staging in LDS either: 10 KB per block, 15 blocks per CU.
template <int NT, bool OVERFLOW, int TH, bool LISTED>
__global__ __launch_bounds__(NT) void raster_kernel(RasterArgs a) {
    constexpr int TW = YM_TILE_W, HM = YM_MAX_KERNEL_HALF;
    constexpr int MAXHITS = LISTED ? 1 : 256 * TH / 32; // chunk boxes an unlisted block lists for its tile before it walks all boxes
    constexpr int RW = YM_RASTER_RW;                   // 64-bit words per bitmap row, + 1 so a funnel read never leaves the row
    constexpr int LPR = TW / 8;                        // lanes per tile row (8 cells each)
    static_assert(RW == (TW + 2 * HM + 63) / 64 + 1, "YM_RASTER_RW");
    // dynamic LDS (YM_RASTER_LDS_BYTES: sized by the kernel half in use, not by the largest one -- 11 KB on the usual kernels):
    // grow | occ | row tables | lut; the finished tile (`outb`) takes the place of occ and the tables once the row pass is done.
    // Rows of grow / outb are GP = 72 bytes apart: the column pass reads and writes them with lane = row.
    constexpr int GP = YM_RASTER_GP;
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_dyn[];
    const int OHh = TH + 2 * a.g.half_kernel;
    unsigned char *grow = rs_dyn;                                                                 // [OH * GP]
    unsigned long long *occ = reinterpret_cast<unsigned long long *>(rs_dyn + (size_t)OHh * GP);  // [OH * RW]
    uint2 *rtab = reinterpret_cast<uint2 *>(occ + (size_t)OHh * RW);                              // [n_rowtab * 128]
    unsigned char *outb = reinterpret_cast<unsigned char *>(occ);                                 // [TH * GP]
    unsigned char *lut = rs_dyn + YM_RASTER_LDS_BYTES(TH, a.g.half_kernel, a.n_rowtab) - (size_t)((2 * a.g.half_kernel * a.g.half_kernel + 2 + 15) / 16 * 16); // [2 h h + 2]
    __shared__ unsigned colany[TW / 8][4]; // per 8-cell column group: bit ry = the row pass found a wall within reach in halo row ry
    __shared__ int s_hits[MAXHITS], s_left[MAXHITS];
    __shared__ int s_nhits;
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    // the row tables leave first (3 KB per block, from L2); they are put into LDS when the block's bitmap is cleared
    constexpr int TPT = (5 * 128 + NT - 1) / NT; // table entries per thread (at most five tables)
    uint2 tq[TPT];
#pragma unroll
    for (int u = 0; u < TPT; u++) tq[u] = (tid + u * NT) < a.n_rowtab * 128 ? a.rowtab[tid + u * NT] : make_uint2(0u, 0u);
    // grid (x, B).  With a work list (batches) block i takes entry i of its item's list; without one (a few items: one
    // more launch would cost more than it saves) block i is tile i of the sub-grid and finds out by itself whether any
    // chunk box reaches it.
    // (hq[u]: hit slot tid / 16 + u * (NT / 16) of the entry, loaded by the caller together with the entry)
    auto one_tile = [&](unsigned entry, int bi, const uint16_t *hl, const unsigned (&hq)[4]) { // (bi = b; opaque to the optimiser in the list loop, see below)
    const int tile = (int)(entry & 0x7fffu);
    const int h = a.g.half_kernel;
    const int OW = TW + 2 * h, OH = TH + 2 * h;
    const int tiy = tile / a.tiles_x, tix = tile - tiy * a.tiles_x;
    const int tx0 = tix * TW, ty0 = tiy * TH;
    YM_STAMP(a, 4);
    // candidate chunks: YM_BOX_CELLS consecutive cells of one base scan whose bounding box touches tile + halo
    const int n_cchunks = YM_N_BOXES(a.max_n);
    const int n_boxes = a.max_base * n_cchunks;
    const int4 *bbox = a.bbox + (size_t)bi * n_boxes;
    const int lo_x = tx0 - h, hi_x = tx0 + TW + h - 1, lo_y = ty0 - h, hi_y = ty0 + TH + h - 1;
    uint8_t *grid = a.grid + (size_t)bi * a.grid_stride;
    // thread -> 8 consecutive cells (x8 ..) of tile rows y0, y0 + NT / LPR, ...
    const int y0 = tid / LPR, x8 = (tid % LPR) * 8;
    const size_t plane_bytes = (size_t)(a.g.pitch / 2) * a.g.win_w;
    uint8_t *planes = a.planes + (size_t)bi * a.grid_stride;
    // the window row-major and its even / odd column planes (v_perm_b32 byte gathers) for 8 cells of tile row y
    // (offsets inside an item's window as 32-bit numbers -- the host refuses windows beyond 2 GB --: a 64-bit multiply-add per
    //  address issues at a quarter of the rate)
    auto store8 = [&](int y, uint32_t p0, uint32_t p1) {
        if (ty0 + y < a.g.win_w) {
            const uint32_t row = (uint32_t)(ty0 + y), col = (uint32_t)(tx0 + x8);
            // (24-bit multiplies: a 32-bit v_mul_lo_u32 issues at a quarter of the rate)
            if (!a.planes_only) *reinterpret_cast<uint2 *>(grid + (__umul24(row, (uint32_t)a.g.pitch) + col)) = make_uint2(p0, p1);
            if (!a.no_planes) {
                uint8_t *pl = planes + (__umul24(row, (uint32_t)(a.g.pitch / 2)) + col / 2u);
                *reinterpret_cast<uint32_t *>(pl) = __builtin_amdgcn_perm(p1, p0, 0x06040200u);
                *reinterpret_cast<uint32_t *>(pl + plane_bytes) = __builtin_amdgcn_perm(p1, p0, 0x07050301u);
            }
        }
    };
    // What is known of the tile's memory (round 4): a tile is either known to be zero, or every 8 x 8 sub-block of it says so
    // for itself.  A sub-block that is zero now and was zero before is not stored again: walls are thin, so that is half of a
    // wall tile's bytes -- and the raster's time on a box whose memory takes writes slowly is its stores.
    static_assert(LPR == 8 && (NT / LPR) % 8 == 0, "a wave stores whole bands of eight rows");
    uint8_t *tz = a.tile_zero + ((size_t)bi * a.tiles_y + tiy) * a.tiles_x + tix;
    uint8_t *sz = a.sub_zero + (((size_t)bi * a.tiles_y + tiy) * a.tiles_x + tix) * 8;
    const unsigned tz_old = *tz;
    const uint2 sz_old = *reinterpret_cast<const uint2 *>(sz);
    const unsigned long long sz_old64 = (unsigned long long)sz_old.x | (unsigned long long)sz_old.y << 32;
    auto old_zero = [&](int y) { // bit g: the band of row y, column group g, is known to hold zeros
        // (one 64-bit shift: picking .x or .y by the row sent the pair to scratch)
        return tz_old ? 0xffu : (unsigned)(sz_old64 >> (y & 56)) & 0xffu;
    };
    // the rows y (one band per wave: lane / 8 = row of the band, lane % 8 = column group) with `flags`: the band's new knowledge
    auto store_band = [&](int y, uint32_t p0, uint32_t p1, bool flags) {
        const unsigned long long nz = __ballot((p0 | p1) != 0u);
        unsigned long long f = nz | (nz >> 32);
        f |= f >> 16;
        f |= f >> 8;
        const unsigned nzb = (unsigned)f & 0xffu, oldb = old_zero(y), g = (unsigned)tid & 7u;
        if (((nzb >> g) & 1u) || !((oldb >> g) & 1u)) store8(y, p0, p1);
        if (flags && (tid & 63) == 0) sz[y >> 3] = (uint8_t)(~nzb);
    };
    auto zero_tile = [&]() {
        for (int y = y0; y < TH; y += NT / LPR) store_band(y, 0u, 0u, false);
    };
    if (entry & 0x8000u) { // no chunk reaches this tile, but its memory still holds an earlier call's bytes
        zero_tile();
        __syncthreads(); // (every wave has read what was known of the tile before thread 0 says "all zero": a wave that read the
                         //  new flag instead would skip its rows -- seen with many blocks in flight)
        if (tid == 0) *tz = 1;
        return;
    }
    // LISTED: the entry says how many chunks reach the tile; their first cells are in the entry's hit slots
    const int hn = LISTED ? ((a.hits && (entry >> 16) != 0xffffu) ? (int)(entry >> 16) : -1) : -1; // -1: walk the boxes
    if constexpr (!LISTED) {
    if (tid == 0) s_nhits = 0;
    }
    int4 bb_first = make_int4(INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN); // this thread's first box: kept for the second pass
    if constexpr (!LISTED) { // decide "no box at all" before touching LDS
        int my_hits = 0;
        for (int c = tid; c < n_boxes; c += NT) {
            const int4 bb = bbox[c];
            if (c == tid) bb_first = bb;
            my_hits += (bb.x <= hi_x && bb.z >= lo_x && bb.y <= hi_y && bb.w >= lo_y) ? 1 : 0;
        }
        if (__syncthreads_or(my_hits) == 0) {
            // empty tile: zeros -- unless this memory is already known to be zero from an earlier call
            if (tz_old == 0) {
                zero_tile();
                __syncthreads();
                if (tid == 0) *tz = 1;
            }
            return;
        }
        // chunks whose box touches tile + halo, compacted so that the cell loads of several chunks are in flight together
        for (int c = tid; c < n_boxes; c += NT) {
            const int4 bb = c == tid ? bb_first : bbox[c]; // (a single match: 170 boxes, one per thread, no second load)
            if (bb.x <= hi_x && bb.z >= lo_x && bb.y <= hi_y && bb.w >= lo_y) {
                const int at = atomicAdd(&s_nhits, 1);
                if (at < MAXHITS) { // the chunk's first cell and how many it holds (the division once per hit, not per cell)
                    const int slot = c / n_cchunks, first = (c - slot * n_cchunks) * YM_BOX_CELLS;
                    s_hits[at] = slot * a.max_n + first;
                    s_left[at] = a.max_n - first;
                }
            }
        }
    }
    // LISTED: the cells of the first NT * 4 work items (hit slot, cell of the chunk) leave now, before the LDS tables are cleared
    const int2 *cells = a.cells + (size_t)bi * a.max_base * a.max_n;
    const int n_cells = a.max_base * a.max_n;
    int2 cc0[4];
    if constexpr (LISTED) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int hi = tid / YM_BOX_CELLS + u * (NT / YM_BOX_CELLS);
            // (a chunk's 16 cells may run past its scan's last reading into the next scan's first ones: the same item's occupied
            //  cells, stamped again -- harmless; past the item's last cell they would be another item's)
            const int ci = (int)hq[u] + (tid % YM_BOX_CELLS);
            cc0[u] = make_int2(YM_CELL_NONE, YM_CELL_NONE);
            if (hi < hn && ci < n_cells) cc0[u] = cells[ci];
        }
    }
    for (int i = tid; i < OH * RW; i += NT) occ[i] = 0ull;
    if (tid < (TW / 8) * 4) (&colany[0][0])[tid] = 0u;
#pragma unroll
    for (int u = 0; u < TPT; u++) // (per tile: the finished tile overwrites them)
        if ((tid + u * NT) < a.n_rowtab * 128) rtab[tid + u * NT] = tq[u];
    for (int i = tid; i <= 2 * h * h + 1; i += NT) lut[i] = i <= 2 * h * h ? a.lut[i] : (unsigned char)0;
    __syncthreads();
    unsigned *occ32 = reinterpret_cast<unsigned *>(occ);
    int any = 0;
    static_assert(RW == 3, "row_words");
    // (ly * RW * 2 + word as ONE 24-bit multiply-add: hipcc folds "* 6" and the byte scaling into a quarter-rate v_mul_lo_u32 by 24)
    auto row_words = [](int ly, int word) { int r; asm("v_mad_u32_u24 %0, %1, 6, %2" : "=v"(r) : "v"(ly), "v"(word)); return r; };
    auto stamp = [&](const int2 c2) {
        const int lx = c2.x - lo_x, ly = c2.y - lo_y;
        if (c2.x != YM_CELL_NONE && lx >= 0 && lx < OW && ly >= 0 && ly < OH) {
            atomicOr(&occ32[row_words(ly, lx >> 5)], 1u << (lx & 31));
            any = 1;
        }
    };
    auto walk_boxes = [&]() { // every box of the item, the cells of those that touch tile + halo
        for (int c0 = 0; c0 < n_boxes; c0 += NT) {
            const int c = c0 + tid;
            bool hit = false;
            if (c < n_boxes) {
                const int4 bb = bbox[c];
                hit = bb.x <= hi_x && bb.z >= lo_x && bb.y <= hi_y && bb.w >= lo_y;
            }
            unsigned long long mask = __ballot(hit);
            while (mask) {
                const int bit = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int chunk = c0 + (tid & ~63) + bit; // wave-uniform
                const int slot = chunk / n_cchunks, ci = chunk - slot * n_cchunks;
                const int i = ci * YM_BOX_CELLS + (tid & 63);
                if ((tid & 63) < YM_BOX_CELLS && i < a.max_n) stamp(cells[(size_t)slot * a.max_n + i]);
            }
        }
    };
    int nhits = 0;
    if constexpr (LISTED) {
        nhits = hn;
        if (hn >= 0) {
#pragma unroll
            for (int u = 0; u < 4; u++) stamp(cc0[u]);
            // (a tile more than NT / 4 chunks reach: the remaining hit slots, then their cells -- two more round trips)
            for (int h0 = NT / 4; h0 < hn; h0 += NT / 4) {
                int2 cc[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int hi = h0 + tid / YM_BOX_CELLS + u * (NT / YM_BOX_CELLS);
                    cc[u] = make_int2(YM_CELL_NONE, YM_CELL_NONE);
                    if (hi < hn) {
                        const int ci = (int)hl[hi] + (tid % YM_BOX_CELLS);
                        if (ci < n_cells) cc[u] = cells[ci];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) stamp(cc[u]);
            }
        } else {
            walk_boxes();
        }
    } else {
    nhits = s_nhits;
    if (nhits <= MAXHITS) {
        // work item = (hit chunk, cell of the chunk); 4 items per thread in flight
        const int nwork = nhits * YM_BOX_CELLS;
        for (int w0 = 0; w0 < nwork; w0 += 4 * NT) {
            int2 cc[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int w = w0 + u * NT + tid;
                cc[u] = make_int2(YM_CELL_NONE, YM_CELL_NONE);
                if (w < nwork && (w % YM_BOX_CELLS) < s_left[w / YM_BOX_CELLS]) cc[u] = cells[s_hits[w / YM_BOX_CELLS] + (w % YM_BOX_CELLS)];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) stamp(cc[u]);
        }
    } else {
        // more hit chunks than the list holds: walk every box (rare)
        walk_boxes();
    }
    }
    any = __syncthreads_or(any);
    YM_STAMP(a, 5);
    if (a.stamps && tid == 0) { // development statistics: listed tiles with work, those that hold a cell in reach, hit chunks
        atomicAdd(a.stamps + 27, 1ull);
        atomicAdd(a.stamps + 28, any ? 1ull : 0ull);
        atomicAdd(a.stamps + 29, (unsigned long long)nhits);
    }
    if (!any) {
        if (tz_old == 0) {
            zero_tile();
            __syncthreads();
            if (tid == 0) *tz = 1;
        }
        return;
    }
    if (tid == 0) *tz = 0;
    // row pass: nearest occupied |dx| <= h, 255 = none.  Bit x+h of a bitmap row is tile column x.
    // Work item = 8 consecutive cells of one (halo) row.  Walls are thin: most 8-cell groups see no bit within
    // reach at all and leave after one test.
    if (a.n_rowtab) {
        // the usual kernels (h <= 12), round 4: the 8 + 2h bits a group sees, seven at a time, index tables that hold the eight
        // cells' distances to the nearest set bit among those seven (127: none in reach); the group's distances are the
        // byte-wise minimum of the n_rowtab entries -- ~36 vector instructions per group instead of ~110 for eight bit scans
        // to the left and to the right.  Byte-wise minimum of values < 128: the borrow-free difference (a | 0x80) - b keeps
        // bit 7 of a byte exactly where a >= b.
        const unsigned long long gmask = (1ull << (2 * h + 8)) - 1ull;
        // byte-wise minimum of two entries (x |= min): v_min_u32 with byte selects (SDWA), each replacing one byte of its result;
        // the two dwords of an entry alternate, because an SDWA write that preserves the rest of its register must not follow
        // the instruction that wrote that register (one wait state; scripts/exp/sdwa_min.hip: a single chain of four is wrong in
        // byte 1).  (The portable form -- the borrow-free difference (x | 0x80) - y keeps bit 7 of a byte where x >= y, spread to a
        // mask, select -- came out of hipcc as nine instructions per dword with a quarter-rate v_mul_lo_u32 by 255 among them.)
        auto bmin2 = [](uint2 &x, const uint2 y) {
#define YM_BM(k) "v_min_u32_sdwa %0, %0, %2 dst_sel:BYTE_" #k " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #k " src1_sel:BYTE_" #k "\n\t" \
                 "v_min_u32_sdwa %1, %1, %3 dst_sel:BYTE_" #k " dst_unused:UNUSED_PRESERVE src0_sel:BYTE_" #k " src1_sel:BYTE_" #k "\n\t"
            asm(YM_BM(0) YM_BM(1) YM_BM(2) YM_BM(3) : "+v"(x.x), "+v"(x.y) : "v"(y.x), "v"(y.y));
#undef YM_BM
        };
        const int ntab = a.n_rowtab, mshift = a.rowtab_shift;
        for (int i = tid; i < OH * LPR; i += NT) {
            const int ry = i / LPR, rx = (i % LPR) * 8;
            // (the 8 + 2h <= 32 bits from bit rx on: one 32-bit funnel shift over two dwords of the row, not two 64-bit shifts)
            const unsigned *rw = occ32 + ry * (RW * 2) + (rx >> 5);
            const unsigned sw = __builtin_amdgcn_alignbit(rw[1], rw[0], (unsigned)rx & 31u) & (unsigned)gmask; // bits rx .. rx + 7 + 2h
            if (sw) { // (a group without a wall in reach writes nothing: the column pass only reads flagged rows)
                uint2 g;
                if (mshift >= 0) { // (block-uniform)
                    const unsigned s4 = sw << mshift;                       // 28 bits: groups 0 .. 3
                    const unsigned r4 = __builtin_bitreverse32(s4) >> 4;    // the same 28 bits mirrored: group 3 first
                    g = rtab[s4 & 127u];
                    bmin2(g, rtab[128 + ((s4 >> 7) & 127u)]);
                    uint2 m = rtab[r4 & 127u];                              // groups 3 and 2, cells in reverse order
                    bmin2(m, rtab[128 + ((r4 >> 7) & 127u)]);
                    bmin2(g, make_uint2(__builtin_amdgcn_perm(0u, m.y, 0x00010203u), __builtin_amdgcn_perm(0u, m.x, 0x00010203u)));
                } else {
                    g = rtab[sw & 127u];
                    for (int j = 1; j < ntab; j++) bmin2(g, rtab[j * 128 + ((sw >> (7 * j)) & 127u)]);
                }
                *reinterpret_cast<uint2 *>(&grow[ry * GP + rx]) = g;
                atomicOr(&colany[i % LPR][ry >> 5], 1u << (ry & 31));
            }
        }
    } else if (2 * h + 8 <= 32) {
        // (kept for a matcher without tables) the 8 cells of a group see 2h + 8 <= 32 bits of the bitmap row, 32-bit bit scans
        const unsigned wmask = (1u << (2 * h + 1)) - 1u, lmask = (1u << h) - 1u;
        const unsigned long long gmask = (1ull << (2 * h + 8)) - 1ull;
        for (int i = tid; i < OH * LPR; i += NT) {
            const int ry = i / LPR, rx = (i % LPR) * 8;
            const int w = rx >> 6, sft = rx & 63;
            const unsigned long long lo = occ[ry * RW + w], hi = occ[ry * RW + w + 1];
            const unsigned sw = (unsigned)((sft ? ((lo >> sft) | (hi << (64 - sft))) : lo) & gmask); // bits rx .. rx + 7 + 2h
            if (sw) { // (a group without a wall in reach writes nothing: the column pass only reads flagged rows)
                uint32_t out[2] = {0u, 0u};
#pragma unroll
                for (int q = 0; q < 8; q++) { // (no branch per cell: hipcc turns each into a saveexec / branch pair)
                    const unsigned win = (sw >> q) & wmask;
                    const unsigned right = win >> h, left = win & lmask;
                    const unsigned dr = right ? (unsigned)(__ffs((int)right) - 1) : 255u;
                    const unsigned dl = left ? (unsigned)(h - 31 + __clz((int)left)) : 255u;
                    out[q >> 2] |= (dr < dl ? dr : dl) << (8 * (q & 3));
                }
                *reinterpret_cast<uint2 *>(&grow[ry * GP + rx]) = make_uint2(out[0], out[1]);
                atomicOr(&colany[i % LPR][ry >> 5], 1u << (ry & 31));
            }
        }
    } else {
        const unsigned long long wmask = (1ull << (2 * h + 1)) - 1ull, lmask = (1ull << h) - 1ull;
        const unsigned long long gmask = (1ull << (2 * h + 8)) - 1ull;
        for (int i = tid; i < OH * LPR; i += NT) {
            const int ry = i / LPR, rx = (i % LPR) * 8;
            const int w = rx >> 6, sft = rx & 63;
            const unsigned long long lo = occ[ry * RW + w], hi = occ[ry * RW + w + 1];
            const unsigned long long sw = (sft ? ((lo >> sft) | (hi << (64 - sft))) : lo) & gmask; // bits rx .. rx + 7 + 2h
            if (sw) {
                uint32_t out[2] = {0xffffffffu, 0xffffffffu};
    #pragma unroll
                for (int q = 0; q < 8; q++) {
                    const unsigned long long win = (sw >> q) & wmask;
                    if (win) {
                        const unsigned long long right = win >> h, left = win & lmask;
                        const int dr = right ? (__ffsll((long long)right) - 1) : 255;
                        const int dl = left ? (h - 63 + __clzll((long long)left)) : 255;
                        const unsigned g = (unsigned)(dr < dl ? dr : dl);
                        out[q >> 2] = (out[q >> 2] & ~(0xffu << (8 * (q & 3)))) | (g << (8 * (q & 3)));
                    }
                }
                *reinterpret_cast<uint2 *>(&grow[ry * GP + rx]) = make_uint2(out[0], out[1]);
                atomicOr(&colany[i % LPR][ry >> 5], 1u << (ry & 31));
            }
        }
    }
    __syncthreads();
    YM_STAMP(a, 6);
    // column pass: 8 cells per lane as four pairs of 16-bit lanes (cells 0|2, 1|3, 4|6, 5|7): the candidate
    // g*g + dy*dy is at most 255^2 + h^2 < 65536, so one v_pk_mad_u16 + one v_pk_min_u16 serve two cells.
    // Only the halo rows in which the row pass found a wall within reach of this column group are visited (bit scan of
    // the group's row mask): walls are thin, most (row, group) pairs have none and write zeros at once.
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const unsigned max_d2 = (unsigned)(2 * h * h);
    const unsigned long long tapmask = (1ull << (2 * h + 1)) - 1ull;
    // Round 4 (second half): a wave takes the rows of ONE column group (two groups of a 32-row tile), lane = row, not eight rows of
    // all eight groups: along a wall that crosses the rows every lane of a group's wave has the same taps and the waves of the
    // groups out of reach have none (a wave costs what its busiest lane costs).  The finished bytes go through LDS so that
    // the stores stay row-wise (64 contiguous bytes of a window row per 8 lanes, sub-block knowledge per band).
    {
        constexpr int RPW = TH >= 64 ? 64 : TH, GPW = 64 / RPW; // rows and column groups per wave
        static_assert(TH % RPW == 0 && TH <= 64 && (TH * LPR) % NT == 0, "a wave takes whole column groups of the tile");
        const int wv = tid >> 6, ln = tid & 63;
        for (int it = 0; it < (TH * LPR) / NT; it++) {
            const int grp = (it * (NT / 64) + wv) * GPW + ln / RPW, y = ln % RPW, gx8 = grp * 8;
            const unsigned *ca = colany[grp];
            const unsigned long long ca_lo = (unsigned long long)ca[0] | (unsigned long long)ca[1] << 32, ca_hi = (unsigned long long)ca[2] | (unsigned long long)ca[3] << 32; // (y + 2h <= TH - 1 + 40 < 128)
            us2 mn2[4];
#pragma unroll
            for (int q = 0; q < 4; q++) mn2[q] = (us2){0xffff, 0xffff};
            // bit t: halo row y + t, dy = t - h  (the usual kernels: a 32-bit funnel shift over two dwords of the group's row flags)
            const unsigned long long m64 = 2 * h + 1 <= 32 ? (unsigned long long)(__builtin_amdgcn_alignbit(ca[(y >> 5) + 1], ca[y >> 5], (unsigned)y & 31u) & (unsigned)tapmask)
                                                           : (y ? (ca_lo >> y) | (ca_hi << (64 - y)) : ca_lo) & tapmask;
            uint32_t packed[2] = {0u, 0u};
            if (m64 != 0ull) { // (else: no wall in reach of these eight cells)
                auto tap = [&](int t) {
                    const int dy = t - h;
                    const uint2 gg = *reinterpret_cast<const uint2 *>(&grow[(y + t) * GP + gx8]);
                    int d2i;
                    asm("v_mul_i32_i24 %0, %1, %1" : "=v"(d2i) : "v"(dy)); // (asm: hipcc widens __mul24 of a small value to a quarter-rate v_mul_lo_u32)
                    const unsigned short d2 = (unsigned short)d2i;
                    const us2 dd = (us2){d2, d2};
                    const uint32_t u[4] = {gg.x & 0x00ff00ffu, (gg.x >> 8) & 0x00ff00ffu, gg.y & 0x00ff00ffu, (gg.y >> 8) & 0x00ff00ffu};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        us2 gq;
                        __builtin_memcpy(&gq, &u[q], 4);
                        mn2[q] = __builtin_elementwise_min(mn2[q], (us2)(gq * gq + dd)); // g = 255 or 127 (none) is larger than any real distance
                    }
                };
                if (2 * h + 1 <= 32) { // (block-uniform; the usual kernels: the tap mask is a 32-bit word, its scan half the instructions)
                    unsigned m = (unsigned)m64;
                    while (m) {
                        const int t = __ffs((int)m) - 1;
                        m &= m - 1u;
                        tap(t);
                    }
                } else {
                    unsigned long long m = m64;
                    while (m) {
                        const int t = __ffsll((long long)m) - 1;
                        m &= m - 1ull;
                        tap(t);
                    }
                }
                // (lut[max_d2 + 1] = 0: anything farther is clamped to that index, two cells per v_pk_min_u16, no branch per cell)
                const unsigned short cap = (unsigned short)(max_d2 + 1u);
#pragma unroll
                for (int q = 0; q < 4; q++) mn2[q] = __builtin_elementwise_min(mn2[q], (us2){cap, cap});
                unsigned mn[8];
                mn[0] = mn2[0].x; mn[2] = mn2[0].y; mn[1] = mn2[1].x; mn[3] = mn2[1].y;
                mn[4] = mn2[2].x; mn[6] = mn2[2].y; mn[5] = mn2[3].x; mn[7] = mn2[3].y;
#pragma unroll
                for (int q = 0; q < 8; q++) packed[q >> 2] |= (uint32_t)lut[mn[q]] << (8 * (q & 3));
            }
            *reinterpret_cast<uint2 *>(&outb[y * GP + gx8]) = make_uint2(packed[0], packed[1]);
        }
    }
    __syncthreads();
    for (int y = y0; y < TH; y += NT / LPR) {
        const uint2 v = *reinterpret_cast<const uint2 *>(&outb[y * GP + x8]);
        store_band(y, v.x, v.y, true);
    }
    YM_STAMP(a, 7);
    }; // one_tile
    if constexpr (LISTED) {
        // The grid's x size is a guess (the longest list of the previous call + 1/8, yagmatch.hip): launching one block per
        // tile of the sub-grid cost 80 us per 1024 items in blocks that only found the list exhausted.  Block x takes entry
        // x; what a longer list holds beyond the grid is done by the OVERFLOW instantiation, a second launch of a few blocks
        // per item that walk the rest (its loop keeps every per-item constant in registers: 106 SGPRs, 9 spilled -- not
        // something the one-tile-per-block kernel should pay for).
        // An entry and its first hit slots are loaded together with the item's count, whatever the count says (the list
        // has tile_cap >= gridDim.x slots per item): one round trip, not two.
        const uint32_t *list = a.tile_list + (size_t)b * a.tile_cap;
        const uint16_t *hits = a.hits ? a.hits + (size_t)b * a.tile_cap * YM_TILE_HITS : nullptr;
        auto load_entry = [&](int e, unsigned &entry, const uint16_t *&hl, unsigned (&hq)[4]) {
            entry = list[e];
            hl = hits ? hits + (size_t)e * YM_TILE_HITS : nullptr;
#pragma unroll
            for (int u = 0; u < 4; u++) hq[u] = hl ? (unsigned)hl[tid / YM_BOX_CELLS + u * (NT / YM_BOX_CELLS)] : 0u;
        };
        static_assert(NT / 4 <= YM_TILE_HITS, "the first round of hit slots must exist");
        if (!OVERFLOW) {
            unsigned entry, hq[4];
            const uint16_t *hl;
            load_entry((int)blockIdx.x, entry, hl, hq);
            const int count = a.tile_count[b];
            if (blockIdx.x == 0 && b == 0 && tid == 0 && a.tile_max_host) *a.tile_max_host = *a.tile_max;
            if ((int)blockIdx.x < count) one_tile(entry, b, hl, hq);
        } else {
            const int count = a.tile_count[b];
            for (int e = a.first_overflow + (int)blockIdx.x; e < count; e += (int)gridDim.x) {
                unsigned entry, hq[4];
                const uint16_t *hl;
                load_entry(e, entry, hl, hq);
                one_tile(entry, b, hl, hq);
                __syncthreads(); // (the LDS tables are reused)
            }
        }
    } else {
        const int sy = (int)blockIdx.x / a.ltx, sx = (int)blockIdx.x - sy * a.ltx;
        // rotate the tile column by the row: a sub-grid width that is a multiple of 8 would otherwise pin every
        // tile column (i.e. every wall) to one XCD
        const unsigned hq[4] = {0u, 0u, 0u, 0u};
        one_tile((unsigned)((a.tile_y0 + sy) * a.tiles_x + a.tile_x0 + (sx + 3 * sy + 5 * b) % a.ltx), b, nullptr, hq);
    }
}

}  // namespace ym
