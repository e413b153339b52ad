// ym_host_matcher.hpp -- host runtime: ym_map, ym_occupancy, ym_batch, ym_matcher (workspace, caches, options)
// Part of yagmatch.hip (included at file scope); not a header of its own.
struct ym_map {
    int device;
    int width, height;
    double *d_cgrid;  // the float correlation grid as the reference holds it
    uint8_t *d_g8;    // int(100 * cell): what scoring reads
};

struct ym_occupancy {
    int device;
    ym_occupancy_info info;
    std::vector<uint8_t> image; // [height][width], row 0 = lowest y
};

struct ym_batch {
    std::vector<const ym_scan *> queries; // the distinct query scans (one for ym_batch_create; ym_pairs_create: up to one per item)
    std::vector<int32_t> item_query;      // per item: its query's index in `queries`
    std::vector<const ym_scan *> scans;
    std::vector<int32_t> offsets;
    mutable std::vector<int> cache_hints; // per scan: its entry in the owning matcher's point cache (validated on use)
    mutable std::vector<int> query_hints; // the same per query
    uint64_t uid = 0; // unique per created batch
};

struct ym_matcher {
    ym_config cfg;
    int device;
    hipStream_t own_stream, stream;
    // the region path's pair lists (bin_kernel: ONE block per query of the call, ~85 us) need nothing of the raster: they are
    // built on a second stream next to tiles + raster and joined before the region kernel
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlap_lists = true;
    bool staged_queries = true;       // a synchronous match reads a just-created query scan from its staging slot instead of waiting
    int tile_h_forced = 0;            // tests: 32 or 64 rows per raster tile whatever the call
    int sticky_tall_left = 0;         // small calls that still take the tall tiles of the last large batch (plan_sizes)
    int tall_pattern = 0;             // 0: no large batch yet, 1: the last call of a large window was a large batch, 2: a small call followed it
    bool tall_alternates = false;     // the matcher has served a small call between two large batches
    int tall_tiles_min_window = 768;  // window width (cells) from which a batch of 512+ items gets 64-row tiles (256 items: 108 against 111 us of raster)
    YmGeom geom;                 // config part filled at create; window part per call
    std::vector<uint8_t> kernel; // Karto smear kernel (ksize x ksize)
    std::vector<double> kernel_f; // yagpy: the float kernel (helpers.py:86-97), for maps built from occupancy images
    DevBuf<double> kernel_f_dev;
    DevBuf<double2> map_pts;      // match against a map: the query point set
    int z2max = 0;               // largest squared cell distance whose kernel value is 100
    DevBuf<uint8_t> ktab;
    DevBuf<uint8_t> rowtab;   // the raster's row-pass tables (upload_lut)
    int n_rowtab = 0, rowtab_shift = -1;
    // workspace
    DevBuf<unsigned char> desc_dev; // batch call descriptors (single calls travel in the kernel arguments)
    DevBuf<YmItemState> states;
    DevBuf<double2> qlocal;    // [query slots][max_n] sensor-frame query points
    DevBuf<int32_t> qnp;       // [query slots]
    DevBuf<unsigned char> tmp_cache; // batches: per-call cache slots of base scans the point cache cannot hold
    DevBuf<int2> cells;
    DevBuf<int4> bbox;
    DevBuf<uint8_t> grid;
    DevBuf<uint8_t> planes;    // even/odd column planes of every window
    DevBuf<uint8_t> tile_zero; // per raster tile: window memory known to be zero (skips rewriting empty tiles)
    DevBuf<uint8_t> sub_zero;  // per raster tile that is not: which of its 8 x 8 sub-blocks are (8 bytes per tile)
    size_t tz_sig[6] = {0, 0, 0, 0, 0, 0}; // memory/geometry the flags are valid for
    // per workspace item: tile rectangle (x0, y0, x1, y1) outside which the item's window memory is known to be zero.
    // Items [0, tz_covered) have valid flags and rectangles; a call only rasterises (and cleans) items [0, B), so the
    // state of the items past B must survive it.
    std::vector<std::array<int, 4>> item_dirty;
    int tz_covered = 0;
    // Calls whose correlate stages from the row-major window write the window only (CallPlan::win_only); the column planes of the
    // items they touch then lag behind.  The knowledge above describes WINDOW memory and stays valid through such calls; what a later
    // call that reads the planes needs is every tile of ITS items written once more -- item by item, not the whole matcher (round 4
    // kept the mode in the signature: one single match between two batches of 4096 cost the second a full raster of all 4096 windows)
    std::vector<unsigned char> planes_stale;
    DevBuf<double2> ctrig;     // (cos, sin) per coarse angle
    DevBuf<int32_t> foffsets;  // fine lookup tables
    DevBuf<int32_t> hypcell;
    DevBuf<uint16_t> partial;  // per beam-chunk partial sums of the coarse lattice
    DevBuf<uint16_t> rg_entries; // region correlate: per query slot of a call the (beam, angle) pairs sorted by region
    DevBuf<int32_t> rg_starts;
    DevBuf<uint32_t> rg_rbox;    // per query slot, region and angle block: the box its patches read of the region
    DevBuf<uint32_t> rg_walk;    // per query slot and angle block: the walk of the wave-specialised region correlate (region_walk_kernel)
    int n_cus = 0;               // compute units of the device
    size_t bin_lds_limit = 64 * 1024; // dynamic LDS bin_whole_kernel may use so far (experimental builds)
    size_t binp_lds_limit = 64 * 1024; // ... and bin_kernel
    int chain_margin = 1;        // tiles (64 cells) added around the predicted raster rectangle of a chained step
    uint64_t cache_gen = 1;      // bumped whenever the point cache changes (entries created, re-posed, dropped) or an option is set
    int64_t seq_segments = 0, seq_faults = 0, seq_sync_steps = 0; // ym_map_sequence: chained segments, those cut short, synchronous steps
    // Batches below this size take the direct correlate kernel: a block of either LDS correlate walks all regions of its item,
    // ~190 us whatever the batch, while the direct kernel's time grows with the batch from ~15 us (measured, both lattices:
    // 8 chains 142 / 109 us against 220 / 225 per enqueue, 64 chains equal, 256 chains 675 / 759 against 485 / 452)
    int lds_min_batch = 64;
    int rg_min_batch = 48;  // the region correlate from this many items on (round 4: 48 items 180 -> 169 us, 56 items 196 -> 184; below 44 the direct kernel wins)
    int prepare_threads = 0;     // development: 512 = the single-item prepare kernel with 512 threads per scan too
    int last_wh = 0;             // half width of the previous call's device window (cells, before clamping)
    bool use_scan_structure = true; // base scans' trigger chains come from ym_scan_create's structure_kernel where that is exact
    bool poll_completion = true; // single matches: the host polls a pinned word instead of waiting for the stream event
    int corr_region_nw = 0;  // development: waves (= angles) per region-correlate block
    int item_min_batch = 1 << 30; // batches from this many items on take correlate_item_kernel
    bool item_lds_set = false;
    int raster_planes_only = 0; // timing experiment (option 36): the raster does not write the row-major window
    int raster_no_rowtab = 0;   // tests (option 37): the raster's row pass by bit scans instead of its tables
    int corr_region_rsplit = 0; // 0 = by batch size, 1 = never split an item's regions over blocks, n = always n blocks
    int tile_list_min_batch = 48; // option 40: batches from this size on get raster work lists (tiles_kernel)
    int keep_planes = 0;         // option 39: 1 = every call writes the column planes and the region correlate stages from them
    int corr_region_pad_lds = 0; // development (option 38): dynamic LDS bytes the region correlate is launched with and does not use (fewer blocks per CU)
    int corr_region_dbg = 0;  // development (timing only): 1 = the loader waves move nothing, 2 = the gather waves gather nothing
    int corr_region_form = 0; // 2 = the wave-specialised region correlate (gather waves + loader waves) instead of correlate_region_kernel
    // the pair lists of the last single-query call that built them: what they were built from.  A call with the same key finds them
    // in the list buffers and builds nothing (no bin_kernel, no second stream, no join) -- they depend on the query's readings and
    // pose, the window and the lattice alone, like the projected points the point cache keeps (round 5; option 45 = 0: off)
    struct ListKey {
        uint64_t qid; double pose[3]; YmGeom g; YmLattice lc;
        int32_t nw, parts, nrx, nry, rg_h, force, nregions, ng; size_t es, ss; const void *pe, *ps, *pb;
    };
    ListKey list_key;
    bool list_key_valid = false, list_cache_on = true;
    int64_t list_cache_hits = 0;
    int rg2_min_batch = 1 << 30; // batches from this many items on take correlate_region2_kernel (option 32 = 5: always where it can)
    int rg2_h = 128;          // option 43: class rows a region of correlate_region2_kernel owns (80, 100 or 128)
    size_t rg2_lds_limit = 0;
    int corr_fuse_score = 0; // tests: 2 = the region correlate never scores itself (score_kernel does)
    // gather correlate: per query slot of a call the (beam, angle) units sorted by region, the bin table, the work
    // lists and the counters they are built with; the lane -> (row, segment) table of the lattice
    DevBuf<uint32_t> ga_units;
    DevBuf<int32_t> ga_starts;
    DevBuf<int32_t> ga_work;
    DevBuf<uint32_t> ga_counters;
    DevBuf<uint32_t> ga_lane_job;
    std::vector<uint32_t> ga_lane_job_host; // what ga_lane_job holds
    size_t ga_lds_limit = 64 * 1024;        // dynamic LDS gather_kernel may use so far
    DevBuf<uint32_t> sums;     // coarse sums, then fine sums
    DevBuf<double> resp;
    DevBuf<double> blockmax;
    DevBuf<double> probs;
    // yagpy: the coarse pass's integer sums come from the production correlate kernels where the item's roundings provably form a
    // lattice (ym_k_yagpy.hpp, yag_lattice_kernel); option 46 = 0: every item through yag_score_kernel, the rule as written
    int yag_fast = 1;
    int last_corr_form = -1; // which coarse correlate the last call launched: 0 correlate_kernel, 1 correlate_region_kernel, 2 gather_kernel, -1 none
    DevBuf<unsigned long long> yag_counters; // [0] items through the production kernels, [1] fallbacks, [2] pairs checked exhaustively, [3] pairs that failed
    DevBuf<double> yaxes;      // yagpy: xvals, yvals, tvals per item
    DevBuf<double2> yrot;      // yagpy: points rotated per angle
    DevBuf<double> seq_pose;   // device-chained sequences: [0..2] the next step's odometry prior, [4 + 3k ..] the pose step k of the segment found
    DevBuf<int32_t> seq_fault; // ... and the first step the host has to repeat (0: none)
    PinnedBuf seq_results;     // ... and the result state of every step of a segment
    DevBuf<unsigned long long> stamps; // phase time stamps (development aid)
    bool stamps_on = false;
    int corr_u = 0;      // development: force the number of beams in flight per lane (16, 32, 48)
    int full_raster = 0; // development: launch every raster tile
    // point cache: world point readings + trigger chain of resident base scans, per (scan id, pose) -- what Karto's
    // LocalizedRangeScan keeps in m_PointReadings until the pose is set again.  One arena, bump-allocated; everything
    // that touches it runs on this matcher's stream, so recomputing a slot in place is ordered after its readers.
    struct CacheEntry { uint64_t id; size_t off; int n; double pose[3]; uint64_t stale_in_call; };
    uint64_t call_counter = 0;
    std::vector<CacheEntry> cache_entries;
    std::unordered_map<uint64_t, int> cache_index;
    DevBuf<unsigned char> cache_arena;
    size_t cache_used = 0;
    size_t cache_limit = (size_t)16 << 30; // bytes; beyond it the cache starts over
    int cache_off = 0;                     // development: 1 = never cache (every call projects every scan)
    int64_t cache_hits = 0, cache_misses = 0;
    DevBuf<unsigned> sel_scratch; // select on long chains: hash, states and neighbour lists in global memory
    DevBuf<unsigned> sel_tables;  // select on a few items (split form): hash keys and earliest-point table, zero between calls
    DevBuf<uint4> sel_rec;        // ... and the record per point
    DevBuf<unsigned> sel_slot;    // ... and the point's slot (between the hash and the neighbour launch)
    DevBuf<uint32_t> tile_list; // raster work list per item
    DevBuf<int32_t> tile_count;
    DevBuf<int32_t> tile_max;        // [1] longest raster work list of the call
    DevBuf<uint16_t> tile_hits;      // per entry of the work list: the chunks that reach its tile (YM_TILE_HITS slots)
    int32_t *tile_max_host = nullptr; // pinned: the raster kernel leaves that number here, the next call sizes its grid by it
    int finish_form = 0; // development: 1 = fine_kernel + final_kernel even on batches, 2 = finish_kernel always
    int corr_chunks = 0; // development: force the number of beam chunks of the correlate kernel
    int corr_pad_lds = 0; // development: extra dynamic LDS per correlate block (limits blocks per CU)
    int corr_cw = 0;      // development: force the chunk-waves per correlate block (1, 2, 4)
    int corr_dedup = 0;     // development / tests: 1 = always merge equal consecutive lookup offsets, 2 = never
    int corr_region = 0;    // tests: 1 = neither LDS correlate, 2 = their per-cell path, 3 = their "lists do not fit" path, 4 = the gather
                            // correlate also where the region correlate would run
    int corr_region_na = 0; // development: jobs (angle, lattice part) per wave of the gather correlate (1..4)
    int corr_region_parts = 0; // development / tests: blocks per item of the gather correlate (each takes a share of the angles)
    int corr_region_cap = 0;   // tests: units per LDS buffer (a multiple of 64; small values force chunked regions)
    int corr_region_lds = 0;   // development / tests: LDS bytes a gather block may use (small values force many regions)
    int raster_gx = 0;      // tests: raster blocks per item (0 = by the previous call's longest work list)
    int raster_hits_per_tile = 0; // tests: hit slots per entry of the raster's work list (0 = YM_TILE_HITS, -1 = no hit lists)
    int keep_sums = 0;      // development: keep the coarse integer sums of batches too (ym_debug_sums)
    int finish_threads = 0; // development: force the finish kernel's block size (256 / 1024)
    int select_global = 0; // development / tests: always evaluate the order-dependent smear rule with the global-memory kernel
    int select_split_max = 8; // items up to which the rule runs in its split form (tests: 0 = the one-block kernel always)
    DevBuf<double> tmp_ranges;   // device copy of ranges for the descriptor-based entry
    PinnedBuf tmp_ranges_host;
    Slot slots[kAsyncSlots + 1]; // last one serves the synchronous entry points
    // geometry of the last launched call (debug getters)
    YmGeom last_geom;
    YmLattice last_lat[2];
    int last_B = 0, last_max_n = 0, last_max_base = 0, last_nt_stride = 0, last_dim_stride = 0;
    size_t last_grid_stride = 0, last_sums_stride[2] = {0, 0};
    size_t sums_pass_offset[2] = {0, 0};
    bool last_valid = false;
    // profiling
    bool profiling = false;
    ProfEvents prof[3];
};
