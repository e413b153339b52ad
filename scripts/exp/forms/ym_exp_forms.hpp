// ym_exp_forms.hpp -- the correlate forms that were built, measured bit-exact and SLOWER than what the product library runs, with the
// list builder they read.  Not part of libyagmatch.so: `make -C yag_slam_amd/csrc experimental` builds ../libyagmatch_exp.so with
// -DYM_EXPERIMENTAL and this directory on the include path (load it with YM_LIB_PATH=...; debug option 32 selects a form, the GPU tests
// run them when the library has them).  The studies that cite them: profiles/r04_region_study.md, profiles/r05_region_study.md.
//   ym_k_region_ws.hpp   bin_whole_kernel (round-5 lists: one block per query, all angles), correlate_region_ws_kernel (wave-specialised,
//                        option 32 = 2: 3.15 - 3.48 ms against 2.73), region_walk_kernel, region_percell_kernel
//   ym_k_item.hpp        correlate_item_kernel (one block per item, 32-bit sums in LDS, option 32 = 3: 4.72 ms), correlate_pool_kernel
//                        (two blocks per item, 16-bit sums in LDS, option 32 = 4: 4.45 ms)
//   ym_k_region2.hpp     correlate_region2_kernel<H> (sixteen waves per block, several per angle, option 32 = 5: 2.99 - 3.25 ms)
#pragma once
#ifndef YM_EXPERIMENTAL
#error "the experimental correlate forms are compiled only with -DYM_EXPERIMENTAL"
#endif
#include "ym_k_region_ws.hpp"
#include "ym_k_region2.hpp"
#include "ym_k_item.hpp"
