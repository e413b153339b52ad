// ym_kernels.hpp -- gfx950 kernels of the correlative scan matcher (karto semantics).
//
// Pipeline of one call (B independent items; one item = one query scan vs one chain of base scans):
//   prepare_kernel    point readings, valid-point filter, grid cells, coarse lookup table
//                     (Karto LocalizedRangeScan::Update, ScanMatcher::FindValidPoints, AddScan,
//                      GridIndexLookup::ComputeOffsets)
//   raster_kernel     correlation-grid window with the Gaussian max-smear
//                     (ScanMatcher::AddScans + CorrelationGrid::SmearPoint)
//   correlate_kernel  integer gather-reduce over the coarse (x, y, theta) lattice, split over beam
//                     chunks into partial sums (CorrelateScan loops + GetResponse)
//   score_kernel      partial sums -> sums -> response (+ penalty), per-block maxima
//   finish_kernel     coarse arg-max / tie mean / positional covariance, then the whole fine pass
//                     (offsets, 3x3xN correlate, arg-max, angular covariance) in one block per item
//                     (CorrelateScan tail, ComputePositionalCovariance, ComputeAngularCovariance)
// All fp64 arithmetic is written operation-for-operation like oracle/ym_oracle.c and the library is
// compiled with -ffp-contract=off, so responses are bit-identical to the oracle's.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "ym_types.h"

// the kernels, one header per stage
#include "ym_k_common.hpp"
#include "ym_k_prepare.hpp"
#include "ym_k_raster.hpp"
#include "ym_k_correlate.hpp"
#include "ym_k_finish.hpp"
#include "ym_k_region.hpp"
#ifdef YM_EXPERIMENTAL // the correlate forms that lost (scripts/exp/forms; `make experimental` puts that directory on the include path)
#include "ym_exp_forms.hpp"
#endif
#include "ym_k_gather.hpp"
#include "ym_k_yagpy.hpp"
#include "ym_k_occupancy.hpp"
