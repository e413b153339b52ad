// yagmatch.hip -- host runtime + C ABI of libyagmatch.so (see include/yagmatch.h).
//
// One ym_matcher = one HIP stream + one device workspace.  A call (B items) is: one H2D copy of
// the call descriptor, a fixed sequence of kernel launches (ym_kernels.hpp), one D2H copy of the
// per-item result states.  Everything between stays in HBM.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <ctime>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/yagmatch.h"
#include "ym_kernels.hpp"

// The host runtime in units (one translation unit: the kernels are templates and inline functions of ym_kernels.hpp):
#include "ym_host_types.hpp"    // error text, device guard, buffers, Call / CallPlan / Slot
#include "ym_host_pool.hpp"     // ym_scan, the per-device scan pool
#include "ym_host_matcher.hpp"  // ym_map, ym_occupancy, ym_batch, ym_matcher

namespace {
#include "ym_host_setup.hpp"    // config -> geometry, tables, lattices, profiling events
#include "ym_host_plan.hpp"     // the planner of one call
#include "ym_host_enqueue.hpp"  // the launches of one call
#include "ym_host_call.hpp"     // launch, collect, scans -> call descriptors
}  // namespace

// =================================================================== C ABI (include/yagmatch.h)
extern "C" {
#include "ym_abi_matcher.hpp"
#include "ym_abi_scans.hpp"
#include "ym_abi_match.hpp"
#include "ym_abi_maps.hpp"
#include "ym_abi_debug.hpp"
}  // extern "C"
