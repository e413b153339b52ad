"""ctypes binding of libyagmatch.so (include/yagmatch.h).  Thin: structs, prototypes, error mapping.

The library is built in-tree (yag_slam_amd/libyagmatch.so) by `build()` = `make -C csrc`.  There is
no fallback: if the library cannot be loaded, or it reports no HIP device, every entry raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YM_LIB_PATH") or os.path.join(_HERE, "libyagmatch.so")  # (the override: A/B timing of two builds)

YM_OK = 0
SEM = {"karto": 0, "yagpy": 1}

EXPORTS = (
    "ym_version", "ym_build_id", "ym_device_count", "ym_last_error", "ym_create", "ym_destroy", "ym_get_config",
    "ym_set_stream", "ym_synchronize", "ym_scan_create", "ym_scans_create", "ym_scans_destroy", "ym_scan_set_pose", "ym_scans_set_poses", "ym_scan_get_pose",
    "ym_scan_size", "ym_scan_structure_trusted", "ym_scan_destroy", "ym_match", "ym_match_scans", "ym_map_sequence", "ym_process_scan", "ym_sequence_stats", "ym_async_slots",
    "ym_match_scans_async", "ym_wait", "ym_match_batch", "ym_batch_create", "ym_batch_destroy", "ym_batch_size",
    "ym_match_pairs", "ym_pairs_create",
    "ym_batch_run_async", "ym_batch_wait", "ym_debug_grid_info",
    "ym_debug_grid", "ym_debug_sums", "ym_debug_query_local", "ym_debug_cells", "ym_debug_option", "ym_debug_stamps",
    "ym_profile_enable",
    "ym_profile_read", "ym_cache_stats", "ym_debug_counters",
    "ym_coarse_dims", "ym_match_slice_begin", "ym_match_slice_finish",
    "ym_occupancy_create", "ym_occupancy_get_info", "ym_occupancy_read", "ym_occupancy_destroy",
    "ym_map_from_occupancy", "ym_map_from_grid", "ym_map_size", "ym_map_read", "ym_map_destroy", "ym_match_map",
)


class YmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libyagmatch error %d: %s" % (code, msg))
        self.code = code


class YmConfig(C.Structure):
    _fields_ = [
        ("angle_variance_penalty", C.c_double),
        ("distance_variance_penalty", C.c_double),
        ("coarse_search_angle_offset", C.c_double),
        ("coarse_angle_resolution", C.c_double),
        ("fine_search_angle_resolution", C.c_double),
        ("range_threshold", C.c_double),
        ("minimum_angle_penalty", C.c_double),
        ("minimum_distance_penalty", C.c_double),
        ("search_size", C.c_double),
        ("resolution", C.c_double),
        ("smear_deviation", C.c_double),
        ("use_response_expansion", C.c_int32),
        ("semantics", C.c_int32),
    ]


class YmScanDesc(C.Structure):
    _fields_ = [
        ("ranges", C.POINTER(C.c_double)),
        ("n", C.c_int32),
        ("reserved", C.c_int32),
        ("min_angle", C.c_double),
        ("max_angle", C.c_double),
        ("angle_increment", C.c_double),
        ("min_range", C.c_double),
        ("max_range", C.c_double),
        ("range_threshold", C.c_double),
        ("pose", C.c_double * 3),
    ]


class YmResult(C.Structure):
    _fields_ = [
        ("response", C.c_double),
        ("pose", C.c_double * 3),
        ("cov", C.c_double * 9),
        ("coarse_response", C.c_double),
        ("hypotheses", C.c_int64),
        ("coarse_dims", C.c_int32 * 3),
        ("fine_dims", C.c_int32 * 3),
        ("n_query_points", C.c_int32),
        ("expansions", C.c_int32),
        ("status", C.c_int32),
        ("reserved", C.c_int32),
    ]


class YmOccupancyInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("offset_x", C.c_double), ("offset_y", C.c_double),
                ("resolution", C.c_double)]


class YmMapSearch(C.Structure):
    _fields_ = [("xy_search", C.c_double), ("xy_step", C.c_double), ("angle_search", C.c_double), ("angle_step", C.c_double),
                ("grid_resolution", C.c_double), ("penalize", C.c_int32), ("reserved", C.c_int32)]


class YmGridInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("width", "height", "pitch", "origin_x", "origin_y", "storage_w",
                                        "storage_h", "roi_x", "roi_y", "roi_w", "roi_h")] + [
        ("pad", C.c_int32), ("offset_x", C.c_double), ("offset_y", C.c_double)]


_lib = None


def build(verbose=False):
    """hipcc --offload-arch=gfx950 the kernels + C ABI into yag_slam_amd/libyagmatch.so."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc")]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def _share_hip_runtime_with_torch():
    """A process can hold ONE HSA runtime (the second one cannot open the GPU), and PyTorch-ROCm wheels bundle their own
    libamdhip64 / libhsa-runtime64.  If torch is imported first, libyagmatch's `libamdhip64.so.7` dependency resolves to
    the copy torch already loaded and both share it; the other way round torch loads a second runtime and then reports
    "No HIP GPUs are available".  So when torch is installed it is imported before the library is loaded (dist.py and
    bench.py use both in one process).  YM_SKIP_TORCH_PRELOAD=1 opts out for torch-free deployments."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("YM_SKIP_TORCH_PRELOAD"):
        return
    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def lib():
    """Load (building first if the .so is absent) and prototype the library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)
    L.ym_version.restype = C.c_int
    L.ym_device_count.restype = C.c_int
    L.ym_last_error.restype = C.c_char_p
    L.ym_build_id.restype = C.c_char_p
    L.ym_create.restype = vp
    L.ym_create.argtypes = [C.POINTER(YmConfig), C.c_int]
    L.ym_destroy.argtypes = [vp]
    L.ym_destroy.restype = None
    L.ym_get_config.argtypes = [vp, C.POINTER(YmConfig)]
    L.ym_set_stream.argtypes = [vp, vp]
    L.ym_synchronize.argtypes = [vp]
    L.ym_scan_create.restype = vp
    L.ym_scan_create.argtypes = [C.c_int, C.POINTER(YmScanDesc)]
    L.ym_scan_set_pose.argtypes = [vp, C.c_double, C.c_double, C.c_double]
    L.ym_scans_set_poses.argtypes = [C.POINTER(vp), dp, C.c_int]
    L.ym_scan_get_pose.argtypes = [vp, dp]
    L.ym_scan_size.argtypes = [vp]
    L.ym_scan_destroy.argtypes = [vp]
    L.ym_scans_create.argtypes = [C.c_int, vp, C.c_int, vp]
    L.ym_scans_destroy.argtypes = [vp, C.c_int]
    L.ym_scans_destroy.restype = None
    L.ym_scan_destroy.restype = None
    L.ym_match.argtypes = [vp, C.POINTER(YmScanDesc), C.POINTER(YmScanDesc), C.c_int, C.c_int, C.c_int,
                           C.POINTER(YmResult)]
    L.ym_match_scans.argtypes = [vp, vp, C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.POINTER(YmResult)]
    L.ym_async_slots.argtypes = [vp]
    L.ym_match_scans_async.argtypes = [vp, vp, C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]
    L.ym_wait.argtypes = [vp, C.c_int, C.POINTER(YmResult)]
    L.ym_match_batch.argtypes = [vp, vp, C.POINTER(vp), ip, C.c_int, C.c_int, C.c_int, C.POINTER(YmResult),
                                 C.POINTER(YmResult), ip]
    L.ym_batch_create.restype = vp
    L.ym_batch_create.argtypes = [vp, vp, C.POINTER(vp), ip, C.c_int]
    L.ym_pairs_create.restype = vp
    L.ym_pairs_create.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), ip, C.c_int]
    L.ym_match_pairs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), ip, C.c_int, C.c_int, C.c_int, C.POINTER(YmResult)]
    L.ym_batch_destroy.argtypes = [vp]
    L.ym_batch_destroy.restype = None
    L.ym_batch_size.argtypes = [vp]
    L.ym_batch_run_async.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int64, vp]
    L.ym_batch_wait.argtypes = [vp, C.c_int, C.POINTER(YmResult), C.POINTER(YmResult), ip]
    L.ym_debug_grid_info.argtypes = [vp, C.c_int, C.POINTER(YmGridInfo)]
    L.ym_debug_grid.argtypes = [vp, C.c_int, C.POINTER(C.c_uint8), C.c_int64]
    L.ym_debug_sums.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_int64]
    L.ym_debug_query_local.argtypes = [vp, C.c_int, dp, C.c_int32, ip]
    L.ym_debug_cells.argtypes = [vp, C.c_int, ip, C.c_int64, ip]
    L.ym_debug_option.argtypes = [vp, C.c_int, C.c_int]
    L.ym_debug_stamps.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64), C.c_int32]
    L.ym_profile_enable.argtypes = [vp, C.c_int]
    L.ym_profile_read.argtypes = [vp, C.c_int, dp, C.POINTER(C.c_int64), C.c_int]
    L.ym_cache_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.ym_debug_counters.argtypes = [vp, C.POINTER(C.c_int64), C.c_int32]
    L.ym_coarse_dims.argtypes = [vp, ip]
    L.ym_scan_structure_trusted.argtypes = [vp, C.c_int]
    L.ym_process_scan.argtypes = [vp, vp, C.POINTER(vp), C.c_int, dp, dp, C.c_int, C.c_int, C.POINTER(YmResult)]
    L.ym_sequence_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.ym_map_sequence.argtypes = [vp, C.POINTER(vp), dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(YmResult), ip]
    L.ym_match_slice_begin.argtypes = [vp, vp, C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    L.ym_match_slice_finish.argtypes = [vp, C.POINTER(YmResult)]
    L.ym_occupancy_create.restype = vp
    L.ym_occupancy_create.argtypes = [C.POINTER(vp), C.c_int, C.c_double, C.c_double]
    L.ym_occupancy_get_info.argtypes = [vp, C.POINTER(YmOccupancyInfo)]
    L.ym_occupancy_read.argtypes = [vp, C.POINTER(C.c_uint8), C.c_int64]
    L.ym_occupancy_destroy.argtypes = [vp]
    L.ym_occupancy_destroy.restype = None
    L.ym_map_from_occupancy.restype = vp
    L.ym_map_from_occupancy.argtypes = [vp, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int]
    L.ym_map_from_grid.restype = vp
    L.ym_map_from_grid.argtypes = [vp, dp, C.c_int, C.c_int]
    L.ym_map_size.argtypes = [vp, ip, ip]
    L.ym_map_read.argtypes = [vp, dp, C.c_int64]
    L.ym_map_destroy.argtypes = [vp]
    L.ym_map_destroy.restype = None
    L.ym_match_map.argtypes = [vp, vp, C.c_double, C.c_double, C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.POINTER(YmMapSearch),
                               C.POINTER(YmResult)]
    _lib = L
    return L


def build_id():
    """the library's build id: a hash of the sources it was built from (include/yagmatch.h, ym_build_id)"""
    return lib().ym_build_id().decode()


def check(rc):
    if rc != YM_OK:
        raise YmError(rc, lib().ym_last_error().decode(errors="replace"))
    return rc


def last_error():
    return lib().ym_last_error().decode(errors="replace")


def config_struct(cfg, semantics="karto"):
    """ScanMatcherConfig / dict -> YmConfig"""
    d = cfg if isinstance(cfg, dict) else cfg.as_dict()
    c = YmConfig()
    for name, _ in YmConfig._fields_:
        if name == "semantics":
            c.semantics = SEM[semantics] if isinstance(semantics, str) else int(semantics)
        elif name == "use_response_expansion":
            c.use_response_expansion = int(bool(d["use_response_expansion"]))
        elif name == "minimum_distance_penalty":
            c.minimum_distance_penalty = float(d.get("minimum_distance_penalty", 0.5))
        else:
            setattr(c, name, float(d[name]))
    return c
