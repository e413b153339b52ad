"""ctypes binding of the CPU oracle (oracle/ym_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py -- never by the product package (yag_slam_amd/).  See oracle/ym_oracle.h for which
reference lines each routine restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SEM_KARTO, SEM_YAGPY, SEM_MIXED = 0, 1, 2
_SEM = {"karto": SEM_KARTO, "yagpy": SEM_YAGPY}

# the switches of ym_oracle.h (bit n set = delta n follows the reference's Python path)
DELTAS = ("D1_CELL_VALUE", "D2_KERNEL_HALF", "D3_GRID_SIZE", "D4_ORIGIN", "D5_ROUNDING", "D6_VALID_FILTER", "D7_RANGE_GATE",
          "D8_RESTAMP", "D9_COARSE_LATTICE", "D10_FINE_LATTICE", "D11_NORMALISER", "D12_PENALTY", "D13_TIES", "D14_POS_COV",
          "D15_ANG_COV", "D16_EXPANSION_CLAMP", "D17_DEFAULT_FINE", "D18_LOOKUP")
ALL_PY = (1 << len(DELTAS)) - 1


def delta_bit(name):
    return 1 << DELTAS.index(name)

# default_config / default_config_loop of the reference (/root/reference/yag_slam/helpers.py:339-361)
# restated as data so the oracle does not depend on the product package.
DEFAULT_CONFIG = {
    "angle_variance_penalty": 0.3,
    "distance_variance_penalty": 0.5,
    "coarse_search_angle_offset": 0.349,
    "coarse_angle_resolution": 0.0349,
    "fine_search_angle_resolution": 0.00349,
    "use_response_expansion": True,
    "range_threshold": 20,
    "minimum_angle_penalty": 0.9,
    "search_size": 0.5,
    "resolution": 0.01,
    "smear_deviation": 0.05,
}
DEFAULT_CONFIG_LOOP = dict(DEFAULT_CONFIG, resolution=0.05, search_size=4.0)


class OrcConfig(C.Structure):
    _fields_ = [
        ("angle_variance_penalty", C.c_double),
        ("distance_variance_penalty", C.c_double),
        ("coarse_search_angle_offset", C.c_double),
        ("coarse_angle_resolution", C.c_double),
        ("fine_search_angle_resolution", C.c_double),
        ("use_response_expansion", C.c_int),
        ("range_threshold", C.c_double),
        ("minimum_angle_penalty", C.c_double),
        ("minimum_distance_penalty", C.c_double),
        ("search_size", C.c_double),
        ("resolution", C.c_double),
        ("smear_deviation", C.c_double),
        ("semantics", C.c_int),
        ("threads", C.c_int),
        ("delta_mask", C.c_uint),
    ]


class OrcScan(C.Structure):
    _fields_ = [
        ("ranges", C.POINTER(C.c_double)),
        ("n", C.c_int),
        ("min_angle", C.c_double),
        ("angle_increment", C.c_double),
        ("min_range", C.c_double),
        ("range_threshold", C.c_double),
        ("pose", C.c_double * 3),
    ]


class OrcResult(C.Structure):
    _fields_ = [
        ("response", C.c_double),
        ("pose", C.c_double * 3),
        ("cov", C.c_double * 9),
        ("coarse_dims", C.c_int * 3),
        ("fine_dims", C.c_int * 3),
        ("n_query_points", C.c_int),
        ("expansions", C.c_int),
        ("hypotheses", C.c_longlong),
    ]


_lib = None


def build(native=False):
    """Compile the oracle with gcc (oracle/Makefile). native=True -> oracle/_native/ with -march=native."""
    args = ["make", "-s", "-C", _HERE]
    if native:
        args += ["ARCH=native", "OUT=_native/libym_oracle.so"]
    subprocess.check_call(args)
    return os.path.join(_HERE, "_native" if native else "", "libym_oracle.so")


def load(path=None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    # (YM_ORACLE_LIB: another build of the same file, e.g. the sanitizer build `make -C oracle asan`)
    p = path or os.environ.get("YM_ORACLE_LIB") or os.path.join(_HERE, "libym_oracle.so")
    if not os.path.exists(p):
        build()
    lib = C.CDLL(p)
    lib.orc_create.restype = C.c_void_p
    lib.orc_create.argtypes = [C.POINTER(OrcConfig)]
    lib.orc_destroy.argtypes = [C.c_void_p]
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_match.argtypes = [C.c_void_p, C.POINTER(OrcScan), C.POINTER(OrcScan), C.c_int, C.c_int,
                              C.c_int, C.POINTER(OrcResult)]
    ip = C.POINTER(C.c_int)
    lib.orc_grid_u8.restype = C.POINTER(C.c_uint8)
    lib.orc_grid_u8.argtypes = [C.c_void_p] + [ip] * 7
    lib.orc_grid_f64.restype = C.POINTER(C.c_double)
    lib.orc_grid_f64.argtypes = [C.c_void_p, ip]
    lib.orc_grid_offset.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.orc_sums.restype = C.POINTER(C.c_uint32)
    lib.orc_sums.argtypes = [C.c_void_p, C.c_int, ip, ip, ip]
    lib.orc_responses.restype = C.POINTER(C.c_double)
    lib.orc_responses.argtypes = [C.c_void_p, C.c_int, ip, ip, ip]
    lib.orc_raster_points.restype = C.POINTER(C.c_double)
    lib.orc_raster_points.argtypes = [C.c_void_p, ip]
    lib.orc_last_serial_seconds.restype = C.c_double
    lib.orc_last_serial_seconds.argtypes = [C.c_void_p]
    lib.orc_occupancy_grid.argtypes = [C.POINTER(OrcScan), C.POINTER(C.c_double), C.c_int, C.c_double, C.c_double, ip, ip,
                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.c_longlong]
    lib.orc_query_local.restype = C.POINTER(C.c_double)
    lib.orc_query_local.argtypes = [C.c_void_p, ip]
    dp = C.POINTER(C.c_double)
    lib.orc_point_readings.argtypes = [C.POINTER(OrcScan), C.c_int, dp, dp]
    lib.orc_valid_points.argtypes = [dp, dp, C.c_int, C.c_double, C.c_double, C.c_int,
                                     C.POINTER(C.c_uint8)]
    lib.orc_kernel_karto.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_uint8)]
    lib.orc_kernel_yagpy.argtypes = [C.c_double, C.c_double, dp]
    lib.orc_arange.argtypes = [C.c_double, C.c_double, C.c_double, dp, C.c_int]
    if path is None:
        _lib = lib
    return lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def make_scan(ranges, min_angle, angle_increment, min_range, range_threshold, pose):
    """-> (OrcScan, keepalive ndarray).  `pose` = (x, y, heading)."""
    r = np.ascontiguousarray(ranges, dtype=np.float64)
    s = OrcScan()
    s.ranges = _dptr(r)
    s.n = int(r.shape[0])
    s.min_angle = float(min_angle)
    s.angle_increment = float(angle_increment)
    s.min_range = float(min_range)
    s.range_threshold = float(range_threshold)
    s.pose[0], s.pose[1], s.pose[2] = (float(v) for v in pose)
    return s, r


def scan_from(obj):
    """OrcScan from any duck-typed LocalizedRangeScan (.ranges .min_angle .angle_increment
    .min_range .range_threshold .corrected_pose.{x,y,euler[-1]})."""
    p = obj.corrected_pose
    return make_scan(obj.ranges, obj.min_angle, obj.angle_increment, obj.min_range,
                     obj.range_threshold, (p.x, p.y, p.euler[-1]))


def make_config(d=None, semantics="karto", loop=False, threads=1, minimum_distance_penalty=0.5, delta_mask=None):
    cfg = dict(DEFAULT_CONFIG_LOOP if loop else DEFAULT_CONFIG)
    if d:
        cfg.update(d)
    c = OrcConfig()
    for k, v in cfg.items():
        if hasattr(c, k):
            setattr(c, k, int(v) if k == "use_response_expansion" else float(v))
    c.minimum_distance_penalty = float(cfg.get("minimum_distance_penalty", minimum_distance_penalty))
    c.semantics = _SEM[semantics] if isinstance(semantics, str) else int(semantics)
    if delta_mask is not None:  # per-switch choice between the Karto and the Python behaviour
        c.semantics = SEM_MIXED
        c.delta_mask = int(delta_mask)
    c.threads = int(threads)
    return c


class Oracle:
    """One matcher instance == Scan2DMatcherCpp / Scan2DMatcherPy of the reference."""

    def __init__(self, config_dict=None, semantics="karto", loop=False, threads=1, lib=None, delta_mask=None):
        self.lib = lib or load()
        self.semantics = semantics
        self.cfg = make_config(config_dict, semantics, loop, threads, delta_mask=delta_mask)
        self.ctx = self.lib.orc_create(C.byref(self.cfg))
        if not self.ctx:
            raise ValueError(self.lib.orc_last_error().decode())

    def __del__(self):
        if getattr(self, "ctx", None):
            self.lib.orc_destroy(self.ctx)
            self.ctx = None

    def match_raw(self, query, base, penalty=True, do_fine=True):
        """query: OrcScan; base: list[OrcScan] -> OrcResult (raises on error)."""
        arr = (OrcScan * max(1, len(base)))(*base)
        res = OrcResult()
        rc = self.lib.orc_match(self.ctx, C.byref(query), arr, len(base), int(bool(penalty)),
                                int(bool(do_fine)), C.byref(res))
        if rc != 0:
            raise RuntimeError(self.lib.orc_last_error().decode())
        return res

    def last_serial_seconds(self):
        return float(self.lib.orc_last_serial_seconds(self.ctx))

    def match_scan(self, query, base_scans, penalty=True, do_fine=True):
        """Duck-typed scans in, dict out."""
        keep = []
        q, k = scan_from(query)
        keep.append(k)
        bs = []
        for b in base_scans:
            s, k = scan_from(b)
            bs.append(s)
            keep.append(k)
        r = self.match_raw(q, bs, penalty, do_fine)
        return {
            "response": r.response,
            "pose": np.array(r.pose[:]),
            "cov": np.array(r.cov[:]).reshape(3, 3),
            "coarse_dims": tuple(r.coarse_dims[:]),
            "fine_dims": tuple(r.fine_dims[:]),
            "n_query_points": r.n_query_points,
            "expansions": r.expansions,
            "hypotheses": r.hypotheses,
        }

    # ---- intermediates of the last match -------------------------------------------------
    def grid_u8(self):
        v = [C.c_int() for _ in range(7)]
        p = self.lib.orc_grid_u8(self.ctx, *[C.byref(x) for x in v])
        w, h, pitch, rx, ry, rw, rh = (x.value for x in v)
        a = np.ctypeslib.as_array(p, shape=(h, pitch)).copy()
        return a[:, :w], dict(width=w, height=h, pitch=pitch, roi=(rx, ry, rw, rh))

    def grid_f64(self):
        n = C.c_int()
        p = self.lib.orc_grid_f64(self.ctx, C.byref(n))
        if not p or n.value == 0:
            return None
        return np.ctypeslib.as_array(p, shape=(n.value, n.value)).copy()

    def grid_offset(self):
        ox, oy = C.c_double(), C.c_double()
        self.lib.orc_grid_offset(self.ctx, C.byref(ox), C.byref(oy))
        return ox.value, oy.value

    def _volume(self, fn, pass_, dtype):
        nx, ny, nt = C.c_int(), C.c_int(), C.c_int()
        p = fn(self.ctx, pass_, C.byref(nx), C.byref(ny), C.byref(nt))
        if not p or nx.value * ny.value * nt.value == 0:
            return None
        a = np.ctypeslib.as_array(p, shape=(ny.value, nx.value, nt.value)).copy()
        return a.transpose(2, 0, 1)  # -> [it][iy][ix]

    def sums(self, pass_=0):
        """integer correlation sums, canonical layout [itheta][iy][ix]"""
        return self._volume(self.lib.orc_sums, pass_, np.uint32)

    def responses(self, pass_=0):
        return self._volume(self.lib.orc_responses, pass_, np.float64)

    def raster_points(self):
        n = C.c_int()
        p = self.lib.orc_raster_points(self.ctx, C.byref(n))
        if n.value == 0:
            return np.zeros((0, 2))
        return np.ctypeslib.as_array(p, shape=(n.value, 2)).copy()

    def query_local(self):
        n = C.c_int()
        p = self.lib.orc_query_local(self.ctx, C.byref(n))
        if n.value == 0:
            return np.zeros((0, 2))
        return np.ctypeslib.as_array(p, shape=(n.value, 2)).copy()


# ---- stand-alone pieces -------------------------------------------------------------------
def point_readings(ranges, min_angle, angle_increment, min_range, range_threshold, pose,
                   semantics="karto"):
    lib = load()
    s, keep = make_scan(ranges, min_angle, angle_increment, min_range, range_threshold, pose)
    xs = np.zeros(max(1, s.n))
    ys = np.zeros(max(1, s.n))
    n = lib.orc_point_readings(C.byref(s), _SEM[semantics], _dptr(xs), _dptr(ys))
    return xs[:n].copy(), ys[:n].copy()


def valid_points(xs, ys, vpx, vpy, semantics="karto"):
    lib = load()
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    ys = np.ascontiguousarray(ys, dtype=np.float64)
    keep = np.zeros(max(1, len(xs)), dtype=np.uint8)
    lib.orc_valid_points(_dptr(xs), _dptr(ys), len(xs), float(vpx), float(vpy), _SEM[semantics],
                         keep.ctypes.data_as(C.POINTER(C.c_uint8)))
    return keep[:len(xs)].astype(bool)


def kernel_karto(resolution, smear):
    lib = load()
    n = lib.orc_kernel_karto(resolution, smear, None)
    k = np.zeros((n, n), dtype=np.uint8)
    lib.orc_kernel_karto(resolution, smear, k.ctypes.data_as(C.POINTER(C.c_uint8)))
    return k


def kernel_yagpy(resolution, smear):
    lib = load()
    n = lib.orc_kernel_yagpy(resolution, smear, None)
    k = np.zeros((n, n))
    lib.orc_kernel_yagpy(resolution, smear, _dptr(k))
    return k


def arange(start, stop, step):
    lib = load()
    n = lib.orc_arange(start, stop, step, None, 0)
    out = np.zeros(max(1, n))
    lib.orc_arange(start, stop, step, _dptr(out), n)
    return out[:n]


def occupancy_grid(scans, resolution, range_threshold):
    """duck-typed scans (with .max_range) -> (image uint8 [h][w], (offset_x, offset_y)); see orc_occupancy_grid"""
    lib = load()
    keep, arr = [], []
    for sc in scans:
        s, k = scan_from(sc)
        arr.append(s)
        keep.append(k)
    a = (OrcScan * len(arr))(*arr)
    mr = (C.c_double * len(arr))(*[float(getattr(sc, "max_range", 1e300)) for sc in scans])
    w, h, ox, oy = C.c_int(), C.c_int(), C.c_double(), C.c_double()
    if lib.orc_occupancy_grid(a, mr, len(arr), float(resolution), float(range_threshold), C.byref(w), C.byref(h), C.byref(ox),
                              C.byref(oy), None, 0) != 0:
        raise RuntimeError(lib.orc_last_error().decode())
    im = np.zeros((h.value, w.value), dtype=np.uint8)
    if lib.orc_occupancy_grid(a, mr, len(arr), float(resolution), float(range_threshold), C.byref(w), C.byref(h), C.byref(ox),
                              C.byref(oy), im.ctypes.data_as(C.POINTER(C.c_uint8)), im.size) != 0:
        raise RuntimeError(lib.orc_last_error().decode())
    return im, (ox.value, oy.value)
