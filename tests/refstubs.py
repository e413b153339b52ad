"""Import the REFERENCE (/root/reference, build container only) with the third-party modules this image lacks replaced
by in-memory stand-ins: numba (njit = identity), cv2 (empty), sba_cpp.SPA2d (records nodes and
constraints, `compute` leaves the poses alone), karto_scanmatcher (attribute bags; ScanMatcherConfig lists the 11 config
keys so that serde's dir() finds them) and -- the point of it -- tiny_tf.tf.Transform = yag_slam_amd.transform.Transform,
the pose class a yag-slam user of this package would hold.  Nothing is written next to the reference's code."""
import os
import sys
import types

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "yag_slam"))


def install():
    if "yag_slam" in sys.modules:
        return
    sys.dont_write_bytecode = True

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda fn: fn

    mod("numba", njit=njit, prange=range)
    mod("cv2")

    class ScanMatcherConfig:
        angle_variance_penalty = 0.0
        distance_variance_penalty = 0.0
        coarse_search_angle_offset = 0.0
        coarse_angle_resolution = 0.0
        fine_search_angle_resolution = 0.0
        use_response_expansion = False
        range_threshold = 0.0
        minimum_angle_penalty = 0.0
        search_size = 0.0
        resolution = 0.0
        smear_deviation = 0.0

    class Wrapper:
        def __init__(self, config):
            self.config = config

    class Bag:
        def __init__(self, *a):
            self.args = a

    class Pose2:
        def __init__(self, x=0.0, y=0.0, yaw=0.0):
            self.x, self.y, self.yaw = x, y, yaw

    def create_occupancy_grid(*a, **k):
        raise NotImplementedError

    mod("karto_scanmatcher", ScanMatcherConfig=ScanMatcherConfig, Wrapper=Wrapper, LaserScanConfig=type("LaserScanConfig", (Bag,), {}),
        LocalizedRangeScan=type("LocalizedRangeScan", (Bag,), {}), Pose2=Pose2, create_occupancy_grid=create_occupancy_grid)

    class Node:
        def __init__(self, x, y, yaw):
            self.x, self.y, self.yaw = x, y, yaw

    class SPA2d:
        def __init__(self):
            self.nodes, self.constraints, self.computes = [], [], 0

        def add_node(self, x, y, yaw, num):
            self.nodes.append(Node(x, y, yaw))

        def add_constraint(self, a, b, x, y, yaw, info):
            self.constraints.append((a, b, x, y, yaw, info))

        def compute(self, *a):
            self.computes += 1

    mod("sba_cpp", SPA2d=SPA2d)
    from yag_slam_amd.transform import Transform
    tt = mod("tiny_tf")
    tt.tf = mod("tiny_tf.tf", Transform=Transform)
    sys.path.insert(0, REF)
