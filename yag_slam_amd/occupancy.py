"""Occupancy-grid rendering: drop-in for `karto_scanmatcher.create_occupancy_grid(scans, resolution, range_threshold)`
as yag-slam calls it (/root/reference/yag_slam/graph_slam.py:341-342, /root/reference/ros1/slam_node_ros1:187-202,
/root/reference/yag_slam/helpers.py:595-603).  The returned object has what those callers touch: `.image` (uint8,
[height][width], 0 occupied / 200 unknown / 255 free), `.width`, `.height`, `.offset` (`.x`, `.y`: world position of
cell (0, 0)).  Rendered on the device from the scans' resident twins (include/yagmatch.h, ym_occupancy_*).  The wheel's
algorithm is not in the reference tree: restated from open_karto's OccupancyGrid, parity unpinned."""
import ctypes as C

import numpy as np

from . import _capi
from .transform import Pose2


class OccupancyGrid(object):
    def __init__(self, image, offset, resolution):
        self.image = image
        self.height, self.width = image.shape
        self.offset = offset
        self.resolution = resolution


def create_occupancy_grid(scans, resolution, range_threshold, device=0):
    """scans: yag_slam_amd.models.LocalizedRangeScan (or their native handles, as the reference passes `v.obj._scan`)"""
    L = _capi.lib()
    handles = [s.native(device) if hasattr(s, "native") else s for s in scans]
    arr = (C.c_void_p * max(1, len(handles)))(*handles)
    h = L.ym_occupancy_create(arr, len(handles), float(resolution), float(range_threshold))
    if not h:
        raise _capi.YmError(-1, _capi.last_error())
    try:
        info = _capi.YmOccupancyInfo()
        _capi.check(L.ym_occupancy_get_info(h, C.byref(info)))
        image = np.empty((info.height, info.width), dtype=np.uint8)
        _capi.check(L.ym_occupancy_read(h, image.ctypes.data_as(C.POINTER(C.c_uint8)), image.size))
    finally:
        L.ym_occupancy_destroy(h)
    return OccupancyGrid(image, Pose2(info.offset_x, info.offset_y, 0.0), info.resolution)
