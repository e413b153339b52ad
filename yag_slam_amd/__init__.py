"""yag_slam_amd -- MI355X-native correlative scan matcher, drop-in for yag-slam's match_scan path.

Importing the package never touches the GPU; the HIP library (csrc/ -> libyagmatch.so) is loaded
on first use and raises if it is missing or no device is present -- there is no CPU fallback.
"""
from .config import default_config, default_config_loop, make_config, ScanMatcherConfig  # noqa: F401
from .transform import Transform, Pose2  # noqa: F401

__all__ = ["default_config", "default_config_loop", "make_config", "ScanMatcherConfig", "Transform",
           "Pose2"]
