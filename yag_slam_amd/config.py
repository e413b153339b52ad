"""Matcher configuration surface (drop-in for the reference's config dicts).

The reference configures a matcher with a plain dict of 11 keys, copied onto an attribute bag
(`default_config`, `default_config_loop`, `make_config`: /root/reference/yag_slam/helpers.py:339-376),
and serialises that bag back into a dict (/root/reference/yag_slam/serde.py:88-92).  This module
keeps those names, defaults and the smear-deviation assertion (helpers.py:370) so configs
round-trip unchanged.  `minimum_distance_penalty` is Karto's MinimumDistancePenalty (upstream
default 0.5); yag-slam never sets it but the Karto matcher uses it, so it is carried explicitly.
"""

CONFIG_KEYS = (
    "angle_variance_penalty",
    "distance_variance_penalty",
    "coarse_search_angle_offset",
    "coarse_angle_resolution",
    "fine_search_angle_resolution",
    "use_response_expansion",
    "range_threshold",
    "minimum_angle_penalty",
    "search_size",
    "resolution",
    "smear_deviation",
)

default_config = dict(zip(CONFIG_KEYS, (0.3, 0.5, 0.349, 0.0349, 0.00349, True, 20, 0.9, 0.5, 0.01, 0.05)))

# loop-closure matcher: coarser, wider (helpers.py:353-361)
default_config_loop = dict(default_config, resolution=0.05, search_size=4.0)

EXTRA_DEFAULTS = {"minimum_distance_penalty": 0.5}


class ScanMatcherConfig(object):
    """Attribute bag: one attribute per config key, like karto_scanmatcher.ScanMatcherConfig."""

    def __init__(self, **kw):
        for k, v in EXTRA_DEFAULTS.items():
            setattr(self, k, v)
        for k, v in kw.items():
            setattr(self, k, v)

    def as_dict(self):
        return {k: getattr(self, k) for k in CONFIG_KEYS + tuple(EXTRA_DEFAULTS) if hasattr(self, k)}

    def __repr__(self):
        return "ScanMatcherConfig(%s)" % ", ".join("%s=%r" % kv for kv in self.as_dict().items())


def check_smear(resolution, smear_deviation):
    lo, hi = 0.5 * resolution, 10 * resolution
    assert lo <= smear_deviation <= hi, f"Smear deviation must be between {lo} and {hi}"


def make_config(d=None, loop=False):
    """dict (or None) -> ScanMatcherConfig, defaults filled in, smear bounds asserted."""
    params = dict(default_config_loop if loop else default_config)
    if d:
        params.update(d)
    check_smear(params["resolution"], params["smear_deviation"])
    return ScanMatcherConfig(**params)


def print_config(config):
    for k, v in config.as_dict().items():
        print("{}: {}".format(k, v))
