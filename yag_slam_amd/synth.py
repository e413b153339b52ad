"""Synthetic lidar scenes for benchmarks and parity tests.

Own code, no counterpart in the reference (its `load_intel_dataset` is an empty stub,
/root/reference/yag_slam/helpers.py:607-610).  An 8 m x 6 m rectangular room with K seeded
axis-aligned box obstacles, exact ray casting, a 1081-beam / 270 degree sensor and seeded
Gaussian range noise.  Everything here is numpy on the host: it produces INPUTS only.
"""
import math

import numpy as np

# sensor of the BASELINE configs: 1081 beams over 270 deg
N_BEAMS = 1081
MIN_ANGLE = -2.35619449
ANGLE_INCREMENT = 0.00436332313
MAX_ANGLE = MIN_ANGLE + (N_BEAMS - 1) * ANGLE_INCREMENT
MIN_RANGE = 0.05
MAX_RANGE = 30.0
RANGE_THRESHOLD = 20.0
SCENE_SEED = 1234
NOISE_SEED_BASE = 1000
SIGMA_RANGE = 0.01


class Scene:
    """Line-segment world: room walls + boxes.  segs[:, 0:2] = p0, segs[:, 2:4] = p1."""

    def __init__(self, width=8.0, height=6.0, n_boxes=6, seed=SCENE_SEED):
        self.width, self.height = float(width), float(height)
        rng = np.random.default_rng(seed)
        segs = [
            (0, 0, width, 0), (width, 0, width, height), (width, height, 0, height), (0, height, 0, 0),
        ]
        # boxes live in a 1.2 m band along the walls so that the interior
        # [1.2, W-1.2] x [1.2, H-1.2] stays free for sensor trajectories
        band = 1.2
        self.boxes = []
        tries = 0
        while len(self.boxes) < n_boxes and tries < 10000:
            tries += 1
            w, h = rng.uniform(0.3, 0.9, size=2)
            cx = rng.uniform(0.1 + w / 2, width - 0.1 - w / 2)
            cy = rng.uniform(0.1 + h / 2, height - 0.1 - h / 2)
            x0, x1, y0, y1 = cx - w / 2, cx + w / 2, cy - h / 2, cy + h / 2
            inside_free = (x1 > band and x0 < width - band and y1 > band and y0 < height - band)
            if inside_free:
                continue
            if any(not (x1 < b[0] or x0 > b[1] or y1 < b[2] or y0 > b[3]) for b in self.boxes):
                continue
            self.boxes.append((x0, x1, y0, y1))
            segs += [(x0, y0, x1, y0), (x1, y0, x1, y1), (x1, y1, x0, y1), (x0, y1, x0, y0)]
        self.segs = np.array(segs, dtype=np.float64)

    def cast(self, x, y, theta, n_beams=N_BEAMS, min_angle=MIN_ANGLE, inc=ANGLE_INCREMENT,
             max_range=MAX_RANGE):
        """Exact ranges (no noise) for a sensor at (x, y, theta)."""
        ang = theta + min_angle + np.arange(n_beams) * inc
        dx, dy = np.cos(ang)[:, None], np.sin(ang)[:, None]
        p0x, p0y = self.segs[None, :, 0], self.segs[None, :, 1]
        ex, ey = self.segs[None, :, 2] - p0x, self.segs[None, :, 3] - p0y
        # solve  o + t d = p0 + u e
        den = dx * ey - dy * ex
        with np.errstate(divide="ignore", invalid="ignore"):
            t = ((p0x - x) * ey - (p0y - y) * ex) / den
            u = ((p0x - x) * dy - (p0y - y) * dx) / den
        ok = (np.abs(den) > 1e-12) & (t > 1e-9) & (u >= 0.0) & (u <= 1.0)
        t = np.where(ok, t, np.inf)
        r = t.min(axis=1)
        return np.where(np.isfinite(r), r, max_range + 1.0)

    def scan_ranges(self, pose, index=0, sigma=SIGMA_RANGE, dirty=False, **kw):
        """Noisy ranges for scan number `index` (noise seed 1000 + index)."""
        r = self.cast(pose[0], pose[1], pose[2], **kw)
        rng = np.random.default_rng(NOISE_SEED_BASE + index)
        r = r + rng.normal(0.0, sigma, size=r.shape)
        if dirty:  # 1 % NaN + 1 % over-range
            n = r.shape[0]
            k = max(1, n // 100)
            idx = rng.permutation(n)
            r[idx[:k]] = np.nan
            r[idx[k:2 * k]] = kw.get("max_range", MAX_RANGE) + 1.0
        return r


def single_match_poses():
    """cfg1/cfg2: 10 base scans at x = 2.0 + 0.1 i, y = 3, theta = 0; query truth and prior."""
    base = [(2.0 + 0.1 * i, 3.0, 0.0) for i in range(10)]
    return base, (3.07, 3.04, 0.05), (3.00, 3.00, 0.0)


def loop_trajectory(n, seed=7, cx=4.0, cy=3.0, rx=2.3, ry=1.3):
    """cfg3: n poses along a smooth closed loop inside the free interior, heading = tangent.
    Step length is ~0.1 m for n = 2000 over ~18 laps of arc; returns (truth, odom_prior)."""
    rng = np.random.default_rng(seed)
    per = 2 * math.pi * math.sqrt((rx * rx + ry * ry) / 2.0)
    s = np.arange(n) * 0.1
    ph = 2 * math.pi * s / per
    x = cx + rx * np.cos(ph)
    y = cy + ry * np.sin(ph)
    th = np.arctan2(ry * np.cos(ph), -rx * np.sin(ph))
    truth = np.stack([x, y, th], axis=1)
    prior = truth + np.concatenate(
        [rng.normal(0, 0.03, size=(n, 2)), rng.normal(0, 0.01, size=(n, 1))], axis=1)
    return truth, prior


def chain_poses(n_chains, chain_len=10, seed=99, scene=None):
    """cfg4: n_chains candidate chains of chain_len poses; chain 0 is the true neighbourhood of
    the cfg2 query, the others start at seeded poses in the free interior."""
    scene = scene or Scene()
    rng = np.random.default_rng(seed)
    chains = []
    base, _, _ = single_match_poses()
    chains.append(base[:chain_len])
    lo_x, hi_x = 1.4, scene.width - 1.4 - 0.1 * chain_len
    for _ in range(1, n_chains):
        x0 = rng.uniform(lo_x, hi_x)
        y0 = rng.uniform(1.4, scene.height - 1.4)
        chains.append([(x0 + 0.1 * i, y0, 0.0) for i in range(chain_len)])
    return chains


# ---- the BASELINE workloads as resident scans (bench.py and the -m gpu tests share these builders) ----------------
def resident_scan(ranges, pose, range_threshold=RANGE_THRESHOLD):
    """LocalizedRangeScan of the synthetic sensor at `pose` (x, y, heading)"""
    from .models import LocalizedRangeScan
    return LocalizedRangeScan(ranges, MIN_ANGLE, MAX_ANGLE, ANGLE_INCREMENT, MIN_RANGE, MAX_RANGE, range_threshold,
                              float(pose[0]), float(pose[1]), float(pose[2]))


def single_match_scans(scene=None):
    """cfg1/cfg2: (query at the odometry prior, 10 base scans), noise seeds 1000 + scan index"""
    scene = scene or Scene()
    base_poses, q_truth, q_prior = single_match_poses()
    base = [resident_scan(scene.scan_ranges(p, index=i), p) for i, p in enumerate(base_poses)]
    return resident_scan(scene.scan_ranges(q_truth, index=10), q_prior), base


def trajectory_jobs(n):
    truth, _ = loop_trajectory(n)
    return [(truth[i], i) for i in range(n)]


def trajectory_scans(n, scene=None, ranges=None):
    """cfg3: n scans along the seeded loop; every scan starts at the first pose and carries its odometry pose
    (= truth + noise), the way yag-slam's node hands scans to GraphSlam.process_scan.  `ranges`: pre-generated
    scan_ranges_many(trajectory_jobs(n)).  Returns (truth, scans)."""
    from .transform import Transform
    scene = scene or Scene()
    truth, prior = loop_trajectory(n)
    if ranges is None:
        ranges = scan_ranges_many(trajectory_jobs(n), scene)
    scans = []
    for i in range(n):
        s = resident_scan(ranges[i], truth[0])
        s.odom_pose = Transform(prior[i][0], prior[i][1], 0.0, prior[i][2])
        scans.append(s)
    scans[0].odom_pose = Transform(truth[0][0], truth[0][1], 0.0, truth[0][2])
    return truth, scans


def loop_batch_scans(n_chains, lo=0, hi=None, chain_len=10, scene=None, ranges=None):
    """cfg4: the cfg2 query against `n_chains` candidate chains of `chain_len` scans at seeded poses; chain 0 is the
    query's true neighbourhood.  Only chains [lo, hi) are built (a rank's shard); scan (c, i) has noise seed
    1000 + 100000 + c * chain_len + i whatever the shard.  `ranges`: what scan_ranges_many(loop_batch_jobs(...)) gave
    for the same arguments, when they were generated ahead of time.  Returns (query, chains[lo:hi])."""
    scene = scene or Scene()
    hi = n_chains if hi is None else hi
    jobs = loop_batch_jobs(n_chains, lo, hi, chain_len, scene)
    if ranges is None:
        ranges = scan_ranges_many(jobs, scene)
    _, q_truth, q_prior = single_match_poses()
    query = resident_scan(scene.scan_ranges(q_truth, index=10), q_prior)
    scans = [resident_scan(r, j[0]) for r, j in zip(ranges, jobs)]
    return query, [scans[k * chain_len:(k + 1) * chain_len] for k in range(hi - lo)]


# ---- bulk generation: ray casting is ~1 ms of numpy per scan, the loop-closure workload has 40 960 of them ---------
_JOB_SCENE = None


def _ranges_job(job):
    pose, index = job
    return _JOB_SCENE.scan_ranges(pose, index=index)


def scan_ranges_many(jobs, scene=None, workers=1):
    """[(pose, noise index), ...] -> list of range arrays, optionally on a fork pool.  Call it with workers > 1 only
    BEFORE the process touches the GPU (a forked child must not inherit an initialised HIP runtime)."""
    global _JOB_SCENE
    _JOB_SCENE = scene or Scene()
    jobs = list(jobs)
    if workers <= 1 or len(jobs) < 64:
        return [_ranges_job(j) for j in jobs]
    import multiprocessing as mp
    with mp.get_context("fork").Pool(workers) as pool:
        return pool.map(_ranges_job, jobs, chunksize=max(1, len(jobs) // (8 * workers)))


def loop_batch_jobs(n_chains, lo=0, hi=None, chain_len=10, scene=None):
    """the (pose, noise index) list behind loop_batch_scans(...)'s chains, chain-major"""
    hi = n_chains if hi is None else hi
    poses = chain_poses(n_chains, chain_len=chain_len, scene=scene)
    return [(poses[c][i], 100000 + c * chain_len + i) for c in range(lo, hi) for i in range(chain_len)]
