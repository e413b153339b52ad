#!/usr/bin/env python3
"""bench.py -- pose hypotheses/sec of the MI355X correlative scan matcher.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`, as a child process, before anything here
touches the GPU) and relays rank 0's JSON line; under torch.distributed.run it is a rank.

The metric line (BASELINE.json configs[1], "cfg2"): 1081-beam synthetic scans, search 0.5 m / 0.349 rad,
resolution 0.01 m, coarse + fine pass with the odometry penalty, Karto semantics.  One STEP = one pass of the hot
path over a batch of `--batch` independent single-match problems of that exact configuration (one query against
`--batch` distinct candidate 10-scan chains, each with its own correlation grid, coarse + fine search, covariances),
issued as enqueues of `--launch-batch` problems, all inputs resident in HBM.  With N GPUs every rank runs its own
batch per step (weak scaling) and the ranks exchange their best (response, pose) records with one RCCL all-gather
per step.  `value` = lattice points scored by all ranks / wall time of K steps.

`config.by_config` carries the other BASELINE configs measured in the same run: cfg1 (CPU oracle), cfg2 as ONE
unbatched match_scan call, cfg3 (2000-scan sequential mapping), cfg4 (1 query vs 4096 distinct chains, the chains
sharded over the ranks = strong scaling, RCCL arg-max), cfg5 (stress lattice) -- and what production would see of the
metric workload: `cfg2x_batch_sweep` (1 query x N chains per enqueue, N = 8 ... 4096), `cfg2x_fresh_query` (a NEW query
every enqueue: descriptor copy, query projection and the query's pair lists inside the timed loop) and `cfg2x_cold`
(every chain re-posed before every enqueue: the point cache misses on every scan).  A leg that raised is listed under
`leg_errors` and makes the exit code non-zero.

`roofline`: `bound` = the resource of the dominant kernel with the highest measured utilisation, `frac` <= 1 against that
resource's peak.  The kernel's duration is measured in this run (HIP events on the launch stream); the counters behind
the utilisations (VALU / LDS / HBM, `rocprofv3 --pmc`) are replayed from the committed `profiles/*.json` of the same
kernel and say so (`replayed_from`).  `algorithmic_frac` keeps SURVEY.md 8(d)'s figure (one grid byte per beam and
hypothesis against the HBM peak): above 1 means the bytes are served from LDS, not that a roof was broken.
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
STRESS_CONFIG = dict(search_size=2.0, resolution=0.005, coarse_search_angle_offset=0.785)  # BASELINE configs[4]
CFG3_SCANS = 2000
CFG4_CHAINS = 4096


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60, help="timed steps (60 x 17.4 ms: a timed region of a second)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16384, help="independent cfg2 matches per step per GPU")
    ap.add_argument("--launch-batch", type=int, default=4096, help="matches per enqueue (workspace size)")
    ap.add_argument("--lanes", type=int, default=4, help="matchers (stream + workspace each) the enqueues of a step alternate over: with one "
                    "per enqueue (four) the kernels of the four enqueues of a step overlap freely -- 18.1 ms per step against 19.1 "
                    "with two lanes and 19.9 with one (scripts/dev/lanes_ab.sh, same box)")
    ap.add_argument("--only", default="", help="comma list of {cfg2x,single,cfg3,cfg4,cfg5,cpu}: run only these legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-distinct-queries", action="store_true", help="skip the cfg2x_distinct_queries leg (one query per item)")
    ap.add_argument("--only-headline", action="store_true", help="profiling runs: do not time the other form of the cfg2x workload")
    ap.add_argument("--headline", default="distinct_queries", choices=("one_query", "distinct_queries"),
                    help="which cfg2x workload the metric line times: every item against the SAME query object, or every item with its own query")
    ap.add_argument("--no-production-legs", action="store_true",
                    help="skip cfg2x_batch_sweep / _fresh_query / _cold (profiling runs: only launches of the metric's batch size)")
    ap.add_argument("--legs", default="", help="development: comma list of production legs of cfg2x to run (cfg2x_yagpy, cfg2x_fresh_scans, ...); default all")
    ap.add_argument("--cfg4-chains", type=int, default=CFG4_CHAINS)
    ap.add_argument("--cfg3-scans", type=int, default=CFG3_SCANS)
    ap.add_argument("--corr-u", type=int, default=0, help="development: beams in flight per lane in the correlate kernel")
    ap.add_argument("--corr-chunks", type=int, default=0, help="development: beam chunks per angle in the correlate kernel")
    ap.add_argument("--corr-region", type=int, default=0, help="development: 1 = direct correlate on batches too")
    ap.add_argument("--corr-region-na", type=int, default=0, help="development: jobs per wave of the gather correlate")
    ap.add_argument("--corr-pad-lds", type=int, default=0, help="development: extra LDS bytes per correlate block")
    return ap.parse_args(argv)


def self_launch(args):
    """`python bench.py --gpus N` as a plain command: run the N ranks as a CHILD process (this process has not touched
    the GPU, and never will), relay the JSON line, return the child's exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
        return p.returncode  # non-zero when a leg failed: the line then lists it under "leg_errors"
    return p.returncode if p.returncode != 0 else 1


def profile_json(name):
    try:
        return json.load(open(os.path.join(REPO, "profiles", name)))
    except Exception:
        return None


DEVICE = {"cus": 256, "clock_hz": 2.4e9}  # MI355X; main() replaces both with what the device reports


def replayed_roofline(workload, kernel, kernel_s, valu_peak_key="region_correlate_body", cus=None):
    """`roofline` block of `kernel` as it ran in `workload`: its counters per launch from profiles/counters.json (rocprofv3 --pmc
    passes of that workload, scripts/profile_pmc.sh) over the kernel duration `kernel_s` (seconds; measured live where the
    caller can, else the profiled run's) against the measured peaks of profiles/issue_peaks.json.  bound = the resource with
    the highest utilisation.  None when the tree holds no counters for the kernel."""
    ctr, peaks = profile_json("counters.json"), profile_json("issue_peaks.json")
    try:
        k = ctr["workloads"][workload]["kernels"][kernel]
    except (KeyError, TypeError):
        return None
    live = kernel_s is not None
    if kernel_s is None:
        kernel_s = k.get("us", 0.0) * 1e-6
    if not kernel_s or not peaks:
        return None
    clocks = kernel_s * DEVICE["clock_hz"] * (cus if cus is not None else DEVICE["cus"])  # (cus = 1 for a one-block kernel)
    vp = peaks.get(valu_peak_key, {}).get("peak_per_cu_clk") or peaks["generic"]["slow_class_per_cu_clk"]
    res = {}
    if k.get("SQ_INSTS_VALU"):
        res["valu_issue"] = {"achieved": k["SQ_INSTS_VALU"] / clocks, "peak": vp, "unit": "wave-instructions per CU and clock",
                             "peak_source": "profiles/issue_peaks.json: " + valu_peak_key}
    if k.get("SQ_LDS_IDX_ACTIVE"):
        res["lds"] = {"achieved": k["SQ_LDS_IDX_ACTIVE"] / clocks, "peak": 1.0, "unit": "LDS busy cycles per CU and clock"}
    if k.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        res["vector_l1"] = {"achieved": k["TCP_TOTAL_CACHE_ACCESSES_sum"] / clocks, "peak": 1.0, "unit": "cache-line visits per CU and clock"}
    # FETCH_SIZE is in KiB and counts HALF the bytes of the 128-byte lines a kernel's 16-byte-per-lane loads request from the fabric --
    # calibrated on this kernel's own access pattern (192 contiguous bytes per row, rows a window pitch apart, and every other row)
    # and on a plain stream: scripts/exp/fetch_calib.hip, profiles/r05_fetch_calibration.md (factor 2.000 in all three)
    hbm_bytes = (k.get("FETCH_SIZE", 0.0) * 2.0 + k.get("WRITE_SIZE", 0.0)) * 1024.0
    if hbm_bytes:
        res["hbm"] = {"achieved": hbm_bytes / kernel_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "read_bytes": k.get("FETCH_SIZE", 0.0) * 2048.0,
                      "written_bytes": k.get("WRITE_SIZE", 0.0) * 1024.0}
    for r_ in res.values():
        r_["frac"] = r_["achieved"] / r_["peak"]
    if not res:
        return None
    bound = max(res, key=lambda k_: res[k_]["frac"])
    wc = k.get("SQ_WAVE_CYCLES", 0.0)
    # the counters describe the library they were collected on: its build id (a hash of the kernel sources, compiled into the library) must
    # be that of the library that runs now, else the utilisations are another kernel's -- `stale`, and no fraction is claimed
    from yag_slam_amd import _capi
    running = _capi.build_id()
    stale = ctr.get("build_id") != running
    if stale:
        return {"bound": bound, "kernel": kernel, "achieved": None, "peak": res[bound]["peak"], "unit": res[bound]["unit"], "frac": None, "traffic": None,
                "stale": True, "stale_because": "profiles/counters.json was collected on build %s, this run's library is build %s: re-run scripts/profile_pmc.sh" % (
                    ctr.get("build_id"), running), "kernel_us": kernel_s * 1e6, "kernel_us_is": "measured in this run" if live else "the profiled run's",
                "replayed_from": None}
    return {"bound": bound, "stale": False, "build_id": running, "grid_in_the_counter_passes": k.get("grid"), "kernel": kernel, "achieved": res[bound]["achieved"], "peak": res[bound]["peak"], "unit": res[bound]["unit"],
            "frac": res[bound]["frac"], "traffic": hbm_bytes or None, "hbm_frac": res["hbm"]["frac"] if "hbm" in res else None,
            "resources": res, "kernel_us": kernel_s * 1e6, "kernel_us_is": "measured in this run" if live else "the profiled run's",
            "kernel_us_in_the_counter_passes": k.get("us"),  # (under rocprofv3 the kernel runs a few per cent longer; the counts per launch are divided by the LIVE duration)
            "wave_time": {"waiting_at_waitcnt_or_barrier": k.get("SQ_WAIT_ANY", 0.0) / wc, "issue_stalled": k.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                          "issuing": k.get("SQ_ACTIVE_INST_ANY", 0.0) / wc} if wc else None,
            "l2_hit_rate": k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]) if k.get("TCC_HIT_sum") else None,
            "replayed_from": "profiles/counters.json (%s, workload %s: %s)" % (ctr.get("tag"), workload, ctr["workloads"][workload]["command"])}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds=6.0):
    """The CPU oracle (karto semantics, -O3 -march=native on this host) on the same cfg2 problem."""
    from oracle import oracle as orc
    from tests.util import cfg2_scans
    try:
        lib = orc.load(orc.build(native=True))
    except Exception:
        lib = orc.load()
    q, base = cfg2_scans()
    out = {}
    # multi-thread leg: OpenMP over the coarse lattice, capped at 32 threads (more only adds fork/join
    # cost to a 20 ms problem)
    ncores = min(os.cpu_count() or 1, 32)
    for label, threads in (("single", 1), ("all", ncores)):
        o = orc.Oracle(None, "karto", threads=threads, lib=lib)
        qs, keep = orc.scan_from(q)
        bs = [orc.scan_from(b) for b in base]
        arr = [b[0] for b in bs]
        r = o.match_raw(qs, arr, True, True)
        n, t0 = 0, time.perf_counter()
        serial = 0.0
        while time.perf_counter() - t0 < seconds:
            o.match_raw(qs, arr, True, True)
            serial += o.last_serial_seconds()
            n += 1
        dt = time.perf_counter() - t0
        out[label] = dict(hyp_per_s=n * r.hypotheses / dt, matches=n, seconds=dt, threads=threads,
                          serial_fraction=serial / dt, ms_per_match=dt / n * 1e3)
    return out


def physical_cores():
    """(physical cores, logical cpus, one cpu of every core) this process may run on: distinct (package, core) pairs of the cpus in
    its affinity mask"""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    cores = {}
    for c in allowed:
        try:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            key = (open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip())
        except OSError:
            key = ("?", str(c))
        cores.setdefault(key, c)
    return len(cores), len(allowed), sorted(cores.values())


def cpu_throughput(seconds=6.0):
    """The CPU figure that belongs beside a GPU THROUGHPUT: independent cfg2 matches on every physical core of the host at once,
    one oracle context (own correlation grid, one thread) per core, each thread pinned to its core -- what a host-side farm of
    Karto matchers would do with the same workload.  A C driver (oracle/ym_throughput.c, built here with -march=native): no
    Python in the loop."""
    import tempfile
    from tests.util import cfg2_scans
    from yag_slam_amd.config import default_config
    oracle_dir = os.path.join(REPO, "oracle")
    subprocess.check_call(["make", "-s", "-B", "-C", oracle_dir, "throughput", "ARCH=native"])  # (-B: never a binary built on another host)
    q, base = cfg2_scans()
    c = default_config
    head = [c["angle_variance_penalty"], c["distance_variance_penalty"], c["coarse_search_angle_offset"], c["coarse_angle_resolution"],
            c["fine_search_angle_resolution"], float(bool(c["use_response_expansion"])), c["range_threshold"], c["minimum_angle_penalty"],
            0.5, c["search_size"], c["resolution"], c["smear_deviation"], float(len(base)), float(len(q.ranges)),
            q.min_angle, q.angle_increment, q.min_range, q.range_threshold]
    rows = [np.concatenate([[s.corrected_pose.x, s.corrected_pose.y, s.corrected_pose.euler[-1]], s.ranges]) for s in [q] + list(base)]
    phys, logical, cpus = physical_cores()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "cfg2.bin")
        np.concatenate([np.array(head, dtype=np.float64)] + rows).astype(np.float64).tofile(path)
        p = subprocess.run([os.path.join(oracle_dir, "_native", "ym_throughput"), path, str(seconds), ",".join(str(c_) for c_ in cpus)],
                           capture_output=True, text=True, check=True)
    r = json.loads(p.stdout.strip().splitlines()[-1])
    n, dt, hyp = r["matches"], r["seconds"], r["hypotheses_per_match"]
    return dict(hyp_per_s=n * hyp / dt, matches=n, seconds=dt, cores=r["threads"], logical_cpus=logical, matches_per_s=n / dt,
                ms_per_match_per_core=dt / max(1, n) * r["threads"] * 1e3, min_max_matches_per_thread=[r["min_matches_per_thread"], r["max_matches_per_thread"]])


# ------------------------------------------------------------------------------------------------ inputs
def generate_inputs(args, rank, world, legs):
    """Every range array the run needs, generated on the host BEFORE the GPU is touched (fork pool)."""
    from yag_slam_amd import dist as ymdist
    from yag_slam_amd import synth
    scene = synth.Scene()
    workers = max(1, min(32, (os.cpu_count() or 1) // max(1, world)))
    out = {"scene": scene}
    if "cfg2x" in legs or "single" in legs:
        base_poses, q_truth, q_prior = synth.single_match_poses()
        exact = [scene.cast(*p) for p in base_poses]
        n = args.batch if "cfg2x" in legs else 1
        noisy = []
        for c in range(n):  # chain c: the cfg2 poses with its own noise seeds, so every item rasterises a different grid
            rng = np.random.default_rng(100000 * (rank + 1) + c)
            noisy.append([e + rng.normal(0.0, synth.SIGMA_RANGE, size=e.shape) for e in exact])
        out["cfg2"] = (scene.scan_ranges(q_truth, index=10), q_prior, base_poses, noisy)
        if "cfg2x" in legs and not args.no_distinct_queries:
            # one query PER ITEM: its own true pose (seeded, within the search box of its prior), its own readings, its own prior
            rng = np.random.default_rng(424242 + rank)
            dq_truth = np.array(q_truth) + np.concatenate([rng.uniform(-0.05, 0.05, size=(n, 2)), rng.uniform(-0.03, 0.03, size=(n, 1))], axis=1)
            dq_prior = np.array(q_prior) + np.concatenate([rng.uniform(-0.02, 0.02, size=(n, 2)), rng.uniform(-0.01, 0.01, size=(n, 1))], axis=1)
            out["cfg2_queries"] = (synth.scan_ranges_many([(tuple(dq_truth[c]), 200000 + c) for c in range(n)], scene, workers), dq_prior)
    if "cfg2x" in legs and not args.only_headline and not args.no_distinct_queries:
        # cfg2x_trajectory: batch (query, chain) pairs along the cfg3 trajectory -- heterogeneous poses and headings
        out["traj"] = synth.scan_ranges_many(synth.trajectory_jobs(args.batch + 10), scene, workers)
    if "cfg3" in legs and rank == 0:
        out["cfg3"] = synth.scan_ranges_many(synth.trajectory_jobs(args.cfg3_scans), scene, workers)
    if "cfg4" in legs:
        lo, hi = ymdist.shard_range(args.cfg4_chains, rank, world)
        out["cfg4"] = (lo, hi, synth.scan_ranges_many(synth.loop_batch_jobs(args.cfg4_chains, lo, hi, scene=scene), scene, workers))
    return out


# ------------------------------------------------------------------------------------------------ legs
def leg_single(m, query, chain, hyp_per_match):
    """cfg2 as the reference runs it: ONE match_scan call, no batching"""
    n1 = 300
    for _ in range(20):
        m.match_scan(query, chain, True, True)
    t1 = time.perf_counter()
    for _ in range(n1):
        m.match_scan(query, chain, True, True)
    sync_s = (time.perf_counter() - t1) / n1
    t1 = time.perf_counter()
    for i in range(n1):
        if i >= 8:
            m.wait(i % 8)
        m.match_scan_async(query, chain, True, True, slot=i % 8)
    for sl in range(8):
        m.wait(sl)
    pipe_s = (time.perf_counter() - t1) / n1
    return {"lattice": "26x26x21 + 3x3x11", "hypotheses_per_match": hyp_per_match,
            "sync_us_per_match": sync_s * 1e6, "pipelined_us_per_match": pipe_s * 1e6,
            "scan_matches_per_s": 1.0 / sync_s, "hypotheses_per_s": hyp_per_match / sync_s,
            "scan_matches_per_s_pipelined": 1.0 / pipe_s, "hypotheses_per_s_pipelined": hyp_per_match / pipe_s}


def leg_cfg3(m, ranges, n):
    """BASELINE configs[2]: n scans through the call pattern of GraphSlam.process_scan (running chain of 10, grid
    rebuilt at every step), every scan resident.  Three drivers on the same trajectory, each timed with its driver:
    `device_chain`  SequentialMapper.process_scans(device_chain=True): the steps enqueued back to back, each step's pose
                    handed to the next on the device (ym_map_sequence; no host round trip inside a segment of 128);
    `library_loop`  process_scans: one synchronous match per step, the loop inside the library;
    `per_scan_calls` process_scan per scan from Python, what a robot's node does.
    The headline figure of the leg is the first; the other two are bit-identical to each other, the first agrees with
    them to rounding (its odometry priors are composed on the device)."""
    from yag_slam_amd import synth
    from yag_slam_amd.mapping import SequentialMapper
    from yag_slam_amd.models import native_many
    out = {}
    for driver in ("device_chain", "library_loop", "per_scan_calls"):
        truth, scans = synth.trajectory_scans(n, ranges=ranges)
        native_many(scans, m.device)
        mapper = SequentialMapper(m)
        t0 = time.perf_counter()
        if driver == "per_scan_calls":
            results = [mapper.process_scan(s) for s in scans]
        else:
            results = mapper.process_scans(scans, device_chain=(driver == "device_chain"))
        dt = time.perf_counter() - t0
        hyp = sum(r.meta["hypotheses"] for r in results if r is not None)
        err = np.array([[s.corrected_pose.x - t[0], s.corrected_pose.y - t[1]] for s, t in zip(scans, truth)])
        out[driver] = {"scans": n, "seconds": dt, "scan_matches_per_s": (n - 1) / dt, "hypotheses_per_s": hyp / dt,
                       "hypotheses": hyp, "max_position_error_m": float(np.hypot(err[:, 0], err[:, 1]).max()),
                       "final_pose": [scans[-1].corrected_pose.x, scans[-1].corrected_pose.y, scans[-1].corrected_pose.euler[-1]]}
    res = dict(out["device_chain"])
    res["driver"] = "ym_map_sequence, device_chain (SequentialMapper.process_scans)"
    for d in ("library_loop", "per_scan_calls"):
        res[d] = {k: out[d][k] for k in ("seconds", "scan_matches_per_s", "hypotheses_per_s")}
    res["synchronous_drivers_agree_bitwise"] = out["library_loop"]["final_pose"] == out["per_scan_calls"]["final_pose"]
    res["device_chain_final_pose_difference"] = float(np.abs(np.array(out["device_chain"]["final_pose"]) -
                                                             np.array(out["library_loop"]["final_pose"])).max())
    return res


def leg_cfg4(loop_m, gen, args, rank, world, torch, dist):
    """BASELINE configs[3]: one query against 4096 distinct candidate chains (loop config, penalty off, coarse only,
    /root/reference/yag_slam/graph_slam.py:217-220), the chains sharded over the ranks, RCCL arg-max of the best."""
    from yag_slam_amd import dist as ymdist
    from yag_slam_amd import synth
    lo, hi, ranges = gen["cfg4"]
    query, chains = synth.loop_batch_scans(args.cfg4_chains, lo, hi, scene=gen["scene"], ranges=ranges)
    from yag_slam_amd.models import native_many
    native_many([s for ch in chains for s in ch], loop_m.device)
    sh = ymdist.ShardedLoopMatcher.from_local_shard(loop_m, query, chains, lo, args.cfg4_chains, rank, world)
    reps = 16  # (the first enqueue's host work is not hidden behind a predecessor: enough repetitions to amortise it)
    records = torch.zeros((reps, ymdist.RECORD), dtype=torch.float64, device="cuda")
    gathered = torch.zeros((reps, world * ymdist.RECORD), dtype=torch.float64, device="cuda")

    class _Done(object):
        def wait(self):
            return True

    def once(r):
        sh.run_async(records[r], False, False, slot=r)
        if dist is not None:
            if ymdist._staged_through_host(records[r]):  # (the one-GPU dry run over gloo: through host copies, on the matcher's stream)
                with torch.cuda.stream(sh.torch_stream):
                    h = torch.empty(gathered[r].shape, dtype=torch.float64)
                    dist.all_gather_into_tensor(h, records[r].cpu())
                    gathered[r].copy_(h)
                return _Done()
            return dist.all_gather_into_tensor(gathered[r], records[r], async_op=True)
        gathered[r].copy_(records[r])
        return None

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # exact form (per-chain results, response expansion folded in), for the check and the one-shot latency; the first call
    # of a matcher also sizes its workspace and fills the point cache, so the timed one is the second
    sh.match(records[0], False, False, slot=0)
    sync()
    t0 = time.perf_counter()
    win, allrec, per = sh.match(records[0], False, False, slot=0)
    sync()
    one_shot = time.perf_counter() - t0
    hyp_local = int(per.array["hypotheses"].sum()) if per else 0
    local_best = None
    if per:
        resp = per.array["response"]
        local_best = (float(resp.max()), lo + int(np.argmax(resp)))
    # pipelined repetitions
    w = once(0)
    if sh.batch is not None:
        sh.batch.wait(0, per_chain=False)
    if w is not None:
        w.wait()
    sync()

    def timed_pass():
        t0 = time.perf_counter()
        works = [once(r) for r in range(reps)]
        sync()
        dt_ = (time.perf_counter() - t0) / reps
        for r in range(reps):
            if sh.batch is not None:
                sh.batch.wait(r, per_chain=False)
            if works[r] is not None:
                works[r].wait()
        return dt_
    # first pass: 15 of the 16 result slots meet the batch for the first time (the host builds and plans the 40 961-scan
    # call for each: 2-4 ms, next to 5 ms of GPU); second pass: every slot replays its plan (0.13 ms of host per enqueue)
    dt_first = timed_pass()
    dt = timed_pass()
    # the per-GPU share of the 8-GPU run: this rank's first 512 chains alone (strong scaling predicted from one GPU: t(4096) / (8 t(512)))
    shard = None
    if world == 1 and len(chains) >= 4096 and not args.no_production_legs:  # (profiling runs: only launches of the config's own size)
        sb = loop_m.make_batch(query, chains[:512])
        srec = torch.zeros((reps, ymdist.RECORD), dtype=torch.float64, device="cuda")
        for r in range(reps):
            sb.run_async(False, False, slot=r, chain_id_base=0, dev_best_out=srec[r].data_ptr())
        for r in range(reps):
            sb.wait(r, per_chain=False)
        sync()
        t0 = time.perf_counter()
        for r in range(reps):
            sb.run_async(False, False, slot=r, chain_id_base=0, dev_best_out=srec[r].data_ptr())
        sync()
        sdt = (time.perf_counter() - t0) / reps
        for r in range(reps):
            sb.wait(r, per_chain=False)
        shard = {"chains": 512, "ms_per_query": sdt * 1e3, "predicted_strong_scaling_efficiency_at_8_gpus": dt / (8.0 * sdt),
                 "what": "one rank's shard of the 8-GPU run (512 of the 4096 chains) on this GPU, enqueues back to back; the prediction leaves out the "
                         "all-gather of eight 64-byte records.  No multi-GPU run was measured"}
    # the dominant kernel of this config (the gather correlate) against its counters: duration measured here (HIP events on
    # the matcher's stream around the correlate stage, lists excluded), counters replayed from profiles/counters.json
    # (EVERY rank runs the pass: it holds the all-gathers of its enqueues.  Until round 6 only rank 0 did -- sixteen collectives the other ranks
    #  never entered: the N > 1 path would have hung here; found by the two-rank dry run on one GPU, scripts/dev/r06_two_ranks_bench.sh)
    roof = None
    loop_m.profile(True)
    timed_pass()
    corr_ms, corr_n = loop_m.profile_read(0)
    loop_m.profile_read(1); loop_m.profile_read(2)
    loop_m.profile(False)
    if rank == 0:
        if corr_n:
            # (round 5: the kernel funnels with v_perm like the region correlate -- that loop body's measured peak; its name carries
            #  the launch bound since then, the tree's older counter files do not)
            roof = (replayed_roofline("cfg4", "ym::gather_kernel<1, 2, 4, 1024>", corr_ms / corr_n * 1e-3, "region_correlate_body") or
                    replayed_roofline("cfg4", "ym::gather_kernel<1, 2, 4>", corr_ms / corr_n * 1e-3, "region_correlate_body_alignbyte_form"))
            if roof is not None:
                nbeams = 1081
                roof["algorithmic_bytes_per_launch"] = float(hi - lo) * 41 * 41 * 21 * nbeams  # one byte per valid beam and hypothesis
                roof["algorithmic_GBps"] = roof["algorithmic_bytes_per_launch"] / (corr_ms / corr_n * 1e-3) / 1e9
    t = torch.tensor([dt, float(hyp_local)], dtype=torch.float64, device="cuda")
    if dist is not None:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, hyp_total = float(tmax[0].item()), float(t[1].item())
    else:
        hyp_total = float(hyp_local)
    g = gathered[reps - 1].view(world, ymdist.RECORD).cpu().numpy()
    wi = ymdist.pick_best(g)
    winner = {"chain": int(g[wi, 1]), "response": float(g[wi, 0]), "pose": [float(v) for v in g[wi, 2:5]]}
    if local_best is not None and world == 1:
        assert winner["chain"] == local_best[1] and winner["response"] == local_best[0], (winner, local_best)
    assert float(win[1].item()) == g[wi, 1]
    return {"chains": args.cfg4_chains, "chains_per_gpu": hi - lo, "lattice": "41x41x21", "scaling": "strong",
            "ms_per_query": dt * 1e3, "ms_per_query_first_use_of_the_slots": dt_first * 1e3, "one_shot_ms_incl_results": one_shot * 1e3,
            "chain_matches_per_s": args.cfg4_chains / dt, "hypotheses_per_s": hyp_total / dt,
            "hypotheses": hyp_total, "winner": winner, "roofline": roof, "cfg4_shard_512": shard,
            "collective": "all_gather of one 64-byte best record per rank" if dist is not None else "none"}


def leg_cfg5(device, query, chain, rank, world, torch, dist):
    """BASELINE configs[4]: the stress lattice (search 2.0 m at 0.005 m, +-0.785 rad) as one match on one GPU; with
    N > 1 also split over the ranks by coarse angle (grid replicated, response volume all-gathered over RCCL)"""
    from yag_slam_amd import dist as ymdist
    from yag_slam_amd.scan_matching import ScanMatcher
    m = ScanMatcher(STRESS_CONFIG, device=device)
    out = {}
    r = None
    if rank == 0:
        r = m.match_scan(query, chain, True, True)
        for _ in range(3):
            m.match_scan(query, chain, True, True)
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            m.match_scan(query, chain, True, True)
        dt = (time.perf_counter() - t0) / n
        m.profile(True)
        for _ in range(10):
            m.match_scan(query, chain, True, True)
        corr_ms, corr_n = m.profile_read(0)
        m.profile(False)
        cd, nq = r.meta["coarse_dims"], r.meta["n_query_points"]
        corr_s = corr_ms / max(corr_n, 1) * 1e-3
        alg = cd[0] * cd[1] * cd[2] * nq
        out = {"lattice": "%dx%dx%d + %dx%dx%d" % (tuple(cd) + tuple(r.meta["fine_dims"])),
               "hypotheses_per_match": r.meta["hypotheses"], "us_per_match": dt * 1e6, "scan_matches_per_s": 1.0 / dt,
               "hypotheses_per_s": r.meta["hypotheses"] / dt, "correlate_kernel_us": corr_s * 1e6,
               "correlate_algorithmic_GBps": alg / corr_s / 1e9, "correlate_frac_of_hbm_peak": alg / corr_s / 1e9 / HBM_PEAK_GBS}
        # the direct correlate (duration measured here) and the one-block
        # step of the order-dependent smear rule (duration from the profiled run), against their counters
        out["roofline"] = replayed_roofline("cfg5", "ym::correlate_kernel<2, 16, 1>", corr_s, "generic")
        out["roofline_select"] = replayed_roofline("cfg5", "ym::select_relax_kernel", None, "generic", cus=1)
        if out["roofline_select"]:
            out["roofline_select"]["note"] = ("the one-block step of the order-dependent smear rule (a chain of dependent decisions; its parallel steps are "
                                              "select_hash_kernel and select_neighbours_kernel): fractions are of that CU; no resource is the bound, the latency of its LDS round trips is")
    if dist is not None:  # (also the one-rank dry run of the RCCL path, YM_BENCH_FORCE_DIST)
        sp = ymdist.AngleSplitMatcher(m, rank, world)
        g = sp.match_scan(query, chain, True, True)
        for _ in range(3):
            sp.match_scan(query, chain, True, True)
        torch.cuda.synchronize()
        dist.barrier()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            sp.match_scan(query, chain, True, True)
        torch.cuda.synchronize()
        dist.barrier()
        dts = (time.perf_counter() - t0) / n
        t = torch.tensor([dts], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            same = (g.response == r.response and g.covariance == r.covariance and
                    (g.best_pose.x, g.best_pose.y, g.best_pose.euler[-1]) == (r.best_pose.x, r.best_pose.y, r.best_pose.euler[-1]))
            out["split_by_angle"] = {"gpus": world, "us_per_match": float(t.item()) * 1e6,
                                     "hypotheses_per_s": r.meta["hypotheses"] / float(t.item()),
                                     "identical_to_one_gpu": bool(same), "angles_per_gpu": sp.per,
                                     "collective": "all_gather of %d doubles + all_reduce(MAX) of %d doubles per match" % (
                                         sp.resp.numel(), sp.probs.numel())}
    m.close()
    return out


# ------------------------------------------------------------------------------------------------ main
def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if os.environ.get("YM_BENCH_WATCHDOG"):  # development: after that many seconds every thread's stack goes to stderr and the rank exits
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["YM_BENCH_WATCHDOG"]), exit=True)
    # stdout carries exactly ONE line, the JSON; libraries that print there (RCCL's version banner) go to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # development: YM_BENCH_ONE_GPU=1 puts every rank on device 0 and moves the collectives over gloo (records staged through the host) --
    # the code path of --gpus N with N real ranks on a box with ONE GPU (RCCL refuses two ranks on one device).  A dry run of the
    # N > 1 logic (sharding, aggregation, barriers), not a scaling measurement: the ranks share the device.
    one_gpu = os.environ.get("YM_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    legs = set(args.only.split(",")) if args.only else {"cfg2x", "single", "cfg3", "cfg4", "cfg5", "cpu"}
    if args.no_cpu_baseline:
        legs.discard("cpu")
    args.batch = max(args.launch_batch, args.batch // args.launch_batch * args.launch_batch)

    gen = generate_inputs(args, rank, world, legs)  # host only; fork pool; nothing has touched the GPU yet

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libyagmatch has no CPU fallback")
    torch.cuda.set_device(local_rank)
    # one explicit stream for everything (matcher launches, RCCL, copies).  Not torch's default stream: that is the null
    # stream, whose handle (0) means "the matcher's own stream" to ym_set_stream.
    torch.cuda.set_stream(torch.cuda.Stream())
    dist = None
    if world > 1 or os.environ.get("YM_BENCH_FORCE_DIST"):  # the env var exercises the RCCL path with one rank
        import torch.distributed as dist
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    class _Done(object):  # (what an async collective returns, for the host-staged form)
        def wait(self):
            return True

    def all_gather_records(out, mine):
        """all_gather_into_tensor of device tensors: RCCL, asynchronous -- or, in the one-GPU dry run, through host copies over gloo"""
        if not one_gpu:
            return dist.all_gather_into_tensor(out, mine, async_op=True)
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, mine.cpu())
        out.copy_(h)
        return _Done()

    os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # cpu_baseline leg: no spinning OpenMP workers
    from yag_slam_amd import dist as ymdist
    from yag_slam_amd import synth
    from yag_slam_amd.models import ScanBlock, native_many
    from yag_slam_amd.scan_matching import MatchBatch, ScanMatcher

    props = torch.cuda.get_device_properties(local_rank)
    DEVICE["cus"] = int(props.multi_processor_count)
    DEVICE["clock_hz"] = float(getattr(props, "clock_rate", 2400000)) * 1e3  # (kHz)
    m = ScanMatcher(None, device=local_rank)
    stream = torch.cuda.current_stream()
    m.set_stream(stream.cuda_stream)
    if args.corr_u > 0:
        m.debug_option(3, args.corr_u)
    if args.corr_chunks:
        m.debug_option(5, args.corr_chunks)
    if args.corr_pad_lds:
        m.debug_option(4, args.corr_pad_lds)
    if args.corr_region:
        m.debug_option(14, args.corr_region)
    if args.corr_region_na:
        m.debug_option(15, args.corr_region_na)
    # development: YM_BENCH_OPTS="option:value,..." is applied to every matcher of the metric leg (A/B runs of a debug option)
    dev_opts = [tuple(int(x) for x in kv.split(":")) for kv in os.environ.get("YM_BENCH_OPTS", "").split(",") if kv]
    for o_, v_ in dev_opts:
        m.debug_option(o_, v_)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    line = {"metric": "pose hypotheses/sec", "value": None, "unit": "hypotheses/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic", "config": {}, "roofline": None,
            # which form of the cfg2 x batch workload `value` is measured on (rounds 2 - 4: "one_query"; from round 5 on every item has
            # its own query: not like for like with the earlier rounds' values -- the old form is config.by_config.cfg2x_one_query)
            "headline_workload": args.headline}
    line["device"] = {"name": props.name, "compute_units": DEVICE["cus"], "clock_hz": DEVICE["clock_hz"]}
    by_config = {}

    query = chains = None
    if "cfg2" in gen:
        q_ranges, q_prior, base_poses, noisy = gen["cfg2"]
        query = synth.resident_scan(q_ranges, q_prior)
        chains = [[synth.resident_scan(r, p) for r, p in zip(ch, base_poses)] for ch in noisy]
        t_create = time.perf_counter()
        native_many([s for ch in chains for s in ch], local_rank)  # (ym_scans_create: one upload and one launch per 2048 scans)
        line["setup"] = {"scans_created": sum(len(ch) for ch in chains), "seconds": time.perf_counter() - t_create,
                         "how": "models.native_many -> ym_scans_create (round 5: one ym_scan_create launch per scan, ~20 us each)"}

    # ---------------------------------------------------------------- the metric line: cfg2 x batch
    if "cfg2x" in legs:
        LB = args.launch_batch
        E = args.batch // LB  # enqueues per step
        # lanes: independent matchers (own stream + workspace); enqueue e of a step goes to lane e % lanes, so the
        # small tail kernels of one enqueue overlap the next one's big ones
        lanes = [m] + [ScanMatcher(None, device=local_rank) for _ in range(max(1, args.lanes) - 1)]
        for lm in lanes[1:]:
            for o_, v_ in dev_opts:
                lm.debug_option(o_, v_)
        lane_streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in lanes[1:]]
        for lm, ls in zip(lanes[1:], lane_streams[1:]):
            lm.set_stream(ls.cuda_stream)
        NL = len(lanes)
        batches = [lanes[e % NL].make_batch(query, chains[e * LB:(e + 1) * LB]) for e in range(E)]
        # the same chains, every item with its OWN query (own readings, own prior): ym_pairs_create
        pbatches = dqueries = None
        if "cfg2_queries" in gen:
            dq_ranges, dq_prior = gen["cfg2_queries"]
            dqueries = [synth.resident_scan(r, p) for r, p in zip(dq_ranges, dq_prior)]
            native_many(dqueries, local_rank)
            pbatches = [lanes[e % NL].make_pairs_batch(dqueries[e * LB:(e + 1) * LB], chains[e * LB:(e + 1) * LB]) for e in range(E)]
        headline_distinct = args.headline == "distinct_queries" and pbatches is not None
        nslots = min(64, 2 * ((E + NL - 1) // NL))
        nbuf = 2
        records = torch.zeros((nbuf, E, ymdist.RECORD), dtype=torch.float64, device="cuda")
        gathered = torch.zeros((nbuf, world * E * ymdist.RECORD), dtype=torch.float64, device="cuda")
        works = [None] * nbuf
        used = [[None] * nslots for _ in lanes]  # per lane: which batch object a slot's call belongs to
        counter = [0] * NL

        def step(i, bs, one_at_a_time=False, marks=None):
            b = i % nbuf
            if works[b] is not None:
                works[b].wait()  # its gathered records are about to be overwritten
                works[b] = None
            for e in range(E):
                ln = e % NL
                s = counter[ln] % nslots
                counter[ln] += 1
                if used[ln][s] is not None:
                    used[ln][s].wait(s, per_chain=False)  # recycle the slot (long since finished)
                bs[e].run_async(True, True, slot=s, chain_id_base=rank * args.batch + e * LB,
                                dev_best_out=records[b, e].data_ptr())
                used[ln][s] = bs[e]
                if one_at_a_time:  # (the profiling pass: a kernel's duration means something only while nothing runs beside it)
                    bs[e].wait(s, per_chain=False)
                    used[ln][s] = None
            if marks is not None:  # when this step's work is complete on the device: the latest of its lanes' marks
                evs = [torch.cuda.Event(enable_timing=True) for _ in lane_streams]
                for ev_, ls in zip(evs, lane_streams):
                    ev_.record(ls)
                marks.append(evs)
            if dist is not None:
                # cross-rank arg-max payload: E 64-byte records per rank.  Asynchronous: RCCL's stream waits for this
                # step's records (on every lane's stream), the launch streams go straight on to the next step
                for ls in lane_streams[1:]:
                    lane_streams[0].wait_stream(ls)
                works[b] = all_gather_records(gathered[b], records[b].view(-1))

        def drain():
            for ln in range(NL):
                for s in range(nslots):
                    if used[ln][s] is not None:
                        used[ln][s].wait(s, per_chain=False)
                        used[ln][s] = None
            for b in range(nbuf):
                if works[b] is not None:
                    works[b].wait()
                    works[b] = None

        def timed_steps(bs, steps, warmup):
            """`warmup` untimed steps, then `steps` steps between two barriers; returns (seconds, max over the ranks; per-step
            completion intervals in ms on this rank, from HIP events on the lanes' streams)"""
            import gc
            for i in range(warmup):
                step(i, bs)
            drain()
            gc.collect()
            gc.disable()  # (a generation-2 pass over the 170 000 resident scan objects takes 80 ms: not inside a 0.9 s measurement)
            barrier()
            marks = []
            start = torch.cuda.Event(enable_timing=True)
            start.record(lane_streams[0])
            t0 = time.perf_counter()
            for i in range(steps):
                step(i, bs, marks=marks)
            barrier()
            dt_ = time.perf_counter() - t0
            gc.enable()
            if dist is not None:  # the last step's gather carries this rank's own records in their place
                b_last = (steps - 1) % nbuf
                works[b_last].wait()
                got = gathered[b_last].view(world, E, ymdist.RECORD)[rank]
                assert torch.equal(got, records[b_last])
            drain()
            t = torch.tensor([dt_], dtype=torch.float64, device="cuda")
            if dist is not None:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ends = [max(start.elapsed_time(ev_) for ev_ in evs) for evs in marks]
            per_step = [b_ - a_ for a_, b_ in zip([0.0] + ends[:-1], ends)]
            return float(t.item()), per_step

        def spread(per_step):
            v = sorted(per_step)
            return {"min": v[0], "median": v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2]), "max": v[-1],
                    "what": "ms between the completion of consecutive steps on the device (HIP events on the lanes' streams; the steps are "
                            "enqueued back to back, so the first one also carries the pipeline's fill)"}

        # correctness of what is timed: first enqueue against the single-call path
        per, best, bi = (batches[0].run_async(True, True, slot=0) or batches[0].wait(0))
        hyp_per_match = per[0].meta["hypotheses"]
        assert all(p.meta["hypotheses"] == hyp_per_match for p in per)
        hyp_step = hyp_per_match * args.batch
        ref = m.match_scan(query, chains[0], True, True)
        checked = not os.environ.get("YM_BENCH_SKIP_CHECK")  # (development: timing-only builds of the library give wrong sums)
        assert not checked or (ref.response == per[0].response and ref.covariance == per[0].covariance)
        if pbatches is not None and checked:  # ... and of the distinct-query form: three of its items against their single calls
            pper, _, _ = (pbatches[0].run_async(True, True, slot=0) or pbatches[0].wait(0))
            assert all(p.meta["hypotheses"] == hyp_per_match for p in pper)
            for i_ in (0, LB // 2, LB - 1):
                ref = m.match_scan(dqueries[i_], chains[i_], True, True)
                assert ref.response == pper[i_].response and ref.covariance == pper[i_].covariance, i_

        hb = pbatches if headline_distinct else batches
        dt, per_step = timed_steps(hb, args.steps, args.warmup)

        # roofline of the dominant kernel (coarse correlate): HIP events on the launch stream, second pass
        for lm in lanes:
            lm.profile(True)
        for i in range(min(args.steps, 4)):
            step(i, hb, one_at_a_time=NL > 1)
        drain()
        corr_ms = corr_n = call_ms = call_n = 0
        for lm in lanes:
            a_, b_ = lm.profile_read(0)
            c_, d_ = lm.profile_read(2)
            corr_ms, corr_n, call_ms, call_n = corr_ms + a_, corr_n + b_, call_ms + c_, call_n + d_
            lm.profile(False)

        nq = per[0].meta["n_query_points"]
        cd = per[0].meta["coarse_dims"]
        alg_bytes = LB * cd[0] * cd[1] * cd[2] * nq  # 1 grid byte per valid beam per coarse hypothesis
        corr_s = corr_ms / max(corr_n, 1) * 1e-3
        achieved = alg_bytes / corr_s / 1e9
        ms_per_step = dt / args.steps * 1e3
        line["value"] = hyp_step * world * args.steps / dt
        line["ms_per_step"] = ms_per_step
        line["ms_per_step_spread"] = spread(per_step)
        line["config"] = {
            "workload": ("cfg2 x batch: %d INDEPENDENT single-match problems per step per GPU -- every item its own 1081-beam query (own "
                         "readings, own prior) against its own 10-scan chain" if headline_distinct else
                         "cfg2 x batch: ONE 1081-beam query against %d distinct 10-scan chains per step per GPU (every item its own "
                         "correlation grid and search; the query object and its pair lists are shared by the items of an enqueue)") % args.batch +
                        ", search 0.5 m / 0.349 rad, resolution 0.01, coarse 26x26x21 + fine 3x3x11, penalty "
                        "on, Karto semantics, issued as %d enqueues of %d, resident batches enqueued again every step (plan replay, warm point cache)" % (E, LB),
            "queries_per_enqueue": LB if headline_distinct else 1,
            "batch_per_gpu": args.batch, "launch_batch": LB, "lanes": NL, "hypotheses_per_match": hyp_per_match,
            "scan_matches_per_s": args.batch * world * args.steps / dt,
            "timed_seconds": dt,
            "collective": "all_gather of %d 64-byte best records per rank per step" % E if dist is not None else "none",
        }
        # the other form of the workload, same lanes, same step function, fewer steps
        if pbatches is not None and not args.only_headline:
            ob = batches if headline_distinct else pbatches
            odt, oper = timed_steps(ob, max(4, args.steps // 2), 2)
            osteps = max(4, args.steps // 2)
            by_config["cfg2x_one_query" if headline_distinct else "cfg2x_distinct_queries"] = {
                "what": ("every item against the SAME query object" if headline_distinct else
                         "every item its OWN query (own readings, own prior): %d distinct queries per enqueue, ym_pairs_create" % LB) +
                        "; same chains, lanes and step function as the metric line",
                "queries_per_enqueue": 1 if headline_distinct else LB, "steps": osteps, "ms_per_step": odt / osteps * 1e3,
                "ms_per_step_spread": spread(oper), "hypotheses_per_s": hyp_step * world * osteps / odt,
                "scan_matches_per_s": args.batch * world * osteps / odt,
                "ratio_to_metric_line": (hyp_step * world * osteps / odt) / line["value"]}
        # the same step function on HETEROGENEOUS problems: (query, chain) pairs along the cfg3 trajectory -- query i = scan 10 + i at its
        # odometry prior, chain i = the ten scans before it at their true poses: every item its own pose, heading, window contents,
        # region lists and patch counts (the metric's items all sit within 5 cm of ONE pose, as SURVEY 8(d) specifies them)
        if "traj" in gen and pbatches is not None:
            t_truth, t_prior = synth.loop_trajectory(args.batch + 10)
            tscans = [synth.resident_scan(r_, p_) for r_, p_ in zip(gen["traj"], t_truth)]
            tq = [synth.resident_scan(gen["traj"][10 + i_], t_prior[10 + i_]) for i_ in range(args.batch)]
            native_many(tscans + tq, local_rank)
            tb = [lanes[e % NL].make_pairs_batch(tq[e * LB:(e + 1) * LB], [tscans[i_:i_ + 10] for i_ in range(e * LB, (e + 1) * LB)]) for e in range(E)]
            tper, _, _ = (tb[0].run_async(True, True, slot=0) or tb[0].wait(0))
            for i_ in (0, LB // 3, LB - 1):  # (what is timed is right: three items against their single calls)
                ref = m.match_scan(tq[i_], tscans[i_:i_ + 10], True, True)
                assert not checked or (ref.response == tper[i_].response and ref.covariance == tper[i_].covariance), i_
            tsteps = max(4, args.steps // 2)
            tdt, tps = timed_steps(tb, tsteps, 2)
            thyp = float(tper.array["hypotheses"].sum()) / LB * args.batch
            for lm in lanes:
                lm.profile(True)
            for i in range(2):
                step(i, tb, one_at_a_time=NL > 1)
            drain()
            tk = [0.0] * 6
            for lm in lanes:
                for w_ in range(3):
                    a_, b_ = lm.profile_read(w_)
                    tk[2 * w_] += a_
                    tk[2 * w_ + 1] += b_
                lm.profile(False)
            by_config["cfg2x_trajectory"] = {
                "what": "%d (query, chain) pairs per step along the cfg3 trajectory (query i = scan 10 + i at its odometry prior against the ten scans "
                        "before it at their true poses): heterogeneous poses, headings and windows; same lanes and step function as the metric line" % args.batch,
                "steps": tsteps, "ms_per_step": tdt / tsteps * 1e3, "ms_per_step_spread": spread(tps), "hypotheses_per_s": thyp * world * tsteps / tdt,
                "scan_matches_per_s": args.batch * world * tsteps / tdt, "ratio_to_metric_line": (thyp * world * tsteps / tdt) / line["value"],
                "region_correlate_us": tk[0] / max(tk[1], 1) * 1e3, "raster_us": tk[2] / max(tk[3], 1) * 1e3, "call_us_gpu": tk[4] / max(tk[5], 1) * 1e3,
                "metric_line_region_correlate_us": corr_s * 1e6, "metric_line_call_us_gpu": call_ms / max(call_n, 1) * 1e3,
                "mean_response": float(tper.array["response"].mean())}
            del tb
        region = LB >= 8 and args.corr_region != 1
        # (the default region correlate of a large batch stages from the row-major window: template argument WIN = true)
        kernel = "ym::correlate_region_kernel<8, true>" if region else "ym::correlate_kernel<2, 16, 4>"
        if region and any(o_ == 39 and v_ for o_, v_ in dev_opts):
            kernel = "ym::correlate_region_kernel<8>"
        step_alg = hyp_step * nq  # coarse + fine lattice points x one byte per valid beam
        rl = replayed_roofline("cfg2x", kernel, corr_s) if LB == 4096 else None
        if rl is None:
            rl = {"bound": "hbm", "kernel": kernel, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                  "kernel_us": corr_s * 1e6, "replayed_from": None}
        rl.update({
            "call_us_gpu": call_ms / max(call_n, 1) * 1e3,
            # one grid byte per valid beam and hypothesis, against the HBM peak.  Not a roofline fraction:
            # the bytes are gathered from LDS, where a staged byte is read ~19 times
            "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_GBps": achieved, "algorithmic_frac": achieved / HBM_PEAK_GBS,
            # the whole step against the same figure: every kernel of the call, launch gaps and host work included
            "algorithmic_frac_step": step_alg / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
        })
        # the other kernels of an enqueue, from the same counter passes (durations: the kernel-trace pass of the profiled run, one
        # enqueue at a time -- these kernels are not timed live): what each is bound by, at a glance
        ctr = profile_json("counters.json")
        try:
            others = {}
            for kn, kv in ctr["workloads"]["cfg2x"]["kernels"].items():
                if not kn.startswith("ym::") or kn == kernel or kv.get("us", 0.0) < 30.0 or "structure_kernel" in kn or "points_kernel" in kn:
                    continue
                us = kv["us"] if "raster" not in kn else min(kv["us"], kv.get("min_us", kv["us"]) * 1.05)  # (the raster's mean holds a first-call outlier)
                clk = us * 1e-6 * DEVICE["clock_hz"] * DEVICE["cus"]
                hb = kv.get("FETCH_SIZE", 0.0) * 2048.0 + kv.get("WRITE_SIZE", 0.0) * 1024.0
                others[kn] = {"us_in_the_profiled_run": us, "valu_per_cu_clk": kv.get("SQ_INSTS_VALU", 0.0) / clk, "lds_busy": kv.get("SQ_LDS_IDX_ACTIVE", 0.0) / clk,
                              "vector_l1_line_visits_per_cu_clk": kv.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / clk, "hbm_GBps": hb / (us * 1e-6) / 1e9,
                              "hbm_frac": hb / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                              "waiting_share_of_wave_time": kv.get("SQ_WAIT_ANY", 0.0) / kv["SQ_WAVE_CYCLES"] if kv.get("SQ_WAVE_CYCLES") else None}
            rl["other_kernels_of_an_enqueue"] = others
        except (KeyError, TypeError):
            pass
        line["roofline"] = rl
        line["config"]["point_cache"] = dict(zip(("hits", "misses"), m.cache_stats()))

        # ---- what production would see of this workload (same run, after the timed region of the metric)
        def timed_enqueues(bs, n, m=m):
            """n back-to-back enqueues cycling over the batch objects `bs` on matcher m; seconds per enqueue, GPU ms per call"""
            for i in range(min(8, n)):  # (every result slot the timed loop uses has met a call of this size: its pinned buffers exist)
                bs[i % len(bs)].run_async(True, True, slot=i % 8)
            for sl in range(min(8, n)):
                bs[0].wait(sl, per_chain=False)
            torch.cuda.synchronize()
            m.profile(True)
            t1 = time.perf_counter()
            for i in range(n):
                if i >= 8:
                    bs[0].wait(i % 8, per_chain=False)
                bs[i % len(bs)].run_async(True, True, slot=i % 8)
            for sl in range(min(8, n)):
                bs[0].wait(sl, per_chain=False)
            torch.cuda.synchronize()
            sec = (time.perf_counter() - t1) / n
            cms, cn = m.profile_read(2)
            m.profile_read(0)
            m.profile_read(1)
            m.profile(False)
            return sec, cms / max(cn, 1)

        def leg_sweep():
            out = {"what": "1 query x N distinct chains per enqueue (resident, warm point cache), enqueues back to back"}
            for n in (8, 64, 512, 4096):
                if n > args.batch:
                    continue
                sec, gpu_ms = timed_enqueues([m.make_batch(query, chains[:n])], 64 if n <= 512 else 12)
                out[str(n)] = {"hypotheses_per_s": hyp_per_match * n / sec, "us_per_enqueue": sec * 1e6, "gpu_us_per_enqueue": gpu_ms * 1e3,
                               "us_per_match": sec * 1e6 / n}
            return out

        def leg_fresh_query():
            # a loop-closure query is NEW every call: two queries (the same readings at two priors) alternate, so every
            # enqueue copies its descriptor, projects its query and sorts the query's (beam, angle) pairs again
            q2 = synth.resident_scan(q_ranges, (q_prior[0] + 0.013, q_prior[1] - 0.007, q_prior[2] + 0.004))
            q2.native(local_rank)
            n = min(LB, args.batch)
            sec, gpu_ms = timed_enqueues([m.make_batch(query, chains[:n]), m.make_batch(q2, chains[:n])], 16)
            return {"chains_per_enqueue": n, "hypotheses_per_s": hyp_per_match * n / sec, "us_per_enqueue": sec * 1e6,
                    "gpu_us_per_enqueue": gpu_ms * 1e3, "what": "two queries alternate: no enqueue repeats its predecessor's descriptor"}

        def leg_cold():
            # every chain re-posed before every enqueue: the point cache misses on every base scan (points_kernel projects
            # them again), the descriptor is new, no plan is replayed.  The pose writes are ONE ym_scans_set_poses call per
            # enqueue (what a graph optimisation does to every vertex, graph_slam.py:263-272); round 3 made one ctypes call
            # per scan and spent 1.7 ms of Python on 5120 of them against 0.85 ms of GPU.
            import ctypes as C_
            n = min(LB, args.batch)
            b1 = m.make_batch(query, chains[:n])
            flat = [s_ for ch in chains[:n] for s_ in ch]
            poses = np.array([(s_.corrected_pose.x, s_.corrected_pose.y, s_.corrected_pose.euler[-1]) for s_ in flat], dtype=np.float64)
            hs = (C_.c_void_p * len(flat))(*[s_.native(local_rank) for s_ in flat])
            setp = m._lib.ym_scans_set_poses
            dp = C_.POINTER(C_.c_double)
            moved = [np.ascontiguousarray(poses + np.array([1e-4 * k, 0.0, 0.0])) for k in (1, 2)]
            m.profile(True)
            reps, host = 12, 0.0
            for i in range(2):  # (the slots meet the batch once, untimed)
                setp(hs, moved[i % 2].ctypes.data_as(dp), len(flat))
                b1.run_async(True, True, slot=i % 8)
            for sl in range(2):
                b1.wait(sl, per_chain=False)
            torch.cuda.synchronize()
            m.profile_read(2)
            t1 = time.perf_counter()
            for i in range(reps):
                th = time.perf_counter()
                setp(hs, moved[i % 2].ctypes.data_as(dp), len(flat))
                host += time.perf_counter() - th
                if i >= 8:
                    b1.wait(i % 8, per_chain=False)
                b1.run_async(True, True, slot=i % 8)
            for sl in range(8):
                b1.wait(sl, per_chain=False)
            torch.cuda.synchronize()
            sec = (time.perf_counter() - t1) / reps
            cms, cn = m.profile_read(2)
            m.profile(False)
            setp(hs, poses.ctypes.data_as(dp), len(flat))  # the chains go back to where the other legs expect them
            return {"chains_per_enqueue": n, "scans_reposed_per_enqueue": len(flat), "gpu_us_per_enqueue": cms / max(cn, 1) * 1e3,
                    "hypotheses_per_s_gpu": hyp_per_match * n / (cms / max(cn, 1) * 1e-3),
                    "wall_us_per_enqueue": sec * 1e6, "hypotheses_per_s": hyp_per_match * n / sec,
                    "host_pose_writes_us_per_enqueue": host / reps * 1e6,
                    "what": "every base scan re-posed (one ym_scans_set_poses call) before every enqueue: point cache misses on all of them, no plan replay"}

        def leg_alternating():
            # one matcher serving both kinds of call in turn: a single match (correlates from the column planes) between two
            # enqueues of a large batch (window only).  Until round 5 each change of kind dropped everything the matcher knew of
            # its windows' memory: the batch after a single match rasterised every tile of every item again.
            n = min(LB, args.batch)
            b1 = m.make_batch(query, chains[:n])
            for _ in range(2):
                b1.run_async(True, True, slot=0)
                b1.wait(0, per_chain=False)
                m.match_scan(query, chains[1], True, True)
            torch.cuda.synchronize()
            reps, tb, ts = 6, 0.0, 0.0
            for _ in range(reps):
                t1 = time.perf_counter()
                b1.run_async(True, True, slot=0)
                b1.wait(0, per_chain=False)
                t2 = time.perf_counter()
                m.match_scan(query, chains[1], True, True)
                t3 = time.perf_counter()
                tb += t2 - t1
                ts += t3 - t2
            t1 = time.perf_counter()
            for _ in range(reps):
                b1.run_async(True, True, slot=0)
                b1.wait(0, per_chain=False)
            only = (time.perf_counter() - t1) / reps
            return {"chains_per_enqueue": n, "us_per_batch_after_a_single_match": tb / reps * 1e6, "us_per_single_match_after_a_batch": ts / reps * 1e6,
                    "us_per_batch_alone": only * 1e6, "what": "synchronous calls on ONE matcher: enqueue of %d chains + wait, one match_scan, and again" % n}

        def leg_yagpy():
            # the metric workload in the REFERENCE'S PYTHON semantics (Scan2DMatcherPy.match_scan, /root/reference/yag_slam/scan_matching.py:175-222:
            # coarse 25 x 25 x 10 + fine ~5 x 5 x 10): the semantics whose integer sum volumes are pinned on reference-made vectors, through the
            # same production correlate kernel as the metric line (items whose roundings yag_lattice_kernel proves to form a lattice), beside
            # the pair-by-pair kernel that evaluates the Python rule as written (debug option 46 = 0)
            n = min(LB, args.batch)
            out = {"what": "%d independent matches per enqueue (own query each), \"yagpy\" semantics; the coarse sums of every item proven regular "
                           "come from correlate_region_kernel, of the others from yag_score_kernel" % n, "items_per_enqueue": n}
            for label, fast in (("production_kernels", 1), ("pairwise_kernel", 0)):
                ym = ScanMatcher(None, semantics="yagpy", device=local_rank)
                ym.set_stream(stream.cuda_stream)
                ym.debug_option(46, fast)
                try:
                    pb = ym.make_pairs_batch(dqueries[:n], chains[:n])
                    per_, _, _ = (pb.run_async(True, True, slot=0) or pb.wait(0))
                    hyp = int(sum(p_.meta["hypotheses"] for p_ in per_))
                    sec, gpu_ms = timed_enqueues([pb], 10 if fast else 4, m=ym)
                    cnt = ym.debug_counters()
                    out[label] = {"hypotheses_per_s": hyp / sec, "us_per_enqueue": sec * 1e6, "gpu_us_per_enqueue": gpu_ms * 1e3, "hypotheses_per_enqueue": hyp,
                                  "scan_matches_per_s": n / sec, "items_through_the_production_kernels": cnt["yag_fast_items"],
                                  "items_that_fell_back": cnt["yag_fallback_items"], "pairs_checked_exhaustively": cnt["yag_pairs_checked"],
                                  "coarse_correlate": cnt["last_correlate"], "first_item": {"response": per_[0].response, "coarse_dims": per_[0].meta["coarse_dims"]}}
                finally:
                    ym.close()
            a_, b_ = out["production_kernels"], out["pairwise_kernel"]
            out["identical_first_item"] = a_["first_item"] == b_["first_item"]
            out["speedup_over_the_pairwise_kernel"] = b_["us_per_enqueue"] / a_["us_per_enqueue"]
            return out

        def leg_fresh_scans():
            # the realistic form of the metric workload: N robots, one FRESH scan each per step.  Every enqueue creates its 4096 query
            # scans from host arrays (ym_scans_create: host-side box / longest reading / beam spacing per scan, one upload, one launch
            # per 2048 scans), builds the pairs batch, enqueues it, and -- once its results are in -- destroys batch and scans
            # (ym_scans_destroy).  Creation, batch building and destruction are INSIDE the timed loop.  One Python thread per lane (ctypes
            # releases the GIL): what a node serving many robots does.  The chains stay resident (the robots' running chains).
            import threading
            n = min(LB, args.batch)
            nl = max(1, args.lanes)
            sensor = (synth.MIN_ANGLE, synth.MAX_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE, synth.MAX_RANGE, synth.RANGE_THRESHOLD)
            R = np.ascontiguousarray(np.stack(dq_ranges[:args.batch]))
            Pq = np.ascontiguousarray(np.asarray(dq_prior[:args.batch], dtype=np.float64))
            groups = max(1, args.batch // n)
            chain_h = [np.array([s_.native(local_rank) for ch in chains[g * n:(g + 1) * n] for s_ in ch], dtype=np.uint64) for g in range(groups)]
            offs = np.arange(n + 1, dtype=np.int32) * len(chains[0])
            lms = [ScanMatcher(None, device=local_rank) for _ in range(nl)]
            t_create = [0.0] * nl
            results = [None] * nl

            def lane(li, reps):
                lm = lms[li]
                live = []
                tc = 0.0
                for i in range(reps):
                    g = (li + i * nl) % groups
                    t1 = time.perf_counter()
                    blk = ScanBlock(R[g * n:(g + 1) * n], Pq[g * n:(g + 1) * n], sensor, device=local_rank)
                    tc += time.perf_counter() - t1
                    hb = MatchBatch.from_handles(lm, blk.handles, chain_h[g], offs)
                    if len(live) == 2:  # two enqueues in flight per lane
                        ob, oblk, osl = live.pop(0)
                        ob.wait(osl, per_chain=False)
                        ob.close()
                        oblk.release()
                    hb.run_async(True, True, slot=i % 2)
                    live.append((hb, blk, i % 2))
                last = None
                for ob, oblk, osl in live:
                    last = ob.wait(osl)[0]
                    ob.close()
                    oblk.release()
                t_create[li] = tc / reps
                results[li] = ((li + (reps - 1) * nl) % groups, last)

            def run_lanes(reps):
                threads = [threading.Thread(target=lane, args=(li, reps)) for li in range(nl)]
                t1 = time.perf_counter()
                for t_ in threads:
                    t_.start()
                for t_ in threads:
                    t_.join()
                torch.cuda.synchronize()
                return (time.perf_counter() - t1) / (reps * nl)

            # untimed: every matcher sizes its workspace and meets the chains; the scan pool grows to what the lanes keep in flight (the
            # blocks of destroyed scans come back once every stream has passed them -- the pool never synchronises the device)
            run_lanes(4)
            reps = 12
            sec = run_lanes(reps)
            # the last enqueue of lane 0 against the resident form of the same items (same readings, same priors, same chains)
            g, last = results[0]
            refb = m.make_pairs_batch(dqueries[g * n:(g + 1) * n], chains[g * n:(g + 1) * n])
            refb.run_async(True, True, slot=0)
            want = refb.wait(0)[0]
            same = bool(np.array_equal(want.array["response"], last.array["response"]) and np.array_equal(want.array["pose"], last.array["pose"]) and
                        np.array_equal(want.array["cov"], last.array["cov"]))
            for lm in lms:
                lm.close()
            hps = hyp_per_match * n / sec
            return {"what": "%d NEW query scans per enqueue: created (ym_scans_create), matched against resident chains (ym_pairs_create + run) and destroyed "
                            "(ym_scans_destroy), all inside the timed loop; %d lanes, one Python thread each" % (n, nl),
                    "scans_created_per_enqueue": n, "us_per_enqueue": sec * 1e6, "hypotheses_per_s": hps, "scan_matches_per_s": n / sec,
                    "scan_creation_us_per_enqueue_per_thread": float(np.mean(t_create)) * 1e6, "ratio_to_metric_line": hps / line["value"] * world,
                    "identical_to_the_resident_form": same}

        if rank == 0 and not args.no_production_legs:
            sweep_legs = [("cfg2x_batch_sweep", leg_sweep), ("cfg2x_fresh_query", leg_fresh_query), ("cfg2x_cold", leg_cold),
                          ("cfg2x_alternating", leg_alternating)]
            if dqueries is not None:
                sweep_legs += [("cfg2x_fresh_scans", leg_fresh_scans), ("cfg2x_yagpy", leg_yagpy)]
        else:
            sweep_legs = []
        if args.legs:
            sweep_legs = [(k_, f_) for k_, f_ in sweep_legs if k_ in args.legs.split(",")]
        del batches, pbatches
        for lm in lanes[1:]:
            lm.close()
    else:
        sweep_legs = []

    if dist is not None:
        dist.barrier()

    emitted = [False]

    def emit():
        if rank == 0 and not emitted[0]:
            emitted[0] = True
            line["config"]["by_config"] = by_config
            line["leg_errors"] = sorted(k_ for k_, v_ in by_config.items() if isinstance(v_, dict) and "error" in v_)
            if line["value"] is None:  # a development run of single legs: not a metric line
                line["metric"] = "partial run (--only %s)" % args.only
            os.write(json_fd, (json.dumps(line) + "\n").encode())

    # ---------------------------------------------------------------- the other BASELINE configs, same run
    # (a leg that fails must not take the metric line with it: its entry then carries the error)
    def guarded(name, fn, collective=False):
        # (no cyclic garbage collection inside a leg, as timeit does: with 200 000 scan objects alive a generation-2 pass
        #  takes 80 ms and lands in whichever Python-inclusive figure is being timed -- 10 k instead of 16 k scans per
        #  second in the per-scan cfg3 driver, 88 instead of 11 ms for cfg4's one-shot call)
        import gc
        gc.collect()
        gc.disable()
        try:
            out = fn()
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            out = {"error": "%s: %s" % (type(e).__name__, e)}
            if collective and dist is not None:
                # the other ranks are inside the same collective leg: print what there is, then let the launcher end the job
                if rank == 0:
                    by_config[name] = out
                    emit()
                raise
        finally:
            gc.enable()
        if rank == 0 and out is not None:
            by_config[name] = out

    for name_, fn_ in sweep_legs:
        guarded(name_, fn_)
    if "single" in legs and rank == 0:
        guarded("cfg2_single_match", lambda: leg_single(m, query, chains[0], m.match_scan(query, chains[0], True, True).meta["hypotheses"]))
    if "cfg3" in legs and rank == 0:
        guarded("cfg3_sequential_mapping", lambda: leg_cfg3(m, gen["cfg3"], args.cfg3_scans))
    if "cfg5" in legs and (rank == 0 or world > 1):
        q5, b5 = synth.single_match_scans(gen["scene"])
        guarded("cfg5_stress", lambda: leg_cfg5(local_rank, q5, b5, rank, world, torch, dist), collective=world > 1)
    if "cfg4" in legs:
        def run_cfg4():
            loop_m = ScanMatcher(None, loop=True, device=local_rank)
            try:
                return leg_cfg4(loop_m, gen, args, rank, world, torch, dist)
            finally:
                loop_m.close()
        guarded("cfg4_loop_closure_batch", run_cfg4, collective=world > 1)
    if "cpu" in legs and rank == 0 and world == 1:  # the CPU baseline is a single-GPU-run item
        cb = cpu_baseline()
        try:
            ct = cpu_throughput()
        except Exception as e:  # noqa: BLE001 -- (no compiler on the host, a cgroup that forbids pinning ...: the other two figures still stand)
            import traceback
            traceback.print_exc()
            ct = {"hyp_per_s": None, "cores": None, "logical_cpus": None, "matches_per_s": None, "ms_per_match_per_core": None,
                  "min_max_matches_per_thread": None, "matches": 0, "seconds": 0.0, "error": "%s: %s" % (type(e).__name__, e)}
        line["cpu_baseline"] = {
            "value": cb["single"]["hyp_per_s"], "unit": "hypotheses/s", "cores": 1, "kind": "port",
            "sample": "%d cfg2 matches (coarse+fine, penalty) in %.1f s, oracle/ym_oracle.c karto semantics, "
                      "-O3 -march=native, 1 thread" % (cb["single"]["matches"], cb["single"]["seconds"]),
            "host": {"cpu_model": cpu_model(), "logical_cpus": os.cpu_count() or 1, "physical_cores": ct["cores"],
                     "logical_cpus_in_affinity_mask": ct["logical_cpus"]},
            # what belongs beside a GPU THROUGHPUT figure: independent matches on every physical core at once
            "throughput": {"value": ct["hyp_per_s"], "unit": "hypotheses/s", "cores": ct["cores"], "scan_matches_per_s": ct["matches_per_s"],
                           "ms_per_match_per_core": ct["ms_per_match_per_core"],
                           "matches_per_thread_min_max": ct["min_max_matches_per_thread"],
                           "error": ct.get("error"),
                           "sample": "%d independent cfg2 matches (coarse+fine, penalty) in %.1f s on %s pinned threads, one per physical core, each "
                                     "with its own oracle context (own correlation grid, one thread), oracle/ym_throughput.c + ym_oracle.c karto "
                                     "semantics, -O3 -march=native" % (ct["matches"], ct["seconds"], ct["cores"])},
            "all_cores": {"value": cb["all"]["hyp_per_s"], "cores": cb["all"]["threads"],
                          "serial_fraction": cb["all"]["serial_fraction"],
                          "sample": "%d matches in %.1f s, OpenMP over the coarse lattice; grid clear and rasterisation "
                                    "stay serial as in Karto (serial_fraction = their share of the wall time), host has "
                                    "%d logical cpus" % (cb["all"]["matches"], cb["all"]["seconds"], os.cpu_count() or 1)},
        }
        # (BASELINE.md holds no published number for this metric: vs_baseline stays null; against this run's own CPU figures:)
        if line["value"] and ct.get("hyp_per_s"):
            line["vs_cpu_baseline"] = {"all_physical_cores": line["value"] / ct["hyp_per_s"], "one_thread": line["value"] / cb["single"]["hyp_per_s"],
                                       "what": "value / cpu_baseline.throughput.value and / cpu_baseline.value: context, not credit (a CPU port of the oracle, not the reference's wheel)"}
        by_config["cfg1_cpu_single_match"] = {
            "ms_per_match": cb["single"]["ms_per_match"], "scan_matches_per_s": 1e3 / cb["single"]["ms_per_match"],
            "hypotheses_per_s": cb["single"]["hyp_per_s"], "what": "oracle/ym_oracle.c, karto semantics, 1 thread"}
    emit()
    if dist is not None:
        dist.destroy_process_group()
    if any(isinstance(v_, dict) and "error" in v_ for v_ in by_config.values()):
        sys.exit(3)  # the line is out; a leg failed


if __name__ == "__main__":
    main()
