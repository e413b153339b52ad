#!/usr/bin/env python3
"""bench.py -- pose hypotheses/sec of the MI355X correlative scan matcher.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched under torch.distributed.run)

Workload (BASELINE.json configs[1], "cfg2"): 1081-beam synthetic scans, search 0.5 m / 0.349 rad,
resolution 0.01 m, coarse + fine pass with the odometry penalty.  One STEP = one enqueue of a batch
of `--batch` independent single-match problems of that exact configuration (one query against
`--batch` candidate 10-scan chains, each with its own correlation grid, coarse + fine search,
covariances), all inputs resident in HBM.  With N GPUs every rank runs its own shard of
`--batch` chains per step (weak scaling) and the ranks exchange their best (response, pose) with one
RCCL all-gather per step.  `value` = lattice points scored by all ranks / wall time of K steps.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def build_inputs(batch, rank):
    """cfg2 scans: one query at the odometry prior, `batch` chains of 10 base scans.  Chain c uses
    the cfg2 poses with its own noise seeds so every item rasterises a different grid."""
    from yag_slam_amd import synth
    from yag_slam_amd.models import LocalizedRangeScan
    scene = synth.Scene()
    base_poses, q_truth, q_prior = synth.single_match_poses()
    mk = lambda r, p: LocalizedRangeScan(r, synth.MIN_ANGLE, synth.MAX_ANGLE, synth.ANGLE_INCREMENT, synth.MIN_RANGE,
                                         synth.MAX_RANGE, synth.RANGE_THRESHOLD, p[0], p[1], p[2])
    query = mk(scene.scan_ranges(q_truth, index=10), q_prior)
    exact = [scene.cast(*p) for p in base_poses]
    chains = []
    for c in range(batch):
        rng = np.random.default_rng(100000 * (rank + 1) + c)
        chains.append([mk(e + rng.normal(0.0, synth.SIGMA_RANGE, size=e.shape), p) for e, p in zip(exact, base_poses)])
    return query, chains


def traffic_bytes(batch):
    """HBM bytes per correlate launch from the PMC pass recorded under profiles/ (FETCH_SIZE with the gfx950
    x2 correction of MI355X_MICROARCH.md); only valid for the batch size it was measured at."""
    try:
        t = json.load(open(os.path.join(REPO, "profiles", "traffic_correlate.json")))
        return t["hbm_bytes_per_launch"] if int(t["batch"]) == int(batch) else None
    except Exception:
        return None


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
        while time.perf_counter() - t0 < seconds:
            o.match_raw(qs, arr, True, True)
            n += 1
        dt = time.perf_counter() - t0
        out[label] = dict(hyp_per_s=n * r.hypotheses / dt, matches=n, seconds=dt, threads=threads)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="independent cfg2 matches per step per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--corr-u", type=int, default=0, help="development: beams in flight per lane in the correlate kernel")
    ap.add_argument("--corr-chunks", type=int, default=0, help="development: beam chunks per angle in the correlate kernel")
    ap.add_argument("--corr-pad-lds", type=int, default=0, help="development: extra LDS bytes per correlate block")
    ap.add_argument("--correlate-variant", type=int, default=-1, help="development: force a coarse correlate kernel form")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libyagmatch has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("YM_BENCH_FORCE_DIST"):  # the env var exercises the RCCL path with one rank
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # cpu_baseline leg: no spinning OpenMP workers
    from yag_slam_amd import dist as ymdist
    from yag_slam_amd.scan_matching import ScanMatcher

    m = ScanMatcher(None, device=local_rank)
    stream = torch.cuda.current_stream()
    m.set_stream(stream.cuda_stream)
    if args.correlate_variant >= 0:
        m.debug_option(0, args.correlate_variant)
    if args.corr_u > 0:
        m.debug_option(3, args.corr_u)
    if args.corr_chunks:
        m.debug_option(5, args.corr_chunks)
    if args.corr_pad_lds:
        m.debug_option(4, args.corr_pad_lds)
    query, chains = build_inputs(args.batch, rank)
    batch = m.make_batch(query, chains)
    nslots = 8  # result slots cycled by the pipelined loop (all touched during warm-up)
    records = torch.zeros((nslots, ymdist.RECORD), dtype=torch.float64, device="cuda")
    gathered = torch.zeros((nslots, world * ymdist.RECORD), dtype=torch.float64, device="cuda")
    works = [None] * nslots

    def step(i):
        s = i % nslots
        if i >= nslots:
            batch.wait(s, per_chain=False)  # recycle the slot (long since finished)
        if works[s] is not None:
            works[s].wait()                 # its gathered records are about to be overwritten
        batch.run_async(True, True, slot=s, chain_id_base=rank * args.batch, dev_best_out=records[s].data_ptr())
        if dist is not None:
            # cross-rank arg-max payload: one 64-byte record per rank.  Asynchronous: RCCL's stream waits for this
            # step's record, the launch stream goes straight on to the next step
            works[s] = dist.all_gather_into_tensor(gathered[s], records[s], async_op=True)

    def drain(n):
        for s in range(min(n, nslots)):
            batch.wait(s, per_chain=False)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # correctness of what is timed: first step against the single-call path
    per, best, bi = (batch.run_async(True, True, slot=0) or batch.wait(0))
    hyp_per_match = per[0].meta["hypotheses"]
    hyp_step = sum(p.meta["hypotheses"] for p in per)
    ref = m.match_scan(query, chains[0], True, True)
    assert ref.response == per[0].response and ref.covariance == per[0].covariance

    for i in range(args.warmup):
        step(i)
    drain(args.warmup)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    drain(args.steps)
    if dist is not None:  # the last step's gather carries this rank's own record in its place
        s_last = (args.steps - 1) % nslots
        assert torch.equal(gathered[s_last].view(world, ymdist.RECORD)[rank], records[s_last])
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # roofline of the dominant kernel (coarse correlate): HIP events on the launch stream, second pass
    m.profile(True)
    for i in range(min(args.steps, 50)):
        step(i)
    drain(min(args.steps, 50))
    corr_ms, corr_n = m.profile_read(0)
    call_ms, call_n = m.profile_read(2)
    m.profile(False)

    # the same problem as ONE match_scan call (BASELINE configs[1] as the reference runs it: no batching), for reference
    single = None
    if rank == 0:
        torch.cuda.synchronize()
        n1 = 300
        for _ in range(20):
            m.match_scan(query, chains[0], True, True)
        t1 = time.perf_counter()
        for _ in range(n1):
            m.match_scan(query, chains[0], True, True)
        sync_s = (time.perf_counter() - t1) / n1
        t1 = time.perf_counter()
        for i in range(n1):
            if i >= 8:
                m.wait(i % 8)
            m.match_scan_async(query, chains[0], True, True, slot=i % 8)
        for sl in range(8):
            m.wait(sl)
        pipe_s = (time.perf_counter() - t1) / n1
        single = {"sync_us_per_match": sync_s * 1e6, "pipelined_us_per_match": pipe_s * 1e6,
                  "hypotheses_per_s_sync": hyp_per_match / sync_s, "hypotheses_per_s_pipelined": hyp_per_match / pipe_s}

    if rank == 0:
        nq = per[0].meta["n_query_points"]
        cd = per[0].meta["coarse_dims"]
        coarse_hyp_launch = args.batch * cd[0] * cd[1] * cd[2]
        alg_bytes = coarse_hyp_launch * nq  # 1 grid byte per valid beam per hypothesis
        corr_s = corr_ms / max(corr_n, 1) * 1e-3
        achieved = alg_bytes / corr_s / 1e9
        total_hyp = hyp_step * world * args.steps
        line = {
            "metric": "pose hypotheses/sec",
            "value": total_hyp / dt,
            "unit": "hypotheses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": "cfg2 x batch: %d independent single-match problems per step per GPU (1081-beam query vs "
                            "10-scan chain, search 0.5 m / 0.349 rad, resolution 0.01, coarse 26x26x21 + fine 3x3x11, "
                            "penalty on), Karto semantics" % args.batch,
                "batch_per_gpu": args.batch,
                "hypotheses_per_match": hyp_per_match,
                "scan_matches_per_s": args.batch * world * args.steps / dt,
                "collective": "all_gather of one 64-byte best record per rank per step" if world > 1 else "none",
                "single_match": single,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "ym::correlate_kernel<2, 16>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic_bytes(args.batch),
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_us": corr_s * 1e6,
                "call_us_gpu": call_ms / max(call_n, 1) * 1e3,
            },
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a single-GPU-run item
            cb = cpu_baseline()
            line["cpu_baseline"] = {
                "value": cb["single"]["hyp_per_s"], "unit": "hypotheses/s", "cores": 1, "kind": "port",
                "sample": "%d cfg2 matches (coarse+fine, penalty) in %.1f s, oracle/ym_oracle.c karto semantics, "
                          "-O3 -march=native, 1 thread" % (cb["single"]["matches"], cb["single"]["seconds"]),
                "host": {"cpu_model": cpu_model(), "logical_cpus": os.cpu_count() or 1},
                "all_cores": {"value": cb["all"]["hyp_per_s"], "cores": cb["all"]["threads"],
                              "sample": "%d matches in %.1f s, OpenMP over the coarse lattice (grid clear and rasterisation stay serial, as in Karto), host has %d cores" % (cb["all"]["matches"], cb["all"]["seconds"], os.cpu_count() or 1)},
            }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
