"""Run on the GPU box after scripts/profile_pmc.sh: the rocprofv3 counter CSVs of every workload (too large to travel) -> one small
JSON, gpurun_out/<tag>_counters.json; copied to profiles/counters.json it is what bench.py replays as `roofline.resources` of
the metric line and of the cfg4 / cfg5 sub-blocks.
    python3 scripts/summarise_counters.py r04p
Per workload and kernel: mean of each counter over the LARGEST launches of the kernel (the upper three quarters by value, so
that the small launches of bench.py's self-checks drop out), and the kernel's mean duration from the --kernel-trace pass."""
import collections, csv, glob, json, os, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = {"cfg2x": "python3 bench.py --only cfg2x --no-production-legs --only-headline --lanes 1 --steps 2 --warmup 1",
             "cfg4": "python3 bench.py --only cfg4 --no-production-legs --lanes 1 --steps 2 --warmup 1",
             "cfg5": "python3 bench.py --only cfg5 --lanes 1 --steps 2 --warmup 1"}


def short(name):
    return name.split("(")[0].replace("void ", "")


def main():
    tag = sys.argv[1]
    sys.path.insert(0, REPO)
    from yag_slam_amd import _capi
    out = {"tag": tag, "build_id": _capi.build_id(),  # (the library the passes ran on: bench.py replays the counters only on the same build)
           "how": "rocprofv3 --pmc <one group per pass> (nothing else traced) + one --kernel-trace --stats pass, per workload; "
                              "scripts/profile_pmc.sh; FETCH_SIZE in KiB as reported (x 1024 x 2 = bytes on gfx950, MI355X_MICROARCH.md)",
           "workloads": {}}
    for wl, cmd in WORKLOADS.items():
        kern = collections.defaultdict(lambda: collections.defaultdict(list))
        grids = collections.defaultdict(collections.Counter)
        for f in glob.glob(os.path.join(REPO, "gpurun_out", "%s_%s_*" % (tag, wl), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if "ym::" in k:
                    kern[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    if r.get("Grid_Size"):
                        grids[k][(int(r["Grid_Size"]), int(r.get("Workgroup_Size") or 0))] += 1
        stats = {}
        for f in glob.glob(os.path.join(REPO, "gpurun_out", "%s_%s_trace" % (tag, wl), "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                stats[short(r["Name"])] = {"us": float(r["AverageNs"]) * 1e-3, "calls": int(r["Calls"]), "total_us": float(r["TotalDurationNs"]) * 1e-3,
                                           "min_us": float(r["MinNs"]) * 1e-3, "max_us": float(r["MaxNs"]) * 1e-3}
        ks = {}
        for k in sorted(set(kern) | set(stats)):
            d = dict(stats.get(k, {}))
            for c, v in kern.get(k, {}).items():
                v = sorted(v)
                v = v[len(v) // 4:] if len(v) > 3 else v
                d[c] = sum(v) / len(v)
            if grids.get(k):  # the LARGEST launch of the kernel in the counter passes: work-items and work-group size
                g = max(grids[k])
                d["grid"] = {"work_items": g[0], "workgroup": g[1], "blocks": g[0] // g[1] if g[1] else None}
            ks[k] = d
        if ks:
            out["workloads"][wl] = {"command": cmd, "kernels": ks}
    dst = os.path.join(REPO, "gpurun_out", "%s_counters.json" % tag)
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for wl, w in out["workloads"].items():
        top = sorted(w["kernels"].items(), key=lambda kv: -kv[1].get("total_us", 0.0))[:6]
        print(wl, ", ".join("%s %.0f us x %d" % (k.replace("ym::", ""), v.get("us", 0.0), v.get("calls", 0)) for k, v in top))


if __name__ == "__main__":
    main()
