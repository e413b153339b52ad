"""Turn the rocprofv3 outputs of scripts/profile_round.sh / profile_pmc.sh (merged under gpurun_out/) into the small
summaries kept under profiles/.    python scripts/summarise_profiles.py r02p 1024"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CU, CLOCK_HZ = 256, 2.4e9


def kernel_means(tag, name):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in (glob.glob(os.path.join(REPO, "gpurun_out", "%s_%s" % (tag, name), "**", "*counter_collection.csv"), recursive=True) +
              glob.glob(os.path.join(REPO, "gpurun_out", "%s_cfg2x_%s" % (tag, name), "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "ym::" in k:
                out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, d in out.items():
        res[k] = {}
        for c, v in d.items():
            v = sorted(v)
            v = v[len(v) // 4:] if len(v) > 3 else v  # drop the small single-match launches of bench.py's self-check
            res[k][c] = sum(v) / len(v)
    return res


def main():
    tag, batch = sys.argv[1], int(sys.argv[2])
    short = tag[:3] + "_" + tag[3:] if len(tag) == 4 else tag
    # (on the GPU box the counter CSVs are hundreds of MB: summarise there into a directory under gpurun_out/ and copy that)
    prof = os.path.join(REPO, sys.argv[3]) if len(sys.argv) > 3 else os.path.join(REPO, "profiles")
    os.makedirs(prof, exist_ok=True)
    for src, dst in (("stats", "bench_kernel_stats"), ("single", "single_match_kernel_stats"), ("stress", "stress_kernel_stats")):
        f = os.path.join(REPO, "gpurun_out", "%s_%s" % (tag, src), "%s_kernel_stats.csv" % tag)
        if os.path.exists(f):
            rows = [r for r in csv.reader(open(f))]
            with open(os.path.join(prof, "%s_%s.csv" % (short, dst)), "w") as o:
                w = csv.writer(o)
                for r in rows:
                    if r and ("ym::" in r[0] or r[0] == "Name"):
                        w.writerow(r)
    for src in ("bench.json", "bench_under_rocprof.json"):
        f = os.path.join(REPO, "gpurun_out", "%s_%s" % (tag, src))
        if os.path.exists(f) and os.path.getsize(f) > 0:
            shutil.copy(f, os.path.join(prof, "%s_%s" % (short, src.replace("bench.json", "bench_line.json"))))
    fetch, l2, tcp, sq = kernel_means(tag, "fetch"), kernel_means(tag, "l2"), kernel_means(tag, "tcp"), kernel_means(tag, "sq")
    sq2 = kernel_means(tag, "sq2")
    stats = {}
    f = os.path.join(REPO, "gpurun_out", "%s_stats" % tag, "%s_kernel_stats.csv" % tag)
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            stats[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"])
    rows = []
    W2 = ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU", "SQ_WAVES")
    if sq2:
        with open(os.path.join(prof, "%s_pmc_waves_batch%d.csv" % (short, batch)), "w") as o:
            o.write("# rocprofv3 --pmc %s (one pass, nothing else traced): python3 bench.py --only cfg2x --no-production-legs --steps 2 --warmup 1  (launch batch %d)\n" % (" ".join(W2), batch))
            o.write("# means per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles of wave time (MI355X_MICROARCH.md): "
                    "waiting (s_waitcnt, barrier) + issue stalls + issuing ~ wave cycles\n")
            o.write("kernel," + ",".join(W2) + ",wait_any_share,wait_inst_share,active_share\n")
            for k in sorted(sq2):
                d = sq2[k]
                wc = max(d.get("SQ_WAVE_CYCLES", 0.0), 1.0)
                o.write("%s,%s,%.3f,%.3f,%.3f\n" % (k, ",".join("%.0f" % d.get(c, 0.0) for c in W2), d.get("SQ_WAIT_ANY", 0.0) / wc,
                                                   d.get("SQ_WAIT_INST_ANY", 0.0) / wc, d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc))
    for k in sorted(set(fetch) | set(l2) | set(tcp) | set(sq)):
        rows.append((k, fetch.get(k, {}).get("FETCH_SIZE", 0.0), l2.get(k, {}).get("TCC_HIT_sum", 0.0), l2.get(k, {}).get("TCC_MISS_sum", 0.0),
                     tcp.get(k, {}).get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0), tcp.get(k, {}).get("TCP_TOTAL_ACCESSES_sum", 0.0),
                     sq.get(k, {}).get("SQ_INSTS_VALU", 0.0), sq.get(k, {}).get("SQ_INSTS_LDS", 0.0),
                     sq.get(k, {}).get("SQ_LDS_IDX_ACTIVE", 0.0), sq.get(k, {}).get("SQ_LDS_BANK_CONFLICT", 0.0)))
    pmc_csv = os.path.join(prof, "%s_pmc_batch%d.csv" % (short, batch))
    with open(pmc_csv, "w") as o:
        o.write("# rocprofv3 --pmc FETCH_SIZE | --pmc TCC_HIT_sum TCC_MISS_sum | --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum | "
                "--pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT (four separate passes, nothing else traced): "
                "python3 bench.py --only cfg2x --no-production-legs --steps 2 --warmup 1  (launch batch %d)\n" % batch)
        o.write("# means per launch.  FETCH_SIZE in KiB as reported; gfx950 counts half of a 16-B/lane stream (MI355X_MICROARCH.md): bytes ~= 2*1024*FETCH_SIZE\n")
        o.write("kernel,FETCH_SIZE_KiB,TCC_HIT_sum,TCC_MISS_sum,l2_hit_rate,TCP_TOTAL_CACHE_ACCESSES_sum,TCP_TOTAL_ACCESSES_sum,"
                "SQ_INSTS_VALU,SQ_INSTS_LDS,SQ_LDS_IDX_ACTIVE,SQ_LDS_BANK_CONFLICT\n")
        for k, fz, h, m, ca, ta, iv, il, la, lc in rows:
            o.write("%s,%.1f,%.0f,%.0f,%.3f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f\n" % (k, fz, h, m, h / max(h + m, 1.0), ca, ta, iv, il, la, lc))
    # the dominant kernel of the bench step: the region-staged correlate on batches (else the direct one)
    dom = [r for r in rows if "correlate_region_kernel" in r[0]] or [r for r in rows if "correlate_kernel<2, 16, 4>" in r[0]]
    if dom:
        k, fz, h, m, ca, ta, iv, il, la, lc = dom[0]
        json.dump({"kernel": k, "batch": batch, "fetch_size_kib_per_launch": fz, "gfx950_wide_read_correction": 2.0,
                   "hbm_bytes_per_launch": fz * 1024 * 2.0, "l2_hit_rate": h / max(h + m, 1.0), "source": "profiles/" + os.path.basename(pmc_csv)},
                  open(os.path.join(prof, "traffic_correlate.json"), "w"), indent=1)
        dur = stats.get(k)
        if dur and "region" in k and iv:
            clk = dur * 1e-9 * CLOCK_HZ
            peaks = json.load(open(os.path.join(REPO, "profiles", "issue_peaks.json")))["region_correlate_body"]
            json.dump({"kernel": k, "batch": batch, "kernel_us_under_rocprof": dur * 1e-3,
                       "valu": {"counter": "SQ_INSTS_VALU", "per_launch": iv, "per_cu_clk": iv / (CU * clk),
                                "peak_per_cu_clk": peaks["peak_per_cu_clk"], "frac": iv / (CU * clk) / peaks["peak_per_cu_clk"],
                                "peak_source": "profiles/issue_peaks.json (round 4): " + peaks["what"],
                                "other_measurements_of_the_peak": peaks["other_measurements"],
                                "frac_against_round_3_peak_0.96": iv / (CU * clk) / 0.96},
                       "waves": sq2.get(k),
                       "lds": {"counter": "SQ_LDS_IDX_ACTIVE", "cycles_per_launch": la, "bank_conflict_cycles": lc,
                               "frac": la / (CU * clk), "peak_source": "LDS busy cycles / (256 CUs x kernel clocks)"},
                       "source": "profiles/" + os.path.basename(pmc_csv)},
                      open(os.path.join(prof, "issue_correlate.json"), "w"), indent=1)
        elif dur and ca:
            per_cu_clk = ca / (CU * dur * 1e-9 * CLOCK_HZ)
            json.dump({"kernel": k, "batch": batch, "counter": "TCP_TOTAL_CACHE_ACCESSES_sum", "per_launch": ca,
                       "kernel_us_under_rocprof": dur * 1e-3, "per_cu_clk": per_cu_clk,
                       "peak_per_cu_clk": 1.0, "frac": per_cu_clk / 1.0,
                       "peak_source": "profiles/r02_ta_coalescing_experiment.md: one cache-line visit per clock per CU "
                                      "(64 lanes in 64 lines = 65 clk per wave load)",
                       "source": "profiles/" + os.path.basename(pmc_csv)},
                      open(os.path.join(prof, "l1_correlate.json"), "w"), indent=1)
    print(open(pmc_csv).read())
    for n in ("traffic_correlate.json", "issue_correlate.json"):
        if os.path.exists(os.path.join(prof, n)):
            print(n, open(os.path.join(prof, n)).read())


if __name__ == "__main__":
    main()
