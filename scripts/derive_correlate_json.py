"""profiles/counters.json + profiles/issue_peaks.json -> profiles/issue_correlate.json and profiles/traffic_correlate.json: what binds the
dominant kernel of the bench step (correlate_region_kernel), with the settled issue peak and every other measurement of it
beside it (the round-3 review's item 1c).  bench.py itself reads counters.json; these two files are the human-readable digest."""
import json, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda n: os.path.join(REPO, "profiles", n)
ctr, peaks = json.load(open(P("counters.json"))), json.load(open(P("issue_peaks.json")))
K = "ym::correlate_region_kernel<8, true>"
k = ctr["workloads"]["cfg2x"]["kernels"][K]
us = min(k["min_us"] * 1.02, k["us"])  # (the mean includes the launches of the bench's self-check; the steady launches sit at the minimum)
clk = us * 1e-6 * 2.4e9 * 256
body = peaks["region_correlate_body"]
json.dump({"kernel": K, "batch": 4096, "kernel_us_under_rocprof": us,
           "valu": {"counter": "SQ_INSTS_VALU", "per_launch": k["SQ_INSTS_VALU"], "per_cu_clk": k["SQ_INSTS_VALU"] / clk,
                    "peak_per_cu_clk": body["peak_per_cu_clk"], "frac": k["SQ_INSTS_VALU"] / clk / body["peak_per_cu_clk"],
                    "peak_source": "profiles/issue_peaks.json: " + body["what"], "peak_by_waves_per_simd": body["by_waves_per_simd"],
                    "other_measurements_of_the_peak": body["other_measurements"],
                    "frac_against_round_3_peak_0.961": k["SQ_INSTS_VALU"] / clk / 0.961},
           "salu_per_cu_clk": k["SQ_INSTS_SALU"] / clk,
           "lds": {"counter": "SQ_LDS_IDX_ACTIVE", "cycles_per_launch": k["SQ_LDS_IDX_ACTIVE"], "bank_conflict_cycles": k["SQ_LDS_BANK_CONFLICT"],
                   "frac": k["SQ_LDS_IDX_ACTIVE"] / clk},
           "vector_l1": {"counter": "TCP_TOTAL_CACHE_ACCESSES_sum", "line_visits_per_launch": k["TCP_TOTAL_CACHE_ACCESSES_sum"],
                         "frac": k["TCP_TOTAL_CACHE_ACCESSES_sum"] / clk},
           "wave_time": {"waiting_at_waitcnt_or_barrier": k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"], "issue_stalled": k["SQ_WAIT_INST_ANY"] / k["SQ_WAVE_CYCLES"],
                         "issuing": k["SQ_ACTIVE_INST_ANY"] / k["SQ_WAVE_CYCLES"]},
           "source": "profiles/counters.json (%s)" % ctr["tag"]}, open(P("issue_correlate.json"), "w"), indent=1)
json.dump({"kernel": K, "batch": 4096, "fetch_size_kib_per_launch": k["FETCH_SIZE"], "gfx950_wide_read_correction": 2.0,
           "hbm_read_bytes_per_launch": k["FETCH_SIZE"] * 2048.0, "hbm_written_bytes_per_launch": k["WRITE_SIZE"] * 1024.0,
           "hbm_bytes_per_launch": k["FETCH_SIZE"] * 2048.0 + k["WRITE_SIZE"] * 1024.0,
           "l2_hit_rate": k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]), "source": "profiles/counters.json (%s)" % ctr["tag"]},
          open(P("traffic_correlate.json"), "w"), indent=1)
print(open(P("issue_correlate.json")).read()[:900])
