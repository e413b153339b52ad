#!/bin/bash
# round 5, first GPU pass: the whole GPU suite on the product library, the forms behind -DYM_EXPERIMENTAL on the experimental build,
# the bench line, and the FETCH_SIZE calibration (scripts/exp/fetch_calib.hip) under rocprofv3
tag=${1:-r05a}
out=gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q > $out/${tag}_gputests.log 2>&1; echo "gpu tests rc $?" >> $out/${tag}_gputests.log
tail -5 $out/${tag}_gputests.log
YM_LIB_PATH=$PWD/yag_slam_amd/libyagmatch_exp.so timeout 600 python -m pytest tests -m gpu -x -q -k "region_correlate_equals or cfg2_batch_512" > $out/${tag}_exptests.log 2>&1; echo "exp tests rc $?" >> $out/${tag}_exptests.log
tail -3 $out/${tag}_exptests.log
timeout 900 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; echo "bench rc $?"
tail -c 600 $out/${tag}_bench.err
for w in 0 1 2; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/${tag}_calib_${w}_$ctr
    timeout 120 rocprofv3 --pmc $ctr --output-format csv -d $out/${tag}_calib_${w}_$ctr -o calib -- ./scripts/exp/fetch_calib $w > $out/${tag}_calib_${w}_$ctr.log 2>&1
  done
done
TAG=$tag python3 - <<'PY'
import csv, glob, os
tag = os.environ.get("TAG", "r05a")
for w in (0, 1, 2):
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = "gpurun_out/%s_calib_%d_%s" % (tag, w, ctr)
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            rows = list(csv.DictReader(open(f)))
            vals = [(r["Kernel_Name"][:30], float(r["Counter_Value"])) for r in rows if r.get("Counter_Name") == ctr]
            print(w, ctr, vals)
        print(open("gpurun_out/%s_calib_%d_%s.log" % (tag, w, ctr)).read().strip().splitlines()[-1:])
PY
