timeout 900 python -m pytest tests -m gpu -x -q -k "region_correlate or cfg2_batch_512 or match_pairs or large_batch" 2>&1 | tail -5
scripts/dev/r05_ab.sh base= h128=32:5,43:128 h100=32:5,43:100 h80=32:5,43:80 base2=
