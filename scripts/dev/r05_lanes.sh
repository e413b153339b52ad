cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH_FLAGS="--only-headline --launch-batch 4096 --lanes 4" scripts/dev/r05_ab.sh L4096x4=
BENCH_FLAGS="--only-headline --launch-batch 2048 --lanes 8" scripts/dev/r05_ab.sh L2048x8=
BENCH_FLAGS="--only-headline --launch-batch 2048 --lanes 4" scripts/dev/r05_ab.sh L2048x4=
BENCH_FLAGS="--only-headline --launch-batch 4096 --lanes 4 --batch 32768" scripts/dev/r05_ab.sh B32768=
BENCH_FLAGS="--only-headline --launch-batch 8192 --lanes 2" scripts/dev/r05_ab.sh L8192x2=
