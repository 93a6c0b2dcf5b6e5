#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the bench line, the rocprofv3 kernel-trace summary of the same command and
# the two PMC passes, all under gpurun_out/$TAG/.  Fold afterwards with tools/fold_profiles.sh $TAG.
set -o pipefail
TAG=${1:-r01x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err && echo "bench ok" &&
for c in 3 4 5; do python3 bench.py --config $c --no-cpu-baseline > $OUT/bench_line_cfg$c.json 2>> $OUT/bench.err; done &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 bench.py --no-cpu-baseline --no-analysis-fwd > $OUT/bench_line_under_rocprof.json 2> $OUT/rocprof.err && echo "stats ok" &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-events --no-analysis-fwd > /dev/null 2>> $OUT/rocprof.err && echo "pmc fetch ok" &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-events --no-analysis-fwd > /dev/null 2>> $OUT/rocprof.err && echo "pmc write ok"
find $OUT -name "*.csv" | head -20
