#!/bin/bash
# Runs ON THE GPU BOX (gpurun): the bench line, the rocprofv3 kernel-trace summary of the same command and
# the two PMC passes, all under gpurun_out/$TAG/.  Fold afterwards with tools/fold_profiles.sh $TAG.
set -o pipefail
TAG=${1:-r01x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench.err && echo "bench ok" &&
for c in 3 4 5 2h 3k hmr hmrh; do python3 bench.py --config $c --no-cpu-baseline > $OUT/bench_line_cfg$c.json 2>> $OUT/bench.err; done &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 bench.py --no-cpu-baseline --no-analysis-fwd > $OUT/bench_line_under_rocprof.json 2> $OUT/rocprof.err && echo "stats ok" &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-events --no-analysis-fwd > /dev/null 2>> $OUT/rocprof.err && echo "pmc fetch ok" &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-events --no-analysis-fwd > /dev/null 2>> $OUT/rocprof.err && echo "pmc write ok"
# config 3 (bf16 storage): the same three passes, plus the kernel trace of its analysis + hyperprior forward alone
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats3 -o s -- python3 bench.py --config 3 --no-cpu-baseline --no-analysis-fwd > $OUT/bench_line_cfg3_under_rocprof.json 2>> $OUT/rocprof.err && echo "cfg3 stats ok" &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc3_f -o f -- python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-profile-events --no-analysis-fwd > /dev/null 2>> $OUT/rocprof.err && echo "cfg3 pmc fetch ok" &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc3_w -o w -- python3 bench.py --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-profile-events --no-analysis-fwd > /dev/null 2>> $OUT/rocprof.err && echo "cfg3 pmc write ok" &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fwd3 -o f -- python3 tools/trace_fwd.py 3 20 > $OUT/fwd3.txt 2>> $OUT/rocprof.err && echo "cfg3 fwd trace ok" &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fwd3_f -o f -- python3 tools/trace_fwd.py 3 3 > /dev/null 2>> $OUT/rocprof.err &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/fwd3_w -o w -- python3 tools/trace_fwd.py 3 3 > /dev/null 2>> $OUT/rocprof.err && echo "cfg3 fwd pmc ok"
find $OUT -name "*.csv" | head -40
