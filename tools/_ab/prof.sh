#!/bin/bash
# kernel-trace stats of a config-3 bench run with the in-tree library -> gpurun_out/abprof/
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/abprof && mkdir -p gpurun_out/abprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abprof -o s -- python3 bench.py --config 3 --no-cpu-baseline --no-analysis-fwd > gpurun_out/abprof/line.json 2> gpurun_out/abprof/err.txt
