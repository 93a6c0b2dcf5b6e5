#!/bin/bash
# After `gpurun -- bash tools/refresh_profiles.sh TAG`: copy the judged summaries from gpurun_out/TAG into profiles/.
set -e
TAG=${1:?tag}
S=gpurun_out/$TAG
cp $S/bench_line.json profiles/${TAG}_bench_line.json
for c in 3 4 5; do cp $S/bench_line_cfg$c.json profiles/${TAG}_bench_line_cfg$c.json; done
cp $S/bench_line_under_rocprof.json profiles/${TAG}_bench_line_under_rocprof.json
cp $S/stats/s_kernel_stats.csv profiles/${TAG}_kernel_stats_bench_cfg2.csv
python tools/pmc_traffic.py $S/pmc_f/f_counter_collection.csv $S/pmc_w/w_counter_collection.csv profiles/${TAG}_pmc_traffic.json
ls -la profiles/${TAG}_*
