#!/bin/bash
# After `gpurun -- bash tools/refresh_profiles.sh TAG`: copy the judged summaries from gpurun_out/TAG into profiles/.
set -e
TAG=${1:?tag}
S=gpurun_out/$TAG
cp $S/bench_line.json profiles/${TAG}_bench_line.json
for c in 3 4 5 2h 3k hmr hmrh; do cp $S/bench_line_cfg$c.json profiles/${TAG}_bench_line_cfg$c.json; done
cp $S/bench_line_under_rocprof.json profiles/${TAG}_bench_line_under_rocprof.json
cp $S/stats/s_kernel_stats.csv profiles/${TAG}_kernel_stats_bench_cfg2.csv
python tools/pmc_traffic.py $S/pmc_f/f_counter_collection.csv $S/pmc_w/w_counter_collection.csv profiles/${TAG}_pmc_traffic.json
cp $S/stats3/s_kernel_stats.csv profiles/${TAG}_kernel_stats_bench_cfg3.csv
cp $S/bench_line_cfg3_under_rocprof.json profiles/${TAG}_bench_line_cfg3_under_rocprof.json
python tools/pmc_traffic.py $S/pmc3_f/f_counter_collection.csv $S/pmc3_w/w_counter_collection.csv profiles/${TAG}_pmc_traffic_cfg3.json
cp $S/fwd3/f_kernel_stats.csv profiles/${TAG}_kernel_stats_fwd_cfg3.csv
python tools/fwd_kernels.py $S/fwd3/f_kernel_trace.csv stem_gdn > profiles/${TAG}_fwd_timeline_cfg3.txt
cat $S/fwd3.txt >> profiles/${TAG}_fwd_timeline_cfg3.txt
python tools/pmc_traffic.py $S/fwd3_f/f_counter_collection.csv $S/fwd3_w/w_counter_collection.csv profiles/${TAG}_pmc_traffic_fwd_cfg3.json
ls -la profiles/${TAG}_*
