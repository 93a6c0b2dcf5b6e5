#!/bin/bash
# ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE passes over tools/bench_halo.py (the 128-channel 5x5 s2 layer, halo kernel vs
# implicit GEMM) -> gpurun_out/$1/halo_pmc.json
set -o pipefail
TAG=${1:-r03x}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/hp_f -o f -- python3 tools/bench_halo.py 128 > $OUT/halo_pmc_run.txt 2> $OUT/halo_pmc.err &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/hp_w -o w -- python3 tools/bench_halo.py 128 >> $OUT/halo_pmc_run.txt 2>> $OUT/halo_pmc.err &&
python3 tools/pmc_traffic.py $(find $OUT/hp_f -name "*counter_collection.csv") $(find $OUT/hp_w -name "*counter_collection.csv") $OUT/halo_pmc.json &&
python3 -c "
import json,sys
d=json.load(open('$OUT/halo_pmc.json'))['kernels']
for k,v in d.items():
    if 'halo' in k or 'igemm_bf16' in k: print(k, v)
"
