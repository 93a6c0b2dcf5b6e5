#!/bin/bash
# same-box A/B of two builds of liblic_hip.so: tools/_ab/liblic_hip_old.so against the in-tree one
set -o pipefail
L=neural_image_compression_amd/liblic_hip.so
cp $L /tmp/new.so
CFG=${1:-3}
for i in 1 2; do
  cp /tmp/new.so $L && python3 bench.py --config $CFG --no-cpu-baseline --no-analysis-fwd 2>/dev/null >> gpurun_out/ab_new.txt &&
  cp tools/_ab/liblic_hip_old.so $L && python3 bench.py --config $CFG --no-cpu-baseline --no-analysis-fwd 2>/dev/null >> gpurun_out/ab_old.txt || exit 1
done
cp /tmp/new.so $L
