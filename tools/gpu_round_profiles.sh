#!/bin/bash
# Everything profiles/ needs for a round, in one GPU session: per-layer conv timings, MFMA-pipe busy by PMC for the three
# VGG9 widths, HBM traffic of the dominant launch (separate PMC passes), bench + kernel stats.  usage: gpu_round_profiles.sh <tag>
set -u
TAG=${1:-r02}
mkdir -p gpurun_out/$TAG
for m in small base wide; do
  timeout 120 python tools/conv_bench.py $m 200 20 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/conv_layers_$m.txt
  bash tools/gpu_mfma_util.sh ${m}_VGG9_cl_$([ $m = small ] && echo 128_128 || echo 512_512) $TAG/mfma_util_$m > /dev/null 2>&1
done
bash tools/gpu_traffic.sh $TAG/traffic_fwdpool fwdpool 200 64 64 32 5 > gpurun_out/$TAG/traffic_fwdpool.txt 2>&1
bash tools/gpu_traffic.sh $TAG/traffic_dgrad dgrad_unpool 200 64 64 32 5 > gpurun_out/$TAG/traffic_dgrad.txt 2>&1
bash tools/gpu_prof.sh $TAG/bench "--no-cpu-baseline --no-configs --no-sweep" > gpurun_out/$TAG/prof.log 2>&1
find gpurun_out/$TAG -type d -name "*FETCH_SIZE" -o -type d -name "*WRITE_SIZE" | xargs rm -rf
ls gpurun_out/$TAG
