#!/bin/bash
# HAT and PackNet on wide_VGG9 (BASELINE configs[4] model) through the driver at Tiny-ImageNet shapes: 2 tasks, 5-epoch cap
set -u
mkdir -p gpurun_out
: > gpurun_out/sweep_wide2.jsonl
for m in HAT packnet; do
  echo "== $m"
  ( time timeout 420 python tools/sweep.py --tasks 2 --epochs 5 --method $m --model wide_VGG9_cl_512_512 --root /tmp/clhip_sweep_$m 2> gpurun_out/sweep_wide2_$m.err | tail -1 | tee -a gpurun_out/sweep_wide2.jsonl | cut -c1-600 ) 2>&1 | tail -5
  grep -v "amdgpu.ids\|Warning\|warn" gpurun_out/sweep_wide2_$m.err | tail -3
  rm -rf /tmp/clhip_sweep_$m
done
