#!/bin/bash
# BASELINE configs[4] end to end at its real input size: HAT and PackNet on wide_VGG9_cl_512_512 through the driver, 2 tasks of
# 3x224x224 images (iNaturalist geometry, data/dataset.py:97), 5-value LR grid, bounded epochs, --test.
set -u
mkdir -p gpurun_out
: > gpurun_out/sweep_wide224.jsonl
for m in HAT packnet; do
  echo "== $m"
  ( time timeout 600 python tools/sweep.py --tasks 2 --epochs 4 --method $m --model wide_VGG9_cl_512_512 --hw 224 --sizes 2000,400,400 --batch 50 --friendly-init --root /tmp/clhip_sweep_$m 2> gpurun_out/sweep_wide224_$m.err | tail -1 | tee -a gpurun_out/sweep_wide224.jsonl | cut -c1-700 ) 2>&1 | tail -5
  grep -v "amdgpu.ids\|Warning\|warn" gpurun_out/sweep_wide224_$m.err | tail -4
  rm -rf /tmp/clhip_sweep_$m
done
