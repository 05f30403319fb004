#!/bin/bash
# PackNet / HAT on wide_VGG9 (BASELINE configs[4] model) through the driver: 2 tasks, 10-epoch cap
set -u
mkdir -p gpurun_out
: > gpurun_out/sweep_wide.jsonl
for m in packnet HAT; do
  echo "== $m"
  timeout 110 python tools/sweep.py --tasks 2 --epochs 10 --method $m --model wide_VGG9_cl_512_512 --root /tmp/clhip_sweep_$m 2> gpurun_out/sweep_wide_$m.err | tail -1 | tee -a gpurun_out/sweep_wide.jsonl | cut -c1-400
  rc=${PIPESTATUS[0]}; [ "$rc" != "0" ] && { echo "rc=$rc"; grep -v "amdgpu.ids\|Warning\|warn" gpurun_out/sweep_wide_$m.err | tail -6; }
  rm -rf /tmp/clhip_sweep_$m
done
