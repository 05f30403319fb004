#!/bin/bash
# 3-task sweeps at BASELINE.json's full shapes for the other configs (through the build's driver, reference defaults)
set -u
mkdir -p gpurun_out
: > gpurun_out/sweep_methods.jsonl
run() {  # method model
  echo "== $1 $2"
  timeout ${3:-170} python tools/sweep.py --tasks 3 --method "$1" --model "$2" --root /tmp/clhip_sweep_$1 --cpu-train-rate 0 2> gpurun_out/sweep_$1.err | tail -1 | tee -a gpurun_out/sweep_methods.jsonl | cut -c1-420
  rc=${PIPESTATUS[0]}; [ "$rc" != "0" ] && { echo "rc=$rc"; tail -5 gpurun_out/sweep_$1.err; }
  rm -rf /tmp/clhip_sweep_$1
}
run MAS base_VGG9_cl_512_512
run SI base_VGG9_cl_512_512
run GEM small_VGG9_cl_128_128
run packnet wide_VGG9_cl_512_512
run HAT wide_VGG9_cl_512_512
