#!/bin/bash
# Ablation variants (garbage results, timing only): which part of a pipeline costs what.  usage: gpu_abl.sh <outdir> <model> <variant...>
out=$1; model=$2; shift 2
mkdir -p gpurun_out/$out
for v in "" "$@"; do
  echo "== variant ${v:-base}"
  CLHIP_LIB=clsurvey_amd/libclhip${v:+_$v}.so timeout 120 python tools/conv_bench.py $model 200 20 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/$out/abl_${v:-base}_$model.txt
done
