#!/bin/bash
set -u
mkdir -p gpurun_out/r05k; export TMPDIR=/tmp
python tools/bs_shapes.py alex wide224 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05k/bs_shapes.txt
