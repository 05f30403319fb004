#!/bin/bash
# round 6, third session, call 6: output stores write-through at agent scope (sc1; CLHIP_ST_AUX=16, the build) against the default policy
# (libclhip_st0.so: conv3x3.hip, bsconv.hip, bswgrad.hip with -DCLHIP_ST_AUX=0): parity of the touched kernels, per-launch and step timing
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bs.py tests/test_gpu_bswgrad.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r06c6_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "fused or pool or engine_matches or full_size" 2>&1 | tail -3 | tee -a gpurun_out/r06c6_tests.log
for rep in 1 2; do
for lib in libclhip.so libclhip_st0.so; do
  echo "== $lib"
  CLHIP_LIB=$PWD/clsurvey_amd/$lib timeout 300 python tools/conv_bench.py small 200 20 2>&1 | grep -v amdgpu.ids | grep -E "3x64@64|64x64@32|TOTAL|ALL"
  CLHIP_LIB=$PWD/clsurvey_amd/$lib timeout 300 python bench.py --no-cpu-baseline --no-configs --no-sweep --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'], 'img/s', d['value'])"
done
done 2>&1 | tee gpurun_out/r06c6_ab.txt
