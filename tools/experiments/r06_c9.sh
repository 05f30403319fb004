#!/bin/bash
# round 6, third session, call 9: the weight-image launches of a pass on a stream of their own beside the first layer's forward
# (CLHIP_PREP_SIDE=1, the default) against the main stream (=0): engine parity, step A/B, AlexNet step A/B
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modes.py tests/test_gpu_switches.py tests/test_gpu_fc_tail.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee gpurun_out/r06c9_tests.log
B="bench.py --no-cpu-baseline --no-configs --no-sweep --steps 200 --warmup 20"
for rep in 1 2 3; do
for side in 1 0; do
  echo -n "CLHIP_PREP_SIDE=$side  "
  CLHIP_PREP_SIDE=$side timeout 300 python $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ms %.4f' % d['ms_per_step'])"
done
done 2>&1 | tee gpurun_out/r06c9_ab.txt
for side in 1 0 1 0; do echo -n "CLHIP_PREP_SIDE=$side  "; CLHIP_PREP_SIDE=$side timeout 200 python tools/alexnet_step.py 128 2>&1 | grep -v amdgpu.ids | tail -1; done | tee -a gpurun_out/r06c9_ab.txt
