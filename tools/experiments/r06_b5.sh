#!/bin/bash
# round 6, second session, call 5: conv3x3_wgrad_c3_unpool_kernel with the pooled gradient + codes through a wave-private LDS image
# (main) against the build with direct per-channel loads (u3nolds) and the build before this session (u3old): bitwise, parity, timing
set -u
mkdir -p gpurun_out/r06b5; export TMPDIR=/tmp
O=gpurun_out/r06b5; P=$PWD
for v in u3old u3nolds main; do
  L=clsurvey_amd/libclhip_$v.so; [ $v = main ] && L=clsurvey_amd/libclhip.so
  echo "== $v"
  CLHIP_LIB=$L python tools/experiments/u3_dump.py $O/dump_$v.npz 2>&1 | grep -v amdgpu.ids | head -2
  CLHIP_LIB=$L timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "wgrad or engine_matches or full_size" 2>&1 | tail -1
  for i in 1 2 3; do CLHIP_LIB=$L timeout 120 python tools/conv_bench.py small 200 20 2>&1 | grep -E "bwd_weight_unpool +3x64@64|^ALL" | tr '\n' ' '; echo; done
done
python - <<'PY'
import numpy as np
a=np.load("gpurun_out/r06b5/dump_u3old.npz"); b=np.load("gpurun_out/r06b5/dump_main.npz"); c=np.load("gpurun_out/r06b5/dump_u3nolds.npz")
print("main bitwise u3old:", all(np.array_equal(a[k], b[k]) for k in a.files))
print("u3nolds bitwise u3old:", all(np.array_equal(a[k], c[k]) for k in a.files))
PY
