#!/bin/bash
mkdir -p gpurun_out/l1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_conv_relu_pool or full_size or unpool or wgrad" 2>&1 | tail -3
python tools/l1_dump.py /tmp/new.pt; CLHIP_LIB=clsurvey_amd/libclhip_old3.so python tools/l1_dump.py /tmp/old.pt
python - <<PY
import torch
a, b = torch.load("/tmp/new.pt"), torch.load("/tmp/old.pt")
for k in a:
    print(k, "values equal", torch.equal(a[k][0], b[k][0]), "idx equal", torch.equal(a[k][1], b[k][1]), "dw equal", torch.equal(a[k][2], b[k][2]), "db equal", torch.equal(a[k][3], b[k][3]), float(a[k][2].abs().max()))
PY
for v in "" "$@"; do
  echo "== ${v:-base}"
  CLHIP_LIB=clsurvey_amd/libclhip${v:+_$v}.so timeout 120 python tools/conv_bench.py small 200 20 2>&1 | grep "3x64"
done
