#!/bin/bash
# Round 5, GPU session L: 5 x 5 taps on the bf16-split kernel (AlexNet conv2), the 28 x 28 rule; AlexNet step before / after.
set -u
mkdir -p gpurun_out/r05l; export TMPDIR=/tmp
O=gpurun_out/r05l
timeout 900 python -m pytest tests/test_gpu_bs.py -m gpu -x -q -p no:cacheprovider -k "5x5 or weight_image or refuses" > $O/test_bs5.txt 2>&1; echo "tests rc $?"; tail -4 $O/test_bs5.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py -m gpu -x -q -p no:cacheprovider -k "alexnet or conv2d" > $O/test_alex.txt 2>&1; echo "alex tests rc $?"; tail -4 $O/test_alex.txt
timeout 200 python tools/alexnet_step.py 128 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/alexnet_bs.txt
CLHIP_BS=0 timeout 200 python tools/alexnet_step.py 128 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/alexnet_bs0.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05l/conv5.txt
import sys, torch
sys.path.insert(0, ".")
from clsurvey_amd import ops
from tools.bs_bench import timed
x = torch.randn(128, 64, 27, 27, device="cuda").relu_(); w = torch.randn(192, 64, 5, 5, device="cuda") * 0.03; b = torch.zeros(192, device="cuda")
dy = torch.randn(128, 192, 27, 27, device="cuda")
print("conv 64->192 5x5 @27x27 N=128: f32 halo kernel fwd %.1f bwd %.1f us; bf16-split fwd %.1f bwd %.1f us; floor(bs) %.1f" % (
    timed(lambda: ops.conv2d_fwd(x, w, b, 1, 2, True)), timed(lambda: ops.conv2d_bwd_data(dy, w, (128, 64, 27, 27), 1, 2, x)),
    timed(lambda: ops.conv5x5_bs_fwd(x, w, b, True)), timed(lambda: ops.conv5x5_bs_bwd_data(dy, w, x)), 2.0 * 25 * 64 * 192 * 729 * 128 * 6 / 2.5e15 * 1e6))
PY
